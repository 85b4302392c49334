out=/root/repo/gpurun_out/hs
mkdir -p $out && cd /tmp && export TMPDIR=/tmp
ONLY=1 timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out -- python3 /root/repo/tools/time_host_stream.py > $out/log.txt 2>&1
echo rc=$?
tail -2 $out/log.txt
python3 - <<PY
import csv,glob,os
d=max(glob.glob('$out/*/'), key=os.path.getmtime)
k=list(csv.DictReader(open(glob.glob(d+'*kernel_trace.csv')[0])))
m=list(csv.DictReader(open(glob.glob(d+'*memory_copy_trace.csv')[0])))
print(m[0].keys())
ev=[]
for r in k: ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'K '+r['Kernel_Name'][:30]))
for r in m: ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'M '+r.get('Direction','')+' '+r.get('Name','')[:20]))
ev.sort()
t0=ev[len(ev)//2][0]
for s,e,n in ev[len(ev)//2: len(ev)//2+60]: print("%9.1f %9.1f us  %s"%((s-t0)/1e3,(e-t0)/1e3,n))
PY
