# HBM traffic per launch (separate --pmc passes as the microarch guide prescribes): FETCH_SIZE, WRITE_SIZE in KiB-ish units
for ctr in FETCH_SIZE WRITE_SIZE; do
out=/root/repo/gpurun_out/traffic_$ctr
mkdir -p $out && cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out -- python3 /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/bench.json 2> $out/err.log
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda:[0,0.0])
for f in glob.glob('$out/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0][-40:]; agg[k][0]+=1; agg[k][1]+=float(r['Counter_Value'])
for k,(n,v) in sorted(agg.items(), key=lambda kv:-kv[1][1])[:8]: print("$ctr %-42s dispatches %3d avg %14.1f"%(k,n,v/n))
PY
done
