# usage: bash tools/pmc_sgm.sh <tag> "<counters>"   — SQ counters of the disparity estimator's kernels (tools/time_sgm.py), averaged per dispatch
tag=$1; ctrs=$2
out=/root/repo/gpurun_out/$tag
mkdir -p $out && cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out -- python3 /root/repo/tools/time_sgm.py > $out/time.log 2> $out/err.log
echo rc=$?
python3 - <<PY
import csv,glob,collections
fs=glob.glob('$out/*/*counter_collection.csv')
agg=collections.defaultdict(lambda: collections.defaultdict(lambda: [0,0.0]))
for f in fs:
    for r in csv.DictReader(open(f)):
        kn=r['Kernel_Name']
        if 'sgm' not in kn: continue
        kn=kn.split('(')[1] if kn.startswith('(') else kn
        kn=kn[kn.find('k_sgm'):][:34]
        a=agg[kn][r['Counter_Name']]; a[0]+=1; a[1]+=float(r['Counter_Value'])
for kn in sorted(agg):
    print(kn, ' '.join("%s=%.3gM"%(k,v/n/1e6) for k,(n,v) in sorted(agg[kn].items())))
PY
