# usage: bash tools/pmc.sh <tag> "<counters>" <kernel substring>
tag=$1; ctrs=$2; kn=$3
out=/root/repo/gpurun_out/$tag
mkdir -p $out && cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out -- python3 /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --distinct 4 ${PMC_ARGS:-} > $out/bench.json 2> $out/err.log
echo rc=$?
python3 - <<PY
import csv,glob,collections
fs=glob.glob('$out/*/*counter_collection.csv')
agg=collections.defaultdict(lambda: [0,0.0])
for f in fs:
    for r in csv.DictReader(open(f)):
        if '$kn' in r['Kernel_Name']:
            k=r['Counter_Name']; agg[k][0]+=1; agg[k][1]+=float(r['Counter_Value'])
for k,(n,v) in sorted(agg.items()): print("%-28s dispatches %3d  avg %16.1f"%(k,n,v/n))
PY
