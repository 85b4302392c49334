# usage: bash tools/prof_dbg.sh "<debug values>" <kernel substring>
for d in $1; do
export MOD_DEBUG=$d
out=/root/repo/gpurun_out/pd$d
mkdir -p $out && cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 /root/repo/bench.py --steps 5 --warmup 1 --no-cpu-baseline --distinct 4 > $out/bench.json 2> $out/err.log
python3 - <<PY
import csv,glob
f=glob.glob('$out/*/*_kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if '$2' in r['Name']: print("debug=$d %-40s avg_us %10.2f"%(r['Name'][:40], float(r['AverageNs'])/1e3))
PY
done
