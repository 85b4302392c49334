# usage: bash tools/prof.sh <tag> [bench args]; writes gpurun_out/<tag>/ (kernel stats csv + bench json)
tag=$1; shift
out=/root/repo/gpurun_out/$tag
mkdir -p $out && cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 /root/repo/bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" > $out/bench.json 2> $out/err.log
echo rc=$?
python3 - <<PY
import csv,glob
f=glob.glob('$out/*/*_kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if float(r['Percentage'])>0.5: print("%-60s calls %5s avg_us %10.2f  %5s%%"%(r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
PY
