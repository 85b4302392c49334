# usage (GPU box): bash tools/prof_sgm.sh <tag>   — rocprofv3 kernel stats of tools/time_sgm.py (8 paths then 4 paths, 16 frames, 1280x720)
tag=$1
out=$PWD/gpurun_out/$tag
repo=$PWD
mkdir -p $out && cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $repo/tools/time_sgm.py > $out/time.log 2> $out/err.log
echo rc=$?
python3 - <<PY
import csv,glob
f=glob.glob('$out/*/*_kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if float(r['Percentage'])>0.3: print("%-70s calls %5s avg_us %10.2f  %5s%%"%(r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
PY
tail -2 $out/time.log
