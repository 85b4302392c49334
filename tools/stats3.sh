repo=$PWD
cd /tmp && export TMPDIR=/tmp
for i in 1 2 3; do
  out=$repo/gpurun_out/final_stats$i; rm -rf $out; mkdir -p $out
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $repo/bench.py --no-cpu-baseline --no-config5 --no-latency > $out/bench.json 2> $out/err.log
  python3 - <<PY
import csv,glob,json
f=glob.glob('$out/*/*_kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'k_scene_flow' in r['Name']: print('$i', r['Calls'], float(r['AverageNs'])/1e3)
j=json.load(open('$out/bench.json')); print('  bench events ms', j['roofline']['avg_launch_ms'], j['value'])
PY
done
