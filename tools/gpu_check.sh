#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/gpu_check.sh <tag> [soak seconds] — GPU test suite, soak, bench on both workloads;
# everything goes to gpurun_out/<tag>_*; prints a short summary.  Steps are chained: a failing step stops the run.
tag=$1; soak=${2:-60}
python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_tests.log 2>&1; rc=$?
tail -3 gpurun_out/${tag}_tests.log
[ $rc -ne 0 ] && { grep -E "^E|Error|FAILED" gpurun_out/${tag}_tests.log | head -20; exit 1; }
if [ "$soak" != "0" ]; then
  timeout -k 10 $((soak + 120)) python tools/soak.py $soak $RANDOM > gpurun_out/${tag}_soak.log 2>&1; rc=$?
  tail -1 gpurun_out/${tag}_soak.log
  [ $rc -ne 0 ] && { grep MISMATCH gpurun_out/${tag}_soak.log; exit 1; }
fi
python bench.py --no-cpu-baseline > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || { tail -5 gpurun_out/${tag}_bench.err; exit 1; }
python bench.py --workload pairs --no-cpu-baseline > gpurun_out/${tag}_bench_pairs.json 2> gpurun_out/${tag}_bench_pairs.err || { tail -5 gpurun_out/${tag}_bench_pairs.err; exit 1; }
python bench.py --no-labels --no-cpu-baseline > gpurun_out/${tag}_bench_nolabels.json 2> gpurun_out/${tag}_bench_nolabels.err || { tail -5 gpurun_out/${tag}_bench_nolabels.err; exit 1; }
python tools/bench_summary.py gpurun_out/${tag}_bench.json gpurun_out/${tag}_bench_pairs.json gpurun_out/${tag}_bench_nolabels.json
