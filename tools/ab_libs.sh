#!/bin/bash
# usage (GPU box): bash tools/ab_libs.sh <tag> <bench args...> -- <lib1> <lib2> ...   A/B of library builds through bench.py (MOD_SF_LIB)
tag=$1; shift
args=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do args+=("$1"); shift; done
shift
for rep in $(seq 1 ${REPS:-2}); do
for lib in "$@"; do
  n=$(basename $lib .so)
  MOD_SF_LIB=$PWD/moving_object_detector_amd/$lib python bench.py --no-cpu-baseline "${args[@]}" > gpurun_out/${tag}_${n}_$rep.json 2> gpurun_out/${tag}_${n}_$rep.err || { tail -3 gpurun_out/${tag}_${n}_$rep.err; exit 1; }
  python tools/bench_summary.py gpurun_out/${tag}_${n}_$rep.json
done
done
