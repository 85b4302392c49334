#!/bin/bash
# usage (GPU box, repo root): bash tools/r5_step.sh <tag> [soak seconds] [libs for the in-process A/B ...]
# GPU suite -> soak through the CHECKED build -> soak of the product build -> in-process A/B of library builds with all stage timers.
# Steps are chained: a failing step stops the run (no GPU step after a failed one).
tag=$1; soak=${2:-60}; shift; shift
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_tests.log 2>&1; rc=$?
tail -3 gpurun_out/${tag}_tests.log
[ $rc -ne 0 ] && { grep -E "^E|Error|FAILED" gpurun_out/${tag}_tests.log | head -20; exit 1; }
if [ "$soak" != "0" ]; then
  MOD_SF_LIB=$PWD/moving_object_detector_amd/libmod_sf_checked.so timeout -k 10 $((soak + 120)) python tools/soak.py $soak $RANDOM > gpurun_out/${tag}_soak_checked.log 2>&1; rc=$?
  tail -1 gpurun_out/${tag}_soak_checked.log
  [ $rc -ne 0 ] && { grep MISMATCH gpurun_out/${tag}_soak_checked.log; exit 1; }
  timeout -k 10 $((soak + 120)) python tools/soak.py $soak $RANDOM > gpurun_out/${tag}_soak.log 2>&1; rc=$?
  tail -1 gpurun_out/${tag}_soak.log
  [ $rc -ne 0 ] && { grep MISMATCH gpurun_out/${tag}_soak.log; exit 1; }
fi
if [ $# -gt 0 ]; then
  ALL_STAGES=1 STAGGER_IN=1 REPS=${REPS:-6} timeout -k 10 400 python tools/ab_inproc.py "$@" > gpurun_out/${tag}_ab.log 2>&1 || { tail -5 gpurun_out/${tag}_ab.log; exit 1; }
  cat gpurun_out/${tag}_ab.log
fi
