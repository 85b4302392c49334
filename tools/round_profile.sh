# Produces the round's committed evidence under gpurun_out/final/: bench JSON (with cpu baseline), rocprofv3 kernel stats of
# the same command, FETCH_SIZE / WRITE_SIZE passes, SQ instruction counters of the two main kernels, a 64-pair and a pairs-workload line.
# Run on the GPU box from the repo root: bash tools/round_profile.sh ; then (anywhere) python tools/collect_profiles.py r04_final
# (the profiled commands skip bench.py's config-5 leg: its small scene-flow launches would dilute the per-launch averages)
out=$PWD/gpurun_out/final
rm -rf $out; mkdir -p $out
timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-config5 --no-latency --workload pairs > $out/bench_pairs.json 2> $out/bench_pairs.err
timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-config5 --no-latency --frames 64 > $out/bench_64.json 2> $out/bench_64.err
timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-config5 --no-latency --camera kitti > $out/bench_kitti.json 2> $out/bench_kitti.err
timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-config5 --no-latency --workload nominal > $out/bench_nominal.json 2> $out/bench_nominal.err
# the committed line (the driver's own command) AFTER the four short runs: the first GPU process on a fresh box runs its kernels 3-6 % slower
# than the ones that follow it (round 5: 95.3 k first, 98.9-101.1 k in the three traced runs minutes later on the same box)
timeout -k 10 500 python3 bench.py --steps 20 --warmup 5 > $out/bench_default.json 2> $out/bench_default.err
echo "bench rc=$?"
repo=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $repo/bench.py --no-cpu-baseline --no-config5 --no-latency > $out/stats_bench.json 2> $out/stats.err
echo "stats rc=$?"
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/pmc_$ctr -- python3 $repo/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-config5 --no-latency > $out/pmc_$ctr.json 2> $out/pmc_$ctr.err
  echo "$ctr rc=$?"
done
# effective clock per kernel (MI355X_MICROARCH.md "DVFS give-back": GRBM_GUI_ACTIVE / 8 XCDs / kernel wall time)
timeout -k 10 400 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/pmc_clk -- python3 $repo/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-config5 --no-latency > $out/pmc_clk.json 2> $out/pmc_clk.err
echo "clk rc=$?"
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/pmc_sq$i -- python3 $repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-config5 --no-latency --frames 64 > $out/pmc_sq$i.json 2> $out/pmc_sq$i.err
  echo "sq$i rc=$?"
done
# on-GPU disparity: timings, config-5 end-to-end lines, kernel trace and SQ counters of tools/time_sgm.py
cd $repo
timeout -k 10 200 python3 tools/time_sgm.py > $out/sgm_720.log 2> /dev/null
SGM_F=10 timeout -k 10 200 python3 tools/time_sgm.py 1920 1080 > $out/sgm_1080.log 2> /dev/null
timeout -k 10 300 python3 tools/bench_config5.py > $out/config5_1080.json 2> $out/config5.err
timeout -k 10 300 python3 tools/bench_config5.py --width 1280 --height 720 > $out/config5_720.json 2>> $out/config5.err
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/sgm_stats -- python3 $repo/tools/time_sgm.py > /dev/null 2> $out/sgm_stats.err
echo "sgm stats rc=$?"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $out/sgm_pmc -- python3 $repo/tools/time_sgm.py > /dev/null 2> $out/sgm_pmc.err
echo "sgm pmc rc=$?"
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/sgm_clk -- python3 $repo/tools/time_sgm.py > /dev/null 2> $out/sgm_clk.err
echo "sgm clk rc=$?"
timeout -k 10 200 python3 $repo/bench.py --no-cpu-baseline --no-config5 --no-latency --objects-only > $out/bench_objects_only.json 2> $out/bench_objects_only.err
timeout -k 10 200 python3 $repo/bench.py --no-cpu-baseline --no-config5 --no-latency --chunks 1 > $out/bench_chunks1.json 2> $out/bench_chunks1.err
timeout -k 10 200 python3 $repo/bench.py --no-cpu-baseline --no-config5 --no-latency --chunks 2 > $out/bench_chunks2.json 2> $out/bench_chunks2.err
timeout -k 10 200 python3 $repo/bench.py --no-cpu-baseline --no-config5 --no-latency --chunks 1 --steps 200 --warmup 50 > $out/bench_chunks1_long.json 2> $out/bench_chunks1_long.err
timeout -k 10 200 python3 $repo/bench.py --no-cpu-baseline --no-config5 --no-latency --chunks 2 --steps 200 --warmup 50 > $out/bench_chunks2_long.json 2> $out/bench_chunks2_long.err
ls $out
