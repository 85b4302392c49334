# Produces the round's committed evidence under gpurun_out/final/: bench JSON (with cpu baseline), rocprofv3 kernel stats of
# the same command, FETCH_SIZE / WRITE_SIZE passes.  Run on the GPU box: bash tools/round_profile.sh
out=/root/repo/gpurun_out/final
mkdir -p $out
timeout -k 10 500 python3 bench.py > $out/bench_default.json 2> $out/bench_default.err
echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 /root/repo/bench.py --no-cpu-baseline > $out/stats_bench.json 2> $out/stats.err
echo "stats rc=$?"
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/pmc_$ctr -- python3 /root/repo/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $out/pmc_$ctr.json 2> $out/pmc_$ctr.err
  echo "$ctr rc=$?"
done
ls $out
