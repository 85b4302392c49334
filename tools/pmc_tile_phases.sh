# SQ instruction counters of k_ccl_tile per cumulative phase: the ablation build returns from the kernel after a phase (MOD_DEBUG bits
# 1<<14 .. 1<<18, 256 = without phase B).  Needs (here, before gpurun): make -C moving_object_detector_amd/csrc ABLATE=1 CHECKED=1 OUT=../libmod_sf_ablate.so
# usage (GPU box, repo root): bash tools/pmc_tile_phases.sh
export MOD_SF_LIB=$PWD/moving_object_detector_amd/libmod_sf_ablate.so
for d in 16384 32768 65536 131072 256 262144 0; do
  echo "== MOD_DEBUG=$d"
  MOD_DEBUG=$d PMC_ARGS="--frames 64" bash tools/pmc.sh pmc_ph_$d "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES" k_ccl_tile 2>&1 | grep -E "SQ_"
done
