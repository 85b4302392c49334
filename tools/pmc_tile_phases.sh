export MOD_SF_LIB=$PWD/moving_object_detector_amd/libmod_sf_ablate.so
for d in 16384 32768 65536 131072 256 262144 0; do
  echo "== MOD_DEBUG=$d"
  MOD_DEBUG=$d PMC_ARGS="--frames 64" bash tools/pmc.sh pmc_ph_$d "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES" k_ccl_tile 2>&1 | grep -E "SQ_"
done
