# DEBUGS other than 0 need the ablation build: make -C moving_object_detector_amd/csrc clean && make -C moving_object_detector_amd/csrc ABLATE=1
for d in ${DEBUGS:-0}; do
  MOD_DEBUG=$d timeout -k 10 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --distinct 4 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('debug=$d pairs/s', round(d['value']), {k: round(v,3) for k,v in r['kernels_ms_per_launch'].items()})"
done
