for F in ${FRAMES_LIST:-64 128 256}; do
  timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --distinct 16 --frames $F 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('frames=$F pairs/s', round(d['value']), 'ms/step', round(d['ms_per_step'],3), {k: round(v,3) for k,v in r['kernels_ms_per_launch'].items()})"
done
