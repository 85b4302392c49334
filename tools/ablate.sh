for d in ${DEBUGS:-0 1}; do
  MOD_DEBUG=$d timeout -k 10 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --distinct 4 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); g=d['roofline']['groups']['cluster']; print('debug=$d ccl_ms', round(g['ccl_ms'],3), 'objects_ms', round(g['objects_ms'],3), 'pairs/s', round(d['value']))"
done
