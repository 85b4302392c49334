"""One line per bench JSON: value, ms per step, roofline fraction, per-kernel ms."""
import json, sys
for p in sys.argv[1:]:
    try:
        j = json.loads(open(p).read().strip().splitlines()[-1])
    except Exception as e:
        print(p, "unreadable:", e); continue
    r = j["roofline"]
    print(f'{p}: {j["value"]:.0f} pairs/s x{j["n_gpus"]}  {j["ms_per_step"]:.3f} ms/step  frac {r["frac"]:.3f}  ' +
          " ".join(f'{k.replace("k_", "")}={v:.3f}' for k, v in r["kernels_ms_per_launch"].items()))
