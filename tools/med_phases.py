"""k_median / k_median_ties phase clocks (PHASE_COUNTERS build) on the bench stream + the cluster-size distribution.
MOD_SF_LIB=.../libmod_sf_pc.so MOD_DEBUG=128 python tools/med_phases.py [frames]"""
import ctypes as C, os, sys
os.environ.setdefault("MOD_DEBUG", "128")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from moving_object_detector_amd import synth, capi
from moving_object_detector_amd.pipeline import Context
W, H, F, G = 1280, 720, int(sys.argv[1]) if len(sys.argv) > 1 else 64, 16
cam, sq = synth.make_sequence(W, H, G, seed=4)
idx = [i % G for i in range(F)]
dev = torch.device("cuda:0")
d = torch.from_numpy(sq["disparity"]).to(dev)
ctx = Context(W, H, max_frames=F)
ctx.set_camera(cam); ctx.set_params(synth.Params())
ws = ctx.workspace(F)
b = ctx.make_batch(d[1:][idx].contiguous(), d[:-1][idx].contiguous(), torch.from_numpy(sq["flow"]).to(dev)[idx].contiguous(), sq["t"][idx], sq["q"][idx], sq["dt"][idx])
lib = ctx.lib
lib.mod_debug_counters.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
out = (C.c_uint64 * 96)()
for it in range(3):
    ctx.process(b, ws); ctx.synchronize()
    lib.mod_debug_counters(ctx.h, out)
objs = ctx.objects_to_host(ws)
sizes = np.concatenate([o["n_points"] for o in objs[:G]])
print("clusters per frame %.2f; sizes: min %d median %d mean %d max %d; share > 8192: %.2f, > 16384: %.2f, > 32768: %.2f" % (
    len(sizes) / G, sizes.min(), np.median(sizes), sizes.mean(), sizes.max(), (sizes > 8192).mean(), (sizes > 16384).mean(), (sizes > 32768).mean()))
n = max(int(out[41]), 1)
print("k_median per cluster-carrying block (100 MHz ticks -> us), summed over blocks / busy blocks: load+range %.1f  rounds %.1f  exact rank %.1f  ties+flag %.1f" % tuple(out[i] / 100.0 / n for i in (26, 27, 28, 29)))
print("  per busy block: round-0 pass %.1f scan %.1f | later rounds pass %.1f scan %.1f | compaction pass %.1f  (us)" % tuple(out[i] / 100.0 / n for i in (30, 31, 32, 33, 34)))
print("k_median blocks: busy %d, mean busy-block life %.1f us, first start -> last end %.1f us" % (out[41], out[40] / n / 100.0, (out[43] - out[42]) / 100.0))
print("k_median_ties (us, total over the launch): box %.1f  mask %.1f  place %.1f  (layout total %.1f)  hbm partition %.1f  lds %.1f" % tuple(out[i] / 100.0 for i in (20, 21, 22, 23, 24, 25)))
for name, base in (("HBM arrays", 0), ("LDS arrays, range > 2048", 8), ("LDS arrays, range <= 2048", 44)):
    n = max(int(out[base + 7]), 1)
    print(f"tie partition steps on {name}: {int(out[base + 7])} steps; shader cycles per step and section (pivot, count pass, prefix, position pass, swaps, decision): "
          + " ".join(f"{out[base + i] / n:.0f}" for i in range(6)))
