"""How many clusters of the bench stream take the tie replay (k_median_ties)?  Needs a diagnostic build (mod_debug_read):
MOD_SF_LIB=$PWD/moving_object_detector_amd/libmod_sf_checked.so python tools/count_ties.py [frames]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from moving_object_detector_amd import synth, capi
from moving_object_detector_amd.pipeline import Context
W, H, F, G = 1280, 720, int(sys.argv[1]) if len(sys.argv) > 1 else 16, 16
cam, sq = synth.make_sequence(W, H, G, seed=4)
idx = [i % G for i in range(F)]
dev = torch.device("cuda:0")
d = torch.from_numpy(sq["disparity"]).to(dev)
ctx = Context(W, H, max_frames=F)
ctx.set_camera(cam); ctx.set_params(synth.Params())
ws = ctx.workspace(F)
b = ctx.make_batch(d[1:][idx].contiguous(), d[:-1][idx].contiguous(), torch.from_numpy(sq["flow"]).to(dev)[idx].contiguous(), sq["t"][idx], sq["q"][idx], sq["dt"][idx])
ctx.process(b, ws); ctx.synchronize()
lib = ctx.lib
lib.mod_debug_read.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64]
MO = ctx.max_objects
ci = np.zeros((F, MO, 8), np.int32)
lib.mod_debug_read(ctx.h, 1, ci.ctypes.data, ci.nbytes)
K = ws["n_clusters"].cpu().numpy()
tot = tied = 0
for f in range(F):
    for k in range(K[f]):
        tot += 1
        if ci[f, k, 5] == 2:
            tied += 1
            print(f"frame {f} cluster {k}: size {ci[f, k, 1]} replayed")
print(f"{tied} of {tot} clusters took the replay")
