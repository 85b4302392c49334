"""Why tiles leave the bit-plane path (k_ccl_bits -> tilelist -> k_ccl_tile_list) on the bench streams.  Needs a DIAGNOSTIC build that was
never committed: one `atomicAdd(&a.dbg[52 ...], 1)` at each of k_ccl_bits' three bail sites and one per active tile (PHASE_COUNTERS=1 for
mod_debug_counters).  Round 5, 16 frames of 1280x720: 3590 active tiles, 63 bails, ALL of them "more than 4 depth classes" (nominal
workload: 2755 / 44 + 3 "class not clear of the one before").  With kMaxClasses = 8 the bails drop to 8 — and the tile stage gets
slower where it matters (8 pairs: 60 -> 75 us, 64 pairs: 92 -> 110 us; 512 pairs: 0.372 -> 0.363 ms): one wave peeling eight classes
takes longer than the four-wave union-find kernel takes for the tile.  Not adopted.
MOD_SF_LIB=.../libmod_sf_bail.so MOD_DEBUG=0 python tools/bail_reasons.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from moving_object_detector_amd import synth, capi
from moving_object_detector_amd.pipeline import Context
W, H, F, G = 1280, 720, 16, 16
for name, kw in (("default stream", {}), ("nominal", dict(synth.NOMINAL))):
    cam, sq = synth.make_sequence(W, H, G, seed=4, **kw)
    dev = torch.device("cuda:0")
    d = torch.from_numpy(sq["disparity"]).to(dev)
    ctx = Context(W, H, max_frames=F); ctx.set_camera(cam); ctx.set_params(synth.Params())
    ws = ctx.workspace(F)
    b = ctx.make_batch(d[1:].contiguous(), d[:-1].contiguous(), torch.from_numpy(sq["flow"]).to(dev), sq["t"], sq["q"], sq["dt"])
    lib = ctx.lib; lib.mod_debug_counters.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    out = (C.c_uint64 * 96)()
    ctx.process(b, ws); ctx.synchronize(); lib.mod_debug_counters(ctx.h, out)
    ctx.process(b, ws); ctx.synchronize(); lib.mod_debug_counters(ctx.h, out)
    print(name, "active tiles", out[58], "bails: > 4 classes", out[52], "| class not clear of the one before, at pass 1 / 2 / 3:", out[54], out[55], out[56], "| > 32 pieces", out[57])
    ctx.close()
