"""Placement experiment inside one process: the six output planes (and the three input planes) at skewed offsets from one block each.
The default layout puts plane i at i * F * N * 4 bytes — a multiple of 8 MiB at 512 x 1280 x 720 — so the six stores (and three
loads) of a wave carry the same low address bits.  python tools/ab_skew.py  (prints the scene-flow stage time per layout)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from moving_object_detector_amd import capi, synth, pipeline

W, H, F, G = 1280, 720, int(os.environ.get("FRAMES", 512)), 16
N = W * H
cam, sq = synth.make_sequence(W, H, G, seed=4)
idx = [i % G for i in range(F)]
dev = torch.device("cuda:0")
if os.environ.get("LIB"):                              # another build of the library (name under moving_object_detector_amd/)
    capi.LIB_PATH = os.path.join(ROOT, "moving_object_detector_amd", os.environ["LIB"])
d = torch.from_numpy(sq["disparity"]).to(dev)
d_now0, d_prev0 = d[1:][idx].contiguous(), d[:-1][idx].contiguous()
flow0 = torch.from_numpy(sq["flow"]).to(dev)[idx].contiguous()
ts, qs, dts = sq["t"][idx], sq["q"][idx], sq["dt"][idx]
ctx = pipeline.Context(W, H, max_frames=F)
ctx.set_camera(capi.camera_struct(cam)); ctx.set_params(capi.params_struct(synth.Params()))
ws0 = ctx.workspace(F)
PAD = 128 << 20
out_block = torch.empty(6 * F * N + 6 * PAD // 4, dtype=torch.float32, device=dev)
in_block = torch.empty(4 * F * N + 4 * PAD // 4, dtype=torch.float32, device=dev)


def layout(skew_out, skew_in):
    ws = dict(ws0)
    ws["planes"] = [out_block[(i * F * N + i * skew_out // 4):][:F * N].view(F, H, W) for i in range(6)]
    o = [0, F * N + skew_in // 4, 2 * F * N + 2 * skew_in // 4]
    dn = in_block[o[0]:][:F * N].view(F, H, W); dn.copy_(d_now0)
    dp = in_block[o[1]:][:F * N].view(F, H, W); dp.copy_(d_prev0)
    fl = in_block[o[2]:][:2 * F * N].view(F, H, W, 2); fl.copy_(flow0)
    return ws, ctx.make_batch(dn, dp, fl, ts, qs, dts)


K, M = 1 << 10, 1 << 20
cases = [(0, 0)] + [(s, 0) for s in (128 * K, 256 * K, 512 * K, M, M + 4 * K, 2 * M, 3 * M, 4 * M, 5 * M // 4, 1396736, 7 * M)] + \
        [(M + 4 * K, M + 4 * K), (M, M), (2 * M, 2 * M), (0, M), (0, 2 * M)]
if os.environ.get("CASES"):
    cases = [tuple(int(v) for v in c.split(":")) for c in os.environ["CASES"].split(",")]
res = {c: [] for c in cases}
for rep in range(int(os.environ.get("REPS", 3))):
    for c in (cases if rep % 2 == 0 else cases[::-1]):
        ws, batch = layout(*c)
        for _ in range(2):
            ctx.process(batch, ws)
        ctx.set_profiling(True, stages=[capi.MOD_STAGE_SCENE_FLOW]); ctx.reset_stage_times()
        for _ in range(8):
            ctx.process(batch, ws)
        torch.cuda.synchronize()
        tot, n = ctx.stage_time(capi.MOD_STAGE_SCENE_FLOW)
        res[c].append(tot / n)
for c in cases:
    print(f"skew out {c[0]:>8d} B  in {c[1]:>6d} B : " + " ".join(f"{x:.3f}" for x in res[c]) + f" | mean {np.mean(res[c]):.3f} ms")
