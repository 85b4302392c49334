"""A/B of library builds INSIDE one process: the same input and output buffers (so the same physical placement, which is what makes
two processes differ by +-4 %), contexts of the builds taken in turns.  python tools/ab_inproc.py libA.so libB.so ...  (names under
moving_object_detector_amd/).  Times mod_process_dev's scene-flow stage (stage timers) and the whole step (wall clock)."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from moving_object_detector_amd import capi, synth
from moving_object_detector_amd import pipeline

W, H, F, G = 1280, 720, int(os.environ.get("FRAMES", 512)), 16
cam, sq = synth.make_sequence(W, H, G, seed=4)
idx = [i % G for i in range(F)]
dev = torch.device("cuda:0")
d = torch.from_numpy(sq["disparity"]).to(dev)
d_now, d_prev = d[1:][idx].contiguous(), d[:-1][idx].contiguous()
flow = torch.from_numpy(sq["flow"]).to(dev)[idx].contiguous()
if int(os.environ.get("STAGGER_IN", "0")):             # the three input planes laid out as bench.py lays them out
    a, b, c = pipeline.staggered([F * H * W, F * H * W, 2 * F * H * W], torch.float32, dev)
    a, b, c = a.view(F, H, W), b.view(F, H, W), c.view(F, H, W, 2)
    a.copy_(d_now); b.copy_(d_prev); c.copy_(flow)
    d_now, d_prev, flow = a, b, c
ts, qs, dts = sq["t"][idx], sq["q"][idx], sq["dt"][idx]
ctxs = []
ws = None
for name in sys.argv[1:]:
    capi._lib = None
    capi.LIB_PATH = os.path.join(ROOT, "moving_object_detector_amd", name)
    ctx = pipeline.Context(W, H, max_frames=F)
    ctx.set_camera(capi.camera_struct(cam)); ctx.set_params(capi.params_struct(synth.Params()))
    if ws is None:
        ws = ctx.workspace(F)
    batch = ctx.make_batch(d_now, d_prev, flow, ts, qs, dts)
    ctxs.append((name, ctx, batch))
    for _ in range(3):
        ctx.process(batch, ws)
    torch.cuda.synchronize()
res = {n: [] for n, _, _ in ctxs}
ALL = bool(int(os.environ.get("ALL_STAGES", "0")))      # event pairs around every stage (costs the step a few percent)
stages = {}
for rep in range(int(os.environ.get("REPS", 6))):
    for name, ctx, batch in (ctxs if rep % 2 == 0 else ctxs[::-1]):
        ctx.set_profiling(True, stages=None if ALL else [capi.MOD_STAGE_SCENE_FLOW]); ctx.reset_stage_times()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            ctx.process(batch, ws)
        torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 10
        tot, n = ctx.stage_time(capi.MOD_STAGE_SCENE_FLOW)
        res[name].append((tot / n, wall * 1e3))
        if ALL:
            st = [ctx.stage_time(i) for i in range(capi.MOD_STAGE_COUNT)]
            stages.setdefault(name, []).append([t / max(k, 1) for t, k in st])
for name in res:
    sf = [a for a, _ in res[name]]; wl = [b for _, b in res[name]]
    print(f"{name:32s} scene_flow ms: " + " ".join(f"{x:.3f}" for x in sf) + f" | mean {np.mean(sf):.3f}  step ms mean {np.mean(wl):.3f}")
    if ALL:
        m = np.mean(np.array(stages[name]), axis=0)
        print("    " + "  ".join(f"{capi.STAGE_NAMES[i]}={m[i]:.3f}" for i in range(capi.MOD_STAGE_COUNT)))
