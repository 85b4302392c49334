"""mod_process_dev in chunks (ModConfig.batch_chunks), A/B INSIDE one process: the same input and output buffers, one context per
setting, taken in turns.  Checks that every setting gives the un-chunked call's labels and objects, then times the whole step (wall
clock), the scene-flow kernel and the cluster group (stage timers 0 and 7).
python tools/chunk_ab.py [chunks ...]   (the round's logs under profiles/r05_logs/ also hold settings | 0x100 (scene-flow kernel cut too) and
| 0x200 (no step order / chunks held one kernel apart, depending on the build): measurement switches of commits 557698a ... d9702f0 that
have left the library again)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from moving_object_detector_amd import capi, synth
from moving_object_detector_amd import pipeline

W, H, F, G = 1280, 720, int(os.environ.get("FRAMES", 512)), 16
settings = [int(a, 0) for a in sys.argv[1:]] or [1, 2, 3, 4]
cam, sq = synth.make_sequence(W, H, G, seed=4)
idx = [i % G for i in range(F)]
dev = torch.device("cuda:0")
d = torch.from_numpy(sq["disparity"]).to(dev)
a, b, c = pipeline.staggered([F * H * W, F * H * W, 2 * F * H * W], torch.float32, dev)
d_now, d_prev, flow = a.view(F, H, W), b.view(F, H, W), c.view(F, H, W, 2)
d_now.copy_(d[1:][idx]); d_prev.copy_(d[:-1][idx]); flow.copy_(torch.from_numpy(sq["flow"]).to(dev)[idx])
ts, qs, dts = sq["t"][idx], sq["q"][idx], sq["dt"][idx]
ctxs, ws, want = [], None, None
for st in settings:
    ctx = pipeline.Context(W, H, max_frames=F, batch_chunks=st)
    ctx.set_camera(capi.camera_struct(cam)); ctx.set_params(capi.params_struct(synth.Params()))
    if ws is None:
        ws = ctx.workspace(F)
    batch = ctx.make_batch(d_now, d_prev, flow, ts, qs, dts)
    ws["labels"].fill_(-7); ws["objects"].zero_(); ws["n_objects"].fill_(-1)
    for _ in range(3):
        ctx.process(batch, ws)
    torch.cuda.synchronize()
    got = (ws["labels"].cpu().numpy().copy(), ws["n_objects"].cpu().numpy().copy(), [o.tobytes() for o in ctx.objects_to_host(ws)])
    if want is None:
        want = got
    same = np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]) and got[2] == want[2]
    print(f"setting {st:#x}: results {'identical to' if same else 'DIFFER from'} setting {settings[0]:#x}", flush=True)
    assert same
    ctxs.append((st, ctx, batch))
res = {st: [] for st in settings}
for rep in range(int(os.environ.get("REPS", 6))):
    for st, ctx, batch in (ctxs if rep % 2 == 0 else ctxs[::-1]):
        ctx.set_profiling(True, stages=[capi.MOD_STAGE_SCENE_FLOW, capi.MOD_STAGE_CLUSTER_GROUP]); ctx.reset_stage_times()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            ctx.process(batch, ws)
        torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 10
        sf, n = ctx.stage_time(capi.MOD_STAGE_SCENE_FLOW)
        cl, m = ctx.stage_time(capi.MOD_STAGE_CLUSTER_GROUP)
        res[st].append((wall * 1e3, sf / 10, cl / max(m, 1)))      # scene flow: per STEP (a cut kernel has several launches per step)
for st in settings:
    r = np.array(res[st])
    print(f"setting {st:#6x}: step ms " + " ".join(f"{x:.3f}" for x in r[:, 0]) + f" | mean {r[:, 0].mean():.3f} = {F / r[:, 0].mean():.1f} k pairs/s"
          f" | scene flow {r[:, 1].mean():.3f}  cluster group {r[:, 2].mean():.3f}")
