"""Does a HIP graph of mod_process_dev's launches shorten a small step?  Captures one call with torch.cuda.CUDAGraph (the context
enqueues on torch's current stream) and replays it.  python tools/graph_probe.py [frames]
Needs a PROBE build of the library: upload_frame_consts (mod_sf.hip) waits for and records an event of its pinned ring, which a
capturing stream refuses — the probe build skipped both while hipStreamIsCapturing() said "active" (six lines, never committed; the
captured graph then replays the constants of the capture, which is all a timing needs).  Round 5 result (profiles/README.md): the
replay is exactly as fast as the direct launches when steps follow each other (0.1035 vs 0.1038 ms for 1 pair, 0.317 vs 0.315 for 8,
0.9025 vs 0.9048 for 64) and 3-10 us faster when the host waits after every step: the step is bound by the GPU-side cost of its
dependent launches, not by the host's launch calls — so the library cuts launches instead of replaying them."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from moving_object_detector_amd import capi, synth, pipeline

W, H, G = 1280, 720, 16
F = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cam, sq = synth.make_sequence(W, H, G, seed=4)
dev = torch.device("cuda:0")
idx = [i % G for i in range(F)]
d = torch.from_numpy(sq["disparity"]).to(dev)
d_now, d_prev = d[1:][idx].contiguous(), d[:-1][idx].contiguous()
flow = torch.from_numpy(sq["flow"]).to(dev)[idx].contiguous()
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    ctx = pipeline.Context(W, H, max_frames=F)
    ctx.set_camera(capi.camera_struct(cam)); ctx.set_params(capi.params_struct(synth.Params()))
    ws = ctx.workspace(F)
    batch = ctx.make_batch(d_now, d_prev, flow, sq["t"][idx], sq["q"][idx], sq["dt"][idx])
    for _ in range(5):
        ctx.process(batch, ws)
    side.synchronize()
    want = ws["labels"].clone(), ws["objects"].clone()

    def timeit(fn, steps=400, sync_each=False):
        for _ in range(10):
            fn()
        side.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
            if sync_each:
                side.synchronize()
        side.synchronize()
        return 1e3 * (time.perf_counter() - t0) / steps

    print(f"direct : {timeit(lambda: ctx.process(batch, ws)):.4f} ms back to back, {timeit(lambda: ctx.process(batch, ws), sync_each=True):.4f} ms synchronous", flush=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side, capture_error_mode="relaxed"):
        ctx.process(batch, ws)
    ws["labels"].fill_(-9)
    g.replay(); side.synchronize()
    print("graph replay results identical:", bool(torch.equal(ws["labels"], want[0]) and torch.equal(ws["objects"], want[1])), flush=True)
    print(f"graph  : {timeit(g.replay):.4f} ms back to back, {timeit(g.replay, sync_each=True):.4f} ms synchronous")
