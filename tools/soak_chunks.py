"""Soak of the chunked cluster stage (ModConfig.batch_chunks): random sizes, batch lengths (64 ... 200 frames), every parameter of both
.cfg files over its range, quantised (tie-rich) and hostile inputs — the chunked call's outputs against the un-chunked call's, byte
for byte (the un-chunked path is what tools/soak.py checks against the oracle).  python tools/soak_chunks.py [seconds] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from moving_object_detector_amd import synth
from moving_object_detector_amd.pipeline import Context

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t_end = time.time() + budget
it = 0
while time.time() < t_end:
    it += 1
    W = int(rng.choice([64, 131, 160, 257, 320])); H = int(rng.choice([48, 67, 96, 120]))
    F = int(rng.integers(64, 201)); G = 8
    seed = int(rng.integers(0, 1 << 30))
    cam, batch = synth.make_batch(W, H, G, seed=seed)
    prm = synth.Params(dynamic_flow_diff=int(rng.choice([1, 2, 5])), cluster_size=int(rng.choice([1, 5, 40, 200])),
                       neighbor_distance=int(rng.integers(1, 11)), depth_diff=float(rng.choice([0.01, 0.05, 0.15, 1.0])),
                       dynamic_speed=float(rng.choice([0.01, 0.1, 0.3, 1.0])))
    mode = int(rng.integers(0, 3))
    if mode == 1:
        batch["flow"] = (np.round(batch["flow"] * 2) / 2).astype(np.float32)
        batch["disparity_now"] = np.round(batch["disparity_now"]).astype(np.float32)
    if mode == 2:
        m = rng.random(batch["disparity_now"].shape) < 0.02
        batch["disparity_now"][m] = rng.choice(np.array([np.nan, np.inf, 0.0, -1.0, 1e-30, 500.0], np.float32), int(m.sum()))
    idx = [int(i) for i in rng.integers(0, G, F)]
    outs = []
    for chunks in (1, int(rng.choice([2, 3, 4]))):
        ctx = Context(W, H, max_frames=F, max_objects=W * H // max(prm.cluster_size, 1) + 1, batch_chunks=chunks)
        ctx.set_camera(cam); ctx.set_params(prm)
        ws = ctx.workspace(F)
        dev = ctx.device
        b = ctx.make_batch(torch.from_numpy(batch["disparity_now"][idx]).to(dev), torch.from_numpy(batch["disparity_prev"][idx]).to(dev),
                           torch.from_numpy(batch["flow"][idx]).to(dev), batch["t"][idx], batch["q"][idx], batch["dt"][idx])
        for _ in range(2):                                # twice over the same scratch
            ws["labels"].fill_(-7); ws["objects"].zero_(); ws["n_objects"].fill_(-1)
            assert ctx.process(b, ws) == 0
            ctx.synchronize()
        n = ws["n_objects"].cpu().numpy()
        raw = ws["objects"].cpu().numpy()
        outs.append((ws["planes"].cpu().numpy().tobytes(), ws["labels"].cpu().numpy().tobytes(), n.tobytes(), ws["n_clusters"].cpu().numpy().tobytes(),
                     [raw[f, : n[f]].tobytes() for f in range(F)], chunks))
        ctx.close()
    a, c = outs
    if a[:5] != c[:5]:
        print("MISMATCH", dict(W=W, H=H, F=F, seed=seed, prm=prm, chunks=c[5], mode=mode)); sys.exit(1)
    print(f"ok #{it}: {W}x{H}x{F} chunks {c[5]} mode {mode} n={prm.neighbor_distance} cs={prm.cluster_size}", flush=True)
print("chunk soak passed:", it, "configurations")
