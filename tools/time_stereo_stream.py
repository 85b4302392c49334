"""PCIe-inclusive rate of BASELINE config 5 through the host boundary at 1920x1080 (and 1280x720): stereo images + flow in, objects out.
  two calls : mod_sgm_compute_host (disparity to the host) + mod_submit_frame_host (disparity back to the GPU) — round 2's only host-fed path
  one call  : mod_submit_stereo_host — the disparity plane stays in the pipe's ring in HBM
Pinned host buffers, MOD_PIPELINE_DEPTH frames in flight, labels + objects back (no 32 B/px cloud).  usage (GPU box): python tools/time_stereo_stream.py"""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from moving_object_detector_amd import capi, synth
from moving_object_detector_amd.pipeline import Context

out = {}
for (W, H, NF) in ((1920, 1080, 24), (1280, 720, 40)):
    D, G, CAP = 128, 2, 64
    N = W * H
    imgs = [synth.make_stereo_images(W, H, 300 + k, D, n_boxes=5) for k in range(G)]
    cam = synth.make_camera(W, H)
    cam.min_disparity, cam.max_disparity = np.float32(0.0), np.float32(D - 1)
    ctx = Context(W, H, max_frames=1)
    ctx.set_camera(cam); ctx.set_params(synth.Params())
    pin = lambda shape, dt: torch.empty(shape, dtype=dt, pin_memory=True)
    tl, tr, tf_, td = pin((G, H, W), torch.uint8), pin((G, H, W), torch.uint8), pin((G, H, W, 2), torch.float32), pin((capi.MOD_PIPELINE_DEPTH + 1, H, W), torch.float32)
    tlab = pin((capi.MOD_PIPELINE_DEPTH, N), torch.int32)
    left, right, flow, disp, labels = tl.numpy(), tr.numpy(), tf_.numpy(), td.numpy(), tlab.numpy()
    for k in range(G):
        left[k], right[k], flow[k] = imgs[k][0], imgs[k][1], synth.make_box_flow(imgs[k][2], 24.0)
    tfs = capi.transforms_array(np.zeros((G, 3)), np.tile(np.array([[0.0, 0.0, 0.0, 1.0]]), (G, 1)))
    sp = capi.ModSgmParams(D, 6, 96, 8, 1, 1)
    objs = [(capi.ModObject * CAP)() for _ in range(capi.MOD_PIPELINE_DEPTH)]
    n, t = C.c_int32(0), C.c_int32(0)
    dt = 1.0 / 15.0

    def run(one_call):
        tickets = []
        ctx.lib.mod_forget_previous(ctx.h)
        t0 = time.perf_counter()
        for i in range(NF):
            f, s = i % G, i % capi.MOD_PIPELINE_DEPTH
            if len(tickets) == capi.MOD_PIPELINE_DEPTH:
                assert ctx.lib.mod_collect_frame_host(ctx.h, tickets.pop(0), C.byref(n)) == 0
            if one_call:
                rc = ctx.lib.mod_submit_stereo_host(ctx.h, left[f].ctypes.data, right[f].ctypes.data, C.byref(sp), flow[f].ctypes.data, C.byref(tfs[f]), dt,
                                                    None, labels[s].ctypes.data, objs[s], CAP, None, C.byref(t))
            else:
                d = disp[i % (capi.MOD_PIPELINE_DEPTH + 1)]
                assert ctx.lib.mod_sgm_compute_host(ctx.h, left[f].ctypes.data, right[f].ctypes.data, C.byref(sp), d.ctypes.data) == 0
                # frame 0 ends at a guard before anything is enqueued, so frame 1 brings its previous disparity along (what the host
                # mirror's submit() does with its parked copy); from then on the previous plane is resident
                dprev = disp[(i - 1) % (capi.MOD_PIPELINE_DEPTH + 1)].ctypes.data if i == 1 else None
                rc = ctx.lib.mod_submit_frame_host(ctx.h, d.ctypes.data, dprev, flow[f].ctypes.data, C.byref(tfs[f]), dt, None, labels[s].ctypes.data, objs[s], CAP,
                                                   C.byref(t))
            if rc == 0:
                tickets.append(t.value)
            else:
                assert rc == capi.MOD_SKIP_NO_DISPARITY_PREV and i == 0, (rc, ctx.lib.mod_last_error(ctx.h))
        while tickets:
            assert ctx.lib.mod_collect_frame_host(ctx.h, tickets.pop(0), C.byref(n)) == 0
        return NF / (time.perf_counter() - t0)

    run(True); run(False)                                            # warm-up (scratch allocation)
    one, two = run(True), run(False)
    out[f"{W}x{H}"] = {"one_call_frames_per_s": one, "two_calls_frames_per_s": two, "objects_in_last_frame": n.value,
                       "bytes_over_pcie_per_frame": {"one_call": (2 + 8 + 4) * N, "two_calls": (2 + 4 + 4 + 8 + 4) * N}}
    ctx.close()
print(json.dumps({"metric": "stereo pairs/s through the host boundary incl. PCIe, config 5 (images -> on-GPU SGM -> scene flow + clusters)", **out}))
