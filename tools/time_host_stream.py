"""PCIe-inclusive rate of the host boundary: mod_process_frame_host (synchronous) vs mod_submit/collect_frame_host
(MOD_PIPELINE_DEPTH frames in flight, pinned buffers), 1280x720, with and without the 32 B/px cloud coming back."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from moving_object_detector_amd import capi, synth
from moving_object_detector_amd.pipeline import Context

W, H, G, NF, CAP = 1280, 720, 4, 60, 64
cam, b = synth.make_batch(W, H, G, seed=0)
N = W * H
ctx = Context(W, H, max_frames=1)
ctx.set_camera(cam)
ctx.set_params(synth.Params())
tfs = capi.transforms_array(b["t"], b["q"])


def pinned(shape, dtype):
    t = torch.empty(shape, dtype=dtype, pin_memory=True)
    return t, t.numpy()


ONLY = os.environ.get("ONLY")   # e.g. ONLY=pinned,nocloud,pipe for a profiler run
for use_pinned in ((True,) if ONLY else (False, True)):
    if use_pinned:
        keep = [pinned((G,) + b[k].shape[1:], torch.float32) for k in ("disparity_now", "disparity_prev", "flow")]
        for (t, a), k in zip(keep, ("disparity_now", "disparity_prev", "flow")):
            a[...] = b[k]
        dn, dp, fl = (a for _, a in keep)
        kc, clouds = pinned((capi.MOD_PIPELINE_DEPTH, N, 8), torch.float32)
        kl, labels = pinned((capi.MOD_PIPELINE_DEPTH, N), torch.int32)
    else:
        dn, dp, fl = b["disparity_now"], b["disparity_prev"], b["flow"]
        clouds = np.zeros((capi.MOD_PIPELINE_DEPTH, N, 8), np.float32)
        labels = np.zeros((capi.MOD_PIPELINE_DEPTH, N), np.int32)
    objs = [(capi.ModObject * CAP)() for _ in range(capi.MOD_PIPELINE_DEPTH)]
    n, t = C.c_int32(0), C.c_int32(0)
    for want_cloud in ((False,) if ONLY else (True, False)):
        # synchronous
        t0 = time.perf_counter()
        for i in range(0 if ONLY else NF):
            f = i % G
            rc = ctx.lib.mod_process_frame_host(ctx.h, dn[f].ctypes.data, dp[f].ctypes.data, fl[f].ctypes.data, C.byref(tfs[f]), 0.1,
                                                clouds[0].ctypes.data if want_cloud else None, labels[0].ctypes.data, objs[0], CAP, C.byref(n))
            assert rc == 0
        sync = NF / max(time.perf_counter() - t0, 1e-9)
        # pipelined
        tickets = []
        t0 = time.perf_counter()
        for i in range(NF):
            f, s = i % G, i % capi.MOD_PIPELINE_DEPTH
            if len(tickets) == capi.MOD_PIPELINE_DEPTH:
                assert ctx.lib.mod_collect_frame_host(ctx.h, tickets.pop(0), C.byref(n)) == 0
            rc = ctx.lib.mod_submit_frame_host(ctx.h, dn[f].ctypes.data, dp[f].ctypes.data, fl[f].ctypes.data, C.byref(tfs[f]), 0.1,
                                               clouds[s].ctypes.data if want_cloud else None, labels[s].ctypes.data, objs[s], CAP, C.byref(t))
            assert rc == 0, ctx.lib.mod_last_error(ctx.h)
            tickets.append(t.value)
        while tickets:
            assert ctx.lib.mod_collect_frame_host(ctx.h, tickets.pop(0), C.byref(n)) == 0
        pipe = NF / (time.perf_counter() - t0)
        mb = (16 + 4 + (32 if want_cloud else 0)) * N / 1e6
        print(f"host buffers {'pinned' if use_pinned else 'pageable'}, cloud back: {want_cloud}: {mb:.1f} MB per frame over PCIe; "
              f"synchronous {sync:.0f} frames/s, pipelined {pipe:.0f} frames/s ({pipe * mb / 1e3:.1f} GB/s), objects in last frame {n.value}")
ctx.close()
