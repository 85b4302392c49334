#!/usr/bin/env python3
"""bench.py — stereo pairs/s of the scene-flow + cluster hot path at 1280x720 on N MI355X GPUs.

A "step" is one pass of the whole hot path (mod_process_dev: fused scene-flow kernel -> dynamic mask -> windowed
connected components -> per-cluster objects) over one batch of F synthetic 1280x720 stereo pairs that are already
resident in HBM when the timed region starts.  Multi-GPU = frame sharding (weak scaling: every rank owns F pairs);
the only collective is the RCCL broadcast of the camera-intrinsics/parameter block before the timed region.

Prints ONE JSON line on rank 0 (contract in the task statement), with two extra objects:
  roofline     — the dominant kernel (longest average launch; HIP events on the launch stream inside the timed region),
                 priced with SURVEY.md §8(d)'s algorithmic bytes per pixel; per-kernel numbers of an untimed breakdown
                 pass (events around every kernel group) alongside
  cpu_baseline — the CPU oracle ("port": op-for-op restatement of the reference loops) timed on this box's host cores
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

# RCCL across processes needs dmabuf IPC on this pool's driver: with the legacy IPC mode hipIpcGetMemHandle fails with "invalid
# argument" (stated by the GPU pool's environment notes; the launchers export it, an external `torchrun ... bench.py` may not).
# The HSA runtime reads its environment when it loads, so this has to stand before `import torch` — here, at import of this file.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
IPC_ENV_SET_BEFORE_TORCH = "torch" not in sys.modules and os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") == "0"

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# numpy / torch are imported inside main(): with --gpus N > 1 this process only starts the N ranks and must not touch the GPU

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s; 6.29 TB/s measured float4 copy)
HBM_COPY_GBS = 6290.0
# measured on this pool in round 4 with timing-only builds of the scene-flow kernel (profiles/README.md): its plane stores alone run at
# 5.65 TB/s whether one plane or six are written and whatever the cache policy, its loads alone at 6.4 TB/s, and both together take
# the SUM of the two — so 24 B written + 16 B read per pixel cannot move faster than 40 / (24 / 5650 + 16 / 6400) = 5.93 TB/s
HBM_WRITE_GBS, HBM_READ_GBS = 5650.0, 6400.0
SF_RW_BOUND_GBS = 40.0 / (24.0 / HBM_WRITE_GBS + 16.0 / HBM_READ_GBS)
B_SCENE_FLOW = 40              # bytes/px: read disp_now 4 + disp_prev 4 + flow 8, write x,y,z,vx,vy,vz 24
B_CLUSTER = 20                 # bytes/px: read z,vx,vy,vz 16, write label 4
B_FUSED = 44                   # bytes/px: 16 in + 24 out + 4 label (mask never round-trips as a plane)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks of the job (default: WORLD_SIZE when a launcher set it, else 1)")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=512, help="stereo pairs per step and per GPU")
    ap.add_argument("--distinct", type=int, default=16, help="distinct synthetic pairs generated per GPU (tiled to --frames)")
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--camera", default="zed", choices=["zed", "kitti"], help="synthetic camera: ZED-like (config 2) or KITTI-style (config 3)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=24, help="pairs timed on the CPU (bounded sample)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for rehearsals)")
    ap.add_argument("--share-device", action="store_true", help="rehearsal on a 1-GPU box: every rank uses cuda:0")
    ap.add_argument("--workload", default="sequence", choices=["sequence", "pairs", "nominal"],
                    help="sequence: one synthetic stream (frame t's now-disparity is frame t+1's previous), sharded over the ranks "
                         "as contiguous chunks + a one-plane disparity halo (SURVEY.md 8(e)); pairs: round 1's independent pairs; "
                         "nominal: the same stream with SURVEY.md 8(d)'s nominal object depth (4-15 m), speed (0.5-2 m/s) and "
                         "dt = 1/15 s, where the default uses 4-9 m, 1-2 m/s and 0.1 s (DESIGN.md section 10)")
    ap.add_argument("--no-labels", action="store_true",
                    help="do not produce the cluster-label plane (the reference renders its cluster image only for subscribers)")
    ap.add_argument("--objects-only", action="store_true",
                    help="the call of a node that serves ~moving_objects alone: no x / y planes (nobody takes the cloud, "
                         "scene_flow_constructor.cpp:141-142) and no cluster-label plane; NOT the headline configuration")
    ap.add_argument("--chunks", type=lambda v: int(v, 0), default=0,
                    help="ModConfig.batch_chunks: 2..4 = cluster stage in that many chunks of frames side by side (worth 1-3 %% in a long run, "
                         "nothing in a process's first calls: include/mod_sf.h); 0 / 1 = one piece (default)")
    ap.add_argument("--seed", type=int, default=4, help="scene seed of the synthetic stream")
    ap.add_argument("--no-dist", action="store_true",
                    help="N = 1 only: skip the process group (by default a one-rank RCCL group is brought up so that the single-GPU "
                         "line drives the communicator, the device broadcast and the device all-reduce of the N-rank run)")
    ap.add_argument("--no-latency", action="store_true", help="skip the frame-at-a-time leg (steps of 1 / 8 / 64 pairs, outside `value`)")
    ap.add_argument("--no-config5", action="store_true", help="skip the short config-5 leg (on-GPU SGM disparity, outside `value`)")
    ap.add_argument("--launch-check", action="store_true",
                    help="rendezvous + config broadcast + stream sharding only, no GPU work (exercises the N-rank launch path on a CPU box)")
    ap.add_argument("--fail-rank", type=int, default=None,
                    help="rehearsal of a rank failure: this rank raises after the rendezvous (the job must exit non-zero, name the rank on "
                         "stderr and print no JSON line)")
    return ap.parse_args()


def cpu_baseline(cam, prm, batch, n_sample, sgm=None):
    """Times the oracle on this box's host cores on the first n_sample pairs of rank 0's batch.  `sgm` = (left, right, D, gpu
    disparity) of one frame of the config-5 leg: the CPU restatement of the disparity estimator (oracle/sgm_ref.cpp) is timed on
    it (one frame, one core) and the GPU's plane compared with its result."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor

    from oracle import pyoracle
    pyoracle.lib()
    n = min(n_sample, batch["disparity_now"].shape[0])

    def one(f, mode):
        ref = pyoracle.construct(cam, prm, batch["disparity_now"][f], batch["disparity_prev"][f], batch["flow"][f],
                                 batch["t"][f], batch["q"][f], float(batch["dt"][f]), mode)
        t1 = time.perf_counter()
        lab, objs, K = pyoracle.cluster(ref["cloud"] if mode == "faithful" else ref, prm, mode)
        return t1, ref, lab, objs

    res = {}
    refs = []
    passes = {"faithful": 6, "tidy": 4}              # ~10 s + ~3 s of single-core work on the sample
    for mode in ("faithful", "tidy"):
        sf = cl = 0.0
        for rep in range(passes[mode]):
            for f in range(n):
                t0 = time.perf_counter()
                t1, ref, lab, objs = one(f, mode)
                t2 = time.perf_counter()
                sf += t1 - t0
                cl += t2 - t1
                if mode == "tidy" and rep == 0:
                    refs.append((ref, lab, objs))
        m = n * passes[mode]
        res[mode] = {"scene_flow_ms": 1e3 * sf / m, "cluster_ms": 1e3 * cl / m, "pairs_per_s": m / (sf + cl)}
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    workers = max(1, cores)                            # every core this process may run on
    reps = max(n, 4 * workers)
    t0 = time.perf_counter()
    with ThreadPoolExecutor(workers) as ex:
        list(ex.map(lambda f: one(f % n, "faithful")[0], range(reps)))
    allcore = reps / (time.perf_counter() - t0)
    out = {
        "value": res["faithful"]["pairs_per_s"], "unit": "stereo pairs/s", "cores": 1, "kind": "port",
        "sample": f"{n} of rank 0's 1280x720 pairs x {passes['faithful']} passes, oracle in 'faithful' mode (reference layouts, "
                  f"column-major walks, per-frame allocations), single thread = what the reference uses for this path",
        "ms_per_frame": {"scene_flow": res["faithful"]["scene_flow_ms"], "cluster": res["faithful"]["cluster_ms"]},
        "tidy_1core": res["tidy"],
        "all_cores": {"value": allcore, "threads": workers, "host_cores": cores, "mode": "faithful, frame-sharded"},
    }
    if sgm is not None:
        from oracle import pysgm
        left, right, D, gpu_disp = sgm
        pysgm.lib()
        t0 = time.perf_counter()
        want = pysgm.compute(left, right, D)
        dt = time.perf_counter() - t0
        out["config5_sgm"] = {"frames_per_s": 1.0 / dt, "cores": 1, "kind": "port", "sample": f"one {left.shape[1]}x{left.shape[0]} image pair, D = {D}, 8 paths",
                              "gpu_disparity_matches_oracle": bool(np.array_equal(gpu_disp, want))}
    return out, refs


def config5_leg(dev, local_rank):
    """BASELINE config 5's data path, short and OUTSIDE `value`: synthetic stereo images -> on-GPU SGM disparity
    (replaces sgm_gpu::SgmGpu::computeDisparity, scene_flow_constructor.cpp:35,267) -> scene flow + clusters, all HBM-resident.
    16 frames of 1280x720 and 10 of 1920x1080 per step = TWO groups of the estimator each (8 / 5 frames: the winner-take-all of
    group k overlaps the path aggregation of group k + 1, which is how the estimator is meant to run), plus one 8-frame step at
    1280x720 for the un-overlapped single-group figure.  Prices the disparity estimator against its HBM
    roofline (8 paths x D bytes/px written + read back = 2 KB/px at D = 128) and its VALU bound.  Returns (numbers, check):
    `check` = (left, right, D, GPU disparity) of one 720p frame, which cpu_baseline() hands to the CPU restatement."""
    import ctypes as C

    import numpy as np
    import torch

    from moving_object_detector_amd import capi, synth
    from moving_object_detector_amd.pipeline import Context

    D, PATHS = 128, 8
    check = None
    out = {"disparities": D, "paths": PATHS, "data": "synthetic stereo images with moving boxes; flow = 0 + a shift on the boxes; identity ego-motion"}
    for (W, H, F, key) in ((1280, 720, 8, "720_one_group"), (1280, 720, 16, "720"), (1920, 1080, 10, "1080")):
        imgs = [synth.make_stereo_images(W, H, 300 + k, D, n_boxes=5) for k in range(2)]
        cam = synth.make_camera(W, H)
        cam.min_disparity, cam.max_disparity = np.float32(0.0), np.float32(D - 1)
        ctx = Context(W, H, max_frames=F, device=local_rank)
        ctx.set_camera(cam)
        ctx.set_params(synth.Params())
        idx = [i % 2 for i in range(F)]
        left = torch.from_numpy(np.stack([p[0] for p in imgs])).to(dev)[idx].contiguous()
        right = torch.from_numpy(np.stack([p[1] for p in imgs])).to(dev)[idx].contiguous()
        # a stream that stands still (now == previous image pair, identity ego-motion) except for the boxes, which the flow moves
        flow = torch.from_numpy(np.stack([synth.make_box_flow(p[2], shift=24.0) for p in imgs])).to(dev)[idx].contiguous()
        ts = np.zeros((F, 3)); qs = np.tile(np.array([[0.0, 0.0, 0.0, 1.0]]), (F, 1)); dts = np.full(F, 1.0 / 15.0)
        d_now = torch.zeros((F, H, W), dtype=torch.float32, device=dev)
        sp = capi.ModSgmParams(D, 6, 96, PATHS, 1, 1)
        ws = ctx.workspace(F)
        batch = ctx.make_batch(d_now, d_now, flow, ts, qs, dts)      # previous = now: the same image pair twice

        def step():
            rc = ctx.lib.mod_sgm_compute_dev(ctx.h, F, left.data_ptr(), right.data_ptr(), C.byref(sp), d_now.data_ptr())
            assert rc == 0, rc
            ev[1].record()
            assert ctx.process(batch, ws) == 0

        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        step()
        torch.cuda.synchronize()
        steps, sgm_ms = 4, 0.0
        t0 = time.perf_counter()
        for _ in range(steps):
            ev[0].record()
            step()
            ev[2].record()
            torch.cuda.synchronize()
            sgm_ms += ev[0].elapsed_time(ev[1])
        el = time.perf_counter() - t0
        sgm_frame_ms = sgm_ms / (steps * F)
        out["pairs_per_s_" + key] = F * steps / el
        out["sgm_ms_per_frame_" + key] = sgm_frame_ms
        out["objects_per_frame_" + key] = float(ws["n_objects"].float().mean())
        out["valid_disparity_share_" + key] = float((d_now >= 0).float().mean())
        if key == "720":
            n = W * H
            gbs = 2 * PATHS * D * n / (sgm_frame_ms * 1e-3) / 1e9
            # a step of four path lines is 84 VALU instructions per wave (927 M per dispatch of 11 M wave-steps, profiles/README.md
            # round 3) = 21 per line step, 4 cycles each, over 1024 SIMDs at 2.4 GHz
            valu_ms = PATHS * n * 21 * 4 / (1024 * 2.4e9) * 1e3
            out["sgm_ms_per_frame"] = sgm_frame_ms
            # the clock the chip HOLDS inside the path grid after seconds of back-to-back calls (in-kernel s_memtime / s_memrealtime
            # stamps of a diagnostic build, tools/sgm_clock_probe.py, round 5; GRBM_GUI_ACTIVE in the counter passes: 2.10-2.12)
            SGM_CLOCK_GHZ = 2.157
            out["sgm_roofline"] = {"bytes_per_px": 2 * PATHS * D, "GBps": gbs, "frac": gbs / HBM_PEAK_GBS, "valu_bound_ms": valu_ms,
                                   "frac_of_valu_bound": valu_ms / sgm_frame_ms,
                                   "clock_held_ghz_measured_round5": SGM_CLOCK_GHZ,
                                   "frac_of_valu_bound_at_clock_held": valu_ms * 2.4 / SGM_CLOCK_GHZ / sgm_frame_ms,
                                   "at": "1280x720, 16 frames per step = two groups of 8, pipelined",
                                   "one_group_ms_per_frame": out.get("sgm_ms_per_frame_720_one_group")}
            check = (imgs[0][0], imgs[0][1], D, d_now[0].cpu().numpy())
        ctx.close()
        del left, right, flow, d_now, ws, batch
        torch.cuda.empty_cache()
    return out, check


def latency_leg(dev, local_rank, cam_s, prm_s, host, W, H, sizes=(1, 8, 64)):
    """Frame-at-a-time figures, OUTSIDE `value`: the reference processes ONE pair per callback at 15 frames/s
    (scene_flow_constructor.cpp:364-399, zed_common.yaml:24-25) and BASELINE's metric names ms/frame.  Device-resident inputs,
    mod_process_dev on steps of 1 / 8 / 64 pairs: `ms` = a step's time with the steps enqueued back to back (what a node that
    pipelines its frames sees), `ms_synchronous` = enqueue + wait, one step at a time (a node that needs frame t's objects before
    it hands over frame t + 1), per-kernel times from a pass with all stage timers on."""
    import numpy as np
    import torch

    from moving_object_detector_amd import capi
    from moving_object_detector_amd.pipeline import Context

    out = {"what": "mod_process_dev on device-resident steps of 1 / 8 / 64 pairs, six planes + labels + objects; ms = back-to-back steps, "
                   "ms_synchronous = one step at a time with a wait after each"}
    G = host["disparity_now"].shape[0]
    for F in sizes:
        idx = [i % G for i in range(F)]
        ctx = Context(W, H, max_frames=F, device=local_rank)
        ctx.set_camera(cam_s)
        ctx.set_params(prm_s)
        ws = ctx.workspace(F)
        d_now = torch.from_numpy(np.ascontiguousarray(host["disparity_now"][idx])).to(dev)
        d_prev = torch.from_numpy(np.ascontiguousarray(host["disparity_prev"][idx])).to(dev)
        flow = torch.from_numpy(np.ascontiguousarray(host["flow"][idx])).to(dev)
        batch = ctx.make_batch(d_now, d_prev, flow, host["t"][idx], host["q"][idx], host["dt"][idx])
        steps = max(20, 400 // F)
        for _ in range(10):
            ctx.process(batch, ws)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            ctx.process(batch, ws)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / steps
        t0 = time.perf_counter()
        for _ in range(steps):
            ctx.process(batch, ws)
            ctx.synchronize()
        ms_sync = 1e3 * (time.perf_counter() - t0) / steps
        ctx.set_profiling(True, stages=[i for i in range(capi.MOD_STAGE_COUNT) if i != capi.MOD_STAGE_CLUSTER_GROUP])
        ctx.reset_stage_times()
        for _ in range(20):
            ctx.process(batch, ws)
        torch.cuda.synchronize()
        st = [ctx.stage_time(i) for i in range(capi.MOD_STAGE_COUNT)]
        ctx.set_profiling(False)
        out[f"pairs_{F}"] = {"ms": ms, "pairs_per_s": F / (ms * 1e-3), "ms_synchronous": ms_sync,
                             "objects_per_frame": float(ws["n_objects"].float().mean()),
                             "kernels_ms": {capi.STAGE_NAMES[i]: (t / n if n else None) for i, (t, n) in enumerate(st)
                                            if i != capi.MOD_STAGE_CLUSTER_GROUP}}
        ctx.close()
        del ws, batch, d_now, d_prev, flow
    return out


def main():
    """Rank-level failures are fatal and visible: one line on stderr that names the rank, a non-zero exit code (the launcher then
    stops the other ranks and fails too), and no JSON line — the line is written last, by rank 0, after every rank's timed work."""
    rank = os.environ.get("RANK", "0")
    try:
        run()
    except SystemExit:
        raise
    except BaseException as e:                               # noqa: BLE001
        import traceback
        traceback.print_exc()
        sys.stderr.write(f"bench.py: rank {rank} failed: {type(e).__name__}: {e}\n")
        sys.stderr.flush()
        os._exit(1)                                          # not sys.exit: a hung communicator must not keep the process in its destructors


def run():
    args = parse()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and (args.gpus or 1) > 1:
        # `python bench.py --gpus N`: this process becomes the launcher.  It has made no GPU call (torch is not even imported),
        # starts N fresh ranks exactly as the driver would (torch.distributed.run, one process per GPU) and relays their exit code.
        from moving_object_detector_amd.launch import spawn_ranks
        sys.exit(spawn_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus))
    world = int(env_world or "1")
    if args.gpus is None:
        args.gpus = world                     # started by a launcher without --gpus: WORLD_SIZE is the job's size
    if world != args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU and pass the same N\n")
        sys.exit(2)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # stdout carries ONE JSON line and nothing else: librccl prints a version banner on file descriptor 1 when its first communicator
    # comes up, so descriptor 1 is pointed at stderr for the whole run and the line goes out through a saved copy of the real one
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        os.write(real_stdout, (json.dumps(obj) + "\n").encode())

    import numpy as np
    import torch
    import torch.distributed as dist

    from moving_object_detector_amd import capi, synth
    from moving_object_detector_amd import dist as mdist

    if not args.launch_check:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
        if args.share_device:
            local_rank = 0
        # a launcher may hand every rank ONE visible device (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES per rank): the device index
        # is then 0 whatever LOCAL_RANK says; with all devices visible it is LOCAL_RANK
        ndev = torch.cuda.device_count()
        if ndev == 1:
            local_rank = 0
        elif local_rank >= ndev:
            # fewer visible devices than ranks per node (and more than one): folding ranks onto the same GPU would report
            # throughput from oversubscribed devices — refuse (--share-device is the deliberate one-GPU rehearsal)
            raise RuntimeError(f"LOCAL_RANK {local_rank} but only {ndev} visible devices: launch one rank per visible GPU "
                               f"(or expose exactly one device per rank)")
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank) if not args.launch_check else torch.device("cpu")
    on_dev = args.backend == "nccl" and not args.launch_check
    backend = "gloo" if args.launch_check else args.backend
    group_error = None
    if world > 1:
        mdist.init_group(backend, rank, world, device=dev if on_dev else None)   # "nccl" is RCCL on ROCm
    elif not args.no_dist and not args.launch_check:
        # one rank: the same group code, on a communicator of one (RCCL is loaded and runs the broadcast / all-reduce kernels
        # on this GPU).  A box whose RCCL cannot come up still gets its line: the failure is reported, not hidden.
        try:
            mdist.init_group(backend, 0, 1, device=dev if on_dev else None)
        except Exception as e:                               # noqa: BLE001
            group_error = f"{type(e).__name__}: {e}"[:300]
    group_up = mdist.group_is_up()

    W, H, F = args.width, args.height, args.frames
    G = max(1, min(args.distinct, F))
    # rank 0 owns the camera + dynamic_reconfigure parameters (reference defaults) and broadcasts them over RCCL
    cam_s = prm_s = None
    if rank == 0:
        cam_s = capi.camera_struct(synth.make_camera(W, H, args.camera))
        prm_s = capi.params_struct(synth.Params())
    cam_s, prm_s = mdist.broadcast_config(cam_s, prm_s, src=0, device=dev if on_dev else None)

    total = world * G                                   # distinct frames of the whole job's stream
    lo, hi = mdist.shard_range(total, rank, world)
    if args.fail_rank is not None and rank == args.fail_rank:
        raise RuntimeError("rehearsed failure (--fail-rank)")
    if args.launch_check:
        counts = mdist.gather_counts(torch.tensor([lo, hi], dtype=torch.int32))
        if group_up:                                      # as in the real run: the group goes down before rank 0 works alone
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            emit({"launch_check": True, "n_gpus": world, "backend": "gloo", "camera_width": cam_s.width,
                              "cluster_size": prm_s.cluster_size, "shards": [c.tolist() for c in counts],
                              "ipc_env_set_before_torch": IPC_ENV_SET_BEFORE_TORCH,
                              "hsa_enable_ipc_mode_legacy": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"),
                              "group_down_before_solo_legs": not dist.is_initialized()})
        return

    from moving_object_detector_amd.pipeline import PLANES, Context

    # this rank's share of the stream, tiled to F frames at distinct HBM addresses
    if args.workload in ("sequence", "nominal"):
        # one stream of `total` frames; a rank materialises its contiguous chunk + the one-plane disparity halo
        # seed 4: the stream's share of dynamic pixels (12.7 %) and its cluster mix (4-6 clusters of 3-46 k px per frame) are close
        # to round 1's pair workload (13.1 %, 4-9 clusters); seeds differ 2.7x in dynamic share
        kw = dict(synth.NOMINAL) if args.workload == "nominal" else {}
        mk = lambda first, frames: synth.make_sequence(W, H, frames, seed=args.seed, first=first, camera=args.camera, **kw)[1]
        sh = mdist.local_stream(mk, total, rank, world)
        cam = synth.make_camera(W, H, args.camera)
        host = {"disparity_now": sh["disparity_now"], "disparity_prev": sh["disparity_prev"], "flow": sh["flow"],
                "t": sh["t"], "q": sh["q"], "dt": sh["dt"]}
    else:
        cam, host = synth.make_batch(W, H, G, seed=0, first_frame=rank * G, camera=args.camera)
    idx = [i % G for i in range(F)]
    # the three input planes come from one block with staggered bases, like the six output planes (pipeline.PLANE_STAGGER_BYTES)
    from moving_object_detector_amd.pipeline import staggered
    d_now, d_prev, flow = staggered([F * H * W, F * H * W, 2 * F * H * W], torch.float32, dev)
    d_now, d_prev, flow = d_now.view(F, H, W), d_prev.view(F, H, W), flow.view(F, H, W, 2)
    d_now.copy_(torch.from_numpy(np.ascontiguousarray(host["disparity_now"])).to(dev)[idx])
    d_prev.copy_(torch.from_numpy(np.ascontiguousarray(host["disparity_prev"])).to(dev)[idx])
    flow.copy_(torch.from_numpy(np.ascontiguousarray(host["flow"])).to(dev)[idx])
    ts, qs, dts = host["t"][idx], host["q"][idx], host["dt"][idx]

    if args.objects_only:
        args.no_labels = True
    ctx = Context(W, H, max_frames=F, device=local_rank, batch_chunks=args.chunks)
    ctx.set_camera(cam_s)
    ctx.set_params(prm_s)
    ws = ctx.workspace(F, labels=not args.no_labels, xy=not args.objects_only)
    batch = ctx.make_batch(d_now, d_prev, flow, ts, qs, dts)

    def barrier():
        if group_up:
            dist.barrier()

    # the warm-up steps run exactly as the timed ones do: same stage timers on (a call's path depends on them: ModConfig.batch_chunks)
    ctx.set_profiling(True, stages=[capi.MOD_STAGE_SCENE_FLOW, capi.MOD_STAGE_CLUSTER_GROUP])
    for _ in range(args.warmup):
        ctx.process(batch, ws)
    torch.cuda.synchronize()
    # Timed region: HIP events (on the stream the kernels run on) bracket the kernel the roofline prices, the scene-flow kernel,
    # and the cluster group as a whole (first launch to last); event pairs around every kernel would cost the timed region ~8 %
    # of stream time (and keep a --chunks run from running in chunks).
    ctx.set_profiling(True, stages=[capi.MOD_STAGE_SCENE_FLOW, capi.MOD_STAGE_CLUSTER_GROUP])
    ctx.reset_stage_times()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ctx.process(batch, ws)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    sf_timed = ctx.stage_time(capi.MOD_STAGE_SCENE_FLOW)
    cl_timed = ctx.stage_time(capi.MOD_STAGE_CLUSTER_GROUP)
    # Untimed breakdown pass over the same batch: every kernel bracketed (reported next to the roofline kernel); with these
    # timers on the call runs its cluster stage in ONE chunk, each kernel alone on the GPU.
    ctx.set_profiling(True)
    ctx.reset_stage_times()
    for _ in range(max(1, min(args.steps, 10))):
        ctx.process(batch, ws)
    torch.cuda.synchronize()
    stage = [ctx.stage_time(i) for i in range(capi.MOD_STAGE_COUNT)]
    ctx.set_profiling(False)
    elapsed = mdist.max_over_ranks(elapsed, device=dev if on_dev else None)     # the slowest rank's time (device all-reduce on RCCL)
    # The job's collective work ends here: the group goes down on every rank BEFORE rank 0's solo legs (config 5, ~15 s of CPU
    # baseline), so that no rank sits in an RCCL barrier with a watchdog running while rank 0 computes alone.
    coll = {"executed": bool(group_up), "backend": (dist.get_backend() if group_up else None),
            "world_size": (dist.get_world_size() if group_up else 0)}
    if group_up:
        dist.barrier()
        dist.destroy_process_group()

    if rank == 0:
        N = W * H
        pairs = world * F * args.steps
        value = pairs / elapsed
        ms = [(t / n if n else float("nan")) for t, n in stage]                 # average launch duration per kernel
        breakdown_sf_ms = ms[capi.MOD_STAGE_SCENE_FLOW]
        ms[capi.MOD_STAGE_SCENE_FLOW] = sf_timed[0] / sf_timed[1]               # the priced kernel: from the timed region
        # the bytes of the planes a non-headline call does not produce are not priced either
        B_SCENE_FLOW = 40 - (8 if args.objects_only else 0)
        B_CLUSTER = 20 - (4 if args.no_labels else 0)
        B_FUSED = 44 - (8 if args.objects_only else 0) - (4 if args.no_labels else 0)
        per_kernel = [capi.MOD_STAGE_SCENE_FLOW] + list(capi.MOD_PER_KERNEL_CLUSTER_STAGES)
        kernels = {capi.STAGE_NAMES[i]: ms[i] for i in per_kernel}
        sf_ms = ms[capi.MOD_STAGE_SCENE_FLOW]
        cl_sum_ms = sum(ms[i] for i in capi.MOD_PER_KERNEL_CLUSTER_STAGES)     # the group's kernels one after the other, each alone
        cl_ms = cl_timed[0] / cl_timed[1]                                       # the group as the timed region ran it
        sf_gbs = F * N * B_SCENE_FLOW / (sf_ms * 1e-3) / 1e9
        cl_gbs = F * N * B_CLUSTER / (cl_ms * 1e-3) / 1e9
        dom = max(per_kernel, key=lambda i: ms[i])                             # the single kernel with the longest launch
        # algorithmic bytes of that kernel per launch: the scene-flow kernel moves its group's 40 B/px; a cluster kernel is
        # priced with the whole cluster group's 20 B/px (the group's bytes are not separable per kernel)
        dom_bytes = F * N * (B_SCENE_FLOW if dom == capi.MOD_STAGE_SCENE_FLOW else B_CLUSTER)
        ach = dom_bytes / (ms[dom] * 1e-3) / 1e9
        traffic = None
        import glob
        tfiles = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))   # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
        tpath = tfiles[-1] if tfiles else None                                          # the latest round's
        if tpath:
            tj = json.load(open(tpath))
            if tj.get("frames_per_launch") == F and tj.get("width") == W and tj.get("height") == H:
                traffic = tj["kernels"].get(capi.STAGE_NAMES[dom].split("+")[0], {}).get("hbm_bytes_per_launch")
        roof = {
            "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
            "kernel": capi.STAGE_NAMES[dom], "algorithmic_bytes_per_launch": dom_bytes,
            "algorithmic_bytes_per_px": B_SCENE_FLOW if dom == capi.MOD_STAGE_SCENE_FLOW else B_CLUSTER,
            "frames_per_launch": F, "avg_launch_ms": ms[dom],
            "measured_in": "timed region" if dom == capi.MOD_STAGE_SCENE_FLOW else "breakdown pass (all stage timers on)",
            "scene_flow_ms_in_breakdown_pass": breakdown_sf_ms, "frac_of_measured_copy_ceiling": ach / HBM_COPY_GBS,
            "frac_of_measured_read_write_bound": (ach / SF_RW_BOUND_GBS) if (dom == capi.MOD_STAGE_SCENE_FLOW and not args.objects_only) else None,
            "traffic_from_committed_profile": traffic is not None,       # a separate rocprofv3 --pmc run of this command, not this run
            "traffic_source": (f"profiles/{os.path.basename(tpath)} at commit {tj.get('head', '?')} (rocprofv3 --pmc FETCH_SIZE x2 on gfx950 + "
                               f"WRITE_SIZE in separate passes of this same command; counters cannot be read inside the run)"
                               if traffic else None),
            "kernels_ms_per_launch": kernels,
            "groups": {
                "scene_flow": {"ms_per_launch": sf_ms, "GBps": sf_gbs, "frac": sf_gbs / HBM_PEAK_GBS, "bytes_per_px": B_SCENE_FLOW},
                "cluster": {"ms_per_launch": cl_ms, "GBps": cl_gbs, "frac": cl_gbs / HBM_PEAK_GBS, "bytes_per_px": B_CLUSTER,
                            "measured_in": "timed region: HIP events around the whole group, first launch to last"
                                           + (f" (in {args.chunks & 0xff} chunks side by side: ModConfig.batch_chunks)" if (args.chunks & 0xff) > 1 else ""),
                            "kernels_one_after_the_other_ms": cl_sum_ms,
                            "frac_kernels_one_after_the_other": F * N * B_CLUSTER / (cl_sum_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
                "fused_end_to_end": {"GBps": F * N * B_FUSED / ((sf_ms + cl_ms) * 1e-3) / 1e9, "bytes_per_px": B_FUSED},
            },
        }
        n_chk = min(args.cpu_sample, G)
        planes = ws["planes"][:, :n_chk].cpu().numpy()   # the sampled pairs' outputs, for the same-run check against the oracle
        checked_planes = [i for i, k in enumerate(PLANES) if not (args.objects_only and k in ("x", "y"))]
        labels = ws["labels"][:n_chk].cpu().numpy() if ws["labels"] is not None else None
        c5 = c5_check = None
        lat = None
        if not args.no_latency or not args.no_config5:
            ctx.close()                                       # frees the 512-pair context's scratch before the SGM volumes come
            ctx._ws = None
            del batch, ws, d_now, d_prev, flow
            torch.cuda.empty_cache()
        if not args.no_latency:                               # rank 0 alone, the group is down (like config 5 and the CPU baseline)
            try:
                lat = latency_leg(dev, local_rank, cam_s, prm_s, host, W, H)
            except Exception as e:                            # noqa: BLE001  (outside `value`)
                lat = {"error": f"{type(e).__name__}: {e}"[:300]}
        if not args.no_config5:
            try:
                c5, c5_check = config5_leg(dev, local_rank)
            except Exception as e:                            # noqa: BLE001  (the leg is outside `value`; its failure must not cost the line)
                c5 = {"error": f"{type(e).__name__}: {e}"[:300]}
        cpu = None
        if not args.no_cpu_baseline:                     # rank 0 only, also when N > 1 (the group is down by now: the other ranks have left)
            prm = synth.Params()
            cpu, refs = cpu_baseline(cam, prm, host, args.cpu_sample, sgm=c5_check)
            ok = True
            for f, (ref, lab, objs) in enumerate(refs):
                if f >= n_chk:
                    break
                for i in checked_planes:
                    a, r = planes[i, f], ref[PLANES[i]]
                    ok &= bool(((a.view(np.uint32) == r.view(np.uint32)) | (np.isnan(a) & np.isnan(r))).all())
                if labels is not None:
                    ok &= bool(np.array_equal(labels[f], lab))
            cpu["gpu_outputs_match_oracle"] = ok
        line = {
            "metric": "stereo pairs/sec (scene-flow+cluster) at 1280x720", "value": value, "unit": "stereo pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "ms_per_frame": 1e3 * elapsed / (args.steps * F), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32+f64", "data": "synthetic",
            "config": {"workload": f"{W}x{H} synthetic sequence ({args.camera}-style camera), scene-flow + cluster kernels only (disparity/flow precomputed, "
                                   f"HBM-resident), reference default parameters",
                       "frames_per_step_per_gpu": F, "distinct_frames_per_gpu": G,
                       "outputs": ("z + velocity planes, dynamic mask, objects (objects-only call: no x / y planes)" if args.objects_only
                                   else "six cloud planes, dynamic mask, objects") + ("" if args.no_labels else ", cluster-label plane"),
                       "batch_chunks": args.chunks,
                       "stream": ("one synthetic stream, contiguous chunk per rank + one disparity plane of halo" if args.workload == "sequence"
                                  else "the same with SURVEY.md 8(d)'s nominal object depth 4-15 m, speed 0.5-2 m/s, dt 1/15 s" if args.workload == "nominal"
                                  else "independent synthetic pairs per rank"),
                       "sharding": f"frames x{world}",
                       "collective": ("one broadcast of the intrinsics/params block before the timed region + one all-reduce (MAX) of the elapsed time"
                                      if coll["executed"] else "none: no process group in this run"),
                       "collective_executed": coll["executed"], "collective_backend": coll["backend"],
                       "collective_on_device": bool(coll["executed"] and on_dev), "collective_world_size": coll["world_size"],
                       "collective_error": group_error},
            "roofline": roof, "cpu_baseline": cpu,
        }
        if lat is not None:
            line["latency"] = lat
        if c5 is not None:
            line["config5"] = c5
        emit(line)
    ctx.close()


if __name__ == "__main__":
    main()
