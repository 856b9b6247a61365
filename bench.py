#!/usr/bin/env python3
"""bench.py -- the ray-march hot path on N GPUs of one node (contract: see the task brief).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one frame: rad pre-pass + ray march of this rank's screen bands (+ one RCCL
gather of RGBA8 bands to rank 0 when N > 1).  Inputs are resident in HBM before the timed
region.  Workload at N = 1 is BASELINE.json configs[2] (C3): 1024^3 f32 volume, 1920x1080,
step 1/512, RGBA transfer function, early-ray termination.  N > 1 scales the frame with N
(weak scaling; N = 8 is configs[3] (C4): 3840x2160, step 1/1024).
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REPO, "volume-viz_amd", "python"))

import numpy as np
import torch
import torch.distributed as dist

import volviz_amd as vv
from volviz_amd import sharding

HBM_PEAK = 8.0e12          # B/s, MI355X_MICROARCH.md "HBM3E peak BW" (spec)
BRICK = 8                  # brick edge of the algorithmic-byte model

# frame / step per GPU count: per-GPU sample budget stays that of C3 (weak scaling)
FRAMES = {1: (1920, 1080, 512), 2: (2716, 1528, 512), 4: (3840, 2160, 512), 8: (3840, 2160, 1024)}


def ramp_tf() -> np.ndarray:
    """Synthetic RGBA colour ramp (all reference tables are grey): alpha = 0.03 v^2, so rays
    through the noise volume reach the ERT threshold after ~400 samples."""
    v = np.arange(256, dtype=np.float64) / 255.0
    tf = np.stack([v, 1.0 - v, np.abs(2.0 * v - 1.0), 0.03 * v * v], axis=1)
    return tf.astype(np.float32).reshape(1024)


def workload(args, world):
    if args.config == "c3":
        n = 1024
        fw = args.frame_of or world
        W, H, steps = FRAMES.get(fw, (int(1920 * fw ** 0.5), int(1080 * fw ** 0.5), 512))
    elif args.config == "c5":     # 2048^3 f32 (32 GiB, > 4 GiB addressing path), Phong: BASELINE configs[4] on one GPU
        n, (W, H, steps) = 2048, (1920, 1080, 2048)
        args.phong = True
    elif args.config == "c2":
        n, (W, H, steps) = 256, (1280, 720, 256)
    else:   # c1 geometry on the GPU (the CPU-runnable case)
        n, (W, H, steps) = 128, (512, 512, 128)
    if args.size:
        n = args.size
    return n, W, H, steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="c3", choices=["c1", "c2", "c3", "c5"])
    ap.add_argument("--volume", default="noise", choices=["noise", "brain"])
    ap.add_argument("--tf", default="ramp", choices=["ramp", "head", "engine"])
    ap.add_argument("--view", default="a", choices=["a", "b"])
    ap.add_argument("--orbit", default="", help="theta,phi in degrees: camera on the orbit of radius 4 (diagnostic; overrides --view)")
    ap.add_argument("--size", type=int, default=0, help="override the volume edge (debug)")
    ap.add_argument("--voxel", default="f32", choices=["f32", "u8"], help="u8 is a diagnostic variant, not the C3 metric")
    ap.add_argument("--filter", default="tex8", choices=["tex8", "exact"])
    ap.add_argument("--ert", default="reference", choices=["reference", "true"])
    ap.add_argument("--phong", action="store_true", help="Phong-shaded path (diagnostic; C3 is unshaded)")
    ap.add_argument("--frame-of", type=int, default=0, help="render the frame/step an N-GPU run would use (check aid)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world == 1 and args.gpus > 1:
        sys.exit("launch N > 1 with torch.distributed.run (one process per GPU)")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a HIP device: the product has no CPU path")
    # VV_BENCH_SHARE_GPU=1 (developer rehearsal on a 1-GPU box): every rank uses cuda:0 and the
    # gather goes through gloo on host copies.  Never used for reported numbers.
    share = os.environ.get("VV_BENCH_SHARE_GPU") == "1"
    if share:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    n, W, H, steps = workload(args, world)
    ctx = vv.Context(local)
    # everything (kernels, events, the gather) runs on one non-default stream: vv_render only enqueues on a caller
    # stream (with no stream it would run on the context's own stream and synchronise every frame)
    tstream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(tstream)
    stream = vv.stream_handle(tstream)

    # ---- synthetic volume, generated and promoted on the device, replicated per GPU ----
    v8 = torch.empty(n * n * n, dtype=torch.uint8, device=dev)
    if args.volume == "noise":
        ctx.generate_noise_device(v8.data_ptr(), n, n, n, 0x9E3779B9, stream)
    else:
        ctx.generate_default_brain_device(v8.data_ptr(), n, n, n, stream)
    v32 = torch.empty(n * n * n, dtype=torch.float32, device=dev)
    ctx.promote_device(v8.data_ptr(), v32.data_ptr(), n * n * n, stream)
    tf = {"ramp": ramp_tf(), "head": vv.transfer_preset(vv.TF_HEAD), "engine": vv.transfer_preset(vv.TF_ENGINE)}[args.tf]
    if args.voxel == "u8":
        ctx.load_volume_device(v8.data_ptr(), vv.VOXEL_U8, n, n, n, tf, stream)
    else:
        ctx.load_volume_device(v32.data_ptr(), vv.VOXEL_F32, n, n, n, tf, stream)
    torch.cuda.synchronize()
    host_vol = None
    want_cpu = (not args.no_cpu_baseline) and rank == 0 and world == 1
    if want_cpu:
        host_vol = (v32 if args.voxel == "f32" else v8).cpu().numpy().reshape(n, n, n)
    del v32, v8
    torch.cuda.empty_cache()

    cam = vv.Camera() if args.view == "a" else vv.Camera.orbit(4.0, np.pi / 3, np.pi / 5)
    if args.orbit:
        th, ph = (float(v) for v in args.orbit.split(","))
        cam = vv.Camera.orbit(4.0, np.radians(th), np.radians(ph))
    step = 1.0 / steps
    base = dict(step=step, filter=vv.FILTER_TEX8 if args.filter == "tex8" else vv.FILTER_EXACT,
                ert_mode=vv.ERT_REFERENCE if args.ert == "reference" else vv.ERT_TRUE,
                shard=sharding.shard_option(world, rank))
    opts = vv.make_options(**base)

    # Frames are double-buffered: the gather of frame k (one RCCL collective) overlaps the
    # march of frame k+1; VV_BENCH_SYNC_GATHER=1 waits for each gather before the next frame instead.
    G = sharding.FrameGatherer(H, W, world, rank, device=dev)
    sync_gather = share or os.environ.get("VV_BENCH_SYNC_GATHER") == "1"
    frame = G.frames[0]

    def submit(b):
        if world == 1:
            return
        if not share:
            G.submit(b)
            if sync_gather:
                G.finish(b)
            return
        torch.cuda.synchronize()                 # rehearsal on one GPU: gloo on host copies
        out = sharding.gather_frame(G.frames[b].cpu(), world, rank)
        if rank == 0:
            G.frames[b].copy_(out)

    def one_frame(o, b=0):
        G.finish(b)
        ctx.render_device(W, H, cam, G.frames[b].data_ptr(), options=o, stream=stream, phong=args.phong)
        submit(b)

    # ---- untimed instrumented pass: executed samples + bricks touched (byte model) ----
    nb = (n + BRICK - 1) // BRICK
    bitmap = torch.zeros((nb * nb * nb + 31) // 32, dtype=torch.int32, device=dev)
    iopts = vv.make_options(count_samples=True, touched_bricks=bitmap.data_ptr(), **base)
    ctx.render_device(W, H, cam, frame.data_ptr(), options=iopts, stream=stream, phong=args.phong)
    torch.cuda.synchronize()
    samples = ctx.last_sample_count()
    if os.environ.get("VV_STATS"):      # developer statistics from a counters-only frame (no brick marking)
        ctx.render_device(W, H, cam, frame.data_ptr(), options=vv.make_options(count_samples=True, **base), stream=stream, phong=args.phong)
        torch.cuda.synchronize()
        print("stats", ctx.debug_counters().tolist(), "instrumented frame ms", ctx.last_frame_ms(), file=sys.stderr)
    words = bitmap.cpu().numpy().view(np.uint32)
    bricks = int(np.unpackbits(words.view(np.uint8)).sum())
    rows_owned = len(sharding.owned_rows(H, world, rank))
    vbytes = 4 if args.voxel == "f32" else 1
    bytes_rank = bricks * BRICK ** 3 * vbytes + 4 * W * rows_owned + 4096     # SURVEY 8d B_frame
    rdev = torch.device("cpu") if share else dev
    tot = torch.tensor([samples, bytes_rank], dtype=torch.float64, device=rdev)
    if world > 1:
        dist.all_reduce(tot)
    samples_all, bytes_all = float(tot[0]), float(tot[1])

    # ---- warm-up, then the timed region ----
    if world > 1:
        one_frame(opts, 0); one_frame(opts, 1)      # opens the point-to-point channels even when --warmup 0
    for k in range(args.warmup):
        one_frame(opts, k & 1)
    G.drain()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        b = k & 1
        G.finish(b)
        ev[k][0].record()
        ctx.render_device(W, H, cam, G.frames[b].data_ptr(), options=opts, stream=stream, phong=args.phong)
        ev[k][1].record()
        submit(b)
    G.drain()                                        # every frame is complete on rank 0 inside the timed region
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t1 = time.perf_counter()
    frame = G.frames[(args.steps - 1) & 1]
    el = torch.tensor([t1 - t0], dtype=torch.float64, device=rdev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el[0])
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))    # rad + march kernels of this rank

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    if os.environ.get("VV_BENCH_FRAME_SHA"):
        import hashlib
        print("frame_sha", hashlib.sha256(frame[:H].cpu().numpy().tobytes()).hexdigest()[:16], file=sys.stderr)
    ms_per_step = elapsed / args.steps * 1e3
    value = samples_all * args.steps / elapsed / 1e6
    achieved = bytes_rank / (kern_ms * 1e-3)
    traffic = None
    pj = os.path.join(REPO, "profiles", "pmc_traffic.json")
    if os.path.exists(pj):
        try:
            traffic = None if args.orbit else json.load(open(pj)).get(f"{args.config}-{args.volume}-{args.tf}-{args.view}-n{world}")
        except Exception:
            traffic = None
    out = {
        "metric": "Msamples/s (rays x steps), 1024^3 f32 volume @1080p", "value": round(value, 1),
        "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.voxel, "data": "synthetic",
        "config": {"workload": f"{args.config.upper()}: {n}^3 {args.voxel} {args.volume} volume, {W}x{H}, step 1/{steps}, "
                               f"{args.tf} RGBA TF, ERT {args.ert}, {args.filter} filter, view {args.orbit or args.view}" + (", Phong" if args.phong else ""),
                   "volume": [n, n, n], "frame": [W, H], "steps_per_unit_length": steps,
                   "sharding": "single GPU" if world == 1 else f"bands of {sharding.BAND_PX} pixel rows round-robin over {world} GPUs, volume replicated, 1 RCCL gather/frame"},
        "executed_samples_per_frame": int(samples_all), "upper_bound_samples_WxHxS": W * H * steps,
        "kernel_ms_rank0": round(kern_ms, 4),
        "roofline": {"bound": "hbm", "kernel": "march_kernel (+rad_kernel)", "achieved": round(achieved / 1e9, 1),
                     "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": round(achieved / HBM_PEAK, 4),
                     "traffic": traffic, "algorithmic_bytes_per_launch": int(bytes_rank),
                     "bytes_per_sample": round(bytes_rank / max(samples, 1), 3)},
    }

    # Not part of `value`: the same workload seen from a direction off the memory axis (SURVEY 8d's
    # second camera), where vv_render samples the bricked copy of the volume.
    if world == 1 and args.config == "c3" and args.view == "a" and not args.orbit and not os.environ.get("VV_BENCH_NO_EXTRA"):
        cam_b = vv.Camera.orbit(4.0, np.pi / 3, np.pi / 5)
        ctx.render_device(W, H, cam_b, frame.data_ptr(), options=vv.make_options(count_samples=True, **base), stream=stream, phong=args.phong)
        torch.cuda.synchronize()
        samples_b = ctx.last_sample_count()
        for _ in range(2):
            ctx.render_device(W, H, cam_b, frame.data_ptr(), options=opts, stream=stream, phong=args.phong)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ctx.render_device(W, H, cam_b, frame.data_ptr(), options=opts, stream=stream, phong=args.phong)
        e1.record()
        torch.cuda.synchronize()
        ms_b = e0.elapsed_time(e1) / 10
        out["rotated_view"] = {"camera": "orbit r=4, theta=60 deg, phi=36 deg", "ms_per_frame": round(ms_b, 4),
                               "value": round(samples_b / ms_b / 1e3, 1), "unit": "Msamples/s",
                               "executed_samples_per_frame": int(samples_b)}

    if want_cpu:
        # the CPU restatement (oracle/vvo.c, OpenMP) on this box's host cores, same workload.
        # The GPU box shares its host: 16 cores is the share of one GPU.
        sys.path.insert(0, os.path.join(REPO, "tests"))
        import oracle_lib as O
        cores = int(os.environ.get("VV_CPU_THREADS", min(len(os.sched_getaffinity(0)), 16)))
        nby = (H + 13) // 14
        mid = nby // 2
        cbase = {k: v for k, v in base.items() if k != "shard"}
        t = time.perf_counter()
        _, s1 = O.render(host_vol, tf, W, H, cam, options=vv.make_options(slab_rows=(mid, mid + 1), **cbase), threads=cores)
        dt1 = time.perf_counter() - t
        rows = int(max(1, min(nby, args.cpu_seconds / max(dt1, 1e-3))))
        lo = max(0, min(nby - rows, mid - rows // 2))
        copts = vv.make_options(slab_rows=(lo, lo + rows), **cbase)
        t = time.perf_counter()
        sN, reps = 0, 0
        while reps == 0 or (rows == nby and time.perf_counter() - t < args.cpu_seconds and reps < 32):
            sN += O.render(host_vol, tf, W, H, cam, options=copts, threads=cores)[1]
            reps += 1
        dt = time.perf_counter() - t
        # single-thread rate on one centre slab row (BASELINE.md section 3 asks for 1 and nproc)
        t1 = time.perf_counter()
        _, s1t = O.render(host_vol, tf, W, H, cam, options=vv.make_options(slab_rows=(mid, mid + 1), **cbase), threads=1)
        d1t = time.perf_counter() - t1
        out["cpu_baseline_1thread"] = {"value": round(s1t / d1t / 1e6, 2), "unit": "Msamples/s", "cores": 1, "kind": "port",
                                       "sample": f"centre slab row ({s1t} samples in {d1t:.1f} s)"}
        what = f"{reps} x the whole frame" if rows == nby else f"slab rows [{lo},{lo + rows}) of {nby} of the same frame"
        out["cpu_baseline"] = {"value": round(sN / dt / 1e6, 2), "unit": "Msamples/s", "cores": cores, "kind": "port",
                               "sample": f"{what} ({sN} samples in {dt:.1f} s; oracle/vvo.c, OpenMP, {cores} threads)"}
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
