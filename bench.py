#!/usr/bin/env python3
"""bench.py -- the ray-march hot path on N GPUs of one node (contract: see the task brief).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
    python bench.py --gpus N ...        (no launcher: starts the N ranks itself, as a child torch.distributed.run, before any GPU call)

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


def csrc_sha() -> str:
    """Hash of the kernel / C-ABI sources: profiles/pmc_traffic.json carries the hash it was measured with, and a
    traffic figure taken with other kernels is not reported."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(REPO, "volume-viz_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h", ".cpp")):
            h.update(name.encode()); h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def measured_traffic(key):
    """HBM-side bytes per launch from the committed PMC passes (tools/pmc_traffic.py), or None (+ why)."""
    pj = os.path.join(REPO, "profiles", "pmc_traffic.json")
    if not os.path.exists(pj):
        return None, "no profiles/pmc_traffic.json"
    try:
        j = json.load(open(pj))
    except Exception as e:
        return None, f"unreadable: {e}"
    if j.get("csrc_sha") != csrc_sha():
        return None, f"stale: measured with csrc {j.get('csrc_sha')}, this tree is {csrc_sha()} (re-run tools/pmc_traffic.py)"
    return j.get("entries", {}).get(key), None


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as a child `python -m torch.distributed.run` (one process per GPU,
    rendezvous on 127.0.0.1) and return its exit code -- a rank that fails makes the launcher stop the others and return non-zero.  This
    process has not touched a GPU (importing torch does not), and it never replaces itself: it waits for the child and exits with its code."""
    import socket
    import subprocess
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL between processes needs it on this pool
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


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
    ap.add_argument("--rays", default="analytic", choices=["analytic", "images"], help="images: march from two 3x first-pass images resident in HBM, the runCuda-shaped call (diagnostic)")
    ap.add_argument("--frame-of", type=int, default=0, help="render the frame/step an N-GPU run would use (check aid)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus))                     # before any GPU call of this process
    if world != args.gpus:
        sys.exit(f"--gpus {args.gpus} but the launcher started {world} rank(s)")
    if os.environ.get("VV_BENCH_DRYRUN") == "1":
        # rehearsal of the launch path without a device (tests/test_sharding.py, CPU): rendezvous, one collective, the line's frame -- then stop
        if os.environ.get("VV_BENCH_DRYRUN_FAIL") == str(rank):
            sys.exit(3)                                      # (a rank that dies: the launcher must report it)
        if world > 1:
            dist.init_process_group("gloo")
            t = torch.tensor([float(rank + 1)], dtype=torch.float64)
            dist.all_reduce(t)
            ok = float(t[0]) == world * (world + 1) / 2
            dist.barrier()
            dist.destroy_process_group()
        else:
            ok = True
        if rank == 0:
            print(json.dumps({"dryrun": True, "n_gpus": world, "collective": {"backend": "gloo", "ranks": world}, "all_reduce_ok": ok,
                              "frame": list(workload(args, world)[1:3]), "local_rank": local}))
        sys.exit(0 if ok else 1)
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
    tf = {"ramp": ramp_tf(), "head": vv.transfer_preset(vv.TF_HEAD), "engine": vv.transfer_preset(vv.TF_ENGINE)}[args.tf]
    upload = None
    if args.config == "c5" and args.voxel == "f32" and not args.size:
        # C5 as BASELINE.json states it: every GPU streams its replica from pinned host memory, slab by slab through ONE
        # pinned buffer (u8 slabs, promoted to f32 on the device: vv_load_volume_stream_*).  The host copy stands in for
        # a file reader; the upload is reported beside the metric, never inside it.
        torch.cuda.synchronize()
        per = 32
        pin = torch.empty((per, n, n), dtype=torch.uint8).pin_memory()
        host8 = np.empty((n, n, n), np.uint8)
        v8v = v8.view(n, n, n)
        for z0 in range(0, n, per):
            pin.copy_(v8v[z0:z0 + per]); host8[z0:z0 + per] = pin.numpy()
        del v8v, v8
        torch.cuda.empty_cache()

        def slabs():
            for z0 in range(0, n, per):
                pin.numpy()[...] = host8[z0:z0 + per]
                yield z0, pin.numpy()
        t0 = time.perf_counter()
        ctx.load_volume_streamed(slabs(), vv.VOXEL_F32, n, n, n, tf)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        upload = {"what": "u8 slabs of 32 slices from one pinned buffer, promoted to f32 on the device (host memcpy into the buffer included)",
                  "seconds": round(dt, 3), "GB_per_s": round(n ** 3 / dt / 1e9, 2)}
        del host8, pin
        v8 = v32 = torch.empty(0, dtype=torch.uint8, device=dev)
    else:
        v32 = torch.empty(n * n * n, dtype=torch.float32, device=dev)
        ctx.promote_device(v8.data_ptr(), v32.data_ptr(), n * n * n, stream)
        if args.voxel == "u8":
            ctx.load_volume_device(v8.data_ptr(), vv.VOXEL_U8, n, n, n, tf, stream)
        else:
            ctx.load_volume_device(v32.data_ptr(), vv.VOXEL_F32, n, n, n, tf, stream)
    torch.cuda.synchronize()
    host_vol = None
    want_cpu = (not args.no_cpu_baseline) and rank == 0 and world == 1
    if want_cpu and upload is None:
        host_vol = (v32 if args.voxel == "f32" else v8).cpu().numpy().reshape(n, n, n)
    del v32, v8
    torch.cuda.empty_cache()

    cam = vv.Camera() if args.view == "a" else vv.Camera.orbit(4.0, np.pi / 3, np.pi / 5)
    if args.orbit:
        th, ph = (float(v) for v in args.orbit.split(","))
        cam = vv.Camera.orbit(4.0, np.radians(th), np.radians(ph))
    main_rays = None
    if args.rays == "images":
        _iw, _ih = 3 * W, 3 * H
        _dfront = torch.empty(_ih * _iw * 4, dtype=torch.uint8, device=dev); _dback = torch.empty_like(_dfront)
        ctx.first_pass_device(_iw, _ih, cam, _dfront.data_ptr(), _dback.data_ptr(), stream)
        main_rays = vv.device_image_rays(_dfront.data_ptr(), _dback.data_ptr(), _iw, _ih, hint=cam)
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
        ctx.render_device(W, H, cam, G.frames[b].data_ptr(), options=o, stream=stream, phong=args.phong, rays=main_rays)
        submit(b)

    # ---- untimed instrumented pass: executed samples + bricks touched (byte model) ----
    nb = (n + BRICK - 1) // BRICK
    vbytes = 4 if args.voxel == "f32" else 1
    rows_owned = len(sharding.owned_rows(H, world, rank))

    def instrumented(camera, phong, rays=None):
        """(executed samples, algorithmic bytes of SURVEY 8d: B_frame) of one frame of this rank"""
        bitmap = torch.zeros((nb * nb * nb + 31) // 32, dtype=torch.int32, device=dev)
        io = vv.make_options(count_samples=True, touched_bricks=bitmap.data_ptr(), **base)
        ctx.render_device(W, H, camera, frame.data_ptr(), options=io, stream=stream, phong=phong, rays=rays)
        torch.cuda.synchronize()
        ns = ctx.last_sample_count()
        words = bitmap.cpu().numpy().view(np.uint32)
        nbricks = int(np.unpackbits(words.view(np.uint8)).sum())
        return ns, nbricks * BRICK ** 3 * vbytes + 4 * W * rows_owned + 4096

    def line_bytes(camera, phong, rays=None, c=None, Wx=None, Hx=None, o=None):
        """The same frame counted at the granularity the memory system fetches at: distinct 128-byte lines of the layout the frame
        samples (a) under its executed in-volume samples (`compulsory`: what must cross the L2's memory side at least once) and (b)
        under every gather the kernel issues, idle lanes and out-of-volume samples included (`issued`); + the frame's own bytes."""
        c = c or ctx; Wx = Wx or W; Hx = Hx or H; ob = dict(base) if o is None else dict(o)
        bits = max(c.device_bytes()[:3]) // 128 + 64
        res = []
        for every in (False, True):
            bm = torch.zeros((bits + 31) // 32, dtype=torch.int32, device=dev)
            io = vv.make_options(touched_lines=bm.data_ptr(), touched_line_bits=bits, touched_lines_all=every, **ob)
            c.render_device(Wx, Hx, camera, frame.data_ptr() if c is ctx else scratch_frame(Wx, Hx).data_ptr(), options=io, stream=stream, phong=phong, rays=rays)
            torch.cuda.synchronize()
            res.append(int(np.unpackbits(bm.cpu().numpy().view(np.uint8)).sum()) * 128 + 4 * Wx * (rows_owned if c is ctx else Hx) + 4096)
            del bm
        out_l = {"compulsory_line_bytes": res[0], "issued_line_bytes": res[1]}
        # ... and per thread block: distinct (block, line) pairs of every gather issued = what the frame fetches if blocks share nothing and a
        # block never loses a line it still needs.  traffic above this figure is re-fetching inside blocks; this figure above `issued` is the
        # overlap between blocks' footprints (partial lines at tile edges, apron rays of the Phong kernel)
        if os.environ.get("VV_BENCH_BLOCK_LINES", "1") != "0":
            lg = 28
            tab = torch.zeros(1 << lg, dtype=torch.int64, device=dev)
            io = vv.make_options(touched_block_lines=tab.data_ptr(), touched_block_lines_log2=lg, touched_lines_all=True, **ob)
            c.render_device(Wx, Hx, camera, frame.data_ptr() if c is ctx else scratch_frame(Wx, Hx).data_ptr(), options=io, stream=stream, phong=phong, rays=rays)
            torch.cuda.synchronize()
            npairs = int((tab != 0).sum().item())
            del tab
            out_l["block_line_bytes"] = npairs * 128 + 4 * Wx * (rows_owned if c is ctx else Hx) + 4096
            if npairs > (1 << lg) * 0.6:
                out_l["block_line_note"] = "hash set more than 60 % full: a lower bound"
        return out_l

    _scratch = {}

    def scratch_frame(Wx, Hx):
        if (Wx, Hx) not in _scratch:
            _scratch[(Wx, Hx)] = torch.zeros(Hx * Wx, dtype=torch.int32, device=dev)
        return _scratch[(Wx, Hx)]

    samples, bytes_rank = instrumented(cam, args.phong, main_rays)
    if main_rays is not None:
        bytes_rank += 8 * W * rows_owned                   # the two texels each pixel's ray is read from
    try:
        lines_main = line_bytes(cam, args.phong, main_rays) if world == 1 else None
    except Exception as e:                      # an instrument beside the metric: never at its cost
        lines_main = {"error": f"{type(e).__name__}: {e}"}
    if os.environ.get("VV_STATS"):      # developer statistics from a counters-only frame (no brick marking)
        ctx.render_device(W, H, cam, frame.data_ptr(), options=vv.make_options(count_samples=True, **base), stream=stream, phong=args.phong)
        torch.cuda.synchronize()
        print("stats", ctx.debug_counters().tolist(), "instrumented frame ms", ctx.last_frame_ms(), file=sys.stderr)
    rdev = torch.device("cpu") if share else dev
    tot = torch.tensor([samples, bytes_rank], dtype=torch.float64, device=rdev)
    if world > 1:
        dist.all_reduce(tot)
    samples_all, bytes_all = float(tot[0]), float(tot[1])

    # ---- device spin-up (untimed, before the W warm-up steps): after idling the GPU needs a few hundred frames to reach
    #      its steady state (MI355X, C3: frames 1-25 after the set-up take 1.05 ms, from about frame 50 on 1.00 ms --
    #      power state and translation caches; measured with --warmup 3 / 50 / 300).  A renderer runs in that steady state;
    #      the timed region is still exactly K complete frames.  VV_BENCH_SPINUP=0 disables it. ----
    # The same K frames are first timed WITHOUT it (after the W warm-up steps only: what a driver calling `--warmup 5` would
    # see from a cold device) and reported as `ms_per_step_first` / `roofline.frac_first`, so both readings are on record.
    spinup = int(os.environ.get("VV_BENCH_SPINUP", "300"))
    cold_ms = None
    # One GPU: the K frames are timed by ONE pair of HIP events around the region (on the stream the kernels are launched on), and vv_render's own
    # per-frame event pair is switched off (vv_set_frame_timing): every event record is a packet the stream retires between two frames, 2-4 us each.
    # N > 1 keeps a pair per frame (the ranks' kernel times are reported one by one).
    if world == 1:
        ctx.set_frame_timing(False)
        for _ in range(args.warmup):
            ctx.render_device(W, H, cam, G.frames[0].data_ptr(), options=opts, stream=stream, phong=args.phong, rays=main_rays)
        ce0, ce1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        ce0.record()
        for k in range(args.steps):
            ctx.render_device(W, H, cam, G.frames[0].data_ptr(), options=opts, stream=stream, phong=args.phong, rays=main_rays)
        ce1.record()
        torch.cuda.synchronize()
        cold_ms = ce0.elapsed_time(ce1) / args.steps
    for _ in range(spinup):
        ctx.render_device(W, H, cam, G.frames[0].data_ptr(), options=opts, stream=stream, phong=args.phong, rays=main_rays)
    torch.cuda.synchronize()

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
    if world == 1:
        ev[0][0].record()
        for k in range(args.steps):
            ctx.render_device(W, H, cam, G.frames[k & 1].data_ptr(), options=opts, stream=stream, phong=args.phong, rays=main_rays)
        ev[0][1].record()
    else:
        for k in range(args.steps):
            b = k & 1
            G.finish(b)
            ev[k][0].record()
            ctx.render_device(W, H, cam, G.frames[b].data_ptr(), options=opts, stream=stream, phong=args.phong, rays=main_rays)
            ev[k][1].record()
            submit(b)
    G.drain()                                        # every frame is complete on rank 0 inside the timed region
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t1 = time.perf_counter()
    frame = G.frames[(args.steps - 1) & 1]
    # N > 1: the gathered frame is checked (untimed) against the same frame rendered unsharded on rank 0 -- a throughput
    # number from a gather that scrambled or dropped bands would be worthless
    gather_check = "n/a (single GPU)"
    if world > 1 and rank == 0:
        whole = torch.full_like(frame, 0)
        whole.copy_(frame)                                  # pixels the frame never writes (column W-1, row H-1) compare equal
        wo = vv.make_options(**{k: v for k, v in base.items() if k != "shard"})
        ctx.render_device(W, H, cam, whole.data_ptr(), options=wo, stream=stream, phong=args.phong)
        torch.cuda.synchronize()
        nbad = int((whole[:H] != frame[:H]).any(dim=-1).sum().item())
        gather_check = "ok" if nbad == 0 else f"mismatch ({nbad} of {H * W} pixels)"
    el = torch.tensor([t1 - t0], dtype=torch.float64, device=rdev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el[0])
    # rad + march kernels of this rank, per frame (one GPU: the region's HIP-event time / K, frame to frame)
    kern_ms = ev[0][0].elapsed_time(ev[0][1]) / args.steps if world == 1 else float(np.mean([a.elapsed_time(b) for a, b in ev]))
    kern_all = torch.tensor([kern_ms], dtype=torch.float64, device=rdev)
    if world > 1:
        gl = [torch.zeros_like(kern_all) for _ in range(world)]
        dist.all_gather(gl, kern_all)
        kern_ranks = [round(float(g[0]), 4) for g in gl]
        backend, nranks = dist.get_backend(), dist.get_world_size()
    else:
        kern_ranks, backend, nranks = [round(kern_ms, 4)], None, 1

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
    key = f"{args.config}-{args.volume}-{args.tf}-{args.view}{'-phong' if args.phong else ''}{'-u8' if args.voxel == 'u8' else ''}{'-images' if main_rays is not None else ''}-n{world}"
    traffic, traffic_note = (None, "diagnostic camera") if args.orbit else measured_traffic(key)
    if traffic is None and traffic_note is None:
        traffic_note = f"no PMC passes committed for {key}"
    out = {
        "metric": "Msamples/s (rays x steps), 1024^3 f32 volume @1080p" if (args.config == "c3" and n == 1024 and args.voxel == "f32")
                  else f"Msamples/s (rays x steps), {n}^3 {args.voxel} volume @{W}x{H} (diagnostic configuration, not BASELINE.json's metric)",
        "value": round(value, 1),
        "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.voxel, "data": "synthetic",
        "config": {"workload": f"{args.config.upper()}: {n}^3 {args.voxel} {args.volume} volume, {W}x{H}, step 1/{steps}, "
                               f"{args.tf} RGBA TF, ERT {args.ert}, {args.filter} filter, view {args.orbit or args.view}" + (", Phong" if args.phong else ""),
                   "volume": [n, n, n], "frame": [W, H], "steps_per_unit_length": steps,
                   "sharding": "single GPU" if world == 1 else f"bands of {sharding.BAND_PX} pixel rows round-robin over {world} GPUs, volume replicated, 1 RCCL gather/frame"},
        "executed_samples_per_frame": int(samples_all), "upper_bound_samples_WxHxS": W * H * steps,
        "spinup_frames": spinup, "kernel_ms_rank0": round(kern_ms, 4), "kernel_ms_per_rank": kern_ranks,
        "ms_per_step_first": None if cold_ms is None else round(cold_ms, 4),
        "collective": None if world == 1 else {"backend": backend, "ranks": nranks, "per_frame": "1 gather of RGBA8 bands to rank 0"},
        "gather_check": gather_check,
        "roofline": {"bound": "hbm", "kernel": ("march_phong_kernel" if args.phong else "march_kernel (+rad_kernel)"), "achieved": round(achieved / 1e9, 1),
                     "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": round(achieved / HBM_PEAK, 4),
                     "frac_first": None if cold_ms is None else round(bytes_rank / (cold_ms * 1e-3) / HBM_PEAK, 4),
                     "protocol": f"frac: {args.steps} frames after {spinup} untimed spin-up frames + {args.warmup} warm-up (steady state); "
                                 f"frac_first: the {args.steps} frames right after the {args.warmup} warm-up frames (device still ramping up)",
                     "traffic": traffic, "algorithmic_bytes_per_launch": int(bytes_rank),
                     "bytes_per_sample": round(bytes_rank / max(samples, 1), 3)},
    }
    if lines_main:
        out["roofline"].update(lines_main)
        if traffic and "compulsory_line_bytes" in lines_main:
            out["roofline"]["traffic_over_compulsory_lines"] = round(traffic / lines_main["compulsory_line_bytes"], 3)
    # `traffic` counts requests at the L2's memory side; Infinity-Cache (MALL) hits are inside it and no gfx950 counter separates them
    # (TCC_EA0_RDREQ_DRAM == TCC_EA0_RDREQ on streams that are certainly MALL hits: profiles/r05_mall_reread.txt).  The same file shows why the split
    # does not matter for this kernel: a re-read served by the MALL costs what a DRAM read costs on that path (6.6 vs 6.2-6.4 TB/s), only an L2 hit is cheap.
    out["roofline"]["traffic_hbm"] = None
    out["roofline"]["traffic_hbm_note"] = ("not separable by counters on gfx950 (TCC_EA0_RDREQ_DRAM equals TCC_EA0_RDREQ on certain MALL hits); "
                                           "re-reads served by the Infinity Cache stream at 6.6-6.8 TB/s against 6.2-6.4 from DRAM (profiles/r05_mall_reread.txt): `traffic` is the figure that bounds the kernel")
    out["roofline"]["traffic_source"] = (f"committed PMC pass (profiles/pmc_traffic.json, taken with csrc {csrc_sha()}: FETCH_SIZE x calibrated factor + WRITE_SIZE, tools/pmc_traffic.py); "
                                         "not measured inside this run") if traffic is not None else "none"
    if traffic_note:
        out["roofline"]["traffic_note"] = traffic_note
    if upload is not None:
        out["volume_upload"] = upload

    # Not part of `value`: the other shipped march kernels on the same workload, each with its own algorithmic bytes
    # (instrumented, untimed pass) and roofline fraction: the camera off the memory axis (SURVEY 8d's second camera:
    # vv_render samples the bricked copy) and the Phong-shaded frame (march_phong_kernel).
    plain_c3 = world == 1 and args.config == "c3" and args.view == "a" and not args.orbit and not args.phong and args.voxel == "f32" and main_rays is None and not args.size
    if plain_c3 and not os.environ.get("VV_BENCH_NO_EXTRA"):
        def timed(camera, phong, reps=20, rays=None):
            for _ in range(60):           # steady state on the other layout / kernel (see the spin-up above)
                ctx.render_device(W, H, camera, frame.data_ptr(), options=opts, stream=stream, phong=phong, rays=rays)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                ctx.render_device(W, H, camera, frame.data_ptr(), options=opts, stream=stream, phong=phong, rays=rays)
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps

        cam_b = vv.Camera.orbit(4.0, np.pi / 3, np.pi / 5)
        # What a frame costs when the copy of the volume its view samples is not resident yet (vv_render then builds it: hipMalloc + one kernel over
        # the volume + a stream synchronisation), and what loading with vv_prepare_layouts(VV_LAYOUT_POLICY) -- the mirror's cudaLoadVolume does that --
        # costs instead.  Wall clock around the call + a device synchronisation.
        try:
            torch.cuda.synchronize()
            st0 = ctx.layout_state()
            t0 = time.perf_counter()
            ctx.render_device(W, H, cam_b, frame.data_ptr(), options=opts, stream=stream)
            torch.cuda.synchronize()
            first_b = (time.perf_counter() - t0) * 1e3
            t0 = time.perf_counter()
            ctx.prepare_layouts(vv.LAYOUT_POLICY)
            torch.cuda.synchronize()
            prep = (time.perf_counter() - t0) * 1e3
            ctx.set_layout_policy(0, False)                    # from here on no frame builds anything
            st1 = ctx.layout_state()
            out["residency"] = {"what": "device memory of the C3 context: linear volume + the optional copies (include/volviz.h); budget = what the copies together may take",
                                "bytes": {k: st1[k] for k in ("linear", "bricked", "zpair", "zfast", "xpair", "budget")},
                                "resident_before_the_first_oblique_frame": {k: st0[k] for k in ("bricked", "zfast")},
                                "first_frame_ms_oblique_view_copy_not_resident": round(first_b, 3),
                                "prepare_remaining_layouts_ms": round(prep, 3),
                                "note": "a host that calls vv_prepare_layouts(VV_LAYOUT_POLICY) at load (the mirror's cudaLoadVolume / PaintLoop::loadVolume do) never sees the first figure"}
        except Exception as e:
            out["residency"] = {"error": f"{type(e).__name__}: {e}"}
        # The view-robust number: C3 over a fixed orbit of 12 cameras on the reference's orbit (glwidget.cpp:435-445: radius 4, up = +y), polar angles
        # 30 / 60 / 90 degrees x azimuths -90 (the headline's axis), -54, 0 (side view), 36 degrees.
        try:
            per = []
            for th in (30.0, 60.0, 90.0):
                for ph in (-90.0, -54.0, 0.0, 36.0):
                    co = vv.Camera.orbit(4.0, np.radians(th), np.radians(ph))
                    ns_o, by_o = instrumented(co, False)
                    ms_o = timed(co, False, reps=15)
                    per.append({"theta": th, "phi": ph, "ms": round(ms_o, 4), "Msamples": round(ns_o / 1e6, 1), "frac": round(by_o / (ms_o * 1e-3) / HBM_PEAK, 4),
                                "layout": ctx.last_launch()["layout"]})
            out["orbit"] = {"what": "C3 over 12 cameras of the reference's orbit (radius 4; theta 30 / 60 / 90 deg x phi -90 / -54 / 0 / 36 deg); each with its own algorithmic bytes; layout 1 = linear, 2 = bricked, 4 = z-fastest",
                            "mean_ms": round(float(np.mean([q["ms"] for q in per])), 4), "worst_ms": max(q["ms"] for q in per),
                            "mean_frac": round(float(np.mean([q["frac"] for q in per])), 4), "worst_frac": min(q["frac"] for q in per), "per_camera": per,
                            "builds_inside_frames": ctx.layout_state()["builds_in_render"] - st1["builds_in_render"]}
        except Exception as e:
            out["orbit"] = {"error": f"{type(e).__name__}: {e}"}
        for name, camera, phong, what, tkey in (
                ("rotated_view", cam_b, False, "camera on the orbit r=4, theta=60 deg, phi=36 deg: march_kernel on the bricked copy", "c3-noise-ramp-b-n1"),
                ("side_view", vv.Camera(origin=(-4.0, 0.0, 0.0)), False, "camera on the -x axis (screen x along the volume's z): march_kernel on the z-fastest copy", "c3-noise-ramp-side-n1"),
                ("phong", cam, True, "view a with central-difference gradient + Phong: march_phong_kernel", "c3-noise-ramp-a-phong-n1")):
            try:
                ns_x, by_x = instrumented(camera, phong)
                ms_x = timed(camera, phong)
            except Exception as e:          # a diagnostic beside the metric: never at the cost of the metric line
                out[name] = {"what": what, "error": f"{type(e).__name__}: {e}"}
                continue
            tr_x, note_x = measured_traffic(tkey)
            out[name] = {"what": what, "ms_per_frame": round(ms_x, 4), "value": round(ns_x / ms_x / 1e3, 1), "unit": "Msamples/s",
                         "executed_samples_per_frame": int(ns_x),
                         "roofline": {"bound": "hbm", "achieved": round(by_x / (ms_x * 1e-3) / 1e9, 1), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                                      "frac": round(by_x / (ms_x * 1e-3) / HBM_PEAK, 4), "traffic": tr_x, "algorithmic_bytes_per_launch": int(by_x)}}
            if note_x:
                out[name]["roofline"]["traffic_note"] = note_x
            try:
                out[name]["roofline"].update(line_bytes(camera, phong))
            except Exception as e:
                out[name]["roofline"]["line_bytes_error"] = f"{type(e).__name__}: {e}"

        # The call the reference's host makes: runCuda marches from the two first-pass images, drawn at three times the render size
        # (glwidget.cpp:291,358; kernel.cu:317-321).  Images resident in HBM (vv_first_pass on the device), the camera that drew them
        # passed as the launch-policy hint; the images' sampled texels (8 B per pixel) join the algorithmic bytes.
        try:
            iw, ih = 3 * W, 3 * H
            dfront = torch.empty(ih * iw * 4, dtype=torch.uint8, device=dev); dback = torch.empty_like(dfront)
            ctx.first_pass_device(iw, ih, cam, dfront.data_ptr(), dback.data_ptr(), stream)
            irays = vv.device_image_rays(dfront.data_ptr(), dback.data_ptr(), iw, ih, hint=cam)
            ns_i, by_i = instrumented(cam, False, irays)
            by_i += 8 * W * rows_owned
            ms_i = timed(cam, False, rays=irays)
            ll = ctx.last_launch()
            ms_q = timed(cam, False, rays=vv.analytic_rays(cam, quantize8=True))     # the same rays without the images: end points rounded to RGBA8
            out["images_path"] = {"what": f"view a marched from two {iw}x{ih} RGBA8 first-pass images resident in HBM (VV_RAYS_IMAGES, the runCuda-shaped call) with the camera as launch hint",
                                  "ms_per_frame": round(ms_i, 4), "value": round(ns_i / ms_i / 1e3, 1), "unit": "Msamples/s", "executed_samples_per_frame": int(ns_i),
                                  "vs_analytic_rays": round(ms_i / (kern_ms if kern_ms > 0 else float("nan")), 4),
                                  "same_rays_without_images_ms": round(ms_q, 4), "vs_same_rays_without_images": round(ms_i / ms_q, 4),
                                  "note": "the 8-bit end points of the first-pass contract (firstpass.frag:4) bunch neighbouring rays on a 1/255 grid: that, not the image fetch, is the difference to analytic rays",
                                  "launch": {k: ll[k] for k in ("tile_log2w", "blk_log2w", "unroll", "lds_reserve", "layout", "view_known")},
                                  "roofline": {"bound": "hbm", "achieved": round(by_i / (ms_i * 1e-3) / 1e9, 1), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                                               "frac": round(by_i / (ms_i * 1e-3) / HBM_PEAK, 4), "traffic": measured_traffic("c3-noise-ramp-a-images-n1")[0], "algorithmic_bytes_per_launch": int(by_i)}}
            del dfront, dback
        except Exception as e:
            out["images_path"] = {"error": f"{type(e).__name__}: {e}"}

    # ---- sub-records beside the metric (never at its cost): the other single-GPU configurations of BASELINE.json and the other kernel
    #      families of the path, each on its own context so that the C3 volume above stays what `value` was measured on ----
    def sub_timed(c, Wx, Hx, camera, o, phong, reps, warm=30):
        fr = torch.zeros(Hx * Wx, dtype=torch.int32, device=dev)
        for _ in range(warm):
            c.render_device(Wx, Hx, camera, fr.data_ptr(), options=o, stream=stream, phong=phong)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            c.render_device(Wx, Hx, camera, fr.data_ptr(), options=o, stream=stream, phong=phong)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps, fr

    def sub_instrumented(c, Wx, Hx, camera, stepx, phong, nvol, vb, fr):
        nbx = (nvol + BRICK - 1) // BRICK
        bm = torch.zeros((nbx ** 3 + 31) // 32, dtype=torch.int32, device=dev)
        c.render_device(Wx, Hx, camera, fr.data_ptr(), options=vv.make_options(step=stepx, count_samples=True, touched_bricks=bm.data_ptr()), stream=stream, phong=phong)
        torch.cuda.synchronize()
        nsx = c.last_sample_count()
        nbr = int(np.unpackbits(bm.cpu().numpy().view(np.uint8)).sum())
        return nsx, nbr * BRICK ** 3 * vb + 4 * Wx * Hx + 4096

    def pmc_sub(key):
        pj = os.path.join(REPO, "profiles", "pmc_sub.json")
        try:
            j = json.load(open(pj))
            return j.get("entries", {}).get(key), (None if j.get("csrc_sha") == csrc_sha() else f"committed with csrc {j.get('csrc_sha')}, this tree is {csrc_sha()}")
        except Exception as e:
            return None, f"no profiles/pmc_sub.json ({type(e).__name__})"

    def sub_records():
        if not (plain_c3 and not os.environ.get("VV_BENCH_NO_EXTRA")):
            return
        cam0 = vv.Camera()
        # -- C2 (BASELINE.json configs[1]): 256^3 f32, 1280x720, step 1/256, grey table (Head).  Lives in the caches: not HBM-bound;
        #    reported as Msamples/s with the VALU-issue fraction of a committed PMC pass
        try:
            c2 = vv.Context(local); c2.set_frame_timing(False)
            n2, W2, H2 = 256, 1280, 720
            a8 = torch.empty(n2 ** 3, dtype=torch.uint8, device=dev); c2.generate_noise_device(a8.data_ptr(), n2, n2, n2, 0x9E3779B9, stream)
            a32 = torch.empty(n2 ** 3, dtype=torch.float32, device=dev); c2.promote_device(a8.data_ptr(), a32.data_ptr(), n2 ** 3, stream)
            c2.load_volume_device(a32.data_ptr(), vv.VOXEL_F32, n2, n2, n2, vv.transfer_preset(vv.TF_HEAD), stream); torch.cuda.synchronize()
            o2 = vv.make_options(step=1.0 / 256)
            ms2, fr2 = sub_timed(c2, W2, H2, cam0, o2, False, 100, warm=100)
            ns2, by2 = sub_instrumented(c2, W2, H2, cam0, 1.0 / 256, False, n2, 4, fr2)
            pm, pnote = pmc_sub("c2")
            out["c2"] = {"what": "BASELINE config C2: 256^3 f32 noise volume, 1280x720, step 1/256, Head (grey) table, view a: march_kernel on the z-pair copy; cache-resident, not HBM-bound",
                         "ms_per_frame": round(ms2, 4), "value": round(ns2 / ms2 / 1e3, 1), "unit": "Msamples/s", "executed_samples_per_frame": int(ns2),
                         "launch": c2.last_launch(), "algorithmic_bytes_per_launch": int(by2),
                         "valu_issue_fraction": None if pm is None else pm.get("valu_issue_fraction"), "lds_issue_fraction": None if pm is None else pm.get("lds_issue_fraction"),
                         "pmc_source": "committed PMC pass (profiles/pmc_sub.json: SQ_ACTIVE_INST_VALU / (32 x GRBM_GUI_ACTIVE), tools/pmc_sub.py)" + (f"; {pnote}" if pnote else "")}
            # -- slice sampler (slice_kernel, kernel.cu:543-644) on the same volume: 256^2 (the reference's size, params.h:17) and 1024^2, device output
            import ctypes as C
            sl = {}
            sc = (C.c_float * 3)(1.0, 1.0, 1.0)
            for hw in (256, 1024):
                buf = torch.zeros(hw * hw, dtype=torch.float32, device=dev)
                def call():
                    c2._chk(c2.lib.vv_slice(c2.h, buf.data_ptr(), hw, hw, 0.1, 0.2, 0.3, vv.CORONAL, C.byref(sc), 0, vv.FILTER_TEX8, 1, stream))
                for _ in range(20): call()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(100): call()
                e1.record(); torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 10.0
                sl[f"{hw}x{hw}"] = {"us_per_slice": round(us, 2), "Msamples_per_s": round(hw * hw / us, 1), "bytes_written": 4 * hw * hw,
                                    "bound": "launch latency" if us < 10 else "gather"}
            out["slice"] = {"what": "vv_slice (coronal plane, TEX8 filter) of the 256^3 volume into a device buffer, enqueue-only calls back to back", **sl}
            c2.close(); del a8, a32, fr2
        except Exception as e:
            out["c2"] = {"error": f"{type(e).__name__}: {e}"}
        # -- the reference's own voxel type (kernel.cu:46,459: u8; f32 volumes are this build's extension): C3's frame on the 1024^3 volume as u8
        try:
            cu = vv.Context(local); cu.set_frame_timing(False)
            nu = 1024
            u8v = torch.empty(nu ** 3, dtype=torch.uint8, device=dev); cu.generate_noise_device(u8v.data_ptr(), nu, nu, nu, 0x9E3779B9, stream)
            cu.load_volume_device(u8v.data_ptr(), vv.VOXEL_U8, nu, nu, nu, tf, stream); torch.cuda.synchronize()
            ou = vv.make_options(step=1.0 / 512)
            msu, fru = sub_timed(cu, W, H, cam0, ou, False, 20, warm=60)
            nsu, byu = sub_instrumented(cu, W, H, cam0, 1.0 / 512, False, nu, 1, fru)
            pmu, pnoteu = pmc_sub("u8")
            out["u8_volume"] = {"what": "C3's frame and step on the same noise volume stored as u8, the reference's voxel type (1 GiB): march_kernel on the z-pair copy",
                                "ms_per_frame": round(msu, 4), "value": round(nsu / msu / 1e3, 1), "unit": "Msamples/s", "executed_samples_per_frame": int(nsu), "launch": cu.last_launch(),
                                # 0.9 GB in 0.6 ms is 1.5 TB/s: this frame is not bound by HBM.  It issues vector instructions most of the time (the bytes of a sample are unpacked
                                # and converted in the VALU): the bound is VALU issue, reported as the fraction of issue slots used (committed PMC pass, tools/pmc_sub.py)
                                "roofline": {"bound": "valu", "achieved": None if pmu is None else pmu.get("valu_issue_fraction"), "peak": 1.0, "unit": "fraction of VALU issue cycles",
                                             "frac": None if pmu is None else pmu.get("valu_issue_fraction"),
                                             "wave_wait_fraction": None if pmu is None else pmu.get("wave_wait_fraction"),
                                             "hbm_frac": round(byu / (msu * 1e-3) / HBM_PEAK, 4), "traffic": measured_traffic("c3-noise-ramp-a-u8-n1")[0],
                                             "algorithmic_bytes_per_launch": int(byu),
                                             "pmc_source": "committed PMC pass (profiles/pmc_sub.json, tools/pmc_sub.py)" + (f"; {pnoteu}" if pnoteu else "")}}
            cu.close(); del u8v, fru
        except Exception as e:
            out["u8_volume"] = {"error": f"{type(e).__name__}: {e}"}
        # -- generator (drawDefaultBrain, volumegenerator.cpp:100-119): HIP against the CPU restatement on one thread (as the reference is), 128^3 and 1024^3
        try:
            cg = vv.Context(local); cg.set_frame_timing(False)
            gen = {}
            for ng in (128, 1024):
                g8 = torch.empty(ng ** 3, dtype=torch.uint8, device=dev)
                for _ in range(3): cg.generate_default_brain_device(g8.data_ptr(), ng, ng, ng, stream)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10): cg.generate_default_brain_device(g8.data_ptr(), ng, ng, ng, stream)
                e1.record(); torch.cuda.synchronize()
                msg = e0.elapsed_time(e1) / 10
                gen[f"{ng}^3"] = {"hip_ms": round(msg, 4), "hip_GB_per_s_written": round(ng ** 3 / msg / 1e6, 1), "frac_of_hbm_peak": round(ng ** 3 / (msg * 1e-3) / HBM_PEAK, 4)}
                del g8
            if want_cpu:
                sys.path.insert(0, os.path.join(REPO, "tests"))
                import oracle_lib as O
                t = time.perf_counter(); O.draw_default_brain(128, 128, 128); d128 = time.perf_counter() - t
                t = time.perf_counter(); O.draw_default_brain(1024, 1024, 16); d16 = time.perf_counter() - t
                gen["128^3"].update({"cpu_port_1thread_ms": round(d128 * 1e3, 1), "speedup": round(d128 * 1e3 / gen["128^3"]["hip_ms"], 1)})
                gen["1024^3"].update({"cpu_port_1thread_ms_extrapolated": round(d16 * 64 * 1e3, 0), "speedup": round(d16 * 64 * 1e3 / gen["1024^3"]["hip_ms"], 0),
                                      "cpu_sample": f"16 slices of 1024 x 1024 voxels ({d16:.2f} s; every drawEllipsoid visits every voxel, so the cost per voxel does not depend on the slice) x 64"})
            out["generator"] = {"what": "vv_generate_default_brain (8 ellipsoids fused; 1024^3: ellipsoid_rows_kernel, 128^3: ellipsoid_kernel = launch latency) into a device buffer; bound at 1024^3: VALU issue 68 % (profiles/r04_generator.txt), write roofline = HBM peak", **gen}
            cg.close()
        except Exception as e:
            out["generator"] = {"error": f"{type(e).__name__}: {e}"}
        # -- C5 (BASELINE.json configs[4]) on one GPU: 2048^3 f32 streamed from one pinned host buffer, gradient + Phong, step 1/2048
        try:
            ctx.load_volume(np.zeros((4, 4, 4), np.uint8), tf)          # release the 4 GiB C3 volume first
            torch.cuda.empty_cache()
            c5 = vv.Context(local); c5.set_frame_timing(False)
            n5, W5, H5, per = 2048, 1920, 1080, 32
            b8 = torch.empty(n5 ** 3, dtype=torch.uint8, device=dev); c5.generate_noise_device(b8.data_ptr(), n5, n5, n5, 0x9E3779B9, stream); torch.cuda.synchronize()
            pin = torch.empty((per, n5, n5), dtype=torch.uint8).pin_memory()
            host8 = np.empty((n5, n5, n5), np.uint8)
            bv = b8.view(n5, n5, n5)
            for z0 in range(0, n5, per):
                pin.copy_(bv[z0:z0 + per]); host8[z0:z0 + per] = pin.numpy()
            del bv, b8; torch.cuda.empty_cache()

            def slabs5():
                for z0 in range(0, n5, per):
                    pin.numpy()[...] = host8[z0:z0 + per]
                    yield z0, pin.numpy()
            t0u = time.perf_counter()
            c5.load_volume_streamed(slabs5(), vv.VOXEL_F32, n5, n5, n5, tf); torch.cuda.synchronize()
            up = time.perf_counter() - t0u
            del host8, pin
            o5 = vv.make_options(step=1.0 / 2048)
            ms5, fr5 = sub_timed(c5, W5, H5, cam0, o5, True, 10, warm=10)
            ns5, by5 = sub_instrumented(c5, W5, H5, cam0, 1.0 / 2048, True, n5, 4, fr5)
            out["c5"] = {"what": "BASELINE config C5 on one GPU: 2048^3 f32 noise volume (32 GiB) streamed as u8 slabs of 32 slices through one pinned host buffer and promoted on the device, "
                                 "1920x1080, step 1/2048, colour-ramp table, central-difference gradient + Phong, view a: march_phong_kernel (64-bit addressing build)",
                         "ms_per_frame": round(ms5, 4), "value": round(ns5 / ms5 / 1e3, 1), "unit": "Msamples/s", "executed_samples_per_frame": int(ns5),
                         "upload_seconds": round(up, 3), "upload_GB_per_s": round(n5 ** 3 / up / 1e9, 2),
                         "upload_note": "includes this script's single-threaded host memcpy of every 128 MiB slab into the one pinned buffer (about two thirds of it); the library's streamer alone: tools/time_upload.py",
                         "roofline": {"bound": "hbm", "achieved": round(by5 / (ms5 * 1e-3) / 1e9, 1), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                                      "frac": round(by5 / (ms5 * 1e-3) / HBM_PEAK, 4), "traffic": measured_traffic("c5-noise-ramp-a-phong-n1")[0], "algorithmic_bytes_per_launch": int(by5)},
                         "device_bytes": c5.layout_state()}
            c5.close(); del fr5
        except Exception as e:
            out["c5"] = {"error": f"{type(e).__name__}: {e}"}

    def cpu_baselines():
        if want_cpu:
            # The CPU restatement (oracle/vvo.c, OpenMP) on this box's host cores (the GPU box shares its host: 16 cores is
            # the share of one GPU).  BASELINE.md section 3 asks for config C1: 128^3 drawDefaultBrain volume (u8, as the
            # reference stores it), 512x512, step 1/128, Head transfer function, camera (0,0,-4); 1 thread and all threads.
            sys.path.insert(0, os.path.join(REPO, "tests"))
            import oracle_lib as O
            cores = int(os.environ.get("VV_CPU_THREADS", min(len(os.sched_getaffinity(0)), 16)))
            c1_vol = O.draw_default_brain(128, 128, 128)
            c1_tf = vv.transfer_preset(vv.TF_HEAD)
            c1_cam = vv.Camera()
            budget = max(2.0, args.cpu_seconds * 0.5)
            t = time.perf_counter(); sN = 0; reps = 0
            while reps == 0 or (time.perf_counter() - t < budget and reps < 4096):
                sN += O.render(c1_vol, c1_tf, 512, 512, c1_cam, threads=cores)[1]; reps += 1
            dt = time.perf_counter() - t
            out["cpu_baseline"] = {"value": round(sN / dt / 1e6, 2), "unit": "Msamples/s", "cores": cores, "kind": "port",
                                   "workload": "C1 (BASELINE.md section 3) -- NOT the workload of `value` (C3); the port on C3 itself is `cpu_baseline_c3`",
                                   "ms_per_frame": round(dt / reps * 1e3, 2),
                                   "sample": f"config C1: {reps} x the whole 512x512 frame of the 128^3 drawDefaultBrain volume, Head TF, step 1/128 "
                                             f"({sN} samples in {dt:.1f} s; oracle/vvo.c, OpenMP, {cores} threads)"}
            t = time.perf_counter()
            s1 = O.render(c1_vol, c1_tf, 512, 512, c1_cam, threads=1)[1]
            d1 = time.perf_counter() - t
            out["cpu_baseline_1thread"] = {"value": round(s1 / d1 / 1e6, 2), "unit": "Msamples/s", "cores": 1, "kind": "port",
                                           "ms_per_frame": round(d1 * 1e3, 1), "sample": f"config C1, one frame ({s1} samples in {d1:.1f} s)"}
        if want_cpu and host_vol is not None:
            # the same port on the bench workload itself (C3), a slab-row band sized to the remaining time
            nby = (H + 13) // 14
            mid = nby // 2
            cbase = {k: v for k, v in base.items() if k != "shard"}
            t = time.perf_counter()
            O.render(host_vol, tf, W, H, cam, options=vv.make_options(slab_rows=(mid, mid + 1), **cbase), threads=cores)
            dt1 = time.perf_counter() - t
            rows = int(max(1, min(nby, budget / max(dt1, 1e-3))))
            lo = max(0, min(nby - rows, mid - rows // 2))
            t = time.perf_counter()
            sN = O.render(host_vol, tf, W, H, cam, options=vv.make_options(slab_rows=(lo, lo + rows), **cbase), threads=cores)[1]
            dt = time.perf_counter() - t
            out["cpu_baseline_c3"] = {"value": round(sN / dt / 1e6, 2), "unit": "Msamples/s", "cores": cores, "kind": "port",
                                      "sample": f"slab rows [{lo},{lo + rows}) of {nby} of the bench frame itself ({sN} samples in {dt:.1f} s)"}

    try:
        sub_records()
    except Exception as e:
        out["sub_records_error"] = f"{type(e).__name__}: {e}"
    try:
        cpu_baselines()
    except Exception as e:          # the baseline is a report beside the metric: it must never cost the metric line
        out["cpu_baseline"] = {"value": None, "unit": "Msamples/s", "cores": 0, "kind": "port", "sample": f"not taken: {type(e).__name__}: {e}"}
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
