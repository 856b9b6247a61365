"""N > 1 host path on CPU: two gloo ranks each fill their bands of a frame (rendered here by
the oracle, standing in for the GPU kernel, with the same shard predicate), then run the
product's gather / re-assembly code (volviz_amd.sharding) exactly as bench.py does."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as O
import volviz_amd as vv
from volviz_amd import sharding


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, W, H, phong, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    vol = O.draw_default_brain(24, 24, 24)
    tf = O.transfer_preset(vv.TF_ENGINE)
    cam = vv.Camera.orbit(4.0, 1.0, 0.6)
    hp = sharding.padded_height(H, world)
    frame = np.full((hp, W, 4), 7, np.uint8)
    O.render(vol, tf, W, H, cam, phong=phong, options=vv.make_options(shard=sharding.shard_option(world, rank)),
             out=frame[:H])
    t = torch.from_numpy(frame)
    out = sharding.gather_frame(t, world, rank)
    if rank == 0:
        q.put(out.numpy()[:H].copy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("W,H,phong", [(60, 130, False), (45, 43, True)])
def test_two_rank_gather_reassembles_frame(world, W, H, phong):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, W, H, phong, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    vol = O.draw_default_brain(24, 24, 24)
    want, _ = O.render(vol, O.transfer_preset(vv.TF_ENGINE), W, H, vv.Camera.orbit(4.0, 1.0, 0.6), phong=phong, fill=7)
    assert np.array_equal(got, want)


def test_band_bookkeeping():
    for H in (1, 14, 56, 57, 1080, 2160):
        for world in (1, 2, 4, 8):
            rows = [sharding.owned_rows(H, world, r) for r in range(world)]
            assert sorted(sum(rows, [])) == list(range(H))
            assert sharding.padded_height(H, world) % (sharding.BAND_PX * world) == 0
            assert sharding.padded_height(H, world) >= H
    f = torch.arange(2 * 56 * 3 * 4, dtype=torch.uint8).reshape(2 * 56, 3, 4)
    assert torch.equal(sharding.compact(f, 2, 1)[0], f[56:112])


def _stream_worker(rank, world, port, W, H, n_frames, q):
    """bench.py's frame loop with FrameGatherer: frame k is 'rendered' as a function of (k, row)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = sharding.FrameGatherer(H, W, world, rank)
    rows = sharding.owned_rows(H, world, rank)
    done = []
    for k in range(n_frames):
        b = k & 1
        f = g.finish(b)
        if rank == 0 and f is not None:
            done.append(f[:H].clone().numpy())
        g.frames[b][rows] = torch.tensor([[(k * 37 + y) % 251 for _ in range(4)] for y in rows], dtype=torch.uint8)[:, None, :]
        g.submit(b)
    order = [(n_frames - 2 + i) & 1 for i in range(2)] if n_frames >= 2 else [0]
    for b in order:                                     # oldest outstanding frame first
        f = g.finish(b)
        if rank == 0 and f is not None:
            done.append(f[:H].clone().numpy())
    if rank == 0:
        q.put(done)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_frames", [(2, 5), (3, 4), (2, 1)])
def test_async_double_buffered_gather(world, n_frames):
    W, H = 9, 130
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_stream_worker, args=(r, world, port, W, H, n_frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    done = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert len(done) == n_frames
    for k, f in enumerate(done):
        want = np.array([[(k * 37 + y) % 251] * 4 for y in range(H)], np.uint8)[:, None, :].repeat(W, axis=1)
        assert np.array_equal(f, want), k


def test_native_gather_band_bookkeeping():
    """vv_mgpu_band_rows (host arithmetic of the native RCCL gather, include/volviz_mgpu.h): for every band the rank and
    the pixel rows it sends are exactly the rows vv_render's shard predicate lets that rank write -- rows of the slab-row
    range, never row H-1 (kernel.cu:297-298) -- so the landing-frame copy can never touch a pixel the rank did not render."""
    import ctypes as C
    repo = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    lib = C.CDLL(os.path.join(repo, "volume-viz_amd", "lib", "libvolviz_mgpu.so"))
    f = lib.vv_mgpu_band_rows
    f.argtypes = [C.c_int] * 5 + [C.POINTER(C.c_int)] * 3
    for H in (2, 14, 15, 29, 43, 56, 57, 113, 300, 1080, 2160):
        nby = H // 14 + (1 if H % 14 else 0)
        for n in (1, 2, 3, 4, 8):
            for rb, re in ((0, 0), (0, nby), (1, max(1, nby - 1)), (nby // 2, nby)):
                if rb > re or re > nby:
                    continue
                lo, hi = (0, nby) if (rb, re) == (0, 0) else (rb, re)
                want = {r: set() for r in range(n)}
                for y in range(H - 1):
                    s = y // 14
                    if lo <= s < hi:
                        want[(s // 4) % n].add(y)
                got = {r: set() for r in range(n)}
                nbands = (H + 55) // 56
                for b in range(nbands):
                    rk, ya, yb = C.c_int(), C.c_int(), C.c_int()
                    assert f(H, n, rb, re, b, C.byref(rk), C.byref(ya), C.byref(yb)) == 0
                    assert 0 <= ya.value <= yb.value <= H - 1 and rk.value == b % n
                    got[rk.value] |= set(range(ya.value, yb.value))
                assert got == want, (H, n, rb, re)
                rk, ya, yb = C.c_int(), C.c_int(), C.c_int()
                assert f(H, n, rb, re, nbands, C.byref(rk), C.byref(ya), C.byref(yb)) != 0      # band out of range


def _bench(args, env_extra, timeout=600):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra)
    repo = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    return subprocess.run([sys.executable, os.path.join(repo, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (the shape of the driver's N = 1 command): bench.py starts the ranks itself, as a
    child torch.distributed.run, before touching a GPU.  Rehearsed here without a device (VV_BENCH_DRYRUN=1: rendezvous on 127.0.0.1, one gloo
    collective, then stop where device selection would begin): rank 0 prints one JSON line for N ranks with the N-GPU frame."""
    import json
    r = _bench(["--gpus", "2", "--steps", "2", "--warmup", "1"], {"VV_BENCH_DRYRUN": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["collective"]["ranks"] == 2 and j["all_reduce_ok"] and j["frame"] == [2716, 1528]


def test_bench_reports_a_failed_rank():
    """A rank that dies makes the whole command fail (non-zero exit, no JSON line taken for a result)."""
    r = _bench(["--gpus", "2", "--steps", "2", "--warmup", "1"], {"VV_BENCH_DRYRUN": "1", "VV_BENCH_DRYRUN_FAIL": "1"})
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_bench_refuses_a_mismatched_launcher():
    r = _bench(["--gpus", "4"], {"VV_BENCH_DRYRUN": "1", "WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "--gpus 4" in (r.stderr + r.stdout)
