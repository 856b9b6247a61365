"""developer tool (1 GPU): parity fuzzing far outside the distributions of the committed sweeps -- degenerate volume
shapes, tables with colours / opacities outside [0, 1], eyes inside the cube, extreme scales, steps and thresholds,
shards -- cycling every layout and launch form.  usage: python3 tools/fuzz_wild.py <first seed> <last seed + 1>"""
import os, sys
import numpy as np
REPO = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, os.path.join(REPO, "volume-viz_amd", "python"))
import torch  # noqa: F401  (its HIP runtime first)
import oracle_lib as O
import volviz_amd as vv

KNOBS = ("VV_ZFAST", "VV_BRICKED", "VV_ZPAIR", "VV_FORCE_BIG", "VV_PITCH_FORCE", "VV_UNROLL", "VV_BLOCK_W", "VV_TILE_LOG2W", "VV_TAIL", "VV_LPT", "VV_LPT_RUN", "VV_RECT", "VV_PHONG_BRICKS")
ENVS = [{}, {"VV_ZFAST": "1"}, {"VV_BRICKED": "1"}, {"VV_ZPAIR": "1"}, {"VV_FORCE_BIG": "1"}, {"VV_PITCH_FORCE": "1"}, {"VV_UNROLL": "2"}, {"VV_UNROLL": "3", "VV_BRICKED": "1"},
        {"VV_ZFAST": "1", "VV_ZPAIR": "0"}, {"VV_UNROLL": "2", "VV_FORCE_BIG": "1"}, {"VV_BRICKED": "0"}, {"VV_FORCE_BIG": "1", "VV_BRICKED": "1"},
        {"VV_BLOCK_W": "64", "VV_TILE_LOG2W": "5"}, {"VV_BLOCK_W": "128", "VV_TILE_LOG2W": "5"}, {"VV_BLOCK_W": "8", "VV_TILE_LOG2W": "3", "VV_BRICKED": "1"}, {"VV_TILE_LOG2W": "4"}, {"VV_TAIL": "0"}, {"VV_BLOCK_W": "16", "VV_TILE_LOG2W": "3"},
        {"VV_LPT": "1"}, {"VV_LPT": "1", "VV_LPT_RUN": "3", "VV_BRICKED": "1"}, {"VV_RECT": "0"}, {"VV_PHONG_BRICKS": "1"}, {"VV_LPT": "1", "VV_ZFAST": "1"}]


def case(seed):
    rng = np.random.default_rng(900000 + seed)
    dims = tuple(int(v) for v in rng.choice([1, 2, 3, 4, 5, 7, 8, 16, 17, 31, 40], size=3))
    vol = rng.integers(0, 256, size=dims[::-1], dtype=np.uint8)
    if rng.random() < 0.3:
        vol = O.noise_u8(*dims, int(rng.integers(1, 2**31)))
    if rng.random() < 0.6:
        vol = vol.astype(np.float32) / np.float32(255)
        if rng.random() < 0.5:
            vol = (vol * np.float32(rng.choice([1.3, 2.0, 0.5])) - np.float32(rng.choice([0.1, 0.5, 0.0]))).astype(np.float32)
    tf = rng.uniform(0, 1, (256, 4)).astype(np.float32)
    if rng.random() < 0.3:
        tf[:, :3] = rng.uniform(-0.5, 1.5, (256, 3)).astype(np.float32)
    mode = int(rng.integers(0, 6))
    if mode == 0: tf[:, 3] *= np.float32(0.03)
    elif mode == 1: tf[:, 3] = rng.uniform(-0.5, 2.5, 256).astype(np.float32)
    elif mode == 2: tf[:, 3] = 0
    elif mode == 3: tf[:, 3] = 1
    elif mode == 4: tf[: int(rng.integers(0, 200)), 3] = 0
    W = int(rng.choice([2, 3, 14, 15, 16, 29, 43, int(rng.integers(2, 120))])); H = int(rng.choice([2, 3, 14, 15, 29, int(rng.integers(2, 100))]))
    scale = tuple(float(v) for v in rng.choice([1.0, 1.0, 0.1, 0.5, 1.57, 4.0], size=3))
    r = float(rng.choice([0.3, 0.9, 1.0, 1.2, 2.0, 4.0, 8.0]))
    th = float(rng.choice([np.pi / 2, 0.01, np.pi - 0.01, rng.uniform(0.05, np.pi - 0.05)])); ph = float(rng.choice([0.0, np.pi / 2, -np.pi / 2, np.pi, rng.uniform(-np.pi, np.pi)]))
    cam = vv.Camera.orbit(r, th, ph, scale=scale)
    st = int(rng.choice([vv.SLICE_NONE, vv.SLICE_NONE, vv.SLICE_PLANE, vv.SLICE_PLANE_CUT]))
    sp = vv.make_slice_params(st, tuple(rng.uniform(-0.2, 1.2, size=3)), tuple(rng.normal(size=3)))
    step = None if rng.random() < 0.3 else float(rng.choice([1 / 3, 1 / 16, 1 / 64, 1 / 300]))
    shard = None
    if rng.random() < 0.25:
        cnt = int(rng.integers(2, 5)); shard = (int(rng.choice([4, 8])), cnt, int(rng.integers(0, cnt)))
    o = dict(step=step, filter=int(rng.choice([vv.FILTER_TEX8, vv.FILTER_EXACT])), ert_mode=int(rng.choice([vv.ERT_REFERENCE, vv.ERT_TRUE])),
             ert_threshold=float(rng.choice([0.01, 0.5, 0.95, 0.999, 1.5])), count_samples=bool(rng.random() < 0.7), shard=shard)
    phong = bool(rng.random() < 0.45)
    # (second stream, added later: anisotropic steps -- pin 8 -- and slab-row ranges)
    rng2 = np.random.default_rng(5000000 + seed)
    if rng2.random() < 0.2:
        o["step"] = tuple(float(v) for v in rng2.choice([1 / 5, 1 / 16, 1 / 40, 1 / 64], size=3))
    if shard is None and rng2.random() < 0.2:
        nby = H // 14 + (1 if H % 14 else 0)
        a = int(rng2.integers(0, nby + 1)); b = int(rng2.integers(a, nby + 1))
        o["slab_rows"] = (a, b)
    return vol, tf, W, H, cam, sp, phong, o


def main():
    lo, hi = int(sys.argv[1]), int(sys.argv[2])
    ctx = vv.Context(0)
    bad = 0
    only = os.environ.get("FUZZ_ENV_ONLY")          # index into ENVS: run only the seeds that use that entry
    for seed in range(lo, hi):
        if only is not None and seed % len(ENVS) != int(only):
            continue
        env = ENVS[seed % len(ENVS)]
        for k in KNOBS:
            os.environ.pop(k, None)
        os.environ.update(env)
        vol, tf, W, H, cam, sp, phong, o = case(seed)
        try:
            ctx.load_volume(vol, tf)
            opts = vv.make_options(**o)
            if seed % 3 == 2:      # every third case through the device-buffer form on a caller's stream (bench.py's path)
                buf = torch.full((H, W, 4), 0x3C, dtype=torch.uint8, device="cuda")
                st = torch.cuda.Stream()
                with torch.cuda.stream(st):
                    ctx.render_device(W, H, cam, buf.data_ptr(), slice=sp, phong=phong, options=opts, stream=vv.stream_handle(st))
                st.synchronize()
                got = buf.cpu().numpy()
            else:
                got = ctx.render(W, H, cam, slice=sp, phong=phong, options=opts, fill=0x3C)
            n_got = ctx.last_sample_count() if o["count_samples"] else None
        except vv.VolvizError as e:
            print("REFUSED seed", seed, env, vol.shape, W, H, e); continue
        want, n = O.render(vol, tf, W, H, cam, slice=sp, phong=phong, options=opts, fill=0x3C)
        if not np.array_equal(got, want) or (n_got is not None and n_got != n):
            bad += 1
            print("MISMATCH seed", seed, env, vol.shape, vol.dtype, W, H, "phong", phong, o, "pixels", int((got != want).any(axis=-1).sum()), n_got, n, flush=True)
            if bad > 8:
                break
        if seed % 100 == 0:
            print("seed", seed, "ok so far, mismatches", bad, flush=True)
    print(f"seeds {lo}..{hi - 1}: {bad} mismatches")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
