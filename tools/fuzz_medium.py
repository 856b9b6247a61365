"""developer tool (1 GPU): parity fuzzing at sizes where vv_render's own policy picks the layout (bricked copy from 2 M
voxels, z-pair copy, re-pitched rows for widths that are multiples of 256 voxels, several strips / slabs per XCD): no knob
is forced -- unless FUZZ_KNOBS=1, which cycles the layout / launch-form knobs of tools/fuzz_wild.py over the same cases.
usage: python3 tools/fuzz_medium.py <first seed> <last seed + 1>"""
import os, sys
import numpy as np
REPO = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, os.path.join(REPO, "volume-viz_amd", "python"))
import torch  # noqa: F401
import oracle_lib as O
import volviz_amd as vv


def case(seed):
    rng = np.random.default_rng(31000000 + seed)
    dims = tuple(int(v) for v in rng.choice([64, 100, 128, 160, 200, 256], size=3))
    if rng.random() < 0.3:
        dims = (256, int(rng.choice([64, 128])), int(rng.choice([64, 128])))          # rows of 1 KiB (f32): re-pitched
    kind = int(rng.integers(0, 3))
    vol = O.draw_default_brain(*dims) if kind == 0 else O.noise_u8(*dims, int(rng.integers(1, 2**31)))
    if rng.random() < 0.6:
        vol = vol.astype(np.float32) / np.float32(255)
    tf = vv.transfer_preset(int(rng.choice([vv.TF_ENGINE, vv.TF_HEAD, vv.TF_MRI]))) if rng.random() < 0.6 else rng.uniform(0, 1, (256, 4)).astype(np.float32)
    W, H = int(rng.integers(60, 420)), int(rng.integers(40, 300))
    scale = tuple(float(v) for v in rng.choice([1.0, 1.0, 1.0, 0.8, 1.57], size=3))
    cam = vv.Camera.orbit(float(rng.uniform(1.5, 5.0)), float(rng.uniform(0.1, np.pi - 0.1)), float(rng.uniform(-np.pi, np.pi)), scale=scale)
    if rng.random() < 0.3:
        cam = vv.Camera(origin=(float(rng.uniform(-0.5, 0.5)), float(rng.uniform(-0.5, 0.5)), float(rng.choice([-4.0, 4.0, -2.5]))), scale=scale)
    st = int(rng.choice([vv.SLICE_NONE, vv.SLICE_NONE, vv.SLICE_PLANE, vv.SLICE_PLANE_CUT]))
    sp = vv.make_slice_params(st, tuple(rng.uniform(0.2, 0.8, size=3)), tuple(rng.normal(size=3)))
    shard = None
    if rng.random() < 0.25:
        cnt = int(rng.integers(2, 9)); shard = (4, cnt, int(rng.integers(0, cnt)))
    o = dict(step=float(rng.choice([1 / 64, 1 / 128, 1 / 200, 1 / 256])), filter=int(rng.choice([vv.FILTER_TEX8, vv.FILTER_EXACT])),
             ert_mode=int(rng.choice([vv.ERT_REFERENCE, vv.ERT_TRUE])), ert_threshold=float(rng.choice([0.5, 0.95, 0.999])),
             count_samples=bool(rng.random() < 0.5), shard=shard)
    return vol, tf, W, H, cam, sp, bool(rng.random() < 0.4), o


def main():
    lo, hi = int(sys.argv[1]), int(sys.argv[2])
    ctx = vv.Context(0)
    bad = 0
    used = {"bricks": 0, "zpair": 0}
    knobs = os.environ.get("FUZZ_KNOBS") == "1"
    KNOBS = ("VV_BRICKED", "VV_ZPAIR", "VV_ZFAST", "VV_FORCE_BIG", "VV_PITCH_FORCE", "VV_UNROLL")
    ENVS = [{"VV_BRICKED": "1"}, {"VV_ZPAIR": "1"}, {"VV_FORCE_BIG": "1"}, {"VV_PITCH_FORCE": "1"}, {"VV_ZFAST": "1"}, {"VV_UNROLL": "2"},
            {"VV_UNROLL": "3", "VV_BRICKED": "1"}, {"VV_ZFAST": "1", "VV_ZPAIR": "0"}, {"VV_BRICKED": "0"}, {"VV_FORCE_BIG": "1", "VV_BRICKED": "1"}]
    for seed in range(lo, hi):
        vol, tf, W, H, cam, sp, phong, o = case(seed)
        if knobs:
            for k in KNOBS:
                os.environ.pop(k, None)
            os.environ.update(ENVS[seed % len(ENVS)])
        ctx.load_volume(vol, tf)
        opts = vv.make_options(**o)
        got = ctx.render(W, H, cam, slice=sp, phong=phong, options=opts, fill=0x3C)
        n_got = ctx.last_sample_count() if o["count_samples"] else None
        if o["count_samples"]:
            cnt = ctx.debug_counters(); used["bricks"] += int(cnt[2] > 0); used["zpair"] += int(cnt[3] > 0)
        want, n = O.render(vol, tf, W, H, cam, slice=sp, phong=phong, options=opts, fill=0x3C, threads=16)
        if not np.array_equal(got, want) or (n_got is not None and n_got != n):
            bad += 1
            print("MISMATCH seed", seed, vol.shape, vol.dtype, W, H, "phong", phong, o, "pixels", int((got != want).any(axis=-1).sum()), n_got, n, flush=True)
            if bad > 8:
                break
        if seed % 25 == 0:
            print("seed", seed, "mismatches so far", bad, used, flush=True)
    print(f"seeds {lo}..{hi - 1}: {bad} mismatches; instrumented frames on bricks / z-pair copy: {used}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
