"""developer tool (1 GPU): wild parity fuzzing of the rows beside the ray march -- slice sampler (canonical, advanced),
procedural generator, first pass and renders fed by first-pass images or quantised analytic rays -- against the oracle.
usage: python3 tools/fuzz_wild_aux.py <first seed> <last seed + 1>"""
import os, sys
import numpy as np
REPO = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, os.path.join(REPO, "volume-viz_amd", "python"))
import torch  # noqa: F401
import oracle_lib as O
import volviz_amd as vv


def run_case(ctx, seed):
    """Returns a list of (what, ok) for one seed."""
    rng = np.random.default_rng(1700000 + seed)
    out = []
    kind = seed % 4
    if kind == 0:                                     # slice sampler
        dims = tuple(int(v) for v in rng.choice([1, 2, 3, 5, 8, 17, 40, 64], size=3))
        vol = rng.integers(0, 256, size=dims[::-1], dtype=np.uint8)
        if rng.random() < 0.5:
            vol = (vol.astype(np.float32) / np.float32(255) * np.float32(rng.choice([1.0, 1.7])) - np.float32(rng.choice([0.0, 0.3]))).astype(np.float32)
        ctx.load_volume(vol, vv.transfer_preset(vv.TF_HEAD))
        h, w = int(rng.choice([1, 2, 17, 64, 96, int(rng.integers(1, 130))])), int(rng.choice([1, 3, 16, 64, int(rng.integers(1, 130))]))
        scale = tuple(float(v) for v in rng.choice([1.0, 0.1, 0.8, 1.57, 4.0], size=3))
        filt = int(rng.choice([vv.FILTER_TEX8, vv.FILTER_EXACT]))
        for orient in (vv.SAGITTAL, vv.CORONAL, vv.HORIZONTAL, vv.FREE_FORM):
            d = [float(v) for v in rng.uniform(-1.2, 1.2, size=3)]
            got = ctx.slice(h, w, *d, orientation=orient, scale=scale, filter=filt, fill=-1.0)
            want = O.slice(vol, h, w, *d, orientation=orient, scale=scale, filter=filt, fill=-1.0)
            out.append((f"slice {dims} {vol.dtype} {h}x{w} orient {orient} {scale} filt {filt}", np.array_equal(got, want)))
        m = O.slice_matrix(*[float(v) for v in rng.uniform(-1.0, 1.0, size=3)], *[float(v) for v in rng.uniform(-7, 7, size=3)])
        got = ctx.slice_advanced(h, w, m, scale=scale, filter=filt, fill=-1.0)
        want = O.slice_advanced(vol, h, w, m, scale=scale, filter=filt, fill=-1.0)
        out.append((f"slice_advanced {dims} {vol.dtype} {h}x{w} {scale}", np.array_equal(got, want)))
    elif kind == 1:                                   # generator
        dims = tuple(int(v) for v in rng.choice([1, 2, 7, 16, 33, 64, 100], size=3))
        ne = int(rng.integers(0, 12))
        centers = rng.uniform(-0.3, 1.3, (ne, 3)); axes = rng.uniform(0.01, 0.9, (ne, 3)); colors = rng.integers(0, 256, ne).astype(np.uint8)
        if ne and rng.random() < 0.3:
            axes[0] = rng.choice([1e-6, 0.0, 5.0], size=3)
        got = ctx.generate_ellipsoids(*dims, centers, axes, colors)
        out.append((f"ellipsoids {dims} n={ne}", np.array_equal(got, O.draw_ellipsoids(*dims, centers, axes, colors))))
        out.append((f"default brain {dims}", np.array_equal(ctx.generate_default_brain(*dims), O.draw_default_brain(*dims))))
    else:                                             # first pass + renders fed by its images / quantised rays
        scale = tuple(float(v) for v in rng.choice([1.0, 1.0, 0.5, 1.57], size=3))
        r = float(rng.choice([0.5, 1.3, 2.5, 4.0])); th = float(rng.uniform(0.05, np.pi - 0.05)); ph = float(rng.uniform(-np.pi, np.pi))
        cam = vv.Camera.orbit(r, th, ph, scale=scale)
        W, H = int(rng.integers(2, 70)), int(rng.integers(2, 60))
        k = int(rng.choice([1, 2, 3]))
        gf, gb = ctx.first_pass(k * W, k * H, cam)
        of, ob = O.first_pass(cam, k * W, k * H)
        out.append((f"first pass {k * W}x{k * H} r={r}", bool(np.array_equal(gf, of) and np.array_equal(gb, ob))))
        dims = tuple(int(v) for v in rng.integers(4, 36, size=3))
        vol = O.noise_u8(*dims, int(rng.integers(1, 2**31)))
        if rng.random() < 0.5:
            vol = vol.astype(np.float32) / np.float32(255)
        tf = vv.transfer_preset(int(rng.choice([vv.TF_ENGINE, vv.TF_HEAD, vv.TF_MRI])))
        ctx.load_volume(vol, tf)
        phong = bool(rng.random() < 0.5)
        o = vv.make_options(step=float(rng.choice([1 / 16, 1 / 64])), count_samples=True)
        if kind == 2:
            rs_g, rs_o = vv.image_rays(gf, gb), vv.image_rays(of, ob)
        else:
            rs_g = rs_o = vv.analytic_rays(cam, quantize8=True)
        got = ctx.render(W, H, cam, rays=rs_g, phong=phong, options=o, fill=9)
        n_got = ctx.last_sample_count()
        want, n = O.render(vol, tf, W, H, cam, rays=rs_o, phong=phong, options=o, fill=9)
        out.append((f"render from {'images' if kind == 2 else 'quantised rays'} {dims} {vol.dtype} {W}x{H} phong={phong} r={r}", bool(np.array_equal(got, want) and n_got == n)))
    return out


def main():
    lo, hi = int(sys.argv[1]), int(sys.argv[2])
    ctx = vv.Context(0)
    bad = 0
    for seed in range(lo, hi):
        try:
            res = run_case(ctx, seed)
        except vv.VolvizError as e:
            print("REFUSED seed", seed, e); continue
        for what, ok in res:
            if not ok:
                bad += 1
                print("MISMATCH seed", seed, what, flush=True)
        if bad > 10:
            break
        if seed % 500 == 0:
            print("seed", seed, "mismatches so far", bad, flush=True)
    print(f"seeds {lo}..{hi - 1}: {bad} mismatches")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
