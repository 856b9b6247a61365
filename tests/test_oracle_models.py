"""How much do frames move if the choices the oracle PINS are made the other way?

The ray-march oracle is parity-unpinned by the reference (kernel.cu needs nvcc; the reference holds no golden frames), so
"bit-exact" in tests/test_gpu_parity.py means: equal to OUR restatement under OUR pins.  This file measures, on the CPU,
how far that can be from a CUDA build whose arithmetic nobody can read off the sources: the same restatement (oracle/vvo.c)
is built under three other arithmetic models (VVO_MODEL, oracle/Makefile `models`):

  fmad      a*b+c contracted wherever the compiler can (nvcc's default --fmad=true)
  fast      fmad + x/y as x*rcp(y), sqrt as x*rsqrt(x), rsqrt as rcp(sqrt), FTZ/DAZ  (what -use_fast_math adds, .pro:52)
  textrunc  texture-unit weights truncated to 8 fractional bits instead of rounded to nearest even (pin 2)

and every committed golden frame case, C1 with and without Phong and a C3-like noise volume are rendered
under all four.  The product equals the "pins" model bit for bit (GPU tests), so the distances below are the product's too.
This cannot make parity green; it tells a reader what "bit-exact against our pins" is worth.  Measured (see the asserts):

  * unshaded frames: every model stays within BASELINE.md's tolerance in the statistic that matters (>= 99.9 % of the pixels
    within 1 LSB on frames with enough pixels for 0.1 % to mean something; never more than 3 LSB);
  * Phong frames under `fast` do NOT: up to ~20 LSB on ~10 % of the pixels of the engine-table cases.  Cause (traced): with
    the six divisions by 255.f turned into multiplications by a rounded reciprocal AND contraction on, `r - l` of two EQUAL
    bytes becomes fma(r, c, -(l * c)) = the rounding error of l * c, not 0; the `gradient != 0` test of kernel.cu:180 then
    normalises a vector that is zero in exact arithmetic and the plateau is lit at the full 0.3 instead of 0.04.  Whether
    nvcc 5.5 fused exactly there is not knowable from the sources; the reference's Phong output on plateaus is
    ill-conditioned with respect to it, ours takes the IEEE reading (pin 3).
"""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
import oracle_lib as O
import volviz_amd as vv
from make_fixtures import frame_cases, PLANE_POINT, PLANE_NORMAL

REPO = os.path.normpath(os.path.join(os.path.dirname(__file__), ".."))


def _ramp_tf():
    v = np.arange(256, dtype=np.float64) / 255.0
    return np.stack([v, 1.0 - v, np.abs(2.0 * v - 1.0), 0.03 * v * v], axis=1).astype(np.float32).reshape(1024)


def _cases():
    vols = {"brain32": O.draw_default_brain(32, 32, 32), "brain64": O.draw_default_brain(64, 64, 64)}
    tfs = {"engine": vv.TF_ENGINE, "head": vv.TF_HEAD, "mri": vv.TF_MRI}
    out = []
    for name, kw in frame_cases():
        out.append((name, vols[kw["vol"]], O.transfer_preset(tfs[kw["tf"]]), kw["W"], kw["H"], vv.Camera(**kw["cam"]),
                    vv.make_slice_params(kw["slice_type"], PLANE_POINT, PLANE_NORMAL), kw["phong"], None))
    c1 = O.draw_default_brain(128, 128, 128)
    head = vv.transfer_preset(vv.TF_HEAD)
    out.append(("C1 (128^3 brain, 512x512, head table)", c1, head, 512, 512, vv.Camera(), None, False, vv.make_options(step=1 / 128)))
    out.append(("C1 + Phong", c1, head, 512, 512, vv.Camera(), None, True, vv.make_options(step=1 / 128)))
    nz = O.noise_u8(128, 128, 128, 0x9E3779B9).astype(np.float32) / np.float32(255)
    out.append(("noise 128^3 f32, colour ramp (C3's content), view a", nz, _ramp_tf(), 480, 270, vv.Camera(), None, False, vv.make_options(step=1 / 256)))
    out.append(("noise 128^3 f32, colour ramp, view b", nz, _ramp_tf(), 480, 270, vv.Camera.orbit(4.0, np.pi / 3, np.pi / 5), None, False,
                vv.make_options(step=1 / 256)))
    return out


def _hist(a, b):
    d = np.abs(a.astype(int) - b.astype(int)).max(axis=-1)
    return [int((d == k).sum()) for k in range(4)] + [int((d >= 4).sum())], int(d.max()), int(d.size)


def test_frames_under_other_arithmetic_models():
    rows = []
    for name, vol, tf, W, H, cam, sp, phong, opts in _cases():
        ref, n0 = O.render(vol, tf, W, H, cam, slice=sp, phong=phong, options=opts, fill=0x5A)
        for m in O.MODELS[1:]:
            img, n = O.render(vol, tf, W, H, cam, slice=sp, phong=phong, options=opts, fill=0x5A, model=m)
            h, mx, tot = _hist(img, ref)
            rows.append((name, m, phong, tot, h, mx, n - n0))
    lines = ["# LSB distance (max over RGBA per pixel) of oracle frames under other arithmetic models from the pinned oracle (= the product)",
             "# tests/test_oracle_models.py; columns: case | model | pixels | =0 | 1 | 2 | 3 | >=4 | max | % within 1 LSB | executed-sample difference"]
    for name, m, phong, tot, h, mx, dn in rows:
        lines.append(f"{name:52s} | {m:8s} | {tot:7d} | {h[0]:7d} | {h[1]:6d} | {h[2]:5d} | {h[3]:4d} | {h[4]:5d} | {mx:3d} | {100.0 * (h[0] + h[1]) / tot:8.4f} | {dn:+d}")
    report = "\n".join(lines) + "\n"
    out = os.environ.get("VV_MODELS_REPORT")
    if out:
        open(out, "w").write(report)
    print(report)
    for name, m, phong, tot, h, mx, dn in rows:
        within1 = (h[0] + h[1]) / tot
        what = f"{name} under {m}: {h} max {mx}"
        assert abs(dn) <= 16, what                       # chunk / ERT decisions move by a handful of samples at most
        if not phong:
            assert mx <= 3, what
            assert within1 >= (0.999 if tot >= 20000 else 0.998), what      # (a 56 x 56 frame has 3136 pixels: 0.1 % = 3 pixels)
        elif m != "fast":
            assert mx <= 11 and within1 >= 0.987, what   # truncated weights move classification indices on the binary brain's edges
        # Phong under `fast`: not bounded (module docstring); the unconditioned cases are the engine-table ones
    worst_fast_phong = min((h[0] + h[1]) / tot for name, m, phong, tot, h, mx, dn in rows if phong and m == "fast")
    assert worst_fast_phong < 0.999, "the fast model no longer moves Phong frames: update the module docstring and DESIGN.md"
