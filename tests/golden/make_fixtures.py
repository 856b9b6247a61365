#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/.  Run in the dev container only
(needs /root/reference and oracle/_ref built by `make -C oracle ref`):

    python tests/golden/make_fixtures.py

Sources of truth
  * tf_{engine,head,mri}.f32      parsed from the reference's transfer_functions.h:4-9
                                  (1024 little-endian float32 each: data, not source text)
  * brain_32.u8, brain_aniso.u8   output of the reference's own volumegenerator.cpp
                                  (VolumeGenerator::drawDefaultBrain) compiled unmodified
  * generator_hashes.json         sha256 of larger reference-generator outputs, including
                                  seeded random ellipsoid sets (inputs are in the JSON)
  * brain_16.t3d                  written by the reference's VolumeGenerator::saveas_raw
  * slice_matrices.json           SliceWidget::getTransformationMatrix composition evaluated
                                  with the reference's compiled cs123math/CS123Matrix.cpp
  * frames_oracle.npz             RGBA8 frames / slices rendered by the CPU oracle
                                  (oracle-derived regression pins; the reference has no
                                  golden images and kernel.cu cannot be built here)
"""
import hashlib
import json
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
import oracle_lib as O  # noqa: E402
import volviz_amd as vv  # noqa: E402

REF = os.environ.get("VV_REFERENCE", "/root/reference")


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def ellipsoid_cases():
    """Seeded random ellipsoid sets: (name, dims, centers, axes, colors)."""
    cases = []
    for seed, dims, n in ((1, (24, 24, 24), 3), (2, (40, 17, 9), 5), (3, (64, 48, 32), 8), (4, (33, 1, 7), 2)):
        rng = np.random.default_rng(seed)
        centers = rng.uniform(0.1, 0.9, (n, 3)).astype(np.float32)
        axes = rng.uniform(0.05, 0.5, (n, 3)).astype(np.float32)
        colors = rng.integers(1, 256, n).astype(np.uint8)
        cases.append((f"ellipsoids_seed{seed}", dims, centers, axes, colors))
    return cases


def frame_cases():
    """(name, kwargs) for oracle-rendered golden frames; shared with the tests."""
    cases = []
    cam_a = dict(origin=(0.0, 0.0, -4.0))
    cam_b = dict(origin=tuple(float(v) for v in vv.Camera.orbit(4.0, np.pi / 3, np.pi / 5).origin))
    for tfname in ("engine", "head", "mri"):
        for st, stn in ((vv.SLICE_NONE, "none"), (vv.SLICE_PLANE, "plane"), (vv.SLICE_PLANE_CUT, "cut")):
            for phong in (False, True):
                cases.append((f"brain32_{tfname}_{stn}_{'phong' if phong else 'flat'}_56",
                              dict(vol="brain32", tf=tfname, W=56, H=56, cam=cam_b if phong else cam_a,
                                   slice_type=st, phong=phong)))
    cases.append(("brain64_head_none_flat_170", dict(vol="brain64", tf="head", W=170, H=170, cam=cam_a,
                                                     slice_type=vv.SLICE_NONE, phong=False)))
    cases.append(("brain64_engine_cut_phong_170", dict(vol="brain64", tf="engine", W=170, H=170, cam=cam_b,
                                                       slice_type=vv.SLICE_PLANE_CUT, phong=True)))
    return cases


PLANE_POINT, PLANE_NORMAL = (0.5, 0.5, 0.45), (0.2, -0.3, 0.93)


def make_cut_planes(ref):
    """cut_planes_pro.json: the cutting plane of the free-form slice view (window.cpp:425-441 -> GLWidget::setSlicePro) from the compiled
    reference (oracle/ref_shim.cpp ref_cut_plane_pro): slider-like values (offsets in [-1, 1], angles in [-pi, pi] as the sliders give them,
    window.cpp:408-414) and a few wild ones."""
    rng = np.random.default_rng(11)
    params = [(0, 0, 0, 0, 0, 0), (0.1, -0.2, 0.3, 0.5, -1.0, 2.0), (0, 0, 0, float(np.float32(np.pi)), 0, 0), (0, 0, 0, 0, float(np.float32(np.pi / 2)), 0)] + \
             [tuple(float(np.float32(v)) for v in np.concatenate([rng.uniform(-1, 1, 3), rng.uniform(-np.pi, np.pi, 3)])) for _ in range(20)] + \
             [tuple(float(np.float32(v)) for v in np.concatenate([rng.uniform(-9, 9, 3), rng.uniform(-40, 40, 3)])) for _ in range(6)]
    planes = []
    for p in params:
        pt = np.zeros(3, np.float32); n = np.zeros(3, np.float32)
        ref.ref_cut_plane_pro(*p, pt.ctypes.data, n.ctypes.data)
        planes.append(dict(params=list(p), point_hex=pt.tobytes().hex(), normal_hex=n.tobytes().hex()))
    json.dump(planes, open(os.path.join(HERE, "cut_planes_pro.json"), "w"), indent=1)


def main():
    if not os.path.isdir(REF):
        sys.exit(f"reference tree {REF} not present")
    ref = O.ref()
    if ref is None:
        sys.exit("oracle/_ref/libvvref.so missing: run `make -C oracle ref`")

    # --- transfer functions ---------------------------------------------------
    src = open(os.path.join(REF, "transfer_functions.h")).read()
    for m in re.finditer(r"float\s+g_transfer(\w+)\[1024\]\s*=\s*\{([^}]*)\}", src, re.S):
        vals = np.array([float(x) for x in m.group(2).split(",") if x.strip()], dtype=np.float64).astype("<f4")
        assert vals.size == 1024
        vals.tofile(os.path.join(HERE, f"tf_{m.group(1).lower()}.f32"))

    # --- generator ----------------------------------------------------------------
    hashes = {}
    b32 = np.zeros((32, 32, 32), np.uint8); ref.ref_default_brain(b32.ctypes.data, 32, 32, 32)
    b32.tofile(os.path.join(HERE, "brain_32.u8"))
    an = np.zeros((52, 36, 20), np.uint8); ref.ref_default_brain(an.ctypes.data, 20, 36, 52)
    an.tofile(os.path.join(HERE, "brain_aniso_20x36x52.u8"))
    for n in (32, 64, 128, 256):
        b = np.zeros((n, n, n), np.uint8); ref.ref_default_brain(b.ctypes.data, n, n, n)
        u, c = np.unique(b, return_counts=True)
        hashes[f"brain_{n}"] = dict(dims=[n, n, n], sha256=sha(b), histogram={int(k): int(v) for k, v in zip(u, c)})
    hashes["brain_aniso_20x36x52"] = dict(dims=[20, 36, 52], sha256=sha(an))
    for name, dims, centers, axes, colors in ellipsoid_cases():
        nx, ny, nz = dims
        out = np.zeros((nz, ny, nx), np.uint8)
        ref.ref_draw_ellipsoids(out.ctypes.data, nx, ny, nz, len(colors), centers.ctypes.data, axes.ctypes.data,
                                colors.ctypes.data)
        hashes[name] = dict(dims=list(dims), sha256=sha(out), centers=centers.tolist(), axes=axes.tolist(),
                            colors=colors.tolist())
    json.dump(hashes, open(os.path.join(HERE, "generator_hashes.json"), "w"), indent=1, sort_keys=True)

    # --- .t3d written by the reference ---------------------------------------------
    ref.ref_save_default_brain(os.path.join(HERE, "brain_16.t3d").encode(), 1, 16, 16, 16)

    # --- slice matrices ---------------------------------------------------------------
    rng = np.random.default_rng(7)
    mats = []
    params = [(0, 0, 0, 0, 0, 0), (0.1, -0.2, 0.3, 0.5, -1.0, 2.0)] + \
             [tuple(float(np.float32(v)) for v in np.concatenate([rng.uniform(-1, 1, 3), rng.uniform(-3.1, 3.1, 3)]))
              for _ in range(14)]
    for p in params:
        m = np.zeros(16, np.float32)
        ref.ref_slice_matrix(*p, m.ctypes.data)
        mats.append(dict(params=list(p), matrix_hex=m.tobytes().hex()))
    json.dump(mats, open(os.path.join(HERE, "slice_matrices.json"), "w"), indent=1)

    make_cut_planes(ref)

    # --- oracle-rendered frames -----------------------------------------------------------
    vols = {"brain32": b32, "brain64": O.draw_default_brain(64, 64, 64)}
    tfs = {"engine": vv.TF_ENGINE, "head": vv.TF_HEAD, "mri": vv.TF_MRI}
    frames = {}
    for name, kw in frame_cases():
        cam = vv.Camera(**kw["cam"])
        sp = vv.make_slice_params(kw["slice_type"], PLANE_POINT, PLANE_NORMAL)
        img, n = O.render(vols[kw["vol"]], O.transfer_preset(tfs[kw["tf"]]), kw["W"], kw["H"], cam, slice=sp,
                          phong=kw["phong"], fill=0x5A)
        frames[name] = img
        frames[name + "__samples"] = np.array([n], np.int64)
    # slice goldens
    for oname, o in (("sagittal", vv.SAGITTAL), ("horizontal", vv.HORIZONTAL), ("coronal", vv.CORONAL)):
        frames[f"slice_brain64_{oname}_64"] = O.slice(vols["brain64"], 64, 64, 0.05, 0.4, 0.3, o, (1.0, 1.0, 0.8))
    frames["slice_brain64_free_64"] = O.slice_advanced(vols["brain64"], 64, 64,
                                                       O.slice_matrix(0.1, -0.05, 0.02, 0.4, -0.3, 0.2), (1.0, 1.0, 1.0))
    np.savez_compressed(os.path.join(HERE, "frames_oracle.npz"), **frames)
    print("fixtures written to", HERE)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "cut_planes":          # only the fixture added in round 5 (the others are unchanged)
        make_cut_planes(O.ref())
    else:
        main()
