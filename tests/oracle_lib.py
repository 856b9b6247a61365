"""ctypes access to the CPU oracle (oracle/_build/libvvoracle.so) and, when it has been
built, to the compiled reference (oracle/_ref/libvvref.so).  Test infrastructure only."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys

import numpy as np

REPO = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(REPO, "volume-viz_amd", "python"))
import volviz_amd as vv  # noqa: E402  (struct definitions shared with the product binding)

ORACLE_SO = os.path.join(REPO, "oracle", "_build", "libvvoracle.so")
REF_SO = os.path.join(REPO, "oracle", "_ref", "libvvref.so")


class vvo_volume(C.Structure):
    _fields_ = [("data", C.c_void_p), ("type", C.c_int), ("nx", C.c_int), ("ny", C.c_int), ("nz", C.c_int)]


_oracle = None
_ref = None
_models = {}
MODELS = ("pins", "fmad", "fast", "textrunc")     # oracle/vvo.c VVO_MODEL; "pins" is the oracle proper


def oracle(model: str = "pins") -> C.CDLL:
    """The oracle library; model != "pins" loads the same restatement built under another arithmetic model
    (oracle/Makefile `models`: measurement only, never a parity target)."""
    global _oracle
    if model != "pins":
        if model not in _models:
            so = os.path.join(REPO, "oracle", "_build", f"libvvoracle_{model}.so")
            if not os.path.exists(so):
                subprocess.check_call(["make", "-C", os.path.join(REPO, "oracle"), "models"], stdout=subprocess.DEVNULL)
            lib = C.CDLL(so)
            lib.vvo_render.argtypes = [C.POINTER(vvo_volume), C.c_void_p, C.c_int, C.c_int, C.POINTER(vv.slice_params),
                                       C.POINTER(vv.camera_params), C.POINTER(vv.shading_params),
                                       C.POINTER(vv.vv_ray_source), C.POINTER(vv.vv_render_options), C.c_void_p, C.c_int]
            lib.vvo_render.restype = C.c_ulonglong
            lib.vvo_model.restype = C.c_char_p
            assert lib.vvo_model().decode() == model
            _models[model] = lib
        return _models[model]
    if _oracle is None:
        if not os.path.exists(ORACLE_SO):
            subprocess.check_call(["make", "-C", os.path.join(REPO, "oracle")], stdout=subprocess.DEVNULL)
        lib = C.CDLL(ORACLE_SO)
        vp, i, f, sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
        lib.vvo_draw_ellipsoid.argtypes = [vp, i, i, i, vp, vp, C.c_uint8]
        lib.vvo_draw_default_brain.argtypes = [vp, i, i, i]
        lib.vvo_transfer_preset.argtypes = [i, vp]
        lib.vvo_tex3d.argtypes = [C.POINTER(vvo_volume), f, f, f, i]; lib.vvo_tex3d.restype = f
        lib.vvo_slice.argtypes = [C.POINTER(vvo_volume), vp, sz, sz, f, f, f, i, vp, i, i]
        lib.vvo_slice_advanced.argtypes = [C.POINTER(vvo_volume), vp, sz, sz, vp, vp, i]
        lib.vvo_slice_matrix.argtypes = [f, f, f, f, f, f, vp]
        lib.vvo_ray_endpoints.argtypes = [C.POINTER(vv.vv_ray_source), C.POINTER(vv.camera_params), i, i, i, i, vp, vp]
        lib.vvo_render.argtypes = [C.POINTER(vvo_volume), vp, i, i, C.POINTER(vv.slice_params),
                                   C.POINTER(vv.camera_params), C.POINTER(vv.shading_params),
                                   C.POINTER(vv.vv_ray_source), C.POINTER(vv.vv_render_options), vp, i]
        lib.vvo_render.restype = C.c_ulonglong
        lib.vvo_generate_noise_u8.argtypes = [vp, i, i, i, C.c_uint32]
        lib.vvo_first_pass.argtypes = [C.POINTER(vv.vv_ray_source), C.POINTER(vv.camera_params), i, i, vp, vp]
        _oracle = lib
    return _oracle


def ref():
    """The compiled reference, or None when oracle/_ref was not built (e.g. on the GPU box
    if it was never built in the dev container)."""
    global _ref
    if _ref is None and os.path.exists(REF_SO):
        lib = C.CDLL(REF_SO)
        vp, i, f = C.c_void_p, C.c_int, C.c_float
        lib.ref_default_brain.argtypes = [vp, i, i, i]
        lib.ref_draw_ellipsoids.argtypes = [vp, i, i, i, i, vp, vp, vp]
        lib.ref_save_default_brain.argtypes = [C.c_char_p, i, i, i, i]
        lib.ref_load_raw.argtypes = [C.c_char_p, i, vp, C.c_long, vp]; lib.ref_load_raw.restype = C.c_long
        lib.ref_slice_matrix.argtypes = [f, f, f, f, f, f, vp]
        if hasattr(lib, "ref_cut_plane_pro"):
            lib.ref_cut_plane_pro.argtypes = [f, f, f, f, f, f, vp, vp]
        _ref = lib
    return _ref


def _vol(vol: np.ndarray) -> vvo_volume:
    assert vol.ndim == 3 and vol.flags.c_contiguous and vol.dtype in (np.uint8, np.float32)
    nz, ny, nx = vol.shape
    return vvo_volume(vol.ctypes.data, vv.VOXEL_U8 if vol.dtype == np.uint8 else vv.VOXEL_F32, nx, ny, nz)


def draw_default_brain(nx, ny, nz) -> np.ndarray:
    out = np.zeros((nz, ny, nx), np.uint8)
    oracle().vvo_draw_default_brain(out.ctypes.data, nx, ny, nz)
    return out


def draw_ellipsoids(nx, ny, nz, centers, axes, colors) -> np.ndarray:
    out = np.zeros((nz, ny, nx), np.uint8)
    centers = np.ascontiguousarray(centers, np.float32).reshape(-1, 3)
    axes = np.ascontiguousarray(axes, np.float32).reshape(-1, 3)
    for c, a, col in zip(centers, axes, colors):
        oracle().vvo_draw_ellipsoid(out.ctypes.data, nx, ny, nz, c.ctypes.data, a.ctypes.data, int(col))
    return out


def transfer_preset(p) -> np.ndarray:
    tf = np.zeros(1024, np.float32)
    oracle().vvo_transfer_preset(p, tf.ctypes.data)
    return tf


def noise_u8(nx, ny, nz, seed) -> np.ndarray:
    out = np.zeros((nz, ny, nx), np.uint8)
    oracle().vvo_generate_noise_u8(out.ctypes.data, nx, ny, nz, seed)
    return out


def tex3d(vol, x, y, z, filter=vv.FILTER_TEX8) -> float:
    v = _vol(vol)
    return float(oracle().vvo_tex3d(C.byref(v), x, y, z, filter))


def slice(vol, height, width, dx=0.0, dy=0.0, dz=0.0, orientation=vv.SAGITTAL, scale=(1, 1, 1),
          legacy=False, filter=vv.FILTER_TEX8, fill=0.0) -> np.ndarray:
    v = _vol(vol)
    buf = np.full(height * width, fill, np.float32)
    sc = np.asarray(scale, np.float32)
    oracle().vvo_slice(C.byref(v), buf.ctypes.data, height, width, dx, dy, dz, orientation, sc.ctypes.data,
                       int(legacy), filter)
    return buf


def slice_advanced(vol, height, width, trans, scale=(1, 1, 1), filter=vv.FILTER_TEX8, fill=0.0) -> np.ndarray:
    v = _vol(vol)
    buf = np.full(height * width, fill, np.float32)
    t = np.ascontiguousarray(trans, np.float32).reshape(16)
    sc = np.asarray(scale, np.float32)
    oracle().vvo_slice_advanced(C.byref(v), buf.ctypes.data, height, width, t.ctypes.data, sc.ctypes.data, filter)
    return buf


def slice_matrix(dx, dy, dz, theta, phi, psi) -> np.ndarray:
    m = np.zeros(16, np.float32)
    oracle().vvo_slice_matrix(dx, dy, dz, theta, phi, psi, m.ctypes.data)
    return m.reshape(4, 4)


def ray_endpoints(rs, cam: vv.Camera, W, H, x, y):
    f = np.zeros(3, np.float32); b = np.zeros(3, np.float32)
    cp = cam.params(W, H)
    oracle().vvo_ray_endpoints(C.byref(rs), C.byref(cp), W, H, x, y, f.ctypes.data, b.ctypes.data)
    return f, b


def first_pass(cam: vv.Camera, W, H, rays=None):
    front = np.zeros((H, W, 4), np.uint8); back = np.zeros((H, W, 4), np.uint8)
    rs = rays if rays is not None else vv.analytic_rays(cam); cp = cam.params(W, H)
    oracle().vvo_first_pass(C.byref(rs), C.byref(cp), W, H, front.ctypes.data, back.ctypes.data)
    return front, back


def render(vol, tf, width, height, cam: vv.Camera, *, slice=None, phong=False, rays=None, options=None,
           out=None, fill=0, threads=0, model="pins"):
    """Returns (rgba [H,W,4] uint8, executed_samples)."""
    v = _vol(vol)
    tf = np.ascontiguousarray(tf, np.float32).reshape(1024)
    if out is None:
        out = np.full((height, width, 4), fill, np.uint8)
    sp = slice if slice is not None else vv.make_slice_params()
    cp = cam.params(width, height)
    sh = vv.shading_params(-1, phong)
    rs = rays if rays is not None else vv.analytic_rays(cam)
    n = oracle(model).vvo_render(C.byref(v), tf.ctypes.data, width, height, C.byref(sp), C.byref(cp), C.byref(sh),
                            C.byref(rs), C.byref(options) if options is not None else None,
                            out.ctypes.data, threads)
    return out, int(n)
