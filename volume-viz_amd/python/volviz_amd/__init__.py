"""ctypes binding of the C-ABI in include/volviz.h (libvolviz_hip.so).

This is the host-side mirror used by tests/, bench.py and __graft_entry__.py.  It is a
thin 1:1 wrapper: every method names the C entry point it calls, which in turn cites
the reference interface it replaces (kernel.cuh / volumegenerator.h).  There is no
Python or CPU implementation of any kernel here -- if the shared library or a HIP
device is missing, calls fail loudly.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
PKG_ROOT = os.path.normpath(os.path.join(_HERE, "..", ".."))          # volume-viz_amd/
REPO_ROOT = os.path.normpath(os.path.join(PKG_ROOT, ".."))
LIB_PATH = os.environ.get("VV_LIB", os.path.join(PKG_ROOT, "lib", "libvolviz_hip.so"))

# kernel.cuh:18-20
SLICE_NONE, SLICE_PLANE, SLICE_PLANE_CUT = -1, 0, 1
# params.h:46
HORIZONTAL, SAGITTAL, CORONAL, FREE_FORM = 0, 1, 2, 4
VOXEL_U8, VOXEL_F32 = 0, 1
LAYOUT_BRICKED, LAYOUT_ZPAIR, LAYOUT_ZFAST, LAYOUT_POLICY = 1, 2, 4, 256
FILTER_TEX8, FILTER_EXACT = 0, 1
ERT_REFERENCE, ERT_TRUE = 0, 1
RAYS_IMAGES, RAYS_ANALYTIC = 0, 1
TF_ENGINE, TF_HEAD, TF_MRI = 0, 1, 2
# `stream` arguments: 0/None = the context's own stream, synchronous; STREAM_DEFAULT_ASYNC = the device's default (null)
# stream, enqueue only (include/volviz.h VV_STREAM_DEFAULT_ASYNC); any other hipStream_t handle = that stream, enqueue only
STREAM_DEFAULT_ASYNC = 1


def stream_handle(torch_stream) -> int:
    """hipStream_t of a torch.cuda.Stream for the `stream` arguments (torch's default stream has handle 0, which the
    C-ABI reads as 'no stream': it maps to STREAM_DEFAULT_ASYNC)."""
    h = int(torch_stream.cuda_stream)
    return h if h != 0 else STREAM_DEFAULT_ASYNC


class slice_params(C.Structure):          # kernel.cuh:26-29
    _fields_ = [("type", C.c_int), ("params", C.c_float * 6)]


class camera_params(C.Structure):         # kernel.cuh:31-35
    _fields_ = [("origin", C.c_float * 3), ("fovX", C.c_float), ("fovY", C.c_float),
                ("scale", C.c_float * 3)]


class shading_params(C.Structure):        # kernel.cuh:37-40
    _fields_ = [("transferPreset", C.c_int), ("phongShading", C.c_bool)]


class vv_ray_source(C.Structure):
    _fields_ = [("mode", C.c_int), ("front", C.c_void_p), ("back", C.c_void_p),
                ("img_w", C.c_int), ("img_h", C.c_int), ("images_on_device", C.c_int),
                ("look", C.c_float * 3), ("up", C.c_float * 3), ("aspect", C.c_float),
                ("quantize8", C.c_int)]


class vv_render_options(C.Structure):
    _fields_ = [("step", C.c_float * 3), ("ert_threshold", C.c_float), ("filter", C.c_int),
                ("ert_mode", C.c_int), ("slab_row_begin", C.c_int), ("slab_row_end", C.c_int),
                ("shard_band", C.c_int), ("shard_count", C.c_int), ("shard_index", C.c_int),
                ("count_samples", C.c_int), ("touched_bricks", C.c_void_p),
                ("touched_lines", C.c_void_p), ("touched_line_bits", C.c_ulonglong), ("touched_lines_all", C.c_int),
                ("touched_block_lines", C.c_void_p), ("touched_block_lines_log2", C.c_int)]


EXPORTS = [
    "vv_init", "vv_shutdown", "vv_last_error", "vv_load_volume_u8", "vv_load_volume_f32",
    "vv_load_volume_device", "vv_set_transfer_function", "vv_render", "vv_slice",
    "vv_slice_advanced", "vv_generate_ellipsoids", "vv_generate_default_brain",
    "vv_promote_u8_to_f32", "vv_generate_noise_u8", "vv_transfer_preset",
    "vv_t3d_read_header", "vv_t3d_read", "vv_t3d_write", "vv_last_frame_ms",
    "vv_last_sample_count", "vv_volume_dims", "vv_slice_matrix", "vv_draw_ellipsoid", "vv_debug_counters",
    "vv_first_pass", "vv_cut_plane_canonical", "vv_cut_plane_from_euler", "vv_cut_plane_to_slice_params", "vv_slice_to_bgra",
    "vv_camera_orbit_drag", "vv_camera_zoom", "vv_cut_plane_from_drag", "vv_cut_plane_drag",
    "vv_prepare_layouts", "vv_set_layout_policy", "vv_layout_state", "vv_device_bytes", "vv_reread_env",
    "vv_load_volume_stream_begin", "vv_load_volume_stream_slices", "vv_load_volume_stream_end", "vv_load_volume_t3d",
    "vv_load_volume_stream_slices_async", "vv_load_volume_stream_wait_source", "vv_dataset_preset", "vv_debug_last_launch", "vv_set_frame_timing", "vv_debug_screen_rect",
]

_lib = None
_libs = {}


def load_library(path: Optional[str] = None) -> C.CDLL:
    """dlopen libvolviz_hip.so and declare the prototypes.  Raises if it is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    if path is not None and path in _libs:
        return _libs[path]
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise RuntimeError(f"{p} not found: build it with `make -C {PKG_ROOT}` "
                           "(python __graft_entry__.py does). There is no fallback path.")
    lib = C.CDLL(p)
    vp, i, f, sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
    lib.vv_init.argtypes = [i, C.POINTER(vp)]
    lib.vv_shutdown.argtypes = [vp]
    lib.vv_last_error.argtypes = [vp]; lib.vv_last_error.restype = C.c_char_p
    lib.vv_load_volume_u8.argtypes = [vp, vp, sz, i, i, i, vp]
    lib.vv_load_volume_f32.argtypes = [vp, vp, sz, i, i, i, vp]
    lib.vv_load_volume_device.argtypes = [vp, vp, i, i, i, i, vp, vp]
    lib.vv_set_transfer_function.argtypes = [vp, vp]
    lib.vv_load_volume_stream_begin.argtypes = [vp, i, i, i, i, vp]
    lib.vv_load_volume_stream_slices.argtypes = [vp, vp, i, i, i]
    lib.vv_load_volume_stream_slices_async.argtypes = [vp, vp, i, i, i]
    lib.vv_load_volume_stream_wait_source.argtypes = [vp]
    lib.vv_load_volume_stream_end.argtypes = [vp]
    lib.vv_load_volume_t3d.argtypes = [vp, C.c_char_p, i, i, vp]
    lib.vv_render.argtypes = [vp, i, i, C.POINTER(slice_params), C.POINTER(camera_params),
                              C.POINTER(shading_params), C.POINTER(vv_ray_source),
                              C.POINTER(vv_render_options), vp, i, vp]
    lib.vv_slice.argtypes = [vp, vp, sz, sz, f, f, f, i, C.POINTER(f * 3), i, i, i, vp]
    lib.vv_slice_advanced.argtypes = [vp, vp, sz, sz, C.POINTER(f * 16), C.POINTER(f * 3), i, i, vp]
    lib.vv_generate_ellipsoids.argtypes = [vp, vp, i, i, i, i, i, vp, vp, vp, vp]
    lib.vv_generate_default_brain.argtypes = [vp, vp, i, i, i, i, vp]
    lib.vv_draw_ellipsoid.argtypes = [vp, vp, i, i, i, i, vp, vp, C.c_uint8, vp]
    lib.vv_promote_u8_to_f32.argtypes = [vp, vp, vp, sz, vp]
    lib.vv_generate_noise_u8.argtypes = [vp, vp, i, i, i, C.c_uint32, vp]
    lib.vv_transfer_preset.argtypes = [i, vp]
    lib.vv_dataset_preset.argtypes = [C.c_char_p, vp, vp]
    lib.vv_t3d_read_header.argtypes = [C.c_char_p, i, C.POINTER(i), C.POINTER(i), C.POINTER(i)]
    lib.vv_t3d_read.argtypes = [C.c_char_p, i, vp, sz]
    lib.vv_t3d_write.argtypes = [C.c_char_p, i, vp, i, i, i]
    lib.vv_last_frame_ms.argtypes = [vp]; lib.vv_last_frame_ms.restype = f
    lib.vv_last_sample_count.argtypes = [vp]; lib.vv_last_sample_count.restype = C.c_ulonglong
    lib.vv_slice_matrix.argtypes = [f, f, f, f, f, f, vp]
    lib.vv_first_pass.argtypes = [vp, i, i, C.POINTER(camera_params), C.POINTER(vv_ray_source), vp, vp, i, vp]
    lib.vv_cut_plane_canonical.argtypes = [i, f, vp, vp]
    lib.vv_cut_plane_from_euler.argtypes = [f, f, f, f, f, f, vp, vp]
    lib.vv_cut_plane_to_slice_params.argtypes = [i, vp, vp, i, C.POINTER(slice_params)]
    lib.vv_slice_to_bgra.argtypes = [vp, sz, sz, vp]
    lib.vv_camera_orbit_drag.argtypes = [vp, i, i, vp, vp]
    lib.vv_camera_zoom.argtypes = [vp, vp, i, vp]
    lib.vv_cut_plane_from_drag.argtypes = [vp, vp, vp, f, vp, vp, vp, vp, vp, vp]
    lib.vv_cut_plane_drag.argtypes = [vp, vp, vp, i, i, i, i]
    lib.vv_debug_counters.argtypes = [vp, vp]
    lib.vv_debug_last_launch.argtypes = [vp, vp]
    lib.vv_set_frame_timing.argtypes = [vp, i]
    lib.vv_debug_screen_rect.argtypes = [i, i, C.POINTER(camera_params), C.POINTER(vv_ray_source), C.POINTER(C.c_double * 4)]
    lib.vv_reread_env.argtypes = [vp]
    lib.vv_prepare_layouts.argtypes = [vp, i, vp]
    lib.vv_device_bytes.argtypes = [vp, vp]
    lib.vv_set_layout_policy.argtypes = [vp, C.c_ulonglong, i]
    lib.vv_layout_state.argtypes = [vp, vp]
    lib.vv_volume_dims.argtypes = [vp, C.POINTER(i * 3), C.POINTER(i)]
    for name in EXPORTS:
        fn = getattr(lib, name)
        if name not in ("vv_last_error", "vv_last_frame_ms", "vv_last_sample_count"):
            fn.restype = i
    if path is None:
        _lib = lib
    else:
        _libs[path] = lib
    return lib


class VolvizError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"volviz error {code}: {msg}")
        self.code = code


def transfer_preset(preset: int) -> np.ndarray:
    """vv_transfer_preset: transfer_functions.h:4-9 tables as float32[1024]."""
    tf = np.zeros(1024, np.float32)
    rc = load_library().vv_transfer_preset(preset, tf.ctypes.data)
    if rc:
        raise VolvizError(rc, "bad preset")
    return tf


def dataset_preset(path: str):
    """vv_dataset_preset: (tf_preset, (sx, sy, sz)) by the file name's ending (glwidget.cpp:678-689), or None."""
    tfp = C.c_int(-1)
    sc = (C.c_float * 3)(0, 0, 0)
    rc = load_library().vv_dataset_preset(path.encode(), C.byref(tfp), sc)
    if rc < 0:
        raise VolvizError(rc, "bad argument")
    return (tfp.value, (sc[0], sc[1], sc[2])) if rc == 1 else None


@dataclass
class Camera:
    """The camera surface of glwidget.cpp:113-114,262-276,338-341 + camera.cpp."""
    origin: Sequence[float] = (0.0, 0.0, -4.0)
    look_at: Sequence[float] = (0.0, 0.0, 0.0)
    up: Sequence[float] = (0.0, 1.0, 0.0)
    fov_y: float = 45.0
    scale: Sequence[float] = (1.0, 1.0, 1.0)

    def params(self, width: int, height: int) -> camera_params:
        aspect = np.float32(width) / np.float32(height)
        cp = camera_params()
        cp.origin[:] = [float(v) for v in self.origin]
        cp.fovY = float(self.fov_y)
        cp.fovX = float(np.float32(self.fov_y) * aspect)        # glwidget.cpp:341
        cp.scale[:] = [float(v) for v in self.scale]
        return cp

    def look(self):
        return [float(np.float32(t) - np.float32(o)) for t, o in zip(self.look_at, self.origin)]

    @staticmethod
    def orbit(r: float, theta: float, phi: float, **kw) -> "Camera":
        """glwidget.cpp:435-445 orbit position."""
        return Camera(origin=(r * np.sin(theta) * np.cos(phi), r * np.cos(theta),
                              r * np.sin(theta) * np.sin(phi)), **kw)


def screen_rect(cam: "Camera", width: int, height: int):
    """vv_debug_screen_rect: (x_min, x_max, y_min, y_max) in pixel coordinates, or None when this camera gets no rectangle.  No device needed."""
    out = (C.c_double * 4)()
    cp = cam.params(width, height); rs = analytic_rays(cam)
    rc = load_library().vv_debug_screen_rect(width, height, C.byref(cp), C.byref(rs), C.byref(out))
    if rc < 0:
        raise VolvizError(rc, "vv_debug_screen_rect")
    return tuple(out) if rc == 1 else None


def analytic_rays(cam: Camera, quantize8: bool = False, aspect: float = 0.0) -> vv_ray_source:
    rs = vv_ray_source()
    rs.mode = RAYS_ANALYTIC
    rs.look[:] = cam.look()
    rs.up[:] = [float(v) for v in cam.up]
    rs.aspect = aspect
    rs.quantize8 = int(quantize8)
    return rs


def _hint(rs: vv_ray_source, cam: Optional["Camera"], aspect: float) -> vv_ray_source:
    if cam is not None:                     # the camera that drew the images: a launch-policy hint (include/volviz.h)
        rs.look[:] = cam.look()
        rs.up[:] = [float(v) for v in cam.up]
        rs.aspect = aspect
    return rs


def image_rays(front: np.ndarray, back: np.ndarray, hint: Optional["Camera"] = None, aspect: float = 0.0) -> vv_ray_source:
    """front/back: uint8 [H_fbo, W_fbo, 4] host arrays (kept alive by the caller)."""
    assert front.dtype == np.uint8 and back.dtype == np.uint8 and front.shape == back.shape
    rs = vv_ray_source()
    rs.mode = RAYS_IMAGES
    rs.front = front.ctypes.data
    rs.back = back.ctypes.data
    rs.img_h, rs.img_w = front.shape[0], front.shape[1]
    rs.images_on_device = 0
    return _hint(rs, hint, aspect)


def device_image_rays(front_ptr: int, back_ptr: int, img_w: int, img_h: int, hint: Optional["Camera"] = None, aspect: float = 0.0) -> vv_ray_source:
    """Two device-resident RGBA8 images (e.g. written by Context.first_pass_device)."""
    rs = vv_ray_source()
    rs.mode = RAYS_IMAGES
    rs.front = front_ptr
    rs.back = back_ptr
    rs.img_h, rs.img_w = img_h, img_w
    rs.images_on_device = 1
    return _hint(rs, hint, aspect)


def make_slice_params(slice_type: int = SLICE_NONE, point=(0.5, 0.5, 0.5), normal=(0.0, 0.0, 1.0)) -> slice_params:
    sp = slice_params()
    sp.type = slice_type
    sp.params[:] = [float(v) for v in (*point, *normal)]
    return sp


def make_options(step=None, ert_threshold=0.0, filter=FILTER_TEX8, ert_mode=ERT_REFERENCE,
                 slab_rows=(0, 0), shard=None, count_samples=False, touched_bricks=0, touched_lines=0, touched_line_bits=0,
                 touched_lines_all=False, touched_block_lines=0, touched_block_lines_log2=0) -> vv_render_options:
    o = vv_render_options()
    if step is not None:
        s = [step] * 3 if np.isscalar(step) else list(step)
        o.step[:] = [float(v) for v in s]
    o.ert_threshold = ert_threshold
    o.filter = filter
    o.ert_mode = ert_mode
    o.slab_row_begin, o.slab_row_end = slab_rows
    if shard is not None:                       # (band, count, index)
        o.shard_band, o.shard_count, o.shard_index = shard
    o.count_samples = int(count_samples)
    o.touched_bricks = touched_bricks
    o.touched_lines = touched_lines
    o.touched_line_bits = touched_line_bits
    o.touched_lines_all = int(touched_lines_all)
    o.touched_block_lines = touched_block_lines
    o.touched_block_lines_log2 = touched_block_lines_log2
    return o


class Context:
    """One vv_context: one device, one volume, one transfer function (kernel.cu:35-51)."""

    def __init__(self, device: int = -1, lib_path: Optional[str] = None):
        self.lib = load_library(lib_path)
        h = C.c_void_p()
        rc = self.lib.vv_init(device, C.byref(h))
        if rc:
            raise VolvizError(rc, (self.lib.vv_last_error(None) or b"").decode())
        self.h = h

    def _chk(self, rc: int):
        if rc:
            raise VolvizError(rc, (self.lib.vv_last_error(self.h) or b"").decode())

    def close(self):
        if getattr(self, "h", None):
            self.lib.vv_shutdown(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # cudaLoadVolume (kernel.cu:456-498)
    def load_volume(self, vol: np.ndarray, tf: np.ndarray):
        """vol: [nz, ny, nx] uint8 or float32 (x fastest); tf: float32[1024]."""
        assert vol.ndim == 3 and vol.flags.c_contiguous
        tf = np.ascontiguousarray(tf, np.float32).reshape(1024)
        nz, ny, nx = vol.shape
        if vol.dtype == np.uint8:
            self._chk(self.lib.vv_load_volume_u8(self.h, vol.ctypes.data, vol.nbytes, nx, ny, nz, tf.ctypes.data))
        elif vol.dtype == np.float32:
            self._chk(self.lib.vv_load_volume_f32(self.h, vol.ctypes.data, vol.nbytes, nx, ny, nz, tf.ctypes.data))
        else:
            raise TypeError("volume must be uint8 or float32")

    def load_volume_device(self, dev_ptr: int, voxel_type: int, nx: int, ny: int, nz: int, tf: np.ndarray, stream: int = 0):
        tf = np.ascontiguousarray(tf, np.float32).reshape(1024)
        self._chk(self.lib.vv_load_volume_device(self.h, dev_ptr, voxel_type, nx, ny, nz, tf.ctypes.data, stream))

    def load_volume_streamed(self, slabs, voxel_type: int, nx: int, ny: int, nz: int, tf: np.ndarray):
        """vv_load_volume_stream_*: `slabs` yields (z0, array[nslices, ny, nx]) in any order; u8 slabs
        into an f32 volume are promoted on the device."""
        tf = np.ascontiguousarray(tf, np.float32).reshape(1024)
        self._chk(self.lib.vv_load_volume_stream_begin(self.h, voxel_type, nx, ny, nz, tf.ctypes.data))
        for z0, arr in slabs:
            arr = np.ascontiguousarray(arr)
            st = VOXEL_U8 if arr.dtype == np.uint8 else VOXEL_F32
            self._chk(self.lib.vv_load_volume_stream_slices(self.h, arr.ctypes.data, st, z0, arr.shape[0]))
        self._chk(self.lib.vv_load_volume_stream_end(self.h))

    def load_volume_t3d(self, path: str, header: bool, voxel_type: int, tf: np.ndarray):
        tf = np.ascontiguousarray(tf, np.float32).reshape(1024)
        self._chk(self.lib.vv_load_volume_t3d(self.h, path.encode(), int(header), voxel_type, tf.ctypes.data))

    def set_transfer_function(self, tf: np.ndarray):
        tf = np.ascontiguousarray(tf, np.float32).reshape(1024)
        self._chk(self.lib.vv_set_transfer_function(self.h, tf.ctypes.data))

    # runCuda (kernel.cu:388-453)
    def render(self, width: int, height: int, cam: Camera, *, slice: Optional[slice_params] = None,
               phong: bool = False, rays: Optional[vv_ray_source] = None,
               options: Optional[vv_render_options] = None, out: Optional[np.ndarray] = None,
               fill: int = 0) -> np.ndarray:
        """Host-buffer render; returns uint8 [H, W, 4] (row 0 = bottom)."""
        if out is None:
            out = np.full((height, width, 4), fill, np.uint8)
        sp = slice if slice is not None else make_slice_params()
        cp = cam.params(width, height)
        sh = shading_params(-1, phong)
        rs = rays if rays is not None else analytic_rays(cam)
        self._chk(self.lib.vv_render(self.h, width, height, C.byref(sp), C.byref(cp), C.byref(sh), C.byref(rs),
                                     C.byref(options) if options is not None else None,
                                     out.ctypes.data, 0, None))
        return out

    def render_device(self, width: int, height: int, cam: Camera, out_ptr: int, *, slice=None, phong=False,
                      rays=None, options=None, stream: int = 0):
        """Device-buffer render enqueued on `stream` (a hipStream_t as int)."""
        sp = slice if slice is not None else make_slice_params()
        cp = cam.params(width, height)
        sh = shading_params(-1, phong)
        rs = rays if rays is not None else analytic_rays(cam)
        self._chk(self.lib.vv_render(self.h, width, height, C.byref(sp), C.byref(cp), C.byref(sh), C.byref(rs),
                                     C.byref(options) if options is not None else None,
                                     out_ptr, 1, stream))

    def first_pass(self, img_w: int, img_h: int, cam: Camera):
        """vv_first_pass: the two RGBA8 FBO images of glwidget.cpp:200-228 for this camera."""
        front = np.zeros((img_h, img_w, 4), np.uint8); back = np.zeros((img_h, img_w, 4), np.uint8)
        cp = cam.params(img_w, img_h); rs = analytic_rays(cam)
        self._chk(self.lib.vv_first_pass(self.h, img_w, img_h, C.byref(cp), C.byref(rs), front.ctypes.data, back.ctypes.data, 0, None))
        return front, back

    def first_pass_device(self, img_w: int, img_h: int, cam: Camera, front_ptr: int, back_ptr: int, stream: int = 0):
        """vv_first_pass into two device buffers of img_w * img_h * 4 bytes, enqueued on `stream`."""
        cp = cam.params(img_w, img_h); rs = analytic_rays(cam)
        self._chk(self.lib.vv_first_pass(self.h, img_w, img_h, C.byref(cp), C.byref(rs), front_ptr, back_ptr, 1, stream))

    def last_frame_ms(self) -> float:
        return float(self.lib.vv_last_frame_ms(self.h))

    def set_frame_timing(self, on: bool) -> None:
        """vv_set_frame_timing: the two hipEventRecord per vv_render behind last_frame_ms(), on (default) or off."""
        self._chk(self.lib.vv_set_frame_timing(self.h, int(bool(on))))

    def debug_counters(self):
        out = np.zeros(16, np.uint64)
        self._chk(self.lib.vv_debug_counters(self.h, out.ctypes.data))
        return out

    def last_launch(self) -> dict:
        """vv_debug_last_launch: what the launch policy chose for the last render."""
        out = (C.c_int * 8)()
        self._chk(self.lib.vv_debug_last_launch(self.h, out))
        k = ("tile_log2w", "blk_log2w", "unroll", "lds_reserve", "layout", "view_known", "density_x1000", "phong")
        return dict(zip(k, list(out)))

    def reread_env(self):
        """vv_reread_env: pick up VV_* knobs changed since the last volume load."""
        self._chk(self.lib.vv_reread_env(self.h))

    def last_sample_count(self) -> int:
        return int(self.lib.vv_last_sample_count(self.h))

    def prepare_layouts(self, which: int = 3, stream=None) -> int:
        """vv_prepare_layouts: build the bricked (1) / z-pair (2) / z-fastest (4, with the x-pair copy) copies now, or LAYOUT_POLICY = every copy
        the launch policy can pick for the loaded volume; returns the resident mask."""
        rc = self.lib.vv_prepare_layouts(self.h, which, stream)
        if rc < 0:
            self._chk(rc)
        return rc

    def set_layout_policy(self, budget_bytes: int = 0, build_in_render: bool = True):
        """vv_set_layout_policy: HBM budget of the optional copies (0 = default) and whether vv_render may build a missing one."""
        self._chk(self.lib.vv_set_layout_policy(self.h, int(budget_bytes), int(build_in_render)))

    def layout_state(self) -> dict:
        """vv_layout_state: resident bytes per layout, the budget, copies built inside vv_render since the load."""
        out = np.zeros(8, np.uint64)
        self._chk(self.lib.vv_layout_state(self.h, out.ctypes.data))
        k = ("linear", "bricked", "zpair", "zfast", "xpair", "budget", "builds_in_render", "build_in_render")
        return dict(zip(k, (int(v) for v in out)))

    def device_bytes(self):
        """vv_device_bytes: [linear volume, bricked copy, z-pair + z-fastest copies, tables + scratch]."""
        out = np.zeros(4, np.uint64)
        self._chk(self.lib.vv_device_bytes(self.h, out.ctypes.data))
        return [int(v) for v in out]

    # invoke_slice_kernel (kernel.cu:506-519) / slicekernel.cu legacy
    def slice(self, height: int, width: int, dx=0.0, dy=0.0, dz=0.0, orientation=SAGITTAL,
              scale=(1.0, 1.0, 1.0), legacy=False, filter=FILTER_TEX8, fill=0.0) -> np.ndarray:
        buf = np.full(height * width, fill, np.float32)
        sc = (C.c_float * 3)(*[float(v) for v in scale])
        self._chk(self.lib.vv_slice(self.h, buf.ctypes.data, height, width, dx, dy, dz, orientation,
                                    C.byref(sc), int(legacy), filter, 0, None))
        return buf

    # invoke_advanced_slice_kernel (kernel.cu:522-541)
    def slice_advanced(self, height: int, width: int, trans: np.ndarray, scale=(1.0, 1.0, 1.0),
                       filter=FILTER_TEX8, fill=0.0) -> np.ndarray:
        buf = np.full(height * width, fill, np.float32)
        t = (C.c_float * 16)(*[float(v) for v in np.asarray(trans, np.float32).reshape(16)])
        sc = (C.c_float * 3)(*[float(v) for v in scale])
        self._chk(self.lib.vv_slice_advanced(self.h, buf.ctypes.data, height, width, C.byref(t), C.byref(sc),
                                             filter, 0, None))
        return buf

    # VolumeGenerator::drawEllipsoid x n / drawDefaultBrain (volumegenerator.cpp:31-119)
    def generate_ellipsoids(self, nx: int, ny: int, nz: int, centers, axes, colors) -> np.ndarray:
        centers = np.ascontiguousarray(centers, np.float32).reshape(-1, 3)
        axes = np.ascontiguousarray(axes, np.float32).reshape(-1, 3)
        colors = np.ascontiguousarray(colors, np.uint8).reshape(-1)
        out = np.zeros((nz, ny, nx), np.uint8)
        self._chk(self.lib.vv_generate_ellipsoids(self.h, out.ctypes.data, 0, nx, ny, nz, len(colors),
                                                  centers.ctypes.data, axes.ctypes.data, colors.ctypes.data, None))
        return out

    def draw_ellipsoid(self, vol: np.ndarray, center, axes, color: int) -> np.ndarray:
        """vv_draw_ellipsoid: one VolumeGenerator::drawEllipsoid applied in place to vol [nz,ny,nx]."""
        assert vol.dtype == np.uint8 and vol.flags.c_contiguous
        nz, ny, nx = vol.shape
        c = np.ascontiguousarray(center, np.float32); a = np.ascontiguousarray(axes, np.float32)
        self._chk(self.lib.vv_draw_ellipsoid(self.h, vol.ctypes.data, 0, nx, ny, nz, c.ctypes.data, a.ctypes.data,
                                             int(color), None))
        return vol

    def generate_default_brain(self, nx: int, ny: int, nz: int) -> np.ndarray:
        out = np.zeros((nz, ny, nx), np.uint8)
        self._chk(self.lib.vv_generate_default_brain(self.h, out.ctypes.data, 0, nx, ny, nz, None))
        return out

    def generate_default_brain_device(self, dev_ptr: int, nx: int, ny: int, nz: int, stream: int = 0):
        self._chk(self.lib.vv_generate_default_brain(self.h, dev_ptr, 1, nx, ny, nz, stream))

    def generate_noise_device(self, dev_ptr: int, nx: int, ny: int, nz: int, seed: int, stream: int = 0):
        self._chk(self.lib.vv_generate_noise_u8(self.h, dev_ptr, nx, ny, nz, seed, stream))

    def promote_device(self, dev_in: int, dev_out: int, n: int, stream: int = 0):
        self._chk(self.lib.vv_promote_u8_to_f32(self.h, dev_in, dev_out, n, stream))


def cut_plane_canonical(orientation: int, displace: float):
    """vv_cut_plane_canonical: GLWidget::setSliceCanonical (glwidget.cpp:743-788)."""
    pt = np.zeros(3, np.float32); n = np.zeros(3, np.float32)
    rc = load_library().vv_cut_plane_canonical(orientation, displace, pt.ctypes.data, n.ctypes.data)
    if rc:
        raise VolvizError(rc, "bad orientation")
    return pt, n


def cut_plane_from_euler(dx, dy, dz, theta, phi, psi):
    """vv_cut_plane_from_euler: the free-form slice view's cutting plane (window.cpp:425-443 -> GLWidget::setSlicePro) -> (point, normal)."""
    pt = np.zeros(3, np.float32); n = np.zeros(3, np.float32)
    rc = load_library().vv_cut_plane_from_euler(dx, dy, dz, theta, phi, psi, pt.ctypes.data, n.ctypes.data)
    if rc:
        raise VolvizError(rc, "bad argument")
    return pt, n


def cut_plane_to_slice_params(slice_type: int, point, normal, flip: bool = False) -> slice_params:
    """vv_cut_plane_to_slice_params: glwidget.cpp:232-258."""
    sp = slice_params()
    pt = np.ascontiguousarray(point, np.float32); n = np.ascontiguousarray(normal, np.float32)
    rc = load_library().vv_cut_plane_to_slice_params(slice_type, pt.ctypes.data, n.ctypes.data, int(flip), C.byref(sp))
    if rc:
        raise VolvizError(rc, "bad slice type")
    return sp


def _f3(v) -> np.ndarray:
    return np.ascontiguousarray(v, np.float32).reshape(3).copy()


def camera_orbit_drag(position, dx: int, dy: int):
    """vv_camera_orbit_drag: right-button drag (glwidget.cpp:432-446) -> (position, look)."""
    p = _f3(position); out = np.zeros(3, np.float32); look = np.zeros(3, np.float32)
    rc = load_library().vv_camera_orbit_drag(p.ctypes.data, dx, dy, out.ctypes.data, look.ctypes.data)
    if rc:
        raise VolvizError(rc, "camera at the origin")
    return out, look


def camera_zoom(position, look, delta: int) -> np.ndarray:
    """vv_camera_zoom: wheel (glwidget.cpp:607-620)."""
    p = _f3(position); l = _f3(look); out = np.zeros(3, np.float32)
    rc = load_library().vv_camera_zoom(p.ctypes.data, l.ctypes.data, delta, out.ctypes.data)
    if rc:
        raise VolvizError(rc, "bad argument")
    return out


def cut_plane_from_drag(position, look, up, aspect: float, press, release):
    """vv_cut_plane_from_drag: left-button drag released (glwidget.cpp:482-535)
    -> (point, normal, plane_up, plane_right)."""
    p = _f3(position); l = _f3(look); u = _f3(up)
    a = np.ascontiguousarray(press, np.float32).reshape(2).copy()
    b = np.ascontiguousarray(release, np.float32).reshape(2).copy()
    out = [np.zeros(3, np.float32) for _ in range(4)]
    rc = load_library().vv_cut_plane_from_drag(p.ctypes.data, l.ctypes.data, u.ctypes.data, aspect, a.ctypes.data,
                                               b.ctypes.data, *[o.ctypes.data for o in out])
    if rc:
        raise VolvizError(rc, "bad argument")
    return tuple(out)


def cut_plane_drag(point, plane_up, plane_right, dx: int, dy: int, width: int, height: int) -> np.ndarray:
    """vv_cut_plane_drag: middle-button drag of an image plane (glwidget.cpp:447-452)."""
    p = _f3(point); u = _f3(plane_up); r = _f3(plane_right)
    rc = load_library().vv_cut_plane_drag(p.ctypes.data, u.ctypes.data, r.ctypes.data, dx, dy, width, height)
    if rc:
        raise VolvizError(rc, "bad argument")
    return p


def slice_to_bgra(buf: np.ndarray, height: int, width: int, fill: int = 0) -> np.ndarray:
    """vv_slice_to_bgra: the image SliceWidget shows (slicewidget.cpp:108-121)."""
    buf = np.ascontiguousarray(buf, np.float32)
    out = np.full((height * width, 4), fill, np.uint8)
    rc = load_library().vv_slice_to_bgra(buf.ctypes.data, height, width, out.ctypes.data)
    if rc:
        raise VolvizError(rc, "bad argument")
    return out


def slice_matrix(dx, dy, dz, theta, phi, psi) -> np.ndarray:
    """vv_slice_matrix: SliceWidget::getTransformationMatrix (slicewidget.cpp:147-165)."""
    m = np.zeros(16, np.float32)
    rc = load_library().vv_slice_matrix(dx, dy, dz, theta, phi, psi, m.ctypes.data)
    if rc:
        raise VolvizError(rc, "angles must be in [-3.2, 3.2)")
    return m.reshape(4, 4)
