"""The C++ host mirror (volume-viz_amd/host/kernel_hip.h: initCuda / cudaLoadVolume / runCuda /
invoke_*_slice_kernel / VolumeGenerator with the reference's names) driven the way
glwidget.cpp and slicewidget.cpp drive kernel.cuh, checked against the oracle."""
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O
import volviz_amd as vv

REPO = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
DEMO = os.path.join(REPO, "volume-viz_amd", "bin", "host_demo")


def test_host_mirror_library_symbols():
    """CPU check: the mirror exports the reference's entry-point names."""
    so = os.path.join(REPO, "volume-viz_amd", "lib", "libvolviz_host.so")
    out = subprocess.check_output(["nm", "-D", "--defined-only", so]).decode()
    for name in ("initCuda", "runCuda", "cudaLoadVolume", "registerCudaResources"):
        assert f" T {name}\n" in out, name
    for frag in ("invoke_slice_kernel", "invoke_advanced_slice_kernel", "VolumeGenerator13drawEllipsoid",
                 "VolumeGenerator16drawDefaultBrain", "VolumeGenerator10saveas_raw", "VolumeGenerator12loadfrom_raw"):
        assert frag in out, frag


@pytest.mark.gpu
def test_host_demo_matches_oracle(tmp_path):
    W, H, fw, fh = 85, 60, 255, 180
    f32 = np.float32
    cam = vv.Camera(origin=(float(f32(3.2360680) * f32(0.8660254)), 2.0, float(f32(2.3511410) * f32(0.8660254))),
                    scale=(1.0, 1.0, 0.8))
    rs_hi = vv.analytic_rays(cam, quantize8=True)
    front = np.zeros((fh, fw, 4), np.uint8); back = np.zeros((fh, fw, 4), np.uint8)
    for y in range(fh):
        for x in range(fw):
            f, b = O.ray_endpoints(rs_hi, cam, fw, fh, x, y)
            front[y, x, :3] = np.round(f * 255); back[y, x, :3] = np.round(b * 255)
    front[..., 3] = back[..., 3] = 255
    front.tofile(tmp_path / "front.rgba"); back.tofile(tmp_path / "back.rgba")
    r = subprocess.run([DEMO, str(tmp_path), str(tmp_path / "front.rgba"), str(tmp_path / "back.rgba"),
                        str(fw), str(fh), str(W), str(H)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    vol = O.draw_default_brain(48, 40, 56)
    assert np.array_equal(np.fromfile(tmp_path / "volume.u8", np.uint8).reshape(56, 40, 48), vol)
    tf = O.transfer_preset(vv.TF_ENGINE)
    sp = vv.make_slice_params(vv.SLICE_PLANE, (0.5, 0.5, 0.45), (0.2, -0.3, 0.93))
    want, _ = O.render(vol, tf, W, H, cam, slice=sp, phong=True, rays=vv.image_rays(front, back), fill=0x5A)
    got = np.fromfile(tmp_path / "frame.rgba", np.uint8).reshape(H, W, 4)
    assert np.array_equal(got, want)
    assert (got[..., 3] > 0).mean() > 0.1
    s = np.fromfile(tmp_path / "slice_coronal.f32", np.float32)
    assert np.array_equal(s, O.slice(vol, 256, 256, 0.05, 0.4, 0.3, vv.CORONAL, (1.0, 1.0, 0.8)))
    s = np.fromfile(tmp_path / "slice_free.f32", np.float32)
    m = O.slice_matrix(0.1, -0.05, 0.02, 0.4, -0.3, 0.2)
    assert np.array_equal(s, O.slice_advanced(vol, 256, 256, m, (1.0, 1.0, 0.8)))
    # the .t3d the mirror wrote is what the reference's loader expects
    raw = open(tmp_path / "demo.t3d", "rb").read()
    assert np.frombuffer(raw[:24], "<u8").tolist() == [48, 40, 56] and raw[24:] == vol.tobytes()


def test_native_library_symbols():
    """CPU check: the optional native libraries export what their headers declare -- the GL-interop half of the
    kernel.cuh mirror (kernel.cu:375-453 over hip_gl_interop.h; compiled here, run only by a GL host) and the
    one-process multi-GPU entry points of include/volviz_mgpu.h."""
    import re
    lib = os.path.join(REPO, "volume-viz_amd", "lib")
    out = subprocess.check_output(["nm", "-D", "--defined-only", os.path.join(lib, "libvolviz_host_gl.so")]).decode()
    for name in ("registerCudaResourcesGL", "runCudaGL", "unregisterCudaResourcesGL"):
        assert f" T {name}\n" in out, name
    und = subprocess.check_output(["nm", "-D", "--undefined-only", os.path.join(lib, "libvolviz_host_gl.so")]).decode()
    for name in ("hipGraphicsGLRegisterImage", "hipGraphicsGLRegisterBuffer", "hipGraphicsMapResources", "hipGraphicsResourceGetMappedPointer", "vv_render"):
        assert name in und, name
    out = subprocess.check_output(["nm", "-D", "--defined-only", os.path.join(lib, "libvolviz_mgpu.so")]).decode()
    hdr = open(os.path.join(REPO, "include", "volviz_mgpu.h")).read()
    declared = set(re.findall(r"\b(vv_mgpu_\w+)\s*\(", hdr))
    assert len(declared) >= 10
    for name in declared:
        assert f" T {name}\n" in out, name
    und = subprocess.check_output(["nm", "-D", "--undefined-only", os.path.join(lib, "libvolviz_mgpu.so")]).decode()
    for name in ("ncclCommInitAll", "ncclSend", "ncclRecv", "ncclGroupStart", "ncclGroupEnd"):
        assert name in und, name


@pytest.mark.gpu
def test_native_multi_gpu_entry_points_on_the_visible_gpus():
    """bin/mgpu_demo: vv_mgpu_init on every visible GPU (one on the test box: the gather degenerates, the per-device
    contexts, the on-device volume generation, streams and the host staging are exercised), frame identical to
    vv_render's.  With more GPUs visible the same binary checks the RCCL gather."""
    demo = os.path.join(REPO, "volume-viz_amd", "bin", "mgpu_demo")
    r = subprocess.run([demo, "0", "96", "400", "300"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert " 0 differing bytes" in r.stdout, r.stdout


def test_paint_loop_symbols():
    """CPU check: the GLWidget-shaped loop (host/paint_loop.h) is part of the mirror library."""
    so = os.path.join(REPO, "volume-viz_amd", "lib", "libvolviz_host.so")
    out = subprocess.check_output(["nm", "-D", "-C", "--defined-only", so]).decode()
    for name in ("PaintLoop::paintGL()", "PaintLoop::resizeGL(int, int)", "PaintLoop::orbitDrag(int, int)", "PaintLoop::setSliceCanonical(int, float)", "PaintLoop::setSlicePro(float, float, float, float, float, float)",
                 "PaintLoop::loadVolume(char const*)"):
        assert name in out, name


@pytest.mark.gpu
def test_paint_loop_matches_oracle(tmp_path):
    """host/paint_loop.cpp = GLWidget::paintGL / resizeGL (glwidget.cpp:188-391) without Qt and GL: first pass into two
    widget-sized images, runCuda at a third of the size only when the frame is dirty, render-time value.  bin/paint_loop_demo
    plays a short session (first paint; a clean repaint; orbit drag + Phong; coronal cross-section + zoom); every frame it
    showed equals the oracle's march over the first-pass images it used, and those equal the oracle's first pass."""
    demo = os.path.join(REPO, "volume-viz_amd", "bin", "paint_loop_demo")
    ww, wh = 510, 384
    r = subprocess.run([demo, str(tmp_path), str(ww), str(wh)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "4 of 5 paints marched" in r.stdout, r.stdout
    W, H = ww // 3, wh // 3
    vol = O.draw_default_brain(64, 64, 64)
    tf = O.transfer_preset(vv.TF_ENGINE)
    pos0 = np.array([0.0, 0.0, -4.0], np.float32)
    pos1, look1 = vv.camera_orbit_drag(pos0, 37, -21)
    pos2 = vv.camera_zoom(pos1, look1, 60)
    pt, nrm = vv.cut_plane_canonical(vv.CORONAL, 0.1)
    sp2 = vv.cut_plane_to_slice_params(vv.SLICE_PLANE_CUT, pt, nrm, False)
    ppt, pnrm = vv.cut_plane_from_euler(0.05, -0.1, 0.02, 0.4, -0.7, 1.1)       # PaintLoop::setSlicePro(dx, dy, dz, theta, phi, psi): window.cpp:425-443
    sp4 = vv.cut_plane_to_slice_params(vv.SLICE_PLANE_CUT, ppt, pnrm, False)
    sessions = [(0, pos0, -pos0, None, False), (1, pos1, look1, None, True), (2, pos2, look1, sp2, True), (4, pos2, look1, sp4, True)]
    lit = 0
    for k, pos, look, sp, phong in sessions:
        front = np.fromfile(tmp_path / f"front{k}.rgba", np.uint8).reshape(wh, ww, 4)
        back = np.fromfile(tmp_path / f"back{k}.rgba", np.uint8).reshape(wh, ww, 4)
        cam_w = vv.Camera(origin=tuple(float(v) for v in pos))
        rs = vv.analytic_rays(cam_w, aspect=ww / wh)
        rs.look[:] = [float(v) for v in look]                  # (after the zoom the view direction is no longer -position)
        of, ob = O.first_pass(cam_w, ww, wh, rays=rs)
        assert np.array_equal(front, of) and np.array_equal(back, ob), f"first pass {k}"
        got = np.fromfile(tmp_path / f"frame{k}.rgba", np.uint8).reshape(H, W, 4)
        want, _ = O.render(vol, tf, W, H, cam_w, slice=sp, phong=phong, rays=vv.image_rays(front, back), fill=0)
        assert np.array_equal(got, want), f"frame {k}"
        lit += int((got[..., 3] > 0).sum())
    assert lit > 4 * W * H // 20


def test_dataset_presets_follow_the_file_name():
    """GLWidget::loadVolume picks table and scale by the file name's ending (glwidget.cpp:678-689): engine.t3d -> Engine table,
    (1, 1, 1); head.t3d -> Engine table, (1, 1, 0.8); VisMale.t3d -> Head table, (1.57, 1, 1); anything else: no rule (the
    caller's table and scale stay).  QString::endsWith is a case-sensitive suffix match."""
    f32 = lambda *v: tuple(float(np.float32(x)) for x in v)
    assert vv.dataset_preset("/home/rmartens/shared/cs224textures/engine.t3d") == (vv.TF_ENGINE, f32(1, 1, 1))
    assert vv.dataset_preset("/home/rmartens/shared/cs224textures/head.t3d") == (vv.TF_ENGINE, f32(1, 1, 0.8))
    assert vv.dataset_preset("VisMale.t3d") == (vv.TF_HEAD, f32(1.57, 1, 1))
    assert vv.dataset_preset("my_subhead.t3d") == (vv.TF_ENGINE, f32(1, 1, 0.8))       # a suffix, not a base name
    for other in ("brain_16.t3d", "HEAD.T3D", "head.t3d.bak", "vismale.t3d", ""):
        assert vv.dataset_preset(other) is None, other


@pytest.mark.gpu
def test_paint_loop_load_volume(tmp_path):
    """PaintLoop::loadVolume = GLWidget::loadVolume (glwidget.cpp:668-710): camera back to (0, 0, -4), table and scale from the
    file name, loadfrom_raw(path, header) -> cudaLoadVolume, dirty, cutting plane dropped.  The committed 16^3 .t3d (written by the
    compiled reference) is loaded under the name *head.t3d: Engine table, scale (1, 1, 0.8); the frame equals the oracle's."""
    import shutil
    demo = os.path.join(REPO, "volume-viz_amd", "bin", "paint_loop_demo")
    src = os.path.join(REPO, "tests", "golden", "brain_16.t3d")
    t3d = tmp_path / "mri_head.t3d"
    shutil.copy(src, t3d)
    ww, wh = 510, 384
    r = subprocess.run([demo, str(tmp_path), str(ww), str(wh), str(t3d)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "5 of 6 paints marched" in r.stdout and "preset 0 scale 1 1 0.8" in r.stdout, r.stdout
    W, H = ww // 3, wh // 3
    raw = np.fromfile(src, np.uint8)
    dims = np.frombuffer(raw[:24].tobytes(), "<u8")
    vol = raw[24:].reshape(int(dims[2]), int(dims[1]), int(dims[0]))
    cam = vv.Camera(origin=(0.0, 0.0, -4.0), scale=(1.0, 1.0, float(np.float32(0.8))))
    front = np.fromfile(tmp_path / "front3.rgba", np.uint8).reshape(wh, ww, 4)
    back = np.fromfile(tmp_path / "back3.rgba", np.uint8).reshape(wh, ww, 4)
    of, ob = O.first_pass(cam, ww, wh, rays=vv.analytic_rays(cam, aspect=ww / wh))
    assert np.array_equal(front, of) and np.array_equal(back, ob)
    got = np.fromfile(tmp_path / "frame3.rgba", np.uint8).reshape(H, W, 4)
    # Phong stays as the session left it (on); the cross-section does not: hasCuttingPlane = false (:706)
    want, _ = O.render(vol, O.transfer_preset(vv.TF_ENGINE), W, H, cam, phong=True, rays=vv.image_rays(front, back), fill=0)
    assert np.array_equal(got, want)
    assert int((got[..., 3] > 0).sum()) > W * H // 20
