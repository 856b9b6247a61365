// paint_loop.cpp -- see paint_loop.h.  Line references are to the reference's glwidget.cpp.
#include "paint_loop.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>

static void die(const char *what, int rc)
{
    fprintf(stderr, "PaintLoop: %s failed: %d (%s)\n", what, rc, vv_last_error(volvizContext()));
    exit(EXIT_FAILURE);                         // the reference's checkCudaErrors behaviour (helper_cuda.h:763-777)
}

PaintLoop::PaintLoop()
{
    m_pos[0] = 0.f; m_pos[1] = 0.f; m_pos[2] = -4.f;           // :113-114
    m_look[0] = 0.f; m_look[1] = 0.f; m_look[2] = 4.f;
    m_up[0] = 0.f; m_up[1] = 1.f; m_up[2] = 0.f;
    m_scale[0] = m_scale[1] = m_scale[2] = 1.f;
    memset(m_cutPoint, 0, sizeof m_cutPoint); memset(m_cutNormal, 0, sizeof m_cutNormal);
}

void PaintLoop::loadVolume(const char *path)
{
    m_pos[0] = 0.f; m_pos[1] = 0.f; m_pos[2] = -4.f;           // :670-671
    m_look[0] = 0.f; m_look[1] = 0.f; m_look[2] = 4.f;
    int rc = vv_dataset_preset(path, &m_tfPreset, m_scale);     // :678-689
    if (rc < 0) die("vv_dataset_preset", rc);
    float transferFunction[1024];
    rc = vv_transfer_preset(m_tfPreset, transferFunction);
    if (rc) die("vv_transfer_preset", rc);
    VolumeGenerator volgen(0, 0, 0);                            // :674
    volgen.loadfrom_raw(path, true);                            // :692
    size_t size;
    byte *texels = volgen.getBytes(size);                       // :696
    cudaLoadVolume(texels, size, volgen.getDims(), transferFunction, &volumeArray);   // :700
    m_renderingDirty = true;                                    // :705-706
    m_hasCuttingPlane = false;
}

void PaintLoop::resizeGL(int width, int height)
{
    m_width = width; m_height = height;
    const float aspect = (float)width / (float)height;          // :333
    m_fovY = 45.f; m_fovX = m_fovY * aspect;                     // :338-341
    for (int i = 0; i < 2; ++i) m_fbo[i].assign((size_t)width * height * 4, 0);                      // :350-356
    const size_t rw = (size_t)(width / m_resolutionScale), rh = (size_t)(height / m_resolutionScale);
    m_resultBuffer.assign(rw * rh * 4, 0);                       // :362-366 (GL_STREAM_DRAW storage, contents undefined there)
    m_resultTexture.assign(rw * rh * 4, 0);                      // :370-379
    registerHostResources(m_fbo[0].data(), m_fbo[1].data(), width, height, m_resultBuffer.data());   // :390
    m_renderingDirty = true;
}

bool PaintLoop::paintGL()
{
    const int width = m_width, height = m_height;
    // ---- the two first-pass FBOs (:199-228): entry / exit points of the proxy cube, scaled by scaleObject ----
    camera_params cameraParams;
    memset(&cameraParams, 0, sizeof cameraParams);
    cameraParams.scale[0] = m_scale[0]; cameraParams.scale[1] = m_scale[1]; cameraParams.scale[2] = m_scale[2];   // :262-266
    cameraParams.origin[0] = m_pos[0]; cameraParams.origin[1] = m_pos[1]; cameraParams.origin[2] = m_pos[2];      // :270-272
    cameraParams.fovX = m_fovX; cameraParams.fovY = m_fovY;                                                        // :274-275
    {
        vv_ray_source rs;
        memset(&rs, 0, sizeof rs);
        rs.mode = VV_RAYS_ANALYTIC;
        rs.look[0] = m_look[0]; rs.look[1] = m_look[1]; rs.look[2] = m_look[2];
        rs.up[0] = m_up[0]; rs.up[1] = m_up[1]; rs.up[2] = m_up[2];
        rs.aspect = (float)width / (float)height;
        int rc = vv_first_pass(volvizContext(), width, height, &cameraParams, &rs, m_fbo[0].data(), m_fbo[1].data(), 0, nullptr);
        if (rc) die("vv_first_pass", rc);
    }
    // ---- slice parameters (:230-258) ----
    slice_params sliceParams;
    memset(&sliceParams, 0, sizeof sliceParams);
    if (m_hasCuttingPlane && m_sliceVis != 0) {
        int rc = vv_cut_plane_to_slice_params(m_sliceVis == 2 ? SLICE_PLANE_CUT : SLICE_PLANE, m_cutPoint, m_cutNormal, m_flip ? 1 : 0, &sliceParams);
        if (rc) die("vv_cut_plane_to_slice_params", rc);
    } else sliceParams.type = SLICE_NONE;
    shading_params shadingParams;                                 // :277-280
    shadingParams.transferPreset = TRANSFER_PRESET_DEFAULT;
    shadingParams.phongShading = m_phong;

    clock_t t = clock();                                          // :288
    const bool marched = m_renderingDirty;
    if (m_renderingDirty) {                                       // :290-292
        runCuda(width / m_resolutionScale, height / m_resolutionScale, sliceParams, cameraParams, shadingParams, volumeArray);
        ++m_runs;
    }
    m_resultTexture = m_resultBuffer;                             // glTexSubImage2D from the unpack buffer (:294-296), every paint
    if (m_renderingDirty) {                                       // :310-317 (glFinish: runCuda above has already fenced)
        t = clock() - t;
        m_lastRenderTime = (float)t / CLOCKS_PER_SEC;
        m_renderingDirty = false;
    }
    return marched;
}

void PaintLoop::setCameraPosition(float x, float y, float z)
{
    m_pos[0] = x; m_pos[1] = y; m_pos[2] = z;
    m_look[0] = -x; m_look[1] = -y; m_look[2] = -z;              // camera->lookAt(origin), :445
    m_renderingDirty = true;
}
void PaintLoop::orbitDrag(int dx, int dy)
{
    float p[3], l[3];
    int rc = vv_camera_orbit_drag(m_pos, dx, dy, p, l);
    if (rc) die("vv_camera_orbit_drag", rc);
    memcpy(m_pos, p, sizeof p); memcpy(m_look, l, sizeof l);
    m_renderingDirty = true;
}
void PaintLoop::zoom(int delta)
{
    float p[3];
    int rc = vv_camera_zoom(m_pos, m_look, delta, p);
    if (rc) die("vv_camera_zoom", rc);
    memcpy(m_pos, p, sizeof p);
    m_renderingDirty = true;
}
void PaintLoop::setScale(float x, float y, float z) { m_scale[0] = x; m_scale[1] = y; m_scale[2] = z; m_renderingDirty = true; }
void PaintLoop::setPhongShading(bool on) { m_phong = on; m_renderingDirty = true; }
void PaintLoop::setSliceVisualization(int vis) { m_sliceVis = vis; m_renderingDirty = true; }
void PaintLoop::setCuttingPlane(const float point[3], const float normal[3], bool flip)
{
    memcpy(m_cutPoint, point, sizeof m_cutPoint); memcpy(m_cutNormal, normal, sizeof m_cutNormal);
    m_flip = flip; m_hasCuttingPlane = true; m_renderingDirty = true;
}
void PaintLoop::setSliceCanonical(int orientation, float displace)
{
    if (m_sliceVis == 0) return;                                       // glwidget.cpp:761: only while a slice visualisation is on
    int rc = vv_cut_plane_canonical(orientation, displace, m_cutPoint, m_cutNormal);
    if (rc) die("vv_cut_plane_canonical", rc);
    m_flip = false; m_hasCuttingPlane = true; m_renderingDirty = true;
}
void PaintLoop::setSlicePro(const float offset[3], const float normal[3])
{
    if (m_sliceVis == 0) return;                                       // glwidget.cpp:745
    for (int a = 0; a < 3; ++a) { m_cutPoint[a] = offset[a]; m_cutNormal[a] = normal[a]; }
    m_flip = false; m_hasCuttingPlane = true; m_renderingDirty = true;  // :746-753
}
void PaintLoop::setSlicePro(float dx, float dy, float dz, float theta, float phi, float psi)
{
    float pt[3], n[3];
    int rc = vv_cut_plane_from_euler(dx, dy, dz, theta, phi, psi, pt, n);
    if (rc) die("vv_cut_plane_from_euler", rc);
    setSlicePro(pt, n);
}
void PaintLoop::clearCuttingPlane() { m_hasCuttingPlane = false; m_renderingDirty = true; }
void PaintLoop::setResolutionScale(int s) { if (s >= 1) { m_resolutionScale = s; if (m_width) resizeGL(m_width, m_height); } }
