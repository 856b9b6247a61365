// paint_loop.h -- the reference's GLWidget::paintGL / resizeGL loop (glwidget.cpp:188-325, 328-391) without Qt and
// without a GL context: same members, same order of work, same dirty-flag and render-time behaviour, calling the
// kernel.cuh mirror (kernel_hip.h) exactly where the reference does.
//
// What GL did in the reference and what stands in for it here:
//   two FBO passes drawing the proxy cube's front / back faces (paintGL :199-228, firstpass.vert/frag)
//        -> vv_first_pass() fills two host RGBA8 images of the widget's size (include/volviz.h)
//   registerCudaResources(fbo0 texture, fbo1 texture, pixel buffer) (resizeGL :390)
//        -> registerHostResources(front, back, width, height, resultBuffer)
//   glTexSubImage2D from the pixel-unpack buffer into resultTexture (:296)
//        -> a copy of resultBuffer into `resultTexture` (a host image of the render size)
//   drawTextureQuad / renderText overlay
//        -> not drawn; lastRenderTime and the resolution string are kept as members
// A GL host keeps its own paintGL and swaps registerCudaResources/runCuda for the *GL forms of kernel_hip_gl.cpp
// (INTEGRATION.md); this class is what can be executed and checked on a machine without a display.
#pragma once
#include <vector>
#include "kernel_hip.h"

class PaintLoop {
public:
    PaintLoop();
    // GLWidget::resizeGL (glwidget.cpp:328-391): FBO images, result buffer and texture for the new size, resources registered
    void resizeGL(int width, int height);
    // GLWidget::paintGL (glwidget.cpp:188-325).  Returns true when the frame was marched (renderingDirty was set).
    bool paintGL();

    // GLWidget::loadVolume (glwidget.cpp:668-710): camera back to (0, 0, -4) looking at the origin, table and scale by the
    // file name's ending (vv_dataset_preset; a name no rule matches keeps the current scale and table -- the reference reads an
    // uninitialised table pointer there -- Engine before any table was chosen), loadfrom_raw(path, header = true) ->
    // cudaLoadVolume, renderingDirty = true, hasCuttingPlane = false.
    void loadVolume(const char *path);
    int transferPreset() const { return m_tfPreset; }
    const float *scale() const { return m_scale; }

    // the setters of the reference's widget that matter to the hot path; each marks the frame dirty (glwidget.cpp:397-460, 743-788)
    void setCameraPosition(float x, float y, float z);                 // position; look re-aimed at the origin
    void orbitDrag(int dx, int dy);                                    // right-button drag, mouseMoveEvent :432-446
    void zoom(int delta);                                              // wheelEvent :607-620
    void setScale(float x, float y, float z);                          // scaleObject
    void setPhongShading(bool on);
    void setSliceVisualization(int vis /* 0 none, 1 plane, 2 cross section (glwidget.h sliceVisualization) */);
    void setCuttingPlane(const float point[3], const float normal[3], bool flipCrossSection);
    void setSliceCanonical(int orientation, float displace);           // :757-788 (ignored while the slice visualisation is off, as in the reference)
    void setSlicePro(const float offset[3], const float normal[3]);    // :743-755: the free-form slice view's plane (vv_cut_plane_from_euler makes it from the sliders, window.cpp:425-443)
    void setSlicePro(float dx, float dy, float dz, float theta, float phi, float psi);   // Window::renderSlice's PRO_SLICING branch + setSlicePro in one call
    bool hasCuttingPlane() const { return m_hasCuttingPlane; }
    void clearCuttingPlane();
    void setResolutionScale(int s);                                    // params.h:10 (3 in the reference)

    int width() const { return m_width; }
    int height() const { return m_height; }
    int renderWidth() const { return m_width / m_resolutionScale; }
    int renderHeight() const { return m_height / m_resolutionScale; }
    bool renderingDirty() const { return m_renderingDirty; }
    float lastRenderTime() const { return m_lastRenderTime; }          // seconds, as the overlay prints it (:312-315)
    unsigned long runCudaCalls() const { return m_runs; }
    const std::vector<unsigned char> &resultTexture() const { return m_resultTexture; }
    const std::vector<unsigned char> &frontFace() const { return m_fbo[0]; }
    const std::vector<unsigned char> &backFace() const { return m_fbo[1]; }
    void *volumeArray = nullptr;                                       // m_volumeArray (vestigial, kernel.cu:393)

private:
    int m_width = 0, m_height = 0, m_resolutionScale = 3;
    bool m_renderingDirty = true;
    float m_lastRenderTime = 0.f;
    unsigned long m_runs = 0;
    float m_pos[3], m_look[3], m_up[3], m_scale[3];
    float m_fovX = 45.f, m_fovY = 45.f;
    bool m_phong = false, m_hasCuttingPlane = false, m_flip = false;
    int m_sliceVis = 0;
    int m_tfPreset = VV_TF_ENGINE;
    float m_cutPoint[3], m_cutNormal[3];
    std::vector<unsigned char> m_fbo[2], m_resultBuffer, m_resultTexture;
};
