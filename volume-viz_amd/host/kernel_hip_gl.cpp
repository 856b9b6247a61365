// kernel_hip_gl.cpp -- the GL-interop half of the kernel.cuh mirror (optional target: `make gl`).
//
// Replaces registerCudaResources / the map-bind-launch-unmap frame of runCuda (kernel.cu:375-386,394-413,452)
// for a host that renders the first pass with OpenGL like glwidget.cpp:198-228,361-390 does:
//   input0 / input1  GL_TEXTURE_2D RGBA8 colour attachments of the two first-pass FBOs (front / back positions)
//   output           GL pixel-buffer object, width * height * 4 bytes, glTexSubImage2D'ed afterwards (glwidget.cpp:294)
// Linked instead of the headless registerCudaResources() of kernel_hip.cpp (which only reports that GL is absent).
// It needs a current GL context at run time, which this repository's test machines do not have: the file is
// compiled and linked by the build, not executed by the tests.
#include "../../include/volviz.h"      // (not kernel_hip.h: its stand-in float3 collides with the HIP headers' own)

vv_context *volvizContext();             // kernel_hip.cpp: the process-wide context behind the mirror

#include <GL/gl.h>
#include <hip/hip_runtime_api.h>
#include <hip/hip_gl_interop.h>

#include <cstdio>
#include <cstdlib>

namespace {
hipGraphicsResource_t g_in0 = nullptr, g_in1 = nullptr, g_out = nullptr;     // kernel.cu:35-36,47
unsigned char *g_front = nullptr, *g_back = nullptr;                         // linear copies of the two textures
size_t g_img_cap = 0;

void chk(hipError_t e, const char *what)
{
    if (e == hipSuccess) return;
    fprintf(stderr, "HIP error at %s: code=%d \"%s\"\n", what, (int)e, hipGetErrorString(e));   // helper_cuda.h:763-777
    exit(EXIT_FAILURE);
}
void chkv(int rc, const char *what)
{
    if (rc == VV_OK) return;
    fprintf(stderr, "volviz error at %s: code=%d \"%s\"\n", what, rc, vv_last_error(volvizContext()));
    exit(EXIT_FAILURE);
}
}

// kernel.cu:375-386
extern "C" void registerCudaResourcesGL(GLuint input0, GLuint input1, GLuint output)
{
    (void)volvizContext();                                                   // initCuda() picks the device
    chk(hipGraphicsGLRegisterImage(&g_in0, input0, GL_TEXTURE_2D, hipGraphicsRegisterFlagsReadOnly), "register input0");
    chk(hipGraphicsGLRegisterImage(&g_in1, input1, GL_TEXTURE_2D, hipGraphicsRegisterFlagsReadOnly), "register input1");
    chk(hipGraphicsGLRegisterBuffer(&g_out, output, hipGraphicsRegisterFlagsWriteDiscard), "register output");
}

// kernel.cu:388-453: map the three resources, march, unmap (the unmap is the frame's fence, :452)
extern "C" void runCudaGL(int width, int height, struct slice_params slice, struct camera_params camera,
                          struct shading_params shading, void *)
{
    if (!g_in0 || !g_in1 || !g_out) { fprintf(stderr, "runCuda: no resources registered\n"); exit(EXIT_FAILURE); }
    hipGraphicsResource_t res[3] = {g_in0, g_in1, g_out};
    chk(hipGraphicsMapResources(3, res, 0), "map");
    hipArray_t a0 = nullptr, a1 = nullptr;
    chk(hipGraphicsSubResourceGetMappedArray(&a0, g_in0, 0, 0), "array 0");   // kernel.cu:396-403
    chk(hipGraphicsSubResourceGetMappedArray(&a1, g_in1, 0, 0), "array 1");
    HIP_ARRAY_DESCRIPTOR d0, d1;
    chk(hipArrayGetDescriptor(&d0, a0), "descriptor 0");
    chk(hipArrayGetDescriptor(&d1, a1), "descriptor 1");
    if (d0.Width != d1.Width || d0.Height != d1.Height) { fprintf(stderr, "runCuda: first-pass textures differ in size\n"); exit(EXIT_FAILURE); }
    const size_t fw = d0.Width, fh = d0.Height, ib = fw * fh * 4;
    if (g_img_cap < 2 * ib) {
        if (g_front) chk(hipFree(g_front), "free");
        chk(hipMalloc((void **)&g_front, 2 * ib), "image copies");
        g_back = g_front + ib; g_img_cap = 2 * ib;
    }
    g_back = g_front + ib;
    // the march samples the first-pass images by point look-ups (kernel.cu:317-318): a linear copy serves as well
    chk(hipMemcpy2DFromArray(g_front, fw * 4, a0, 0, 0, fw * 4, fh, hipMemcpyDeviceToDevice), "copy front");
    chk(hipMemcpy2DFromArray(g_back, fw * 4, a1, 0, 0, fw * 4, fh, hipMemcpyDeviceToDevice), "copy back");
    void *pixels = nullptr; size_t nbytes = 0;
    chk(hipGraphicsResourceGetMappedPointer(&pixels, &nbytes, g_out), "pixel buffer");   // kernel.cu:411-413
    if (nbytes < (size_t)width * height * 4) { fprintf(stderr, "runCuda: pixel buffer too small\n"); exit(EXIT_FAILURE); }
    vv_ray_source rs = {};
    rs.mode = VV_RAYS_IMAGES; rs.front = g_front; rs.back = g_back; rs.img_w = (int)fw; rs.img_h = (int)fh; rs.images_on_device = 1;
    chkv(vv_render(volvizContext(), width, height, &slice, &camera, &shading, &rs, nullptr, (uint8_t *)pixels, 1, nullptr), "runCuda");
    chk(hipGraphicsUnmapResources(3, res, 0), "unmap");                       // kernel.cu:452
}

extern "C" void unregisterCudaResourcesGL()
{
    if (g_in0) chk(hipGraphicsUnregisterResource(g_in0), "unregister"); g_in0 = nullptr;
    if (g_in1) chk(hipGraphicsUnregisterResource(g_in1), "unregister"); g_in1 = nullptr;
    if (g_out) chk(hipGraphicsUnregisterResource(g_out), "unregister"); g_out = nullptr;
    if (g_front) chk(hipFree(g_front), "free"); g_front = g_back = nullptr; g_img_cap = 0;
}
