// kernel_hip.h -- host-side mirror of the reference's kernel.cuh operator surface, for a
// C++ host (glwidget.cpp / slicewidget.cpp style) that wants to keep its call sites.
//
// Same names, argument order and meaning as kernel.cuh:16-61; same error behaviour as the
// reference's checkCudaErrors (print + exit(EXIT_FAILURE), include/helper_cuda.h:763-777).
// Everything forwards to the C-ABI in include/volviz.h -- applications that want error
// codes, streams, sharding or f32 volumes call that directly.
//
// Differences that the missing CUDA/GL types force:
//   * cudaArray* / cudaArray** parameters (vestigial in the reference: ignored at
//     kernel.cu:393,457) are void* / void**;
//   * registerCudaResources takes the GL names when built with a GL-interop host; in this
//     headless build registerHostResources() supplies the same three resources as host
//     memory: FBO0/FBO1 RGBA8 images and the RGBA8 output pixel buffer.
#pragma once
#include <cstddef>
#include "../../include/volviz.h"
#include "volumegenerator_hip.h"

typedef unsigned char byte;                         // volumegenerator.h:21

struct float3 { float x, y, z; };                   // helper_math.h float3 (CUDA vector type)
inline float3 make_float3(float x, float y, float z) { float3 r = {x, y, z}; return r; }

// params.h:46
typedef enum { HORIZONTAL, SAGITTAL, CORONAL, N_CANONICAL_ORIENTATIONS, FREE_FORM } canonicalOrientation;

// params.h:56-74
struct SliceParameters {
    SliceParameters(float x, float y, float z) : dx(x), dy(y), dz(z), theta(0), phi(0), psi(0) {}
    SliceParameters(float x, float y, float z, float t, float ph, float ps) : dx(x), dy(y), dz(z), theta(t), phi(ph), psi(ps) {}
    float dx, dy, dz, theta, phi, psi;
};
struct BufferParameters {
    BufferParameters(size_t height_, size_t width_) : height(height_), width(width_) {}
    size_t height, width;
};

// cs123math/CS123Algebra.h:273-291: row-major float[16]
struct Matrix4x4 { float data[16]; };

extern "C" {
void initCuda();                                                                   // kernel.cuh:44
void registerCudaResources(unsigned input0, unsigned input1, unsigned output);     // kernel.cuh:45 (GL names)
void runCuda(int width, int height, struct slice_params slice, struct camera_params camera,
             struct shading_params shading, void *volumeArray);                    // kernel.cuh:46-51
void cudaLoadVolume(byte *texels, size_t size, Vector3 dims, float transferFunction[1024],
                    void **volumeArray);                                           // kernel.cuh:53-54
}

// headless stand-in for the three GL resources registerCudaResources() registers
void registerHostResources(const unsigned char *front_rgba, const unsigned char *back_rgba,
                           int fbo_width, int fbo_height, unsigned char *pixels_rgba);

void invoke_slice_kernel(float *buffer, BufferParameters bp, SliceParameters sp,
                         canonicalOrientation c, float3 scale);                    // kernel.cuh:59
void invoke_advanced_slice_kernel(float *buffer, BufferParameters bp, Matrix4x4 trans,
                                  float3 scale);                                   // kernel.cuh:61
// slicekernel.cuh:27 (legacy 4-argument form; never built by the reference's .pro)
void invoke_slice_kernel(float *buffer, BufferParameters bp, SliceParameters sp, canonicalOrientation c);

// SliceWidget::getTransformationMatrix (slicewidget.cpp:147-165)
Matrix4x4 getTransformationMatrix(SliceParameters sliceParameters);

// the process-wide context behind the mirror (the reference keeps file-static state too)
vv_context *volvizContext();
