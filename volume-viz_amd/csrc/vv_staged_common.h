// vv_staged_common.h -- pieces shared by the LDS-staged march kernels (block- and wave-private boxes).
#pragma once
#include "vv_device.h"

namespace vv {

struct Box {                                   // block-uniform
    int lox, loy, loz;                         // first voxel index of the box per volume axis (x 16-byte aligned)
    int nx, ny, nz;                            // extent in voxels per axis
    int pitch;                                 // bytes between box rows in LDS
    int slice_pitch;                           // bytes between box slices in LDS
};

__device__ __forceinline__ int wave_min(int v)
{
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ int wave_max(int v)
{
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
    return v;
}

// per-lane ray cursor: the reference's loop nest flattened into "next sample" steps
struct Cursor {
    float dist; int n, i, chunks;              // chunk start distance, samples in chunk, next index, chunk counter
    float px, py, pz;                          // position of sample i (kernel.cu:141 incremental sums)
    bool live, ert;
};

template <int VOXEL, bool TEX8>
__device__ __forceinline__ float lds_trilinear(const char *box, const Box &B, float wx, float wy, float wz,
                                               uint32_t lx, uint32_t ly, uint32_t lz)
{
    const uint32_t a00 = lz * (uint32_t)B.slice_pitch + ly * (uint32_t)B.pitch + lx * (VOXEL == VV_VOXEL_F32 ? 4u : 1u);
    const char *p00 = box + a00, *p10 = p00 + B.pitch, *p01 = p00 + B.slice_pitch, *p11 = p01 + B.pitch;
    float c000, c100, c010, c110, c001, c101, c011, c111;
    if (VOXEL == VV_VOXEL_F32) {
        c000 = ((const float *)p00)[0]; c100 = ((const float *)p00)[1];
        c010 = ((const float *)p10)[0]; c110 = ((const float *)p10)[1];
        c001 = ((const float *)p01)[0]; c101 = ((const float *)p01)[1];
        c011 = ((const float *)p11)[0]; c111 = ((const float *)p11)[1];
    } else {
        c000 = (float)((const uint8_t *)p00)[0]; c100 = (float)((const uint8_t *)p00)[1];
        c010 = (float)((const uint8_t *)p10)[0]; c110 = (float)((const uint8_t *)p10)[1];
        c001 = (float)((const uint8_t *)p01)[0]; c101 = (float)((const uint8_t *)p01)[1];
        c011 = (float)((const uint8_t *)p11)[0]; c111 = (float)((const uint8_t *)p11)[1];
    }
    float c00 = __builtin_fmaf(wx, c100 - c000, c000);
    float c10 = __builtin_fmaf(wx, c110 - c010, c010);
    float c01 = __builtin_fmaf(wx, c101 - c001, c001);
    float c11 = __builtin_fmaf(wx, c111 - c011, c011);
    float c0 = __builtin_fmaf(wy, c10 - c00, c00);
    float c1 = __builtin_fmaf(wy, c11 - c01, c01);
    return __builtin_fmaf(wz, c1 - c0, c0);
}

} // namespace vv
