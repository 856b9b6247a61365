// vv_frustum.h -- the footprint of a pixel tile's frustum in the volume's slices (shared by the slab sweep,
// vv_sweep.hip, and the prefetch wave of march_kernel, vv_raymarch.hip).
#pragma once
#include "vv_device.h"

namespace vv {
namespace sweepk {

constexpr float kMargin = 0.0625f;             // voxels added around the analytic footprint (float rounding is < 0.01)

// Tile frustum in voxel-float coordinates (vb = tex * n - 0.5): eye E and the extreme slopes of the
// four corner rays against the sweep coordinate.  Every ray of the tile runs inside the hull of the
// corner rays (directions are affine in the pixel coordinates, ray_endpoints()).
struct Frustum { float Ex, Er, Es, mx_lo, mx_hi, mr_lo, mr_hi; };

template <int MAJOR>
__device__ __host__ inline void axis_pick(const float v[3], float &x, float &r, float &s)
{
    x = v[0]; r = MAJOR == 2 ? v[1] : v[2]; s = MAJOR == 2 ? v[2] : v[1];
}

// Footprint of the frustum in slice s: voxel columns x0..x1 and rows r0..r1 (inclusive) that any in-volume
// sample interpolating with slice s can touch.  Those are the samples whose sweep coordinate zeta lies in
// [s - 1, s + 1), plus -- slice 1 only, kept for all -- the in-volume samples with zeta in [-0.5, 0), which clamp
// to slice 0 and read slice 1 with weight 0 (the value must still be finite).  The eye lies outside the slab
// range (plan_sweep), so (zeta - Es) keeps one sign and each bound is ONE corner slope times a linear function of
// s: bound(s) = a + m * s, four fused multiply-adds per slice.
struct Foot { int x0, x1, r0, r1; };
struct FootLin { float ax_lo, mx_lo, ax_hi, mx_hi, ar_lo, mr_lo, ar_hi, mr_hi; };
__device__ __host__ inline void foot_linear(const Frustum &F, bool ahead /* zeta - Es > 0 */, FootLin &L)
{
    const float c0 = -1.5f, c1 = 1.0f;                 // the slab is [s + c0, s + c1)
    auto pick = [&](float mlo, float mhi, float E, float &a_lo, float &m_lo, float &a_hi, float &m_hi) {
        // ahead: min over {mlo, mhi} x {z0, z1} is mlo * (mlo >= 0 ? z0 : z1), max is mhi * (mhi >= 0 ? z1 : z0);
        // behind (z < 0): min is mhi * (mhi >= 0 ? z0 : z1), max is mlo * (mlo >= 0 ? z1 : z0)
        m_lo = ahead ? mlo : mhi; m_hi = ahead ? mhi : mlo;
        const float cl = m_lo >= 0.f ? c0 : c1, ch = m_hi >= 0.f ? c1 : c0;
        a_lo = E - kMargin + m_lo * (cl - F.Es);
        a_hi = E + kMargin + m_hi * (ch - F.Es);
    };
    pick(F.mx_lo, F.mx_hi, F.Ex, L.ax_lo, L.mx_lo, L.ax_hi, L.mx_hi);
    pick(F.mr_lo, F.mr_hi, F.Er, L.ar_lo, L.mr_lo, L.ar_hi, L.mr_hi);
}
__device__ __forceinline__ Foot footprint(const FootLin &L, int s, int nx, int nr)
{
    const float fs = (float)s;
    const float xlo = __builtin_fmaf(L.mx_lo, fs, L.ax_lo), xhi = __builtin_fmaf(L.mx_hi, fs, L.ax_hi);
    const float rlo = __builtin_fmaf(L.mr_lo, fs, L.ar_lo), rhi = __builtin_fmaf(L.mr_hi, fs, L.ar_hi);
    Foot f;
    f.x0 = (int)fminf(fmaxf(floorf(xlo), 0.f), (float)(nx - 1));
    f.x1 = (int)fminf(fmaxf(floorf(xhi), 0.f), (float)(nx - 1)) + 1;
    f.r0 = (int)fminf(fmaxf(floorf(rlo), 0.f), (float)(nr - 1));
    f.r1 = (int)fminf(fmaxf(floorf(rhi), 0.f), (float)(nr - 1)) + 1;
    return f;
}


// The frustum of the pixel rectangle [x0, x1] x [y0, y1] (pixel centres, ray_endpoints()) against sweep axis MAJOR.
template <int MAJOR>
__device__ inline void tile_frustum(const FrameParams &P, const VolumeView &V, int x0, int y0, int x1, int y1, Frustum &F)
{
    float h[3], nn[3] = {(float)V.nx, (float)V.ny, (float)V.nz}, E[3];
    for (int a = 0; a < 3; ++a) { h[a] = 0.5f * P.inv_scale[a]; E[a] = (P.cam_pos[a] * h[a] + 0.5f) * nn[a] - 0.5f; }
    axis_pick<MAJOR>(E, F.Ex, F.Er, F.Es);
    F.mx_lo = F.mr_lo = INFINITY; F.mx_hi = F.mr_hi = -INFINITY;
    for (int c = 0; c < 4; ++c) {
        const int px = (c & 1) ? x1 : x0, py = (c & 2) ? y1 : y0;
        const float ndx = (2.0f * ((float)px + 0.5f)) / (float)P.W - 1.0f, ndy = (2.0f * ((float)py + 0.5f)) / (float)P.H - 1.0f;
        const float sx = ndx * P.tan_half_x, sy = ndy * P.tan_half_y;
        float D[3];
        for (int a = 0; a < 3; ++a) D[a] = ((P.side[a] * sx + P.up[a] * sy) + P.look[a]) * h[a] * nn[a];
        float dx, dr, ds;
        axis_pick<MAJOR>(D, dx, dr, ds);
        const float mx = dx / ds, mr = dr / ds;
        F.mx_lo = fminf(F.mx_lo, mx); F.mx_hi = fmaxf(F.mx_hi, mx);
        F.mr_lo = fminf(F.mr_lo, mr); F.mr_hi = fmaxf(F.mr_hi, mr);
    }
}

} // namespace sweepk
} // namespace vv
