// vv_raymarch_wstaged.hip -- wave-private LDS brick cache march (no Phong) for gfx950.
//
// For views that are NOT aligned with the volume's memory axis.  There the gather kernel
// (march_kernel) touches a different 128-byte line with almost every lane of every load
// (C3, view rotated by theta=60, phi=36 degrees: 5.6 ms against 1.6 ms along z), and every
// line is touched again a step or two later by a neighbouring lane -- too late for the 32 KB L1.
//
// Here every wavefront owns an 8x8 pixel tile and a private box in LDS (no block barriers, the
// four waves of a block never wait for each other).  The wave advances through the volume in
// slabs of slices perpendicular to its rays' major axis.  Per slab it
//   1. reduces (wave shuffles) the first slice its rays need and the minor-axis extents of
//      their runs through the slab, sized to the LDS box;
//   2. pulls that axis-aligned box HBM -> LDS with `global_load_lds_dwordx4` (LDS-DMA: 64 lanes
//      x 16 bytes land contiguously, rows of the box are contiguous in the linear volume, so
//      the copy is coalesced whatever the view direction);
//   3. composites every sample whose 2x2x2 footprint is in the box from LDS, with exactly the
//      arithmetic of march_kernel; a sample outside the box falls back to a global fetch.
// When the major axis is x the box rows lie along the rays, which is the best case: a 16 KB box
// serves ~10 samples of each of the 64 rays.
#include "vv_device.h"
#include "vv_kernels.h"
#include "vv_staged_common.h"

namespace vv {

template <int WBOX> struct WaveBoxCfg { static constexpr int bytes = WBOX; };

template <int SLICE, int VOXEL, bool TEX8, bool GRAY, bool INSTR, int WBOX>
__global__ __launch_bounds__(256) void march_wstaged_kernel(FrameParams P, VolumeView V,
                                                            const float4 *__restrict__ tf,
                                                            const float *__restrict__ rad,
                                                            uint32_t *__restrict__ pixels,
                                                            unsigned long long *__restrict__ counter,
                                                            uint32_t *__restrict__ bricks, StripMap M)
{
    __shared__ __attribute__((aligned(16))) char box_all[4 * WBOX];
    __shared__ float lds_tf[1024];

    constexpr uint32_t VSZ = VOXEL == VV_VOXEL_F32 ? 4u : 1u;
    constexpr int PIECE = 16 / (int)VSZ;                               // voxels per 16-byte piece

    const int ntx = (P.W + 31) >> 5;
    const int strip = blockIdx.x / ntx, tile_x = blockIdx.x % ntx;
    {
        float4 e = tf[threadIdx.x];
        lds_tf[threadIdx.x] = e.x; lds_tf[256 + threadIdx.x] = e.y; lds_tf[512 + threadIdx.x] = e.z; lds_tf[768 + threadIdx.x] = e.w;
    }
    __syncthreads();                                                   // the only block barrier

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char *box = box_all + wave * WBOX;
    const int x = (tile_x << 5) + (wave << 3) + (lane & 7);
    const int y = M.y0 + (strip / M.strips_per_band) * M.band_stride_px + (strip % M.strips_per_band) * 8 + (lane >> 3);
    const int xmax = P.W >= 2 ? P.W - 2 : 0, ymax = P.H >= 2 ? P.H - 2 : 0;
    const bool in_frame = x <= xmax && y <= ymax && row_owned(P, y);

    float res_r = 0.f, res_g = 0.f, res_b = 0.f, res_a = 0.f;
    unsigned long long executed = 0;
    bool write_zero = false;
    Ray r;
    bool alive = false;
    if (in_frame) {
        f3 front, back;
        ray_endpoints(P, x, y, front, back);
        float length = vlen3(back.x - front.x, back.y - front.y, back.z - front.z);
        if (length < 0.001f) {
            write_zero = true;                                         // kernel.cu:334-338
        } else {
            float rd;
            if (P.W < 2 || P.H < 2) rd = vlen3(front.x - P.cam_pos[0], front.y - P.cam_pos[1], front.z - P.cam_pos[2]);
            else rd = rad[owner_slab(y, P.H, P.nby, P.conflict_y) * P.nbx + owner_slab(x, P.W, P.nbx, P.conflict_x)];
            setup_ray(P, front, back, rd, r);
            alive = !r.cut_return;
        }
    }
    if (!alive) { r.upper = -1.f; r.dist0 = 0.f; r.sstep = 1.f; r.origin = mk3(0, 0, 0); r.dir = r.origin; r.sdir = r.origin; }

    const f3 sp = mk3(P.slice_point[0], P.slice_point[1], P.slice_point[2]);
    const f3 sn = mk3(P.slice_normal[0], P.slice_normal[1], P.slice_normal[2]);
    const float fnx = (float)V.nx, fny = (float)V.ny, fnz = (float)V.nz;

    // ---- cursor on the first sample (chunk 0, i = 1) ----
    Cursor c;
    c.dist = r.dist0; c.chunks = 0; c.ert = false; c.i = 1;
    c.n = chunk_count(c.dist, r.upper, r.sstep);
    c.live = c.n > 0;
    {
#pragma clang fp contract(off)
        c.px = r.origin.x + r.dir.x * c.dist; c.py = r.origin.y + r.dir.y * c.dist; c.pz = r.origin.z + r.dir.z * c.dist;
    }
    c.px += r.sdir.x; c.py += r.sdir.y; c.pz += r.sdir.z;

    // voxel-space increment per sample (prediction only; never used for a sample's value)
    const float dqx = r.sdir.x * P.inv_scale[0] * fnx, dqy = r.sdir.y * P.inv_scale[1] * fny, dqz = r.sdir.z * P.inv_scale[2] * fnz;

    // ---- wave-uniform major axis and direction: taken from the first live lane ----
    const unsigned long long live0 = __ballot(c.live);
    if (live0 != 0) {
        const int first = __ffsll((long long)live0) - 1;
        const float fx = __shfl(dqx, first), fy = __shfl(dqy, first), fz = __shfl(dqz, first);
        const int m = (fabsf(fx) >= fabsf(fy) && fabsf(fx) >= fabsf(fz)) ? 0 : (fabsf(fy) >= fabsf(fz) ? 1 : 2);
        const int sgn = (m == 0 ? fx : (m == 1 ? fy : fz)) < 0.f ? -1 : 1;
        const int nm = m == 0 ? V.nx : (m == 1 ? V.ny : V.nz);
        const int nn0 = m == 0 ? V.ny : V.nx, nn1 = m == 2 ? V.ny : V.nz;             // sizes of the two minor axes
        const float dm = m == 0 ? dqx : (m == 1 ? dqy : dqz);
        const float vfwd = fmaxf(fabsf(dm), 1e-6f);
        const float du = m == 0 ? dqy : dqx, dv = m == 2 ? dqy : dqz;

        int dpred = 12;                                                // slab thickness the next stage aims for
        const int max_stages = 2 * nm + P.max_chunks + 16;
        for (int stage = 0; stage < max_stages; ++stage) {
            if (!__any(c.live)) break;
            float tx = __builtin_fmaf(c.px - 0.5f, P.inv_scale[0], 0.5f);
            float ty = __builtin_fmaf(c.py - 0.5f, P.inv_scale[1], 0.5f);
            float tz = __builtin_fmaf(c.pz - 0.5f, P.inv_scale[2], 0.5f);
            bool inb = c.live && bounds_check(tx, ty, tz);
            const float qx = __builtin_fmaf(tx, fnx, -0.5f), qy = __builtin_fmaf(ty, fny, -0.5f), qz = __builtin_fmaf(tz, fnz, -0.5f);
            const float qm = m == 0 ? qx : (m == 1 ? qy : qz);
            const int basem = (int)floorf(fminf(fmaxf(qm, 0.f), (float)(nm - 1)));

            // ---- 1. slab: starts at the first base slice any in-volume ray of the wave needs ----
            const int kkey = __builtin_amdgcn_readfirstlane(wave_min(inb && !c.ert ? (sgn > 0 ? basem : -basem) : 0x7fffffff));
            int bs_lo, bs_hi;                                          // base slices this stage serves
            const bool have_slab = kkey != 0x7fffffff;
            Box b;
            b.lox = b.loy = b.loz = 0; b.nx = b.ny = b.nz = 0; b.pitch = 16; b.slice_pitch = 16;
            if (kkey == 0x7fffffff) {
                // nobody waits for data yet (leading / trailing out-of-volume samples, or only rays
                // that early-terminated and take one sample per chunk): no box this stage
                bs_lo = 0; bs_hi = -1;
            } else {
                const int k0 = sgn > 0 ? kkey : -kkey;
                if (sgn > 0) { bs_lo = k0; bs_hi = min(k0 + dpred - 1, nm - 1); }
                else         { bs_hi = k0; bs_lo = max(k0 - dpred + 1, 0); }
                // ---- 2. minor-axis extents of every ray's run through the slab ----
                int lo0 = 0x7fffffff, hi0 = -0x7fffffff, lo1 = 0x7fffffff, hi1 = -0x7fffffff;
                if (c.live && !c.ert) {
                    const float F = sgn > 0 ? qm : -qm;
                    const float E = sgn > 0 ? (float)(bs_hi + 1) : -(float)bs_lo;
                    const float rem = fmaxf((r.upper - ((float)c.i * r.sstep + c.dist)) / r.sstep + 2.0f, 0.f);
                    const float jn = fminf(fmaxf(ceilf((E - F) / vfwd), 0.f), fminf(rem, 96.f));
                    if (jn > 0.f) {
                        const float u0 = m == 0 ? qy : qx, v0 = m == 2 ? qy : qz;
                        const float u1 = u0 + jn * du, v1 = v0 + jn * dv;
                        lo0 = (int)floorf(fminf(u0, u1)) - 1; hi0 = (int)floorf(fmaxf(u0, u1)) + 2;
                        lo1 = (int)floorf(fminf(v0, v1)) - 1; hi1 = (int)floorf(fmaxf(v0, v1)) + 2;
                    }
                }
                lo0 = __builtin_amdgcn_readfirstlane(wave_min(lo0)); hi0 = __builtin_amdgcn_readfirstlane(wave_max(hi0));
                lo1 = __builtin_amdgcn_readfirstlane(wave_min(lo1)); hi1 = __builtin_amdgcn_readfirstlane(wave_max(hi1));
                // box along the march axis (M) and the two minor axes (A, B), mapped onto x, y, z
                const int loM = bs_lo, hiM = min(bs_hi + 1, nm);
                const int loA = max(lo0, 0), hiA = min(hi0, nn0), loB = max(lo1, 0), hiB = min(hi1, nn1);
                const bool empty = lo0 == 0x7fffffff || hiA < loA || hiB < loB;
                int lox = m == 0 ? loM : loA, hix = m == 0 ? hiM : hiA;
                int loy = m == 0 ? loA : (m == 1 ? loM : loB), hiy = m == 0 ? hiA : (m == 1 ? hiM : hiB);
                int loz = m == 2 ? loM : loB, hiz = m == 2 ? hiM : hiB;
                lox &= ~(PIECE - 1);                                   // 16-byte aligned rows
                int nxp = (hix - lox + PIECE) / PIECE;                 // 16-byte pieces per row (covers hix)
                int ny_ = hiy - loy + 1, nz_ = hiz - loz + 1;
                if (!empty && nxp * 16 * ny_ * nz_ > WBOX) {           // shed slices of the march axis
                    if (m == 0) {
                        const int fit = WBOX / (ny_ * nz_ * 16);       // pieces per row that fit
                        if (fit * PIECE < 2) nxp = 0;
                        else {
                            nxp = fit;
                            if (sgn < 0) { lox = (hix + 1 - nxp * PIECE + PIECE - 1) & ~(PIECE - 1); if (lox < 0) lox = 0; }
                            if (sgn > 0) bs_hi = min(bs_hi, lox + nxp * PIECE - 2); else bs_lo = max(bs_lo, lox);
                        }
                    } else if (m == 1) {
                        const int fit = WBOX / (nxp * 16 * nz_);
                        if (fit < 2) nxp = 0;
                        else if (sgn > 0) { hiy = loy + fit - 1; bs_hi = hiy - 1; } else { loy = hiy + 1 - fit; bs_lo = loy; }
                        ny_ = hiy - loy + 1;
                    } else {
                        const int fit = WBOX / (nxp * 16 * ny_);
                        if (fit < 2) nxp = 0;
                        else if (sgn > 0) { hiz = loz + fit - 1; bs_hi = hiz - 1; } else { loz = hiz + 1 - fit; bs_lo = loz; }
                        nz_ = hiz - loz + 1;
                    }
                }
                if (empty) nxp = 0;
                if (bs_lo > bs_hi) { if (sgn > 0) bs_hi = bs_lo; else bs_lo = bs_hi; }   // always serve >= 1 slice
                b.lox = lox; b.loy = loy; b.loz = loz;
                b.nx = nxp * PIECE; b.ny = nxp ? ny_ : 0; b.nz = nxp ? nz_ : 0;
                b.pitch = nxp * 16; b.slice_pitch = nxp * 16 * ny_;
                const int got = bs_hi - bs_lo + 1;
                dpred = got >= dpred ? min(dpred + 4, 64) : max(got, 2);

                // ---- 3. LDS-DMA: piece q of the box (row-major, 16 bytes) lands at box + 16 q ----
                const int total = nxp * b.ny * b.nz;
                const float inv_nxp = 1.0f / (float)max(nxp, 1), inv_ny = 1.0f / (float)max(b.ny, 1);
                const char *g0 = (const char *)V.data + (size_t)b.loz * V.slice_bytes + (size_t)b.loy * V.row_bytes + (size_t)b.lox * VSZ;
                asm volatile("" ::: "memory");                         // earlier LDS reads stay before the refill
                for (int q0 = 0; q0 < total; q0 += 64) {
                    const int q = q0 + lane;
                    const int row = (int)(((float)q + 0.5f) * inv_nxp);          // q / nxp  (exact for q < 2^16)
                    const int col = q - row * nxp;
                    const int zz = (int)(((float)row + 0.5f) * inv_ny);          // row / ny
                    const int yy = row - zz * b.ny;
                    const char *src = g0 + (size_t)zz * V.slice_bytes + (size_t)yy * V.row_bytes + (size_t)col * 16;
                    if (q < total)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                         (__attribute__((address_space(3))) void *)(box + q0 * 16), 16, 0, 0);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the box has landed (this wave's DMA only)
                if (INSTR && lane == 0) { atomicAdd(counter + 1, 1ull); atomicAdd(counter + 3, (unsigned long long)total * 16ull); }
            }

            // ---- 4. composite every sample this slab serves ----
            const uint32_t bx = (uint32_t)b.lox, by_ = (uint32_t)b.loy, bz = (uint32_t)b.loz;
            const uint32_t mx = b.nx > 1 ? (uint32_t)(b.nx - 1) : 0u, my = b.ny > 1 ? (uint32_t)(b.ny - 1) : 0u,
                           mz = b.nz > 1 ? (uint32_t)(b.nz - 1) : 0u;
            for (int guard = 0; guard < 4096; ++guard) {
                tx = __builtin_fmaf(c.px - 0.5f, P.inv_scale[0], 0.5f);               // kernel.cu:136, DESIGN.md pin 3
                ty = __builtin_fmaf(c.py - 0.5f, P.inv_scale[1], 0.5f);
                tz = __builtin_fmaf(c.pz - 0.5f, P.inv_scale[2], 0.5f);
                inb = bounds_check(tx, ty, tz);
                uint32_t ix, iy, iz;
                float wx = axis_coord<TEX8>(tx, fnx, (float)(V.nx - 1), ix);
                float wy = axis_coord<TEX8>(ty, fny, (float)(V.ny - 1), iy);
                float wz = axis_coord<TEX8>(tz, fnz, (float)(V.nz - 1), iz);
                const int im = (int)(m == 0 ? ix : (m == 1 ? iy : iz));
                // a lane takes its next sample now if it needs no data or its base slice is not
                // beyond the slab; early-terminated rays (one sample per chunk) never wait
                const bool mine = c.live && (!inb || c.ert || (have_slab && (sgn > 0 ? im <= bs_hi : im >= bs_lo)));
                if (!__any(mine)) break;
                if (INSTR && lane == 0) atomicAdd(counter + 4, 1ull);
                if (mine) {
                    uint32_t idx = 0;
                    if (inb) {
                        const uint32_t lx = ix - bx, ly = iy - by_, lz = iz - bz;
                        float L;
                        if (lx < mx && ly < my && lz < mz) L = lds_trilinear<VOXEL, TEX8>(box, b, wx, wy, wz, lx, ly, lz);
                        else { L = tex3d_raw<VOXEL, TEX8>(V, tx, ty, tz); if (INSTR) atomicAdd(counter + 2, 1ull); }   // not in the box
                        float sv = (VOXEL == VV_VOXEL_F32) ? L * 255.0f : L;
                        idx = min((uint32_t)sv, 255u);
                    }
                    float cr, cg, cb, ca;
                    ca = lds_tf[768 + idx];
                    cr = lds_tf[idx];
                    if (GRAY) { cg = cb = cr; }
                    else { cg = lds_tf[256 + idx]; cb = lds_tf[512 + idx]; }
                    if (SLICE == SLICE_PLANE) {                                        // kernel.cu:193-198
#pragma clang fp contract(off)
                        float vd = (float)c.i * r.sstep + c.dist;                      // :254
                        float vx = r.origin.x + r.dir.x * vd, vy = r.origin.y + r.dir.y * vd, vz = r.origin.z + r.dir.z * vd;
                        float d = fabsf(sn.x * (vx - sp.x) + sn.y * (vy - sp.y) + sn.z * (vz - sp.z));
                        if (d < .01f) cr = fmaxf(0.f, fminf(cr + (.01f - d) * 100.f, 1.f));
                    }
                    if (INSTR) {
                        executed++;
                        if (bricks && inb) mark_bricks(bricks, V, tx, ty, tz);
                    }
                    if (ca > kEps) {                                                   // :268-270, blend :107-118
#pragma clang fp contract(off)
                        float bf = ca * (1.f - res_a);
                        res_r = res_r + cr * bf;
                        if (!GRAY) { res_g = res_g + cg * bf; res_b = res_b + cb * bf; }
                        res_a = res_a + bf;
                    }
                    bool end_chunk = c.i >= c.n;
                    if (res_a > P.ert_thr) {                                           // :272-274
                        c.ert = true; end_chunk = true;
                        if (P.ert_true) r.upper = -1.f;
                    }
                    if (!end_chunk) {
                        c.i++;
                        c.px += r.sdir.x; c.py += r.sdir.y; c.pz += r.sdir.z;          // :141
                    } else {
#pragma clang fp contract(off)
                        c.dist += r.sstep * kChunkSteps;                               // :277
                        c.chunks++;
                        c.n = chunk_count(c.dist, r.upper, r.sstep);
                        if (c.ert) c.n = min(c.n, 1);
                        c.live = c.n > 0 && c.chunks < P.max_chunks;
                        c.i = 1;
                        c.px = r.origin.x + r.dir.x * c.dist; c.py = r.origin.y + r.dir.y * c.dist; c.pz = r.origin.z + r.dir.z * c.dist;
                        c.px = c.px + r.sdir.x; c.py = c.py + r.sdir.y; c.pz = c.pz + r.sdir.z;
                    }
                }
            }
        }
    }

    if (in_frame) {
        if (GRAY) { res_g = res_r; res_b = res_r; }
        pixels[(size_t)y * P.W + x] = write_zero ? 0u : pack_rgba(res_r, res_g, res_b, res_a);
    }
    if (INSTR) {
        for (int o = 32; o > 0; o >>= 1) executed += __shfl_down(executed, o);
        if (lane == 0 && executed) atomicAdd(counter, executed);
    }
}

// ---------------------------------------------------------------------------
constexpr int kWaveBoxBytes = 16 * 1024;

template <int SLICE, int VOXEL, bool TEX8, bool GRAY, bool INSTR>
static void launch_ws(const MarchArgs &a, hipStream_t s)
{
    const int ntx = (a.P.W + 31) / 32;
    dim3 grid((unsigned)(a.strips.n_strips * ntx));
    hipLaunchKernelGGL((march_wstaged_kernel<SLICE, VOXEL, TEX8, GRAY, INSTR, kWaveBoxBytes>), grid, dim3(256), 0, s,
                       a.P, a.V, a.tf, a.rad, a.pixels, a.counter, a.bricks, a.strips);
}
template <int SLICE, int VOXEL, bool TEX8>
static void wdispatch3(const MarchArgs &a, hipStream_t s)
{
    const bool gray = a.gray && SLICE != SLICE_PLANE;
    if (gray) { if (a.instr) launch_ws<SLICE, VOXEL, TEX8, true, true>(a, s); else launch_ws<SLICE, VOXEL, TEX8, true, false>(a, s); }
    else      { if (a.instr) launch_ws<SLICE, VOXEL, TEX8, false, true>(a, s); else launch_ws<SLICE, VOXEL, TEX8, false, false>(a, s); }
}
template <int SLICE>
static void wdispatch2(const MarchArgs &a, hipStream_t s)
{
    if (a.V_type == VV_VOXEL_F32) { if (a.tex8) wdispatch3<SLICE, VV_VOXEL_F32, true>(a, s); else wdispatch3<SLICE, VV_VOXEL_F32, false>(a, s); }
    else                          { if (a.tex8) wdispatch3<SLICE, VV_VOXEL_U8,  true>(a, s); else wdispatch3<SLICE, VV_VOXEL_U8,  false>(a, s); }
}

void launch_raymarch_wstaged(const MarchArgs &a, hipStream_t s)
{
    switch (a.P.slice_type) {
    case SLICE_PLANE:     wdispatch2<SLICE_PLANE>(a, s); break;
    case SLICE_PLANE_CUT: wdispatch2<SLICE_PLANE_CUT>(a, s); break;
    default:              wdispatch2<SLICE_NONE>(a, s); break;
    }
}

} // namespace vv
