// vv_sweep.hip -- block-wide slab sweep: the ray march with the volume streamed through LDS (gfx950).
//
// Replaces kernel<SLICE_NONE> + mainLoop (kernel.cu:203-367) for unshaded frames whose rays all cross the
// volume's x-y (or x-z) planes in the same direction.  Same arithmetic per sample as march_kernel
// (vv_raymarch.hip), so frames stay bit-identical; what changes is where the eight corners come from.
//
// march_kernel gathers them from HBM through the 32 KB L1 (4 wave-wide gathers per sample, 16 cycles of
// address pipeline each, stalled by every miss).  Here a block of up to 1024 threads owns a wide, short pixel
// tile (e.g. 96 x 8):
//   * nc = wx * wy consumer waves, 32 x 2 pixels each, keep one ray per lane in registers;
//   * nl loader waves walk the slices k = kmin .. kmax the tile's rays cross, in the order the rays
//     cross them, and copy each slice's footprint of the tile -- the bounding box of the tile's
//     frustum in that slice, a few hundred voxels wide and two dozen rows high -- HBM -> LDS with
//     `global_load_lds_dwordx4` (whole 128-byte cells, one image row per wave instruction, no registers)
//     into a ring of `ring` slice slots;
//   * a consumer lane takes its next sample as soon as the two slices it interpolates between have
//     landed (four 8-byte LDS reads), at its own pace: lanes are not in lock step, so a wave needs
//     no common slab of slices and stalls only when the loaders are behind;
//   * flags in LDS replace barriers: `loaded` (slices landed, written by the loaders in order) and
//     `progress[w]` (slices wave w will not read again); a slot is refilled when every wave has let go.
// Every voxel line of a tile's footprint is read once per tile; neighbouring tiles overlap by the
// footprint's rim (x: up to one 128-byte cell, y: two rows), which they mostly find in L2.
//
// Reference-mode early ray termination (kernel.cu:272-274: one sample per later 30-sample chunk) would
// keep the stream running for almost nothing, so a wave whose live rays have all terminated leaves the
// ring and takes those sparse samples by direct gathers like march_kernel.
#include "vv_device.h"
#include "vv_kernels.h"
#include "vv_frustum.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <algorithm>

namespace vv {
namespace sweepk {

constexpr int kInf = 0x3fffffff;
constexpr int kTfBytes = 4096;
constexpr int kCtlBytes = 2048;
constexpr int kRingOff = kTfBytes + kCtlBytes;
constexpr int kLdsMax = 160 * 1024;
constexpr int kMaxChunks = 60;                 // rows (= LDS-DMA instructions) per slice image: a group's must fit the 6-bit vmcnt
constexpr int kPage = 1024;                    // the ring is handed out in pages
constexpr int kPages = (kLdsMax - kRingOff) / kPage;
constexpr int kTab = 32;                       // slices the ring can hold at once (table entries)

typedef int __attribute__((ext_vector_type(4))) i4v;
typedef int __attribute__((ext_vector_type(2))) i2v;
struct Ctl {                                   // control block in LDS
    int landed[4];                             // per loader wave: its rows of the slices k < landed[w] have landed (unused: kInf)
    int kmin, kmax;                            // slice range of the tile (sweep order)
    int err;
    int alloc;                                 // index of the next group to be given its place in the ring
    int progress[16];                          // per consumer wave: slices k < progress[w] are released
    i2v tab[kTab];                            // per slice k (entry k % kTab): byte address of voxel (x 0, row 0) of its image, row pitch
    int box[kTab][4];                          // per slice: x0, x1, r0, r1 held (instrumented builds check against it)
    int owner[kPages + 2];                     // per ring page: the slice whose image occupies it
    int dbg[4][8];                             // debug builds: what each loader wave is doing
};
static_assert(sizeof(Ctl) <= kCtlBytes, "control block");

// minimum over the wave / over each row of 16 lanes by DPP (no LDS traffic)
template <int CTRL, int ROWMASK> __device__ __forceinline__ int dpp_min(int v)
{
    return min(v, __builtin_amdgcn_update_dpp(v, v, CTRL, ROWMASK, 0xf, false));
}
__device__ __forceinline__ int row16_min(int v)       // lane 15 of every row of 16 holds the row's minimum
{
    v = dpp_min<0x111, 0xf>(v); v = dpp_min<0x112, 0xf>(v); v = dpp_min<0x114, 0xf>(v); v = dpp_min<0x118, 0xf>(v);
    return v;
}
__device__ __forceinline__ int wave_min_fast(int v)   // uniform result
{
    v = row16_min(v);
    v = dpp_min<0x142, 0xa>(v);                        // row_bcast:15 into rows 1 and 3
    v = dpp_min<0x143, 0xc>(v);                        // row_bcast:31 into rows 2 and 3
    return __builtin_amdgcn_readlane(v, 63);
}

__device__ __forceinline__ bool any_(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }

// volatile vector loads / stores in the LDS address space (through a generic pointer hipcc falls back to flat_load
// with a full s_waitcnt: hundreds of cycles per access)
typedef int __attribute__((ext_vector_type(4))) i4v_;
typedef int __attribute__((ext_vector_type(2))) i2v_;
__device__ __forceinline__ i4v_ lds_load_i4(const void *p) { return *(const volatile __attribute__((address_space(3))) i4v_ *)p; }
__device__ __forceinline__ i2v_ lds_load_i2(const void *p) { return *(const volatile __attribute__((address_space(3))) i2v_ *)p; }
__device__ __forceinline__ void lds_store_i2(void *p, i2v_ v) { *(volatile __attribute__((address_space(3))) i2v_ *)p = v; }
__device__ __forceinline__ int lds_load_i(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_store_i(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// Four x-pairs (8 bytes at a 4-byte aligned LDS address each).  hipcc turns such a load into ds_read2_b32; the
// unaligned ds_read_b64 (experiment build) returns the right bytes but is an order of magnitude slower.
__device__ __forceinline__ void lds_pairs(const char *a, const char *b, const char *c, const char *d,
                                          float2u &va, float2u &vb, float2u &vc, float2u &vd)
{
#ifndef VV_SWEEP_B64ASM      // measured on MI355X: an unaligned ds_read_b64 costs ~36 cycles per wave instruction (tools/ubench/lds_pairs.hip)
    va = *(const float2u *)a; vb = *(const float2u *)b; vc = *(const float2u *)c; vd = *(const float2u *)d;
#else
    typedef float __attribute__((ext_vector_type(2))) f2;
    f2 ra, rb, rc, rd;
    const uint32_t aa = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char *)a, ab = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char *)b;
    const uint32_t ac = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char *)c, ad = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char *)d;
    asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %5\n\tds_read_b64 %2, %6\n\tds_read_b64 %3, %7\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(ra), "=&v"(rb), "=&v"(rc), "=&v"(rd) : "v"(aa), "v"(ab), "v"(ac), "v"(ad) : "memory");
    va = {ra.x, ra.y}; vb = {rb.x, rb.y}; vc = {rc.x, rc.y}; vd = {rd.x, rd.y};
#endif
}

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
// wait until at most n vector-memory operations of this wave are outstanding (n uniform, 0..kMaxChunks)
__device__ __forceinline__ void wait_vm_n(int n)
{
    n = n > 63 ? 63 : n;                               // the counter has 6 bits; waiting for fewer is only stricter
    switch (n) {
#define VV_W(i) case i: wait_vm<i>(); break;
#define VV_W8(b) VV_W(b) VV_W(b + 1) VV_W(b + 2) VV_W(b + 3) VV_W(b + 4) VV_W(b + 5) VV_W(b + 6) VV_W(b + 7)
    VV_W8(0) VV_W8(8) VV_W8(16) VV_W8(24) VV_W8(32) VV_W8(40) VV_W8(48) VV_W8(56)
#undef VV_W8
#undef VV_W
    default: wait_vm<0>(); break;
    }
}

// ---------------------------------------------------------------------------------------------
template <int MAJOR, bool TEX8, bool GRAY, bool INSTR>
__global__ __launch_bounds__(1024) void sweep_kernel(FrameParams P, VolumeView V,
                                                     const float4 *__restrict__ tf,
                                                     const float *__restrict__ rad,
                                                     uint32_t *__restrict__ pixels,
                                                     unsigned long long *__restrict__ counter,
                                                     uint32_t *__restrict__ bricks, SweepArgs S)
{
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    float *lds_tf = (float *)lds;
    Ctl *ctl = (Ctl *)(lds + kTfBytes);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long t_start = INSTR ? __builtin_readcyclecounter() : 0ull;
    const unsigned long long t_blk0 = S.trace ? __builtin_amdgcn_s_memrealtime() : 0ull;
    // tile of this block.  Raster order dealt to the XCDs by tile row (block L runs on XCD L % 8 under
    // round-robin dispatch: speed only), or the order the host planned.
    int trow, tcol;
    if (S.order) {
        if ((int)blockIdx.x >= S.n_order) return;
        const int t = S.order[blockIdx.x];
        trow = t / S.ntx; tcol = t % S.ntx;
    } else {
        // Centre-out: the tiles in the middle of the frame march the longest rays, the ones at its rim are short or
        // empty; dispatching the long ones first keeps the end of the launch from waiting for a few late, long tiles.
        const int L = blockIdx.x, xcd = L & 7, j = L >> 3;
        const int rr = (j / S.ntx) * 8 + xcd, rc = j % S.ntx;             // rank of the tile row / column
        if (rr >= S.nty) return;                                          // block-uniform, before any barrier
        const int cy = (S.nty - 1) >> 1, cx = (S.ntx - 1) >> 1;
        trow = (rr & 1) ? cy + ((rr + 1) >> 1) : cy - (rr >> 1);
        tcol = (rc & 1) ? cx + ((rc + 1) >> 1) : cx - (rc >> 1);
    }
    if (trow >= S.nty || trow < 0 || tcol < 0 || tcol >= S.ntx) return;   // block-uniform, before any barrier
    const int tile_w = S.wx * 32, tile_h = S.wy * 2;
    const int x0 = tcol * tile_w;
    const int y0 = S.y0 + (trow / S.rows_per_band) * S.band_stride_px + (trow % S.rows_per_band) * tile_h;

    for (int i = threadIdx.x; i < 256; i += blockDim.x) {
        float4 e = tf[i];
        lds_tf[i] = e.x; lds_tf[256 + i] = e.y; lds_tf[512 + i] = e.z; lds_tf[768 + i] = e.w;
    }
    if (threadIdx.x < 16) ctl->progress[threadIdx.x] = kInf;
    if (threadIdx.x == 16) { ctl->kmin = kInf; ctl->kmax = -1; ctl->err = 0; ctl->alloc = 0; }
    if (threadIdx.x >= 32 && threadIdx.x < 36) ctl->landed[threadIdx.x - 32] = (int)threadIdx.x - 32 < S.nl ? 0 : kInf;
    for (int i = threadIdx.x; i < kPages + 2; i += blockDim.x) ctl->owner[i] = -1;
    __syncthreads();

    const int nx = V.nx, nr = MAJOR == 2 ? V.ny : V.nz, ns = MAJOR == 2 ? V.nz : V.ny;
    // slice index s along the sweep axis -> position k in sweep order: s, or ns - s = (s ^ -1) + ns + 1
    const int kmul = S.sgn > 0 ? 1 : -1, kadd = S.sgn > 0 ? 0 : ns;
    const int kxor = S.sgn > 0 ? 0 : -1, kxadd = S.sgn > 0 ? 0 : ns + 1;

    if (wave >= S.nc) {
        // =====================================================================================
        // loader waves
        // =====================================================================================
        const int lw = wave - S.nc;
        // frustum of the tile from its four corner pixels (pixel centres, ray_endpoints())
        Frustum F;
        // the rectangle of the tile's pixels that exist: slopes are ratios of affine functions of the pixel, monotone only
        // where the denominator keeps its sign, which sweep_axis() checked for the frame's pixels and no further
        tile_frustum<MAJOR>(P, V, min(x0, P.W - 1), min(y0, P.H - 1), min(x0 + tile_w - 1, P.W - 1), min(y0 + tile_h - 1, P.H - 1), F);
        FootLin FL;
        foot_linear(F, S.sgn > 0, FL);
        // one LDS-DMA piece = one row of the slice's image: lane l copies bytes [16 l, 16 l + 16) of the row
        // (8 lanes per 128-byte cell; lanes beyond the row's cells are masked)
        const uint64_t Sr = MAJOR == 2 ? V.row_bytes : V.slice_bytes;       // bytes between rows of the image
        const uint64_t Ss = MAJOR == 2 ? V.slice_bytes : V.row_bytes;       // bytes between slices
        __syncthreads();
        const int kmin = __builtin_amdgcn_readfirstlane(lds_load_i(&ctl->kmin)), kmax = __builtin_amdgcn_readfirstlane(lds_load_i(&ctl->kmax));
        // ---- loader wave lw: every nl-th group of `group` slices.  The ring is handed out in 1 KiB pages, in slice
        // order, as a circular first-in-first-out buffer: a group takes ceil(slices * rows * pitch / 1 KiB) pages at
        // `head`, or at page 0 when they do not fit before the end.  Every loader wave derives the same positions from
        // the same footprints (it runs the footprints of the other waves' groups too: a few dozen instructions), so
        // the waves never talk to each other.  Wave w publishes landed[w] = the first slice it has not confirmed yet
        // (its oldest pending group, else the next group it will issue); everything before the minimum over the
        // waves has landed.  A page may be overwritten once the group that owns it is released by every consumer wave.
        // A group shares one footprint (the union over its slices), one allocation and one confirmation: per slice
        // the loaders' bookkeeping would cost more than the copies themselves. ----
        const int G = S.group;
        int gi = 0;                          // index of the next group in the deterministic walk
        int kn = kmin;                       // its first slice
        int head = 0, idle = 0;
        int pk = -1, pn = 0;                 // first slice and instruction count of this wave's pending group (-1: none)
        bool bail = false;
        unsigned long long t_issue = 0, t_land = 0, t_pub = 0, t_idle = 0, tt = 0;
        const unsigned long long t_pro = INSTR ? __builtin_readcyclecounter() - t_start : 0ull;
        // the first group that is mine
        auto group_shape = [&](int k0, int &ke, Foot &f) {
            ke = min(k0 + G - 1, kmax);
            const int sa = kmul * k0 + kadd, sb = kmul * ke + kadd;
            const Foot fa = footprint(FL, sa, nx, nr), fb = footprint(FL, sb, nx, nr);
            f.x0 = __builtin_amdgcn_readfirstlane(min(fa.x0, fb.x0)); f.x1 = __builtin_amdgcn_readfirstlane(max(fa.x1, fb.x1));      // uniform by construction:
            f.r0 = __builtin_amdgcn_readfirstlane(min(fa.r0, fb.r0)); f.r1 = __builtin_amdgcn_readfirstlane(max(fa.r1, fb.r1));      // scalar loops and branches
        };
        if (lane == 0) lds_store_i(&ctl->landed[lw], kmin);
        for (;;) {
            if (INSTR) tt = __builtin_readcyclecounter();
            if (kn > kmax && pk < 0) break;
            // ---- groups of the other waves: only their place in the ring ----
            if (kn <= kmax && (gi % S.nl) != lw) {
                int ke; Foot f;
                group_shape(kn, ke, f);
                const int ncell = min((f.x1 >> 5) - (f.x0 >> 5) + 1, S.pxc), nrows = min(f.r1 - f.r0 + 1, S.ry);
                const int np = ((ke - kn + 1) * nrows * ncell * 128 + kPage - 1) / kPage;
                head = (head + np <= kPages ? head : 0) + np;
                kn = ke + 1; ++gi;
                if (pk < 0 && lane == 0) lds_store_i(&ctl->landed[lw], min(kn, kmax + 1));     // nothing of mine before kn is missing
                continue;
            }
            // ---- my pending group first: it must be confirmed before my next one is issued (one group's instructions
            //      nearly fill the 6-bit vmcnt) ----
            if (pk >= 0) {
                wait_vm<0>();
                pk = -1;
                if (lane == 0) lds_store_i(&ctl->landed[lw], min(kn, kmax + 1));               // all my groups before kn have landed
                if (INSTR) t_land += __builtin_readcyclecounter() - tt;
                continue;
            }
            // ---- my next group: wait for room in the ring, then issue it ----
            {
                const int pr = max(__builtin_amdgcn_readlane(row16_min(lds_load_i(&ctl->progress[lane & 15])), 15), kmin);
                if (pr >= kInf) { bail = true; break; }                 // every consumer wave has left the ring
                int ke; Foot f;
                group_shape(kn, ke, f);
                const int ng = ke - kn + 1, c0 = f.x0 >> 5;
                int ncell = (f.x1 >> 5) - c0 + 1, nrows = f.r1 - f.r0 + 1;
                if (ncell > S.pxc || nrows > S.ry) { if (lane == 0) lds_store_i(&ctl->err, 2); ncell = min(ncell, S.pxc); nrows = min(nrows, S.ry); }
                const int pitch = ncell * 128, img_bytes = nrows * pitch;
                const int np = (ng * img_bytes + kPage - 1) / kPage;
                const int pos = head + np <= kPages ? head : 0;
                // Places are given out in group order (ctl->alloc passes from wave to wave): a later group that took its
                // pages first could sit on pages an earlier group needs while the consumers wait for that earlier group.
                bool ok = ke - pr < kTab && __builtin_amdgcn_readfirstlane(lds_load_i(&ctl->alloc)) == gi;
                if (ok) {
                    bool busy = false;
                    for (int b0 = 0; b0 < np; b0 += 64)
                        if (b0 + lane < np) busy = busy || !(lds_load_i(&ctl->owner[pos + b0 + lane]) < pr);
                    ok = !any_(busy);
                }
                if (!ok) {
#ifdef VV_SWEEP_DEBUG
                    if (lane == 0) { int *d = ctl->dbg[lw]; d[0] = kn; d[1] = ke; d[2] = pr; d[3] = pos; d[4] = np; d[5] = head; d[6] = gi; d[7] = idle; }
#endif
                    // no room: wait without taking issue slots from the consumers that have to make it
                    __builtin_amdgcn_s_setprio(0);
                    __builtin_amdgcn_s_sleep(4);
                    if (INSTR) t_idle += __builtin_readcyclecounter() - tt;
                    if (++idle > (1 << 22)) { if (lane == 0) lds_store_i(&ctl->err, 1); bail = true; break; }
                    continue;
                }
                idle = 0;
                __builtin_amdgcn_s_setprio(3);       // the copies are on every consumer's critical path: issue them ahead of the arithmetic
                for (int b0 = 0; b0 < np; b0 += 64)
                    if (b0 + lane < np) lds_store_i(&ctl->owner[pos + b0 + lane], ke);
                const int img = kRingOff + pos * kPage;
                if (lane < ng) {
                    lds_store_i2(&ctl->tab[(kn + lane) & (kTab - 1)], i2v{img + lane * img_bytes - f.r0 * pitch - c0 * 128, pitch});
                    if (INSTR) { int *bx = ctl->box[(kn + lane) & (kTab - 1)]; bx[0] = c0 * 32; bx[1] = (c0 + ncell) * 32 - 1; bx[2] = f.r0; bx[3] = f.r0 + nrows - 1; }
                }
                if (lane == 0) lds_store_i(&ctl->alloc, gi + 1);         // the next group may take its place (after the owner marks above: LDS is in order)
                const int sa = kmul * kn + kadd;
                const char *gp = (const char *)V.data + (int64_t)sa * (int64_t)Ss + (uint64_t)f.r0 * Sr + (uint64_t)c0 * 128u + (uint32_t)lane * 16u;
                const int64_t gstep = (int64_t)kmul * (int64_t)Ss - (int64_t)nrows * (int64_t)Sr;    // last row of a slice -> first row of the next
                int lb = img;
                if (lane < 8 * ncell) {              // one exec mask for the whole group
#pragma unroll 1
                    for (int j = 0; j < ng; ++j) {
#pragma unroll 1
                        for (int rr = 0; rr < nrows; ++rr) {
#ifndef VV_SWEEP_DUMMY_LOADER            // experiment build: the bookkeeping without the copies (rate of the consumers alone; pixels are wrong)
                            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gp,
                                                             (__attribute__((address_space(3))) void *)(lds + lb), 16, 0, 0);
#endif
                            gp += Sr; lb += pitch;
                        }
                        gp += gstep;
                    }
                }
                __builtin_amdgcn_s_setprio(0);
                pk = kn; pn = ng * nrows;
                head = pos + np; kn = ke + 1; ++gi;
                if (INSTR && lane == 0) atomicAdd(counter + 5, (unsigned long long)ng * nrows * ncell * 128ull);
                if (INSTR) t_issue += __builtin_readcyclecounter() - tt;
            }
        }
        wait_vm<0>();                     // nothing of this wave may land in LDS after it has gone
        // open the gate for good: a consumer that asked for more than kmax would otherwise wait forever (it cannot,
        // but a wrong pixel beats a hung GPU)
        if (lane == 0) lds_store_i(&ctl->landed[lw], kInf);
        if (INSTR && lane == 0) {
            atomicAdd(counter + 8, t_issue); atomicAdd(counter + 9, t_land); atomicAdd(counter + 10, t_pub); atomicAdd(counter + 11, t_idle);
            atomicAdd(counter + 12, t_pro);
        }
        (void)bail; (void)pn;
        return;
    }

    // =========================================================================================
    // consumer waves: 32 x 2 pixels each, wx x wy of them tile the block's tile_w x tile_h pixels
    // =========================================================================================
    const int x = x0 + (wave % S.wx) * 32 + (lane & 31), y = y0 + (wave / S.wx) * 2 + (lane >> 5);
    const int xmax = P.W >= 2 ? P.W - 2 : 0, ymax = P.H >= 2 ? P.H - 2 : 0;
    const bool in_frame = x <= xmax && y <= ymax && row_owned(P, y);

    float res_r = 0.f, res_g = 0.f, res_b = 0.f, res_a = 0.f;
    unsigned long long executed = 0, slots = 0, misses = 0;
    bool write_zero = false;
    Ray r;
    bool alive = false;
    if (in_frame) {
        f3 front, back;
        ray_endpoints(P, x, y, front, back);
        float length = vlen3(back.x - front.x, back.y - front.y, back.z - front.z);
        if (length < 0.001f) {
            write_zero = true;                                           // kernel.cu:334-338
        } else {
            float rd;
            if (P.W < 2 || P.H < 2) rd = vlen3(front.x - P.cam_pos[0], front.y - P.cam_pos[1], front.z - P.cam_pos[2]);
            else rd = rad[owner_slab(y, P.H, P.nby, P.conflict_y) * P.nbx + owner_slab(x, P.W, P.nbx, P.conflict_x)];
            setup_ray(P, front, back, rd, r);
            alive = !r.cut_return;
        }
    }
    if (!alive) { r.upper = -1.f; r.dist0 = 0.f; r.sstep = 1.f; r.origin = mk3(0, 0, 0); r.dir = r.origin; r.sdir = r.origin; }

    // slices this ray can touch, with a margin: positions at both ends of [dist0, upper]
    {
        int klo = kInf, khi = -1;
        if (alive && r.dist0 < r.upper) {
            const float os = MAJOR == 2 ? r.origin.z : r.origin.y, dsv = MAJOR == 2 ? r.dir.z : r.dir.y;
            const float isc = MAJOR == 2 ? P.inv_scale[2] : P.inv_scale[1];
            const float ta = __builtin_fmaf((os + dsv * r.dist0) - 0.5f, isc, 0.5f), tb = __builtin_fmaf((os + dsv * r.upper) - 0.5f, isc, 0.5f);
            const float za = ta * (float)ns - 0.5f, zb = tb * (float)ns - 0.5f;
            const int slo = (int)fminf(fmaxf(floorf(fminf(za, zb) - 1.5f), 0.f), (float)(ns - 1));
            const int shi = (int)fminf(fmaxf(floorf(fmaxf(za, zb) + 1.5f), 0.f), (float)(ns - 1)) + 1;
            klo = S.sgn > 0 ? slo : ns - shi; khi = S.sgn > 0 ? shi : ns - slo;
        }
        klo = wave_min_i(klo); khi = wave_max_i(khi);
        if (lane == 0 && khi >= 0) { atomicMin(&ctl->kmin, klo); atomicMax(&ctl->kmax, khi); lds_store_i(&ctl->progress[wave], 0); }
    }
    __syncthreads();
    const int kmin = __builtin_amdgcn_readfirstlane(lds_load_i(&ctl->kmin));

    // ---------------------------------------------------------------------------------------------------------------
    // Per-wave skewed lock step (round 3; round 2's consumer paced every LANE by itself and spent 386 VALU instructions
    // per step on that).  As in march_skew_kernel (vv_raymarch.hip) lane L takes its sample s = tau - o_L at wave step
    // tau, with o_L chosen so that all lanes of the wave sit within about one sample spacing of a common position along
    // the sweep axis; every lane still executes exactly the reference's operations for its ray.  The wave's slice window
    // of a trip -- [floor(min_L kc_L), floor(max_L kc_L) + 1] in sweep positions -- is then a few slices wide and is
    // tracked with scalars: kc_L(tau) = A_L + tau * D_L per lane (positions are affine in the sample number), the wave
    // extremes are re-anchored by two wave reductions every kAnchor steps and run on the extreme slopes in between.
    // One scalar test per trip against `landed`, one scalar release through progress[wave]; no per-lane waiting.
    // A wave whose window would not fit the ring (lanes entering through different faces at the cube's silhouette) or
    // whose live rays have all terminated early takes its samples by direct gathers instead (`tail`).
    // ---------------------------------------------------------------------------------------------------------------
    constexpr int U = 2;                      // samples per trip (LDS latency is short: depth buys nothing, window costs ring)
    constexpr int kAnchor = 16;
    constexpr float kWinMargin = 0.125f;      // slices; the affine model is exact to ~1e-3
    const float fns = (float)ns;
    float kA = 0.f, kD = 1.f;                 // kc_L(tau) = kA + tau * kD
    int o = 0;
    float Dmin = 1.f, Dmax = 1.f;
    {
        const float isc = MAJOR == 2 ? P.inv_scale[2] : P.inv_scale[1];
        const float p1 = MAJOR == 2 ? r.origin.z + r.sdir.z : r.origin.y + r.sdir.y;              // first sample (dist0 = 0: no cutting plane here)
        const float z1 = __builtin_fmaf(p1 - 0.5f, isc, 0.5f) * fns - 0.5f;                          // its slice coordinate
        const float dz = (MAJOR == 2 ? r.sdir.z : r.sdir.y) * isc * fns;
        const float kc1 = S.sgn > 0 ? z1 : fns - z1;
        kD = fabsf(dz);
        const bool ok = alive && kD > 1e-6f && kc1 == kc1;
        int m = __float_as_int(ok ? fmaxf(kc1 + 1024.f, 0.f) : INFINITY);                            // (+1024: ordered as integers also below 0)
        m = wave_min_fast(m);
        const float cref = __int_as_float(m) - 1024.f;
        if (ok) {
            const float q = rintf((kc1 - cref) / kD);
            o = q >= 0.f ? (q < 30.f ? (int)q : 30) : 0;
        }
        kA = kc1 - (float)(o + 1) * kD;
        // (positive floats order like their bit patterns: minimum / maximum by the integer reductions)
        Dmin = __int_as_float(wave_min_fast(ok ? __float_as_int(kD) : 0x7f800000));
        Dmax = __int_as_float(-wave_min_fast(ok ? -__float_as_int(kD) : 0));
        if (!(Dmin <= Dmax)) { Dmin = 1.f; Dmax = 1.f; }                      // no lane of this wave marches
    }

    float dist = r.dist0;
    bool ert = false, stop = false, active = alive, lastc = false;
    int i = 31 - o;                           // next sample of the lane's chunk; 31 = open the next chunk now
    int n = 0, chunks = 0;
    float px = 0.f, py = 0.f, pz = 0.f;
    int tail = 0;                             // wave-uniform: the wave takes its samples by direct gathers (it has left the ring)
    int pw = 0;                               // published progress of this wave
    int tau = 1;                              // wave step of the trip's first slot
    int tau0 = -1000000;                      // step of the last re-anchoring
    float lo0 = 0.f, hi0 = 0.f;               // wave extremes of kc at tau0
    const int kmaxT = __builtin_amdgcn_readfirstlane(lds_load_i(&ctl->kmax));
    const int tmax = P.max_chunks * 30 + 30 + U;
    int stall_run = 0, n_iter = 0, n_stall = 0;
    const unsigned long long t_loop0 = S.trace ? __builtin_amdgcn_s_memrealtime() : 0ull;

    for (int t = 0; t < tmax; t += U) {
        const bool pending = active && !(lastc && i > n) && !(P.ert_true && ert);
        if (!__any(pending)) break;
        // every lane idle until its next chunk boundary: jump there (march_skew_kernel)
        if (!__any(i <= n && !stop)) {
            int d = pending ? 31 - i : 64;
            d = wave_min_fast(d);
            i += d; tau += d;
        }
        // all live rays terminated early: the remaining samples are one per 30-sample chunk -- not worth the stream
        if (!tail && !__any(pending && !ert)) {
            tail = 1;
            if (lane == 0) lds_store_i(&ctl->progress[wave], kInf);
        }
        if (!tail) {
            // ---- the wave's slice window for this trip ----
            if (tau - tau0 >= kAnchor) {
                const float kc = fminf(fmaxf(kA + (float)tau * kD, (float)kmin), (float)kmaxT);     // pending lanes; >= 0
                const int mn = wave_min_fast(pending ? __float_as_int(kc) : 0x7f800000);
                const int mx = wave_min_fast(pending ? -__float_as_int(kc) : 0);
                lo0 = __int_as_float(mn); hi0 = __int_as_float(-mx); tau0 = tau;
                if (!(hi0 - lo0 + (float)kAnchor * (Dmax - Dmin) + (float)U * Dmax + 2.f * kWinMargin + 3.f <= (float)S.wmax)) {
                    // the window does not fit the ring: this wave marches on direct gathers from here on
                    tail = 1;
                    if (lane == 0) lds_store_i(&ctl->progress[wave], kInf);
                }
            }
        }
        tail = __builtin_amdgcn_readfirstlane(tail);
        if (!tail) {
            const float dt0 = (float)(tau - tau0);
            const float lo_f = lo0 + dt0 * Dmin - kWinMargin, hi_f = hi0 + (dt0 + (float)(U - 1)) * Dmax + kWinMargin;
            const int need_lo = __builtin_amdgcn_readfirstlane(max((int)floorf(lo_f), kmin));
            const int need_hi = __builtin_amdgcn_readfirstlane(min((int)floorf(hi_f) + 1, kmaxT));
            if (need_lo > pw) { pw = need_lo; if (lane == 0) lds_store_i(&ctl->progress[wave], pw); }
            for (;;) {
                const i4v l4 = lds_load_i4(ctl->landed);
                const int kl = __builtin_amdgcn_readfirstlane(min(min(l4.x, l4.y), min(l4.z, l4.w)));
                if (kl > need_hi) break;
                if (INSTR) ++n_stall;
                __builtin_amdgcn_s_sleep(1);
                if (++stall_run > (1 << 21)) {               // watchdog: a wrong pixel beats a hung GPU
                    if (lane == 0) { lds_store_i(&ctl->err, 3); lds_store_i(&ctl->progress[wave], kInf); }
                    tail = 1;
                    break;
                }
            }
            stall_run = 0;
            asm volatile("" ::: "memory");        // the flag is read before any slice data of this step
        }
        ++n_iter;
        if (INSTR) slots += (unsigned long long)U * 64ull;

        uint32_t idx[U];
        int iu[U];
        bool lv[U];
        float ttx[U], tty[U], ttz[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            // ---- chunk boundary of this lane: close the old chunk (:277), open the next (:248-249) ----
            if (__any(i == 31)) {
                if (i == 31) {
#pragma clang fp contract(off)
                    if (chunks > 0) { dist += r.sstep * kChunkSteps; if (P.ert_true && ert) r.upper = -1.f; }
                    i = 1; n = 0;
                    if (active) {
                        if (chunks >= P.max_chunks) active = false;
                        else {
                            n = chunk_count(dist, r.upper, r.sstep);
                            lastc = n < 30;
                            if (ert) n = min(n, 1);                          // DESIGN.md pin 4 (tables here have opacities in [0, 1])
                            px = r.origin.x + r.dir.x * dist; py = r.origin.y + r.dir.y * dist; pz = r.origin.z + r.dir.z * dist;
                            ++chunks;
                            if (n == 0) active = false;
                        }
                    }
                }
            }
            px += r.sdir.x; py += r.sdir.y; pz += r.sdir.z;                  // :141
            const float tx = __builtin_fmaf(px - 0.5f, P.inv_scale[0], 0.5f);
            const float ty = __builtin_fmaf(py - 0.5f, P.inv_scale[1], 0.5f);
            const float tz = __builtin_fmaf(pz - 0.5f, P.inv_scale[2], 0.5f);
            ttx[u] = tx; tty[u] = ty; ttz[u] = tz;
            iu[u] = i; lv[u] = i <= n;
            ++i;
            if (!tail) {
                uint32_t ix, iy, iz;
                const float wx = axis_coord<TEX8>(tx, (float)V.nx, (float)(V.nx - 1), ix);
                const float wy = axis_coord<TEX8>(ty, (float)V.ny, (float)(V.ny - 1), iy);
                const float wz = axis_coord<TEX8>(tz, (float)V.nz, (float)(V.nz - 1), iz);
                const bool inb = bounds_check(tx, ty, tz);
                const int is = (int)(MAJOR == 2 ? iz : iy), ir = (int)(MAJOR == 2 ? iy : iz);
                const int k0 = (is ^ kxor) + kxadd;              // position of slice `is`; slice is + 1 sits at k0 + kmul
                const i2v T0 = lds_load_i2(&ctl->tab[k0 & (kTab - 1)]), T1 = lds_load_i2(&ctl->tab[(k0 + kmul) & (kTab - 1)]);
                const int x4 = (int)(ix << 2);
                const char *p0 = lds + ((int)__umul24((uint32_t)ir, (uint32_t)T0.y) + (T0.x + x4));
                const char *p1 = lds + ((int)__umul24((uint32_t)ir, (uint32_t)T1.y) + (T1.x + x4));
                float2u c00, c10, c01, c11;
                lds_pairs(p0, p0 + T0.y, p1, p1 + T1.y, c00, c10, c01, c11);
                // MAJOR == 2: rows are y, slices z.  MAJOR == 1: rows are z, slices y -- the lerp order stays x, y, z
                const float2u a_ = c00, b_ = MAJOR == 2 ? c10 : c01, c_ = MAJOR == 2 ? c01 : c10, d_ = c11;
                const float e00 = __builtin_fmaf(wx, a_.y - a_.x, a_.x);
                const float e10 = __builtin_fmaf(wx, b_.y - b_.x, b_.x);
                const float e01 = __builtin_fmaf(wx, c_.y - c_.x, c_.x);
                const float e11 = __builtin_fmaf(wx, d_.y - d_.x, d_.x);
                const float f0 = __builtin_fmaf(wy, e10 - e00, e00);
                const float f1 = __builtin_fmaf(wy, e11 - e01, e01);
                const float L = __builtin_fmaf(wz, f1 - f0, f0);
                uint32_t id = min((uint32_t)(L * 255.0f), 255u);
                idx[u] = inb ? id : 0u;
                if (INSTR && lv[u] && inb) {
                    const int *bx0 = ctl->box[k0 & (kTab - 1)], *bx1 = ctl->box[(k0 + kmul) & (kTab - 1)];
                    const bool okb = (int)ix >= bx0[0] && (int)ix + 1 <= bx0[1] && ir >= bx0[2] && ir + 1 <= bx0[3] &&
                                     (int)ix >= bx1[0] && (int)ix + 1 <= bx1[1] && ir >= bx1[2] && ir + 1 <= bx1[3];
                    // (a sample masked by `stop` / early termination below may lie ahead of the window: only live ones count)
                    if (!okb && !stop && !(P.ert_true && ert)) ++misses;
                }
            } else {
                idx[u] = V.big ? sample_index<VV_VOXEL_F32, TEX8, true>(V, tx, ty, tz) : sample_index<VV_VOXEL_F32, TEX8, false>(V, tx, ty, tz);
            }
        }
        tau += U;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (iu[u] == 1) stop = false;                    // the inner loop's break flag is per chunk (:272-274)
            const bool live = lv[u] && !stop && !(P.ert_true && ert);
            float cr, cg, cb, ca;
            ca = lds_tf[768 + idx[u]];
            cr = lds_tf[idx[u]];
            if (GRAY) { cg = cb = cr; }
            else { cg = lds_tf[256 + idx[u]]; cb = lds_tf[512 + idx[u]]; }
            if (INSTR && live) {
                executed++;
                if (bricks && bounds_check(ttx[u], tty[u], ttz[u])) mark_bricks(bricks, V, ttx[u], tty[u], ttz[u]);
            }
            {
                // :268-270 + blend :107-118, predicated (the table is finite)
#pragma clang fp contract(off)
                const float bf = (live && ca > kEps) ? ca * (1.f - res_a) : 0.f;
                res_r = res_r + cr * bf;
                if (!GRAY) { res_g = res_g + cg * bf; res_b = res_b + cb * bf; }
                res_a = res_a + bf;
            }
            const bool hit = live && res_a > P.ert_thr;                      // :272-274
            stop = stop || hit;
            ert = ert || hit;
        }
    }
    if (!tail && lane == 0) lds_store_i(&ctl->progress[wave], kInf);
    if (S.trace && wave == 0 && lane == 0) {
        unsigned long long *tr = S.trace + 8ull * blockIdx.x;
        tr[0] = t_blk0; tr[1] = t_loop0; tr[2] = __builtin_amdgcn_s_memrealtime();
        tr[3] = ((unsigned long long)__builtin_amdgcn_s_getreg(63492) << 32) | (unsigned)__builtin_amdgcn_s_getreg(63508);
        tr[4] = ((unsigned long long)trow << 32) | (unsigned)tcol; tr[5] = ((unsigned long long)(unsigned)kmaxT << 32) | (unsigned)kmin; tr[6] = ((unsigned long long)(unsigned)n_stall << 32) | (unsigned)n_iter; tr[7] = 1;
    }

    if (in_frame) {
        if (GRAY) { res_g = res_r; res_b = res_r; }
        pixels[(size_t)y * P.W + x] = write_zero ? 0u : pack_rgba(res_r, res_g, res_b, res_a);
    }
    if (INSTR) {
        for (int o = 32; o > 0; o >>= 1) { executed += __shfl_down(executed, o); misses += __shfl_down(misses, o); }
        if (lane == 0 && executed) atomicAdd(counter, executed);
        if (lane == 0 && slots) atomicAdd(counter + 1, slots);
        if (lane == 0 && misses) atomicAdd(counter + 4, misses);
        if (lane == 0 && n_stall) atomicAdd(counter + 6, (unsigned long long)n_stall);
        if (lane == 0 && tail) atomicAdd(counter + 13, 1ull);                                     // waves that finished on direct gathers
        if (lane == 0 && wave == 0) { const int e = lds_load_i(&ctl->err); if (e) atomicAdd(counter + 7, 1ull << (16 * (e - 1))); }   // four 16-bit counts: codes 1..4
    }
}

template <int MAJOR, bool TEX8, bool GRAY, bool INSTR>
static void launch_one(const MarchArgs &a, hipStream_t s)
{
    const SweepArgs &S = a.sweep;
    const unsigned nblocks = S.order ? (unsigned)S.n_order : (unsigned)(((S.nty + 7) / 8) * 8 * S.ntx);
    auto kern = sweep_kernel<MAJOR, TEX8, GRAY, INSTR>;
    static bool attr_set = false;
    if (!attr_set) { (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsMax); attr_set = true; }
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3((unsigned)((S.nc + S.nl) * 64)), (size_t)S.lds_bytes, s,
                       a.P, a.V, a.tf, a.rad, a.pixels, a.counter, a.bricks, S);
}
template <int MAJOR>
static void launch_major(const MarchArgs &a, hipStream_t s)
{
    const bool gray = a.gray;
    if (a.tex8) {
        if (gray) { if (a.instr) launch_one<MAJOR, true, true, true>(a, s); else launch_one<MAJOR, true, true, false>(a, s); }
        else      { if (a.instr) launch_one<MAJOR, true, false, true>(a, s); else launch_one<MAJOR, true, false, false>(a, s); }
    } else {
        if (gray) { if (a.instr) launch_one<MAJOR, false, true, true>(a, s); else launch_one<MAJOR, false, true, false>(a, s); }
        else      { if (a.instr) launch_one<MAJOR, false, false, true>(a, s); else launch_one<MAJOR, false, false, false>(a, s); }
    }
}

} // namespace sweepk

void launch_raymarch_sweep(const MarchArgs &a, hipStream_t s)
{
    if (a.sweep.major == 2) sweepk::launch_major<2>(a, s); else sweepk::launch_major<1>(a, s);
}

bool sweep_axis(const FrameParams &P, const VolumeView &V, int &major, int &sgn, const char **why)
{
    const char *dummy; if (!why) why = &dummy;
    major = 0; sgn = 0;
    if (P.ray_mode != VV_RAYS_ANALYTIC || P.quantize8) { *why = "rays from images / quantised"; return false; }
    if (P.W < 2 || P.H < 2) { *why = "degenerate frame"; return false; }
    const double n[3] = {(double)V.nx, (double)V.ny, (double)V.nz};
    double h[3], E[3];
    bool outside = false;
    for (int a = 0; a < 3; ++a) {
        h[a] = 0.5 * (double)P.inv_scale[a];
        E[a] = ((double)P.cam_pos[a] * h[a] + 0.5) * n[a] - 0.5;
        if (fabs((double)P.cam_pos[a]) > (double)P.scale[a] * 1.0001) outside = true;
    }
    if (!outside) { *why = "eye inside the cube"; return false; }         // front = (0,0,0) rays (kernel.cu:317-321 quirk)
    auto dirD = [&](double px, double py, double D[3]) {
        const double sx = ((2.0 * (px + 0.5)) / P.W - 1.0) * P.tan_half_x, sy = ((2.0 * (py + 0.5)) / P.H - 1.0) * P.tan_half_y;
        for (int a = 0; a < 3; ++a) D[a] = (P.side[a] * sx + P.up[a] * sy + P.look[a]) * h[a] * n[a];
    };
    // y or z, the one every ray of the frame crosses most steeply, all in the same direction
    double best_q = 0.0;
    for (int ax = 1; ax <= 2; ++ax) {
        double lo = INFINITY, hi = -INFINITY, q = INFINITY;
        for (int c = 0; c < 4; ++c) {
            double D[3]; dirD((c & 1) ? P.W - 1 : 0, (c & 2) ? P.H - 1 : 0, D);
            const double len = sqrt(D[0] * D[0] + D[1] * D[1] + D[2] * D[2]);
            lo = std::min(lo, D[ax]); hi = std::max(hi, D[ax]); q = std::min(q, fabs(D[ax]) / len);
        }
        if (lo * hi <= 0.0) continue;                                     // rays cross these planes both ways
        // the eye must sit before the first slice (every in-volume point has (zeta - Es) of one sign)
        if (lo > 0.0 ? !(E[ax] < -1.5) : !(E[ax] > n[ax] + 1.5)) continue;
        if (q > best_q) { best_q = q; major = ax; sgn = lo > 0.0 ? 1 : -1; }
    }
    if (!major || best_q < 0.35) { major = 0; *why = "no sweep axis: rays cross the x-y and x-z planes both ways or too flatly, or the eye is beside the slices"; return false; }
    return true;
}

// ---------------------------------------------------------------------------------------------
// Host-side plan: does the frame qualify, along which axis, and how large must the LDS image of a
// slice be.  Mirrors the device's footprint() in double precision with slack, so the device never
// finds a footprint larger than the slot (it would flag err = 2 in instrumented runs).
// ---------------------------------------------------------------------------------------------
void plan_sweep(MarchArgs &A, int y_first, int n_rows_px, int own_bands)
{
    using namespace sweepk;
    SweepArgs &S = A.sweep;
    S.enabled = 0;
    const SweepArgs req = S;                   // requested values (developer knobs, read by vv_api.cpp at init / volume load; -1 = default)
    const bool verbose = req.verbose != 0;
#define VV_NO(why) do { if (verbose) fprintf(stderr, "sweep: not used (%s)\n", why); return; } while (0)
    const FrameParams &P = A.P;
    const VolumeView &V = A.V;
    if (A.phong || A.V_type != VV_VOXEL_F32 || P.slice_type != SLICE_NONE) VV_NO("shaded, u8 or cutting plane");
    if (!P.alpha_unit) VV_NO("table opacities outside [0, 1]");       // the one-sample tail after early termination assumes monotone opacity
    if (P.ray_mode != VV_RAYS_ANALYTIC || P.quantize8) VV_NO("rays from images / quantised");
    if ((V.row_bytes & 15u) || (V.slice_bytes & 15u) || ((uintptr_t)V.data & 15u)) VV_NO("rows not 16-byte aligned");
    if (!(P.step[0] == P.step[1] && P.step[1] == P.step[2])) VV_NO("anisotropic step");      // samples of a chunk must stay on the ray's line
    if (P.W < 2 || P.H < 2 || n_rows_px < 1) VV_NO("degenerate frame");
    const double n[3] = {(double)V.nx, (double)V.ny, (double)V.nz};
    double h[3], E[3];
    for (int a = 0; a < 3; ++a) { h[a] = 0.5 * (double)P.inv_scale[a]; E[a] = ((double)P.cam_pos[a] * h[a] + 0.5) * n[a] - 0.5; }
    auto dirD = [&](double px, double py, double D[3]) {
        const double sx = ((2.0 * (px + 0.5)) / P.W - 1.0) * P.tan_half_x, sy = ((2.0 * (py + 0.5)) / P.H - 1.0) * P.tan_half_y;
        for (int a = 0; a < 3; ++a) D[a] = (P.side[a] * sx + P.up[a] * sy + P.look[a]) * h[a] * n[a];
    };
    int best = 0, best_sgn = 0;
    { const char *why = nullptr; if (!sweep_axis(P, V, best, best_sgn, &why)) VV_NO(why); }
    // sample spacing along the sweep axis, in slices: beyond ~3 whole slices would be streamed for nothing
    {
        const double dz = (double)P.step[best] * (double)P.inv_scale[best] * n[best];
        if (!(dz <= 3.0)) VV_NO("samples more than 3 slices apart");
    }
    const int xa = 0, ra = best == 2 ? 1 : 2, sa = best;
    const int nr = (int)n[ra], ns = (int)n[sa];
    S.major = best; S.sgn = best_sgn;
    // tile shape: wx x wy waves of 32 x 2 pixels.  Wide and short, so that a slice's image has few, long rows
    // (one LDS-DMA instruction per row) and the rim the neighbours re-read is small.
    S.nl = 3; S.wx = 3; S.wy = 4;
    if (req.nl >= 1 && req.nl <= 4) S.nl = req.nl;
    if (req.wx >= 1 && req.wx <= 8) S.wx = req.wx;
    if (req.wy >= 1 && req.wy <= 14) S.wy = req.wy;
    S.group = 3;                             // slices per allocation / confirmation unit of the loaders
    if (req.group >= 1 && req.group <= 8) S.group = req.group;
    if (S.wx * S.wy + S.nl > 16) VV_NO("too many waves");
    const bool forced = req.wx >= 1 || req.wy >= 1;
    for (;;) {
        S.nc = S.wx * S.wy;
        const int tw = 32 * S.wx, th = 2 * S.wy;
        S.ntx = (P.W + tw - 1) / tw;
        if (P.count > 1) {
            S.rows_per_band = (P.band * kSlab + th - 1) / th; S.band_stride_px = P.count * P.band * kSlab;
            S.nty = own_bands * S.rows_per_band;
        } else { S.rows_per_band = 1 << 28; S.band_stride_px = 0; S.nty = (n_rows_px + th - 1) / th; }
        S.y0 = y_first;
        // largest footprint over the tiles that can meet the volume, at both ends of the sweep
        double ext_x = 0.0, ext_r = 0.0;
        for (int t = 0; t < S.nty; ++t) {
            const int py0 = S.y0 + (t / S.rows_per_band) * S.band_stride_px + (t % S.rows_per_band) * th;
            for (int tc = 0; tc < S.ntx; ++tc) {
                double mxl = INFINITY, mxh = -INFINITY, mrl = INFINITY, mrh = -INFINITY;
                for (int c = 0; c < 4; ++c) {
                    double D[3]; dirD(std::min(tc * tw + ((c & 1) ? tw - 1 : 0), P.W - 1), std::min(py0 + ((c & 2) ? th - 1 : 0), P.H - 1), D);   // as the kernel: pixels of the frame only
                    const double mx = D[xa] / D[sa], mr = D[ra] / D[sa];
                    mxl = std::min(mxl, mx); mxh = std::max(mxh, mx); mrl = std::min(mrl, mr); mrh = std::max(mrh, mr);
                }
                double hx_lo = INFINITY, hx_hi = -INFINITY, hr_lo = INFINITY, hr_hi = -INFINITY, ex = 0.0, er = 0.0;
                for (int end = 0; end < 2; ++end) {
                    const double sl = end ? ns : 0;
                    const double z0 = sl - 1.5 - (S.sgn < 0 ? S.group - 1 : 0) - E[sa], z1 = sl + 1.0 + (S.sgn > 0 ? S.group - 1 : 0) - E[sa];   // a group's slab: it extends along the sweep
                    const double xs[4] = {mxl * z0, mxl * z1, mxh * z0, mxh * z1}, rs[4] = {mrl * z0, mrl * z1, mrh * z0, mrh * z1};
                    const double xl = *std::min_element(xs, xs + 4), xh = *std::max_element(xs, xs + 4);
                    const double rl = *std::min_element(rs, rs + 4), rh = *std::max_element(rs, rs + 4);
                    ex = std::max(ex, xh - xl); er = std::max(er, rh - rl);
                    hx_lo = std::min(hx_lo, E[xa] + xl); hx_hi = std::max(hx_hi, E[xa] + xh);
                    hr_lo = std::min(hr_lo, E[ra] + rl); hr_hi = std::max(hr_hi, E[ra] + rh);
                }
                if (verbose && getenv("VV_SWEEP_VERBOSE2")) fprintf(stderr, "  tile (%d,%d): x [%.2f, %.2f] rows [%.2f, %.2f] slopes x [%.3f, %.3f] r [%.3f, %.3f] ex %.2f er %.2f\n", tc, t, hx_lo, hx_hi, hr_lo, hr_hi, mxl, mxh, mrl, mrh, ex, er);
                if (hx_hi < -2.0 || hx_lo > n[xa] + 1.0 || hr_hi < -2.0 || hr_lo > n[ra] + 1.0) continue;   // the tile's frustum misses the volume
                ext_x = std::max(ext_x, ex); ext_r = std::max(ext_r, er);
            }
        }
        const double slack = 2.0 * kMargin + 0.01;
        if (verbose) fprintf(stderr, "sweep: tile %dx%d waves: largest extent %.2f voxels in x, %.2f rows\n", S.wx, S.wy, ext_x, ext_r);
        S.pxc = (int)floor((ext_x + slack + 2.0) / 32.0) + 2;
        S.ry = (int)floor(ext_r + slack) + 3;
        S.pxc = std::min(S.pxc, (V.nx + 31) / 32 + 1);
        S.ry = std::min(S.ry, nr + 1);
        S.slot_bytes = S.pxc * 128 * S.ry;                               // the largest image of a slice
        S.ring = (kPages * kPage) / S.slot_bytes;                        // slices of that size the ring holds (it holds more of the smaller ones)
        if (S.pxc <= 8 && S.ry <= kMaxChunks && S.ring >= 12) break;      // a consumer wave's window (~8-10 slices) + two groups must fit
        // footprint too large for the LDS (sparse pixels): smaller tiles, else no sweep
        if (verbose) fprintf(stderr, "sweep: tile %dx%d waves needs pxc %d ry %d ring %d\n", S.wx, S.wy, S.pxc, S.ry, S.ring);
        if (forced) VV_NO("forced tile shape does not fit");
        if (S.wy > 2) S.wy -= 1; else if (S.wx > 1) { S.wx -= 1; S.wy = 4; } else VV_NO("footprint does not fit the LDS");
    }
    (void)nr;
    // A sample may interpolate between the last slice of one group and the first of the next, and the loaders can only
    // refill behind both: three groups of the largest size must fit or the ring can lock up.  One row of slack per group
    // for the rounding to pages.
    {
        const int gmax = (kPages * kPage) / (3 * (S.slot_bytes + kPage));
        if (gmax < 1) VV_NO("three groups do not fit the ring");
        S.group = std::min(S.group, gmax);
        if (S.group * S.ry > 62) S.group = std::max(1, 62 / S.ry);             // a group's copies must fit the 6-bit vmcnt
        // Widest slice window a consumer wave may hold (in slices of the largest size): the ring must also hold the group being
        // landed, and pages come free only group-wise (up to group - 1 released slices stay resident)
        while (S.group > 1 && S.ring - 2 * S.group < 9) --S.group;
        S.wmax = S.ring - 2 * S.group;
        if (S.wmax < 6) VV_NO("the ring is too small for a wave's slice window");
    }
    S.lds_bytes = kRingOff + kPages * kPage;
    S.order = nullptr; S.n_order = 0; S.trace = nullptr;
    S.enabled = 1;
    if (verbose) fprintf(stderr, "sweep: axis %d sgn %d, tile %dx%d px (%d+%d waves), image <= %d cells x %d rows, group %d, %d x %d tiles\n", S.major, S.sgn, 32 * S.wx, 2 * S.wy, S.nc, S.nl, S.pxc, S.ry, S.group, S.ntx, S.nty);
#undef VV_NO
}

} // namespace vv
