// vv_sweep.hip -- block-wide slab sweep: the ray march with the volume streamed through LDS (gfx950).
//
// Replaces kernel<SLICE_NONE> + mainLoop (kernel.cu:203-367) for unshaded frames whose rays all cross the
// volume's x-y (or x-z) planes in the same direction.  Same arithmetic per sample as march_kernel
// (vv_raymarch.hip), so frames stay bit-identical; what changes is where the eight corners come from.
//
// march_kernel gathers them from HBM through the 32 KB L1 (4 wave-wide gathers per sample, 16 cycles of
// address pipeline each, stalled by every miss) and reads 1.48 x the algorithmic bytes because tiles that share
// lines do not meet in the L2 in time (profiles/r03_traffic_split.txt).  Here a block of NW waves owns a wide,
// short pixel tile (e.g. 64 x 8: NW = 8 waves of 32 x 2 pixels, one ray per lane in registers) and walks the
// slices k = kmin .. kmax its rays cross, in the order they cross them:
//   * every lane is offset in sample index (block-wide "skewed lock step", as march_skew_kernel does per wave) so
//     that at block step tau ALL rays of the tile sit within about one sample spacing of a common front along the
//     sweep axis; the slices a trip of U steps needs are then a window of ~8-10 slices known from scalars
//     (two block-wide extremes re-anchored every kAnchorTrips trips, run forward on the extreme slopes);
//   * ALL waves copy.  The image of slice k in LDS is a fixed S.ry rows x S.pxw voxels in ring slot k mod S.ring, its origin a
//     closed form of k (fixed-point lower bounds of the tile's footprint, SALU arithmetic, the same in every wave): no
//     allocation, no ownership table, no per-slice footprint evaluation.  An image is dealt to the waves in 16-byte units
//     (unit u -> LDS byte 16 u; wave w copies units [64 w, 64 w + 64) (+ 64 NW ...)): one `global_load_lds_dwordx4` with a
//     scalar base and a per-lane offset that is computed once per kernel.  Each trip every wave issues its units of the
//     slices the block will need up to kAhead trips later, waits for ITS OWN units of the slices this trip reads (a counted
//     s_waitcnt vmcnt: copies retire in order) and meets the others at the one barrier of the trip;
//   * corners come from LDS: one 8-byte read of the table (slice -> byte address of voxel (0, 0) of its image; the pitch is
//     the same for every slice), four ds_read2_b32 x-pairs, the lerps in the oracle's order.
// No flags, no polling, no loader / consumer roles (round 2's kernel had all three and lost to them: its flag
// protocol alone cost 1.2 ms per C3 frame), no page ring (round 3's: ~250 VALU per slice placed, in every wave).  This is
// the fourth design (round 4: profiles/r04_sweep_v4.txt); like the three before it, it gives the oracle's frames and loses to
// march_kernel: C3 in 2.2-3.9 ms against 1.0.
// Reference-mode early ray termination (kernel.cu:272-274: one sample per later 30-sample chunk) would keep the
// stream running for almost nothing, so once every live ray of the block has terminated -- or if a window ever
// does not fit the ring -- the block stops copying and each wave finishes on direct gathers like march_kernel.
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
constexpr int kCtlBytes = 1024;
constexpr int kRingOff = kTfBytes + kCtlBytes;
constexpr int kLdsMax = 160 * 1024;
constexpr int kTab = 32;                       // table entries: the ring holds fewer slices than that (S.ring <= kTab - 2)
constexpr int kCopyMax = 6;                    // LDS-DMA instructions a wave issues per slice at most (plan_sweep: units <= kCopyMax * 64 * waves)
constexpr int kFxBits = 14;                    // fixed point of the window origins: slope error < ns / 2^14 voxels (plan_sweep: ns <= 4096)
constexpr float kFxPad = 0.3f;                 // ... covered by this pad below the analytic bound

typedef int __attribute__((ext_vector_type(4))) i4v;
typedef int __attribute__((ext_vector_type(2))) i2v;
struct Ctl {                                   // control block in LDS
    int kmin, kmax;                            // slice range of the tile (sweep order)
    int cref;                                  // block reference position (offset float bits): the front at step 1
    int dmin, dmax;                            // extreme slopes of the block's rays, slices per step (float bits, -bits)
    int err;
    int anc[2][2];                             // double-buffered anchors: {min, -max} of the pending lanes' positions (offset float bits)
    int still[3];                              // trip T, word T % 3: set by a wave that still needs the stream (no static LDS: the ring may take all 160 KB)
    int pad_[1];
    int tab[kTab + 1];                         // per slice k (entry k % kTab; entry kTab repeats entry 0 so that a pair is one 8-byte read): byte address of voxel (x 0, row 0) of its image
    int own[kTab];                             // instrumented builds: the slice whose image entry k % kTab describes
    int box[kTab][4];                          // ... and x0, x1, r0, r1 it holds
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

// One LDS-DMA instruction: lane l copies 16 bytes from its global address to LDS byte address lds_addr + 16 l (lds_addr wave-uniform).
// Written as inline assembly ON PURPOSE: through __builtin_amdgcn_global_load_lds the compiler knows that an asynchronous write to
// LDS is in flight and -- it cannot tell which LDS bytes -- puts `s_waitcnt vmcnt(0)` in front of EVERY later ds_read / ds_write /
// barrier of the wave (seen in the ISA of the first version of this kernel: 7 200 cycles per trip in the copy phase, 4 800 in the
// march).  That was harmless in round 2's kernel, whose copying waves never touched LDS otherwise, and is fatal here, where every
// wave both copies and marches.  The ordering the algorithm needs is established explicitly: a counted `s_waitcnt vmcnt(N)`
// (wait_vm_n) before the trip's barrier.  VMEM operations the compiler does not know about only make ITS counted waits stricter.
__device__ __forceinline__ void lds_dma16(const void *gbase /* wave-uniform */, uint32_t goff, uint32_t lds_addr /* wave-uniform */)
{
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(goff), "s"(gbase), "s"(lds_addr) : "memory");
}
// the trip's barrier: this wave's LDS writes done, then s_barrier (no `vmcnt(0)`: copies stay in flight across it)
__device__ __forceinline__ void trip_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
// wait until at most n vector-memory operations of this wave are outstanding (n uniform)
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

constexpr int kHist = 8;                       // S.ahead <= kHist
constexpr int kAnchorTrips = 8;                // the block's extremes are re-anchored every so many trips
constexpr float kWinMargin = 0.125f;           // slices; the affine position model is exact to ~1e-3
constexpr float kPosOff = 4096.f;              // positions are offset so that their bit patterns order like integers (they can be < 0)

__device__ __forceinline__ int pos_bits(float kc) { return __float_as_int(fminf(fmaxf(kc, -4000.f), 1.0e6f) + kPosOff); }
__device__ __forceinline__ float bits_pos(int b) { return __int_as_float(b) - kPosOff; }

// ---------------------------------------------------------------------------------------------
// kU: sample steps per trip.  Every step more widens the block's slice window by one sample spacing (two slices on C3) --
// ring space that is then not in flight --, every step less means a barrier more per sample.
template <int MAJOR, bool TEX8, bool GRAY, bool INSTR, int kU>
__global__ __launch_bounds__(768) void sweep_kernel(FrameParams P, VolumeView V,
                                                     const float4 *__restrict__ tf,
                                                     const float *__restrict__ rad,
                                                     uint32_t *__restrict__ pixels,
                                                     unsigned long long *__restrict__ counter,
                                                     uint32_t *__restrict__ bricks, SweepArgs S)
{
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    float *lds_tf = (float *)lds;
    Ctl *ctl = (Ctl *)(lds + kTfBytes);

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), NW = S.nc;   // (wave: a scalar for the compiler too)
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
    if (threadIdx.x == 0) {
        ctl->kmin = kInf; ctl->kmax = -1; ctl->err = 0; ctl->cref = 0x7f800000; ctl->dmin = 0x7f800000; ctl->dmax = 0;
        ctl->anc[0][0] = 0x7f800000; ctl->anc[0][1] = 0; ctl->anc[1][0] = 0x7f800000; ctl->anc[1][1] = 0;
        ctl->still[0] = 0; ctl->still[1] = 0; ctl->still[2] = 0;
    }
    if (INSTR) for (int i = threadIdx.x; i < kTab; i += blockDim.x) ctl->own[i] = -1;
    __syncthreads();

    const int nx = V.nx, nr = MAJOR == 2 ? V.ny : V.nz, ns = MAJOR == 2 ? V.nz : V.ny;
    // slice index s along the sweep axis -> position k in sweep order: s, or ns - s = (s ^ -1) + ns + 1
    const int kmul = S.sgn > 0 ? 1 : -1, kadd = S.sgn > 0 ? 0 : ns;
    const int kxor = S.sgn > 0 ? 0 : -1, kxadd = S.sgn > 0 ? 0 : ns + 1;
    const int kpair = S.sgn > 0 ? 0 : -1;                  // slices `is` and `is + 1` sit at k0 + kpair and k0 + kpair + 1
    const float fns = (float)ns;

    // ---- this lane's ray ----
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

    // ---- tile-wide quantities: the slices the tile's rays can touch (with a margin), the block's reference position
    //      (the least advanced first sample) and the extreme slopes ----
    float kc1 = 0.f, kD = 1.f;                   // position of the lane's first sample along the sweep, slices per sample step
    bool ok = false;
    {
        const float isc = MAJOR == 2 ? P.inv_scale[2] : P.inv_scale[1];
        int klo = kInf, khi = -1;
        if (alive && r.dist0 < r.upper) {
            const float os = MAJOR == 2 ? r.origin.z : r.origin.y, dsv = MAJOR == 2 ? r.dir.z : r.dir.y;
            const float ta = __builtin_fmaf((os + dsv * r.dist0) - 0.5f, isc, 0.5f), tb = __builtin_fmaf((os + dsv * r.upper) - 0.5f, isc, 0.5f);
            const float za = ta * fns - 0.5f, zb = tb * fns - 0.5f;
            const int slo = (int)fminf(fmaxf(floorf(fminf(za, zb) - 1.5f), 0.f), (float)(ns - 1));
            const int shi = (int)fminf(fmaxf(floorf(fmaxf(za, zb) + 1.5f), 0.f), (float)(ns - 1)) + 1;
            klo = S.sgn > 0 ? slo : ns - shi; khi = S.sgn > 0 ? shi : ns - slo;
        }
        const float p1 = MAJOR == 2 ? r.origin.z + r.sdir.z : r.origin.y + r.sdir.y;              // first sample (dist0 = 0: no cutting plane here)
        const float z1 = __builtin_fmaf(p1 - 0.5f, isc, 0.5f) * fns - 0.5f;
        const float dz = (MAJOR == 2 ? r.sdir.z : r.sdir.y) * isc * fns;
        kc1 = S.sgn > 0 ? z1 : fns - z1;
        kD = fabsf(dz);
        ok = alive && r.dist0 < r.upper && kD > 1e-6f && kD < 64.f && kc1 == kc1;
        klo = wave_min_fast(klo); khi = -wave_min_fast(-khi);
        const int cb = wave_min_fast(ok ? pos_bits(kc1) : 0x7f800000);
        const int d0 = wave_min_fast(ok ? __float_as_int(kD) : 0x7f800000), d1 = wave_min_fast(ok ? -__float_as_int(kD) : 0);
        if (lane == 0 && khi >= 0) { atomicMin(&ctl->kmin, klo); atomicMax(&ctl->kmax, khi); }
        if (lane == 0) { atomicMin(&ctl->cref, cb); atomicMin(&ctl->dmin, d0); atomicMin(&ctl->dmax, d1); }
    }
    __syncthreads();
    const int kmin = __builtin_amdgcn_readfirstlane(lds_load_i(&ctl->kmin)), kmaxT = __builtin_amdgcn_readfirstlane(lds_load_i(&ctl->kmax));
    float Dmin = __int_as_float(__builtin_amdgcn_readfirstlane(lds_load_i(&ctl->dmin)));
    float Dmax = __int_as_float(-__builtin_amdgcn_readfirstlane(lds_load_i(&ctl->dmax)));
    // the lane's offset: speed only -- any value gives the same pixels
    int o = 0;
    float kA = 0.f;                               // kc_L(tau) = kA + tau * kD
    {
        const float cref = bits_pos(__builtin_amdgcn_readfirstlane(lds_load_i(&ctl->cref)));
        if (ok) { const float q = rintf((kc1 - cref) / kD); o = q >= 0.f ? (q < 30.f ? (int)q : 30) : 0; }
        kA = kc1 - (float)(o + 1) * kD;
    }
    int ring = (kmaxT >= kmin && Dmin <= Dmax && Dmax < 64.f) ? 1 : 0;        // block-uniform: the block streams slices through LDS

    // ---- the copy side: frustum of the tile; the window of the tile in slice k is a closed form in scalars ----
    // Every slice image is S.ry rows of S.pxw voxels (16-byte units, row pitch S.pxw * 4 bytes) in ring slot k mod S.ring; its origin
    // follows the lower bounds of the tile's footprint, x0(k) = ((AX + MX * d) >> 14) & ~3 and r0(k) = (AR + MR * d) >> 14 with
    // d = slice - first slice of the tile, clamped so that the image stays inside the volume's rows: integer arithmetic on wave-uniform
    // values (SALU), the same in every wave, no allocation and no ownership table.  plan_sweep sizes pxw x ry for the widest footprint.
    Frustum F;
    tile_frustum<MAJOR>(P, V, min(x0, P.W - 1), min(y0, P.H - 1), min(x0 + tile_w - 1, P.W - 1), min(y0 + tile_h - 1, P.H - 1), F);
    FootLin FL;
    foot_linear(F, S.sgn > 0, FL);
    const uint32_t Sr = (uint32_t)(MAJOR == 2 ? V.row_bytes : V.slice_bytes);       // bytes between rows of the image
    const uint64_t Ss = MAJOR == 2 ? V.slice_bytes : V.row_bytes;                   // bytes between slices
    const int sl_ref = __builtin_amdgcn_readfirstlane(kmul * max(kmin, 0) + kadd);  // the tile's first slice
    auto fx = [&](float a, float m) {           // bound(sl_ref) - pad, in fixed point (clamped: a tile that misses the volume streams nothing)
        const float v = (__builtin_fmaf(m, (float)sl_ref, a) - kFxPad) * (float)(1 << kFxBits);
        return __builtin_amdgcn_readfirstlane((int)fminf(fmaxf(floorf(v), -1.0e9f), 1.0e9f));
    };
    auto fm = [&](float m) {                    // slope, rounded so that slope * d never exceeds the analytic product (d >= 0 when slices grow along the sweep, else d <= 0)
        const float v = fminf(fmaxf(m, -8.f), 8.f) * (float)(1 << kFxBits);
        return __builtin_amdgcn_readfirstlane((int)(S.sgn > 0 ? floorf(v) : ceilf(v)));
    };
    const int AXi = fx(FL.ax_lo, FL.mx_lo), MXi = fm(FL.mx_lo), ARi = fx(FL.ar_lo, FL.mr_lo), MRi = fm(FL.mr_lo);
    const int pitch = S.pxw * 4, upr = S.pxw >> 2, units = upr * S.ry;               // bytes per image row, 16-byte units per row / per image
    // (a sample clamped to the last voxel / row reads its neighbour nx / nr with weight 0, axis_coord(): the image reaches one further,
    //  into the next row / slice or the padding behind the volume, as the gather kernels' loads do)
    const int x0_max = max(nx + 4 - S.pxw, 0), r0_max = max(nr + 1 - S.ry, 0);
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char *)lds);
    // this lane's 16 bytes of an image: unit u = (wave + j NW) 64 + lane -> LDS byte 16 u (linear), global row u / upr, column u % upr
    uint32_t goff[kCopyMax];
    bool gval[kCopyMax];
    int nq = 0;                                  // LDS-DMA instructions this wave issues per slice
#pragma unroll
    for (int j = 0; j < kCopyMax; ++j) {
        const int u = (wave + j * NW) * 64 + lane, row = u / upr;
        gval[j] = u < units;
        goff[j] = (uint32_t)row * Sr + (uint32_t)(u - row * upr) * 16u;
        if ((wave + j * NW) * 64 < units) nq = j + 1;
    }
    int issued_k = kmin - 1, slot_next = 0;      // highest slice whose copies are issued; the ring slot of the next one (every wave: same values)
    unsigned long long staged = 0;

    // ---- march state (skewed lock step, as march_skew_kernel) ----
    float dist = r.dist0;
    bool ert = false, stop = false, active = alive, lastc = false;
    int i = 31 - o;                           // next sample of the lane's chunk; 31 = open the next chunk now
    int n = 0, chunks = 0;
    float px = 0.f, py = 0.f, pz = 0.f;
    int tau = 1;                              // block step of the trip's first slot (ring mode: the same in every wave)
    int tau0 = 1, apar = 0, trip = 0;
    float lo0 = 0.f, hi0 = 0.f;               // block extremes of kc at tau0
    const int tmax = P.max_chunks * 30 + 30 + kU;
    int n_trips = 0, n_gather_trips = 0;
    unsigned long long tc_issue = 0, tc_wait = 0, tc_barrier = 0, tc_march = 0, tc0 = 0;   // instrumented builds: cycles per phase
    const unsigned long long t_loop0 = S.trace ? __builtin_amdgcn_s_memrealtime() : 0ull;

    auto publish_anchor = [&](int par, int at_tau, bool pend) {
        const float kc = kA + (float)at_tau * kD;
        const int mn = wave_min_fast(pend ? pos_bits(kc) : 0x7f800000), mx = wave_min_fast(pend ? -pos_bits(kc) : 0);
        if (lane == 0 && mn != 0x7f800000) { atomicMin(&ctl->anc[par][0], mn); atomicMin(&ctl->anc[par][1], mx); }
    };
    if (ring) {
        publish_anchor(0, 1, active);
        __syncthreads();
        const int a0 = __builtin_amdgcn_readfirstlane(lds_load_i(&ctl->anc[0][0])), a1 = __builtin_amdgcn_readfirstlane(lds_load_i(&ctl->anc[0][1]));
        if (a0 == 0x7f800000) ring = 0;                                       // no ray of the tile marches
        lo0 = bits_pos(a0); hi0 = bits_pos(-a1);
    }

    for (int t = 0; t < tmax; t += kU) {
        const bool pending = active && !(lastc && i > n) && !(P.ert_true && ert);
        if (!ring && !__any(pending)) break;
        if (ring) {
            // ================= copy side of the trip (identical scalar state in every wave) =================
            const float dt0 = (float)(tau - tau0);
            const float lo_prev = lo0 + (dt0 - (float)kU) * Dmin - kWinMargin;                       // the previous trip may still be computing
            const float hi_now = hi0 + (dt0 + (float)(kU - 1)) * Dmax + kWinMargin;
            const float hi_ahead = hi0 + (dt0 + (float)(kU - 1 + S.ahead * kU)) * Dmax + kWinMargin;
            const int lo_free = __builtin_amdgcn_readfirstlane(max((int)floorf(lo_prev), kmin));                  // slices below are read by nobody any more
            const int need_hi = __builtin_amdgcn_readfirstlane(min((int)floorf(hi_now) + 1, kmaxT));
            const int target_w = min((int)floorf(hi_ahead) + 1, kmaxT);
            if (INSTR) tc0 = __builtin_readcyclecounter();
            // slot k mod ring is free once slice k - ring lies below the window of the previous trip
            const int target = __builtin_amdgcn_readfirstlane(min(target_w, lo_free + S.ring - 1));
            while (issued_k < target) {
                const int k = __builtin_amdgcn_readfirstlane(issued_k) + 1, sl = kmul * k + kadd, d = sl - sl_ref;    // (scalars for the compiler too)
                const int sn = __builtin_amdgcn_readfirstlane(slot_next);
                const int xo = min(max(((AXi + MXi * d) >> kFxBits) & ~3, 0), x0_max);
                const int ro = min(max((ARi + MRi * d) >> kFxBits, 0), r0_max);
                const int img = kRingOff + sn * S.slot_bytes;
                if (threadIdx.x == 0) {
                    const int e = img - ro * pitch - xo * 4;
                    lds_store_i(&ctl->tab[k & (kTab - 1)], e);
                    if ((k & (kTab - 1)) == 0) lds_store_i(&ctl->tab[kTab], e);
                    if (INSTR) { int *bx = ctl->box[k & (kTab - 1)]; bx[0] = xo; bx[1] = xo + S.pxw - 1; bx[2] = ro; bx[3] = ro + S.ry - 1; ctl->own[k & (kTab - 1)] = k; }
                }
                const char *gb = (const char *)V.data + (int64_t)sl * (int64_t)Ss + (uint64_t)(uint32_t)ro * (uint64_t)Sr + (uint64_t)(uint32_t)xo * 4u;
                const uint32_t la = lds_base + (uint32_t)img + (uint32_t)wave * 1024u;
#pragma unroll
                for (int j = 0; j < kCopyMax; ++j)
                    if (j < nq) { if (gval[j]) lds_dma16(gb, goff[j], la + (uint32_t)(j * NW) * 1024u); }
                if (INSTR && wave == 0) staged += (unsigned long long)S.slot_bytes;
                slot_next = sn + 1 == S.ring ? 0 : sn + 1; issued_k = k;
            }
            if (INSTR) { const unsigned long long tn = __builtin_readcyclecounter(); tc_issue += tn - tc0; tc0 = tn; }
            int bail = 0;
            if (issued_k < need_hi) {
                // the slices this very trip reads could not be placed: the block's window does not fit the ring (rays that
                // enter through different faces at the cube's silhouette, strongly diverging rays).  Every wave sees the same
                // numbers: the block leaves the ring here and marches on direct gathers.
                bail = 1;
                if (lane == 0) atomicOr(&ctl->err, 1 << 3);
            } else {
                // wait for THIS wave's units of the slices <= need_hi: copies retire in order, nq per slice
                wait_vm_n(__builtin_amdgcn_readfirstlane((issued_k - need_hi) * nq));
            }
            // the one barrier of the trip: every wave's rows have landed; does any ray still need the stream?
            if (INSTR) { const unsigned long long tn = __builtin_readcyclecounter(); tc_wait += tn - tc0; tc0 = tn; }
            // (word trip % 3: written before this barrier, read after it, cleared two barriers before its next use)
            const int sw = trip % 3;
            const bool wave_needs = __any(pending && !ert);
            if (!bail && lane == 0 && wave_needs) lds_store_i(&ctl->still[sw], 1);
            trip_barrier();
            const int still = __builtin_amdgcn_readfirstlane(lds_load_i(&ctl->still[sw]));
            if (INSTR) { const unsigned long long tn = __builtin_readcyclecounter(); tc_barrier += tn - tc0; tc0 = tn; }
            if (threadIdx.x == 0) lds_store_i(&ctl->still[(trip + 2) % 3], 0);
            if (!still) {
                ring = 0;
                wait_vm<0>();                        // nothing of this wave may land in LDS after the block has moved on
            } else if (trip > 0 && trip % kAnchorTrips == 0) {
                // the anchors published during the previous trip (for this trip's first step)
                const int a0 = __builtin_amdgcn_readfirstlane(lds_load_i(&ctl->anc[apar ^ 1][0])), a1 = __builtin_amdgcn_readfirstlane(lds_load_i(&ctl->anc[apar ^ 1][1]));
                if (a0 != 0x7f800000) { lo0 = bits_pos(a0); hi0 = bits_pos(-a1); tau0 = tau; }
                if (threadIdx.x == 0) { ctl->anc[apar][0] = 0x7f800000; ctl->anc[apar][1] = 0; }   // (read last kAnchorTrips trips ago)
                apar ^= 1;
            }
            ++trip;
        } else {
            // direct gathers: every lane idle until its next chunk boundary -> jump there (march_skew_kernel)
            if (!__any(i <= n && !stop)) {
                int d = pending ? 31 - i : 64;
                d = wave_min_fast(d);
                i += d; tau += d;
            }
            ++n_gather_trips;
        }
        ++n_trips;
        if (INSTR && !ring) tc0 = __builtin_readcyclecounter();
        if (!__any(pending)) {                    // a wave whose rays are done keeps copying its rows for the others
            tau += kU;
            if (ring && (trip % kAnchorTrips) == 0) publish_anchor(apar ^ 1, tau, false);
            continue;
        }
        if (INSTR) slots += (unsigned long long)kU * 64ull;

        uint32_t idx[kU];
        int iu[kU];
        bool lv[kU];
        float ttx[kU], tty[kU], ttz[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
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
            if (ring) {
                uint32_t ix, iy, iz;
                const float wx = axis_coord<TEX8>(tx, (float)V.nx, (float)(V.nx - 1), ix);
                const float wy = axis_coord<TEX8>(ty, (float)V.ny, (float)(V.ny - 1), iy);
                const float wz = axis_coord<TEX8>(tz, (float)V.nz, (float)(V.nz - 1), iz);
                const bool inb = bounds_check(tx, ty, tz);
                const int is = (int)(MAJOR == 2 ? iz : iy), ir = (int)(MAJOR == 2 ? iy : iz);
                const int k0 = (is ^ kxor) + kxadd;              // position of slice `is`; slice is + 1 sits at k0 + kmul
                const i2v T = lds_load_i2(&ctl->tab[(k0 + kpair) & (kTab - 1)]);       // entries of slices `is` and `is + 1`: neighbours in the table
                const int T0 = S.sgn > 0 ? T.x : T.y, T1 = S.sgn > 0 ? T.y : T.x;
                const int x4 = (int)(ix << 2);
                const char *p0 = lds + ((int)__umul24((uint32_t)ir, (uint32_t)pitch) + (T0 + x4));
                const char *p1 = lds + ((int)__umul24((uint32_t)ir, (uint32_t)pitch) + (T1 + x4));
                float2u c00, c10, c01, c11;
                lds_pairs(p0, p0 + pitch, p1, p1 + pitch, c00, c10, c01, c11);
                // MAJOR == 2: rows are y, slices z.  MAJOR == 1: rows are z, slices y -- the lerp order stays x, y, z
                const float2u a_ = c00, b_ = MAJOR == 2 ? c10 : c01, c_ = MAJOR == 2 ? c01 : c10, d_ = c11;
                const float e00 = __builtin_fmaf(wx, a_.y - a_.x, a_.x);
                const float e10 = __builtin_fmaf(wx, b_.y - b_.x, b_.x);
                const float e01 = __builtin_fmaf(wx, c_.y - c_.x, c_.x);
                const float e11 = __builtin_fmaf(wx, d_.y - d_.x, d_.x);
                const float f0 = __builtin_fmaf(wy, e10 - e00, e00);
                const float f1 = __builtin_fmaf(wy, e11 - e01, e01);
                const float L = __builtin_fmaf(wz, f1 - f0, f0);
                const uint32_t id = min((uint32_t)(L * 255.0f), 255u);
                idx[u] = inb ? id : 0u;
                if (INSTR && lv[u] && inb && !stop && !(P.ert_true && ert)) {
                    const int *bx0 = ctl->box[k0 & (kTab - 1)], *bx1 = ctl->box[(k0 + kmul) & (kTab - 1)];
                    const bool okb = (int)ix >= bx0[0] && (int)ix + 1 <= bx0[1] && ir >= bx0[2] && ir + 1 <= bx0[3] &&
                                     (int)ix >= bx1[0] && (int)ix + 1 <= bx1[1] && ir >= bx1[2] && ir + 1 <= bx1[3];
                    // ... and both images must still be the ones of these slices (a page handed on too early would read as a miss)
                    const int o0 = lds_load_i(&ctl->own[k0 & (kTab - 1)]), o1 = lds_load_i(&ctl->own[(k0 + kmul) & (kTab - 1)]);
                    if (!okb || o0 != k0 || o1 != k0 + kmul) ++misses;
                }
            } else {
                idx[u] = V.big ? sample_index<VV_VOXEL_F32, TEX8, true>(V, tx, ty, tz) : sample_index<VV_VOXEL_F32, TEX8, false>(V, tx, ty, tz);
            }
        }
        tau += kU;
#pragma unroll
        for (int u = 0; u < kU; ++u) {
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
        if (INSTR) tc_march += __builtin_readcyclecounter() - tc0;
        // ---- the block's extremes at the next trip's first step, published one trip ahead of their use (`trip` already counts
        //      this trip: the next one re-anchors when it is a multiple of kAnchorTrips) ----
        if (ring && (trip % kAnchorTrips) == 0) {
            const bool pend2 = active && !(lastc && i > n) && !(P.ert_true && ert);
            publish_anchor(apar ^ 1, tau, pend2);
        }
    }
    wait_vm<0>();                             // (a block that never left the ring inside the loop: tmax)
    if (S.trace && wave == 0 && lane == 0) {
        unsigned long long *tr = S.trace + 8ull * blockIdx.x;
        tr[0] = t_blk0; tr[1] = t_loop0; tr[2] = __builtin_amdgcn_s_memrealtime();
        tr[3] = ((unsigned long long)__builtin_amdgcn_s_getreg(63492) << 32) | (unsigned)__builtin_amdgcn_s_getreg(63508);
        tr[4] = ((unsigned long long)trow << 32) | (unsigned)tcol; tr[5] = ((unsigned long long)(unsigned)kmaxT << 32) | (unsigned)kmin; tr[6] = ((unsigned long long)(unsigned)n_gather_trips << 32) | (unsigned)n_trips; tr[7] = 1;
    }

    if (in_frame) {
        if (GRAY) { res_g = res_r; res_b = res_r; }
        pixels[(size_t)y * P.W + x] = write_zero ? 0u : pack_rgba(res_r, res_g, res_b, res_a);
    }
    // errors are always reported (a clamped footprint would mean wrong pixels): counter + 7 holds four 16-bit counts, codes 1..4
    // (code 4 -- a block left the ring because its window did not fit -- is a statistic, not an error: pixels are right)
    // (ctl->err is a bit mask, bit c - 1 for code c, so that the statistic cannot hide an error of the same block)
    if (lane == 0 && wave == 0) {
        const int e = lds_load_i(&ctl->err);
        unsigned long long add = 0ull;
        for (int c = 0; c < 4; ++c) if ((e >> c) & 1) add += 1ull << (16 * c);
        if (add) atomicAdd(counter + 7, add);
    }
    if (INSTR) {
        for (int q = 32; q > 0; q >>= 1) { executed += __shfl_down(executed, q); misses += __shfl_down(misses, q); }
        if (lane == 0 && executed) atomicAdd(counter, executed);
        if (lane == 0 && slots) atomicAdd(counter + 1, slots);
        if (lane == 0 && misses) atomicAdd(counter + 4, misses);
        if (lane == 0 && staged) atomicAdd(counter + 5, staged);
        if (lane == 0) { atomicAdd(counter + 6, (unsigned long long)n_trips); atomicAdd(counter + 13, (unsigned long long)n_gather_trips); }
        if (lane == 0 && wave == 0) { atomicAdd(counter + 8, tc_issue); atomicAdd(counter + 9, tc_wait); atomicAdd(counter + 10, tc_barrier); atomicAdd(counter + 11, tc_march); atomicAdd(counter + 12, (unsigned long long)trip); }
    }
}

// A CU has 160 KB of LDS for a block's static and dynamic allocations together.  The ring takes nearly all of it, so one more
// __shared__ variable in the kernel can push the launch over (round 3: HSA_STATUS_ERROR_INVALID_ALLOCATION aborted the process).
// Once per instantiation: the kernel's static LDS from its attributes, the dynamic limit raised; false = this launch cannot be made.
template <class K>
static bool sweep_launchable(K kern, size_t dynamic_bytes)
{
    static int static_lds = -2;                        // (one per instantiation: K is a distinct function type only per signature, so keyed below -- and per device)
    static const void *keyed = nullptr;
    static int keyed_dev = -1;
    int dv = -1;
    (void)hipGetDevice(&dv);
    if (keyed != (const void *)kern || keyed_dev != dv) {
        hipFuncAttributes fa;
        static_lds = hipFuncGetAttributes(&fa, (const void *)kern) == hipSuccess ? (int)fa.sharedSizeBytes : -1;
        if (static_lds >= 0 && hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsMax - static_lds) != hipSuccess) static_lds = -1;
        (void)hipGetLastError();
        keyed = (const void *)kern; keyed_dev = dv;
    }
    return static_lds >= 0 && sweep_lds_fits((size_t)static_lds, dynamic_bytes);
}
template <int MAJOR, bool TEX8, bool GRAY, bool INSTR>
static bool launch_one(const MarchArgs &a, hipStream_t s)
{
    const SweepArgs &S = a.sweep;
    const unsigned nblocks = S.order ? (unsigned)S.n_order : (unsigned)(((S.nty + 7) / 8) * 8 * S.ntx);
    if (S.steps == 1) {
        auto kern = sweep_kernel<MAJOR, TEX8, GRAY, INSTR, 1>;
        if (!sweep_launchable(kern, (size_t)S.lds_bytes)) return false;
        hipLaunchKernelGGL(kern, dim3(nblocks), dim3((unsigned)(S.nc * 64)), (size_t)S.lds_bytes, s,
                           a.P, a.V, a.tf, a.rad, a.pixels, a.counter, a.bricks, S);
    } else {
        auto kern = sweep_kernel<MAJOR, TEX8, GRAY, INSTR, 2>;
        if (!sweep_launchable(kern, (size_t)S.lds_bytes)) return false;
        hipLaunchKernelGGL(kern, dim3(nblocks), dim3((unsigned)(S.nc * 64)), (size_t)S.lds_bytes, s,
                           a.P, a.V, a.tf, a.rad, a.pixels, a.counter, a.bricks, S);
    }
    return true;
}
template <int MAJOR>
static bool launch_major(const MarchArgs &a, hipStream_t s)
{
    const bool gray = a.gray;
    if (a.tex8) {
        if (gray) return a.instr ? launch_one<MAJOR, true, true, true>(a, s) : launch_one<MAJOR, true, true, false>(a, s);
        return a.instr ? launch_one<MAJOR, true, false, true>(a, s) : launch_one<MAJOR, true, false, false>(a, s);
    }
    if (gray) return a.instr ? launch_one<MAJOR, false, true, true>(a, s) : launch_one<MAJOR, false, true, false>(a, s);
    return a.instr ? launch_one<MAJOR, false, false, true>(a, s) : launch_one<MAJOR, false, false, false>(a, s);
}

} // namespace sweepk

bool sweep_lds_fits(size_t static_bytes, size_t dynamic_bytes) { return static_bytes + dynamic_bytes <= (size_t)sweepk::kLdsMax; }

bool launch_raymarch_sweep(const MarchArgs &a, hipStream_t s)
{
    return a.sweep.major == 2 ? sweepk::launch_major<2>(a, s) : sweepk::launch_major<1>(a, s);
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
    if ((V.row_bytes & 15u) || (V.slice_bytes & 15u) || ((uintptr_t)V.data & 15u) || (V.nx & 3)) VV_NO("rows not 16-byte aligned");
    if (V.nx > 4096 || V.ny > 4096 || V.nz > 4096) VV_NO("more than 4096 slices");     // fixed-point window origins (kFxBits)
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
    S.nl = 0; S.wx = 2; S.wy = 4;              // (every wave copies: no loader waves)
    if (req.wx >= 1 && req.wx <= 8) S.wx = req.wx;
    if (req.wy >= 1 && req.wy <= 14) S.wy = req.wy;
    S.group = 1;
    S.steps = 1;                             // sample steps per trip
    if (req.steps >= 1 && req.steps <= 2) S.steps = req.steps;
    S.ahead = 6;                             // trips the copies run ahead of the march (as far as the ring has room)
    if (req.ahead >= 0 && req.ahead <= kHist) S.ahead = req.ahead;
    if (S.wx * S.wy > 12) VV_NO("too many waves");
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
        // The image of a slice: origin = fixed-point lower bound (at most one voxel below the analytic one: pad + slope error), x aligned
        // down to 4 voxels (16-byte copies); it must reach the footprint's upper end: floor(hi) + 1 - (floor(lo) - 1 [- 3]) + 1 voxels.
        S.pxw = (((int)floor(ext_x + slack) + 8) + 3) & ~3;
        S.ry = (int)floor(ext_r + slack) + 4;
        S.pxw = std::min(S.pxw, V.nx + 4);                                 // (voxel nx / row nr are read with weight 0: the device's x0_max, r0_max)
        S.ry = std::min(S.ry, nr + 1);
        S.pxc = (S.pxw * 4 + 127) / 128;                                  // (row pitch in 128-byte lines, for the record)
        S.slot_bytes = S.pxw * 4 * S.ry;
        // slices a trip reads (its steps + the trilinear neighbour + the skew of the tile's rays around the front) + one trip in flight
        const double dzs = (double)P.step[best] * (double)P.inv_scale[best] * n[best];
        const int need = (int)ceil(S.steps * dzs) + 4 + (int)ceil(S.steps * dzs);
        // two blocks per CU when the ring still holds that, else one block with all of the LDS
        int blocks = req.blocks >= 1 && req.blocks <= 4 ? req.blocks : 0;
        if (!blocks) blocks = (kLdsMax / 2 - kRingOff) / S.slot_bytes >= need ? 2 : 1;
        S.ring = std::min((kLdsMax / blocks - kRingOff) / S.slot_bytes, kTab - 2);
        const int units = (S.pxw / 4) * S.ry;
        if (S.ring >= (forced ? 4 : need) && units <= kCopyMax * 64 * S.nc) break;
        // footprint too large for the LDS (sparse pixels): smaller tiles, else no sweep
        if (verbose) fprintf(stderr, "sweep: tile %dx%d waves needs %d x %d voxels per slice, ring %d of %d\n", S.wx, S.wy, S.pxw, S.ry, S.ring, need);
        if (forced) VV_NO("forced tile shape does not fit");
        if (S.wy > 2) S.wy -= 1; else if (S.wx > 1) { S.wx -= 1; S.wy = 4; } else VV_NO("footprint does not fit the LDS");
    }
    (void)nr;
    S.wmax = S.ring;
    S.lds_bytes = kRingOff + S.ring * S.slot_bytes;
    S.order = nullptr; S.n_order = 0; S.trace = nullptr;
    S.enabled = 1;
    if (verbose) fprintf(stderr, "sweep: axis %d sgn %d, tile %dx%d px (%d waves), image %d voxels x %d rows (ring: %d slices, %d bytes of LDS), %d steps per trip, %d trips ahead, %d x %d tiles\n", S.major, S.sgn, 32 * S.wx, 2 * S.wy, S.nc, S.pxw, S.ry, S.ring, S.lds_bytes, S.steps, S.ahead, S.ntx, S.nty);
#undef VV_NO
}

} // namespace vv
