// vv_raymarch.hip -- front-to-back ray-march kernels for gfx950 (MI355X).
//
// Replaces kernel<_sliceType> and its device functions (kernel.cu:80-367).  Not a
// translation: the reference runs one 16x16 block per 14x14 slab with an LDS byte
// cache and hardware texture fetches; here
//   * rad_kernel      computes the per-slab sphere radius (blockMin, kernel.cu:80-97,329)
//                     once, so the march itself is free of the slab tiling;
//   * march_kernel    (no Phong) gives every wavefront a 32x2 (views along the memory axis) or
//                     8x8 pixel tile, keeps the whole ray state in registers, reconstructs samples with a
//                     branch-free hand-written trilinear gather from linear HBM, and
//                     classifies through a transfer-function table staged in LDS;
//   * march_phong_kernel keeps the slab+apron structure that central differences
//                     across neighbouring rays need (kernel.cu:167-173) with the
//                     32-deep LDS sample cache.
#include "vv_device.h"
#include "vv_kernels.h"

// This file is compiled seven times: as is (linear volume up to 4 GiB), through vv_raymarch_big.hip
// with VV_BIG_VOLUME (linear, 64-bit slice addressing), through vv_raymarch_brick.hip with
// VV_BRICKED (volume sampled from the bricked copy), through vv_raymarch_brick_cached.hip (the same for volumes
// that live in the caches), through vv_raymarch_zpair.hip with VV_ZPAIR (volume sampled from the z-pair copy) and through
// vv_raymarch_zfast.hip with VV_ZFAST (volume sampled from the z-fastest copy: side views) and through vv_raymarch_xpair.hip with
// VV_ZPAIR + VV_XPAIR (the z-pair build on the x-pair copy: side views of small / u8 volumes),
// so that each path pays only for itself.  The builds for cache-resident volumes (this file as is, z-pair, brick_cached)
// are compiled with -fno-slp-vectorize: the packed-f32 code the SLP vectoriser makes of the lerps costs a v_mov per
// operand pair, which VALU-bound frames pay for (C2 -6 %, C1 -4 %, 512^3 -4 %, C2 rotated -6.5 %, Phong -3 ... -6 %),
// while the builds for volumes beyond the caches are 0.5-3 % (rotated + Phong 9 %) faster with it (Makefile).
#if defined(VV_ZPAIR) && defined(VV_XPAIR)
#define VV_BIG_NS xpair
constexpr int kLayout = vv::LAYOUT_ZPAIR;
#elif defined(VV_ZPAIR)
#define VV_BIG_NS zpair
constexpr int kLayout = vv::LAYOUT_ZPAIR;
#elif defined(VV_BRICKED) && defined(VV_BRICKED_CACHED)
#define VV_BIG_NS brickc
constexpr int kLayout = vv::LAYOUT_BRICKED;
#elif defined(VV_BRICKED)
#define VV_BIG_NS brick
constexpr int kLayout = vv::LAYOUT_BRICKED;
#elif defined(VV_ZFAST)
#define VV_BIG_NS zfast
constexpr int kLayout = vv::LAYOUT_ZFAST;
#elif defined(VV_BIG_VOLUME)
#define VV_BIG_NS big
constexpr int kLayout = vv::LAYOUT_LINEAR_BIG;
#else
#define VV_BIG_NS small
constexpr int kLayout = vv::LAYOUT_LINEAR;
#endif

namespace vv {
namespace VV_BIG_NS {

// corner registers and fetch of the layout this translation unit is compiled for
template <int VOXEL> struct CornerSel { using type = Corners<VOXEL>; };
#ifdef VV_ZPAIR
template <> struct CornerSel<VV_VOXEL_F32> { using type = CornersZ; };
template <> struct CornerSel<VV_VOXEL_U8>  { using type = CornersZ8; };
#endif
template <int VOXEL, bool TEX8, class CT>
__device__ __forceinline__ void fetch_any(const VolumeView &V, float px, float py, float pz, CT &C)
{
#ifdef VV_ZPAIR
    fetch_corners_zpair<TEX8>(V, px, py, pz, C);
#else
    fetch_corners<VOXEL, TEX8, kLayout>(V, px, py, pz, C);
#endif
}

// Instrumented frames only: the 128-byte lines (offsets from the sampled layout's base) the gathers of one sample touch -- the address
// arithmetic of fetch_any() for this translation unit's layout, restated (InstrArgs::lines).
template <int VOXEL, bool TEX8>
__device__ __noinline__ void mark_sample_lines(const InstrArgs &I, const VolumeView &V, float px, float py, float pz)
{
    uint32_t ix, iy, iz;
    (void)axis_coord<TEX8>(px, (float)V.nx, (float)(V.nx - 1), ix);
    (void)axis_coord<TEX8>(py, (float)V.ny, (float)(V.ny - 1), iy);
    (void)axis_coord<TEX8>(pz, (float)V.nz, (float)(V.nz - 1), iz);
    constexpr bool F = VOXEL == VV_VOXEL_F32;
#if defined(VV_ZPAIR)
    const uint32_t rec = F ? 8u : 2u, bytes = F ? 16u : 4u;
#ifdef VV_XPAIR
    const uint64_t off = (uint64_t)ix * V.zp_slab_bytes + (uint64_t)iy * V.zp_row_bytes + (uint64_t)iz * rec;
#else
    const uint64_t off = (uint64_t)iz * V.zp_slab_bytes + (uint64_t)iy * V.zp_row_bytes + (uint64_t)ix * rec;
#endif
    mark_line_range(I, off, bytes); mark_line_range(I, off + V.zp_row_bytes, bytes);
#else
    if constexpr (kLayout == LAYOUT_LINEAR || kLayout == LAYOUT_LINEAR_BIG) {
        const uint64_t o = (uint64_t)iz * V.slice_bytes + (uint64_t)iy * V.row_bytes + (F ? ix * 4u : (ix & ~3u));
        mark_line_range(I, o, 8); mark_line_range(I, o + V.row_bytes, 8);
        mark_line_range(I, o + V.slice_bytes, 8); mark_line_range(I, o + V.slice_bytes + V.row_bytes, 8);
    } else if constexpr (kLayout == LAYOUT_ZFAST) {
        const uint64_t o = (uint64_t)ix * V.zf_slice_bytes + (uint64_t)iy * V.zf_row_bytes + (F ? iz * 4u : (iz & ~3u));
        mark_line_range(I, o, 8); mark_line_range(I, o + V.zf_row_bytes, 8);
        mark_line_range(I, o + V.zf_slice_bytes, 8); mark_line_range(I, o + V.zf_slice_bytes + V.zf_row_bytes, 8);
    } else {
        using G = BrickGeom<VOXEL>;
        uint64_t a[4];
        brick_offsets<VOXEL>(V, ix, iy, iz, a);
        if constexpr (F && G::halo == 0) {
            const uint32_t dx = (ix & (G::bx - 1u)) == G::bx - 1u ? G::brick - (G::bx - 1u) * 4u : 4u;
            for (int k = 0; k < 4; ++k) { mark_line_range(I, a[k], 4); mark_line_range(I, a[k] + dx, 4); }
        } else {
            for (int k = 0; k < 4; ++k) mark_line_range(I, a[k], 8);
        }
    }
#endif
}

// ---------------------------------------------------------------------------
// rad pre-pass (blockMin, kernel.cu:80-97,329): one wave per slab, four (clamped) footprint
// pixels per lane, minimum by wave shuffles -- no LDS, no block barrier.  The minimum of a set of
// floats does not depend on the order fminf visits them in, so the value equals the reference's tree.
// ---------------------------------------------------------------------------
// How long the ray through pixel-grid point (gx, gy) -- tile corners of the launch (StripMap), gx in [0, wr], gy in [0, s1 - s0] -- stays inside the cube, as one
// of 64 classes, longest = 0.  Speed only: the same slab test as ray_endpoints, nothing here reaches a pixel.  (Corners, not tile centres: along a silhouette
// edge made by a face seen at a grazing angle the chord jumps from 0 to the face's length, and a tile whose centre misses the cube can hold the frame's longest rays.)
__device__ __forceinline__ int point_life_class(const FrameParams &P, const StripMap &M, int gx, int gy)
{
    const int bl = M.blk_log2w, bh = 256 >> bl;
    const int strip = M.s0 + gy, tile_x = M.tx0 + gx;
    const float x = (float)(tile_x << bl) - 0.5f, y = (float)(M.y0 + (strip / M.strips_per_band) * M.band_stride_px + (strip % M.strips_per_band) * bh) - 0.5f;
    const float sx = ((2.0f * (x + 0.5f)) / (float)P.W - 1.0f) * P.tan_half_x, sy = ((2.0f * (y + 0.5f)) / (float)P.H - 1.0f) * P.tan_half_y;
    float tmin = 0.f, tmax = INFINITY, d2 = 0.f, s2 = 0.f;
    bool miss = false;
    for (int a = 0; a < 3; a++) {
        const float d = P.side[a] * sx + P.up[a] * sy + P.look[a], o = P.cam_pos[a], sc = P.scale[a];
        d2 += d * d; s2 += sc * sc;
        if (d != 0.0f) { const float t1 = (-sc - o) / d, t2 = (sc - o) / d; tmin = fmaxf(tmin, fminf(t1, t2)); tmax = fminf(tmax, fmaxf(t1, t2)); }
        else if (o < -sc || o > sc) miss = true;
    }
    float len = (miss || !(tmin <= tmax)) ? 0.f : (tmax - tmin) * sqrtf(d2);
    const float cls = len / (2.0f * sqrtf(s2)) * 64.f;                    // (the cube's diagonal = the longest chord)
    const int k = cls >= 63.f ? 63 : (cls > 0.f ? (int)cls : 0);
    return 63 - k;
}

constexpr int kMaxUnits = 1024, kMaxPoints = 24576;
__global__ __launch_bounds__(256) void rad_kernel(FrameParams P, float *__restrict__ rad, PixelRect R, uint32_t *__restrict__ pixels,
                                                  StripMap M, uint32_t *__restrict__ order)
{
    if (order && blockIdx.x == gridDim.x - 1) {
        // The extra block: StripMap::order.  Units = runs of order_run x-adjacent tiles of a strip (the tiles that share cache lines stay together, on one
        // XCD); a unit weighs what its tiles' centre rays spend inside the cube.  Units sorted by weight (heaviest first; rank sort, ties in raster order)
        // are dealt to the XCDs to and fro (0..7, 7..0, ...): every XCD gets the same load to within one light unit and ends on its lightest ones.
        // Table slot ((r * 8 + xcd) * run + i) holds tile i of the r-th unit of that XCD, or ~0 (a unit shorter than a run, the last round).
        __shared__ uint32_t w[kMaxUnits];
        __shared__ uint8_t pts[kMaxPoints];
        const int run = M.order_run, nseg = (M.wr + run - 1) / run, ns = M.s1 - M.s0, U = ns * nseg, n = M.wr * ns;
        const int rounds = (U + 7) / 8;
        for (int i = threadIdx.x; i < U; i += 256) w[i] = 0;
        for (int i = threadIdx.x; i < rounds * 8 * run; i += 256) order[i] = ~0u;
        __syncthreads();
        {
            // a tile weighs what the longest of its four corner rays spends in the cube
            const int gw = M.wr + 1, np = gw * (ns + 1);
            for (int i = threadIdx.x; i < np; i += 256) pts[i] = (uint8_t)point_life_class(P, M, i % gw, i / gw);
            __syncthreads();
            for (int t = threadIdx.x; t < n; t += 256) {
                const int tx = t % M.wr, ty = t / M.wr;
                const int c = min(min((int)pts[ty * gw + tx], (int)pts[ty * gw + tx + 1]), min((int)pts[(ty + 1) * gw + tx], (int)pts[(ty + 1) * gw + tx + 1]));
                atomicAdd(&w[ty * nseg + tx / run], 64u - (uint32_t)c);
            }
        }
        __syncthreads();
        for (int u = threadIdx.x; u < U; u += 256) {
            const uint32_t wu = w[u];
            int rank = 0;
            for (int v = 0; v < U; ++v) { const uint32_t wv = w[v]; rank += (wv > wu || (wv == wu && v < u)) ? 1 : 0; }
            const int c = rank & 7, r = rank >> 3, xcd = (r & 1) ? 7 - c : c;
            const int strip_l = u / nseg, x0 = (u % nseg) * run, size = min(run, M.wr - x0);
            uint32_t *dst = order + (size_t)(r * 8 + xcd) * run;
            for (int i = 0; i < size; ++i) dst[i] = (uint32_t)(strip_l * M.wr + x0 + i);
        }
        return;
    }
    const int slab = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (slab >= P.nbx * P.nby) return;                                     // wave-uniform
    const int bx = slab % P.nbx, by = slab / P.nbx;
    // a shard only needs the radii of the slab rows that own its pixel rows: its own bands
    // (bands are whole slab rows) and, when H == 1 (mod 14), the last slab row (pin 10)
    if (!row_owned(P, by * kSlab) && !(P.conflict_y && by == P.nby - 1)) return;
    // Pixels outside the rectangle march_kernel's tiles cover: their rays miss the volume (vv_render: screen_rect), the frame holds 0 there
    // (kernel.cu:334-338).  This wave writes the ones among the slab's 14 x 14 pixels that the frame owns (not column W-1 / row H-1, not another shard's rows).
    const int sx0 = bx * kSlab, sy0 = by * kSlab;
    const bool meets = sx0 < R.x1 && sx0 + kSlab > R.x0 && sy0 < R.y1 && sy0 + kSlab > R.y0;
    const bool inside = sx0 >= R.x0 && sx0 + kSlab <= R.x1 && sy0 >= R.y0 && sy0 + kSlab <= R.y1;
    if (!inside && row_owned(P, sy0)) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int t = lane + 64 * q, tx = t % kSlab, ty = t / kSlab;
            const int x = sx0 + tx, y = sy0 + ty;
            if (t < kSlab * kSlab && x <= P.W - 2 && y <= P.H - 2 && !(x >= R.x0 && x < R.x1 && y >= R.y0 && y < R.y1)) pixels[(size_t)y * P.W + x] = 0u;
        }
    }
    // no radius for a slab none of whose own pixels march_kernel visits (it reads rad[owner_slab(pixel)]); the slabs that own pixel W-2 / H-2
    // without containing it (pin 10) always compute theirs
    if (!rad || (!meets && !(P.conflict_x && bx == P.nbx - 1) && !(P.conflict_y && by == P.nby - 1))) return;      // (Phong frames: no radii wanted, march_phong_kernel finds its own)
    const int lox = slab_lo(bx), upx = slab_up(bx, P.W) - 1, loy = slab_lo(by), upy = slab_up(by, P.H) - 1;
    float m = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int t = lane + 64 * q, tx = t & 15, ty = t >> 4;
        int x = bx * kSlab + tx - 1, y = by * kSlab + ty - 1;              // kernel.cu:294-295
        x = max(lox, min(x, upx));                                         // :307-308
        y = max(loy, min(y, upy));
        f3 front, back;
        ray_endpoints(P, x, y, front, back);
        const float cl = vlen3(front.x - P.cam_pos[0], front.y - P.cam_pos[1], front.z - P.cam_pos[2]);  // :323-325
        m = q == 0 ? cl : fminf(m, cl);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fminf(m, __shfl_xor(m, o));
    if (lane == 0) rad[slab] = m;
}

// ---------------------------------------------------------------------------
// helpers
// ---------------------------------------------------------------------------
__device__ __forceinline__ void stage_tf(float4 *lds_tf, const float4 *__restrict__ tf)
{
    for (int i = threadIdx.x; i < 256; i += blockDim.x) lds_tf[i] = tf[i];
    __syncthreads();
}
// march_kernel keeps the table channel-planar in LDS: entry idx of channel c sits in bank idx % 32, so the data-dependent look-up of a wave
// spreads over all banks (the float4 layout hits 8 of 32).

// ---------------------------------------------------------------------------
// march_kernel: no Phong.  blockDim = 256 = 4 waves; a block owns a 32x8 pixel strip (64x4 / 128x2 with StripMap::blk_log2w 6 / 7), each
// wave a 2^tw x 2^(6-tw) tile of it (32x2 when screen x runs along the volume's x axis, so
// that the lanes of a gather walk one memory row; 8x8 otherwise).  blockIdx.x enumerates
// (strip, tile) pairs of the shard (StripMap).
//
// Tuning (MI355X, measured, profiles/EXPERIMENTS.md part B section 4): the kernel lives off the 32 KB L1 of its CU
// (lanes of one gather share sectors, consecutive rows of a wave share lines), so FEWER resident
// waves are faster: the launcher reserves unused dynamic LDS to cap a CU at 1-3 blocks
// (MarchArgs::lds_reserve) and each lane keeps U samples = 4U gathers in flight instead.
// ---------------------------------------------------------------------------
template <int SLICE, int VOXEL, bool TEX8, bool GRAY, bool INSTR, int U>
__global__ __launch_bounds__(256) void march_kernel(FrameParams P, VolumeView V,
                                                    const float4 *__restrict__ tf,
                                                    const float *__restrict__ rad,
                                                    uint32_t *__restrict__ pixels,
                                                    unsigned long long *__restrict__ counter,
                                                    InstrArgs I, StripMap M)
{
    __shared__ float lds_tf[1024];

#ifdef VV_TIMELINE
    const unsigned long long tl0 = wall_clock64();
#endif
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bl = M.blk_log2w, ntx = M.wr;                               // block = 2^bl x (256 >> bl) pixels; the launch covers ntx tile columns from M.tx0 on
    int strip, tile_x;
    if (M.order) {
        // runs of order_run consecutive tiles of the list go to one XCD (block L runs on XCD L % 8): x-neighbours share an L2 as in the strip order below
        const int L = blockIdx.x, xcd = L & 7, j = L >> 3, pos = ((j / M.order_run) * 8 + xcd) * M.order_run + j % M.order_run;
        const uint32_t t = M.order[pos];                              // (the grid is exactly the table)
        if (t == ~0u) return;                                         // block-uniform, before any barrier
        strip = M.s0 + (int)t / ntx; tile_x = M.tx0 + (int)t % ntx;
    } else if (M.xcd_band > 0) {
        // XCD-aware order (speed only): linear block L runs on XCD L % 8 (round-robin dispatch);
        // XCD k walks bands k, k+8, ... of xcd_band strips so that neighbouring tiles share an L2
        const int L = blockIdx.x, per_band = ntx * M.xcd_band;
        const int xcd = L & 7, j = L >> 3;
        const int band = (j / per_band) * 8 + xcd, w = j % per_band;
        strip = M.s0 + band * M.xcd_band + w / ntx; tile_x = M.tx0 + w % ntx;
    } else { strip = M.s0 + blockIdx.x / ntx; tile_x = M.tx0 + blockIdx.x % ntx; }
    // wave tile = 2^tw x 2^(6-tw) pixels; the 4 waves of a block tile a 32x8 strip
    const int tw = M.tile_log2w, th = 6 - tw;
    const int wx = wave & (((1 << bl) >> tw) - 1), wy = wave >> (bl - tw);
    const int x = (tile_x << bl) + (wx << tw) + (lane & ((1 << tw) - 1));
    const int y = M.y0 + (strip / M.strips_per_band) * M.band_stride_px + (strip % M.strips_per_band) * (256 >> bl) + (wy << th) + (lane >> tw);
    if (strip >= M.s1) return;                             // block-uniform, before any barrier
    // The table entry this thread stages is loaded now and parked in LDS behind the ray set-up (which needs no table): its latency hides behind
    // the set-up's divisions instead of standing in front of them (blockDim.x == 256 == entries: every launch of this kernel).  C3 -0.6 %, C2 -1.1 %.
    const float4 tf_entry = tf[threadIdx.x];
    // pixels the reference never writes: column W-1 / row H-1 (W,H >= 2)
    const int xmax = P.W >= 2 ? P.W - 2 : 0, ymax = P.H >= 2 ? P.H - 2 : 0;
    const bool in_frame = x <= xmax && y <= ymax && row_owned(P, y);

    float res_r = 0.f, res_g = 0.f, res_b = 0.f, res_a = 0.f;
    unsigned long long executed = 0, slots = 0;
    bool write_zero = false;

    Ray r;
    int alive = 0;
    if (in_frame) {
        f3 front, back;
        ray_endpoints(P, x, y, front, back);
        float length = vlen3(back.x - front.x, back.y - front.y, back.z - front.z);
        if (length < 0.001f) {
            write_zero = true;                                       // kernel.cu:334-338
        } else {
            float rd;
            if (P.W < 2 || P.H < 2) {      // degenerate footprint: blockMin scans nothing (kernel.cu:89)
                rd = vlen3(front.x - P.cam_pos[0], front.y - P.cam_pos[1], front.z - P.cam_pos[2]);
            } else {
                int ox = owner_slab(x, P.W, P.nbx, P.conflict_x), oy = owner_slab(y, P.H, P.nby, P.conflict_y);
                rd = rad[oy * P.nbx + ox];
            }
            setup_ray(P, front, back, rd, r);
            alive = r.cut_return ? 0 : 1;
        }
    }
    if (!alive) { r.upper = -1.f; r.dist0 = 0.f; r.sstep = 1.f; r.origin = mk3(0, 0, 0); r.dir = r.origin; r.sdir = r.origin; }
    // (Round 5 also let a block none of whose rays meets the volume -- 62 % of C3's blocks -- leave right here, before the table wait: 0 ... +1 % on
    //  every workload, +1-2 % on the rotated view.  Empty blocks are cheap enough as they are; not kept.  profiles/EXPERIMENTS.md part A5.)
    lds_tf[threadIdx.x] = tf_entry.x; lds_tf[256 + threadIdx.x] = tf_entry.y; lds_tf[512 + threadIdx.x] = tf_entry.z; lds_tf[768 + threadIdx.x] = tf_entry.w;
    __syncthreads();

    float dist = r.dist0;
    bool ert = false;
    const f3 sp = mk3(P.slice_point[0], P.slice_point[1], P.slice_point[2]);
    const f3 sn = mk3(P.slice_normal[0], P.slice_normal[1], P.slice_normal[2]);

    // chunk loop: kernel.cu:248-278.  All lanes of the wave walk chunks together;
    // a lane whose own `while (dist < upper)` has ended simply has n == 0.
    for (int chunk = 0; chunk < P.max_chunks && __any(dist < r.upper); ++chunk) {
        int n = chunk_count(dist, r.upper, r.sstep);
        // reference ERT: a ray past the threshold composites sample 1 of every later chunk and breaks again (:272-274)
        // -- as long as its opacity cannot fall back under the threshold, i.e. for tables with opacities in [0, 1];
        // otherwise every chunk runs the reference's per-sample test in full
        if (ert && P.alpha_unit) n = min(n, 1);
        float px, py, pz;
        {
#pragma clang fp contract(off)
            px = r.origin.x + r.dir.x * dist;                        // :249
            py = r.origin.y + r.dir.y * dist;
            pz = r.origin.z + r.dir.z * dist;
        }
        // The trip count of the sample loop is made wave-uniform once per chunk (5 ballots: n <= 30),
        // and blend / early termination are predicated instead of branched: scalar branches and
        // exec-mask bookkeeping inside this loop cost measurable time (MI355X, C3: 1.60 -> 1.5x ms).
        int nmax = 0;
#pragma unroll
        for (int bit = 16; bit > 0; bit >>= 1)
            if (__any(n >= (nmax | bit))) nmax |= bit;
        bool stop = false;
#ifndef VV_BRICKED
        // Tail: no lane has more than one sample left in this chunk -- rays past the ERT threshold composite sample 1 of every later chunk (pin 4), finished
        // rays none -- and that stays so: a terminated ray stays terminated (opacities in [0, 1]: alpha_unit), and a ray whose chunk holds one sample ends in it.
        // Such chunks would each cost a memory round trip for one sample (and U - 1 gathers nobody uses): take sample 1 of U consecutive chunks in one trip.
        // Same operations per ray in the same order; only chunks that hold nothing are visited differently.  C3 -0.6 ... -1.1 % (gathers -0.9 %, EA bytes -1.6 %),
        // u8 -1.6 %, 512^3 -1.3 %, the 3840 x 2160 / step 1/1024 frame -4.7 %.  Not in the bricked build: with this block present the compiler schedules that
        // build's main loop differently and the rotated view loses 9 % whether the block runs or not (A/B against the previous library, tools/ab_rounds.sh).
        if (nmax <= 1 && M.tail_batch) {
            float du[U], tx[U], ty[U], tz[U];
            int nu[U];
            typename CornerSel<VOXEL>::type C[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
#pragma clang fp contract(off)
                du[u] = u == 0 ? dist : du[u - 1] + r.sstep * kChunkSteps;                // :277
                nu[u] = u == 0 ? n : min(chunk_count(du[u], r.upper, r.sstep), 1);
                if (chunk + u >= P.max_chunks) nu[u] = 0;
                float qx = r.origin.x + r.dir.x * du[u], qy = r.origin.y + r.dir.y * du[u], qz = r.origin.z + r.dir.z * du[u];   // :249
                qx += r.sdir.x; qy += r.sdir.y; qz += r.sdir.z;                            // :141
                tx[u] = __builtin_fmaf(qx - 0.5f, P.inv_scale[0], 0.5f);
                ty[u] = __builtin_fmaf(qy - 0.5f, P.inv_scale[1], 0.5f);
                tz[u] = __builtin_fmaf(qz - 0.5f, P.inv_scale[2], 0.5f);
                fetch_any<VOXEL, TEX8>(V, tx[u], ty[u], tz[u], C[u]);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (INSTR) slots += 64ull * U;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t k = classify_index<VOXEL>(C[u], tx[u], ty[u], tz[u]);
                // a ray that crossed the threshold in this batch under ERT_TRUE has upper = -1 from its next chunk on (:258 below)
                const bool live = nu[u] >= 1 && !(P.ert_true && ert);
                float cr, cg, cb, ca;
                ca = lds_tf[768 + k];
                cr = lds_tf[k];
                if (GRAY) { cg = cb = cr; }
                else { cg = lds_tf[256 + k]; cb = lds_tf[512 + k]; }
                if (SLICE == SLICE_PLANE) {                                              // :193-198
#pragma clang fp contract(off)
                    float vd = 1.f * r.sstep + du[u];                                    // :254
                    float vx = r.origin.x + r.dir.x * vd, vy = r.origin.y + r.dir.y * vd, vz = r.origin.z + r.dir.z * vd;
                    float d = fabsf(sn.x * (vx - sp.x) + sn.y * (vy - sp.y) + sn.z * (vz - sp.z));
                    if (d < .01f) cr = fmaxf(0.f, fminf(cr + (.01f - d) * 100.f, 1.f));
                }
                if (INSTR) {
                    const bool inv = bounds_check(tx[u], ty[u], tz[u]);
                    if (live) {
                        executed++;
                        if (I.bricks && inv) mark_bricks(I.bricks, V, tx[u], ty[u], tz[u]);
                    }
                    if ((I.lines || I.pairs) && (I.lines_all || (live && inv))) mark_sample_lines<VOXEL, TEX8>(I, V, tx[u], ty[u], tz[u]);
                }
                {
#pragma clang fp contract(off)
                    const float bf = (live && ca > kEps) ? ca * (1.f - res_a) : 0.f;
                    res_r = res_r + cr * bf;
                    if (!GRAY) { res_g = res_g + cg * bf; res_b = res_b + cb * bf; }
                    res_a = res_a + bf;
                }
                ert = ert || (live && res_a > P.ert_thr);                                // :272-274
            }
            if (P.ert_true && ert) r.upper = -1.f;
            {
#pragma clang fp contract(off)
                dist = du[U - 1] + r.sstep * kChunkSteps;
            }
            chunk += U - 1;
            continue;
        }
#endif
        if (INSTR) slots += (unsigned long long)((nmax + U - 1) / U * U) * 64ull;   // lane slots this wave spends
        // U samples per trip: their gathers are all issued before the first is consumed.  With
        // the block count per CU capped (lds_reserve) registers are plentiful and the extra
        // loads in flight pay.
        for (int i0 = 1; i0 <= nmax; i0 += U) {
            float tx[U], ty[U], tz[U];
            typename CornerSel<VOXEL>::type C[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                px += r.sdir.x; py += r.sdir.y; pz += r.sdir.z;      // :141 (sample i = i increments)
                // (pos - .5) / scale + .5 as fma(pos - .5, 1/scale, .5)   :136, DESIGN.md pin 3
                tx[u] = __builtin_fmaf(px - 0.5f, P.inv_scale[0], 0.5f);
                ty[u] = __builtin_fmaf(py - 0.5f, P.inv_scale[1], 0.5f);
                tz[u] = __builtin_fmaf(pz - 0.5f, P.inv_scale[2], 0.5f);
                fetch_any<VOXEL, TEX8>(V, tx[u], ty[u], tz[u], C[u]);
            }
            __builtin_amdgcn_sched_barrier(0);           // all 4U gathers are issued before the first is consumed
            uint32_t idx[U];
#pragma unroll
            for (int u = 0; u < U; ++u) idx[u] = classify_index<VOXEL>(C[u], tx[u], ty[u], tz[u]);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = i0 + u;
                const bool live = i <= n && !stop;
                float cr, cg, cb, ca;
                ca = lds_tf[768 + idx[u]];
                cr = lds_tf[idx[u]];
                if (GRAY) { cg = cb = cr; }                      // r == g == b
                else { cg = lds_tf[256 + idx[u]]; cb = lds_tf[512 + idx[u]]; }
                if (SLICE == SLICE_PLANE) {                                              // :193-198
#pragma clang fp contract(off)
                    float vd = (float)i * r.sstep + dist;                                // :254
                    float vx = r.origin.x + r.dir.x * vd, vy = r.origin.y + r.dir.y * vd, vz = r.origin.z + r.dir.z * vd;
                    float d = fabsf(sn.x * (vx - sp.x) + sn.y * (vy - sp.y) + sn.z * (vz - sp.z));
                    if (d < .01f) cr = fmaxf(0.f, fminf(cr + (.01f - d) * 100.f, 1.f));
                }
                if (INSTR) {
                    const bool inv = bounds_check(tx[u], ty[u], tz[u]);
                    if (live) {
                        executed++;
                        if (I.bricks && inv) mark_bricks(I.bricks, V, tx[u], ty[u], tz[u]);
                    }
                    if ((I.lines || I.pairs) && (I.lines_all || (live && inv))) mark_sample_lines<VOXEL, TEX8>(I, V, tx[u], ty[u], tz[u]);
                }
                {
                    // :268-270 + blend :107-118, predicated: with bf == 0 the sums are unchanged
                    // bit for bit (the table is finite: vv_set_transfer_function rejects NaN/Inf)
#pragma clang fp contract(off)
                    const float bf = (live && ca > kEps) ? ca * (1.f - res_a) : 0.f;
                    res_r = res_r + cr * bf;
                    if (!GRAY) { res_g = res_g + cg * bf; res_b = res_b + cb * bf; }
                    res_a = res_a + bf;
                }
                const bool hit = live && res_a > P.ert_thr;                              // :272-274
                stop = stop || hit;
                ert = ert || hit;
            }
        }
        if (P.ert_true && ert) r.upper = -1.f;
        {
#pragma clang fp contract(off)
            dist += r.sstep * kChunkSteps;                                           // :277
        }
    }

    if (in_frame) {
        if (GRAY) { res_g = res_r; res_b = res_r; }
        pixels[(size_t)y * P.W + x] = write_zero ? 0u : pack_rgba(res_r, res_g, res_b, res_a);
    }
#ifdef VV_TIMELINE
    {
        const bool any_live = __any(alive) != 0;
        if (I.timeline && threadIdx.x == 0) {
            unsigned long long *t = I.timeline + 4ull * blockIdx.x;
            t[0] = tl0; t[1] = wall_clock64(); t[2] = ((unsigned long long)strip << 16) | (unsigned)tile_x;
            t[3] = (__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u) | (any_live ? 256u : 0u) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) << 16);      // HW_REG_XCC_ID, HW_REG_HW_ID
        }
    }
#endif
    if (INSTR) {
        for (int o = 32; o > 0; o >>= 1) executed += __shfl_down(executed, o);
        if (lane == 0 && executed) atomicAdd(counter, executed);
        if (lane == 0 && slots) atomicAdd(counter + 1, slots);      // developer statistic: lane utilisation
        if (kLayout == LAYOUT_BRICKED && lane == 0 && executed) atomicAdd(counter + 2, 1ull);   // waves that sampled the bricked copy
        if (kLayout == LAYOUT_ZPAIR && lane == 0 && executed) atomicAdd(counter + 3, 1ull);     // ... the z-pair copy
    }
}

// x / d and sqrt(x), IEEE-exact, without their range handling.  The compiler's expansion of `/` and sqrtf is a fixed core (v_rcp + one
// Newton step, quotient + two residual corrections; v_sqrt + a test of the two neighbouring floats) wrapped in v_div_scale x 2 + v_div_fixup
// resp. a 2^32 pre-scale and a class test: 11 and 14 instructions.  With operands known to be normal and far from the ends of the range --
// decided per frame on the host (FrameParams::safe_div: pixel tangents in [2^-24, 2^8], steps in [1e-5, 16]; numerators are 0 or differences
// of q / 255, so quotients lie in [2^-26, 2^42] and their squares' sum in [2^-52, 2^86]) -- the wrappers do nothing (v_div_scale returns its
// operand, v_div_fmas is a plain fma, v_div_fixup passes normal quotients through) and the cores alone give the same bits: 8 and 8 instructions,
// 18 fewer per lit sample.  Round 4: C3 + Phong -4 %, C2 + Phong -11 %, u8 -9 %, the 3840 x 2160 frame -11 % (profiles/r04_phong_forms.txt F).
__device__ __forceinline__ float ph_div_core(float n, float d)
{
    const float y0 = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, y0, 1.0f);
    const float y = __builtin_fmaf(e, y0, y0);
    const float q0 = n * y;
    const float r0 = __builtin_fmaf(-d, q0, n);
    const float q1 = __builtin_fmaf(r0, y, q0);
    const float r1 = __builtin_fmaf(-d, q1, n);
    return __builtin_fmaf(r1, y, q1);
}
__device__ __forceinline__ float ph_sqrt_core(float x)
{
    const float s0 = __builtin_amdgcn_sqrtf(x);
    const float sm = __uint_as_float(__float_as_uint(s0) - 1u), sp = __uint_as_float(__float_as_uint(s0) + 1u);
    const float t1 = __builtin_fmaf(-sm, s0, x);
    float s = (0.f >= t1) ? sm : s0;
    const float t2 = __builtin_fmaf(-sp, s0, x);
    s = (0.f < t2) ? sp : s;
    return s;
}
// ---------------------------------------------------------------------------
// march_phong_kernel: one block per reference slab (14x14 interior + apron),
// 32-deep byte cache in LDS exactly as kernel.cu:125-145 lays it out, but indexed
// by thread (so clamped apron threads own private entries with identical content).
// Neighbour look-ups clamp to the slab footprint (DESIGN.md pin 6) and every
// thread keeps marching its own chunk sequence until the whole block is done, so
// neighbour entries are never stale (pin 5).
// ---------------------------------------------------------------------------
template <int SLICE, int VOXEL, bool TEX8, bool INSTR>
// Registers: on the linear layout up to 4 GiB the kernel is compiled for 5 waves per SIMD (83-85 VGPRs: the volumes that
// live in the caches want 5 blocks per CU; C1 0.196 -> 0.174 ms, 256^3 1.36 -> 1.28); the variants for volumes beyond
// 4 GiB and for the bricked copy run 2-3 blocks per CU and are 2-7 % faster with the 106 VGPRs the compiler takes by itself.
#if defined(VV_BIG_VOLUME) || defined(VV_BRICKED) || defined(VV_ZFAST)
#define VV_PHONG_OCC
#else
#define VV_PHONG_OCC __attribute__((amdgpu_waves_per_eu(5)))
#endif
__global__ __launch_bounds__(256) VV_PHONG_OCC void march_phong_kernel(FrameParams P, VolumeView V,
                                                          const float4 *__restrict__ tf, SlabMap M,
                                                          uint32_t *__restrict__ pixels,
                                                          unsigned long long *__restrict__ counter,
                                                          InstrArgs I)
{
    __shared__ float4 lds_tf[256];
    __shared__ float red[256];
    __shared__ uint8_t cache[kCacheDepth][256];
    __shared__ float q255[256];              // q / 255.f for every byte q, by the same IEEE division
    __shared__ int any_live[2];              // refresh depth of the chunk of iteration `it`: any_live[it & 1]
    const int tid = threadIdx.x;
#ifdef VV_TIMELINE
    const unsigned long long tl0 = wall_clock64();
#endif
    // XCD-aware order (speed only, as in march_kernel): linear block L runs on XCD L % 8; XCD k takes the
    // grid rows k, k+8, ... so that the slabs of one row, which share volume lines, share an L2
    const int nbxg = M.wg;                   // (slab columns the launch covers, from M.gx0 on)
#ifndef VV_PHONG_BAND
#define VV_PHONG_BAND 1
#endif
    constexpr int BAND = VV_PHONG_BAND;      // vertically adjacent slab rows per XCD, dispatched next to each other
    const int j_ = (int)blockIdx.x >> 3;
    const int gx = M.gx0 + (j_ / BAND) % nbxg, lr = ((j_ / (BAND * nbxg)) * 8 + ((int)blockIdx.x & 7)) * BAND + j_ % BAND;
    if (lr > M.gs1 - M.gs0) return;                            // block-uniform, before any barrier
    const int gy = lr < M.gs1 - M.gs0 ? M.gs0 + lr : M.n_regular;     // (the launch's last row is the extra one)
    stage_tf(lds_tf, tf);
    // kernel.cu:175-177 divides six cached bytes by 255.f per shaded sample; a correctly rounded
    // division is ~10 instructions, a table look-up of the same quotient is one LDS read
    q255[tid] = (float)tid / 255.f;                      // visible after the barriers of the reduction below
    if (tid < 2) any_live[tid] = 0;

    // grid row gy -> slab row of this shard; the last grid row is the "extra" slab row
    // nby-1 that re-writes pixel row H-2 when H == 1 (mod 14) (pin 10): it travels with
    // the shard that owns pixel row H-2.
    const int bx = gx;
    int by;
    if (gy == M.n_regular) { if (!P.conflict_y) return; by = P.nby - 1; }
    else by = M.r0 + (gy / M.band) * M.band_stride + (gy % M.band);
    if (by >= P.nby) return;                                   // block-uniform
    // a sharded grid walks whole bands of slab rows: when H == 1 (mod 14) its last one is the "extra" row, which only
    // the last grid row may march (twice would write the same pixels again and count their samples twice)
    if (gy != M.n_regular && P.conflict_y && by == P.nby - 1) return;
    {
        // ownership is decided on the pixel rows this slab row writes
        int yrow = (P.conflict_y && by == P.nby - 1) ? P.H - 2 : by * kSlab;
        if (yrow > (P.H >= 2 ? P.H - 2 : 0) || !row_owned(P, yrow)) return;   // block-uniform
    }
    const int tx = tid & 15, ty = tid >> 4;
    const int lox = slab_lo(bx), upx = slab_up(bx, P.W), loy = slab_lo(by), upy = slab_up(by, P.H);
    const bool degenerate = (upx - lox) <= 0 || (upy - loy) <= 0;
    int x = bx * kSlab + tx - 1, y = by * kSlab + ty - 1;
    x = max(lox, min(x, upx - 1)); y = max(loy, min(y, upy - 1));
    const bool border = tx == 0 || ty == 0 || tx == 15 || ty == 15;              // kernel.cu:304-305

    f3 front, back;
    ray_endpoints(P, x, y, front, back);
    float cl = vlen3(front.x - P.cam_pos[0], front.y - P.cam_pos[1], front.z - P.cam_pos[2]);
    red[tid] = cl;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) red[tid] = fminf(red[tid], red[tid + s]);
        __syncthreads();
    }
    float rd = degenerate ? cl : red[0];
    float length = vlen3(back.x - front.x, back.y - front.y, back.z - front.z);
    const bool skip = length < 0.001f && !border;                                 // :334
    Ray r;
    setup_ray(P, front, back, rd, r);

    // write ownership (pin 10) and one writer per pixel
    const int ox = owner_slab(x, P.W, P.nbx, P.conflict_x), oy = owner_slab(y, P.H, P.nby, P.conflict_y);
    bool writer = !border && ox == bx && oy == by;
    // among interior threads clamped onto the same pixel keep the one whose unclamped
    // coordinate equals the pixel, or (pin 10 case: none does) the first interior one
    {
        int ux = bx * kSlab + tx - 1, uy = by * kSlab + ty - 1;
        bool xrep = (ux == x) || (bx * kSlab > x && tx == 1);
        bool yrep = (uy == y) || (by * kSlab > y && ty == 1);
        writer = writer && xrep && yrep;
    }
    // neighbour thread indices, clamped to the footprint
    int nl, nr, nt, nb;
    if (degenerate) { nl = nr = nt = nb = tid; }
    else {
        int xl = max(lox, min(x - 1, upx - 1)), xr = max(lox, min(x + 1, upx - 1));
        int yt = max(loy, min(y + 1, upy - 1)), yb = max(loy, min(y - 1, upy - 1));
        int fx0 = bx * kSlab - 1, fy0 = by * kSlab - 1;
        nl = (y - fy0) * 16 + (xl - fx0); nr = (y - fy0) * 16 + (xr - fx0);
        nt = (yt - fy0) * 16 + (x - fx0); nb = (yb - fy0) * 16 + (x - fx0);
    }

    float res_r = 0.f, res_g = 0.f, res_b = 0.f, res_a = 0.f;
    unsigned long long executed = 0;
    float dist = r.dist0;
    bool ert_done = false;
    const bool marching = writer && !skip && !r.cut_return;
    const f3 sp = mk3(P.slice_point[0], P.slice_point[1], P.slice_point[2]);
    const f3 sn = mk3(P.slice_normal[0], P.slice_normal[1], P.slice_normal[2]);

    // How deep a chunk's cache has to be: a compositing ray reads its entries 0 .. n+1 and its neighbours' 1 .. n,
    // n = the samples of the chunk before `vd > upper` (:254).  A ray whose opacity already passed the threshold (it
    // keeps compositing one sample per chunk, pin 4) reads entries 0 .. 2 only: with every table opacity in [0, 1]
    // the accumulated opacity cannot fall back under the threshold (res_a <= 1 stays true in float arithmetic:
    // fl(r + fl(c * fl(1 - r))) <= 1 for r, c in [0, 1]).  The block refreshes the deepest need of its rays -- the
    // reference refreshes all 32 always (:125-145); entries nobody reads are not observable.  The need of the next chunk is
    // posted (a block-wide maximum in LDS) right after a chunk is shaded and read behind the barrier that ends the shading:
    // two barriers per chunk.
    auto post_need = [&](const int slot) {
        int d = 0;
        if (marching && !ert_done && dist < r.upper) {
#pragma clang fp contract(off)
            if (P.alpha_unit && res_a > P.ert_thr) d = 3;
            else if (!(30.f * r.sstep + dist > r.upper)) d = kCacheDepth;
            else {
                int n = 0;
#pragma unroll 1
                for (int i = 1; i < kCacheDepth - 1; ++i) { const float vd = (float)i * r.sstep + dist; if (vd > r.upper) break; n = i; }
                d = n + 2;
            }
        }
        // wave maximum: nearly always decided by two ballots (some ray at full depth, or every ray past the threshold)
        const bool full = __builtin_amdgcn_ballot_w64(d == kCacheDepth) != 0ull, deep = __builtin_amdgcn_ballot_w64(d > 3) != 0ull;
        if (full) d = kCacheDepth;
        else if (!deep) d = __builtin_amdgcn_ballot_w64(d != 0) != 0ull ? 3 : 0;
        else d = wave_max_i(d);
        if ((threadIdx.x & 63) == 0 && d) atomicMax(&any_live[slot], d);
    };
    post_need(0);
    __syncthreads();
    for (int chunk = 0; chunk < P.max_chunks; ++chunk) {
        const bool mine = marching && !ert_done && dist < r.upper;
        const int depth = any_live[chunk & 1];                // complete: posted before the barrier above / at the end of the last chunk
        if (!depth) break;
        if (threadIdx.x == 0) any_live[(chunk + 1) & 1] = 0;      // posted to behind the next barrier, read behind the one after
        // rayMarch: every thread refreshes its 32 cache entries for this chunk  :125-145
        {
            float px, py, pz;
            {
#pragma clang fp contract(off)
                px = r.origin.x + r.dir.x * dist; py = r.origin.y + r.dir.y * dist; pz = r.origin.z + r.dir.z * dist;
            }
            // 4 samples (16 gathers) in flight per trip, issued before any is consumed
            // samples in flight per trip (measured: 2, 4 and 8 are within 1 % of each other except on the
            // bricked copy, where 2 is 6 % ahead of 4)
#ifndef VV_PHONG_PU
#ifdef VV_BRICKED
#define VV_PHONG_PU 2
#else
#define VV_PHONG_PU 4
#endif
#endif
            constexpr int PU = VV_PHONG_PU;
            {
                auto refresh = [&](const int i0) {
                    float tx_[PU], ty_[PU], tz_[PU];
                    typename CornerSel<VOXEL>::type C[PU];
#pragma unroll
                    for (int u = 0; u < PU; ++u) {
                        tx_[u] = __builtin_fmaf(px - 0.5f, P.inv_scale[0], 0.5f);
                        ty_[u] = __builtin_fmaf(py - 0.5f, P.inv_scale[1], 0.5f);
                        tz_[u] = __builtin_fmaf(pz - 0.5f, P.inv_scale[2], 0.5f);
                        fetch_any<VOXEL, TEX8>(V, tx_[u], ty_[u], tz_[u], C[u]);
                        px += r.sdir.x; py += r.sdir.y; pz += r.sdir.z;
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < PU; ++u) {
                        const int i = i0 + u;
                        cache[i][tid] = (uint8_t)classify_index<VOXEL>(C[u], tx_[u], ty_[u], tz_[u]);
                        if (INSTR) {
                            // (entries 1..30 of a compositing ray are the samples it can execute; entries 0 / 31 and the apron threads' are gradient-only)
                            const bool need = mine && i >= 1 && i <= 30 && bounds_check(tx_[u], ty_[u], tz_[u]);
                            if (I.bricks && need) mark_bricks(I.bricks, V, tx_[u], ty_[u], tz_[u]);
                            if ((I.lines || I.pairs) && (I.lines_all || need)) mark_sample_lines<VOXEL, TEX8>(I, V, tx_[u], ty_[u], tz_[u]);
                        }
                    }
                };
                // the full depth keeps its compile-time trip count (the short form alone cost cache-resident volumes 5-13 %)
                if (depth == kCacheDepth) { for (int j0 = 0; j0 < kCacheDepth; j0 += PU) refresh(j0); }
                else { for (int j0 = 0; j0 < depth; j0 += PU) refresh(j0); }
            }
        }
        __syncthreads();
        if (mine) {
            // the ray's own cache column as a rolling window (entries i - 1, i, i + 1): one byte read per sample instead of three, and the table look-up of
            // sample i no longer waits for a byte read of the same iteration (C3 + Phong -0.6 %, C2 + Phong -1.4 %, others +-0.3 %)
            uint32_t s_prev = cache[0][tid], s_cur = cache[1][tid];
            for (int i = 1; i < kCacheDepth - 1; ++i) {
#pragma clang fp contract(off)
                float vd = (float)i * r.sstep + dist;                                 // :254
                if (vd > r.upper) break;
                if (INSTR) executed++;
                const uint32_t s = s_cur, s_next = cache[i + 1][tid];                 // (entry i + 1 <= n + 1: inside the refreshed depth)
                const uint32_t s_before = s_prev;
                s_prev = s_cur; s_cur = s_next;
                float4 e = lds_tf[s];
                float cr = e.x, cg = e.y, cb = e.z, ca = e.w;
                if (ca > kEps) {                                                      // :164 (phong is on)
                    const uint32_t qf = s_before, qa = s_next;
                    const uint32_t ql = cache[i][nl], qr = cache[i][nr], qt = cache[i][nt], qb = cache[i][nb];
                    float direct = 0.f;
                    // all three central differences zero (inside a plateau): the gradient is (0,0,0),
                    // it is not normalised (:180) and direct = clamp(0) = 0 -- skip the divisions
                    if (!(qr == ql && qt == qb && qa == qf)) {
                        float f = q255[qf], a = q255[qa], l = q255[ql], rr = q255[qr], t = q255[qt], b = q255[qb];
                        // :175-178, :259-263.  The three divisions, the square root and the reciprocal are IEEE-exact either way; when the frame's
                        // operands are known to stay far inside the normal range (FrameParams::safe_div) their cores alone give the same bits (ph_div_core)
                        float gx, gy, gz;
                        if (P.safe_div) {
                            gx = ph_div_core(rr - l, P.tan_fov_x * vd); gy = ph_div_core(t - b, P.tan_fov_y * vd); gz = ph_div_core(a - f, r.sstep * 2.f);
                            if (gx != 0.f && gy != 0.f && gz != 0.f) {
                                float inv = ph_div_core(1.0f, ph_sqrt_core(gx * gx + gy * gy + gz * gz));
                                gx *= inv; gy *= inv; gz *= inv;
                            }
                        } else {
                            gx = (rr - l) / (P.tan_fov_x * vd); gy = (t - b) / (P.tan_fov_y * vd); gz = (a - f) / (r.sstep * 2.f);
                            if (gx != 0.f && gy != 0.f && gz != 0.f) {
                                float inv = 1.0f / sqrtf(gx * gx + gy * gy + gz * gz);
                                gx *= inv; gy *= inv; gz *= inv;
                            }
                        }
                        direct = (gx * -1.f + gy * -1.f + gz * 1.f) * 0.3f;           // :183
                        direct = fmaxf(0.f, fminf(direct, 0.3f));
                    }
                    cr = cr * 0.7f + direct; cg = cg * 0.7f + direct; cb = cb * 0.7f + direct;
                }
                if (SLICE == SLICE_PLANE) {
                    float vx = r.origin.x + r.dir.x * vd, vy = r.origin.y + r.dir.y * vd, vz = r.origin.z + r.dir.z * vd;
                    float d = fabsf(sn.x * (vx - sp.x) + sn.y * (vy - sp.y) + sn.z * (vz - sp.z));
                    if (d < .01f) cr = fmaxf(0.f, fminf(cr + (.01f - d) * 100.f, 1.f));
                }
                if (ca > kEps) {
                    float bf = ca * (1.f - res_a);
                    res_r = res_r + cr * bf; res_g = res_g + cg * bf; res_b = res_b + cb * bf; res_a = res_a + bf;
                }
                if (res_a > P.ert_thr) { if (P.ert_true) ert_done = true; break; }
            }
        }
        {
#pragma clang fp contract(off)
            dist += r.sstep * kChunkSteps;
        }
        if (chunk + 1 < P.max_chunks) post_need((chunk + 1) & 1);
        __syncthreads();      // cache is rewritten next iteration
    }

    if (writer)
        pixels[(size_t)y * P.W + x] = skip ? 0u : pack_rgba(res_r, res_g, res_b, res_a);
#ifdef VV_TIMELINE
    if (I.timeline && threadIdx.x == 0) {
        unsigned long long *t = I.timeline + 4ull * blockIdx.x;
        t[0] = tl0; t[1] = wall_clock64(); t[2] = ((unsigned long long)by << 16) | (unsigned)bx;
        t[3] = (__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u) | 256u | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) << 16);
    }
#endif
    if (INSTR) {
        for (int o = 32; o > 0; o >>= 1) executed += __shfl_down(executed, o);
        if ((threadIdx.x & 63) == 0 && executed) atomicAdd(counter, executed);
        if (kLayout == LAYOUT_BRICKED && (threadIdx.x & 63) == 0 && executed) atomicAdd(counter + 2, 1ull);
        if (kLayout == LAYOUT_ZPAIR && (threadIdx.x & 63) == 0 && executed) atomicAdd(counter + 3, 1ull);
    }
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
template <int SLICE, int VOXEL, bool TEX8, bool GRAY, bool INSTR>
static void launch_march(const MarchArgs &a, hipStream_t s)
{
    const int ntx = a.strips.wr, ns = a.strips.s1 - a.strips.s0;          // the tiles under the volume's screen rectangle (StripMap)
    if (ntx <= 0 || ns <= 0) return;
    unsigned nblocks = (unsigned)(ns * ntx);
    if (a.strips.order) {
        const int nseg = (ntx + a.strips.order_run - 1) / a.strips.order_run, units = ns * nseg;
        nblocks = (unsigned)((units + 7) / 8 * 8 * a.strips.order_run);          // (rad_kernel's table: rounds of 8 units)
    } else if (a.strips.xcd_band > 0) {
        const int nbands = (ns + a.strips.xcd_band - 1) / a.strips.xcd_band;
        nblocks = (unsigned)(((nbands + 7) / 8) * 8 * a.strips.xcd_band * ntx);
    }
    dim3 grid(nblocks);
    // a.lds_reserve bytes of (unused) dynamic LDS cap the number of resident blocks per CU:
    // fewer waves share the 32 KB L1, which this gather kernel needs more than latency hiding
    if (a.unroll == 3)
        hipLaunchKernelGGL((march_kernel<SLICE, VOXEL, TEX8, GRAY, INSTR, 3>), grid, dim3(256), (size_t)a.lds_reserve, s,
                           a.P, a.V, a.tf, a.rad, a.pixels, a.counter, a.I, a.strips);
    else
        hipLaunchKernelGGL((march_kernel<SLICE, VOXEL, TEX8, GRAY, INSTR, 2>), grid, dim3(256), (size_t)a.lds_reserve, s,
                           a.P, a.V, a.tf, a.rad, a.pixels, a.counter, a.I, a.strips);
}
template <int SLICE, int VOXEL, bool TEX8, bool INSTR>
static void launch_phong(const MarchArgs &a, hipStream_t s)
{
    const int rows = a.slabs.gs1 - a.slabs.gs0 + 1;           // grid rows, dealt to the 8 XCDs round-robin
    constexpr int BAND = VV_PHONG_BAND;
    if (a.slabs.wg <= 0) return;
    dim3 grid((unsigned)(((rows + 8 * BAND - 1) / (8 * BAND)) * 8 * BAND * a.slabs.wg));
    hipLaunchKernelGGL((march_phong_kernel<SLICE, VOXEL, TEX8, INSTR>), grid, dim3(256), (size_t)a.lds_reserve_phong, s,
                       a.P, a.V, a.tf, a.slabs, a.pixels, a.counter, a.I);
}

template <int SLICE, int VOXEL, bool TEX8>
static void dispatch3(const MarchArgs &a, hipStream_t s)
{
    if (a.phong) {
        if (a.instr) launch_phong<SLICE, VOXEL, TEX8, true>(a, s); else launch_phong<SLICE, VOXEL, TEX8, false>(a, s);
        return;
    }
    const bool gray = a.gray && SLICE != SLICE_PLANE;
    if (gray) { if (a.instr) launch_march<SLICE, VOXEL, TEX8, true, true>(a, s); else launch_march<SLICE, VOXEL, TEX8, true, false>(a, s); }
    else      { if (a.instr) launch_march<SLICE, VOXEL, TEX8, false, true>(a, s); else launch_march<SLICE, VOXEL, TEX8, false, false>(a, s); }
}
template <int SLICE>
static void dispatch2(const MarchArgs &a, hipStream_t s)
{
    if (a.V_type == VV_VOXEL_F32) { if (a.tex8) dispatch3<SLICE, VV_VOXEL_F32, true>(a, s); else dispatch3<SLICE, VV_VOXEL_F32, false>(a, s); }
    else                          { if (a.tex8) dispatch3<SLICE, VV_VOXEL_U8,  true>(a, s); else dispatch3<SLICE, VV_VOXEL_U8,  false>(a, s); }
}

static void launch_rad_impl(const MarchArgs &a, hipStream_t s)
{
    // one wave per slab of the frame; waves of slab rows this shard does not own exit at once
    dim3 grid((unsigned)((a.P.nbx * a.P.nby + 3) / 4) + (a.order_out ? 1u : 0u));        // (+ the block that sorts the tiles: StripMap::order)
    hipLaunchKernelGGL(rad_kernel, grid, dim3(256), 0, s, a.P, a.rad_out, a.rect, a.pixels, a.strips, a.order_out);
}

static void launch_raymarch_impl(const MarchArgs &a, hipStream_t s)
{
    switch (a.P.slice_type) {                       // kernel.cu:429-447
    case SLICE_PLANE:     dispatch2<SLICE_PLANE>(a, s); break;
    case SLICE_PLANE_CUT: dispatch2<SLICE_PLANE_CUT>(a, s); break;
    default:              dispatch2<SLICE_NONE>(a, s); break;
    }
}

} // namespace VV_BIG_NS

#if defined(VV_ZPAIR) && defined(VV_XPAIR)
void launch_raymarch_xpair(const MarchArgs &a, hipStream_t s) { xpair::launch_raymarch_impl(a, s); }
#elif defined(VV_ZPAIR)
void launch_raymarch_zpair(const MarchArgs &a, hipStream_t s) { zpair::launch_raymarch_impl(a, s); }
#elif defined(VV_BRICKED) && defined(VV_BRICKED_CACHED)
void launch_raymarch_bricked_cached(const MarchArgs &a, hipStream_t s) { brickc::launch_raymarch_impl(a, s); }
#elif defined(VV_BRICKED)
void launch_raymarch_bricked(const MarchArgs &a, hipStream_t s) { brick::launch_raymarch_impl(a, s); }
#elif defined(VV_ZFAST)
void launch_raymarch_zfast(const MarchArgs &a, hipStream_t s) { zfast::launch_raymarch_impl(a, s); }
#elif defined(VV_BIG_VOLUME)
void launch_raymarch_big(const MarchArgs &a, hipStream_t s) { big::launch_raymarch_impl(a, s); }
#else
void launch_rad(const MarchArgs &a, hipStream_t s) { small::launch_rad_impl(a, s); }
void launch_raymarch(const MarchArgs &a, hipStream_t s) { small::launch_raymarch_impl(a, s); }
#endif

} // namespace vv
