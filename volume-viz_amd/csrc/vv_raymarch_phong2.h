// vv_raymarch_phong2.h -- march_phong2_kernel: the Phong-shaded march with a double-buffered sample cache and S slabs per block.
// Included by vv_raymarch.hip inside namespace vv::<layout>, once per volume layout.
//
// Same semantics as march_phong_kernel (kernel.cu:125-145 rayMarch, :147-201 shadeVoxel, :248-278 the chunk loop; a reference
// slab + apron per 256 rays, pins 5 and 6), different schedule.  The first kernel alternates two block-wide phases per chunk --
// every thread gathers its 32 cache entries, barrier, the interior threads shade 30 samples one by one, barrier -- so within a
// block memory latency and shading arithmetic never overlap, and every shaded sample waits for four dependent LDS round trips
// (own byte -> table entry -> six neighbour bytes -> six quotients).  Here
//   * the cache has two planes: while chunk c is shaded out of plane c & 1, the gathers of chunk c + 1 are issued in groups of
//     four samples, each group's loads staying in flight across the shading of four samples of chunk c, and land in plane
//     (c + 1) & 1.  One barrier per chunk instead of four;
//   * a thread's 32 entries are contiguous (9 dwords per thread: 32 bytes + 4 of padding, so rows start on all 32 banks) and
//     samples are shaded in batches of four aligned entries: one dword of the own row and one of each neighbour's row serve
//     four samples, the quotients of two samples are fetched together, the arithmetic is predicated instead of branched per
//     lane (wave-uniform skips remain), and only the blend runs sample by sample;
//   * S = 2: a block of 256 threads marches TWO x-adjacent slabs, every thread one ray of each (same position in the slab).  The
//     two rays of a thread are 14 pixels apart, so their gathers fall into the same or neighbouring 128-byte lines and are issued
//     back to back: a 16-pixel slab row spans about 24 voxels = 1.75 lines on the 1024^3 volume at 1080p, two rows side by
//     side 45 voxels = 2.4 lines instead of 3.5.  Each slab keeps its own rad, apron and cache (kernel.cu:329, :167-173).
//
// How deep chunk c + 1 has to be refreshed is decided one chunk early, from the rays' state before chunk c is shaded (a ray
// that is still compositing then may cross the ERT threshold in chunk c: its chunk c + 1 is refreshed deeper than it turns out
// to need -- never shallower: a ray's need does not grow).  Entries nobody reads are not observable; frames are bit-identical.
//
// The kernel serves frames whose shading divisions need no range handling (FrameParams::safe_div, below); vv_render gives the
// others to march_phong_kernel.
#pragma once

// x / d for the three screen-space differences and 1 / sqrt(x) for the normalisation, IEEE-exact.  The compiler's expansion of
// `/` and sqrtf is a fixed core (v_rcp + one Newton step, quotient + two residual corrections; v_sqrt + a test of the two
// neighbouring floats) wrapped in range handling (v_div_scale x 2, v_div_fixup; a 2^32 pre-scale and a class test): 11 and 14
// instructions.  With operands known to be normal and far from the ends of the range -- decided per frame on the host
// (FrameParams::safe_div: pixel tangents in [2^-24, 2^8], steps <= 16; numerators are 0 or differences of q / 255) -- the
// wrappers do nothing and the cores alone give the same bits: 8 and 8 instructions, 18 fewer per shaded sample.
__device__ __forceinline__ float div_core(float n, float d)
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
__device__ __forceinline__ float sqrt_core(float x)
{
    const float s0 = __builtin_amdgcn_sqrtf(x);
    const float sm = __uint_as_float(__float_as_uint(s0) - 1u), sp = __uint_as_float(__float_as_uint(s0) + 1u);
    const float t1 = __builtin_fmaf(-sm, s0, x);
    float s = (0.f >= t1) ? sm : s0;
    const float t2 = __builtin_fmaf(-sp, s0, x);
    s = (0.f < t2) ? sp : s;
    return s;
}

// The same two cores on two samples at a time: gfx950 executes v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 at two results per
// lane and cycle, each half rounded exactly like the scalar instruction, so a pair costs 7 + 2 (v_rcp) instead of 16.
typedef float __attribute__((ext_vector_type(2))) ph_f2;
__device__ __forceinline__ ph_f2 ph_fma2(ph_f2 a, ph_f2 b, ph_f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ ph_f2 div_core2(ph_f2 n, ph_f2 d)
{
    const ph_f2 y0 = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)}, one = {1.0f, 1.0f};
    const ph_f2 e = ph_fma2(-d, y0, one);
    const ph_f2 y = ph_fma2(e, y0, y0);
    const ph_f2 q0 = n * y;
    const ph_f2 r0 = ph_fma2(-d, q0, n);
    const ph_f2 q1 = ph_fma2(r0, y, q0);
    const ph_f2 r1 = ph_fma2(-d, q1, n);
    return ph_fma2(r1, y, q1);
}
__device__ __forceinline__ ph_f2 sqrt_core2(ph_f2 x)
{
    const ph_f2 s0 = {__builtin_amdgcn_sqrtf(x.x), __builtin_amdgcn_sqrtf(x.y)};
    const ph_f2 sm = {__uint_as_float(__float_as_uint(s0.x) - 1u), __uint_as_float(__float_as_uint(s0.y) - 1u)};
    const ph_f2 sp = {__uint_as_float(__float_as_uint(s0.x) + 1u), __uint_as_float(__float_as_uint(s0.y) + 1u)};
    const ph_f2 t1 = ph_fma2(-sm, s0, x), t2 = ph_fma2(-sp, s0, x);
    ph_f2 s;
    s.x = (0.f >= t1.x) ? sm.x : s0.x; s.y = (0.f >= t1.y) ? sm.y : s0.y;
    s.x = (0.f < t2.x) ? sp.x : s.x;   s.y = (0.f < t2.y) ? sp.y : s.y;
    return s;
}

constexpr int kPhRow = 9;                    // dwords per thread and cache plane

// one ray of a thread (a thread marches S of them, one per slab of the block)
template <int VOXEL, int PU>
struct PhRay {
    Ray r;
    float s30, dz2, dist, dnext;
    float res_r, res_g, res_b, res_a;
    float px, py, pz;                        // position of the next cache entry to gather
    int nl, nr, nt, nb;                      // rows (dword offsets into a plane) of the neighbour threads, clamped to the footprint (pin 6)
    int x;
    bool marching, ert_done, writer, skip;
    int n;                                   // samples of the open chunk this lane runs unless it terminates inside (kernel.cu:253-257)
    bool live;                               // still inside the chunk's inner loop
    uint32_t wprev, wcur;                    // own dwords b - 1 and b of the open chunk
    uint32_t gin;                            // byte u = 0xff if sample u of the group in flight lies inside the volume (kernel.cu:65-71)
    typename CornerSel<VOXEL>::type C[PU];   // the group in flight
};

#ifndef VV_PHONG2_WAVES
#define VV_PHONG2_WAVES 3
#endif
template <int SLICE, int VOXEL, bool TEX8, bool INSTR, int S>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(S == 2 ? 2 : VV_PHONG2_WAVES))) void march_phong2_kernel(FrameParams P, VolumeView V,
                                                           const float4 *__restrict__ tf, SlabMap M,
                                                           uint32_t *__restrict__ pixels,
                                                           unsigned long long *__restrict__ counter,
                                                           uint32_t *__restrict__ bricks)
{
    constexpr int PU = 4 / S;                                // samples (4 gathers each) a ray has in flight per step: the block keeps 16 gathers per lane in flight
    constexpr int ROW = kPhRow;
    static_assert(S == 1 || S == 2, "one or two slabs per block");
    using RayT = PhRay<VOXEL, PU>;
    __shared__ float4 lds_tf[256];
    __shared__ uint32_t cache[2][S][256 * ROW];
    __shared__ float q255[256];              // q / 255.f for every byte q, by the same IEEE division
    __shared__ int need[3];                  // refresh depth of chunk c: need[c % 3]
    float *red = (float *)&cache[1][0][0];   // the rad reductions borrow the second plane (dead until the first refresh into it)
    const int tid = threadIdx.x;
    // XCD-aware order (speed only, as in march_kernel): linear block L runs on XCD L % 8; XCD k takes the
    // grid rows k, k+8, ... so that the slabs of one row, which share volume lines, share an L2
    const int nbxg = (P.nbx + S - 1) / S;
    const int j_ = (int)blockIdx.x >> 3;
    const int gx_ = j_ % nbxg, gy_ = (j_ / nbxg) * 8 + ((int)blockIdx.x & 7);
    if (gy_ > M.n_regular) return;                             // block-uniform, before any barrier
    stage_tf(lds_tf, tf);
    q255[tid] = (float)tid / 255.f;                      // visible after the barriers of the reduction below
    if (tid < 3) need[tid] = 0;

    // grid row -> slab row of this shard; the last grid row is the "extra" slab row nby-1 that re-writes pixel
    // row H-2 when H == 1 (mod 14) (pin 10): it travels with the shard that owns pixel row H-2.
    int by;
    if (gy_ == M.n_regular) { if (!P.conflict_y) return; by = P.nby - 1; }
    else by = M.r0 + (gy_ / M.band) * M.band_stride + (gy_ % M.band);
    if (by >= P.nby) return;                                   // block-uniform
    if (gy_ != M.n_regular && P.conflict_y && by == P.nby - 1) return;
    {
        int yrow = (P.conflict_y && by == P.nby - 1) ? P.H - 2 : by * kSlab;
        if (yrow > (P.H >= 2 ? P.H - 2 : 0) || !row_owned(P, yrow)) return;   // block-uniform
    }
    const int tx = tid & 15, ty = tid >> 4;
    const int own = tid * ROW;
    const bool border = tx == 0 || ty == 0 || tx == 15 || ty == 15;              // kernel.cu:304-305
    const int loy = slab_lo(by), upy = slab_up(by, P.H);
    int y = by * kSlab + ty - 1;
    y = max(loy, min(y, upy - 1));
    const bool has1 = S == 1 || gx_ * S + 1 < P.nbx;             // the block's second slab exists (block-uniform)

    RayT R[S];
    float cl[S];
    f3 front[S], back[S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int bx = gx_ * S + ((s == 0 || has1) ? s : 0);     // (a missing second slab shadows the first: it gathers what the first gathers and writes nothing)
        const int lox = slab_lo(bx), upx = slab_up(bx, P.W);
        int x = bx * kSlab + tx - 1;
        x = max(lox, min(x, upx - 1));
        R[s].x = x;
        ray_endpoints(P, x, y, front[s], back[s]);
        cl[s] = vlen3(front[s].x - P.cam_pos[0], front[s].y - P.cam_pos[1], front[s].z - P.cam_pos[2]);
        red[s * 256 + tid] = cl[s];
    }
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (tid < w) {
#pragma unroll
            for (int s = 0; s < S; ++s) red[s * 256 + tid] = fminf(red[s * 256 + tid], red[s * 256 + tid + w]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int s = 0; s < S; ++s) {
        RayT &Q = R[s];
        const bool present = s == 0 || has1;
        const int bx = gx_ * S + (present ? s : 0);
        const int lox = slab_lo(bx), upx = slab_up(bx, P.W);
        const bool degenerate = (upx - lox) <= 0 || (upy - loy) <= 0;
        const int x = Q.x;
        const float rd = degenerate ? cl[s] : red[s * 256];
        const float length = vlen3(back[s].x - front[s].x, back[s].y - front[s].y, back[s].z - front[s].z);
        Q.skip = length < 0.001f && !border;                                      // :334
        setup_ray(P, front[s], back[s], rd, Q.r);
        // write ownership (pin 10) and one writer per pixel
        const int ox = owner_slab(x, P.W, P.nbx, P.conflict_x), oy = owner_slab(y, P.H, P.nby, P.conflict_y);
        bool writer = present && !border && ox == bx && oy == by;
        {
            int ux = bx * kSlab + tx - 1, uy = by * kSlab + ty - 1;
            bool xrep = (ux == x) || (bx * kSlab > x && tx == 1);
            bool yrep = (uy == y) || (by * kSlab > y && ty == 1);
            writer = writer && xrep && yrep;
        }
        Q.writer = writer;
        if (degenerate) { Q.nl = Q.nr = Q.nt = Q.nb = tid; }
        else {
            int xl = max(lox, min(x - 1, upx - 1)), xr = max(lox, min(x + 1, upx - 1));
            int yt = max(loy, min(y + 1, upy - 1)), yb = max(loy, min(y - 1, upy - 1));
            int fx0 = bx * kSlab - 1, fy0 = by * kSlab - 1;
            Q.nl = (y - fy0) * 16 + (xl - fx0); Q.nr = (y - fy0) * 16 + (xr - fx0);
            Q.nt = (yt - fy0) * 16 + (x - fx0); Q.nb = (yb - fy0) * 16 + (x - fx0);
        }
        Q.nl *= ROW; Q.nr *= ROW; Q.nt *= ROW; Q.nb *= ROW;
        Q.res_r = Q.res_g = Q.res_b = Q.res_a = 0.f;
        Q.dist = Q.r.dist0;
        Q.ert_done = false;
        Q.marching = writer && !Q.skip && !Q.r.cut_return;
        {
#pragma clang fp contract(off)
            Q.s30 = Q.r.sstep * kChunkSteps;                                      // :277
            Q.dz2 = Q.r.sstep * 2.f;                                              // :263
            Q.dnext = Q.dist + Q.s30;
        }
        Q.n = 0; Q.live = false; Q.wprev = Q.wcur = 0; Q.gin = 0; Q.px = Q.py = Q.pz = 0.f;
    }
    unsigned long long executed = 0;
    const f3 sp = mk3(P.slice_point[0], P.slice_point[1], P.slice_point[2]);
    const f3 sn = mk3(P.slice_normal[0], P.slice_normal[1], P.slice_normal[2]);

    // Cache depth a ray needs in the chunk that starts at distance D, judged from its present state: a compositing ray reads
    // its entries 0 .. n+1 and its neighbours' 1 .. n, n = the samples of the chunk before `vd > upper` (:254); a ray past the
    // ERT threshold (it keeps compositing one sample per chunk, pin 4) reads entries 0 .. 2 -- with every table opacity in
    // [0, 1] accumulated opacity cannot fall back under the threshold.
    auto need_at = [&](const RayT &Q, float D) -> int {
#pragma clang fp contract(off)
        if (!(Q.marching && !Q.ert_done && D < Q.r.upper)) return 0;
        if (P.alpha_unit && Q.res_a > P.ert_thr) return 3;
        if (!(30.f * Q.r.sstep + D > Q.r.upper)) return kCacheDepth;
        return chunk_count(D, Q.r.upper, Q.r.sstep) + 2;
    };
    // block-wide maximum into need[slot]: nearly always decided by two ballots per wave
    auto post_need = [&](int d, int slot) {
        const bool full = __builtin_amdgcn_ballot_w64(d == kCacheDepth) != 0ull, deep = __builtin_amdgcn_ballot_w64(d > 3) != 0ull;
        if (full) d = kCacheDepth;
        else if (!deep) d = __builtin_amdgcn_ballot_w64(d != 0) != 0ull ? 3 : 0;
        else d = wave_max_i(d);
        if ((threadIdx.x & 63) == 0 && d) atomicMax(&need[slot], d);
    };
    auto need_all = [&](bool next, bool on) -> int {
        int d = 0;
        if (on) {
#pragma unroll
            for (int s = 0; s < S; ++s) d = max(d, need_at(R[s], next ? R[s].dnext : R[s].dist));
        }
        return d;
    };

    // ---- refresh machinery: a running position along the ray, PU samples per step ----
    auto chunk_origin = [&](RayT &Q, float D) {
#pragma clang fp contract(off)
        Q.px = Q.r.origin.x + Q.r.dir.x * D; Q.py = Q.r.origin.y + Q.r.dir.y * D; Q.pz = Q.r.origin.z + Q.r.dir.z * D;   // :249 / :131
    };
    auto issue = [&](RayT &Q) {
        Q.gin = 0;
#pragma unroll
        for (int u = 0; u < PU; ++u) {
            const float tx_ = __builtin_fmaf(Q.px - 0.5f, P.inv_scale[0], 0.5f);
            const float ty_ = __builtin_fmaf(Q.py - 0.5f, P.inv_scale[1], 0.5f);
            const float tz_ = __builtin_fmaf(Q.pz - 0.5f, P.inv_scale[2], 0.5f);
            fetch_any<VOXEL, TEX8>(V, tx_, ty_, tz_, Q.C[u]);
            Q.gin |= bounds_check(tx_, ty_, tz_) ? (0xffu << (8 * u)) : 0u;
            Q.px += Q.r.sdir.x; Q.py += Q.r.sdir.y; Q.pz += Q.r.sdir.z;      // :141
        }
    };
    // entries g * PU .. + PU - 1 of the thread's row, packed
    auto land = [&](RayT &Q, uint32_t *plane, int g) {
        uint32_t w = 0;
#pragma unroll
        for (int u = 0; u < PU; ++u) w |= classify_raw<VOXEL>(Q.C[u]) << (8 * u);
        w &= Q.gin;                                          // sample(): 0 outside the volume (kernel.cu:99-105)
        if (PU == 4) plane[own + g] = w;
        else ((uint16_t *)(plane + own))[g] = (uint16_t)w;
    };

    // ---- shading of the aligned batch b (entries 4b .. 4b+3, samples i = 4b + k in 1 .. 30) of a ray's open chunk ----
    auto shade_batch = [&](RayT &Q, const uint32_t *cur, const int b) {
        // LDS round 1: the next own dword and the neighbours' dword b
        const uint32_t wnext = cur[own + b + 1];                 // (b == 7: the padding dword, never used)
        const uint32_t wl = cur[Q.nl + b], wr = cur[Q.nr + b], wt = cur[Q.nt + b], wb = cur[Q.nb + b];
        const uint32_t wcur = Q.wcur, wprev = Q.wprev;
        bool valid[4], lit[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = 4 * b + k;
            valid[k] = Q.live && i >= 1 && i <= Q.n;
        }
        // LDS rounds 2 and 3: the quotients of the samples some lane has to light, two samples at a time (12 registers, not 24)
        float direct[4], vd[4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float f_[2], a_[2], l_[2], r_[2], t_[2], b_[2];
            bool any_lit[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int k = 2 * h + j;
                const uint32_t qf = k == 0 ? (wprev >> 24) : ((wcur >> (8 * (k - 1))) & 255u);
                const uint32_t qa = k == 3 ? (wnext & 255u) : ((wcur >> (8 * (k + 1))) & 255u);
                const uint32_t ql = (wl >> (8 * k)) & 255u, qr = (wr >> (8 * k)) & 255u, qt = (wt >> (8 * k)) & 255u, qb = (wb >> (8 * k)) & 255u;
                // all three central differences zero (inside a plateau): the gradient is (0,0,0), it is not normalised (:180) and
                // direct = clamp(0) = 0 -- no division needed
                lit[k] = valid[k] && !(qr == ql && qt == qb && qa == qf);
                any_lit[j] = __builtin_amdgcn_ballot_w64(lit[k]) != 0ull;
                if (any_lit[j]) {
                    f_[j] = q255[qf]; a_[j] = q255[qa]; l_[j] = q255[ql]; r_[j] = q255[qr]; t_[j] = q255[qt]; b_[j] = q255[qb];
                } else { f_[j] = a_[j] = l_[j] = r_[j] = t_[j] = b_[j] = 0.f; }
            }
            {
#pragma clang fp contract(off)
                // the two samples k = 2h, 2h + 1 side by side (packed fp32: same roundings as one by one)
                const int i0 = 4 * b + 2 * h;
                const ph_f2 fi = {(float)i0, (float)(i0 + 1)}, ss = {Q.r.sstep, Q.r.sstep}, dd = {Q.dist, Q.dist};
                const ph_f2 vd2 = fi * ss + dd;                                   // :254
                vd[2 * h] = vd2.x; vd[2 * h + 1] = vd2.y;
                direct[2 * h] = direct[2 * h + 1] = 0.f;
                if (any_lit[0] || any_lit[1]) {
                    const ph_f2 tfx = {P.tan_fov_x, P.tan_fov_x}, tfy = {P.tan_fov_y, P.tan_fov_y}, dz = {Q.dz2, Q.dz2};
                    const ph_f2 dx = tfx * vd2, dy = tfy * vd2;                   // :259-262
                    const ph_f2 R2 = {r_[0], r_[1]}, L2 = {l_[0], l_[1]}, T2 = {t_[0], t_[1]}, B2 = {b_[0], b_[1]}, A2 = {a_[0], a_[1]}, F2 = {f_[0], f_[1]};
                    ph_f2 gx = div_core2(R2 - L2, dx), gy = div_core2(T2 - B2, dy), gz = div_core2(A2 - F2, dz);   // :175-178
                    const ph_f2 one = {1.0f, 1.0f};
                    const ph_f2 inv = div_core2(one, sqrt_core2(gx * gx + gy * gy + gz * gz));
                    const ph_f2 nx = gx * inv, ny = gy * inv, nz = gz * inv;
                    // :180: normalised only where all three components are non-zero
                    const bool n0 = gx.x != 0.f && gy.x != 0.f && gz.x != 0.f, n1 = gx.y != 0.f && gy.y != 0.f && gz.y != 0.f;
                    gx.x = n0 ? nx.x : gx.x; gy.x = n0 ? ny.x : gy.x; gz.x = n0 ? nz.x : gz.x;
                    gx.y = n1 ? nx.y : gx.y; gy.y = n1 ? ny.y : gy.y; gz.y = n1 ? nz.y : gz.y;
                    const ph_f2 m1 = {-1.f, -1.f}, c3 = {0.3f, 0.3f};
                    const ph_f2 d2 = (gx * m1 + gy * m1 + gz * one) * c3;         // :183
                    direct[2 * h]     = lit[2 * h]     ? fmaxf(0.f, fminf(d2.x, 0.3f)) : 0.f;
                    direct[2 * h + 1] = lit[2 * h + 1] ? fmaxf(0.f, fminf(d2.y, 0.3f)) : 0.f;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // LDS round 4: the table entries
        float4 e[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) e[k] = lds_tf[(wcur >> (8 * k)) & 255u];
        // the blend runs in order (:268-274)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma clang fp contract(off)
            const int i = 4 * b + k;
            if (i < 1 || i > 30) continue;
            const bool run = valid[k] && Q.live;                                  // (live may have fallen inside this batch)
            float cr = e[k].x, cg = e[k].y, cb = e[k].z;
            const float ca = e[k].w;
            const bool opaque = ca > kEps;                                        // :164 (phong is on)
            if (INSTR && run) {
                executed++;
                if (bricks) {
                    // the bricks under this sample (position as the refresh computes it: i increments from the chunk origin)
                    float qx = Q.r.origin.x + Q.r.dir.x * Q.dist, qy = Q.r.origin.y + Q.r.dir.y * Q.dist, qz = Q.r.origin.z + Q.r.dir.z * Q.dist;
                    for (int q = 0; q < i; ++q) { qx += Q.r.sdir.x; qy += Q.r.sdir.y; qz += Q.r.sdir.z; }
                    const float ux = __builtin_fmaf(qx - 0.5f, P.inv_scale[0], 0.5f), uy = __builtin_fmaf(qy - 0.5f, P.inv_scale[1], 0.5f),
                                uz = __builtin_fmaf(qz - 0.5f, P.inv_scale[2], 0.5f);
                    if (bounds_check(ux, uy, uz)) mark_bricks(bricks, V, ux, uy, uz);
                }
            }
            // :185-190 for opaque samples; the products below are then multiplied by bf = 0 for the others
            cr = opaque ? cr * 0.7f + direct[k] : cr; cg = opaque ? cg * 0.7f + direct[k] : cg; cb = opaque ? cb * 0.7f + direct[k] : cb;
            if (SLICE == SLICE_PLANE) {
                float vx = Q.r.origin.x + Q.r.dir.x * vd[k], vy = Q.r.origin.y + Q.r.dir.y * vd[k], vz = Q.r.origin.z + Q.r.dir.z * vd[k];
                float d = fabsf(sn.x * (vx - sp.x) + sn.y * (vy - sp.y) + sn.z * (vz - sp.z));
                if (d < .01f) cr = fmaxf(0.f, fminf(cr + (.01f - d) * 100.f, 1.f));
            }
            // blend (:107-118), predicated: with bf == 0 the sums are unchanged bit for bit (the table is finite)
            const float bf = (run && opaque) ? ca * (1.f - Q.res_a) : 0.f;
            Q.res_r = Q.res_r + cr * bf; Q.res_g = Q.res_g + cg * bf; Q.res_b = Q.res_b + cb * bf; Q.res_a = Q.res_a + bf;
            if (run && Q.res_a > P.ert_thr) { if (P.ert_true) Q.ert_done = true; Q.live = false; }   // :272-274
        }
        Q.wprev = wcur; Q.wcur = wnext;
    };
    // opens the chunk at Q.dist for shading out of `plane`; returns the number of batches this wave has to run for the ray
    auto open_chunk = [&](RayT &Q, const uint32_t *plane) -> int {
        const bool mine = Q.marching && !Q.ert_done && Q.dist < Q.r.upper;        // kernel.cu:248
        Q.n = mine ? chunk_count(Q.dist, Q.r.upper, Q.r.sstep) : 0;
        // a ray past the threshold composites sample 1 and breaks again (:272-274) -- as long as its opacity cannot fall back
        // under the threshold; otherwise the per-sample test decides
        if (P.alpha_unit && Q.res_a > P.ert_thr) Q.n = min(Q.n, 1);
        Q.live = Q.n > 0;
        Q.wprev = 0; Q.wcur = plane[own];
        const int nmax = wave_max_i(Q.n);
        return nmax ? nmax / 4 + 1 : 0;                                           // samples 1 .. nmax live in batches 0 .. nmax / 4
    };

    // ---- prologue: chunk 0 into plane 0 (nothing to shade yet), and the depth of chunk 1 ----
    post_need(need_all(false, true), 0);
    post_need(need_all(true, P.max_chunks > 1), 1);
    __syncthreads();
    int depth = need[0];                                     // block-uniform
    if (depth) {
#pragma unroll
        for (int s = 0; s < S; ++s) chunk_origin(R[s], R[s].dist);
        for (int g = 0; g * PU < depth; ++g) {
#pragma unroll
            for (int s = 0; s < S; ++s) issue(R[s]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < S; ++s) land(R[s], cache[0][s], g);
        }
    }
    __syncthreads();

    for (int chunk = 0; depth && chunk < P.max_chunks; ++chunk) {
        // plane chunk & 1 holds this chunk at `depth`; need[(chunk + 1) % 3] is complete (barrier above)
        const int s1 = (chunk + 1) % 3, s2 = (chunk + 2) % 3, s0 = chunk % 3;
        const int depth_next = need[s1];
        if (tid == 0) need[s0] = 0;                          // read by everyone one barrier ago, posted to again after the next one
        const int pc = chunk & 1, pn = pc ^ 1;
        int nbat[S], bdone[S];
#pragma unroll
        for (int s = 0; s < S; ++s) { nbat[s] = open_chunk(R[s], cache[pc][s]); bdone[s] = 0; }
        if (depth_next) {
#pragma unroll
            for (int s = 0; s < S; ++s) chunk_origin(R[s], R[s].dnext);
            const int ng = (depth_next + PU - 1) / PU;       // steps: every ray issues PU samples per step
            for (int g0 = 0; g0 < ng; g0 += S) {
#pragma unroll
                for (int t = 0; t < S; ++t) {
                    const int g = g0 + t;
                    if (g < ng) {
#pragma unroll
                        for (int s = 0; s < S; ++s) issue(R[s]);
                        __builtin_amdgcn_sched_barrier(0);   // the step's 16 gathers are in flight ...
                        if (bdone[t] < nbat[t]) { shade_batch(R[t], cache[pc][t], bdone[t]); ++bdone[t]; }   // ... across four samples of shading
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int s = 0; s < S; ++s) land(R[s], cache[pn][s], g);
                    }
                }
            }
        }
        // what is left of this chunk's inner loops
#pragma unroll
        for (int s = 0; s < S; ++s)
            for (; bdone[s] < nbat[s]; ++bdone[s]) shade_batch(R[s], cache[pc][s], bdone[s]);
#pragma unroll
        for (int s = 0; s < S; ++s) {
#pragma clang fp contract(off)
            R[s].dist = R[s].dnext;                          // dist += 30 scaledStep (:277), accumulated in the same order
            R[s].dnext = R[s].dist + R[s].s30;
        }
        post_need(need_all(true, chunk + 2 < P.max_chunks), s2);
        __syncthreads();
        depth = depth_next;
    }

#pragma unroll
    for (int s = 0; s < S; ++s)
        if (R[s].writer)
            pixels[(size_t)y * P.W + R[s].x] = R[s].skip ? 0u : pack_rgba(R[s].res_r, R[s].res_g, R[s].res_b, R[s].res_a);
    if (INSTR) {
        for (int o = 32; o > 0; o >>= 1) executed += __shfl_down(executed, o);
        if ((threadIdx.x & 63) == 0 && executed) atomicAdd(counter, executed);
        if (kLayout == LAYOUT_BRICKED && (threadIdx.x & 63) == 0 && executed) atomicAdd(counter + 2, 1ull);
        if (kLayout == LAYOUT_ZPAIR && (threadIdx.x & 63) == 0 && executed) atomicAdd(counter + 3, 1ull);
    }
}
