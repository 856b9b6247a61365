// vv_raymarch_phong2.h -- march_phong2_kernel: the Phong-shaded march, second form.
// Included by vv_raymarch.hip inside namespace vv::<layout>, once per volume layout.
//
// Same semantics as march_phong_kernel (kernel.cu:125-145 rayMarch, :147-201 shadeVoxel, :248-278 the chunk loop; a reference
// slab + apron per 256 threads, pins 5 and 6) and the same two phases per chunk -- every thread gathers its cache entries,
// barrier, the interior threads shade, barrier -- because what this kernel needs most on MI355X is resident waves (a form
// that overlapped the two phases inside a thread through a second cache plane was built first: 170-240 registers, two waves
// per SIMD, slower on every workload but one; profiles/r04_phong_forms.txt).  What differs from the first kernel:
//   * a thread's 32 entries are contiguous (9 dwords per thread: 32 bytes + 4 of padding, so rows start on all 32 banks) and
//     samples are shaded in batches of four aligned entries: one dword of the own row and one of each neighbour's row serve
//     four samples, the quotients of two samples are fetched together -- two LDS round trips per pair of samples instead of
//     four per sample -- and the arithmetic is predicated instead of branched per lane (wave-uniform skips remain);
//   * the gradient of two samples is computed side by side in packed fp32 (v_pk_fma_f32 ...: two results per lane and cycle,
//     each half rounded like the scalar instruction), with divisions and the square root reduced to their cores (below);
//   * the refresh depth of chunk c + 1 is posted while chunk c is shaded (exact, as in the first kernel) and read behind the
//     barrier that ends the shading: two barriers per chunk instead of four;
//   * once every ray of the block that still marches is past the ERT threshold -- each composites sample 1 of every later
//     chunk (pin 4) and reads entries 0 .. 2 of it -- up to eight such chunks are taken between two barriers: chunk j's four
//     entries live in dword j of the row;
//   * S = 2: a block of 512 threads marches two x-adjacent slabs, wave w of the second slab sharing its SIMD and its moment
//     with wave w of the first: a 16-pixel slab row spans about 24 voxels = 1.75 cache lines on the 1024^3 volume at 1080p,
//     two rows side by side 45 voxels = 2.4 lines instead of 3.5.  Each slab keeps its own rad, apron and cache (:329, :167-173).
// Entries nobody reads are not observable; frames are bit-identical (tests/test_gpu_parity.py runs every Phong case through both).
//
// The kernel serves frames whose shading divisions need no range handling (FrameParams::safe_div); vv_render gives the
// others to march_phong_kernel.
#pragma once

// x / d for the three screen-space differences and 1 / sqrt(x) for the normalisation, IEEE-exact.  The compiler's expansion of
// `/` and sqrtf is a fixed core (v_rcp + one Newton step, quotient + two residual corrections; v_sqrt + a test of the two
// neighbouring floats) wrapped in range handling (v_div_scale x 2, v_div_fixup; a 2^32 pre-scale and a class test): 11 and 14
// instructions.  With operands known to be normal and far from the ends of the range -- decided per frame on the host
// (FrameParams::safe_div: pixel tangents in [2^-24, 2^8], steps <= 16; numerators are 0 or differences of q / 255) -- the
// wrappers do nothing and the cores alone give the same bits.  Two samples at a time: 7 packed instructions + 2 v_rcp for a
// pair of quotients instead of 22.
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

constexpr int kPhRow = 9;                    // dwords per thread in the cache
constexpr int kPhTail = 8;                   // tail chunks (four entries each) taken between two barriers

#ifndef VV_PHONG2_WAVES
#define VV_PHONG2_WAVES 4
#endif
template <int SLICE, int VOXEL, bool TEX8, bool INSTR, int S>
__global__ __launch_bounds__(256 * S) __attribute__((amdgpu_waves_per_eu(VV_PHONG2_WAVES))) void march_phong2_kernel(FrameParams P, VolumeView V,
                                                           const float4 *__restrict__ tf, SlabMap M,
                                                           uint32_t *__restrict__ pixels,
                                                           unsigned long long *__restrict__ counter,
                                                           uint32_t *__restrict__ bricks)
{
#ifndef VV_PHONG2_PU
#define VV_PHONG2_PU 4
#endif
    constexpr int PU = VV_PHONG2_PU;                         // samples (4 gathers each) in flight per thread during the refresh (a divisor of 4 entries per dword: 4 or 8)
    constexpr int ROW = kPhRow;
    static_assert(S == 1 || S == 2, "one or two slabs per block");
    __shared__ float4 lds_tf[256];
    __shared__ uint32_t cache_[S][256 * ROW];
    __shared__ float q255[256];              // q / 255.f for every byte q, by the same IEEE division
    __shared__ int need[2];                  // refresh depth of chunk c: need[c & 1]
    const int slab = S == 1 ? 0 : (int)(threadIdx.x >> 8);   // (wave-uniform)
    const int tid = threadIdx.x & 255;                       // the thread's place in its slab's 16 x 16 footprint
    uint32_t *cache = cache_[slab];
    float *red = (float *)cache;                             // the rad reduction borrows the cache (dead until the first refresh)
    // XCD-aware order (speed only, as in march_kernel): linear block L runs on XCD L % 8; XCD k takes the
    // grid rows k, k+8, ... so that the slabs of one row, which share volume lines, share an L2
    const int nbxg = (P.nbx + S - 1) / S;
    const int j_ = (int)blockIdx.x >> 3;
    const int gx_ = j_ % nbxg, gy_ = (j_ / nbxg) * 8 + ((int)blockIdx.x & 7);
    if (gy_ > M.n_regular) return;                             // block-uniform, before any barrier
    for (int i = threadIdx.x; i < 256; i += 256 * S) { lds_tf[i] = tf[i]; q255[i] = (float)i / 255.f; }
    if (threadIdx.x < 2) need[threadIdx.x] = 0;                // (visible after the barriers of the reduction below)

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
    // the block's second slab may not exist (odd nbx): its threads shadow the first slab's rays and write nothing
    const bool present = slab == 0 || gx_ * S + 1 < P.nbx;
    const int bx = gx_ * S + (present ? slab : 0);
    const int tx = tid & 15, ty = tid >> 4;
    const int own = tid * ROW;
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
    for (int w = 128; w > 0; w >>= 1) {
        if (tid < w) red[tid] = fminf(red[tid], red[tid + w]);
        __syncthreads();
    }
    float rd = degenerate ? cl : red[0];
    float length = vlen3(back.x - front.x, back.y - front.y, back.z - front.z);
    const bool skip = length < 0.001f && !border;                                 // :334
    Ray r;
    setup_ray(P, front, back, rd, r);

    // write ownership (pin 10) and one writer per pixel
    const int ox = owner_slab(x, P.W, P.nbx, P.conflict_x), oy = owner_slab(y, P.H, P.nby, P.conflict_y);
    bool writer = present && !border && ox == bx && oy == by;
    {
        int ux = bx * kSlab + tx - 1, uy = by * kSlab + ty - 1;
        bool xrep = (ux == x) || (bx * kSlab > x && tx == 1);
        bool yrep = (uy == y) || (by * kSlab > y && ty == 1);
        writer = writer && xrep && yrep;
    }
    // rows (dword offsets into the cache) of the neighbour threads, clamped to the footprint (pin 6)
    int nl, nr, nt, nb;
    if (degenerate) { nl = nr = nt = nb = tid; }
    else {
        int xl = max(lox, min(x - 1, upx - 1)), xr = max(lox, min(x + 1, upx - 1));
        int yt = max(loy, min(y + 1, upy - 1)), yb = max(loy, min(y - 1, upy - 1));
        int fx0 = bx * kSlab - 1, fy0 = by * kSlab - 1;
        nl = (y - fy0) * 16 + (xl - fx0); nr = (y - fy0) * 16 + (xr - fx0);
        nt = (yt - fy0) * 16 + (x - fx0); nb = (yb - fy0) * 16 + (x - fx0);
    }
    nl *= ROW; nr *= ROW; nt *= ROW; nb *= ROW;

    float res_r = 0.f, res_g = 0.f, res_b = 0.f, res_a = 0.f;
    unsigned long long executed = 0;
    float dist = r.dist0;
    bool ert_done = false;
    const bool marching = writer && !skip && !r.cut_return;
    const f3 sp = mk3(P.slice_point[0], P.slice_point[1], P.slice_point[2]);
    const f3 sn = mk3(P.slice_normal[0], P.slice_normal[1], P.slice_normal[2]);
    float s30, dz2;
    {
#pragma clang fp contract(off)
        s30 = r.sstep * kChunkSteps;                                              // :277
        dz2 = r.sstep * 2.f;                                                      // :263
    }

    // Cache depth this ray needs in the chunk that starts at distance D: a compositing ray reads its entries 0 .. n+1 and its
    // neighbours' 1 .. n, n = the samples of the chunk before `vd > upper` (:254); a ray past the ERT threshold (it keeps
    // compositing one sample per chunk, pin 4) reads entries 0 .. 2 -- with every table opacity in [0, 1] accumulated opacity
    // cannot fall back under the threshold.
    auto need_at = [&](float D) -> int {
#pragma clang fp contract(off)
        if (!(marching && !ert_done && D < r.upper)) return 0;
        if (P.alpha_unit && res_a > P.ert_thr) return 3;
        if (!(30.f * r.sstep + D > r.upper)) return kCacheDepth;
        return chunk_count(D, r.upper, r.sstep) + 2;
    };
    // block-wide maximum into need[slot]: nearly always decided by two ballots per wave
    auto post_need = [&](int d, int slot) {
        const bool full = __builtin_amdgcn_ballot_w64(d == kCacheDepth) != 0ull, deep = __builtin_amdgcn_ballot_w64(d > 3) != 0ull;
        if (full) d = kCacheDepth;
        else if (!deep) d = __builtin_amdgcn_ballot_w64(d != 0) != 0ull ? 3 : 0;
        else d = wave_max_i(d);
        if ((threadIdx.x & 63) == 0 && d) atomicMax(&need[slot], d);
    };

    // ---- refresh: PU samples (4 PU gathers) in flight, entries j0 .. j0 + 3 of the row packed into dword j0 / 4 ----
    float px, py, pz;
    auto refresh4 = [&](const int dw) {
        uint32_t inb = 0;
        typename CornerSel<VOXEL>::type C[PU];
#pragma unroll
        for (int u = 0; u < PU; ++u) {
            const float tx_ = __builtin_fmaf(px - 0.5f, P.inv_scale[0], 0.5f);
            const float ty_ = __builtin_fmaf(py - 0.5f, P.inv_scale[1], 0.5f);
            const float tz_ = __builtin_fmaf(pz - 0.5f, P.inv_scale[2], 0.5f);
            fetch_any<VOXEL, TEX8>(V, tx_, ty_, tz_, C[u]);
            inb |= bounds_check(tx_, ty_, tz_) ? (0xffu << (8 * u)) : 0u;         // sample(): 0 outside the volume (kernel.cu:65-71, :99-105)
            px += r.sdir.x; py += r.sdir.y; pz += r.sdir.z;                       // :141
        }
        __builtin_amdgcn_sched_barrier(0);
        uint32_t w = 0;
#pragma unroll
        for (int u = 0; u < PU; ++u) w |= classify_raw<VOXEL>(C[u]) << (8 * u);
        cache[own + dw] = w & inb;
    };
    auto chunk_origin = [&](float D) {
#pragma clang fp contract(off)
        px = r.origin.x + r.dir.x * D; py = r.origin.y + r.dir.y * D; pz = r.origin.z + r.dir.z * D;   // :249 / :131
    };

    // ---- shading of the aligned batch in dword b of the rows: samples i0 + k, k = 0 .. 3, of the chunk at distance D; sample
    //      i0 + k runs if bit k of `vmask` is set and the ray is still inside the chunk's inner loop (`live`) ----
    bool live = false;
    auto shade_batch = [&](const float D, const int b, const int i0, const uint32_t vmask, const uint32_t wprev, const uint32_t wcur, const uint32_t wnext) {
        const uint32_t wl = cache[nl + b], wr = cache[nr + b], wt = cache[nt + b], wb = cache[nb + b];
        bool valid[4], lit[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) valid[k] = live && ((vmask >> k) & 1u);
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
                const ph_f2 fi = {(float)(i0 + 2 * h), (float)(i0 + 2 * h + 1)}, ss = {r.sstep, r.sstep}, dd = {D, D};
                const ph_f2 vd2 = fi * ss + dd;                                   // :254
                vd[2 * h] = vd2.x; vd[2 * h + 1] = vd2.y;
                direct[2 * h] = direct[2 * h + 1] = 0.f;
                if (any_lit[0] || any_lit[1]) {
                    const ph_f2 tfx = {P.tan_fov_x, P.tan_fov_x}, tfy = {P.tan_fov_y, P.tan_fov_y}, dz = {dz2, dz2};
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
        // the table entries, then the blend in order (:268-274)
        float4 e[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) e[k] = lds_tf[(wcur >> (8 * k)) & 255u];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma clang fp contract(off)
            const bool run = valid[k] && live;                                    // (live may have fallen inside this batch)
            float cr = e[k].x, cg = e[k].y, cb = e[k].z;
            const float ca = e[k].w;
            const bool opaque = ca > kEps;                                        // :164 (phong is on)
            if (INSTR && run) {
                executed++;
                if (bricks) {
                    // the bricks under this sample (position as the refresh computes it: increments from the chunk origin)
                    float qx = r.origin.x + r.dir.x * D, qy = r.origin.y + r.dir.y * D, qz = r.origin.z + r.dir.z * D;
                    for (int q = 0; q < i0 + k; ++q) { qx += r.sdir.x; qy += r.sdir.y; qz += r.sdir.z; }
                    const float ux = __builtin_fmaf(qx - 0.5f, P.inv_scale[0], 0.5f), uy = __builtin_fmaf(qy - 0.5f, P.inv_scale[1], 0.5f),
                                uz = __builtin_fmaf(qz - 0.5f, P.inv_scale[2], 0.5f);
                    if (bounds_check(ux, uy, uz)) mark_bricks(bricks, V, ux, uy, uz);
                }
            }
            // :185-190 for opaque samples; the products below are then multiplied by bf = 0 for the others
            cr = opaque ? cr * 0.7f + direct[k] : cr; cg = opaque ? cg * 0.7f + direct[k] : cg; cb = opaque ? cb * 0.7f + direct[k] : cb;
            if (SLICE == SLICE_PLANE) {
                float vx = r.origin.x + r.dir.x * vd[k], vy = r.origin.y + r.dir.y * vd[k], vz = r.origin.z + r.dir.z * vd[k];
                float d = fabsf(sn.x * (vx - sp.x) + sn.y * (vy - sp.y) + sn.z * (vz - sp.z));
                if (d < .01f) cr = fmaxf(0.f, fminf(cr + (.01f - d) * 100.f, 1.f));
            }
            // blend (:107-118), predicated: with bf == 0 the sums are unchanged bit for bit (the table is finite)
            const float bf = (run && opaque) ? ca * (1.f - res_a) : 0.f;
            res_r = res_r + cr * bf; res_g = res_g + cg * bf; res_b = res_b + cb * bf; res_a = res_a + bf;
            if (run && res_a > P.ert_thr) { if (P.ert_true) ert_done = true; live = false; }   // :272-274
        }
    };

    // ---- chunk loop ----
    post_need(need_at(dist), 0);
    __syncthreads();
    for (int chunk = 0, it = 0; chunk < P.max_chunks; ++it) {
        const int depth = need[it & 1];                      // block-uniform; complete (barrier above)
        if (!depth) break;
        if (threadIdx.x == 0) need[(it + 1) & 1] = 0;        // posted to behind the next barrier, read behind the one after
        // Tail: every ray of the block that still marches composites at most sample 1 of this chunk, and of every later one
        // (rays past the threshold, pin 4; rays in a last chunk of one sample end with it): kPhTail chunks between two barriers.
        const bool tail = depth <= 3;
        const int T = tail ? min(kPhTail, P.max_chunks - chunk) : 1;
        if (tail) {
            float D = dist;
            for (int j = 0; j < T; ++j) {
                chunk_origin(D);
                refresh4(j);                                 // entries 0 .. 3 of chunk + j in dword j
                {
#pragma clang fp contract(off)
                    D = D + s30;
                }
            }
        } else {
            chunk_origin(dist);
            // the full depth keeps its compile-time trip count
            if (depth == kCacheDepth) { for (int g = 0; g < kCacheDepth / PU; ++g) refresh4(g); }
            else { for (int g = 0; g * PU < depth; ++g) refresh4(g); }
        }
        __syncthreads();
        if (tail) {
            for (int j = 0; j < T; ++j) {
                const bool mine = marching && !ert_done && dist < r.upper;        // kernel.cu:248
                const int n = mine ? chunk_count(dist, r.upper, r.sstep) : 0;
                live = n > 0;
                if (__builtin_amdgcn_ballot_w64(live) != 0ull) {
                    const uint32_t w = cache[own + j];
                    shade_batch(dist, j, 0, 2u, 0u, w, 0u);  // sample 1 alone: entries 0, 1, 2 of the own row, entry 1 of the neighbours'
                }
                {
#pragma clang fp contract(off)
                    dist = dist + s30;                       // :277
                }
            }
        } else {
            const bool mine = marching && !ert_done && dist < r.upper;            // kernel.cu:248
            int n = mine ? chunk_count(dist, r.upper, r.sstep) : 0;               // samples this lane runs unless it terminates inside (:253-257)
            // a ray past the threshold composites sample 1 and breaks again (:272-274) -- as long as its opacity cannot fall back
            // under the threshold; otherwise the per-sample test decides
            if (P.alpha_unit && res_a > P.ert_thr) n = min(n, 1);
            live = n > 0;
            const int nmax = wave_max_i(n);
            if (nmax) {
                uint32_t wprev = 0, wcur = cache[own];
                for (int b = 0; b <= nmax / 4; ++b) {        // samples 1 .. nmax live in batches 0 .. nmax / 4
                    const uint32_t wnext = cache[own + b + 1];                    // (b == 7: the padding dword, never used)
                    const int lo = 4 * b;
                    const uint32_t vmask = ((n >= lo ? (n - lo >= 3 ? 15u : ((2u << (n - lo)) - 1u)) : 0u)) & (b == 0 ? 14u : 15u) & (b == 7 ? 7u : 15u);
                    shade_batch(dist, b, lo, vmask, wprev, wcur, wnext);
                    wprev = wcur; wcur = wnext;
                }
            }
            {
#pragma clang fp contract(off)
                dist = dist + s30;                           // :277
            }
        }
        chunk += T;
        post_need(chunk < P.max_chunks ? need_at(dist) : 0, (it + 1) & 1);
        __syncthreads();
    }

    if (writer)
        pixels[(size_t)y * P.W + x] = skip ? 0u : pack_rgba(res_r, res_g, res_b, res_a);
    if (INSTR) {
        for (int o = 32; o > 0; o >>= 1) executed += __shfl_down(executed, o);
        if ((threadIdx.x & 63) == 0 && executed) atomicAdd(counter, executed);
        if (kLayout == LAYOUT_BRICKED && (threadIdx.x & 63) == 0 && executed) atomicAdd(counter + 2, 1ull);
        if (kLayout == LAYOUT_ZPAIR && (threadIdx.x & 63) == 0 && executed) atomicAdd(counter + 3, 1ull);
    }
}
