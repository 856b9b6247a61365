// vv_raymarch_phong_pair.h -- march_phong_pair_kernel: march_phong_kernel with TWO x-adjacent slabs per block.
// Included by vv_raymarch.hip inside namespace vv::<layout>, once per volume layout.
//
// Same phases, same arithmetic, same cache layout per slab as march_phong_kernel (kernel.cu:125-145 rayMarch, :147-201 shadeVoxel,
// :248-278 the chunk loop; pins 5 and 6); every thread owns the ray at its place in BOTH slabs' 16 x 16 footprints.  Why: at its
// shipped occupancy the Phong march is bound by the bytes it moves (profiles/r04_phong_forms.txt: 9.9 GB at 5.4 TB/s for 3.7 GB of
// work), and most of the excess is 128-byte lines of which a 16-pixel slab row uses ~24 voxels: the two rays of a thread are 14
// pixels apart, so the second slab's gathers -- issued right behind the first's, sample by sample -- fall into lines the first
// just asked for.  The phases stay separate (gathers in flight and shading temporaries never coexist), so the second ray costs
// ~25 registers, not a second wave: the kernel keeps three blocks (six slabs) per CU.
// Each slab keeps its own rad (:329), apron and cache (:167-173); the refresh depth of a chunk is the deeper of the two slabs' needs.
#pragma once

template <int SLICE, int VOXEL, bool TEX8, bool INSTR>
__global__ __launch_bounds__(256) void march_phong_pair_kernel(FrameParams P, VolumeView V,
                                                               const float4 *__restrict__ tf, SlabMap M,
                                                               uint32_t *__restrict__ pixels,
                                                               unsigned long long *__restrict__ counter,
                                                               uint32_t *__restrict__ bricks)
{
    constexpr int S = 2;
    __shared__ float4 lds_tf[256];
    __shared__ uint8_t cache[S][kCacheDepth][256];
    __shared__ float q255[256];              // q / 255.f for every byte q, by the same IEEE division
    __shared__ int any_live;
    float *red = (float *)&cache[0][0][0];   // the rad reductions borrow the caches (dead until the first refresh)
    const int tid = threadIdx.x;
    // XCD-aware order (speed only, as in march_kernel): linear block L runs on XCD L % 8; XCD k takes the
    // grid rows k, k+8, ... so that the slabs of one row, which share volume lines, share an L2
    const int nbxg = (P.nbx + S - 1) / S;
    const int j_ = (int)blockIdx.x >> 3;
    const int gx = j_ % nbxg, gy = (j_ / nbxg) * 8 + ((int)blockIdx.x & 7);
    if (gy > M.n_regular) return;                              // block-uniform, before any barrier
    stage_tf(lds_tf, tf);
    q255[tid] = (float)tid / 255.f;                      // visible after the barriers of the reduction below

    int by;
    if (gy == M.n_regular) { if (!P.conflict_y) return; by = P.nby - 1; }
    else by = M.r0 + (gy / M.band) * M.band_stride + (gy % M.band);
    if (by >= P.nby) return;                                   // block-uniform
    if (gy != M.n_regular && P.conflict_y && by == P.nby - 1) return;
    {
        int yrow = (P.conflict_y && by == P.nby - 1) ? P.H - 2 : by * kSlab;
        if (yrow > (P.H >= 2 ? P.H - 2 : 0) || !row_owned(P, yrow)) return;   // block-uniform
    }
    const int tx = tid & 15, ty = tid >> 4;
    const bool border = tx == 0 || ty == 0 || tx == 15 || ty == 15;              // kernel.cu:304-305
    const int loy = slab_lo(by), upy = slab_up(by, P.H);
    int y = by * kSlab + ty - 1;
    y = max(loy, min(y, upy - 1));
    const bool has1 = gx * S + 1 < P.nbx;                      // the block's second slab exists (block-uniform); else it shadows the first and writes nothing

    struct PRay { Ray r; float res_r, res_g, res_b, res_a, dist; int nl, nr, nt, nb, x; bool marching, ert_done, writer, skip, mine; };
    PRay R[S];
    float cl[S];
    f3 front[S], back[S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int bx = gx * S + ((s == 0 || has1) ? s : 0);
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
        PRay &Q = R[s];
        const bool present = s == 0 || has1;
        const int bx = gx * S + (present ? s : 0);
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
        // neighbour thread indices, clamped to the footprint
        if (degenerate) { Q.nl = Q.nr = Q.nt = Q.nb = tid; }
        else {
            int xl = max(lox, min(x - 1, upx - 1)), xr = max(lox, min(x + 1, upx - 1));
            int yt = max(loy, min(y + 1, upy - 1)), yb = max(loy, min(y - 1, upy - 1));
            int fx0 = bx * kSlab - 1, fy0 = by * kSlab - 1;
            Q.nl = (y - fy0) * 16 + (xl - fx0); Q.nr = (y - fy0) * 16 + (xr - fx0);
            Q.nt = (yt - fy0) * 16 + (x - fx0); Q.nb = (yb - fy0) * 16 + (x - fx0);
        }
        Q.res_r = Q.res_g = Q.res_b = Q.res_a = 0.f;
        Q.dist = Q.r.dist0;
        Q.ert_done = false; Q.mine = false;
        Q.marching = writer && !Q.skip && !Q.r.cut_return;
    }
    __syncthreads();                                          // red[] (= the caches) has been read by everyone
    unsigned long long executed = 0;
    const f3 sp = mk3(P.slice_point[0], P.slice_point[1], P.slice_point[2]);
    const f3 sn = mk3(P.slice_normal[0], P.slice_normal[1], P.slice_normal[2]);

    for (int chunk = 0; chunk < P.max_chunks; ++chunk) {
        // how deep this chunk's caches have to be: the deepest need of the rays of both slabs (march_phong_kernel explains the rule)
        int depth = kCacheDepth;
        {
            int d = 0;
#pragma unroll
            for (int s = 0; s < S; ++s) {
                PRay &Q = R[s];
                Q.mine = Q.marching && !Q.ert_done && Q.dist < Q.r.upper;
                int ds = 0;
                if (Q.mine) {
#pragma clang fp contract(off)
                    if (P.alpha_unit && Q.res_a > P.ert_thr) ds = 3;
                    else if (!(30.f * Q.r.sstep + Q.dist > Q.r.upper)) ds = kCacheDepth;
                    else ds = chunk_count(Q.dist, Q.r.upper, Q.r.sstep) + 2;
                }
                d = max(d, ds);
            }
            if (threadIdx.x == 0) any_live = 0;
            __syncthreads();
            {
                const bool full = __builtin_amdgcn_ballot_w64(d == kCacheDepth) != 0ull, deep = __builtin_amdgcn_ballot_w64(d > 3) != 0ull;
                if (full) d = kCacheDepth;
                else if (!deep) d = __builtin_amdgcn_ballot_w64(d != 0) != 0ull ? 3 : 0;
                else d = wave_max_i(d);
            }
            if ((threadIdx.x & 63) == 0 && d) atomicMax(&any_live, d);
            __syncthreads();
            depth = any_live;
            if (!depth) break;
        }
        // rayMarch (:125-145) for both slabs, sample group by sample group: the second slab's gathers follow the first's at once
        {
            float px[S], py[S], pz[S];
#pragma unroll
            for (int s = 0; s < S; ++s) {
#pragma clang fp contract(off)
                const PRay &Q = R[s];
                px[s] = Q.r.origin.x + Q.r.dir.x * Q.dist; py[s] = Q.r.origin.y + Q.r.dir.y * Q.dist; pz[s] = Q.r.origin.z + Q.r.dir.z * Q.dist;
            }
#ifndef VV_PHONG_PAIR_PU
#define VV_PHONG_PAIR_PU 2
#endif
            constexpr int PU = VV_PHONG_PAIR_PU;                  // samples per slab in flight per trip: 2 x PU x 4 gathers per lane
            auto refresh = [&](const int i0) {
                float tx_[S][PU], ty_[S][PU], tz_[S][PU];
                typename CornerSel<VOXEL>::type C[S][PU];
#pragma unroll
                for (int u = 0; u < PU; ++u) {
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        tx_[s][u] = __builtin_fmaf(px[s] - 0.5f, P.inv_scale[0], 0.5f);
                        ty_[s][u] = __builtin_fmaf(py[s] - 0.5f, P.inv_scale[1], 0.5f);
                        tz_[s][u] = __builtin_fmaf(pz[s] - 0.5f, P.inv_scale[2], 0.5f);
                        fetch_any<VOXEL, TEX8>(V, tx_[s][u], ty_[s][u], tz_[s][u], C[s][u]);
                        px[s] += R[s].r.sdir.x; py[s] += R[s].r.sdir.y; pz[s] += R[s].r.sdir.z;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < PU; ++u) {
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        const int i = i0 + u;
                        cache[s][i][tid] = (uint8_t)classify_index<VOXEL>(C[s][u], tx_[s][u], ty_[s][u], tz_[s][u]);
                        if (INSTR && bricks && R[s].mine && i >= 1 && i <= 30 && bounds_check(tx_[s][u], ty_[s][u], tz_[s][u])) mark_bricks(bricks, V, tx_[s][u], ty_[s][u], tz_[s][u]);
                    }
                }
            };
            // the full depth keeps its compile-time trip count
            if (depth == kCacheDepth) { for (int j0 = 0; j0 < kCacheDepth; j0 += PU) refresh(j0); }
            else { for (int j0 = 0; j0 < depth; j0 += PU) refresh(j0); }
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < S; ++s) {
            PRay &Q = R[s];
            if (Q.mine) {
                for (int i = 1; i < kCacheDepth - 1; ++i) {
#pragma clang fp contract(off)
                    float vd = (float)i * Q.r.sstep + Q.dist;                             // :254
                    if (vd > Q.r.upper) break;
                    if (INSTR) executed++;
                    uint32_t sv = cache[s][i][tid];
                    float4 e = lds_tf[sv];
                    float cr = e.x, cg = e.y, cb = e.z, ca = e.w;
                    if (ca > kEps) {                                                      // :164 (phong is on)
                        const uint32_t qf = cache[s][i - 1][tid], qa = cache[s][i + 1][tid];
                        const uint32_t ql = cache[s][i][Q.nl], qr = cache[s][i][Q.nr], qt = cache[s][i][Q.nt], qb = cache[s][i][Q.nb];
                        float direct = 0.f;
                        // all three central differences zero (inside a plateau): the gradient is (0,0,0),
                        // it is not normalised (:180) and direct = clamp(0) = 0 -- skip the divisions
                        if (!(qr == ql && qt == qb && qa == qf)) {
                            float f = q255[qf], a = q255[qa], l = q255[ql], rr = q255[qr], t = q255[qt], b = q255[qb];
                            float gx_ = (rr - l) / (P.tan_fov_x * vd), gy_ = (t - b) / (P.tan_fov_y * vd),
                                  gz_ = (a - f) / (Q.r.sstep * 2.f);                      // :175-178, :259-263
                            if (gx_ != 0.f && gy_ != 0.f && gz_ != 0.f) {
                                float inv = 1.0f / sqrtf(gx_ * gx_ + gy_ * gy_ + gz_ * gz_);
                                gx_ *= inv; gy_ *= inv; gz_ *= inv;
                            }
                            direct = (gx_ * -1.f + gy_ * -1.f + gz_ * 1.f) * 0.3f;        // :183
                            direct = fmaxf(0.f, fminf(direct, 0.3f));
                        }
                        cr = cr * 0.7f + direct; cg = cg * 0.7f + direct; cb = cb * 0.7f + direct;
                    }
                    if (SLICE == SLICE_PLANE) {
                        float vx = Q.r.origin.x + Q.r.dir.x * vd, vy = Q.r.origin.y + Q.r.dir.y * vd, vz = Q.r.origin.z + Q.r.dir.z * vd;
                        float d = fabsf(sn.x * (vx - sp.x) + sn.y * (vy - sp.y) + sn.z * (vz - sp.z));
                        if (d < .01f) cr = fmaxf(0.f, fminf(cr + (.01f - d) * 100.f, 1.f));
                    }
                    if (ca > kEps) {
                        float bf = ca * (1.f - Q.res_a);
                        Q.res_r = Q.res_r + cr * bf; Q.res_g = Q.res_g + cg * bf; Q.res_b = Q.res_b + cb * bf; Q.res_a = Q.res_a + bf;
                    }
                    if (Q.res_a > P.ert_thr) { if (P.ert_true) Q.ert_done = true; break; }
                }
            }
            {
#pragma clang fp contract(off)
                Q.dist += Q.r.sstep * kChunkSteps;
            }
        }
        __syncthreads();      // the caches are rewritten next iteration
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
