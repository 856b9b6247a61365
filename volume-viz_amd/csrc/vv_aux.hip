// vv_aux.hip -- slice-view sampler, procedural generator and volume staging kernels
// for gfx950.
//
//   slice_kernel          kernel.cu:543-597 (5-arg), slicekernel.cu:51-82 (legacy 4-arg)
//   advanced_slice_kernel kernel.cu:599-644
//   ellipsoid_kernel      VolumeGenerator::drawEllipsoid, volumegenerator.cpp:31-97,
//                         n ellipsoids fused into one pass over the volume
#include <atomic>
#include "vv_device.h"
#include "vv_kernels.h"
#include <cstdlib>

namespace vv {

// Lanes of ONE wave handing data to each other through LDS (ellipsoid_rows_kernel): program order between the store and the other lanes' load,
// stated to the compiler as a wave-scope release / acquire pair around a wave barrier.  No instruction is emitted for it on gfx950 (LDS operations of
// a wave are issued and completed in order); it only forbids the compiler to reorder or forward across it.
__device__ __forceinline__ void vv_wave_lds_order()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}


// ---------------------------------------------------------------------------
// slice view
// ---------------------------------------------------------------------------
template <int VOXEL, bool TEX8>
__global__ __launch_bounds__(256) void slice_kernel(SliceArgs A)
{
#pragma clang fp contract(off)
    const size_t i = threadIdx.x + (size_t)blockIdx.x * 16;      // 16x16 threads, as the reference
    const size_t j = threadIdx.y + (size_t)blockIdx.y * 16;
    if (!(j < A.height && i < A.width)) return;                   // kernel.cu:552
    const size_t offset = j * A.height + i;                       // kernel.cu:550,604 (height as stride)
    if (offset >= A.height * A.width) return;                     // the reference would write out of bounds
    // width > height: (j,i) and (j+1,i-height) share an offset (a write race in the
    // reference).  Pinned like the oracle's loop order: the larger j wins.
    if (i >= A.height && j + 1 < A.height) return;
    const float u = ((float)i) / ((float)A.width), w = ((float)j) / ((float)A.height);
    const float ix = 1.0f / A.scale[0], iy = 1.0f / A.scale[1], iz = 1.0f / A.scale[2];
    float px, py, pz;
    bool check = true;
    if (A.advanced) {
        const float rz = 0.5f, rw = 1.f;                          // kernel.cu:608-618, row-major matrix
        px = A.trans[0] * u + A.trans[1] * w + A.trans[2]  * rz + A.trans[3]  * rw;
        py = A.trans[4] * u + A.trans[5] * w + A.trans[6]  * rz + A.trans[7]  * rw;
        pz = A.trans[8] * u + A.trans[9] * w + A.trans[10] * rz + A.trans[11] * rw;
        px *= ix; py *= iy; pz *= iz;                             // :620-622 (reciprocal, DESIGN.md pin 3)
        px = __builtin_fmaf(px - 0.5f, ix, 0.5f);                 // :624 (scaled a second time)
        py = __builtin_fmaf(py - 0.5f, iy, 0.5f);
        pz = __builtin_fmaf(pz - 0.5f, iz, 0.5f);
    } else {
        px = 0.f; py = 0.f; pz = 0.f;
        if (A.legacy) { px = u; py = w; pz = 0.f; }               // slicekernel.cu:62-64
        else switch (A.orientation) {                             // kernel.cu:559-579
            case VV_SAGITTAL:   pz += 0.f; py += w;   px += u;   break;
            case VV_HORIZONTAL: pz += u;   py += 0.f; px += w;   break;
            case VV_CORONAL:    pz += u;   py += w;   px += 0.f; break;
            default: break;
        }
        px += A.dx; py += A.dy; pz += A.dz;                       // :581-583
        if (A.legacy) check = false;                              // slicekernel.cu:70: unconditional fetch
        else {
            px = __builtin_fmaf(px - 0.5f, ix, 0.5f);             // :585
            py = __builtin_fmaf(py - 0.5f, iy, 0.5f);
            pz = __builtin_fmaf(pz - 0.5f, iz, 0.5f);
        }
    }
    float s = 0.f;
    if (!check || (px < 1.0f && px >= 0.0f && py < 1.0f && py >= 0.0f && pz < 1.0f && pz >= 0.0f)) {
        float L = A.V.big ? tex3d_raw<VOXEL, TEX8, true>(A.V, px, py, pz) : tex3d_raw<VOXEL, TEX8, false>(A.V, px, py, pz);
        s = (VOXEL == VV_VOXEL_U8) ? L / 255.0f : L;              // normalised-float read mode, kernel.cu:46
    }
    A.buffer[offset] = s;
}

void launch_slice(const SliceArgs &a, hipStream_t s)
{
    dim3 block(16, 16), grid((unsigned)((a.width + 15) / 16), (unsigned)((a.height + 15) / 16));
    if (a.V_type == VV_VOXEL_F32) {
        if (a.tex8) hipLaunchKernelGGL((slice_kernel<VV_VOXEL_F32, true>), grid, block, 0, s, a);
        else        hipLaunchKernelGGL((slice_kernel<VV_VOXEL_F32, false>), grid, block, 0, s, a);
    } else {
        if (a.tex8) hipLaunchKernelGGL((slice_kernel<VV_VOXEL_U8, true>), grid, block, 0, s, a);
        else        hipLaunchKernelGGL((slice_kernel<VV_VOXEL_U8, false>), grid, block, 0, s, a);
    }
}

// ---------------------------------------------------------------------------
// first pass images (firstpass.vert:6 / firstpass.frag:4 through the analytic ray source)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void first_pass_kernel(FrameParams P, uint32_t *__restrict__ front, uint32_t *__restrict__ back)
{
#pragma clang fp contract(off)
    const int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (x >= P.W || y >= P.H) return;
    // same arithmetic as ray_endpoints (analytic), plus the visibility of each face
    float ndx = (2.0f * ((float)x + 0.5f)) / (float)P.W - 1.0f;
    float ndy = (2.0f * ((float)y + 0.5f)) / (float)P.H - 1.0f;
    float sx = ndx * P.tan_half_x, sy = ndy * P.tan_half_y;
    float d[3], o[3] = {P.cam_pos[0], P.cam_pos[1], P.cam_pos[2]};
    for (int a = 0; a < 3; a++) d[a] = (P.side[a] * sx + P.up[a] * sy) + P.look[a];
    float tmin = -INFINITY, tmax = INFINITY;
    bool miss = false;
    for (int a = 0; a < 3; a++) {
        float s = P.scale[a];
        if (d[a] != 0.0f) {
            float t1 = (-s - o[a]) / d[a], t2 = (s - o[a]) / d[a];
            tmin = fmaxf(tmin, fminf(t1, t2)); tmax = fminf(tmax, fmaxf(t1, t2));
        } else if (o[a] < -s || o[a] > s) miss = true;
    }
    uint32_t f = 0u, b = 0u;
    if (!(miss || !(tmin <= tmax) || !(tmax > 0.0f))) {
        auto enc = [&](float t) {
            uint32_t v = 0xff000000u;
            for (int a = 0; a < 3; a++) {
                float c = (o[a] + d[a] * t) * 0.5f + 0.5f;
                c = fmaxf(0.f, fminf(c, 1.f));
                v |= (uint32_t)floorf(c * 255.0f + 0.5f) << (8 * a);
            }
            return v;
        };
        b = enc(tmax);
        if (tmin > 0.0f) f = enc(tmin);
    }
    front[(size_t)y * P.W + x] = f;
    back[(size_t)y * P.W + x] = b;
}

void launch_first_pass(const FrameParams &P, uint32_t *front, uint32_t *back, hipStream_t s)
{
    dim3 grid((P.W + 15) / 16, (P.H + 15) / 16);
    hipLaunchKernelGGL(first_pass_kernel, grid, dim3(256), 0, s, P, front, back);
}

// ---------------------------------------------------------------------------
// generator
// ---------------------------------------------------------------------------
struct EllipsoidSet {
    int n;
    float cx[kMaxEllipsoids], cy[kMaxEllipsoids], cz[kMaxEllipsoids];
    float ax[kMaxEllipsoids], ay[kMaxEllipsoids], az[kMaxEllipsoids];
    uint8_t color[kMaxEllipsoids];
};

// The inside test of volumegenerator.cpp:57-59 is separable: ((ex*ex) + (ey*ey)) + (ez*ez) with
// ex = (c.x - i/mx) / a.x depending on i only, ey on j only, ez on k only.  A first small kernel
// evaluates the three squared terms of every ellipsoid for every index of its axis, with the
// reference's own operations (uncontracted binary32, IEEE division); the volume pass then needs two
// additions and a compare per voxel and ellipsoid, and skips an ellipsoid for a whole row when
// fl(eyy + ezz) >= 1 (rounding is monotonic and exx >= 0, so fl(fl(exx + eyy) + ezz) >= fl(eyy + ezz)
// >= 1 there) and for a 16-voxel chunk the same way with the smallest exx of the chunk.  The marker
// slab fi >= 0.99 (:85-87) is i >= i_mark, fi being monotonic in i (found on the host with the same
// float division).  Same bytes as n calls of drawEllipsoid, 30x faster than 16 divisions per voxel.
// Tables (floats), n8 = n rounded up to 8 with +inf in the padding (always skipped):
//   TX[n][nx16]  TY[ny][n8]  TZ[nz][n8]  TXMIN[nx16/16][n8]  TXS[n][nx16/16]  TXE[n][nx16/16]  TXM[n][nx16/16]      (nx16 = nx rounded up to 16)
struct GenTables {
    int nx16, nch, n8; size_t ty, tz, txmin, txs, txe, txm, total;
    __host__ __device__ GenTables(int nx, int ny, int nz, int n)
    {
        nx16 = (nx + 15) & ~15; nch = nx16 / 16; n8 = (n + 7) & ~7;
        ty = (size_t)n * nx16;
        tz = ty + (size_t)ny * n8;
        txmin = tz + (size_t)nz * n8;
        txs = txmin + (size_t)nch * n8;            // TXS[n][nch]: the x term of a chunk's first voxel
        txe = txs + (size_t)n * nch;               // TXE[n][nch]: ... of its last voxel (i clamped to nx - 1)
        txm = txe + (size_t)n * nch;               // TXM[n][nch]: the chunk minima again, ellipsoid-major (lanes of a row wave read consecutive floats)
        total = txm + (size_t)n * nch;
    }
};

__global__ __launch_bounds__(256) void ellipsoid_tables_kernel(float *__restrict__ tab, int nx, int ny, int nz, EllipsoidSet E)
{
#pragma clang fp contract(off)
    const GenTables G(nx, ny, nz, E.n);
    const size_t total = G.total;
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        if (t < G.ty) {                                         // TX
            const int e = (int)(t / G.nx16), i = (int)(t % G.nx16);
            const float fi = ((float)i) / ((float)nx);          // :44
            const float ex = (E.cx[e] - fi) / E.ax[e];
            tab[t] = ex * ex;
        } else if (t < G.tz) {                                  // TY
            const int j = (int)((t - G.ty) / G.n8), e = (int)((t - G.ty) % G.n8);
            const float fj = ((float)j) / ((float)ny);          // :45
            const float ey = e < E.n ? (E.cy[e] - fj) / E.ay[e] : 0.f;
            tab[t] = e < E.n ? ey * ey : __builtin_inff();
        } else if (t < G.txmin) {                               // TZ
            const int k = (int)((t - G.tz) / G.n8), e = (int)((t - G.tz) % G.n8);
            const float fk = ((float)k) / ((float)nz);          // :46
            const float ez = e < E.n ? (E.cz[e] - fk) / E.az[e] : 0.f;
            tab[t] = e < E.n ? ez * ez : __builtin_inff();
        } else if (t >= G.txm) {                                // smallest x term of a chunk, ellipsoid-major
            const int e = (int)((t - G.txm) / G.nch), ch = (int)((t - G.txm) % G.nch);
            float m = __builtin_nanf("");
            for (int v = 0; v < 16; ++v) {
                const int i = ch * 16 + v;
                if (i >= nx) break;
                const float fi = ((float)i) / ((float)nx);
                const float ex = (E.cx[e] - fi) / E.ax[e];
                m = fminf(m, ex * ex);
            }
            tab[t] = m;
        } else if (t >= G.txs) {                                // x term of a chunk's first / last voxel
            const bool last = t >= G.txe;
            const size_t u = t - (last ? G.txe : G.txs);
            const int e = (int)(u / G.nch), ch = (int)(u % G.nch);
            int i = ch * 16 + (last ? 15 : 0);
            if (i > nx - 1) i = nx - 1;
            const float fi = ((float)i) / ((float)nx);
            const float ex = (E.cx[e] - fi) / E.ax[e];
            tab[t] = ex * ex;
        } else {                                                // smallest x term of a 16-voxel chunk
            const int ch = (int)((t - G.txmin) / G.n8), e = (int)((t - G.txmin) % G.n8);
            float m = e < E.n ? __builtin_nanf("") : __builtin_inff();
            for (int v = 0; v < 16 && e < E.n; ++v) {
                const int i = ch * 16 + v;
                if (i >= nx) break;
                const float fi = ((float)i) / ((float)nx);
                const float ex = (E.cx[e] - fi) / E.ax[e];
                m = fminf(m, ex * ex);                          // fminf skips NaNs: a NaN term is never inside
            }
            tab[t] = m;
        }
    }
}

struct EllipsoidColors { int n; uint8_t color[kMaxEllipsoids]; };

// One thread produces 16 consecutive x voxels of one (j,k) row; the store is one 16-byte write.
//
// ROWWAVE (rows of a multiple of 1024 voxels, fresh volumes): the 64 chunks of a wave lie in one row, and the voxels of a row that are
// inside an ellipsoid are an INTERVAL of x -- the x term is unimodal in i (ex = (c.x - i/mx) / a.x is monotonic, its square falls then
// rises) and every rounding on the way to fl(fl(exx + eyy) + ezz) < 1 is monotonic.  So a chunk is inside as a whole iff its first and
// last voxel are, touched at all iff its smallest x term is, and at most two chunks of the row are touched in part: only those two get
// the per-voxel test, by 32 lanes of the wave at once (a ballot carries the 2 x 16 results to their owners, which turn 4 bits at a time
// into byte masks through a 16-entry table).  ~50 instructions per (row, ellipsoid) pair instead of ~96 for 16 tests in every lane, and
// two coalesced table reads instead of sixteen values per lane.  Same bytes (test_generator_*: against the compiled reference).
__device__ __forceinline__ void vv_gen_store16(uint8_t *p, uint4 v)
{
#ifdef VV_GEN_NT
    typedef uint32_t __attribute__((ext_vector_type(4))) u4v;
    __builtin_nontemporal_store(u4v{v.x, v.y, v.z, v.w}, (u4v *)p);
#else
    *(uint4 *)p = v;
#endif
}
template <bool ROWWAVE>
__global__ __launch_bounds__(256) void ellipsoid_kernel(uint8_t *__restrict__ out, int nx, int ny, int nz, int xchunks,
                                                        int i_mark, const float *__restrict__ tab, EllipsoidColors E, int in_place)
{
#pragma clang fp contract(off)
    const GenTables G(nx, ny, nz, E.n);
    const float *TX = tab, *TY = tab + G.ty, *TZ = tab + G.tz, *TXMIN = tab + G.txmin, *TXS = tab + G.txs, *TXE = tab + G.txe, *TXM = tab + G.txm;
    const int n8 = G.n8, nx16 = G.nx16, nch = G.nch;
    __shared__ uint32_t s_lut[16];                 // 4 bits -> 4 byte masks
    if (ROWWAVE) {
        if (threadIdx.x < 16) {
            const uint32_t b = threadIdx.x;
            s_lut[b] = ((b & 1u) ? 0xffu : 0u) | ((b & 2u) ? 0xff00u : 0u) | ((b & 4u) ? 0xff0000u : 0u) | ((b & 8u) ? 0xff000000u : 0u);
        }
        __syncthreads();
    }
    const int lane = threadIdx.x & 63;
    const size_t total = (size_t)xchunks * ny * nz;
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const int xc = (int)(t % xchunks);
        const size_t row = t / xchunks;
        const int j = (int)(row % ny), k = (int)(row / ny);
        const int i0 = xc * 16;
        const size_t base = row * (size_t)nx + i0;
        uint8_t vals[16];
        uint32_t packed4[4] = {0u, 0u, 0u, 0u};                                     // (ROWWAVE keeps the 16 voxels packed)
#pragma unroll
        for (int v = 0; v < 16; ++v) vals[v] = 0;                                   // ctor zero-fill :12-23
        if (in_place) {                                                             // else keep, :60-62
#pragma unroll
            for (int v = 0; v < 16; ++v) if (i0 + v < nx) vals[v] = out[base + v];
        }
        // ellipsoids in batches of 8: the row / chunk terms of a batch are six 16-byte loads
        for (int e0 = 0; e0 < E.n; e0 += 8) {
            // (row wave: j and k are the same in every lane; saying so lets the row terms come through scalar loads and the row test be a scalar branch,
            //  and the chunk minimum is read only for the ellipsoids that touch the row)
            const int ju = ROWWAVE ? __builtin_amdgcn_readfirstlane(j) : j, ku = ROWWAVE ? __builtin_amdgcn_readfirstlane(k) : k;
            const float4 *py = (const float4 *)(TY + (size_t)ju * n8 + e0), *pz = (const float4 *)(TZ + (size_t)ku * n8 + e0);
            const float4 y0 = py[0], y1 = py[1], z0 = pz[0], z1 = pz[1];
            float4 m0 = make_float4(0.f, 0.f, 0.f, 0.f), m1 = m0;
            if (!ROWWAVE) { const float4 *pm = (const float4 *)(TXMIN + (size_t)xc * n8 + e0); m0 = pm[0]; m1 = pm[1]; }
            const float yy[8] = {y0.x, y0.y, y0.z, y0.w, y1.x, y1.y, y1.z, y1.w};
            const float zz[8] = {z0.x, z0.y, z0.z, z0.w, z1.x, z1.y, z1.z, z1.w};
            const float mn[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int e = e0 + q;
                if (e >= E.n) break;
                const float eyy = yy[q], ezz = zz[q];
                if (eyy + ezz >= 1.0f) continue;                    // no voxel of this row is inside (see above)
                if (ROWWAVE) {
                    // (the branch above was taken by the whole wave or by none: the wave is one row)
                    const float xs = TXS[(size_t)e * nch + xc], xe = TXE[(size_t)e * nch + xc];
                    const float xm = TXM[(size_t)e * nch + xc];
                    const bool fs = (xs + eyy) + ezz < 1.0f, fe = (xe + eyy) + ezz < 1.0f, fm = (xm + eyy) + ezz < 1.0f;
                    const bool full = fs && fe, part = fm && !full;
                    const uint32_t col4 = (uint32_t)E.color[e] * 0x01010101u;
                    const unsigned long long bp = __builtin_amdgcn_ballot_w64(part);
#pragma unroll
                    for (int d = 0; d < 4; ++d) packed4[d] = full ? col4 : packed4[d];
                    if (bp != 0ull) {
                        const int ca = __builtin_ctzll(bp);
                        const unsigned long long bp2 = bp & (bp - 1ull);
                        const int cb = bp2 ? __builtin_ctzll(bp2) : -1;
                        const int cg = lane < 16 ? ca : cb;
                        const bool helper = lane < 32 && cg >= 0;
                        const float xv = helper ? TX[(size_t)e * nx16 + 16 * (size_t)(xc - lane + cg) + (lane & 15)] : __builtin_inff();
                        const bool pv = helper && (xv + eyy) + ezz < 1.0f;                  // :57-59
                        const uint32_t m32 = (uint32_t)__builtin_amdgcn_ballot_w64(pv);
                        const uint32_t my16 = lane == ca ? (m32 & 0xffffu) : (lane == cb ? (m32 >> 16) : 0u);
#pragma unroll
                        for (int d = 0; d < 4; ++d) {
                            const uint32_t bm = s_lut[(my16 >> (4 * d)) & 15u];
                            packed4[d] = (packed4[d] & ~bm) | (col4 & bm);
                        }
                    }
                    continue;
                }
                if ((mn[q] + eyy) + ezz >= 1.0f) continue;          // ... nor of this chunk
                const float4 *tx = (const float4 *)(TX + (size_t)e * nx16 + i0);
                const uint8_t col = E.color[e];
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const float4 x4 = tx[q4];
                    const float xs[4] = {x4.x, x4.y, x4.z, x4.w};
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const float qv = (xs[w] + eyy) + ezz;                       // :57-59
                        if (qv < 1.0f) vals[q4 * 4 + w] = col;                      // (double)q < 1.0
                    }
                }
            }
        }
        if (ROWWAVE) {                                          // (rows are multiples of 1024 voxels: whole, aligned chunks)
            if (E.n > 0 && i0 + 15 >= i_mark) {                 // :85-87, after the last drawEllipsoid
#pragma unroll
                for (int v = 0; v < 16; ++v)
                    if (i0 + v >= i_mark) packed4[v >> 2] = (packed4[v >> 2] & ~(0xffu << (8 * (v & 3)))) | (4u << (8 * (v & 3)));
            }
            vv_gen_store16(out + base, make_uint4(packed4[0], packed4[1], packed4[2], packed4[3]));
            continue;
        }
        if (E.n > 0 && i0 + 15 >= i_mark) {                     // :85-87, after the last drawEllipsoid
#pragma unroll
            for (int v = 0; v < 16; ++v) if (i0 + v >= i_mark) vals[v] = 4;
        }
        if (i0 + 16 <= nx && (base & 15) == 0) {
            uint32_t packed[4] = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int v = 0; v < 16; ++v) packed[v >> 2] |= (uint32_t)vals[v] << (8 * (v & 3));
            *(uint4 *)(out + base) = make_uint4(packed[0], packed[1], packed[2], packed[3]);
        } else {
#pragma unroll
            for (int v = 0; v < 16; ++v) if (i0 + v < nx) out[base + v] = vals[v];
        }
    }
}

// The rows-of-whole-waves form once more, for up to 8 ellipsoids whose x tables fit the LDS twice per CU (drawDefaultBrain at 1024^3: 39 KB,
// at 2048^3: 78 KB).  ellipsoid_kernel<true> waits for memory once per row and twice per touching ellipsoid (tables that do not stay in the
// 32 KB L1 beside the stores) and spends ~210 VALU instructions per row, 90 of them on index divisions and on testing all eight ellipsoids.
// Here a block of 16 waves stages TX / TXS / TXE / TXM in LDS once; every wave then walks row segments (1024 voxels):
//   * segments are handed out in tickets of 8 consecutive ones: a block's share is fixed (rotated, so that every block samples the whole volume),
//     its waves draw from a counter in LDS (rows differ tenfold in cost: with a fixed share per wave the slowest of 8192 waves ends a third
//     after the average one; a device-wide counter retires one ticket per ~12 ns); the index arithmetic is scalar (a multiply-high for the division by ny);
//   * a wave works on two segments at a time and has the row terms of its NEXT two already in flight: one 32-lane vector load (y terms in
//     lanes 0..7 / 16..23, z terms in 8..15 / 24..31; a scalar load would be waited for by the first LDS read, they share a counter); the two
//     16-byte stores per lane stay in flight across the wait for that load (`vmcnt(2)`);
//   * which ellipsoids touch the row is one DPP add + compare + ballot; only those are visited (scalar bit loop), everything they need is LDS --
//     including the row terms and the colour, read back as broadcasts: an add or a select with an SGPR operand issues at 0.6 of the rate.
// Same arithmetic, same order of ellipsoids (volumegenerator.cpp:51-63), same bytes.
struct RowsArgs { uint32_t ny_magic; int ny_shift; int segs_log2; uint32_t rot; };      // q = mulhi(row, ny_magic) >> ny_shift (ny >= 2)
template <int SEGS_LOG2>                          // rows of 1024 << SEGS_LOG2 voxels
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(8, 8)))
void ellipsoid_rows_kernel(uint8_t *__restrict__ out, int nx, int ny, int nz, int i_mark, const float *__restrict__ tab, EllipsoidColors E, RowsArgs A)
{
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) float gen_lds[];
    const GenTables G(nx, ny, nz, E.n);
    constexpr int nch = 64 << SEGS_LOG2, nx16 = 1024 << SEGS_LOG2;                   // (G.nch, G.nx16 == nx; G.n8 == 8)
    const int n = E.n;
    // the chunk terms of an ellipsoid side by side -- LC[q][0 / 1 / 2][chunk] = first-voxel / last-voxel / minimum term -- so that one vector
    // address and three immediate offsets reach them
    float *LX = gen_lds, *LC = LX + (size_t)n * nx16;
    char *patch = (char *)(LC + (size_t)3 * n * nch) + (threadIdx.x >> 6) * 32;      // 2 x 16 bytes per wave (below)
    uint32_t *ctr = (uint32_t *)((char *)(LC + (size_t)3 * n * nch) + 16 * 32);       // the block's ticket counter
    if (threadIdx.x == 0) *ctr = 0u;
    // per wave: the row terms of its two current segments (32 floats) and the 8 colours (x 0x01010101), read back as LDS broadcasts: a value
    // that reaches an add or a select through an SGPR (v_readlane) halves that instruction's issue rate (profiles/r04_instruction_rates.txt)
    float *wterms = (float *)((char *)ctr + 16) + (threadIdx.x >> 6) * 40;
    if ((threadIdx.x & 63) < 8) ((uint32_t *)wterms)[32 + (threadIdx.x & 63)] = (int)(threadIdx.x & 63) < n ? (uint32_t)E.color[threadIdx.x & 63] * 0x01010101u : 0u;
    for (int i = threadIdx.x * 4; i < n * nx16; i += blockDim.x * 4) *(float4 *)(LX + i) = *(const float4 *)(tab + i);
    for (int i = threadIdx.x; i < 3 * n * nch; i += blockDim.x) {                    // tab: TXS[n][nch] TXE[n][nch] TXM[n][nch]
        const int t = i / (n * nch), r = i - t * (n * nch), q = r / nch, c = r - q * nch;
        LC[(q * 3 + t) * nch + c] = tab[G.txs + i];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const uint32_t rows = (uint32_t)ny * (uint32_t)nz, total = rows << SEGS_LOG2;                           // (generate_ellipsoids: < 2^31)
    const uint32_t ty_b = (uint32_t)G.ty * 4u, tz_b = (uint32_t)G.tz * 4u;
    // lanes 0..7: the y terms of the segment's row for the 8 ellipsoids, lanes 8..15: the z terms
    // the row terms of two segments in one load: lanes 0..7 the y terms of segment a's row for the 8 ellipsoids, 8..15 its z terms, 16..31 segment b's
    const uint32_t lane4 = (uint32_t)(lane & 7) * 4u;
    auto row_terms = [&](uint32_t sa_, uint32_t sb_ /* wave-uniform */) -> float {
        const uint32_t sa = __builtin_amdgcn_readfirstlane(sa_), sb = __builtin_amdgcn_readfirstlane(sb_);     // (scalars for the compiler too)
        const uint32_t ra = sa >> SEGS_LOG2, rb = sb >> SEGS_LOG2;
        const uint32_t ka = ny > 1 ? __umulhi(ra, A.ny_magic) >> A.ny_shift : ra, ja = ra - ka * (uint32_t)ny;
        const uint32_t kb = ny > 1 ? __umulhi(rb, A.ny_magic) >> A.ny_shift : rb, jb = rb - kb * (uint32_t)ny;
        const uint32_t o0 = ty_b + ja * 32u, o1 = tz_b + ka * 32u, o2 = ty_b + jb * 32u, o3 = tz_b + kb * 32u;      // (scalars)
        const uint32_t off = (lane < 16 ? (lane < 8 ? o0 : o1) : (lane < 24 ? o2 : o3)) + lane4;
        return lane < 32 ? *(const float *)((const char *)tab + off) : 0.f;
    };
    // the marker slab (:85-87: x >= 0.99 after the last drawEllipsoid; n > 0 here) lies in the last segment of a row: this lane's byte masks there
    uint32_t mk[4];
    {
        const int c0 = (((nx >> 10) - 1) * 64 + lane) * 16;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            mk[d] = 0u;
#pragma unroll
            for (int v = 0; v < 4; ++v) if (c0 + 4 * d + v >= i_mark) mk[d] |= 0xffu << (8 * v);
        }
    }
    const uint32_t last_seg = ((uint32_t)nx >> 10) - 1u;
    // (a pair whose second segment does not exist -- odd total -- redoes the last one: the same bytes)
    const uint32_t nmask = (1u << n) - 1u;
    auto do_segment = [&](uint32_t sg /* scalar */, float cur, uint32_t touch /* scalar: bit q = ellipsoid q touches the row */, int lo /* lane of the row's y terms */) {
        const uint32_t row = sg >> SEGS_LOG2, seg = sg - (row << SEGS_LOG2);
        const int xc = (int)seg * 64 + lane, i0 = xc * 16;
        uint32_t packed4[4] = {0u, 0u, 0u, 0u};                                    // ctor zero-fill :12-23
        while (touch) {
            const int q = __builtin_ctz(touch);
            touch &= touch - 1u;
            const float *wq = wterms + q;                                        // (one vector address, three broadcast reads)
            const float eyy = wq[lo], ezz = wq[lo + 8];
            const uint32_t col4 = ((const uint32_t *)wq)[32];
            const float *lc = LC + q * (3 * nch) + xc;
            const float xs = lc[0], xe = lc[nch], xm = lc[2 * nch];
            const bool fs = (xs + eyy) + ezz < 1.0f, fe = (xe + eyy) + ezz < 1.0f, fm = (xm + eyy) + ezz < 1.0f;
            const bool full = fs && fe;
            // (ballots of the plain comparisons: the compiler turns a ballot of `fm && !full` into two more vector instructions)
            const unsigned long long bp = __builtin_amdgcn_ballot_w64(fm) & ~(__builtin_amdgcn_ballot_w64(fs) & __builtin_amdgcn_ballot_w64(fe));
#pragma unroll
            for (int d = 0; d < 4; ++d) packed4[d] = full ? col4 : packed4[d];
            if (bp != 0ull) {
                const int ca = __builtin_ctzll(bp);
                const unsigned long long bp2 = bp & (bp - 1ull);
                const int cb = bp2 ? __builtin_ctzll(bp2) : -1;
                const int cg = lane < 16 ? ca : cb;
                const bool helper = lane < 32 && cg >= 0;
                const float xv = helper ? LX[q * nx16 + 16 * (xc - lane + cg) + (lane & 15)] : __builtin_inff();
                const bool pv = helper && (xv + eyy) + ezz < 1.0f;                  // :57-59
                // the owners of the two chunks park their 16 bytes in LDS, the helpers whose voxel is inside write its colour over them, the
                // owners read the bytes back (LDS operations of one wave complete in order): 6 instructions instead of 24 for turning
                // 2 x 16 ballot bits into byte masks
                const bool own_b = lane == cb, own = lane == ca || own_b;
                char *slot = patch + (own_b ? 16 : 0);
                // (the hardware keeps one wave's LDS operations in order; the wave-scope fences + barriers say so to the compiler too: they emit nothing)
                if (own) *(uint4 *)slot = make_uint4(packed4[0], packed4[1], packed4[2], packed4[3]);
                vv_wave_lds_order();
                if (pv) patch[lane] = (char)col4;
                vv_wave_lds_order();
                if (own) { const uint4 r = *(const uint4 *)slot; packed4[0] = r.x; packed4[1] = r.y; packed4[2] = r.z; packed4[3] = r.w; }
                vv_wave_lds_order();                                                   // (the next ellipsoid's owners overwrite the slots)
            }
        }
        if (seg == last_seg) {
#pragma unroll
            for (int d = 0; d < 4; ++d) packed4[d] = (packed4[d] & ~mk[d]) | (0x04040404u & mk[d]);
        }
        uint8_t *rowp = out + (size_t)row * (size_t)nx;                               // scalar base + 32-bit lane offset
        vv_gen_store16(rowp + (uint32_t)i0, make_uint4(packed4[0], packed4[1], packed4[2], packed4[3]));
    };
    // Work is handed out in tickets of 4 segment pairs (8 consecutive row segments).  Rows differ tenfold in cost, and with a fixed share per
    // wave the slowest of 8192 waves ends a third after the average one (1024^3: 128 rows per wave).  A device-wide counter does not work either
    // -- returning atomics on one address retire at ~12 ns each (measured: 131 072 tickets = 1.6 ms) -- so the split is in two levels: a block's
    // share is fixed, its i-th ticket being i NB + ((b + i R) mod NB) with R ~ 0.618 NB (every block samples the whole volume; without the
    // rotation block b would get the same rows of every slice), and the block's 16 waves draw i from a counter in LDS.  The next ticket is drawn
    // when a ticket is started, and the row terms of its first pair are requested during the last pair of this one: neither wait is exposed.
    const uint32_t pairs = (total + 1u) >> 1, all_tickets = (pairs + 3u) >> 2;
    const uint32_t NB = gridDim.x, tickets = (all_tickets + NB - 1u) / NB;            // tickets per block (the last round may be short: clamped, redone)
    auto ticket_of = [&](uint32_t i /* scalar */) -> uint32_t {                        // i < tickets: the block's i-th ticket; else >= all_tickets
        const uint32_t r = (blockIdx.x + i * A.rot) % NB;                             // (once per 8 row segments)
        return i < tickets ? min(i * NB + r, all_tickets - 1u) : all_tickets;
    };
    auto draw = [&]() -> uint32_t { return lane == 0 ? atomicAdd(ctr, 1u) : 0u; };
    auto terms_of_pair = [&](uint32_t pr) { return row_terms(min(2u * pr, total - 1u), min(2u * pr + 1u, total - 1u)); };
    uint32_t ti = __builtin_amdgcn_readfirstlane(draw());                              // index of the ticket within the block's share
    float pre = terms_of_pair(min(4u * min(ticket_of(ti), all_tickets - 1u), pairs - 1u));
    asm volatile("" :: "v"(pre));                  // (landed before the loop)
    while (ti < tickets) {
        const uint32_t tks = __builtin_amdgcn_readfirstlane(ticket_of(__builtin_amdgcn_readfirstlane(ti)));   // (a scalar for the compiler too)
        const uint32_t ti_next_v = draw();         // (lane 0; read three pairs later)
#pragma unroll
        for (uint32_t st = 0; st < 4u; ++st) {
            const uint32_t pr = min(4u * tks + st, pairs - 1u);
            const uint32_t sga = min(2u * pr, total - 1u), sgb = min(2u * pr + 1u, total - 1u);
            const float cur = pre;
            if (st < 3u) pre = terms_of_pair(min(4u * tks + st + 1u, pairs - 1u));
            else {
                ti = __builtin_amdgcn_readfirstlane(ti_next_v);
                pre = terms_of_pair(min(4u * min(ticket_of(ti), all_tickets - 1u), pairs - 1u));      // (beyond the block's share: a valid address, never used)
            }
            // lanes q and 16 + q (q < 8): eyy + ezz of ellipsoid q for segment a / b (DPP row_shl:8 brings lane + 8's z term); the padding holds +inf
            const float tsum = cur + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(cur), 0x108, 0xf, 0xf, false));
            const uint32_t tb = (uint32_t)__builtin_amdgcn_ballot_w64(tsum < 1.0f);      // else no voxel of the row is inside (ellipsoid_kernel)
            if (lane < 32) wterms[lane] = cur;                                          // (this wave's own reads follow in order)
            vv_wave_lds_order();
            do_segment(sga, cur, tb & nmask, 0);
            do_segment(sgb, cur, (tb >> 16) & nmask, 16);
            vv_wave_lds_order();                                                        // (the next pair's terms overwrite wterms)
        }
    }
}

size_t generate_scratch_floats(int nx, int ny, int nz, int n)
{
    return GenTables(nx, ny, nz, n).total + 16;
}

void launch_generate_ellipsoids(uint8_t *out, int nx, int ny, int nz, int n,
                                const float *centers, const float *axes, const uint8_t *colors,
                                int in_place, float *scratch, hipStream_t s)
{
    EllipsoidSet E;
    EllipsoidColors Cc;
    E.n = n; Cc.n = n;
    for (int e = 0; e < n; ++e) {
        E.cx[e] = centers[3*e]; E.cy[e] = centers[3*e+1]; E.cz[e] = centers[3*e+2];
        E.ax[e] = axes[3*e];    E.ay[e] = axes[3*e+1];    E.az[e] = axes[3*e+2];
        E.color[e] = colors[e]; Cc.color[e] = colors[e];
    }
    // first x index of the marker slab: fi = float(i) / float(nx) is monotonic in i (:44, :85)
    int i_mark = nx;
    for (int i = 0; i < nx; ++i) {
        volatile float fi = ((float)i) / ((float)nx);
        if ((double)fi >= 0.99) { i_mark = i; break; }
    }
    const GenTables G(nx, ny, nz, n);
    size_t tblocks = (G.total + 255) / 256;
    if (tblocks > 4096) tblocks = 4096;
    if (tblocks) hipLaunchKernelGGL(ellipsoid_tables_kernel, dim3((unsigned)tblocks), dim3(256), 0, s, scratch, nx, ny, nz, E);
    const int xchunks = (nx + 15) / 16;
    const size_t total = (size_t)xchunks * ny * nz;
    size_t blocks = (total + 255) / 256;
    // blocks of 256 chunks, grid-stride beyond 65536 of them: busy and empty rows mix better over many short blocks (drawDefaultBrain at 1024^3: 1.03 ms with
    // 1024 blocks, 0.50 with 8192, 0.46 with 65536, 0.51 with one block per 256 chunks; the fill alone 0.26 / 0.27 / 0.20 / 0.22 ms)
    size_t cap = 65536;
    if (const char *e = getenv("VV_GEN_BLOCKS")) { const long v = atol(e); if (v >= 256) cap = (size_t)v; }     // (experiment knob)
    if (blocks > cap) blocks = cap;
    if (blocks == 0) return;
    // rows of whole waves (a multiple of 1024 voxels) of a fresh volume take the interval form
    if (!in_place && n > 0 && xchunks % 64 == 0 && nx % 16 == 0) {
        // ... with the x tables in LDS when they fit it twice per CU (ellipsoid_rows_kernel)
        const size_t lds = ((size_t)n * G.nx16 + 3 * (size_t)n * G.nch) * sizeof(float) + 16 * 32 + 16 + 16 * 160;      // tables + 32 bytes per wave + the ticket counter + row terms and colours per wave
        // the kernel's dynamic LDS limit is raised once per device (a one-process multi-GPU host, include/volviz_mgpu.h, comes here with each of its devices)
        static std::atomic<unsigned long long> dev_ok{0ull}, dev_bad{0ull};        // (a multi-GPU host may come here from one thread per device)
        int rows_ok = 0;
        {
            int dv = 0;
            if (hipGetDevice(&dv) == hipSuccess && dv >= 0 && dv < 64) {
                const unsigned long long bit = 1ull << dv;
                if (!((dev_ok.load() | dev_bad.load()) & bit)) {
                    const bool ok = hipFuncSetAttribute((const void *)ellipsoid_rows_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) == hipSuccess &&
                                    hipFuncSetAttribute((const void *)ellipsoid_rows_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) == hipSuccess &&
                                    hipFuncSetAttribute((const void *)ellipsoid_rows_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) == hipSuccess;
                    (void)hipGetLastError();
                    (ok ? dev_ok : dev_bad).fetch_or(bit);
                }
                rows_ok = (dev_ok.load() & bit) ? 1 : 0;
            }
        }
        const size_t segments = (size_t)ny * nz * (nx / 1024);
        const int segs = nx / 1024;
        if (rows_ok == 1 && n <= 8 && lds <= 80u * 1024u && segments < (1ull << 28) && (segs == 1 || segs == 2 || segs == 4) && !getenv("VV_GEN_NO_LDS")) {
            int dev = 0, cus = 256;
            if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
            unsigned threads = 1024;
            if (const char *e = getenv("VV_GEN_THREADS")) { const long v = atol(e); if (v == 256 || v == 512 || v == 1024) threads = (unsigned)v; }   // (experiment knob)
            size_t nb = (size_t)cus * (2048 / threads);
            if (nb > (segments + 15) / 16) nb = (segments + 15) / 16;
            RowsArgs A;
            int L = 0; while ((2u << L) <= (unsigned)ny) ++L;                          // floor(log2(ny))
            A.ny_magic = ny > 1 ? (uint32_t)(((1ull << (31 + L)) + (uint64_t)ny - 1) / (uint64_t)ny) : 0u;   // exact for row < 2^28 (checked above)
            A.ny_shift = L > 0 ? L - 1 : 0;
            A.segs_log2 = 0; while ((1 << A.segs_log2) < segs) ++A.segs_log2;
            A.rot = (uint32_t)(0.6180339887 * (double)nb) | 1u;                       // the rotation of a block's tickets (ellipsoid_rows_kernel)
            if (A.segs_log2 == 0) hipLaunchKernelGGL(ellipsoid_rows_kernel<0>, dim3((unsigned)nb), dim3(threads), lds, s, out, nx, ny, nz, i_mark, scratch, Cc, A);
            else if (A.segs_log2 == 1) hipLaunchKernelGGL(ellipsoid_rows_kernel<1>, dim3((unsigned)nb), dim3(threads), lds, s, out, nx, ny, nz, i_mark, scratch, Cc, A);
            else hipLaunchKernelGGL(ellipsoid_rows_kernel<2>, dim3((unsigned)nb), dim3(threads), lds, s, out, nx, ny, nz, i_mark, scratch, Cc, A);
            return;
        }
        hipLaunchKernelGGL(ellipsoid_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, s, out, nx, ny, nz, xchunks, i_mark, scratch, Cc, in_place);
    } else
        hipLaunchKernelGGL(ellipsoid_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, s, out, nx, ny, nz, xchunks, i_mark, scratch, Cc, in_place);
}

// ---------------------------------------------------------------------------
// u8 -> f32 promotion (v / 255), 16 voxels per thread
// ---------------------------------------------------------------------------
// b / 255.f for a byte b without the division: q0 = b * c, r = fma(-q0, 255, b), q = fma(r, c, q0) with c = fl(1 / 255) is the correctly
// rounded quotient for every b in 0..255 (checked exhaustively against exact rational arithmetic): one conversion + three full-rate
// instructions instead of a ~10-instruction IEEE division.
__device__ __forceinline__ float byte_over_255(float b)
{
#pragma clang fp contract(off)
    const float c = 1.0f / 255.0f;
    const float q0 = b * c;
    const float r = __builtin_fmaf(-q0, 255.0f, b);
    return __builtin_fmaf(r, c, q0);
}
// One dword (4 voxels) per lane and trip: a wave reads 256 contiguous bytes and writes 1 KB contiguous (the first version converted 16 voxels
// per lane: four stores of 16 bytes 64 bytes apart per lane, and sixteen divisions: 1.75 ms per GiB of voxels).
__global__ __launch_bounds__(256) void promote_kernel(const uint8_t *__restrict__ in, float *__restrict__ out, size_t n)
{
    const size_t nw = n / 4;
    const uint32_t *in32 = (const uint32_t *)in;
    float4 *out4 = (float4 *)out;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t c = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    for (; c + 3 * stride < nw; c += 4 * stride) {           // four dwords in flight per lane
        uint32_t w[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) w[q] = in32[c + q * stride];
#pragma unroll
        for (int q = 0; q < 4; ++q)
            out4[c + q * stride] = make_float4(byte_over_255((float)(w[q] & 0xff)), byte_over_255((float)((w[q] >> 8) & 0xff)),
                                               byte_over_255((float)((w[q] >> 16) & 0xff)), byte_over_255((float)(w[q] >> 24)));
    }
    for (; c < nw; c += stride) {
        const uint32_t w = in32[c];
        out4[c] = make_float4(byte_over_255((float)(w & 0xff)), byte_over_255((float)((w >> 8) & 0xff)),
                              byte_over_255((float)((w >> 16) & 0xff)), byte_over_255((float)(w >> 24)));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const size_t i = nw * 4 + threadIdx.x;
        out[i] = byte_over_255((float)in[i]);
    }
}

void launch_promote_u8_f32(const uint8_t *in, float *out, size_t n, hipStream_t s)
{
    size_t blocks = (n / 4 + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(promote_kernel, dim3((unsigned)blocks), dim3(256), 0, s, in, out, n);
}

// ---------------------------------------------------------------------------
// bricked copy of an f32 volume (VolumeView::bricks, vv_device.h): 4x4x4 voxels per brick,
// rows of 4 + 1 halo voxel, indices beyond the volume clamp to the edge
// ---------------------------------------------------------------------------
size_t brick_copy_bytes(int vtype, int nx, int ny, int nz, uint32_t *sy, uint32_t *sz64)
{
    const size_t brick = vtype == VV_VOXEL_F32 ? BrickGeom<VV_VOXEL_F32>::brick : BrickGeom<VV_VOXEL_U8>::brick;
    const size_t bxv = vtype == VV_VOXEL_F32 ? BrickGeom<VV_VOXEL_F32>::bx : BrickGeom<VV_VOXEL_U8>::bx;
    const bool xcol = vtype == VV_VOXEL_F32 && BrickGeom<VV_VOXEL_F32>::halo == 0;      // no halo voxel: one more (clamped) brick column instead
    const size_t bzv = vtype == VV_VOXEL_F32 ? BrickGeom<VV_VOXEL_F32>::bz : BrickGeom<VV_VOXEL_U8>::bz;
    const size_t nbx = xcol ? (size_t)nx / bxv + 1 : ((size_t)nx + bxv - 1) / bxv, nby = (size_t)ny / 4 + 1, nbz = (size_t)nz / bzv + 1;
    size_t row = nbx * brick;                                     // brick sizes are multiples of 64
    // a row of bricks that is a multiple of 4 KiB gets 64 bytes more (same cache-channel effect as the
    // linear pitch, smaller: rotated C3 -1...-4 %, + Phong -4 %); VV_BRICK_PAD=<bytes, multiple of 64> / 0 overrides
    size_t pad = (vtype == VV_VOXEL_F32 && row % 4096 == 0) ? (BrickGeom<VV_VOXEL_F32>::brick % 128 == 0 ? 128 : 64) : 0;      // u8 bricks (one line each): +7 % with it
    if (const char *e = getenv("VV_BRICK_PAD")) { int t = atoi(e); if (t >= 0 && t <= 4096 && t % 64 == 0) pad = (size_t)t; }
    if (nx >= (1 << 20)) { if (sy) *sy = 0xFFFFFFFFu; if (sz64) *sz64 = 0xFFFFFFFFu; return ~(size_t)0 >> 1; }      // (brick_x_offset's range; ensure_bricks then takes the linear path)
    row += pad;
    const size_t layer = nby * row;
    if (sy) *sy = (uint32_t)row;
    if (sz64) *sz64 = (uint32_t)(layer >> 6);
    return nbz * layer;
}

// one thread per stored element: E elements per brick row (BX voxels + halo [+ padding]), 4 BZ rows per brick (y fastest, then z)
template <typename T, int E, int BX, int BZ, int BRICK_ELEMS>
__global__ __launch_bounds__(256) void brick_kernel(const T *__restrict__ in, size_t row_pitch, size_t slice_pitch, T *__restrict__ out,
                                                    int nx, int ny, int nz, size_t nbx, size_t nby, size_t total,
                                                    size_t brow /* elements per row of bricks */, size_t blayer /* per layer */)
{
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        constexpr int R = 4 * BZ;
        const int e = (int)(t % E), r = (int)((t / E) % R);
        const size_t b = t / (E * R);
        const size_t bx = b % nbx, by = (b / nbx) % nby, bz = b / (nbx * nby);
        const int x = min((int)bx * BX + e, nx - 1), y = min((int)by * 4 + (r & 3), ny - 1), z = min((int)bz * BZ + (r >> 2), nz - 1);
        out[bz * blayer + by * brow + bx * (size_t)BRICK_ELEMS + (size_t)(r * E + e)] =
            e <= BX ? ((const T *)((const char *)in + (size_t)z * slice_pitch + (size_t)y * row_pitch))[x] : T(0);
    }
}

void launch_build_bricks(int vtype, const void *linear, size_t row_pitch, size_t slice_pitch, void *bricks, int nx, int ny, int nz, hipStream_t s)
{
    constexpr int FBX = BrickGeom<VV_VOXEL_F32>::bx;
    const size_t bxv = vtype == VV_VOXEL_F32 ? FBX : 4;
    constexpr int FH = BrickGeom<VV_VOXEL_F32>::halo;
    const bool xcol = vtype == VV_VOXEL_F32 && FH == 0;
    constexpr int FBZ = BrickGeom<VV_VOXEL_F32>::bz;
    const size_t bzv = vtype == VV_VOXEL_F32 ? FBZ : 4;
    const size_t nbx = xcol ? (size_t)nx / bxv + 1 : ((size_t)nx + bxv - 1) / bxv, nby = (size_t)ny / 4 + 1, nbz = (size_t)nz / bzv + 1;
    const size_t per_brick = vtype == VV_VOXEL_F32 ? 4 * FBZ * (FBX + FH) : 128;
    const size_t total = nbx * nby * nbz * per_brick;
    uint32_t sy = 0, sz64 = 0;
    brick_copy_bytes(vtype, nx, ny, nz, &sy, &sz64);
    const size_t esz = vtype == VV_VOXEL_F32 ? 4 : 1, brow = sy / esz, blayer = ((size_t)sz64 << 6) / esz;
    size_t blocks = (total + 255) / 256;
    if (blocks > 256 * 64) blocks = 256 * 64;
    if (vtype == VV_VOXEL_F32)
        hipLaunchKernelGGL((brick_kernel<float, FBX + FH, FBX, FBZ, BrickGeom<VV_VOXEL_F32>::brick / 4>), dim3((unsigned)blocks), dim3(256), 0, s, (const float *)linear, row_pitch, slice_pitch, (float *)bricks, nx, ny, nz, nbx, nby, total, brow, blayer);
    else
        hipLaunchKernelGGL((brick_kernel<uint8_t, 8, 4, 4, 128>), dim3((unsigned)blocks), dim3(256), 0, s, (const uint8_t *)linear, row_pitch, slice_pitch, (uint8_t *)bricks, nx, ny, nz, nbx, nby, total, brow, blayer);
}

// ---------------------------------------------------------------------------
// dense -> pitched copy of the linear volume (vv_api.cpp finalize_layout); rows are multiples of 16 bytes
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void repitch_kernel(const uint4 *__restrict__ in, char *__restrict__ out, size_t row16, size_t ny,
                                                      size_t row_pitch, size_t slice_pitch, size_t total)
{
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const size_t x = t % row16, r = t / row16, y = r % ny, z = r / ny;
        *(uint4 *)(out + z * slice_pitch + y * row_pitch + x * 16) = in[t];
    }
}

void launch_repitch(const void *dense, void *pitched, size_t row_bytes, size_t ny, size_t nz,
                    size_t row_pitch, size_t slice_pitch, hipStream_t s)
{
    const size_t row16 = row_bytes / 16, total = row16 * ny * nz;
    size_t blocks = (total + 255) / 256;
    if (blocks > 256 * 64) blocks = 256 * 64;
    if (blocks == 0) return;
    hipLaunchKernelGGL(repitch_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const uint4 *)dense, (char *)pitched, row16, ny, row_pitch, slice_pitch, total);
}

// ---------------------------------------------------------------------------
// z-pair copy of an f32 volume (VolumeView::zpair, vv_device.h)
// ---------------------------------------------------------------------------
size_t zpair_copy_bytes(int vtype, int nx, int ny, int nz, uint32_t *row_bytes, uint32_t *slab_bytes)
{
    const size_t rec = vtype == VV_VOXEL_F32 ? 8 : 2;
    size_t row = (((size_t)nx + 1) * rec + 3) & ~(size_t)3;
    if (const char *e = getenv("VV_ZPAIR_PAD")) { int t = atoi(e); if (t > 0 && t <= 4096 && t % 8 == 0) row += (size_t)t; }   // experiment knob (no effect measured)
    const size_t slab = ((size_t)ny + 1) * row;
    if (row_bytes) *row_bytes = (uint32_t)row;
    if (slab_bytes) *slab_bytes = (uint32_t)slab;
    return slab * (size_t)nz;
}

// one thread per record; rows of `row_recs` records (u8 rows may end in one unused record of padding)
template <typename T, typename T2>
__global__ __launch_bounds__(256) void zpair_kernel(const T *__restrict__ in, size_t row_pitch, size_t slice_pitch, T2 *__restrict__ out,
                                                    int nx, int ny, int nz, size_t row_recs, size_t total)
{
    const size_t ry = (size_t)ny + 1;
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const int x = min((int)(t % row_recs), nx - 1);
        const size_t row = t / row_recs;
        const int y = min((int)(row % ry), ny - 1), z = (int)(row / ry);
        const char *rowp = (const char *)in + (size_t)y * row_pitch;
        T2 r;
        r.x = ((const T *)(rowp + (size_t)z * slice_pitch))[x];
        r.y = ((const T *)(rowp + (size_t)min(z + 1, nz - 1) * slice_pitch))[x];
        out[t] = r;
    }
}

void launch_build_zpair(int vtype, const void *linear, size_t row_pitch, size_t slice_pitch, void *zpair, int nx, int ny, int nz, hipStream_t s)
{
    uint32_t rb = 0, sb = 0;
    zpair_copy_bytes(vtype, nx, ny, nz, &rb, &sb);
    const size_t row_recs = rb / (vtype == VV_VOXEL_F32 ? 8 : 2);
    const size_t total = row_recs * ((size_t)ny + 1) * (size_t)nz;
    size_t blocks = (total + 255) / 256;
    if (blocks > 256 * 64) blocks = 256 * 64;
    if (vtype == VV_VOXEL_F32)
        hipLaunchKernelGGL((zpair_kernel<float, float2>), dim3((unsigned)blocks), dim3(256), 0, s, (const float *)linear, row_pitch, slice_pitch, (float2 *)zpair, nx, ny, nz, row_recs, total);
    else
        hipLaunchKernelGGL((zpair_kernel<uint8_t, uchar2>), dim3((unsigned)blocks), dim3(256), 0, s, (const uint8_t *)linear, row_pitch, slice_pitch, (uchar2 *)zpair, nx, ny, nz, row_recs, total);
}

// ---------------------------------------------------------------------------
// z-fastest copy of a volume (VolumeView::zfast): out(x, y, z) = in(z, y, x) -- for every y a transpose of the (z, x) plane through
// 32 x 32 tiles in LDS (33 columns: no bank conflicts), reads and writes in whole lines
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void zfast_kernel(const T *__restrict__ in, uint32_t row_pitch, uint64_t slice_pitch, T *__restrict__ out,
                                                    uint32_t zf_row_bytes, uint64_t zf_slice_bytes, int nx, int ny, int nz)
{
    __shared__ T tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;              // 32 x 8 threads
    const int x0 = blockIdx.x * 32, z0 = blockIdx.y * 32, y = blockIdx.z;
    for (int r = ty; r < 32; r += 8) {
        const int z = z0 + r, x = x0 + tx;
        tile[r][tx] = (z < nz && x < nx) ? ((const T *)((const char *)in + (uint64_t)z * slice_pitch + (uint64_t)y * row_pitch))[x] : T(0);
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int x = x0 + r, z = z0 + tx;
        if (x < nx && z < nz) ((T *)((char *)out + (uint64_t)x * zf_slice_bytes + (uint64_t)y * zf_row_bytes))[z] = tile[tx][r];
    }
}

void launch_build_zfast(int vtype, const void *vol, uint32_t row_pitch, uint64_t slice_pitch, void *out, uint32_t zf_row_bytes, uint64_t zf_slice_bytes, int nx, int ny, int nz, hipStream_t s)
{
    // grid.z = ny must stay within the launch limit of 65535 (volumes may have up to 2^24 - 1 rows: vv_load_volume_*); the caller (ensure_zfast) does
    // not ask for this copy beyond it and the bricked copy serves the view
    if (ny > 65535 || (nz + 31) / 32 > 65535) return;
    const dim3 grid((unsigned)((nx + 31) / 32), (unsigned)((nz + 31) / 32), (unsigned)ny);
    if (vtype == VV_VOXEL_F32)
        hipLaunchKernelGGL(zfast_kernel<float>, grid, dim3(256), 0, s, (const float *)vol, row_pitch, slice_pitch, (float *)out, zf_row_bytes, zf_slice_bytes, nx, ny, nz);
    else
        hipLaunchKernelGGL(zfast_kernel<uint8_t>, grid, dim3(256), 0, s, (const uint8_t *)vol, row_pitch, slice_pitch, (uint8_t *)out, zf_row_bytes, zf_slice_bytes, nx, ny, nz);
}

// x-pair copy (the z-pair copy with x and z exchanged), built from the z-fastest copy so that both reads run along z:
// record (z, y, x) = { v(x, y, z), v(x+1, y, z) }, rows of nz + 1 records, slabs of ny + 1 rows, nx slabs; indices beyond the volume clamp
template <typename T, typename T2>
__global__ __launch_bounds__(256) void xpair_kernel(const T *__restrict__ zf, uint32_t zf_row_bytes, uint64_t zf_slice_bytes, T2 *__restrict__ out,
                                                    int nx, int ny, int nz, size_t row_recs, size_t total)
{
    const size_t ry = (size_t)ny + 1;
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const int z = min((int)(t % row_recs), nz - 1);
        const size_t row = t / row_recs;
        const int y = min((int)(row % ry), ny - 1), x = (int)(row / ry);
        const char *rowp = (const char *)zf + (size_t)y * zf_row_bytes;
        T2 r;
        r.x = ((const T *)(rowp + (size_t)x * zf_slice_bytes))[z];
        r.y = ((const T *)(rowp + (size_t)min(x + 1, nx - 1) * zf_slice_bytes))[z];
        out[t] = r;
    }
}

void launch_build_xpair(int vtype, const void *zfast, uint32_t zf_row_bytes, uint64_t zf_slice_bytes, void *xpair, int nx, int ny, int nz, hipStream_t s)
{
    uint32_t rb = 0, sb = 0;
    zpair_copy_bytes(vtype, nz, ny, nx, &rb, &sb);                         // (the z-pair copy's geometry with x and z exchanged)
    const size_t row_recs = rb / (vtype == VV_VOXEL_F32 ? 8 : 2);
    const size_t total = row_recs * ((size_t)ny + 1) * (size_t)nx;
    size_t blocks = (total + 255) / 256;
    if (blocks > 256 * 64) blocks = 256 * 64;
    if (vtype == VV_VOXEL_F32)
        hipLaunchKernelGGL((xpair_kernel<float, float2>), dim3((unsigned)blocks), dim3(256), 0, s, (const float *)zfast, zf_row_bytes, zf_slice_bytes, (float2 *)xpair, nx, ny, nz, row_recs, total);
    else
        hipLaunchKernelGGL((xpair_kernel<uint8_t, uchar2>), dim3((unsigned)blocks), dim3(256), 0, s, (const uint8_t *)zfast, zf_row_bytes, zf_slice_bytes, (uchar2 *)xpair, nx, ny, nz, row_recs, total);
}

// ---------------------------------------------------------------------------
// synthetic noise volume V2 (measurement input, not from the reference)
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t mix32(uint32_t h)
{
    h ^= h >> 16; h *= 0x7feb352dU; h ^= h >> 15; h *= 0x846ca68bU; h ^= h >> 16;
    return h;
}

__global__ __launch_bounds__(256) void noise_kernel(uint8_t *__restrict__ out, int nx, int ny, int nz, uint32_t seed)
{
    const size_t total = (size_t)nx * ny * nz;
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(t % nx);
        const size_t row = t / nx;
        const int y = (int)(row % ny), z = (int)(row / ny);
        uint32_t sum = 0;
        for (int dz = -1; dz <= 1; ++dz) for (int dy = -1; dy <= 1; ++dy) for (int dx = -1; dx <= 1; ++dx) {
            int xx = max(0, min(x + dx, nx - 1)), yy = max(0, min(y + dy, ny - 1)), zz = max(0, min(z + dz, nz - 1));
            uint32_t idx = (uint32_t)xx + (uint32_t)nx * ((uint32_t)yy + (uint32_t)ny * (uint32_t)zz);
            sum += mix32(idx ^ seed) >> 24;
        }
        out[t] = (uint8_t)((sum + 13u) / 27u);
    }
}

void launch_noise_u8(uint8_t *out, int nx, int ny, int nz, uint32_t seed, hipStream_t s)
{
    size_t total = (size_t)nx * ny * nz;
    size_t blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (blocks == 0) return;
    hipLaunchKernelGGL(noise_kernel, dim3((unsigned)blocks), dim3(256), 0, s, out, nx, ny, nz, seed);
}

} // namespace vv
