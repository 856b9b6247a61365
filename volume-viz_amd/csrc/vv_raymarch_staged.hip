// vv_raymarch_staged.hip -- LDS-staged slab march (no Phong) for gfx950 (MI355X).
//
// Same arithmetic, sample for sample, as march_kernel in vv_raymarch.hip (and therefore the
// same bit-exact frames); what changes is where the eight corners of a sample come from.
//
// march_kernel gathers them from HBM/L2 with four 8-byte loads per sample.  That is bound by
// the bytes the gathers pull through the fabric: every step of a wave touches a fresh thin
// patch of the volume, 128-byte lines are used once and partially, and a view that is not
// aligned with the memory axis touches a different line per lane.
//
// Here a 256-thread block owns a 32x8 pixel tile and advances it through the volume in
// *stages*.  A stage is a slab of slices perpendicular to the rays' major axis.  The block
//   1. finds the first slice any of its rays still needs and predicts, from each ray's
//      position and per-sample increment, the box (minor-axis extents) its rays cross inside
//      the slab -- two block-wide min/max reductions;
//   2. copies that axis-aligned box from the linear HBM volume into LDS with coalesced,
//      16-byte, row-contiguous loads (each voxel of the box crosses the fabric once per stage,
//      whatever the view direction);
//   3. lets every lane composite its samples whose 2x2x2 footprint lies in the box, reading
//      the corners from LDS.  A sample the prediction missed is fetched from global memory,
//      so correctness never depends on the prediction.
// Rays keep the reference's chunk / early-termination state machine per lane (kernel.cu:248-278).
#include "vv_device.h"
#include "vv_kernels.h"

namespace vv {

constexpr int kBoxBytes = 32 * 1024;          // LDS box (8 pieces of 16 B per thread); 4 blocks per CU with the 4 KB table
constexpr int kMaxSlices = 24;

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

template <int SLICE, int VOXEL, bool TEX8, bool GRAY, bool INSTR>
__global__ __launch_bounds__(256) void march_staged_kernel(FrameParams P, VolumeView V,
                                                           const float4 *__restrict__ tf,
                                                           const float *__restrict__ rad,
                                                           uint32_t *__restrict__ pixels,
                                                           unsigned long long *__restrict__ counter,
                                                           uint32_t *__restrict__ bricks, StripMap M)
{
    __shared__ __attribute__((aligned(16))) char box[kBoxBytes];
    __shared__ float lds_tf[1024];
    __shared__ int red[16];                    // reduction scratch

    constexpr uint32_t VSZ = VOXEL == VV_VOXEL_F32 ? 4u : 1u;
    constexpr int PIECE = 16 / (int)VSZ;       // voxels per 16-byte piece

    const int ntx = (P.W + 31) >> 5;
    const int strip = blockIdx.x / ntx, tile_x = blockIdx.x % ntx;
    for (int i = threadIdx.x; i < 256; i += 256) {
        float4 e = tf[i];
        lds_tf[i] = e.x; lds_tf[256 + i] = e.y; lds_tf[512 + i] = e.z; lds_tf[768 + i] = e.w;
    }
    if (threadIdx.x < 16) red[threadIdx.x] = 0;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
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
            write_zero = true;                                       // kernel.cu:334-338
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

    // ---- cursor: position the lane on its first sample (chunk 0, i = 1) ----
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

    // ---- block-uniform major axis and direction: taken from the first live lane ----
    __syncthreads();
    if (c.live) atomicMin(&red[0], -(int)(256 - threadIdx.x));       // smallest thread id wins (stored negative)
    __syncthreads();
    const int first = red[0] == 0 ? -1 : 256 + red[0];
    if (first < 0) {                                                   // no ray of this tile samples anything
        if (in_frame) pixels[(size_t)y * P.W + x] = 0u;               // result (0,0,0,0) or the zero-length early-out
        return;
    }
    if ((int)threadIdx.x == first) {
        float ax = fabsf(dqx), ay = fabsf(dqy), az = fabsf(dqz);
        int m = (ax >= ay && ax >= az) ? 0 : (ay >= az ? 1 : 2);
        float d = m == 0 ? dqx : (m == 1 ? dqy : dqz);
        red[1] = m; red[2] = d < 0.f ? -1 : 1;
    }
    __syncthreads();
    const int m = red[1], sgn = red[2];
    const int nm = m == 0 ? V.nx : (m == 1 ? V.ny : V.nz);
    __syncthreads();

    // ------------------------------------------------------------------------------------------
    // Stage pipeline.  Along the major axis a stage serves the samples whose base slice lies in
    // [bs_lo, bs_hi]; `cur` is the box in LDS for that range.  While the block composites stage
    // s from LDS, the box of stage s+1 -- predicted from where every ray will be when it leaves
    // the current slab -- is already in flight from HBM into registers; it is written to LDS
    // when the compute loop has drained.  Samples a box does not cover (mispredicted, lagging,
    // early-terminated rays that take one sample per chunk) are fetched from global memory.
    // ------------------------------------------------------------------------------------------
    constexpr int R = kBoxBytes / 16 / 256;                            // 16-byte pieces per thread and box
    const int a0 = m == 0 ? 1 : 0, a1 = m == 2 ? 1 : 2;               // volume axes of the two minor extents
    const int nn0 = a0 == 0 ? V.nx : V.ny, nn1 = a1 == 1 ? V.ny : V.nz;
    const float dm = m == 0 ? dqx : (m == 1 ? dqy : dqz);
    const float vfwd = fmaxf(fabsf(dm), 1e-6f);                        // slices per sample along the march
    const float du = m == 0 ? dqy : dqx, dv = m == 2 ? dqy : dqz;

    // first slab: starts at the first base slice any ray needs
    int bs_lo, bs_hi;                                                  // base slices the current stage serves
    {
        float tx = __builtin_fmaf(c.px - 0.5f, P.inv_scale[0], 0.5f);
        float ty = __builtin_fmaf(c.py - 0.5f, P.inv_scale[1], 0.5f);
        float tz = __builtin_fmaf(c.pz - 0.5f, P.inv_scale[2], 0.5f);
        float qm = m == 0 ? __builtin_fmaf(tx, fnx, -0.5f) : (m == 1 ? __builtin_fmaf(ty, fny, -0.5f) : __builtin_fmaf(tz, fnz, -0.5f));
        int basem = (int)floorf(fminf(fmaxf(qm, 0.f), (float)(nm - 1)));
        if (threadIdx.x == 0) red[3] = 0x7fffffff;
        __syncthreads();
        int k = wave_min(c.live ? (sgn > 0 ? basem : -basem) : 0x7fffffff);
        if (lane == 0 && k != 0x7fffffff) atomicMin(&red[3], k);
        __syncthreads();
        const int k0 = sgn > 0 ? red[3] : -red[3];
        // an empty slab just before k0: stage 0 composites the leading out-of-volume samples
        // while the first real box is being fetched
        if (sgn > 0) { bs_lo = k0; bs_hi = k0 - 1; } else { bs_hi = k0; bs_lo = k0 + 1; }
    }
    Box cur;
    cur.lox = cur.loy = cur.loz = 0; cur.nx = cur.ny = cur.nz = 0; cur.pitch = 16; cur.slice_pitch = 16;

    int dpred = 8;                                                     // slab thickness the next stage aims for
    const int max_stages = 2 * nm + P.max_chunks + 16;
    for (int stage = 0; stage < max_stages; ++stage) {
        // ---- 1. predict the next slab's box -------------------------------------------------
        float tx = __builtin_fmaf(c.px - 0.5f, P.inv_scale[0], 0.5f);
        float ty = __builtin_fmaf(c.py - 0.5f, P.inv_scale[1], 0.5f);
        float tz = __builtin_fmaf(c.pz - 0.5f, P.inv_scale[2], 0.5f);
        const float qx = __builtin_fmaf(tx, fnx, -0.5f), qy = __builtin_fmaf(ty, fny, -0.5f), qz = __builtin_fmaf(tz, fnz, -0.5f);
        const float qm = m == 0 ? qx : (m == 1 ? qy : qz);
        // nominal next slab
        int nb_lo, nb_hi;
        if (sgn > 0) { nb_lo = bs_hi + 1; nb_hi = min(nb_lo + dpred - 1, nm - 1); }
        else         { nb_hi = bs_lo - 1; nb_lo = max(nb_hi - dpred + 1, 0); }
        int lo0 = 0x7fffffff, hi0 = -0x7fffffff, lo1 = 0x7fffffff, hi1 = -0x7fffffff, kent = 0x7fffffff;
        if (c.live && !c.ert) {
            // forward coordinate F grows along the march; the current slab is served while F < Ecur
            const float F = sgn > 0 ? qm : -qm;
            const float Ecur = sgn > 0 ? (float)(bs_hi + 1) : -(float)bs_lo;
            const float Enext = sgn > 0 ? (float)(nb_hi + 1) : -(float)nb_lo;
            const float rem = fmaxf((r.upper - ((float)c.i * r.sstep + c.dist)) / r.sstep + 2.0f, 0.f);   // samples left on the ray
            float jc = fminf(fmaxf(ceilf((Ecur - F) / vfwd), 0.f), fminf(rem, 96.f));
            float jn = fminf(fmaxf(ceilf((Enext - (F + jc * vfwd)) / vfwd), 0.f), fminf(rem - jc, 96.f));
            if (jn > 0.f) {
                const float u0 = m == 0 ? qy : qx, v0 = m == 2 ? qy : qz;
                const float u1 = u0 + jc * du, v1 = v0 + jc * dv, u2 = u1 + jn * du, v2 = v1 + jn * dv;
                lo0 = (int)floorf(fminf(u1, u2)) - 1; hi0 = (int)floorf(fmaxf(u1, u2)) + 2;
                lo1 = (int)floorf(fminf(v1, v2)) - 1; hi1 = (int)floorf(fmaxf(v1, v2)) + 2;
            }
            // base slice at which the lane will stand when the current slab is done (for jumps)
            const float qe = qm + jc * dm;
            const int be = (int)floorf(fminf(fmaxf(qe, 0.f), (float)(nm - 1)));
            kent = sgn > 0 ? be : -be;
        }
        __syncthreads();                                   // everybody is done with red[] of the previous stage
        if (threadIdx.x == 0) { red[3] = 0x7fffffff; red[4] = 0x7fffffff; red[5] = -0x7fffffff; red[6] = 0x7fffffff; red[7] = -0x7fffffff; red[8] = 0; }
        __syncthreads();
        {
            lo0 = wave_min(lo0); hi0 = wave_max(hi0); lo1 = wave_min(lo1); hi1 = wave_max(hi1); kent = wave_min(kent);
            const bool wl = __any(c.live);
            if (lane == 0) {
                if (lo0 != 0x7fffffff) { atomicMin(&red[4], lo0); atomicMax(&red[5], hi0); atomicMin(&red[6], lo1); atomicMax(&red[7], hi1); }
                if (kent != 0x7fffffff) atomicMin(&red[3], kent);
                if (wl) red[8] = 1;
            }
        }
        __syncthreads();
        if (red[8] == 0) break;                            // no ray of the tile has a sample left
        // every thread derives the same next box from the reduced values
        Box nxt;
        {
            // jump ahead when every predicting ray enters beyond the nominal slab
            if (red[3] != 0x7fffffff) {
                const int ke = sgn > 0 ? red[3] : -red[3];
                if (sgn > 0 && ke > nb_lo) { nb_lo = min(ke, nm - 1); nb_hi = min(nb_lo + dpred - 1, nm - 1); }
                if (sgn < 0 && ke < nb_hi) { nb_hi = max(ke, 0); nb_lo = max(nb_hi - dpred + 1, 0); }
            }
            if (nb_lo > nb_hi) { if (sgn > 0) nb_lo = nb_hi = nm - 1; else nb_lo = nb_hi = 0; }   // range exhausted: serve everything
            // extents along the march axis (M) and the two minor axes, mapped onto x, y, z
            const int loM = nb_lo, hiM = min(nb_hi + 1, nm);
            const int loA = max(red[4], 0), hiA = min(red[5], nn0), loB = max(red[6], 0), hiB = min(red[7], nn1);
            const bool empty = red[4] == 0x7fffffff || hiA < loA || hiB < loB;
            int lox = m == 0 ? loM : loA, hix = m == 0 ? hiM : hiA;
            int loy = m == 0 ? loA : (m == 1 ? loM : loB), hiy = m == 0 ? hiA : (m == 1 ? hiM : hiB);
            int loz = m == 2 ? loM : loB, hiz = m == 2 ? hiM : hiB;
            lox &= ~(PIECE - 1);                                           // 16-byte aligned rows
            int nxp = (hix - lox + PIECE) / PIECE;                         // 16-byte pieces per row (covers hix)
            int ny_ = hiy - loy + 1, nz_ = hiz - loz + 1;
            int pitch = ((nxp & 7) == 0 ? nxp + 1 : nxp) * 16;             // never a multiple of 128 B: rows rotate banks
            if (!empty && (long)pitch * ny_ * nz_ > kBoxBytes) {           // shed slices of the major axis
                if (m == 0) {
                    const int rows = ny_ * nz_;
                    int fit = kBoxBytes / (rows * 16);                     // 16-byte units per row that fit
                    if (fit >= 8 && (fit & 7) == 0) fit -= 1;
                    if (fit * PIECE < 2) nxp = 0;
                    else {
                        nxp = fit; pitch = fit * 16;
                        if (sgn < 0) { lox = (hix + 1 - nxp * PIECE + PIECE - 1) & ~(PIECE - 1); if (lox < 0) lox = 0; }
                        if (sgn > 0) nb_hi = min(nb_hi, lox + nxp * PIECE - 2); else nb_lo = max(nb_lo, lox);   // base slices covered
                    }
                } else if (m == 1) {
                    const int fit = kBoxBytes / (pitch * nz_);             // y slices that fit
                    if (fit < 2) nxp = 0;
                    else if (sgn > 0) { hiy = loy + fit - 1; nb_hi = hiy - 1; } else { loy = hiy + 1 - fit; nb_lo = loy; }
                    ny_ = hiy - loy + 1;
                } else {
                    const int fit = kBoxBytes / (pitch * ny_);             // z slices that fit
                    if (fit < 2) nxp = 0;
                    else if (sgn > 0) { hiz = loz + fit - 1; nb_hi = hiz - 1; } else { loz = hiz + 1 - fit; nb_lo = loz; }
                    nz_ = hiz - loz + 1;
                }
            }
            if (empty) nxp = 0;
            nxt.lox = lox; nxt.loy = loy; nxt.loz = loz;
            nxt.nx = nxp * PIECE; nxt.ny = nxp ? ny_ : 0; nxt.nz = nxp ? nz_ : 0;
            nxt.pitch = pitch; nxt.slice_pitch = pitch * ny_;
            // keep at least one base slice per stage so that the march always advances
            if (nb_lo > nb_hi) { if (sgn > 0) nb_hi = nb_lo; else nb_lo = nb_hi; }
            const int got = nb_hi - nb_lo + 1;
            dpred = got >= dpred ? min(dpred + 2, kMaxSlices) : max(got, 2);
        }

        // ---- 2. issue the loads of the next box: 16-byte pieces, consecutive lanes along a row ----
        uint4 v[R];
        const int nxp_n = nxt.nx / PIECE;
        const int total_n = nxp_n * nxt.ny * nxt.nz;
        const float inv_nxp = 1.0f / (float)max(nxp_n, 1), inv_ny = 1.0f / (float)max(nxt.ny, 1);
        {
            const char *g0 = (const char *)V.data + (size_t)nxt.loz * V.slice_bytes + (size_t)nxt.loy * V.row_bytes + (size_t)nxt.lox * VSZ;
#pragma unroll
            for (int u = 0; u < R; ++u) {
                const int q = u * 256 + (int)threadIdx.x;
                const int row = (int)(((float)q + 0.5f) * inv_nxp);          // q / nxp   (exact for q < 2^16)
                const int col = q - row * nxp_n;
                const int zz = (int)(((float)row + 0.5f) * inv_ny);          // row / ny
                const int yy = row - zz * nxt.ny;
                if (q < total_n)
                    v[u] = *(const uint4 *)(g0 + (size_t)zz * V.slice_bytes + (size_t)yy * V.row_bytes + (size_t)col * 16);
            }
        }
        if (INSTR && threadIdx.x == 0) { atomicAdd(counter + 1, 1ull); atomicAdd(counter + 3, (unsigned long long)total_n * 16ull); }

        // ---- 3. composite every sample the current slab serves ----
        {
            const Box b = cur;
            const uint32_t bx = (uint32_t)b.lox, by_ = (uint32_t)b.loy, bz = (uint32_t)b.loz;
            const uint32_t mx = b.nx > 1 ? (uint32_t)(b.nx - 1) : 0u, my = b.ny > 1 ? (uint32_t)(b.ny - 1) : 0u,
                           mz = b.nz > 1 ? (uint32_t)(b.nz - 1) : 0u;
            for (int guard = 0; guard < 4096; ++guard) {
                tx = __builtin_fmaf(c.px - 0.5f, P.inv_scale[0], 0.5f);             // kernel.cu:136, DESIGN.md pin 3
                ty = __builtin_fmaf(c.py - 0.5f, P.inv_scale[1], 0.5f);
                tz = __builtin_fmaf(c.pz - 0.5f, P.inv_scale[2], 0.5f);
                const bool inb = bounds_check(tx, ty, tz);
                uint32_t ix, iy, iz;
                float wx = axis_coord<TEX8>(tx, fnx, (float)(V.nx - 1), ix);
                float wy = axis_coord<TEX8>(ty, fny, (float)(V.ny - 1), iy);
                float wz = axis_coord<TEX8>(tz, fnz, (float)(V.nz - 1), iz);
                const int im = (int)(m == 0 ? ix : (m == 1 ? iy : iz));
                // a lane takes its next sample now if it needs no data, or its base slice is not
                // beyond the slab (slices behind the slab are served from global memory)
                const bool mine = c.live && (!inb || (sgn > 0 ? im <= bs_hi : im >= bs_lo));
                if (!__any(mine)) break;
                if (INSTR && lane == 0) atomicAdd(counter + 4, 1ull);
                if (mine) {
                    uint32_t idx = 0;
                    if (inb) {
                        const uint32_t lx = ix - bx, ly = iy - by_, lz = iz - bz;
                        float L;
                        if (lx < mx && ly < my && lz < mz) L = lds_trilinear<VOXEL, TEX8>(box, b, wx, wy, wz, lx, ly, lz);
                        else { L = tex3d_raw<VOXEL, TEX8>(V, tx, ty, tz); if (INSTR) atomicAdd(counter + 2, 1ull); }   // not in the box: fetch from HBM
                        float sv = (VOXEL == VV_VOXEL_F32) ? L * 255.0f : L;
                        idx = min((uint32_t)sv, 255u);
                    }
                    float cr, cg, cb, ca;
                    ca = lds_tf[768 + idx];
                    cr = lds_tf[idx];
                    if (GRAY) { cg = cb = cr; }
                    else { cg = lds_tf[256 + idx]; cb = lds_tf[512 + idx]; }
                    if (SLICE == SLICE_PLANE) {                                      // kernel.cu:193-198
#pragma clang fp contract(off)
                        float vd = (float)c.i * r.sstep + c.dist;                    // :254
                        float vx = r.origin.x + r.dir.x * vd, vy = r.origin.y + r.dir.y * vd, vz = r.origin.z + r.dir.z * vd;
                        float d = fabsf(sn.x * (vx - sp.x) + sn.y * (vy - sp.y) + sn.z * (vz - sp.z));
                        if (d < .01f) cr = fmaxf(0.f, fminf(cr + (.01f - d) * 100.f, 1.f));
                    }
                    if (INSTR) {
                        executed++;
                        if (bricks && inb) mark_bricks(bricks, V, tx, ty, tz);
                    }
                    if (ca > kEps) {                                                 // :268-270, blend :107-118
#pragma clang fp contract(off)
                        float bf = ca * (1.f - res_a);
                        res_r = res_r + cr * bf;
                        if (!GRAY) { res_g = res_g + cg * bf; res_b = res_b + cb * bf; }
                        res_a = res_a + bf;
                    }
                    bool end_chunk = c.i >= c.n;
                    if (res_a > P.ert_thr) {                                         // :272-274
                        c.ert = true; end_chunk = true;
                        if (P.ert_true) r.upper = -1.f;
                    }
                    if (!end_chunk) {
                        c.i++;
                        c.px += r.sdir.x; c.py += r.sdir.y; c.pz += r.sdir.z;        // :141
                    } else {
#pragma clang fp contract(off)
                        c.dist += r.sstep * kChunkSteps;                             // :277
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

        // ---- 4. the slab is drained: park the prefetched box in LDS ----
        __syncthreads();
#pragma unroll
        for (int u = 0; u < R; ++u) {
            const int q = u * 256 + (int)threadIdx.x;
            const int row = (int)(((float)q + 0.5f) * inv_nxp);
            const int col = q - row * nxp_n;
            const int zz = (int)(((float)row + 0.5f) * inv_ny);
            const int yy = row - zz * nxt.ny;
            if (q < total_n) *(uint4 *)(box + zz * nxt.slice_pitch + yy * nxt.pitch + col * 16) = v[u];
        }
        cur = nxt; bs_lo = nb_lo; bs_hi = nb_hi;
        __syncthreads();
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
template <int SLICE, int VOXEL, bool TEX8, bool GRAY, bool INSTR>
static void launch_staged(const MarchArgs &a, hipStream_t s)
{
    const int ntx = (a.P.W + 31) / 32;
    dim3 grid((unsigned)(a.strips.n_strips * ntx));
    hipLaunchKernelGGL((march_staged_kernel<SLICE, VOXEL, TEX8, GRAY, INSTR>), grid, dim3(256), 0, s,
                       a.P, a.V, a.tf, a.rad, a.pixels, a.counter, a.bricks, a.strips);
}
template <int SLICE, int VOXEL, bool TEX8>
static void sdispatch3(const MarchArgs &a, hipStream_t s)
{
    const bool gray = a.gray && SLICE != SLICE_PLANE;
    if (gray) { if (a.instr) launch_staged<SLICE, VOXEL, TEX8, true, true>(a, s); else launch_staged<SLICE, VOXEL, TEX8, true, false>(a, s); }
    else      { if (a.instr) launch_staged<SLICE, VOXEL, TEX8, false, true>(a, s); else launch_staged<SLICE, VOXEL, TEX8, false, false>(a, s); }
}
template <int SLICE>
static void sdispatch2(const MarchArgs &a, hipStream_t s)
{
    if (a.V_type == VV_VOXEL_F32) { if (a.tex8) sdispatch3<SLICE, VV_VOXEL_F32, true>(a, s); else sdispatch3<SLICE, VV_VOXEL_F32, false>(a, s); }
    else                          { if (a.tex8) sdispatch3<SLICE, VV_VOXEL_U8,  true>(a, s); else sdispatch3<SLICE, VV_VOXEL_U8,  false>(a, s); }
}

void launch_raymarch_staged(const MarchArgs &a, hipStream_t s)
{
    switch (a.P.slice_type) {
    case SLICE_PLANE:     sdispatch2<SLICE_PLANE>(a, s); break;
    case SLICE_PLANE_CUT: sdispatch2<SLICE_PLANE_CUT>(a, s); break;
    default:              sdispatch2<SLICE_NONE>(a, s); break;
    }
}

} // namespace vv
