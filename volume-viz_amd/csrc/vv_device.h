// vv_device.h -- device-side building blocks shared by the HIP kernels (gfx950 only).
//
// Reference semantics restated here (citations relative to jacobstern/volume-viz):
//   ray end points      kernel.cu:317-321, firstpass.vert:6, glwidget.cpp:198-228,338
//   ray set-up          kernel.cu:323-350, implicit.cu:20-35
//   plane clipping      kernel.cu:234-246, implicit.cu:4-17,38-47
//   texture unit        kernel.cu:46,99-105,485-489 (tex3D linear/clamp/normalised)
//
// Arithmetic that feeds per-ray set-up is written without FMA contraction so that it
// reproduces the strict-float order of the source; the per-sample texture model uses
// explicit fmaf (the reference does this in hardware).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/volviz.h"

namespace vv {

constexpr int   kCacheDepth = 32;          // kernel.cu:24
constexpr float kChunkSteps = 30.f;        // kernel.cu:25
constexpr int   kSlab = 14;                // BLOCK_WIDTH-2, kernel.cu:30,418
constexpr int   kBlock = 16;               // kernel.cu:30-31
constexpr float kSqrt3 = 1.73205081f;      // kernel.cu:33
// float thresholds equivalent to the reference's double-literal comparisons:
//   (double)f > 1e-6  <=>  f >  kEps   ;  (double)f < 1e-6  <=>  f <= kEps
//   (double)t > -1e-6 <=>  t >= -kEps      (kEps = (float)1e-6 = 0x358637BD < 1e-6)
constexpr float kEps = 1e-6f;

struct f3 { float x, y, z; };
__device__ __forceinline__ f3 mk3(float x, float y, float z) { return f3{x, y, z}; }

// Volume as laid out in HBM: linear, x fastest, rows/slices contiguous, with one
// row + one slice + 16 bytes of zero padding past the end so that the (weight-0)
// upper corner of an edge sample may be fetched without clamping the index.
struct VolumeView {
    const void *data;
    int nx, ny, nz;
    uint32_t row_bytes;     // nx * sizeof(voxel)
    uint32_t slice_bytes;   // nx*ny * sizeof(voxel)
    int big;                // volume > 4 GiB or slice >= 16 MiB: 64-bit slice base per sample (BIG kernels)
    // Optional second copy in 4x4x4-voxel bricks for views that are not aligned with the x axis
    // (built on first use, vv_api.cpp).  A brick is 16 rows (y fastest, then z) of 4 voxels + 1 halo
    // voxel (= the next voxel in x, clamped): f32 rows are 20 bytes (brick 320 B), u8 rows are padded
    // to 8 bytes (brick 128 B = one cache line).  Bricks are x fastest.  The grid has one extra brick
    // layer in y and z when needed so that y+1 / z+1 always exist (clamped content), which keeps
    // the sampler free of index clamps like the linear layout.
    const void *bricks;
    uint32_t b_sy;          // bytes per row of bricks   (nbx * brick bytes)
    uint32_t b_sz64;        // bytes per layer of bricks (nby * b_sy) in 64-byte units
    // Optional z-pair copy for views along the memory axis (built on first use):
    // record (x, y, z) = { v(x,y,z), v(x,y,z+1) } (8 bytes for f32, 2 bytes for u8), x fastest, rows of nx+1 records, slabs of
    // ny+1 rows, nz slabs; indices beyond the volume clamp.  The two x-neighbouring records of a row
    // are the four corners (x..x+1, y, z..z+1) of a sample: one 16-byte gather per row instead of
    // two 8-byte gathers, i.e. 2 gathers per sample instead of 4 (a wave-wide gather costs the same
    // 16 cycles for 8 and for 16 bytes per lane, profiles/EXPERIMENTS.md part B section 4).  2x the volume in HBM.
    const void *zpair;
    uint32_t zp_row_bytes;  // (nx+1) records, u8 rows rounded up to 4 bytes
    uint32_t zp_slab_bytes; // (ny+1) * zp_row_bytes
    // Optional z-fastest copy for views whose screen x runs along the volume's z axis (side views; built on first use):
    // voxel (x, y, z) at zfast + x * zf_slice_bytes + y * zf_row_bytes + z * sizeof(voxel), padded like the linear layout.  The lanes of a wave tile then
    // read consecutive z, as they read consecutive x of the linear layout in a front view: the same kernel, the same time.
    const void *zfast;
    uint32_t zf_row_bytes;  // nz * 4 (+ padding)
    uint64_t zf_slice_bytes; // ny * zf_row_bytes
};
template <int VOXEL> struct BrickGeom;
#ifndef VV_BRICK_XLOG2
#define VV_BRICK_XLOG2 2          // f32 bricks are (1 << VV_BRICK_XLOG2) voxels long in x (experiment knob)
#endif
#ifndef VV_BRICK_HALO
#define VV_BRICK_HALO 1           // f32 rows end in a copy of the next voxel in x (0: whole-line bricks, the x pair is two gathers)
#endif
#ifndef VV_BRICK_ZLOG2
#define VV_BRICK_ZLOG2 2          // f32 bricks are (1 << VV_BRICK_ZLOG2) voxels deep in z (experiment knob: 3 = 4x4x8 bricks, profiles/r04_brick_shape.txt)
#endif
#ifndef VV_BRICK_X7
// 1: f32 bricks 7 voxels + 1 halo voxel long in x: rows of 32 bytes, a z-layer of a brick (4 rows) is exactly one 128-byte line, bricks of 512 bytes.
// Measured in round 5 (profiles/r05_brick_aligned.txt) and NOT used: 25-40 % slower than the 320-byte bricks on oblique views, cache-resident volumes
// included, at any brick stride or depth: with aligned layers the z and z + 1 rows of a sample are always two lines (L1 misses + 40 %), whereas the four
// gathers of a sample span 108 bytes of a 320-byte brick and mostly hit one line.
#define VV_BRICK_X7 0
#endif
#ifndef VV_BRICK_EXTRA
#define VV_BRICK_EXTRA 0          // unused bytes behind each f32 brick (experiment knob: brick strides that are not a power of two)
#endif
#if VV_BRICK_X7
template <> struct BrickGeom<VV_VOXEL_F32> { static constexpr uint32_t xlog2 = 0, bx = 7, halo = 1, row = 32,
                                             zlog2 = VV_BRICK_ZLOG2, bz = 1u << zlog2, rows = 4 * bz, brick = rows * row + VV_BRICK_EXTRA; };
#else
template <> struct BrickGeom<VV_VOXEL_F32> { static constexpr uint32_t xlog2 = VV_BRICK_XLOG2, bx = 1u << xlog2, halo = VV_BRICK_HALO, row = (bx + halo) * 4,
                                             zlog2 = VV_BRICK_ZLOG2, bz = 1u << zlog2, rows = 4 * bz, brick = rows * row; };
#endif
// voxel x -> byte offset of its brick in a row of bricks + of the voxel in its brick row.  Bricks of 7: x / 7 through the float reciprocal rounded up
// (0x3E124925 > 1/7): fl(x * r) >= x / 7 for every x, and it stays below the next integer while x / 7 * 2^-23 < 1/7, i.e. for every x a volume can have
// here (brick_copy_bytes refuses nx >= 2^20)
template <int VOXEL>
__device__ __forceinline__ uint32_t brick_x_offset(uint32_t ix)
{
    using G = BrickGeom<VOXEL>;
    if constexpr (VOXEL == VV_VOXEL_F32 && G::bx == 7) {
        const uint32_t q = (uint32_t)((float)ix * 0.142857149f);
        return __umul24(q, G::brick - 28u) + (ix << 2);                 // q * brick + (ix - 7 q) * 4
    } else if constexpr (VOXEL == VV_VOXEL_F32) {
        return __umul24(ix >> G::xlog2, G::brick) + ((ix & (G::bx - 1u)) << 2);
    } else {
        return (ix >> 2) * G::brick;
    }
}
template <> struct BrickGeom<VV_VOXEL_U8>  { static constexpr uint32_t xlog2 = 2, bx = 4, halo = 1, row = 8, zlog2 = 2, bz = 4, rows = 16, brick = 128; };
enum { LAYOUT_LINEAR = 0, LAYOUT_LINEAR_BIG = 1, LAYOUT_BRICKED = 2, LAYOUT_ZPAIR = 3, LAYOUT_ZFAST = 4 };

// Everything a frame needs that is uniform over the launch.
struct FrameParams {
    int W, H;
    int nbx, nby;                 // slab grid, kernel.cu:418-425
    int conflict_x, conflict_y;   // W-1 == 14*(nbx-1) (resp. H): last block re-writes a pixel
    // slab-row shard predicate (include/volviz.h vv_render_options): a pixel row y is rendered
    // iff r = y/14 satisfies  rb <= r < re  &&  (r / band) % count == index
    int rb, re, band, count, index;
    int slice_type;
    float slice_point[3], slice_normal[3];
    float cam_pos[3], scale[3], inv_scale[3];
    float step[3];
    float tan_fov_x, tan_fov_y;   // kernel.cu:221-222 (computed in double on the host)
    float ert_thr;
    int   ert_true;
    int   alpha_unit;             // every table opacity in [0, 1] (a ray past the threshold stays past it)
    int   max_chunks;             // hard bound on the chunk loop (every wave exits)
    int   safe_div;               // Phong: pixel tangents in [2^-24, 2^8] and steps <= 16, so the gradient's divisions and square root
                                  // stay far inside the normal range and need no range handling (vv_raymarch_phong2.h)
    // ray source
    int ray_mode, quantize8;
    const uint8_t *front_img, *back_img; int img_w, img_h;
    float side[3], up[3], look[3];   // orthonormal camera basis (camera.cpp:78-91)
    float tan_half_x, tan_half_y;    // tan(fovY/2)*aspect, tan(fovY/2)
};

__device__ __forceinline__ int wave_min_i(int v) { for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o)); return v; }
__device__ __forceinline__ int wave_max_i(int v) { for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o)); return v; }

// ---------------------------------------------------------------------------
// strict (uncontracted) helpers for per-ray set-up
// ---------------------------------------------------------------------------
#pragma clang fp contract(off)
__device__ __forceinline__ float vlen3(float x, float y, float z)
{
#pragma clang fp contract(off)
    return sqrtf(x * x + y * y + z * z);       // kernel.cu:53-57
}
__device__ __forceinline__ float dot3s(f3 a, f3 b)
{
#pragma clang fp contract(off)
    return a.x * b.x + a.y * b.y + a.z * b.z;
}

// Pixel -> front/back end points.  Mirrors oracle analytic_endpoints / image_endpoints.
__device__ __forceinline__ void ray_endpoints(const FrameParams &P, int x, int y, f3 &front, f3 &back)
{
#pragma clang fp contract(off)
    if (P.ray_mode == VV_RAYS_IMAGES) {
        // kernel.cu:317-321: tex2D point sampling at normalised (x/W, y/H)
        float u = (float)x / (float)P.W, v = (float)y / (float)P.H;
        int tx = (int)floorf(u * (float)P.img_w), ty = (int)floorf(v * (float)P.img_h);
        tx = max(0, min(tx, P.img_w - 1)); ty = max(0, min(ty, P.img_h - 1));
        // one dword per texel (the images are 4-byte aligned: vv_render checks)
        const size_t t = (size_t)ty * P.img_w + tx;
        const uint32_t f = ((const uint32_t *)P.front_img)[t], b = ((const uint32_t *)P.back_img)[t];
        front = mk3((float)(f & 255u) / 255.f, (float)((f >> 8) & 255u) / 255.f, (float)((f >> 16) & 255u) / 255.f);
        back  = mk3((float)(b & 255u) / 255.f, (float)((b >> 8) & 255u) / 255.f, (float)((b >> 16) & 255u) / 255.f);
        return;
    }
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
    front = mk3(0.f, 0.f, 0.f); back = mk3(0.f, 0.f, 0.f);
    if (miss || !(tmin <= tmax) || !(tmax > 0.0f)) return;
    back = mk3((o[0] + d[0] * tmax) * 0.5f + 0.5f, (o[1] + d[1] * tmax) * 0.5f + 0.5f,
               (o[2] + d[2] * tmax) * 0.5f + 0.5f);
    if (tmin > 0.0f)
        front = mk3((o[0] + d[0] * tmin) * 0.5f + 0.5f, (o[1] + d[1] * tmin) * 0.5f + 0.5f,
                    (o[2] + d[2] * tmin) * 0.5f + 0.5f);
    if (P.quantize8) {
        float *p[6] = {&front.x, &front.y, &front.z, &back.x, &back.y, &back.z};
        for (int i = 0; i < 6; i++) {
            float c = fmaxf(0.f, fminf(*p[i], 1.f));
            *p[i] = floorf(c * 255.0f + 0.5f) / 255.f;
        }
    }
}

// Slab footprint of block b along one axis (kernel.cu:297-302) and the write-owner
// rule (oracle pin 10).
__device__ __host__ __forceinline__ int slab_lo(int b) { int v = b * kSlab - 1; return v < 0 ? 0 : v; }
__device__ __host__ __forceinline__ int slab_up(int b, int n) { int v = (b + 1) * kSlab + 1; return v > n - 1 ? n - 1 : v; }
__device__ __host__ __forceinline__ int owner_slab(int p, int n, int nb, int conflict)
{
    return (conflict && p == n - 2) ? nb - 1 : p / kSlab;
}

__device__ __host__ __forceinline__ bool row_owned(const FrameParams &P, int y)
{
    int r = y / kSlab;
    return r >= P.rb && r < P.re && ((r / P.band) % P.count) == P.index;
}

struct Ray {
    f3 origin, dir, sdir;
    float sstep, upper, dist0;
    bool cut_return;
};

// kernel.cu:331-350 + the head of mainLoop (kernel.cu:228-246)
__device__ __forceinline__ void setup_ray(const FrameParams &P, f3 front, f3 back, float rad, Ray &r)
{
#pragma clang fp contract(off)
    f3 cam = mk3(P.cam_pos[0], P.cam_pos[1], P.cam_pos[2]);
    float dx = back.x - front.x, dy = back.y - front.y, dz = back.z - front.z;
    float length = vlen3(dx, dy, dz);
    f3 ray = mk3(dx / length, dy / length, dz / length);          // :340
    f3 pos = front;
    {   // implicit.cu:20-35 with l = -ray
        f3 l = mk3(ray.x * -1.f, ray.y * -1.f, ray.z * -1.f);
        f3 l0p0 = mk3(front.x - cam.x, front.y - cam.y, front.z - cam.z);
        float b = dot3s(l, l0p0);
        float c = dot3s(l0p0, l0p0) - rad * rad;
        float disc = b * b - c;
        if (disc >= 0.f) {
            float t = b * -1.f - sqrtf(disc);
            if (t >= -kEps) pos = mk3(pos.x - ray.x * t, pos.y - ray.y * t, pos.z - ray.z * t);  // :347
        }
    }
    float upper = fminf(kSqrt3, vlen3(back.x - pos.x, back.y - pos.y, back.z - pos.z));   // :350
    r.origin = pos; r.dir = ray;
    r.sdir = mk3(ray.x * P.step[0], ray.y * P.step[1], ray.z * P.step[2]);                // :228
    r.sstep = vlen3(r.sdir.x, r.sdir.y, r.sdir.z);                                        // :229
    r.dist0 = 0.f; r.cut_return = false;
    if (P.slice_type == SLICE_PLANE_CUT) {                                                // :234-246
        f3 p0 = mk3(P.slice_point[0], P.slice_point[1], P.slice_point[2]);
        f3 n  = mk3(P.slice_normal[0], P.slice_normal[1], P.slice_normal[2]);
        f3 bk = mk3(pos.x + ray.x * upper, pos.y + ray.y * upper, pos.z + ray.z * upper);
        float sf = dot3s(n, mk3(pos.x - p0.x, pos.y - p0.y, pos.z - p0.z));
        float sb = dot3s(n, mk3(bk.x - p0.x, bk.y - p0.y, bk.z - p0.z));
        if (sf <= kEps && sb <= kEps) {
            r.cut_return = true;
        } else {
            float denom = dot3s(n, ray);
            bool hit = false;
            if (denom > kEps) {                                   // implicit.cu:4-17 (origin, ray)
                float t = dot3s(mk3(p0.x - pos.x, p0.y - pos.y, p0.z - pos.z), n) / denom;
                if (t >= 0.f) { r.dist0 = t; hit = true; }
            }
            if (!hit) {
                f3 nr = mk3(ray.x * -1.f, ray.y * -1.f, ray.z * -1.f);
                float den2 = dot3s(n, nr);
                if (den2 > kEps) {                                // (back, -ray)
                    float t = dot3s(mk3(p0.x - bk.x, p0.y - bk.y, p0.z - bk.z), n) / den2;
                    if (t >= 0.f) upper -= t;
                }
            }
        }
    }
    r.upper = upper;
}

// Number of inner-loop iterations i = 1..30 that pass `i*sstep + dist <= upper`
// (kernel.cu:253-257), evaluated with the reference's own float expression so the
// count is exact; the predicate is monotone in i.
__device__ __forceinline__ int chunk_count(float dist, float upper, float sstep)
{
#pragma clang fp contract(off)
    if (!(dist < upper)) return 0;                  // while (dist < upper), kernel.cu:248
    // the predicate must hold for i to be counted; NaN compares false -> stop
    float est = (upper - dist) / sstep;
    int n = est >= 30.f ? 30 : (est > 0.f ? (int)est : 0);
    while (n < 30 && !((float)(n + 1) * sstep + dist > upper)) ++n;
    while (n > 0 && ((float)n * sstep + dist > upper)) --n;
    // NaN step: predicate `voxelDist > upper` is false for every i -> all 30 run
    if (sstep != sstep) n = 30;
    return n;
}
#pragma clang fp contract(fast)

// ---------------------------------------------------------------------------
// texture unit model (mirrors oracle tex3d_raw)
// ---------------------------------------------------------------------------
typedef float  __attribute__((ext_vector_type(2), aligned(4))) float2u;   // 4-byte aligned pair
typedef unsigned short __attribute__((aligned(1))) ushort_u;

template <bool TEX8>
__device__ __forceinline__ float axis_coord(float x, float n, float nm1, uint32_t &i)
{
    float xb = __builtin_fmaf(x, n, -0.5f);
    // clamp addressing: for xb < 0 the result is texel 0, for xb >= n-1 texel n-1;
    // clamping the coordinate gives the same value with weight 0.
    xb = __builtin_amdgcn_fmed3f(xb, 0.0f, nm1);
    // xb >= 0: truncation is floor (one v_cvt_u32_f32) and v_fract_f32 is xb - floor(xb) exactly
    i = (uint32_t)xb;
    float a = __builtin_amdgcn_fractf(xb);
    if (TEX8) {
        // round to 8 fractional bits, ties to even: adding 1.5*2^15 makes ulp = 2^-8
        const float magic = 49152.0f;
        float t = a + magic;
        a = t - magic;
    }
    return a;
}

// A sample in flight: the raw x-pair loads of its four (y,z) rows plus the interpolation
// weights.  Fetching (address arithmetic + loads) and finishing (conversion + lerps) are
// separate so that a kernel can issue the loads of several samples back to back and only
// then consume them -- the compiler does not reliably do that on its own, and on MI355X it
// is worth 25 % (profiles/EXPERIMENTS.md part B section 4).
template <int VOXEL> struct Corners;
template <> struct Corners<VV_VOXEL_F32> { float2u a, b, c, d; float wx, wy, wz; };
template <> struct Corners<VV_VOXEL_U8>  { uint32_t a0, a1, b0, b1, c0, c1, d0, d1, sh; float wx, wy, wz; };

template <int VOXEL>
__device__ __forceinline__ void load_rows(const char *b00, uint32_t row_bytes, uint32_t slice_bytes, uint32_t ix,
                                          uint32_t yz, Corners<VOXEL> &C)
{
    const char *b10 = b00 + row_bytes;
    const char *b01 = b00 + slice_bytes;
    const char *b11 = b01 + row_bytes;
    if constexpr (VOXEL == VV_VOXEL_F32) {
        const uint32_t off = ix * 4u + yz;
        C.a = *(const float2u *)(b00 + off); C.b = *(const float2u *)(b10 + off);
        C.c = *(const float2u *)(b01 + off); C.d = *(const float2u *)(b11 + off);
    } else {
        // u8: two aligned dwords per row (8 voxels from x & ~3) cost less in the address/L1
        // pipeline than one unaligned 2-byte gather: neighbouring lanes share the dwords.
        const uint32_t off = (ix & ~3u) + yz;
        C.sh = ix & 3u;
        C.a0 = *(const uint32_t *)(b00 + off); C.a1 = *(const uint32_t *)(b00 + off + 4);
        C.b0 = *(const uint32_t *)(b10 + off); C.b1 = *(const uint32_t *)(b10 + off + 4);
        C.c0 = *(const uint32_t *)(b01 + off); C.c1 = *(const uint32_t *)(b01 + off + 4);
        C.d0 = *(const uint32_t *)(b11 + off); C.d1 = *(const uint32_t *)(b11 + off + 4);
    }
}

// texture coordinates -> weights + loads in flight (no use of the loaded data)
template <int VOXEL, bool TEX8, int LAYOUT = LAYOUT_LINEAR>
__device__ __forceinline__ void fetch_corners(const VolumeView &V, float px, float py, float pz, Corners<VOXEL> &C)
{
    uint32_t ix, iy, iz;
    C.wx = axis_coord<TEX8>(px, (float)V.nx, (float)(V.nx - 1), ix);
    C.wy = axis_coord<TEX8>(py, (float)V.ny, (float)(V.ny - 1), iy);
    C.wz = axis_coord<TEX8>(pz, (float)V.nz, (float)(V.nz - 1), iz);
    // row/slice offsets: 24-bit multiplies are full rate (v_mul_lo_u32 is quarter rate); indices
    // and row_bytes are always < 2^24; volumes whose slice_bytes is not take the BIG path.  The layout
    // is a compile-time choice: a run-time branch here cost 4 % (view along z) to 23 % (rotated view).
    if constexpr (LAYOUT == LAYOUT_LINEAR) {
        // up to 4 GiB: one 32-bit byte offset per sample added to four scalar bases (saddr + voffset)
        load_rows<VOXEL>((const char *)V.data, V.row_bytes, V.slice_bytes, ix,
                         __umul24(iy, V.row_bytes) + __umul24(iz, V.slice_bytes), C);
    } else if constexpr (LAYOUT == LAYOUT_LINEAR_BIG) {
        // volumes above 4 GiB (separate kernel instantiations): one 64-bit address per lane for the
        // corner row, the three others are 64-bit additions of the uniform row / slice pitches
        const uint32_t vsz = VOXEL == VV_VOXEL_F32 ? 4u : 1u;
        const char *zb = (const char *)V.data + (uint64_t)iz * V.slice_bytes + (__umul24(iy, V.row_bytes) + ix * vsz);
        if constexpr (VOXEL == VV_VOXEL_F32) load_rows<VOXEL>(zb, V.row_bytes, V.slice_bytes, 0u, 0u, C);
        else {
            // u8: load_rows wants the x index for its aligned-dword trick
            const char *zr = (const char *)V.data + (uint64_t)iz * V.slice_bytes;
            load_rows<VOXEL>(zr, V.row_bytes, V.slice_bytes, ix, __umul24(iy, V.row_bytes), C);
        }
    } else if constexpr (LAYOUT == LAYOUT_ZFAST) {
        // z-fastest copy: the loads of the linear layout with z in the role of x -- rows (x, y), (x, y+1), (x+1, y), (x+1, y+1), each holding
        // the voxels z and z+1 (f32: one pair; u8: two aligned dwords, shifted in finish_corners) -- are the same eight corners.  f32 corners
        // are handed on under the names the x-pair loads give them; u8 corners are renamed where the bytes are taken apart (VV_ZFAST there).
        const char *xb = (const char *)V.zfast + (uint64_t)ix * V.zf_slice_bytes;
        if constexpr (VOXEL == VV_VOXEL_F32) {
            Corners<VOXEL> T;
            load_rows<VOXEL>(xb + (__umul24(iy, V.zf_row_bytes) + iz * 4u), V.zf_row_bytes, (uint32_t)V.zf_slice_bytes, 0u, 0u, T);
            // T.a = (c000, c001), T.b = (c010, c011), T.c = (c100, c101), T.d = (c110, c111)
            C.a.x = T.a.x; C.a.y = T.c.x; C.b.x = T.b.x; C.b.y = T.d.x;      // a = (c000, c100), b = (c010, c110)
            C.c.x = T.a.y; C.c.y = T.c.y; C.d.x = T.b.y; C.d.y = T.d.y;      // c = (c001, c101), d = (c011, c111)
        } else {
            load_rows<VOXEL>(xb, V.zf_row_bytes, (uint32_t)V.zf_slice_bytes, iz, __umul24(iy, V.zf_row_bytes), C);
        }
    } else {
        // bricked copy: the rows y / y+1 and the slices z / z+1 of a sample sit in the same brick
        // unless (y & 3) == 3 resp. (z & 3) == 3; the x pair always does (halo voxel)
        using G = BrickGeom<VOXEL>;
        const uint32_t ya = iy & 3u, za = iz & (G::bz - 1u);
        const uint32_t oy0 = __umul24(iy >> 2, V.b_sy) + __umul24(ya, G::row);
        const uint32_t oy1 = oy0 + (ya == 3u ? V.b_sy - 3u * G::row : G::row);
        const uint32_t m0  = __umul24(iz >> G::zlog2, V.b_sz64);          // layer offset, 64-byte units
        const uint32_t m1  = m0 + (za == G::bz - 1u ? V.b_sz64 : 0u);
        const uint32_t zi0 = __umul24(za, 4u * G::row), zi1 = za == G::bz - 1u ? 0u : zi0 + 4u * G::row;
        const char *L0 = (const char *)V.bricks + ((uint64_t)m0 << 6);
        const char *L1 = (const char *)V.bricks + ((uint64_t)m1 << 6);
        if constexpr (VOXEL == VV_VOXEL_F32) {
            const uint32_t ox = brick_x_offset<VOXEL>(ix);
            const uint32_t o0 = ox + oy0, o1 = ox + oy1;
            if constexpr (G::halo != 0) {
                C.a = *(const float2u *)(L0 + (o0 + zi0)); C.b = *(const float2u *)(L0 + (o1 + zi0));
                C.c = *(const float2u *)(L1 + (o0 + zi1)); C.d = *(const float2u *)(L1 + (o1 + zi1));
            } else {
                // whole-line bricks: voxel x+1 is the next dword, or the first of the same row in the next brick
                const uint32_t dx = (ix & (G::bx - 1u)) == G::bx - 1u ? G::brick - (G::bx - 1u) * 4u : 4u;
                const char *pa = L0 + (o0 + zi0), *pb = L0 + (o1 + zi0), *pc = L1 + (o0 + zi1), *pd = L1 + (o1 + zi1);
                const float a0 = *(const float *)pa, b0 = *(const float *)pb, c0 = *(const float *)pc, d0 = *(const float *)pd;
                const float a1 = *(const float *)(pa + dx), b1 = *(const float *)(pb + dx), c1 = *(const float *)(pc + dx), d1 = *(const float *)(pd + dx);
                C.a = {a0, a1}; C.b = {b0, b1}; C.c = {c0, c1}; C.d = {d0, d1};
            }
        } else {
            // the whole 8-byte row (voxels 4*(x>>2) .. +4, then padding) in one aligned load;
            // finish_corners shifts voxel x to byte 0 exactly as for the linear layout
            const uint32_t ox = (ix >> 2) * G::brick;
            const uint32_t o0 = ox + oy0, o1 = ox + oy1;
            C.sh = ix & 3u;
            const uint2 ra = *(const uint2 *)(L0 + (o0 + zi0)), rb = *(const uint2 *)(L0 + (o1 + zi0));
            const uint2 rc = *(const uint2 *)(L1 + (o0 + zi1)), rd = *(const uint2 *)(L1 + (o1 + zi1));
            C.a0 = ra.x; C.a1 = ra.y; C.b0 = rb.x; C.b1 = rb.y; C.c0 = rc.x; C.c1 = rc.y; C.d0 = rd.x; C.d1 = rd.y;
        }
    }
}

// ---- z-pair copy (f32): two 16-byte gathers per sample ----
typedef float __attribute__((ext_vector_type(4), aligned(8))) float4u;    // 8-byte aligned quad
typedef float __attribute__((ext_vector_type(2))) float2v;
// naming: c<x><y><z>.  Plain floats (not the loaded vectors) so that the struct stays in registers.
struct CornersZ { float wx, c000, c001, c100, c101, wy, c010, c011, c110, c111, wz; };

template <bool TEX8>
__device__ __forceinline__ void fetch_corners_zpair(const VolumeView &V, float px, float py, float pz, CornersZ &C)
{
    uint32_t ix, iy, iz;
    C.wx = axis_coord<TEX8>(px, (float)V.nx, (float)(V.nx - 1), ix);
    C.wy = axis_coord<TEX8>(py, (float)V.ny, (float)(V.ny - 1), iy);
    C.wz = axis_coord<TEX8>(pz, (float)V.nz, (float)(V.nz - 1), iz);
#ifdef VV_XPAIR
    // x-pair copy (side views): the z-pair copy with x and z in each other's roles -- records {v(x), v(x+1)}, z fastest, slabs along x --
    // read through the same fields of the view; the two z-neighbouring records of a row are the corners (x..x+1, y, z..z+1)
    const char *zb = (const char *)V.zpair + (uint64_t)ix * V.zp_slab_bytes;
    const uint32_t off = __umul24(iy, V.zp_row_bytes) + (iz << 3);
    const float4u r0 = *(const float4u *)(zb + off);                         // (x..x+1, y, z), (x..x+1, y, z+1)
    const float4u r1 = *(const float4u *)(zb + off + V.zp_row_bytes);        // the same for y+1
    C.c000 = r0.x; C.c100 = r0.y; C.c001 = r0.z; C.c101 = r0.w;
    C.c010 = r1.x; C.c110 = r1.y; C.c011 = r1.z; C.c111 = r1.w;
#else
    const char *zb = (const char *)V.zpair + (uint64_t)iz * V.zp_slab_bytes;      // the copy of a 1024^3 volume is 8.6 GB
    const uint32_t off = __umul24(iy, V.zp_row_bytes) + (ix << 3);
    const float4u r0 = *(const float4u *)(zb + off);                         // (x, y, z..z+1), (x+1, y, z..z+1)
    const float4u r1 = *(const float4u *)(zb + off + V.zp_row_bytes);        // the same for y+1
    C.c000 = r0.x; C.c001 = r0.y; C.c100 = r0.z; C.c101 = r0.w;
    C.c010 = r1.x; C.c011 = r1.y; C.c110 = r1.z; C.c111 = r1.w;
#endif
}

// u8 z-pair copy: records of 2 bytes, one (2-byte aligned) dword per row = (c000, c001, c100, c101).
// Copies up to 4 GiB: 32-bit offsets on a scalar base.
struct CornersZ8 { float wx; uint32_t r0; float wy; uint32_t r1; float wz; };
typedef uint32_t __attribute__((aligned(2))) uint32_a2;

template <bool TEX8>
__device__ __forceinline__ void fetch_corners_zpair(const VolumeView &V, float px, float py, float pz, CornersZ8 &C)
{
    uint32_t ix, iy, iz;
    C.wx = axis_coord<TEX8>(px, (float)V.nx, (float)(V.nx - 1), ix);
    C.wy = axis_coord<TEX8>(py, (float)V.ny, (float)(V.ny - 1), iy);
    C.wz = axis_coord<TEX8>(pz, (float)V.nz, (float)(V.nz - 1), iz);
    const char *b0 = (const char *)V.zpair, *b1 = b0 + V.zp_row_bytes;
#ifdef VV_XPAIR
    const uint32_t off = ix * V.zp_slab_bytes + __umul24(iy, V.zp_row_bytes) + (iz << 1);      // dword = (c000, c100, c001, c101)
#else
    const uint32_t off = iz * V.zp_slab_bytes + __umul24(iy, V.zp_row_bytes) + (ix << 1);
#endif
    C.r0 = *(const uint32_a2 *)(b0 + off);
    C.r1 = *(const uint32_a2 *)(b1 + off);
}

__device__ __forceinline__ float finish_corners(const CornersZ8 &C)
{
#ifdef VV_XPAIR
    const float c000 = (float)(C.r0 & 0xffu), c100 = (float)((C.r0 >> 8) & 0xffu), c001 = (float)((C.r0 >> 16) & 0xffu), c101 = (float)(C.r0 >> 24);
    const float c010 = (float)(C.r1 & 0xffu), c110 = (float)((C.r1 >> 8) & 0xffu), c011 = (float)((C.r1 >> 16) & 0xffu), c111 = (float)(C.r1 >> 24);
#else
    const float c000 = (float)(C.r0 & 0xffu), c001 = (float)((C.r0 >> 8) & 0xffu), c100 = (float)((C.r0 >> 16) & 0xffu), c101 = (float)(C.r0 >> 24);
    const float c010 = (float)(C.r1 & 0xffu), c011 = (float)((C.r1 >> 8) & 0xffu), c110 = (float)((C.r1 >> 16) & 0xffu), c111 = (float)(C.r1 >> 24);
#endif
    const float c00 = __builtin_fmaf(C.wx, c100 - c000, c000);
    const float c10 = __builtin_fmaf(C.wx, c110 - c010, c010);
    const float c01 = __builtin_fmaf(C.wx, c101 - c001, c001);
    const float c11 = __builtin_fmaf(C.wx, c111 - c011, c011);
    const float c0 = __builtin_fmaf(C.wy, c10 - c00, c00);
    const float c1 = __builtin_fmaf(C.wy, c11 - c01, c01);
    return __builtin_fmaf(C.wz, c1 - c0, c0);
}

// same operations as finish_corners (x, then y, then z; fma(w, b - a, a)), two at a time
__device__ __forceinline__ float finish_corners(const CornersZ &C)
{
    const float2v lo0 = {C.c000, C.c001}, hi0 = {C.c100, C.c101}, lo1 = {C.c010, C.c011}, hi1 = {C.c110, C.c111};
    const float2v wx = {C.wx, C.wx}, wy = {C.wy, C.wy};
    const float2v cy0 = __builtin_elementwise_fma(wx, hi0 - lo0, lo0);      // (c00, c01)
    const float2v cy1 = __builtin_elementwise_fma(wx, hi1 - lo1, lo1);      // (c10, c11)
    const float2v cz  = __builtin_elementwise_fma(wy, cy1 - cy0, cy0);      // (c0, c1)
    return __builtin_fmaf(C.wz, cz.y - cz.x, cz.x);
}

// loaded rows -> trilinear value in storage units (0..255 for u8, as-is for f32)
template <int VOXEL>
__device__ __forceinline__ float finish_corners(const Corners<VOXEL> &C)
{
    float c000, c100, c010, c110, c001, c101, c011, c111;
    if constexpr (VOXEL == VV_VOXEL_F32) {
        c000 = C.a.x; c100 = C.a.y; c010 = C.b.x; c110 = C.b.y; c001 = C.c.x; c101 = C.c.y; c011 = C.d.x; c111 = C.d.y;
    } else {
        // v_alignbyte shifts each dword pair so that byte 0 is voxel x
        const uint32_t a = __builtin_amdgcn_alignbyte(C.a1, C.a0, C.sh), b = __builtin_amdgcn_alignbyte(C.b1, C.b0, C.sh);
        const uint32_t c = __builtin_amdgcn_alignbyte(C.c1, C.c0, C.sh), d = __builtin_amdgcn_alignbyte(C.d1, C.d0, C.sh);
#ifdef VV_ZFAST      // (the z-fastest build: byte 0 / 1 are voxels z / z+1 of the rows (x, y), (x, y+1), (x+1, y), (x+1, y+1))
        c000 = (float)(a & 0xffu); c001 = (float)((a >> 8) & 0xffu); c010 = (float)(b & 0xffu); c011 = (float)((b >> 8) & 0xffu);
        c100 = (float)(c & 0xffu); c101 = (float)((c >> 8) & 0xffu); c110 = (float)(d & 0xffu); c111 = (float)((d >> 8) & 0xffu);
#else
        c000 = (float)(a & 0xffu); c100 = (float)((a >> 8) & 0xffu); c010 = (float)(b & 0xffu); c110 = (float)((b >> 8) & 0xffu);
        c001 = (float)(c & 0xffu); c101 = (float)((c >> 8) & 0xffu); c011 = (float)(d & 0xffu); c111 = (float)((d >> 8) & 0xffu);
#endif
    }
    float c00 = __builtin_fmaf(C.wx, c100 - c000, c000);
    float c10 = __builtin_fmaf(C.wx, c110 - c010, c010);
    float c01 = __builtin_fmaf(C.wx, c101 - c001, c001);
    float c11 = __builtin_fmaf(C.wx, c111 - c011, c011);
    float c0 = __builtin_fmaf(C.wy, c10 - c00, c00);
    float c1 = __builtin_fmaf(C.wy, c11 - c01, c01);
    return __builtin_fmaf(C.wz, c1 - c0, c0);
}

// Trilinear reconstruction in storage units at normalised texture coordinates p (any value;
// out-of-range is clamped, NaN -> texel 0).
template <int VOXEL, bool TEX8, bool BIG = false>
__device__ __forceinline__ float tex3d_raw(const VolumeView &V, float px, float py, float pz)
{
    Corners<VOXEL> C;
    fetch_corners<VOXEL, TEX8, BIG ? LAYOUT_LINEAR_BIG : LAYOUT_LINEAR>(V, px, py, pz, C);
    return finish_corners<VOXEL>(C);
}

// kernel.cu:65-71 boundsCheck on all three coordinates: p in [0,1).  A float is in
// [0,1) iff its bit pattern, read as unsigned, is below 0x3F800000 (negatives and
// NaNs have larger patterns; -0.0 cannot arise from (x-.5)/s+.5 in round-to-nearest).
__device__ __forceinline__ bool bounds_check(float x, float y, float z)
{
    uint32_t m = max(max(__float_as_uint(x), __float_as_uint(y)), __float_as_uint(z));
    return m < 0x3F800000u;
}

// kernel.cu:99-105 sample(): (uchar)(0xff * tex3D) or 0 outside.  For u8 volumes the
// normalisation and the multiplication cancel, the index is trunc(L) (DESIGN.md pin 2);
// f32 volumes: trunc(255 * L), saturated.
template <int VOXEL> __device__ __forceinline__ float corner_value(const Corners<VOXEL> &C) { return finish_corners<VOXEL>(C); }
template <int VOXEL> __device__ __forceinline__ float corner_value(const CornersZ &C) { return finish_corners(C); }
template <int VOXEL> __device__ __forceinline__ float corner_value(const CornersZ8 &C) { return finish_corners(C); }

// ... without the bounds test (the caller applies it)
template <int VOXEL, class CT>
__device__ __forceinline__ uint32_t classify_raw(const CT &C)
{
    float L = corner_value<VOXEL>(C);
    float s = (VOXEL == VV_VOXEL_F32) ? L * 255.0f : L;
    return min((uint32_t)s, 255u);                  // v_cvt_u32_f32 saturates; NaN -> 0
}
template <int VOXEL, class CT>
__device__ __forceinline__ uint32_t classify_index(const CT &C, float px, float py, float pz)
{
    float L = corner_value<VOXEL>(C);
    float s = (VOXEL == VV_VOXEL_F32) ? L * 255.0f : L;
    uint32_t idx = min((uint32_t)s, 255u);          // v_cvt_u32_f32 saturates; NaN -> 0
    return bounds_check(px, py, pz) ? idx : 0u;
}
template <int VOXEL, bool TEX8, bool BIG = false>
__device__ __forceinline__ uint32_t sample_index(const VolumeView &V, float px, float py, float pz)
{
    Corners<VOXEL> C;
    fetch_corners<VOXEL, TEX8, BIG ? LAYOUT_LINEAR_BIG : LAYOUT_LINEAR>(V, px, py, pz, C);
    return classify_index<VOXEL>(C, px, py, pz);
}

__device__ __forceinline__ uint32_t pack_rgba(float r, float g, float b, float a)
{
    // kernel.cu:359-365: clamp to [0,1], * 0xff, truncate
    uint32_t R = (uint32_t)(fmaxf(0.f, fminf(r, 1.f)) * 255.0f);
    uint32_t G = (uint32_t)(fmaxf(0.f, fminf(g, 1.f)) * 255.0f);
    uint32_t B = (uint32_t)(fmaxf(0.f, fminf(b, 1.f)) * 255.0f);
    uint32_t A = (uint32_t)(fmaxf(0.f, fminf(a, 1.f)) * 255.0f);
    return R | (G << 8) | (B << 16) | (A << 24);
}

// Instrumentation of a (never timed) frame: what the roofline's byte model is counted from.
struct InstrArgs {
    uint32_t *bricks;        // 1 bit per 8^3-voxel brick of the volume touched by an executed in-volume sample (SURVEY 8d's algorithmic bytes), or NULL
    uint32_t *lines;         // 1 bit per 128-byte line of the layout this frame SAMPLES (offsets from the layout's base), or NULL
    uint64_t  line_bits;     // size of `lines` in bits (lines beyond it are not marked)
    int       lines_all;     // 0: lines of executed in-volume samples only (what has to be fetched at line granularity);
                             // 1: lines of every gather the kernel issues, idle lanes and out-of-volume samples included
    unsigned long long *pairs;   // open-addressing hash set of (block, line) pairs, 2^pairs_log2 zero-initialised words, or NULL: the lines each
    uint32_t  pairs_log2;        // block touches, counted per block (what the frame fetches if nothing is shared between blocks)
#ifdef VV_TIMELINE
    unsigned long long *timeline;   // experiment builds only (tools/timeline.py): 4 words per block of march_kernel -- start, end (100 MHz clock), strip << 16 | tile, XCC | live << 8
#endif
};
__device__ __forceinline__ void mark_line_range(const InstrArgs &I, uint64_t off, uint32_t bytes)
{
    for (uint64_t l = off >> 7; l <= (off + bytes - 1u) >> 7; ++l) {
        if (I.lines && l < I.line_bits) {
            const uint32_t bit = 1u << (l & 31);
            if (!(I.lines[l >> 5] & bit)) atomicOr(&I.lines[l >> 5], bit);
        }
        if (I.pairs) {
            const unsigned long long key = ((unsigned long long)(blockIdx.x + 1u) << 36) | (l & 0xFFFFFFFFFull);
            const unsigned long long mask = (1ull << I.pairs_log2) - 1ull;
            unsigned long long h = (key * 0x9E3779B97F4A7C15ull) >> (64u - I.pairs_log2);
            for (int probe = 0; probe < (1 << 14); ++probe, h = (h + 1ull) & mask) {          // (bounded: a full table drops the pair instead of spinning)
                const unsigned long long seen = I.pairs[h];
                if (seen == key) break;
                if (seen == 0ull) { const unsigned long long old = atomicCAS(&I.pairs[h], 0ull, key); if (old == 0ull || old == key) break; }
            }
        }
    }
}

// byte offsets (from VolumeView::bricks) of the four row loads fetch_corners() issues on the bricked copy (instrumentation only)
template <int VOXEL>
__device__ __forceinline__ void brick_offsets(const VolumeView &V, uint32_t ix, uint32_t iy, uint32_t iz, uint64_t a[4])
{
    using G = BrickGeom<VOXEL>;
    const uint32_t ya = iy & 3u, za = iz & (G::bz - 1u);
    const uint32_t oy0 = (iy >> 2) * V.b_sy + ya * G::row;
    const uint32_t oy1 = oy0 + (ya == 3u ? V.b_sy - 3u * G::row : G::row);
    const uint64_t m0 = (uint64_t)(iz >> G::zlog2) * V.b_sz64, m1 = m0 + (za == G::bz - 1u ? V.b_sz64 : 0u);
    const uint32_t zi0 = za * 4u * G::row, zi1 = za == G::bz - 1u ? 0u : zi0 + 4u * G::row;
    const uint32_t ox = brick_x_offset<VOXEL>(ix);
    a[0] = (m0 << 6) + ox + oy0 + zi0; a[1] = (m0 << 6) + ox + oy1 + zi0;
    a[2] = (m1 << 6) + ox + oy0 + zi1; a[3] = (m1 << 6) + ox + oy1 + zi1;
}

__device__ __forceinline__ void mark_bricks(uint32_t *bm, const VolumeView &V, float px, float py, float pz)
{
    // instrumentation only: 8^3-voxel bricks touched by the 2x2x2 footprint of a sample
    float xb = fminf(fmaxf(px * (float)V.nx - 0.5f, 0.f), (float)(V.nx - 1));
    float yb = fminf(fmaxf(py * (float)V.ny - 0.5f, 0.f), (float)(V.ny - 1));
    float zb = fminf(fmaxf(pz * (float)V.nz - 0.5f, 0.f), (float)(V.nz - 1));
    int ix = (int)xb, iy = (int)yb, iz = (int)zb;
    int bnx = (V.nx + 7) >> 3, bny = (V.ny + 7) >> 3;
    for (int c = 0; c < 8; c++) {
        int x = min(ix + (c & 1), V.nx - 1) >> 3, y = min(iy + ((c >> 1) & 1), V.ny - 1) >> 3,
            z = min(iz + (c >> 2), V.nz - 1) >> 3;
        size_t b = ((size_t)z * bny + y) * bnx + x;
        uint32_t bit = 1u << (b & 31);
        if (!(bm[b >> 5] & bit)) atomicOr(&bm[b >> 5], bit);
    }
}


} // namespace vv
