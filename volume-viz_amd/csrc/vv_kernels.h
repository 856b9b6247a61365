// vv_kernels.h -- host-visible launch interface of the HIP kernels (internal).
#pragma once
#include <hip/hip_runtime.h>
#include "vv_device.h"

namespace vv {

// blockIdx.y -> pixel strip (march_kernel) / slab row (march_phong_kernel) of a shard
struct StripMap { int y0, strips_per_band, band_stride_px, tile_log2w, n_strips, xcd_band;
                  int tail_batch;     // march_kernel: chunks that hold at most one sample per lane (rays past the ERT threshold) are taken U at a time
                  int blk_log2w;      // a block covers 2^blk_log2w x (256 >> blk_log2w) pixels (5: 32 x 8; 32 x 2 tiles also 64 x 4 / 128 x 2, 8 x 8 tiles 16 x 16 / 8 x 32); strips are that high
                  // march_kernel launches the tiles of columns [tx0, tx0 + wr) of strips [s0, s1) only: the tiles under the volume's screen rectangle
                  // (vv_render: screen_rect) -- or all of them: tx0 = 0, wr = tile columns of the frame, s0 = 0, s1 = n_strips
                  int tx0, wr, s0, s1;
                  // order != nullptr: block L marches the tile in slot ((j / order_run) * 8 + L % 8) * order_run + j % order_run, j = L / 8, of a table that
                  // rad_kernel's extra block writes (see there): runs of order_run x-adjacent tiles, sorted by the time their rays spend in the cube and dealt
                  // to the XCDs so that all of them carry the same load and end on their lightest runs (speed only).
                  const uint32_t *order; int order_run; };
// the same rectangle in pixels: [x0, x1) x [y0, y1).  Every owned pixel outside it is a pixel whose ray misses the volume: rad_kernel writes its 0,
// and computes no radius for slabs that do not meet the rectangle (march_kernel, which reads them, is not launched there).  No rectangle: x1 = y1 = INT_MAX.
struct PixelRect { int x0, x1, y0, y1; };

struct SlabMap  { int r0, band, band_stride, n_regular;
                  // march_phong_kernel launches slab columns [gx0, gx0 + wg) of the grid rows [gs0, gs1) and the extra row n_regular (pin 10): the slabs under the
                  // volume's screen rectangle, or all of them (gx0 = 0, wg = nbx, gs0 = 0, gs1 = n_regular)
                  int gx0, wg, gs0, gs1; };

struct MarchArgs {
    FrameParams P;
    VolumeView  V;
    int V_type;                 // vv_voxel_type
    bool tex8, gray, phong, instr;
    bool xpair;                 // VolumeView::zpair holds the x-pair copy (side views): launch_raymarch_xpair
    int lds_reserve;            // march_kernel: dynamic LDS bytes reserved only to cap blocks per CU
    int unroll;                 // march_kernel: samples per loop trip (2 or 3)
    int lds_reserve_phong;      // march_phong_kernel: same occupancy cap (its own LDS is 14 KB)
    StripMap strips;                   // march_kernel: strips of 8 pixel rows (n_strips of them)
    PixelRect rect;                    // rad_kernel + march kernels: the pixels the march kernel's tiles / slabs cover
    bool fill_outside;                 // Phong frames: rad_kernel is launched (without radii) to write the pixels outside `rect`
    uint32_t *order_out;               // rad_kernel: where its extra block writes StripMap::order (nullptr: no such block)
    SlabMap slabs;                     // march_phong_kernel grid.y = n_regular + 1
    const float4 *tf;           // device, 256 entries
    const float *rad;           // device, nbx*nby (read by march_kernel)
    float *rad_out;             // same buffer (written by rad_kernel)
    uint32_t *pixels;           // device RGBA8 frame
    unsigned long long *counter;
    InstrArgs I;                // bitmaps of an instrumented frame (vv_render_options::touched_bricks / touched_lines)
};

void launch_rad(const MarchArgs &a, hipStream_t s);
void launch_raymarch(const MarchArgs &a, hipStream_t s);
void launch_raymarch_big(const MarchArgs &a, hipStream_t s);      // same kernels, volumes above 4 GiB
void launch_raymarch_bricked(const MarchArgs &a, hipStream_t s);  // same kernels on VolumeView::bricks
void launch_raymarch_bricked_cached(const MarchArgs &a, hipStream_t s);  // ... the build for volumes up to 1 GiB
void launch_raymarch_zpair(const MarchArgs &a, hipStream_t s);    // same kernels on VolumeView::zpair
void launch_raymarch_zfast(const MarchArgs &a, hipStream_t s);    // same kernels on VolumeView::zfast
void launch_raymarch_xpair(const MarchArgs &a, hipStream_t s);    // same kernels on the x-pair copy (handed over in VolumeView::zpair)
void launch_build_xpair(int vtype, const void *zfast, uint32_t zf_row_bytes, uint64_t zf_slice_bytes, void *xpair, int nx, int ny, int nz, hipStream_t s);
void launch_build_zfast(int vtype, const void *vol, uint32_t row_pitch, uint64_t slice_pitch, void *out, uint32_t zf_row_bytes, uint64_t zf_slice_bytes, int nx, int ny, int nz, hipStream_t s);
size_t zpair_copy_bytes(int vtype, int nx, int ny, int nz, uint32_t *row_bytes, uint32_t *slab_bytes);
void launch_build_zpair(int vtype, const void *linear, size_t row_pitch, size_t slice_pitch, void *zpair, int nx, int ny, int nz, hipStream_t s);
void launch_repitch(const void *dense, void *pitched, size_t row_bytes /* multiple of 16 */, size_t ny, size_t nz,
                    size_t row_pitch, size_t slice_pitch, hipStream_t s);
size_t brick_copy_bytes(int vtype, int nx, int ny, int nz, uint32_t *sy, uint32_t *sz64);
void launch_build_bricks(int vtype, const void *linear, size_t row_pitch, size_t slice_pitch, void *bricks, int nx, int ny, int nz, hipStream_t s);
struct SliceArgs {
    VolumeView V; int V_type; bool tex8;
    float *buffer; size_t height, width;
    float dx, dy, dz; int orientation; int legacy;
    float scale[3];
    float trans[16]; int advanced;
};
void launch_slice(const SliceArgs &a, hipStream_t s);
void launch_first_pass(const FrameParams &P, uint32_t *front, uint32_t *back, hipStream_t s);

size_t generate_scratch_floats(int nx, int ny, int nz, int n);
void launch_generate_ellipsoids(uint8_t *out, int nx, int ny, int nz, int n,
                                const float *centers, const float *axes, const uint8_t *colors,
                                int in_place, float *scratch /* generate_scratch_floats() device floats */, hipStream_t s);
void launch_promote_u8_f32(const uint8_t *in, float *out, size_t n, hipStream_t s);
void launch_noise_u8(uint8_t *out, int nx, int ny, int nz, uint32_t seed, hipStream_t s);

constexpr int kMaxEllipsoids = 64;

} // namespace vv
