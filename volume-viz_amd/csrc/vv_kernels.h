// vv_kernels.h -- host-visible launch interface of the HIP kernels (internal).
#pragma once
#include <hip/hip_runtime.h>
#include "vv_device.h"

namespace vv {

// blockIdx.y -> pixel strip (march_kernel) / slab row (march_phong_kernel) of a shard
struct StripMap { int y0, strips_per_band, band_stride_px, tile_log2w, n_strips, xcd_band;
                  int skew_axis;      // 0: lock-step march_kernel; 1 / 2: march_skew_kernel, lanes aligned along y / z
                  int tail_batch;     // march_kernel: chunks that hold at most one sample per lane (rays past the ERT threshold) are taken U at a time
                  int blk_log2w; };   // a block covers 2^blk_log2w x (256 >> blk_log2w) pixels (5: 32 x 8; 32 x 2 tiles also 64 x 4 / 128 x 2, 8 x 8 tiles 16 x 16 / 8 x 32); strips are that high

struct SlabMap  { int r0, band, band_stride, n_regular; };

// Slab sweep (vv_sweep.hip): a block of nc = wx * wy waves (32 x 2 pixels each; every wave marches AND copies) owns a
// (32 wx) x (2 wy) pixel tile and walks the volume slice by slice along the sweep axis (y or z; rows along x are
// contiguous in both cases).
struct SweepArgs {
    int enabled;
    int verbose;           // print the planner's decision (VV_SWEEP_VERBOSE)
    int major;             // 1: slices are x-z planes (sweep along y); 2: x-y planes (sweep along z)
    int sgn;               // +1: slice index grows along every ray, -1: it falls
    int wx, wy, nc, nl;    // consumer waves across / down the tile, their product, loader waves per block
    int ntx, nty;          // tile grid
    int y0, rows_per_band, band_stride_px;   // pixel row of tile row t: y0 + (t / rows_per_band) * band_stride_px + (t % rows_per_band) * 2 wy
    int group;             // (1: slices are placed one by one)
    int ahead;             // trips the copies run ahead of the march
    int steps;             // sample steps per trip (1 or 2)
    int wmax;              // widest slice window (in slices) a consumer wave may need and still use the ring
    int pxw, ry, ring;     // LDS image of a slice: ry rows of pxw voxels (a multiple of 4); `ring` such slots
    int pxc;               // (row pitch in 128-byte lines: planner output for the record)
    int slot_bytes;        // pxw * 4 * ry
    int blocks;            // requested blocks per CU (0 / -1: the planner's choice)
    int lds_bytes;         // dynamic LDS of the launch
    unsigned long long *trace;   // developer trace (VV_SWEEP_TRACE=1): 8 words per block, or NULL
    const int *order;      // optional tile order (device), n_order entries; NULL = XCD-interleaved raster order
    int n_order;
};

struct MarchArgs {
    FrameParams P;
    VolumeView  V;
    int V_type;                 // vv_voxel_type
    bool tex8, gray, phong, instr;
    bool xpair;                 // VolumeView::zpair holds the x-pair copy (side views): launch_raymarch_xpair
    int lds_reserve;            // march_kernel: dynamic LDS bytes reserved only to cap blocks per CU
    int unroll;                 // march_kernel: samples per loop trip (2 or 3; march_skew_kernel also 1)
    int lds_reserve_phong;      // march_phong_kernel: same occupancy cap (its own LDS is 14 KB)
    int phong_pair;             // march_phong_pair_kernel (two x-adjacent slabs per block) instead of march_phong_kernel
    int phong_v2;               // 0: march_phong_kernel; 1 / 2: march_phong2_kernel (double-buffered sample cache) with one / two slabs per block
    SweepArgs sweep;                   // sweep_kernel (vv_sweep.hip)
    StripMap strips;                   // march_kernel: strips of 8 pixel rows (n_strips of them)
    SlabMap slabs;                     // march_phong_kernel grid.y = n_regular + 1
    const float4 *tf;           // device, 256 entries
    const float *rad;           // device, nbx*nby (read by march_kernel)
    float *rad_out;             // same buffer (written by rad_kernel)
    uint32_t *pixels;           // device RGBA8 frame
    unsigned long long *counter;
    uint32_t *bricks;
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
// block-wide slab sweep, LDS slice ring filled by LDS-DMA (f32, no Phong).  False: the kernel's static + the planned dynamic LDS exceed a CU's 160 KB (or its
// attributes cannot be set): nothing was launched, the caller takes the gather kernel
bool launch_raymarch_sweep(const MarchArgs &a, hipStream_t s);
bool sweep_lds_fits(size_t static_bytes, size_t dynamic_bytes);
// host-side sizing of the sweep for a frame (fills a.sweep; enabled = 0 if the frame does not qualify)
void plan_sweep(MarchArgs &a, int first_tile_row_px, int n_pixel_rows, int own_bands);
// the axis (1 = y, 2 = z) and direction along which every ray of the frame crosses the volume's slices, or false
bool sweep_axis(const FrameParams &P, const VolumeView &V, int &major, int &sgn, const char **why);

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
