/*
 * volviz.h -- C-ABI boundary of the MI355X-native volume ray-marcher.
 *
 * This header is what a host application (the reference's glwidget.cpp /
 * slicewidget.cpp, or any FFI) binds instead of the reference's kernel.cuh.
 * Every entry point cites the reference interface it replaces.  All types are
 * plain C: pointers, sizes, PODs.  No torch / HIP types appear in signatures
 * (a stream is passed as an opaque void* that is a hipStream_t; NULL = the
 * default stream).
 *
 * Conventions
 *   - every function returns VV_OK (0) or a negative vv_status; it never calls
 *     exit() (the reference's checkCudaErrors does: include/helper_cuda.h:763-777);
 *     vv_last_error() returns the text for the last failure on the context.
 *   - calls are synchronous from the caller's view unless a stream is passed
 *     AND the output buffer is device memory, in which case the work is only
 *     enqueued on that stream (reference contract: kernel.cu:452,517 fence each call).
 *   - a context owns one volume, one transfer function, scratch buffers
 *     (reference: file-static globals, kernel.cu:35-51).  Not re-entrant.
 *   - volume layout: u8 or f32, x fastest, then y, then z (kernel.cu:477,
 *     volumegenerator.cpp:41).
 *   - output image: RGBA u8, row-major, row 0 = bottom (GL), W*H*4 bytes
 *     (glwidget.cpp:365-369, kernel.cu:365).
 */
#ifndef VOLVIZ_H
#define VOLVIZ_H

#include <stddef.h>
#include <stdint.h>
#include <stdbool.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- reference PODs, byte-identical to kernel.cuh:18-40 ------------------- */
#ifndef SLICE_NONE
#define SLICE_NONE      -1   /* kernel.cuh:18 */
#define SLICE_PLANE      0   /* kernel.cuh:19 */
#define SLICE_PLANE_CUT  1   /* kernel.cuh:20 */
#endif
#ifndef TRANSFER_PRESET_DEFAULT
#define TRANSFER_PRESET_DEFAULT -1  /* kernel.cuh:22 */
#define TRANSFER_PRESET_ENGINE   0  /* kernel.cuh:23 */
#define TRANSFER_PRESET_MRI      1  /* kernel.cuh:24 */
#endif

struct slice_params {        /* kernel.cuh:26-29 */
    int   type;              /* SLICE_NONE / SLICE_PLANE / SLICE_PLANE_CUT */
    float params[6];         /* plane point xyz, plane normal xyz (cube space) */
};

struct camera_params {       /* kernel.cuh:31-35 */
    float origin[3];         /* world-space eye (glwidget.cpp:271-273) */
    float fovX, fovY;        /* degrees; fovX = fovY*aspect (glwidget.cpp:340-341) */
    float scale[3];          /* object scale (glwidget.cpp:267-269) */
};

struct shading_params {      /* kernel.cuh:37-40 */
    int  transferPreset;     /* carried, never read by the kernel (as in the reference) */
    bool phongShading;
};

/* params.h:46 canonicalOrientation, same enumerator values */
typedef enum {
    VV_HORIZONTAL = 0, VV_SAGITTAL = 1, VV_CORONAL = 2,
    VV_N_CANONICAL_ORIENTATIONS = 3, VV_FREE_FORM = 4
} vv_orientation;

/* ---- status ----------------------------------------------------------------- */
typedef enum {
    VV_OK = 0,
    VV_ERR_INVALID = -1,     /* bad argument */
    VV_ERR_NO_VOLUME = -2,   /* render/slice before a volume was loaded */
    VV_ERR_DEVICE = -3,      /* HIP runtime error */
    VV_ERR_NOMEM = -4,
    VV_ERR_IO = -5
} vv_status;

typedef struct vv_context vv_context;

/* The `stream` argument of the entry points below is a hipStream_t passed as void*:
 *   NULL                     the context's own stream; the call returns when the work is complete (the reference's
 *                            synchronous contract, kernel.cu:452,517);
 *   VV_STREAM_DEFAULT_ASYNC  the device's default (null) stream, enqueue only;
 *   any other handle         that stream, enqueue only: the caller synchronises.                                  */
#define VV_STREAM_DEFAULT_ASYNC ((void *)1)

typedef enum { VV_VOXEL_U8 = 0, VV_VOXEL_F32 = 1 } vv_voxel_type;

/* Trilinear reconstruction model (what tex3D does in kernel.cu:102,589,628).
 *   VV_FILTER_TEX8  : CUDA texture-unit model -- interpolation weights rounded
 *                     to 8 fractional bits (1.8 fixed point).  Default.
 *   VV_FILTER_EXACT : full float weights.                                     */
typedef enum { VV_FILTER_TEX8 = 0, VV_FILTER_EXACT = 1 } vv_filter;

/* Early-ray termination.
 *   VV_ERT_REFERENCE : the reference's behaviour -- `break` leaves only the
 *                      30-sample inner loop (kernel.cu:272-274); every later
 *                      chunk still composites its first sample.  Default.
 *   VV_ERT_TRUE      : the ray stops for good once alpha > threshold.          */
typedef enum { VV_ERT_REFERENCE = 0, VV_ERT_TRUE = 1 } vv_ert_mode;

/* Where front/back ray end points come from (first-pass contract,
 * firstpass.vert:6, glwidget.cpp:198-228).                                    */
typedef enum {
    VV_RAYS_IMAGES = 0,   /* two RGBA8 images = the reference's FBO0/FBO1        */
    VV_RAYS_ANALYTIC = 1  /* ray/box intersection computed from the camera       */
} vv_ray_mode;

typedef struct vv_ray_source {
    int            mode;             /* vv_ray_mode */
    /* VV_RAYS_IMAGES: RGBA8, row 0 = bottom, sampled at (x/W, y/H) with point
     * filtering exactly like tex2D(inTexture0, ..) in kernel.cu:317-318.      */
    const uint8_t *front;            /* img_w*img_h*4 bytes */
    const uint8_t *back;
    int            img_w, img_h;     /* FBO size (reference: widget size = 3x the render size) */
    int            images_on_device; /* 0 = host pointers, 1 = device pointers */
    /* VV_RAYS_ANALYTIC: camera basis (camera.cpp:78-91); eye = camera_params.origin,
     * projection = perspective(fovY, aspect) as glwidget.cpp:338.
     * VV_RAYS_IMAGES: optional HINT (speed only, pixels never depend on it): the camera that drew the
     * images, if the host knows it (glwidget.cpp:188-228 draws them from `camera`).  With a hint vv_render
     * chooses wave tiles, volume layout and occupancy as for analytic rays; without one it estimates the
     * view from the images' centre row (host images; device images only in synchronous calls, which may
     * read them back) and otherwise launches conservatively.  Leave look / up zero for "no hint": ZERO-INITIALISE this struct -- stale
     * values in look / up of an image source are read as a camera and pick tile shape, layout (possibly building a copy of the volume)
     * and occupancy from it; a non-finite or degenerate hint is ignored.  Device images must be 4-byte aligned (VV_ERR_INVALID otherwise). */
    float          look[3];
    float          up[3];
    float          aspect;           /* <= 0 : use W/H */
    int            quantize8;        /* 1 = round end points to RGBA8 like the FBO does */
} vv_ray_source;

typedef struct vv_render_options {
    float    step[3];        /* per-axis step; all 0 => 1/dims (kernel.cu:415)   */
    float    ert_threshold;  /* 0 => .95f (kernel.cu:272)                        */
    int      filter;         /* vv_filter                                        */
    int      ert_mode;       /* vv_ert_mode                                      */
    /* Screen-tile sharding (multi-GPU).  A pixel row y belongs to slab row r = y/14 of the
     * global 14x14 slab grid (kernel.cu:418); this call renders the rows whose r satisfies
     *   begin <= r < end   (0,0 => all)   and   (r / shard_band) % shard_count == shard_index
     * Other rows are left untouched.  shard_count <= 1 disables the interleave;
     * shard_band must be a multiple of 4 slab rows when shard_count > 1.              */
    int      slab_row_begin;
    int      slab_row_end;
    int      shard_band;     /* slab rows per interleave band (e.g. 4 = 56 pixel rows)  */
    int      shard_count;    /* number of shards (ranks)                                */
    int      shard_index;    /* this shard                                               */
    int      count_samples;  /* 1 => count executed samples (vv_last_sample_count)*/
    uint32_t *touched_bricks;/* device bitmap, 1 bit per 8^3 brick, or NULL:     */
                             /* instrumentation for the roofline's byte model    */
    /* The same at the granularity the memory system fetches at: 1 bit per 128-byte line of the LAYOUT THE FRAME SAMPLES (bit = byte offset
     * from that layout's base / 128; vv_device_bytes() bounds the size: the largest resident layout), or NULL.  touched_lines_all = 0 marks
     * the lines of executed in-volume samples (what must be fetched at least once), 1 the lines of every gather the kernel issues.      */
    uint32_t *touched_lines;
    unsigned long long touched_line_bits;   /* size of touched_lines in bits */
    int      touched_lines_all;
    /* ... and per block: a zero-initialised device hash set of 2^log2 64-bit words that receives one entry per (thread block, line) pair
     * under the same rule (count the non-zero words afterwards): the bytes the frame fetches if blocks share nothing.  Size it at
     * >= 2 x the pairs expected.  NULL = off. */
    unsigned long long *touched_block_lines;
    int      touched_block_lines_log2;
} vv_render_options;

/* ---- lifecycle ---------------------------------------------------------------- */
/* replaces initCuda() (kernel.cuh:44, kernel.cu:369-373).  device < 0 => current device. */
int  vv_init(int device, vv_context **out_ctx);
int  vv_shutdown(vv_context *ctx);
const char *vv_last_error(const vv_context *ctx);   /* ctx may be NULL (global error) */

/* ---- volume + transfer function: replaces cudaLoadVolume (kernel.cuh:53, kernel.cu:456-498).
 * texels: host pointer, nx*ny*nz voxels, x fastest.  tf: 256 RGBA float entries.
 * Reloading frees the previous volume (the reference leaks it).                 */
int  vv_load_volume_u8 (vv_context *ctx, const uint8_t *texels, size_t size,
                        int nx, int ny, int nz, const float tf[1024]);
int  vv_load_volume_f32(vv_context *ctx, const float *texels, size_t size,
                        int nx, int ny, int nz, const float tf[1024]);
/* Same, from a buffer already resident in HBM (e.g. written by vv_generate_*). */
int  vv_load_volume_device(vv_context *ctx, const void *dev_texels, int voxel_type,
                           int nx, int ny, int nz, const float tf[1024], void *stream);
int  vv_set_transfer_function(vv_context *ctx, const float tf[1024]);

/* Streamed upload for volumes that should not sit in host memory whole (config C5: 2048^3
 * streamed from pinned host memory).  begin allocates the device volume; each slices call
 * enqueues slices [z0, z0+n) on an internal copy stream (directly if `src` is pinned host
 * memory, through a pinned double buffer otherwise) and returns; src_type may be VV_VOXEL_U8
 * while the volume is VV_VOXEL_F32, in which case the slices are promoted (v/255) on the
 * device.  end waits for the copies.  Slices may arrive in any order.  A slices call returns once the source
 * buffer has been read (pinned sources are copied from directly), so the caller may refill it at once.          */
int  vv_load_volume_stream_begin(vv_context *ctx, int voxel_type, int nx, int ny, int nz,
                                 const float tf[1024]);
int  vv_load_volume_stream_slices(vv_context *ctx, const void *src, int src_type, int z0, int nslices);
/* The same in two halves, for a host that feeds several contexts from one pinned buffer (volviz_mgpu) or wants to
 * overlap its own work with the transfer: _async only enqueues (H2D copy, then the promotion kernel on a second
 * stream), _wait_source returns when the copy engine has read every pinned source handed over so far -- the buffer may
 * then be refilled; promotion kernels may still be running (stream_end waits for them).
 * vv_load_volume_stream_slices == _async followed by _wait_source. */
int  vv_load_volume_stream_slices_async(vv_context *ctx, const void *src, int src_type, int z0, int nslices);
int  vv_load_volume_stream_wait_source(vv_context *ctx);
int  vv_load_volume_stream_end(vv_context *ctx);
/* .t3d file -> device volume in chunks (never holds the file in memory); voxel_type F32 promotes. */
int  vv_load_volume_t3d(vv_context *ctx, const char *path, int header, int voxel_type,
                        const float tf[1024]);

/* ---- ray march: replaces runCuda (kernel.cuh:46-51, kernel.cu:388-453) ----------
 * rgba_out: W*H*4 bytes; out_on_device selects host or device pointer.
 * Pixels the reference never writes (column W-1, row H-1) are left untouched.   */
int  vv_render(vv_context *ctx, int width, int height,
               const struct slice_params *slice,
               const struct camera_params *camera,
               const struct shading_params *shading,
               const vv_ray_source *rays,
               const vv_render_options *opts,     /* NULL => reference defaults */
               uint8_t *rgba_out, int out_on_device, void *stream);

/* ---- slice view: replaces invoke_slice_kernel (kernel.cuh:59, kernel.cu:506-519)
 * and invoke_advanced_slice_kernel (kernel.cuh:61, kernel.cu:522-541).
 * buffer: height*width floats; element (j,i) is stored at j*height+i exactly as
 * the reference does (kernel.cu:550,604).  legacy != 0 selects the 4-argument
 * slicekernel.cu:51-82 semantics (sagittal only, no scale, no bounds check).   */
int  vv_slice(vv_context *ctx, float *buffer, size_t height, size_t width,
              float dx, float dy, float dz, int orientation,
              const float scale[3], int legacy, int filter,
              int out_on_device, void *stream);
int  vv_slice_advanced(vv_context *ctx, float *buffer, size_t height, size_t width,
                       const float trans[16] /* row-major, CS123Algebra.h:278-281 */,
                       const float scale[3], int filter,
                       int out_on_device, void *stream);

/* The first pass on its own (firstpass.vert:6, firstpass.frag:4, glwidget.cpp:198-228): the
 * RGBA8 images the reference's two FBOs would hold for this camera -- cube-space entry (front
 * faces) and exit (back faces) positions, UNORM8-rounded, alpha 255 where a face is visible and
 * the clear colour (0,0,0,0) elsewhere.  A host without GL can feed them to VV_RAYS_IMAGES.
 * rays must be VV_RAYS_ANALYTIC.  Images are img_w x img_h x 4 bytes, row 0 = bottom.        */
int  vv_first_pass(vv_context *ctx, int img_w, int img_h, const struct camera_params *camera,
                   const vv_ray_source *rays, uint8_t *front_rgba, uint8_t *back_rgba,
                   int out_on_device, void *stream);

/* Cutting plane of the canonical slice views (GLWidget::setSliceCanonical, glwidget.cpp:743-788)
 * and the orientation rule applied before the plane goes into slice_params (glwidget.cpp:243-258:
 * the normal is negated when flip_cross_section && n.y < -1e-6, or !flip && n.y > 1e-6).       */
int  vv_cut_plane_canonical(int orientation, float displace, float point[3], float normal[3]);
/* Cutting plane of the free-form ("pro") slice view: Window::renderSlice's PRO_SLICING branch (window.cpp:425-443), which hands
 * GLWidget::setSlicePro (glwidget.cpp:743-755)  point = (dx, dy, dz) + .5  and  normal = T(.5) Rx(theta) Ry(phi) Rz(psi) T(-.5) (0,0,1,0),
 * binary32 with the operation order of cs123math (bit-identical to the compiled reference: tests/golden/cut_planes_pro.json).  Host-only. */
int  vv_cut_plane_from_euler(float dx, float dy, float dz, float theta, float phi, float psi, float point[3], float normal[3]);
int  vv_cut_plane_to_slice_params(int slice_type, const float point[3], const float normal[3],
                                  int flip_cross_section, struct slice_params *out);

/* Camera / cutting-plane controls of the 3D view (host-only arithmetic, no device work).
 * The reference does these with Qt 4 value types (float-stored QVector3D read back as qreal,
 * double QMatrix4x4) inside its mouse handlers; Qt is not available to pin them bit for bit, so
 * they are restated in double precision on float inputs/outputs and tested to 1e-5.
 *   vv_camera_orbit_drag    right-button drag (glwidget.cpp:432-446): spherical orbit about the
 *                           origin, theta clamped to [0.1, pi-0.1], look re-aimed at the origin
 *   vv_camera_zoom          wheel (glwidget.cpp:607-620): position += look * delta/200
 *   vv_cut_plane_from_drag  left-button drag released (glwidget.cpp:482-535): the plane through the
 *                           eye ray of the release point and the near-plane point of the press
 *                           point; window coordinates are fractions of the widget (y down);
 *                           point comes back in cube space [0,1]^3, normal un-normalised as in
 *                           the reference; perspective(45 deg, aspect, 0.1, 100) (glwidget.cpp:338)
 *   vv_cut_plane_drag       middle-button drag of such a plane (glwidget.cpp:447-452)          */
int  vv_camera_orbit_drag(const float position[3], int dx, int dy, float out_position[3], float out_look[3]);
int  vv_camera_zoom(const float position[3], const float look[3], int delta, float out_position[3]);
int  vv_cut_plane_from_drag(const float position[3], const float look[3], const float up[3], float aspect,
                            const float press[2], const float release[2],
                            float point[3], float normal[3], float plane_up[3], float plane_right[3]);
int  vv_cut_plane_drag(float point[3], const float plane_up[3], const float plane_right[3],
                       int dx, int dy, int width, int height);

/* Slice buffer -> the BGRA image SliceWidget shows (slicewidget.cpp:108-121): grey value
 * (unsigned)(f*255), alpha 255, written mirrored at bits[size - offset] with offset = j*height+i;
 * bits[0] and elements whose index falls outside stay untouched.  bgra: width*height*4 bytes. */
int  vv_slice_to_bgra(const float *slice, size_t height, size_t width, uint8_t *bgra);

/* Slice transform of the free-form slice view: SliceWidget::getTransformationMatrix
 * (slicewidget.cpp:147-165) = T(+.5) T(dx,dy,dz) Rx(theta) Ry(phi) Rz(psi) T(-.5),
 * binary32, row-major, multiplied left to right (cs123math/CS123Matrix.cpp:27-62).
 * Angles must lie in [-3.2, 3.2) (the reference asserts this).  Host-only.       */
int  vv_slice_matrix(float dx, float dy, float dz, float theta, float phi, float psi,
                     float out[16]);

/* ---- procedural generator: replaces VolumeGenerator::drawEllipsoid / drawDefaultBrain
 * (volumegenerator.cpp:31-119).  One fused pass applies the n ellipsoids in order
 * to a zero-filled volume, bit-identical to n successive drawEllipsoid calls.
 * out: nx*ny*nz bytes (host or device).                                          */
int  vv_generate_ellipsoids(vv_context *ctx, uint8_t *out, int out_on_device,
                            int nx, int ny, int nz, int n,
                            const float *centers /* n*3 */, const float *axes /* n*3 */,
                            const uint8_t *colors /* n */, void *stream);
/* One drawEllipsoid call (volumegenerator.cpp:31-97) applied IN PLACE to an existing volume:
 * voxels outside the ellipsoid keep their value, the marker slab fi >= 0.99 is set to 4. */
int  vv_draw_ellipsoid(vv_context *ctx, uint8_t *vol, int vol_on_device,
                       int nx, int ny, int nz, const float center[3], const float axes[3],
                       uint8_t color, void *stream);
int  vv_generate_default_brain(vv_context *ctx, uint8_t *out, int out_on_device,
                               int nx, int ny, int nz, void *stream);
/* u8 -> f32 promotion v/255 on the device (build extension for f32 configs). */
int  vv_promote_u8_to_f32(vv_context *ctx, const uint8_t *dev_in, float *dev_out,
                          size_t n, void *stream);
/* Synthetic "noise" volume V2 of the measurement plan: smooth hash field. */
int  vv_generate_noise_u8(vv_context *ctx, uint8_t *dev_out, int nx, int ny, int nz,
                          uint32_t seed, void *stream);

/* ---- transfer-function presets: transfer_functions.h:4-9 as closed forms -------- */
typedef enum { VV_TF_ENGINE = 0, VV_TF_HEAD = 1, VV_TF_MRI = 2 } vv_tf_preset;
int  vv_transfer_preset(int preset, float tf_out[1024]);

/* ---- dataset presets: the rule of GLWidget::loadVolume (glwidget.cpp:678-689) ----
 * The reference picks the transfer table and the object scale from the file name's ending:
 *     "engine.t3d"  -> g_transferEngine, scale (1, 1, 1)        "head.t3d" -> g_transferEngine, scale (1, 1, 0.8)
 *     "VisMale.t3d" -> g_transferHead,   scale (1.57, 1, 1)
 * Returns 1 and fills *tf_preset (a vv_tf_preset) and scale[3] when a rule matches; returns 0 and leaves both
 * untouched otherwise (the reference then reads an uninitialised table pointer and keeps the previous scale: the
 * caller's current table and scale are this library's reading of that); VV_ERR_INVALID (< 0) for NULL arguments.
 * Case-sensitive suffix match, as QString::endsWith. */
int  vv_dataset_preset(const char *path, int *tf_preset, float scale[3]);

/* ---- .t3d container (volumegenerator.cpp:147-220): 3 x u64 LE header + bytes ---- */
int  vv_t3d_read_header(const char *path, int header, int *nx, int *ny, int *nz);
int  vv_t3d_read (const char *path, int header, uint8_t *dst, size_t capacity);
int  vv_t3d_write(const char *path, int header, const uint8_t *src, int nx, int ny, int nz);

/* ---- optional second layouts of the loaded volume (no reference counterpart: cudaArray hides its layout) -----------------------
 * Which copy a frame samples is a launch-policy decision; results never depend on it.
 *   VV_LAYOUT_BRICKED  4x4x4-voxel bricks with an x halo (1.25x an f32 volume, 2x a u8 volume): frames whose screen x direction is more
 *                      than ~14 degrees off the volume's x and z axes; volumes of 2 M voxels and more
 *   VV_LAYOUT_ZFAST    rows along z (1x the volume, both voxel types): side views (screen x within ~14 degrees of the volume's z axis).
 *                      Preparing it also builds the x-pair copy (2x) where the policy would sample it (unshaded side views of u8 volumes
 *                      and of f32 volumes up to 512 MiB)
 *   VV_LAYOUT_ZPAIR    {v(z), v(z+1)} records (2x the volume): unshaded frames along the x axis, f32 up to 512 MiB, u8 up to 2 GiB
 *   VV_LAYOUT_POLICY   every copy of the above that vv_render's policy can pick for the loaded volume
 * RESIDENCY.  All copies together stay within a budget: by default the larger of 8 GiB and 2.5 x the linear volume (C3's 4.3 GB volume: the
 * bricked + z-fastest copies, 9.7 GB; C5's 32 GiB volume: 72 GiB), vv_set_layout_policy changes it.  A copy that does not fit evicts the
 * copies sampled least recently; if that is not enough it is not built and frames take the next layout of the policy (at worst the linear
 * volume).  By default vv_render builds a missing copy on the first frame that wants it: a kernel over the volume (1024^3 f32: bricked 4.2 ms,
 * z-fastest 2.5 ms; x 8 at 2048^3), a hipMalloc and a stream synchronisation -- also inside an enqueue-only call.  A host that must not
 * stall in its paint loop calls vv_prepare_layouts(VV_LAYOUT_POLICY) after loading (the mirror's cudaLoadVolume and PaintLoop::loadVolume
 * do) and may switch building in vv_render off (build_in_render = 0): frames then sample what is resident.
 * All copies are dropped when another volume is loaded.
 * vv_prepare_layouts returns the bit mask of the requested layouts that are resident afterwards, or a negative vv_status.               */
enum { VV_LAYOUT_BRICKED = 1, VV_LAYOUT_ZPAIR = 2, VV_LAYOUT_ZFAST = 4, VV_LAYOUT_POLICY = 256 };
int  vv_prepare_layouts(vv_context *ctx, int which, void *stream);
int  vv_set_layout_policy(vv_context *ctx, unsigned long long budget_bytes /* 0 = default */, int build_in_render /* default 1 */);
/* out = { linear volume, bricked, z-pair, z-fastest, x-pair copy (bytes; 0 = not resident), budget for the copies, copies built inside
 *         vv_render since the volume was loaded, build_in_render } */
int  vv_layout_state(const vv_context *ctx, unsigned long long out[8]);
/* Device memory held by the context: out[0] linear volume, [1] bricked copy, [2] z-pair + z-fastest + x-pair copies,
 * [3] tables and scratch (bytes). */
int  vv_device_bytes(const vv_context *ctx, unsigned long long out[4]);

/* ---- metrics (the reference only has a clock() overlay) ---------------- */
float              vv_last_frame_ms(const vv_context *ctx);      /* hipEvent time of the last vv_render (-1: not timed) */
/* Every vv_render brackets its kernels with two hipEventRecord (what vv_last_frame_ms reads): two more packets the stream has to retire per
 * frame, ~2-4 us each back to back.  A host that times whole runs itself (bench.py) or does not time at all switches them off (on = 0);
 * vv_last_frame_ms then returns -1.  Default: on, the reference's lastRenderTime overlay (glwidget.cpp:288-293) wants it. */
int                vv_set_frame_timing(vv_context *ctx, int on);
unsigned long long vv_last_sample_count(vv_context *ctx);        /* executed samples, if count_samples */
int                vv_debug_last_launch(vv_context *ctx, int out[8]);  /* what the launch policy chose for the last vv_render (developer aid): wave tile log2 width,
                                                                       * block log2 width, samples per trip, LDS reserve, layout (0 linear, 1 linear/64-bit, 2 bricked,
                                                                       * 3 z-pair, 4 z-fastest, 5 x-pair), view known to the policy (0 / 1), density x 1000, Phong (0 / 1) */
/* The rectangle of pixel coordinates (x_min, x_max, y_min, y_max, margin included) outside of which vv_render lets its pre-pass write (0,0,0,0) instead of
 * marching (analytic ray sources): returns 1 and fills out[4], or 0 when this camera gets no rectangle (a cube corner at or behind the eye's plane, a
 * margin wider than the frame).  Needs no device and no context: tests/test_host.py checks it against the oracle's ray-box test pixel by pixel. */
int                vv_debug_screen_rect(int width, int height, const struct camera_params *cam, const vv_ray_source *rays, double out[4]);
int                vv_debug_counters(vv_context *ctx, unsigned long long out[16]);  /* developer statistics of the last instrumented frame: [0] executed samples,
                                                                                     * [1] lane slots spent, [2] / [3] waves that sampled the bricked / a pair copy */
/* The VV_* developer knobs of the environment are read when a context is created and at every volume load, never
 * per frame; this reads them again (tests that flip a knob between two frames of one volume). */
int                vv_reread_env(vv_context *ctx);
int                vv_volume_dims(const vv_context *ctx, int dims[3], int *voxel_type);

#ifdef __cplusplus
}
#endif
#endif /* VOLVIZ_H */
