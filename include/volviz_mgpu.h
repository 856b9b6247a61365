/* volviz_mgpu.h -- one process, N GPUs: the screen-tile shard of SURVEY.md 8(e) behind a C-ABI
 * (libvolviz_mgpu.so; links libvolviz_hip.so and RCCL).
 *
 * The reference is single-GPU (kernel.cu:369-373 picks one device); this is the path BASELINE.json's
 * north_star adds: the frame is cut into bands of 4 slab rows (56 pixel rows) dealt round-robin to the
 * devices (vv_render_options.shard_*), the volume is replicated, nothing is exchanged during the march,
 * and the bands are gathered once per frame over xGMI into the frame on device 0:
 * ncclCommInitAll + one ncclGroupStart { ncclSend per band on its device, ncclRecv per band on device 0 into a
 * landing frame } ncclGroupEnd  (rccl.h:236,700,722), then the written (W-1)-pixel part of each received row is copied
 * into the destination frame, so the pixels the reference never writes keep the caller's bytes.
 * bench.py's torchrun path (one process per GPU, torch.distributed gather) is the same design for Python hosts.
 */
#ifndef VOLVIZ_MGPU_H
#define VOLVIZ_MGPU_H
#include "volviz.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct vv_mgpu vv_mgpu;

/* devices == NULL: devices 0 .. n_devices-1.  Creates one vv_context per device and the RCCL communicators. */
int  vv_mgpu_init(int n_devices, const int *devices, vv_mgpu **out);
int  vv_mgpu_shutdown(vv_mgpu *m);
int  vv_mgpu_size(const vv_mgpu *m);
/* The context of rank r: load the (replicated) volume / transfer function through the usual entry points. */
vv_context *vv_mgpu_context(vv_mgpu *m, int rank);
/* Convenience: the same host volume on every device. */
int  vv_mgpu_load_volume_u8 (vv_mgpu *m, const uint8_t *texels, size_t size, int nx, int ny, int nz, const float tf[1024]);
int  vv_mgpu_load_volume_f32(vv_mgpu *m, const float *texels, size_t size, int nx, int ny, int nz, const float tf[1024]);
/* The volume generated on every device (no host copy, no PCIe): drawDefaultBrain at nx x ny x nz, as u8 or promoted to f32. */
int  vv_mgpu_generate_default_brain(vv_mgpu *m, int voxel_type, int nx, int ny, int nz, const float tf[1024]);
/* C5: each device streams the u8 volume slab by slab from the caller's (pinned) host memory through
 * vv_load_volume_stream_* -- promoted to f32 on the device when voxel_type is VV_VOXEL_F32.  A slab is enqueued on every
 * device before any is waited for (each device has its own PCIe link); promotion overlaps the next slab's copies. */
int  vv_mgpu_stream_volume_u8(vv_mgpu *m, const uint8_t *texels, int voxel_type, int nx, int ny, int nz,
                              int slices_per_call, const float tf[1024]);

/* One frame: every device marches its bands (opts->shard_* are overwritten; other fields as in vv_render), then one
 * RCCL gather.  rgba_out: W*H*4 bytes on the host (out_on_device = 0) or on device 0 (1).  Returns when the frame is
 * complete.  Pixels the reference never writes (column W-1, row H-1) keep the caller's bytes. */
int  vv_mgpu_render(vv_mgpu *m, int width, int height, const struct slice_params *slice,
                    const struct camera_params *camera, const struct shading_params *shading,
                    const vv_ray_source *rays, const vv_render_options *opts,
                    uint8_t *rgba_out, int out_on_device);
/* The gather's bookkeeping (host arithmetic, no device): band b of a frame of height H rendered on n devices belongs to
 * *rank = b % n; of its 56 pixel rows that rank writes [*y_begin, *y_end) -- clipped to the slab-row range of the options
 * (0, 0 = all) and to row H-2, since row H-1 is never written (kernel.cu:297-298).  vv_mgpu_render sends exactly these
 * rows and copies columns 0 .. W-2 of them into the destination. */
int  vv_mgpu_band_rows(int height, int n_devices, int slab_row_begin, int slab_row_end, int band,
                       int *rank, int *y_begin, int *y_end);
/* per-rank march time of the last frame in ms (hipEvent), and the gather's */
int  vv_mgpu_last_times(vv_mgpu *m, float *march_ms /* [n] */, float *gather_ms);
const char *vv_mgpu_last_error(const vv_mgpu *m);

#ifdef __cplusplus
}
#endif
#endif
