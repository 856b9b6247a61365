// vv_mgpu.cpp -- the C-ABI of include/volviz_mgpu.h: N devices in one process, bands gathered by RCCL.
#include "../../include/volviz_mgpu.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <cstring>
#include <string>
#include <vector>

namespace {
constexpr int kSlab = 14, kBand = 4;          // kernel.cu:418; bands of 4 slab rows (vv_render_options.shard_band)
}

struct vv_mgpu {
    int n = 0;
    std::vector<int> dev;
    std::vector<vv_context *> ctx;
    std::vector<ncclComm_t> comm;
    std::vector<hipStream_t> stream;
    std::vector<hipEvent_t> e0, e1;
    hipEvent_t g0 = nullptr, g1 = nullptr;
    std::vector<uint8_t *> frame; std::vector<size_t> frame_cap;     // per-device full-size frames (device 0: staging for host output)
    uint8_t *land = nullptr; size_t land_cap = 0;                    // device 0: where the other ranks' bands are received
    std::string err;
};

static std::string g_merr;
static int mfail(vv_mgpu *m, int code, const std::string &s) { if (m) m->err = s; else g_merr = s; return code; }
#define MH(m, e) do { hipError_t e_ = (e); if (e_ != hipSuccess) return mfail((m), VV_ERR_DEVICE, std::string(#e) + ": " + hipGetErrorString(e_)); } while (0)
#define MN(m, e) do { ncclResult_t r_ = (e); if (r_ != ncclSuccess) return mfail((m), VV_ERR_DEVICE, std::string(#e) + ": " + ncclGetErrorString(r_)); } while (0)

extern "C" {

const char *vv_mgpu_last_error(const vv_mgpu *m) { return m ? m->err.c_str() : g_merr.c_str(); }
int vv_mgpu_size(const vv_mgpu *m) { return m ? m->n : 0; }
vv_context *vv_mgpu_context(vv_mgpu *m, int r) { return (m && r >= 0 && r < m->n) ? m->ctx[r] : nullptr; }

int vv_mgpu_init(int n, const int *devices, vv_mgpu **out)
{
    if (!out || n < 1) return mfail(nullptr, VV_ERR_INVALID, "vv_mgpu_init: bad argument");
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count < 1) return mfail(nullptr, VV_ERR_DEVICE, "vv_mgpu_init: no HIP device");
    vv_mgpu *m = new vv_mgpu();
    m->n = n;
    for (int r = 0; r < n; ++r) {
        const int d = devices ? devices[r] : r;
        if (d < 0 || d >= count) { delete m; return mfail(nullptr, VV_ERR_INVALID, "vv_mgpu_init: device index out of range"); }
        for (int q = 0; q < r; ++q) if (m->dev[q] == d) { delete m; return mfail(nullptr, VV_ERR_INVALID, "vv_mgpu_init: a device listed twice"); }
        m->dev.push_back(d);
    }
    m->ctx.assign(n, nullptr); m->stream.assign(n, nullptr); m->e0.assign(n, nullptr); m->e1.assign(n, nullptr);
    m->frame.assign(n, nullptr); m->frame_cap.assign(n, 0);
    for (int r = 0; r < n; ++r) {
        int rc = vv_init(m->dev[r], &m->ctx[r]);
        if (rc) { std::string e = vv_last_error(nullptr); vv_mgpu_shutdown(m); return mfail(nullptr, rc, "vv_mgpu_init: " + e); }
        if (hipSetDevice(m->dev[r]) != hipSuccess || hipStreamCreate(&m->stream[r]) != hipSuccess ||
            hipEventCreate(&m->e0[r]) != hipSuccess || hipEventCreate(&m->e1[r]) != hipSuccess) {
            vv_mgpu_shutdown(m); return mfail(nullptr, VV_ERR_DEVICE, "vv_mgpu_init: stream / event creation failed");
        }
    }
    if (hipSetDevice(m->dev[0]) != hipSuccess || hipEventCreate(&m->g0) != hipSuccess || hipEventCreate(&m->g1) != hipSuccess) {
        vv_mgpu_shutdown(m); return mfail(nullptr, VV_ERR_DEVICE, "vv_mgpu_init: event creation failed");
    }
    if (n > 1) {
        m->comm.assign(n, nullptr);
        ncclResult_t r = ncclCommInitAll(m->comm.data(), n, m->dev.data());
        if (r != ncclSuccess) { std::string e = ncclGetErrorString(r); m->comm.clear(); vv_mgpu_shutdown(m); return mfail(nullptr, VV_ERR_DEVICE, "ncclCommInitAll: " + e); }
    }
    *out = m;
    return VV_OK;
}

int vv_mgpu_shutdown(vv_mgpu *m)
{
    if (!m) return VV_OK;
    for (int r = 0; r < m->n; ++r) {
        (void)hipSetDevice(m->dev[r]);
        if (r < (int)m->stream.size() && m->stream[r]) (void)hipStreamSynchronize(m->stream[r]);
        if (r < (int)m->comm.size() && m->comm[r]) (void)ncclCommDestroy(m->comm[r]);
        if (r < (int)m->frame.size() && m->frame[r]) (void)hipFree(m->frame[r]);
        if (r < (int)m->e0.size() && m->e0[r]) (void)hipEventDestroy(m->e0[r]);
        if (r < (int)m->e1.size() && m->e1[r]) (void)hipEventDestroy(m->e1[r]);
        if (r < (int)m->stream.size() && m->stream[r]) (void)hipStreamDestroy(m->stream[r]);
        if (r < (int)m->ctx.size() && m->ctx[r]) (void)vv_shutdown(m->ctx[r]);
    }
    if (m->land) { (void)hipSetDevice(m->dev[0]); (void)hipFree(m->land); }
    if (m->g0) (void)hipEventDestroy(m->g0);
    if (m->g1) (void)hipEventDestroy(m->g1);
    delete m;
    return VV_OK;
}

int vv_mgpu_load_volume_u8(vv_mgpu *m, const uint8_t *t, size_t size, int nx, int ny, int nz, const float tf[1024])
{
    if (!m) return mfail(nullptr, VV_ERR_INVALID, "vv_mgpu_load_volume_u8: NULL");
    for (int r = 0; r < m->n; ++r) { int rc = vv_load_volume_u8(m->ctx[r], t, size, nx, ny, nz, tf); if (rc) return mfail(m, rc, vv_last_error(m->ctx[r])); }
    return VV_OK;
}
int vv_mgpu_load_volume_f32(vv_mgpu *m, const float *t, size_t size, int nx, int ny, int nz, const float tf[1024])
{
    if (!m) return mfail(nullptr, VV_ERR_INVALID, "vv_mgpu_load_volume_f32: NULL");
    for (int r = 0; r < m->n; ++r) { int rc = vv_load_volume_f32(m->ctx[r], t, size, nx, ny, nz, tf); if (rc) return mfail(m, rc, vv_last_error(m->ctx[r])); }
    return VV_OK;
}

int vv_mgpu_generate_default_brain(vv_mgpu *m, int vtype, int nx, int ny, int nz, const float tf[1024])
{
    if (!m || nx < 1 || ny < 1 || nz < 1) return mfail(m, VV_ERR_INVALID, "vv_mgpu_generate_default_brain: bad argument");
    const size_t nvox = (size_t)nx * ny * nz;
    for (int r = 0; r < m->n; ++r) {
        MH(m, hipSetDevice(m->dev[r]));
        uint8_t *d8 = nullptr; float *d32 = nullptr;
        MH(m, hipMalloc((void **)&d8, nvox));
        int rc = vv_generate_default_brain(m->ctx[r], d8, 1, nx, ny, nz, nullptr);
        if (!rc && vtype == VV_VOXEL_F32) {
            if (hipMalloc((void **)&d32, nvox * 4) != hipSuccess) rc = VV_ERR_NOMEM;
            if (!rc) rc = vv_promote_u8_to_f32(m->ctx[r], d8, d32, nvox, nullptr);
            if (!rc) rc = vv_load_volume_device(m->ctx[r], d32, VV_VOXEL_F32, nx, ny, nz, tf, nullptr);
        } else if (!rc) rc = vv_load_volume_device(m->ctx[r], d8, VV_VOXEL_U8, nx, ny, nz, tf, nullptr);
        if (d32) (void)hipFree(d32);
        (void)hipFree(d8);
        if (rc) return mfail(m, rc, std::string("vv_mgpu_generate_default_brain: ") + vv_last_error(m->ctx[r]));
    }
    return VV_OK;
}

int vv_mgpu_stream_volume_u8(vv_mgpu *m, const uint8_t *t, int vtype, int nx, int ny, int nz, int per, const float tf[1024])
{
    if (!m || !t || per < 1) return mfail(m, VV_ERR_INVALID, "vv_mgpu_stream_volume_u8: bad argument");
    for (int r = 0; r < m->n; ++r) { int rc = vv_load_volume_stream_begin(m->ctx[r], vtype, nx, ny, nz, tf); if (rc) return mfail(m, rc, vv_last_error(m->ctx[r])); }
    // Slab by slab: the slab is enqueued on EVERY device (each has its own PCIe link and copy engine), then the
    // source reads are waited for -- not the promotion kernels, which overlap the next slab's copies.
    const size_t slice = (size_t)nx * ny;
    int rc = VV_OK, bad = -1;
    for (int z = 0; z < nz && !rc; z += per) {
        const int n = z + per <= nz ? per : nz - z;
        for (int r = 0; r < m->n && !rc; ++r) { rc = vv_load_volume_stream_slices_async(m->ctx[r], t + (size_t)z * slice, VV_VOXEL_U8, z, n); if (rc) bad = r; }
        for (int r = 0; r < m->n; ++r) { int r2 = vv_load_volume_stream_wait_source(m->ctx[r]); if (r2 && !rc) { rc = r2; bad = r; } }
    }
    for (int r = 0; r < m->n; ++r) { int r2 = vv_load_volume_stream_end(m->ctx[r]); if (r2 && !rc) { rc = r2; bad = r; } }
    if (rc) return mfail(m, rc, vv_last_error(m->ctx[bad < 0 ? 0 : bad]));
    return VV_OK;
}

// Band bookkeeping of the gather (host arithmetic only; unit-tested without a GPU).  Band b = pixel rows
// [56 b, 56 b + 56) belongs to rank b % n.  Of those rows a rank's vv_render writes the ones inside the slab-row range
// [rb, re) of the options (0, 0 = all) and never row H-1 (kernel.cu:297-298); columns 0 .. W-2 of each.
int vv_mgpu_band_rows(int H, int n, int slab_row_begin, int slab_row_end, int band, int *rank, int *y_begin, int *y_end)
{
    if (H < 1 || n < 1 || band < 0 || !rank || !y_begin || !y_end) return VV_ERR_INVALID;
    const int band_px = kBand * kSlab, nbands = (H + band_px - 1) / band_px;
    if (band >= nbands) return VV_ERR_INVALID;
    const int nby = H / kSlab + ((H % kSlab) ? 1 : 0);
    int rb = 0, re = nby;
    if (!(slab_row_begin == 0 && slab_row_end == 0)) { rb = slab_row_begin; re = slab_row_end; }
    if (rb < 0 || re > nby || rb > re) return VV_ERR_INVALID;
    int ya = band * band_px, yb = ya + band_px;
    if (ya < rb * kSlab) ya = rb * kSlab;
    if (yb > re * kSlab) yb = re * kSlab;
    if (yb > H - 1) yb = H - 1;                    // row H-1 is never written
    if (ya > yb) ya = yb;                         // empty: the band lies outside the slab-row range
    *rank = band % n; *y_begin = ya; *y_end = yb;
    return VV_OK;
}

int vv_mgpu_render(vv_mgpu *m, int W, int H, const struct slice_params *slice, const struct camera_params *cam,
                   const struct shading_params *shading, const vv_ray_source *rays, const vv_render_options *opts,
                   uint8_t *rgba_out, int out_on_device)
{
    if (!m || !rgba_out || W < 1 || H < 1) return mfail(m, VV_ERR_INVALID, "vv_mgpu_render: bad argument");
    const int n = m->n;
    const size_t fb = (size_t)W * H * 4, row = (size_t)W * 4;
    // frames: rank r > 0 marches into its own full-size frame; device 0 holds the destination (the caller's buffer, or a
    // staging frame seeded with the caller's bytes) and, for n > 1, a landing frame the bands are received into
    for (int r = 0; r < n; ++r) {
        const bool need = r > 0 || !out_on_device;
        if (need && m->frame_cap[r] < fb) {
            MH(m, hipSetDevice(m->dev[r]));
            if (m->frame[r]) MH(m, hipFree(m->frame[r]));
            m->frame[r] = nullptr; m->frame_cap[r] = 0;
            MH(m, hipMalloc((void **)&m->frame[r], fb));
            m->frame_cap[r] = fb;
        }
    }
    if (n > 1 && m->land_cap < fb) {
        MH(m, hipSetDevice(m->dev[0]));
        if (m->land) MH(m, hipFree(m->land));
        m->land = nullptr; m->land_cap = 0;
        MH(m, hipMalloc((void **)&m->land, fb));
        m->land_cap = fb;
    }
    uint8_t *dst0 = out_on_device ? rgba_out : m->frame[0];
    MH(m, hipSetDevice(m->dev[0]));
    if (!out_on_device) MH(m, hipMemcpyAsync(dst0, rgba_out, fb, hipMemcpyHostToDevice, m->stream[0]));   // untouched pixels keep the caller's bytes
    // ---- march: every device its bands, all enqueued before anything is waited for ----
    int rc = VV_OK;
    std::string why;
    for (int r = 0; r < n && !rc; ++r) {
        vv_render_options o;
        if (opts) o = *opts; else memset(&o, 0, sizeof o);
        if (n > 1) { o.shard_band = kBand; o.shard_count = n; o.shard_index = r; }
        if (hipSetDevice(m->dev[r]) != hipSuccess || hipEventRecord(m->e0[r], m->stream[r]) != hipSuccess) { rc = VV_ERR_DEVICE; why = "event record failed"; break; }
        rc = vv_render(m->ctx[r], W, H, slice, cam, shading, rays, &o, r == 0 ? dst0 : m->frame[r], 1, (void *)m->stream[r]);
        if (rc) { why = std::string("rank ") + std::to_string(r) + ": " + vv_last_error(m->ctx[r]); break; }
        if (hipEventRecord(m->e1[r], m->stream[r]) != hipSuccess) { rc = VV_ERR_DEVICE; why = "event record failed"; }
    }
    // ---- gather: one send / receive pair per band of a rank > 0, whole pixel rows (contiguous) into the landing frame; then
    //      only the pixels that rank wrote -- columns 0 .. W-2 of the rows vv_mgpu_band_rows() names -- go into the destination,
    //      so column W-1 and row H-1 keep the caller's bytes (a rank's own frame is never seeded) ----
    const int sr0 = opts ? opts->slab_row_begin : 0, sr1 = opts ? opts->slab_row_end : 0;
    if (!rc && hipSetDevice(m->dev[0]) == hipSuccess) (void)hipEventRecord(m->g0, m->stream[0]);
    if (!rc && n > 1) {
        const int band_px = kBand * kSlab, nbands = (H + band_px - 1) / band_px;
        ncclResult_t nr = ncclGroupStart();
        bool open = nr == ncclSuccess;
        for (int b = 0; b < nbands && nr == ncclSuccess; ++b) {
            int r, ya, yb;
            if (vv_mgpu_band_rows(H, n, sr0, sr1, b, &r, &ya, &yb) != VV_OK) { rc = VV_ERR_INVALID; why = "slab row range out of bounds"; break; }
            if (r == 0 || yb <= ya) continue;
            const size_t off = (size_t)ya * row, bytes = (size_t)(yb - ya) * row;
            nr = ncclSend(m->frame[r] + off, bytes, ncclUint8, 0, m->comm[r], m->stream[r]);
            if (nr == ncclSuccess) nr = ncclRecv(m->land + off, bytes, ncclUint8, r, m->comm[0], m->stream[0]);
        }
        if (open) { ncclResult_t ne = ncclGroupEnd(); if (nr == ncclSuccess) nr = ne; }      // a group is never left open
        if (nr != ncclSuccess && !rc) { rc = VV_ERR_DEVICE; why = std::string("RCCL: ") + ncclGetErrorString(nr); }
        if (!rc && W >= 2) {
            if (hipSetDevice(m->dev[0]) != hipSuccess) { rc = VV_ERR_DEVICE; why = "hipSetDevice failed"; }
            for (int b = 0; b < nbands && !rc; ++b) {
                int r, ya, yb;
                (void)vv_mgpu_band_rows(H, n, sr0, sr1, b, &r, &ya, &yb);
                if (r == 0 || yb <= ya) continue;
                if (hipMemcpy2DAsync(dst0 + (size_t)ya * row, row, m->land + (size_t)ya * row, row, (size_t)(W - 1) * 4, (size_t)(yb - ya),
                                     hipMemcpyDeviceToDevice, m->stream[0]) != hipSuccess) { rc = VV_ERR_DEVICE; why = "band copy failed"; }
            }
        }
    }
    if (!rc && hipSetDevice(m->dev[0]) == hipSuccess) (void)hipEventRecord(m->g1, m->stream[0]);
    if (!rc && !out_on_device && hipMemcpyAsync(rgba_out, dst0, fb, hipMemcpyDeviceToHost, m->stream[0]) != hipSuccess) { rc = VV_ERR_DEVICE; why = "read-back failed"; }
    // every stream is drained on every path: nothing of this frame is in flight when the call returns
    for (int r = n - 1; r >= 0; --r) {
        if (hipSetDevice(m->dev[r]) != hipSuccess || hipStreamSynchronize(m->stream[r]) != hipSuccess) { if (!rc) { rc = VV_ERR_DEVICE; why = "stream synchronisation failed"; } }
    }
    (void)hipGetLastError();
    if (rc) return mfail(m, rc, "vv_mgpu_render: " + why);
    return VV_OK;
}

int vv_mgpu_last_times(vv_mgpu *m, float *march_ms, float *gather_ms)
{
    if (!m) return VV_ERR_INVALID;
    for (int r = 0; r < m->n && march_ms; ++r) {
        if (hipSetDevice(m->dev[r]) != hipSuccess || hipEventElapsedTime(&march_ms[r], m->e0[r], m->e1[r]) != hipSuccess) return VV_ERR_DEVICE;
    }
    if (gather_ms && (hipSetDevice(m->dev[0]) != hipSuccess || hipEventElapsedTime(gather_ms, m->g0, m->g1) != hipSuccess)) return VV_ERR_DEVICE;
    return VV_OK;
}

} // extern "C"
