// vv_api.cpp -- the C-ABI of include/volviz.h over the HIP kernels.
//
// Replaces the host half of kernel.cu (initCuda/registerCudaResources/runCuda/
// cudaLoadVolume, kernel.cu:369-498; invoke_*_slice_kernel, kernel.cu:506-541).
// There is no CPU fallback: every entry point that computes needs a HIP device and
// fails with VV_ERR_DEVICE otherwise.
#include "../../include/volviz.h"
#include "vv_kernels.h"

#include <hip/hip_runtime.h>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <algorithm>
#include <vector>

using namespace vv;

// Developer / experiment knobs (VV_* environment variables).  Read when the context is created and again at every
// volume load -- never on the per-frame path.
struct vv_knobs {
    int tile_log2w = -1, xcd_band = -1, unroll = -1, lds_reserve = -1, lds_reserve_phong = -1;
    int bricked = -1, zpair = -1, zfast = -1, force_big = 0;
    int block_w = -1, tail = -1, rect = -1, lpt = -1, lpt_run = -1, phong_bricks = -1;
    static int geti(const char *name, int dflt) { const char *e = getenv(name); return e ? atoi(e) : dflt; }
    void read()
    {
        tile_log2w = geti("VV_TILE_LOG2W", -1); xcd_band = geti("VV_XCD_BAND", -1); unroll = geti("VV_UNROLL", -1);
        lds_reserve = geti("VV_LDS_RESERVE", -1); lds_reserve_phong = geti("VV_LDS_RESERVE_PHONG", -1);
        bricked = geti("VV_BRICKED", -1); zpair = geti("VV_ZPAIR", -1);
        block_w = geti("VV_BLOCK_W", -1); tail = geti("VV_TAIL", -1);
        zfast = geti("VV_ZFAST", -1); force_big = getenv("VV_FORCE_BIG") != nullptr;
        rect = geti("VV_RECT", -1); lpt = geti("VV_LPT", -1); lpt_run = geti("VV_LPT_RUN", -1); phong_bricks = geti("VV_PHONG_BRICKS", -1);
    }
};

struct vv_context {
    int device = 0;
    vv_knobs knobs;
    hipStream_t stream = nullptr;          // used when the caller passes no stream
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false, time_frames = true;      // vv_set_frame_timing
    // volume
    void *d_vol = nullptr; size_t vol_bytes = 0; int vtype = VV_VOXEL_U8; int nx = 0, ny = 0, nz = 0;
    size_t row_pitch = 0, slice_pitch = 0, alloc_bytes = 0;   // linear layout in HBM (bytes); vol_bytes stays nx*ny*nz*voxel
    // bricked copy of an f32 volume for views off the memory axis (built on first use, dropped on reload)
    void *d_bricks = nullptr; bool bricks_valid = false, bricks_failed = false; uint32_t b_sy = 0, b_sz64 = 0; size_t bricks_bytes = 0;
    // z-pair copy of an f32 volume for views along the memory axis (same life cycle)
    void *d_zpair = nullptr; bool zpair_valid = false; uint32_t zp_row = 0, zp_slab = 0; size_t zpair_bytes = 0;
    void *d_zfast = nullptr; bool zfast_valid = false, zfast_failed = false; uint32_t zf_row = 0; uint64_t zf_slice = 0; size_t zfast_bytes = 0;   // z-fastest copy (side views)
    void *d_xpair = nullptr; bool xpair_valid = false, xpair_failed = false; uint32_t xp_row = 0, xp_slab = 0; size_t xpair_bytes = 0;           // x-pair copy (side views of small / u8 volumes)
    // residency policy of the optional copies (vv_set_layout_policy): HBM budget for all of them together (0 = default, layout_budget()),
    // whether vv_render may build a missing copy by itself, and when each copy was last sampled (the least recently used one goes first)
    size_t layout_budget_bytes = 0; bool build_in_render = true;
    unsigned long long frame_no = 0, last_used[4] = {0, 0, 0, 0};          // CP_BRICKS, CP_ZPAIR, CP_ZFAST, CP_XPAIR
    unsigned long long builds_in_render = 0;                                  // copies built inside vv_render since the volume was loaded
    float last_density = 1e9f;                                                // what the launch policy took the last frame's sampling density to be
    // transfer function
    float4 *d_tf = nullptr; bool tf_gray = false; bool have_tf = false;
    bool tf_alpha_unit = false;          // every opacity of the table lies in [0, 1]: accumulated opacity never decreases
    // scratch
    float *d_rad = nullptr; size_t rad_cap = 0;
    uint32_t *d_order = nullptr; size_t order_cap = 0;      // StripMap::order of the frame in flight
    uint8_t *d_frame = nullptr; size_t frame_cap = 0;
    uint8_t *d_img = nullptr; size_t img_cap = 0;
    float *d_slice = nullptr; size_t slice_cap = 0;
    float *d_gen = nullptr; size_t gen_cap = 0;       // per-axis tables of the ellipsoid generator
    // launch-policy estimate taken from device-resident first-pass images (vv_render, synchronous calls), kept while the view stays
    struct view_key_t { float cam[8]; const void *front, *back; int img_w, img_h, W, H; } view_key;
    bool view_key_valid = false, view_est_ok = false; float view_side[3] = {0, 0, 0}, view_px_cube = 0.f;
    std::vector<uint8_t> row_buf;
    int last_launch[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // vv_debug_last_launch
    unsigned long long *d_counter = nullptr;
    bool counter_valid = false;          // counters of the last instrumented frame
    // streamed upload
    hipStream_t copy_stream = nullptr, promo_stream = nullptr;   // H2D copies / u8 -> f32 promotion kernels
    void *pin[2] = {nullptr, nullptr}; hipEvent_t pin_ev[2] = {nullptr, nullptr}; int pin_next = 0;   // pin_ev[b]: staging buffers b free again
    hipEvent_t src_ev[2] = {nullptr, nullptr}; int src_last = -1;       // src_ev[b]: the copy engine has read piece b's source
    uint8_t *d_stage[2] = {nullptr, nullptr};     // u8 slices awaiting promotion
    bool streaming = false;
    std::string err;
};

static std::string g_err;

static int fail(vv_context *c, int code, const std::string &msg)
{
    if (c) c->err = msg; else g_err = msg;
    return code;
}
#define HIPCHK(c, expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) \
    return fail((c), VV_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)

// `stream` of the C-ABI: NULL = the context's own stream, the call returns when the work is done;
// VV_STREAM_DEFAULT_ASYNC = the device's default (null) stream, enqueue only; anything else = that stream, enqueue only.
static inline hipStream_t pick_stream(const vv_context *c, void *stream)
{
    if (!stream) return c->stream;
    if (stream == VV_STREAM_DEFAULT_ASYNC) return (hipStream_t)0;
    return (hipStream_t)stream;
}

static void drop_bricks(vv_context *c)
{
    if (c->d_bricks) (void)hipFree(c->d_bricks);
    c->d_bricks = nullptr; c->bricks_valid = false; c->bricks_failed = false; c->bricks_bytes = 0;
    if (c->d_zpair) (void)hipFree(c->d_zpair);
    c->d_zpair = nullptr; c->zpair_valid = false; c->zpair_bytes = 0;
    if (c->d_zfast) (void)hipFree(c->d_zfast);
    c->d_zfast = nullptr; c->zfast_valid = false; c->zfast_failed = false; c->zfast_bytes = 0;
    if (c->d_xpair) (void)hipFree(c->d_xpair);
    c->d_xpair = nullptr; c->xpair_valid = false; c->xpair_failed = false; c->xpair_bytes = 0;
    for (int k = 0; k < 4; ++k) c->last_used[k] = 0;
    c->builds_in_render = 0;
    c->view_key_valid = false;             // (a new volume: the launch-policy estimate taken from first-pass images is re-taken)
}

static int ensure(vv_context *c, void **p, size_t *cap, size_t need)
{
    if (*cap >= need && *p) return VV_OK;
    if (*p) { HIPCHK(c, hipFree(*p)); *p = nullptr; *cap = 0; }
    HIPCHK(c, hipMalloc(p, need));
    *cap = need;
    return VV_OK;
}

static int finalize_layout(vv_context *c, hipStream_t st);
static int camera_basis(vv_context *c, FrameParams &P, const camera_params *cam, const vv_ray_source *rays, int W, int H);
typedef vv_context::view_key_t vv_view_key;

// ---- launch policy for image ray sources (speed only; the pixels never depend on it) ----
// An image source may carry the camera that drew it in the fields analytic sources use (look, up, aspect): a hint.
static bool image_source_has_hint(const vv_ray_source *r)
{
    if (r->mode != VV_RAYS_IMAGES) return false;
    const float l2 = r->look[0] * r->look[0] + r->look[1] * r->look[1] + r->look[2] * r->look[2];
    const float cx = r->look[1] * r->up[2] - r->look[2] * r->up[1], cy = r->look[2] * r->up[0] - r->look[0] * r->up[2], cz = r->look[0] * r->up[1] - r->look[1] * r->up[0];
    return std::isfinite(l2) && l2 > 0.f && std::isfinite(cx + cy + cz) && (cx * cx + cy * cy + cz * cz) > 0.f;
}
static int clamp_row(long long t, int n) { return (int)(t < 0 ? 0 : (t > n - 1 ? n - 1 : t)); }
static vv_view_key view_key_of(const camera_params *cam, const vv_ray_source *r, int W, int H)
{
    vv_view_key k;
    memset(&k, 0, sizeof k);
    memcpy(k.cam, cam, sizeof(float) * 8);
    k.front = r->front; k.back = r->back; k.img_w = r->img_w; k.img_h = r->img_h; k.W = W; k.H = H;
    return k;
}
// The centre pixel row of the frame, looked up in the images exactly as the kernels do (kernel.cu:317-321): where it enters and
// leaves the cube's silhouette, and how the rays' mid-depth points move through cube space between those pixels.  side = that
// direction (unit), px_cube = its length per pixel.  False when the row misses the cube.
static bool estimate_view_from_images(const uint8_t *front, const uint8_t *back, int img_w, int img_h, int W, int H, float side[3], float *px_cube)
{
    if (!front || !back || img_w < 1 || img_h < 1 || W < 8 || H < 1) return false;
    const int y = H / 2;
    const int ty = clamp_row((long long)floorf((float)y / (float)H * (float)img_h), img_h);
    const uint8_t *fr = front + (size_t)ty * img_w * 4, *br = back + (size_t)ty * img_w * 4;
    auto texel = [&](int x) { return clamp_row((long long)floorf((float)x / (float)W * (float)img_w), img_w); };
    auto hit = [&](int x) { const int t = texel(x) * 4; return fr[t] != br[t] || fr[t + 1] != br[t + 1] || fr[t + 2] != br[t + 2]; };
    int xl = -1, xr = -1;
    for (int x = 0; x <= W - 2; ++x) if (hit(x)) { xl = x; break; }
    if (xl < 0) return false;
    for (int x = W - 2; x >= xl; --x) if (hit(x)) { xr = x; break; }
    if (xr - xl < 8) return false;
    const int a = (xl + xr) / 2 - (xr - xl) / 8, b = (xl + xr) / 2 + (xr - xl) / 8;   // the middle quarter: rays that cross the cube front to back
    float d[3], len2 = 0.f;
    for (int k = 0; k < 3; ++k) {
        const int ta = texel(a) * 4 + k, tb = texel(b) * 4 + k;
        d[k] = (((float)fr[tb] + (float)br[tb]) - ((float)fr[ta] + (float)br[ta])) / (2.f * 255.f);
        len2 += d[k] * d[k];
    }
    if (!(len2 > 0.f)) return false;
    const float len = sqrtf(len2);
    side[0] = d[0] / len; side[1] = d[1] / len; side[2] = d[2] / len;
    *px_cube = len / (float)(b - a);
    return true;
}
static bool ensure_bricks(vv_context *c, hipStream_t st);
static bool ensure_zpair(vv_context *c, hipStream_t st);
static bool ensure_zfast(vv_context *c, hipStream_t st);
static bool ensure_xpair(vv_context *c, hipStream_t st);

// ---- residency of the optional copies -----------------------------------------------------------------------------------------
// All copies of a volume together stay within a budget (default: the larger of 8 GiB and 2.5 x the linear volume -- room for the bricked and the
// z-fastest copy of an f32 volume, C3: 9.7 GB of copies beside the 4.3 GB volume, C5: 72 GiB beside 32 GiB; every copy of a u8 volume up to 1 GiB).
// A copy that would not fit evicts copies no frame has sampled more recently (least recently used first); if that is not enough it is not built
// and the frame takes the next layout of the policy.  Results never depend on any of this.
enum { CP_BRICKS = 0, CP_ZPAIR = 1, CP_ZFAST = 2, CP_XPAIR = 3, CP_N = 4 };
static size_t layout_budget(const vv_context *c)
{
    if (c->layout_budget_bytes) return c->layout_budget_bytes;
    return std::max<size_t>((size_t)8 << 30, c->alloc_bytes / 2 * 5);
}
static bool copy_valid(const vv_context *c, int k) { return k == CP_BRICKS ? c->bricks_valid : k == CP_ZPAIR ? c->zpair_valid : k == CP_ZFAST ? c->zfast_valid : c->xpair_valid; }
static size_t copy_bytes(const vv_context *c, int k) { return !copy_valid(c, k) ? 0 : k == CP_BRICKS ? c->bricks_bytes : k == CP_ZPAIR ? c->zpair_bytes : k == CP_ZFAST ? c->zfast_bytes : c->xpair_bytes; }
static void copy_drop(vv_context *c, int k)
{
    void **p = k == CP_BRICKS ? &c->d_bricks : k == CP_ZPAIR ? &c->d_zpair : k == CP_ZFAST ? &c->d_zfast : &c->d_xpair;
    if (*p) (void)hipFree(*p);                       // (hipFree waits for the device: frames still in flight on a caller's stream have finished with it)
    *p = nullptr;
    if (k == CP_BRICKS) { c->bricks_valid = false; c->bricks_bytes = 0; } else if (k == CP_ZPAIR) { c->zpair_valid = false; c->zpair_bytes = 0; }
    else if (k == CP_ZFAST) { c->zfast_valid = false; c->zfast_bytes = 0; } else { c->xpair_valid = false; c->xpair_bytes = 0; }
}
// Room for `need` more bytes of copies?  `keep`: bit mask of copies that must stay (the ones the frame being set up samples or builds from).
static bool make_room(vv_context *c, size_t need, unsigned keep)
{
    const size_t budget = layout_budget(c);
    if (need > budget) return false;
    for (;;) {
        size_t used = 0;
        for (int k = 0; k < CP_N; ++k) used += copy_bytes(c, k);
        if (used + need <= budget) return true;
        int victim = -1;
        for (int k = 0; k < CP_N; ++k)
            if (copy_valid(c, k) && !(keep & (1u << k)) && (victim < 0 || c->last_used[k] < c->last_used[victim])) victim = k;
        if (victim < 0) return false;
        copy_drop(c, victim);
    }
}


extern "C" {

const char *vv_last_error(const vv_context *ctx) { return ctx ? ctx->err.c_str() : g_err.c_str(); }

// ---- lifecycle: initCuda, kernel.cu:369-373 ------------------------------------
int vv_init(int device, vv_context **out)
{
    if (!out) return fail(nullptr, VV_ERR_INVALID, "vv_init: out_ctx is NULL");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0)
        return fail(nullptr, VV_ERR_DEVICE, "vv_init: no HIP device available (this library has no CPU path)");
    if (device < 0) { if (hipGetDevice(&device) != hipSuccess) device = 0; }
    if (device >= count) return fail(nullptr, VV_ERR_INVALID, "vv_init: device index out of range");
    vv_context *c = new vv_context();
    c->device = device;
    c->knobs.read();
    if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&c->stream) != hipSuccess ||
        hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess ||
        hipMalloc((void **)&c->d_counter, 16 * sizeof(unsigned long long)) != hipSuccess ||
        hipMalloc((void **)&c->d_tf, 256 * sizeof(float4)) != hipSuccess) {
        delete c;
        return fail(nullptr, VV_ERR_DEVICE, "vv_init: device set-up failed");
    }
    *out = c;
    return VV_OK;
}

int vv_shutdown(vv_context *c)
{
    if (!c) return VV_OK;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    if (c->d_vol) hipFree(c->d_vol);
    drop_bricks(c);
    if (c->d_tf) hipFree(c->d_tf);
    if (c->d_rad) hipFree(c->d_rad);
    if (c->d_order) hipFree(c->d_order);
    if (c->d_frame) hipFree(c->d_frame);
    if (c->d_img) hipFree(c->d_img);
    if (c->d_slice) hipFree(c->d_slice);
    if (c->d_gen) hipFree(c->d_gen);
    if (c->d_counter) hipFree(c->d_counter);
    for (int i = 0; i < 2; ++i) {
        if (c->pin[i]) hipHostFree(c->pin[i]);
        if (c->pin_ev[i]) hipEventDestroy(c->pin_ev[i]);
        if (c->src_ev[i]) hipEventDestroy(c->src_ev[i]);
        if (c->d_stage[i]) hipFree(c->d_stage[i]);
    }
    if (c->copy_stream) hipStreamDestroy(c->copy_stream);
    if (c->promo_stream) hipStreamDestroy(c->promo_stream);
    if (c->ev0) hipEventDestroy(c->ev0);
    if (c->ev1) hipEventDestroy(c->ev1);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
    return VV_OK;
}

// ---- volume + TF: cudaLoadVolume, kernel.cu:456-498 ------------------------------
int vv_set_transfer_function(vv_context *c, const float tf[1024])
{
    if (!c || !tf) return fail(c, VV_ERR_INVALID, "vv_set_transfer_function: NULL argument");
    HIPCHK(c, hipSetDevice(c->device));
    for (int i = 0; i < 1024; ++i)
        if (!std::isfinite(tf[i])) return fail(c, VV_ERR_INVALID, "vv_set_transfer_function: table entries must be finite");
    bool gray = true, unit = true;
    for (int i = 0; i < 256; ++i)
        if (!(tf[4*i] == tf[4*i+1] && tf[4*i] == tf[4*i+2])) { gray = false; break; }
    for (int i = 0; i < 256; ++i)
        if (!(tf[4*i+3] >= 0.f && tf[4*i+3] <= 1.f)) { unit = false; break; }
    // pageable host memory: the copy is staged before hipMemcpy returns      kernel.cu:495-496
    HIPCHK(c, hipMemcpy(c->d_tf, tf, 1024 * sizeof(float), hipMemcpyHostToDevice));
    c->tf_gray = gray; c->tf_alpha_unit = unit; c->have_tf = true;
    return VV_OK;
}

static int install_volume(vv_context *c, const void *src, bool src_on_device, int vtype,
                          int nx, int ny, int nz, const float tf[1024], hipStream_t s)
{
    if (!c || !src) return fail(c, VV_ERR_INVALID, "load_volume: NULL argument");
    if (nx < 1 || ny < 1 || nz < 1) return fail(c, VV_ERR_INVALID, "load_volume: dims must be >= 1");
    if (vtype != VV_VOXEL_U8 && vtype != VV_VOXEL_F32) return fail(c, VV_ERR_INVALID, "load_volume: bad voxel type");
    const size_t vsz = vtype == VV_VOXEL_F32 ? 4 : 1;
    const size_t bytes = (size_t)nx * ny * nz * vsz;
    if ((size_t)nx * ny * vsz > 0xFFFFFFF0ull)
        return fail(c, VV_ERR_INVALID, "load_volume: one slice must stay below 4 GiB");
    if ((size_t)nx * vsz >= (1u << 24) || nx >= (1 << 24) || ny >= (1 << 24) || nz >= (1 << 24))
        return fail(c, VV_ERR_INVALID, "load_volume: a volume row must be below 16 MiB and each dimension below 2^24");
    HIPCHK(c, hipSetDevice(c->device));
    c->knobs.read();
    // one slice + one row + 16 bytes of zero padding: weight-0 corner fetches of edge
    // samples land here instead of needing index clamps (see vv_device.h VolumeView)
    const size_t pad = (size_t)nx * ny * vsz + 2 * (size_t)nx * vsz + 4096;
    drop_bricks(c);
    if (c->d_vol) { HIPCHK(c, hipFree(c->d_vol)); c->d_vol = nullptr; }   // the reference leaks here
    HIPCHK(c, hipMalloc(&c->d_vol, bytes + pad));
    hipStream_t st = s ? (s == (hipStream_t)VV_STREAM_DEFAULT_ASYNC ? (hipStream_t)0 : s) : c->stream;
    HIPCHK(c, hipMemsetAsync((char *)c->d_vol + bytes, 0, pad, st));
    if (src_on_device) HIPCHK(c, hipMemcpyAsync(c->d_vol, src, bytes, hipMemcpyDeviceToDevice, st));
    else { HIPCHK(c, hipMemcpyAsync(c->d_vol, src, bytes, hipMemcpyHostToDevice, st)); }
    c->vol_bytes = bytes; c->vtype = vtype; c->nx = nx; c->ny = ny; c->nz = nz;
    c->row_pitch = (size_t)nx * vsz; c->slice_pitch = c->row_pitch * ny; c->alloc_bytes = bytes + pad;
    { int rc = finalize_layout(c, st); if (rc) return rc; }
    if (!src_on_device || !s) HIPCHK(c, hipStreamSynchronize(st));
    if (tf) return vv_set_transfer_function(c, tf);
    return VV_OK;
}

int vv_load_volume_u8(vv_context *c, const uint8_t *texels, size_t size, int nx, int ny, int nz, const float tf[1024])
{
    if (size != (size_t)nx * ny * nz) return fail(c, VV_ERR_INVALID, "vv_load_volume_u8: size != nx*ny*nz");
    return install_volume(c, texels, false, VV_VOXEL_U8, nx, ny, nz, tf, nullptr);
}
int vv_load_volume_f32(vv_context *c, const float *texels, size_t size, int nx, int ny, int nz, const float tf[1024])
{
    if (size != (size_t)nx * ny * nz * 4) return fail(c, VV_ERR_INVALID, "vv_load_volume_f32: size != nx*ny*nz*4 bytes");
    return install_volume(c, texels, false, VV_VOXEL_F32, nx, ny, nz, tf, nullptr);
}
int vv_load_volume_device(vv_context *c, const void *dev, int vtype, int nx, int ny, int nz, const float tf[1024], void *stream)
{
    return install_volume(c, dev, true, vtype, nx, ny, nz, tf, (hipStream_t)stream);
}

// developer statistics of the last instrumented launch (staged kernel): [0] executed samples,
// [1] stages, [2] samples served from global memory, [3] bytes staged into LDS, [4] wave compute trips
int vv_prepare_layouts(vv_context *c, int which, void *stream)
{
    if (!c) return fail(nullptr, VV_ERR_INVALID, "vv_prepare_layouts: NULL context");
    if (!c->d_vol) return fail(c, VV_ERR_NO_VOLUME, "vv_prepare_layouts: no volume loaded");
    if (which & ~(VV_LAYOUT_BRICKED | VV_LAYOUT_ZPAIR | VV_LAYOUT_ZFAST | VV_LAYOUT_POLICY)) return fail(c, VV_ERR_INVALID, "vv_prepare_layouts: unknown layout bit");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = pick_stream(c, stream);
    const bool pair_ok = c->vtype == VV_VOXEL_U8 || c->vol_bytes <= (512ull << 20);          // (the z-pair / x-pair conditions of vv_render)
    if (which & VV_LAYOUT_POLICY) {
        // every copy vv_render's policy can pick for this volume, in the order of what an orbit needs most: oblique views, side views, the pair copies
        if ((size_t)c->nx * c->ny * c->nz >= (1ull << 21)) which |= VV_LAYOUT_BRICKED | VV_LAYOUT_ZFAST;
        if (pair_ok) which |= VV_LAYOUT_ZPAIR;
    }
    int built = 0;
    if (which & VV_LAYOUT_BRICKED) c->bricks_failed = false;       // an explicit request retries after an earlier shortage of HBM
    if (which & VV_LAYOUT_ZFAST) { c->zfast_failed = false; c->xpair_failed = false; }
    // (a copy built here counts as just used: the next one must not evict it to make room for itself)
    if ((which & VV_LAYOUT_BRICKED) && ensure_bricks(c, st)) { built |= VV_LAYOUT_BRICKED; c->last_used[CP_BRICKS] = ++c->frame_no; }
    if ((which & VV_LAYOUT_ZFAST) && ensure_zfast(c, st)) {
        built |= VV_LAYOUT_ZFAST; c->last_used[CP_ZFAST] = ++c->frame_no;
        // side views of unshaded frames take the x-pair copy where front views take the z-pair copy: built with the z-fastest copy it is made from
        if (pair_ok && ensure_xpair(c, st)) c->last_used[CP_XPAIR] = ++c->frame_no;
    }
    if ((which & VV_LAYOUT_ZPAIR) && ensure_zpair(c, st)) { built |= VV_LAYOUT_ZPAIR; c->last_used[CP_ZPAIR] = ++c->frame_no; }
    return built;
}

int vv_reread_env(vv_context *c)
{
    if (!c) return VV_ERR_INVALID;
    c->knobs.read();
    return VV_OK;
}

int vv_set_layout_policy(vv_context *c, unsigned long long budget_bytes, int build_in_render)
{
    if (!c) return fail(nullptr, VV_ERR_INVALID, "vv_set_layout_policy: NULL context");
    c->layout_budget_bytes = (size_t)budget_bytes;
    c->build_in_render = build_in_render != 0;
    return VV_OK;
}

int vv_layout_state(const vv_context *c, unsigned long long out[8])
{
    if (!c || !out) return VV_ERR_INVALID;
    out[0] = c->d_vol ? c->alloc_bytes : 0;
    for (int k = 0; k < CP_N; ++k) out[1 + k] = copy_bytes(c, k);
    out[5] = c->d_vol ? layout_budget(c) : c->layout_budget_bytes;
    out[6] = c->builds_in_render;
    out[7] = c->build_in_render ? 1 : 0;
    return VV_OK;
}

int vv_device_bytes(const vv_context *c, unsigned long long out[4])
{
    if (!c || !out) return VV_ERR_INVALID;
    out[0] = c->d_vol ? c->alloc_bytes : 0;
    out[1] = c->bricks_valid ? c->bricks_bytes : 0;
    out[2] = (c->zpair_valid ? c->zpair_bytes : 0) + (c->zfast_valid ? c->zfast_bytes : 0) + (c->xpair_valid ? c->xpair_bytes : 0);
    out[3] = c->rad_cap + c->frame_cap + c->img_cap + c->slice_cap + 4096 + 8 * sizeof(unsigned long long);
    return VV_OK;
}

int vv_debug_counters(vv_context *c, unsigned long long out[16])
{
    if (!c || !out || !c->counter_valid) return VV_ERR_INVALID;
    if (hipEventSynchronize(c->ev1) != hipSuccess) return VV_ERR_DEVICE;
    if (hipMemcpy(out, c->d_counter, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return VV_ERR_DEVICE;
    return VV_OK;
}

// ---- streamed upload ---------------------------------------------------------------------
static const size_t kStageBytes = 64u << 20;      // pinned / device staging buffers

int vv_load_volume_stream_begin(vv_context *c, int vtype, int nx, int ny, int nz, const float tf[1024])
{
    if (!c) return fail(nullptr, VV_ERR_INVALID, "stream_begin: NULL context");
    if (nx < 1 || ny < 1 || nz < 1 || (vtype != VV_VOXEL_U8 && vtype != VV_VOXEL_F32))
        return fail(c, VV_ERR_INVALID, "stream_begin: bad dims / voxel type");
    const size_t vsz = vtype == VV_VOXEL_F32 ? 4 : 1;
    if ((size_t)nx * ny * vsz > 0xFFFFFFF0ull) return fail(c, VV_ERR_INVALID, "stream_begin: one slice must stay below 4 GiB");
    if ((size_t)nx * vsz >= (1u << 24) || ny >= (1 << 24) || nz >= (1 << 24))
        return fail(c, VV_ERR_INVALID, "stream_begin: a volume row must be below 16 MiB and each dimension below 2^24");
    HIPCHK(c, hipSetDevice(c->device));
    c->knobs.read();
    const size_t bytes = (size_t)nx * ny * nz * vsz, pad = (size_t)nx * ny * vsz + 2 * (size_t)nx * vsz + 4096;
    drop_bricks(c);
    if (c->d_vol) { HIPCHK(c, hipFree(c->d_vol)); c->d_vol = nullptr; }
    HIPCHK(c, hipMalloc(&c->d_vol, bytes + pad));
    if (!c->copy_stream) HIPCHK(c, hipStreamCreate(&c->copy_stream));
    if (!c->promo_stream) HIPCHK(c, hipStreamCreate(&c->promo_stream));
    HIPCHK(c, hipMemsetAsync((char *)c->d_vol + bytes, 0, pad, c->copy_stream));
    c->src_last = -1;
    c->vol_bytes = bytes; c->vtype = vtype; c->nx = nx; c->ny = ny; c->nz = nz;
    c->row_pitch = (size_t)nx * vsz; c->slice_pitch = c->row_pitch * ny; c->alloc_bytes = bytes + pad;
    c->streaming = true;
    if (tf) return vv_set_transfer_function(c, tf);
    return VV_OK;
}

// Enqueue only.  H2D copies run on copy_stream, promotion kernels on promo_stream behind the copy's event, so the copy of
// the next piece overlaps the promotion of this one.  Two events per staging slot: src_ev (the copy engine has read the
// source: what a caller that refills a pinned buffer has to wait for) and pin_ev (copy AND promotion done: the slot's
// staging buffers may be reused).
int vv_load_volume_stream_slices_async(vv_context *c, const void *src, int src_type, int z0, int nslices)
{
    if (!c || !c->streaming) return fail(c, VV_ERR_INVALID, "stream_slices: no stream_begin");
    if (!src || z0 < 0 || nslices < 1 || z0 + nslices > c->nz) return fail(c, VV_ERR_INVALID, "stream_slices: bad slice range");
    if (src_type != c->vtype && !(src_type == VV_VOXEL_U8 && c->vtype == VV_VOXEL_F32))
        return fail(c, VV_ERR_INVALID, "stream_slices: source type must match the volume (or be u8 for an f32 volume)");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t ssz = src_type == VV_VOXEL_F32 ? 4 : 1, dsz = c->vtype == VV_VOXEL_F32 ? 4 : 1;
    const size_t slice_vox = (size_t)c->nx * c->ny;
    const bool promote = src_type != c->vtype;
    hipPointerAttribute_t attr;
    bool pinned = hipPointerGetAttributes(&attr, src) == hipSuccess && attr.type == hipMemoryTypeHost;
    (void)hipGetLastError();                       // an unregistered pointer is not an error for us
    // work in pieces of at most kStageBytes of source data
    size_t total_vox = slice_vox * (size_t)nslices, done = 0;
    const size_t piece_vox_max = ((kStageBytes / ssz) / 16) * 16;
    while (done < total_vox) {
        const size_t nv = std::min(piece_vox_max, total_vox - done);
        const int b = c->pin_next; c->pin_next ^= 1;
        if (!c->pin_ev[b]) HIPCHK(c, hipEventCreate(&c->pin_ev[b]));
        if (!c->src_ev[b]) HIPCHK(c, hipEventCreate(&c->src_ev[b]));
        HIPCHK(c, hipEventSynchronize(c->pin_ev[b]));          // staging slot b free again
        const char *hsrc = (const char *)src + done * ssz;
        if (!pinned) {
            if (!c->pin[b]) HIPCHK(c, hipHostMalloc(&c->pin[b], kStageBytes, hipHostMallocDefault));
            memcpy(c->pin[b], hsrc, nv * ssz);
            hsrc = (const char *)c->pin[b];
        }
        char *dst = (char *)c->d_vol + ((size_t)z0 * slice_vox + done) * dsz;
        if (!promote) {
            HIPCHK(c, hipMemcpyAsync(dst, hsrc, nv * ssz, hipMemcpyHostToDevice, c->copy_stream));
            HIPCHK(c, hipEventRecord(c->src_ev[b], c->copy_stream));
            HIPCHK(c, hipEventRecord(c->pin_ev[b], c->copy_stream));
        } else {
            if (!c->d_stage[b]) HIPCHK(c, hipMalloc((void **)&c->d_stage[b], kStageBytes));
            HIPCHK(c, hipMemcpyAsync(c->d_stage[b], hsrc, nv, hipMemcpyHostToDevice, c->copy_stream));
            HIPCHK(c, hipEventRecord(c->src_ev[b], c->copy_stream));
            HIPCHK(c, hipStreamWaitEvent(c->promo_stream, c->src_ev[b], 0));
            launch_promote_u8_f32(c->d_stage[b], (float *)dst, nv, c->promo_stream);
            HIPCHK(c, hipGetLastError());
            HIPCHK(c, hipEventRecord(c->pin_ev[b], c->promo_stream));
        }
        // (pageable sources were copied into our own staging buffer above.)  An outstanding pinned source stays outstanding when a
        // pageable piece follows: copies are ordered on copy_stream, so this piece's event covers the earlier ones.
        if (pinned || c->src_last >= 0) c->src_last = b;
        done += nv;
    }
    return VV_OK;
}

// Returns when the copy engine has read every pinned source buffer handed to ..._slices_async so far (copies are in
// order on one stream: the last one's event covers the earlier ones).  Does not wait for promotion kernels.
int vv_load_volume_stream_wait_source(vv_context *c)
{
    if (!c) return fail(nullptr, VV_ERR_INVALID, "stream_wait_source: NULL context");
    if (c->src_last < 0) return VV_OK;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipEventSynchronize(c->src_ev[c->src_last]));
    c->src_last = -1;
    return VV_OK;
}

// A pinned source is read by the copy engine straight from the caller's buffer: do not return before that read is over, or
// a producer that refills the buffer would race it.  (The transfer of a pageable source keeps overlapping the caller's
// next fill; the promotion of a pinned piece overlaps it too.)
int vv_load_volume_stream_slices(vv_context *c, const void *src, int src_type, int z0, int nslices)
{
    int rc = vv_load_volume_stream_slices_async(c, src, src_type, z0, nslices);
    if (rc) return rc;
    return vv_load_volume_stream_wait_source(c);
}

int vv_load_volume_stream_end(vv_context *c)
{
    if (!c || !c->streaming) return fail(c, VV_ERR_INVALID, "stream_end: no stream_begin");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->copy_stream));
    HIPCHK(c, hipStreamSynchronize(c->promo_stream));
    c->streaming = false; c->src_last = -1;
    { int rc = finalize_layout(c, c->copy_stream); if (rc) return rc; }
    HIPCHK(c, hipStreamSynchronize(c->copy_stream));
    return VV_OK;
}

int vv_load_volume_t3d(vv_context *c, const char *path, int header, int vtype, const float tf[1024])
{
    if (!c || !path) return fail(c, VV_ERR_INVALID, "vv_load_volume_t3d: NULL argument");
    int nx, ny, nz;
    int rc = vv_t3d_read_header(path, header, &nx, &ny, &nz);
    if (rc) return fail(c, rc, "vv_load_volume_t3d: cannot read the header");
    FILE *f = fopen(path, "rb");                 // before the old volume is given up
    if (!f) return fail(c, VV_ERR_IO, "vv_load_volume_t3d: cannot open file");
    if (header && fseek(f, 24, SEEK_SET) != 0) { fclose(f); return fail(c, VV_ERR_IO, "vv_load_volume_t3d: seek failed"); }
    rc = vv_load_volume_stream_begin(c, vtype, nx, ny, nz, tf);
    if (rc) { fclose(f); return rc; }
    const size_t slice = (size_t)nx * ny;
    int per = (int)std::max<size_t>(1, (32u << 20) / slice);
    std::string buf;
    buf.resize(slice * (size_t)per);
    rc = VV_OK;
    for (int z = 0; z < nz && rc == VV_OK; z += per) {
        const int n = std::min(per, nz - z);
        size_t got = fread(&buf[0], 1, slice * n, f);
        if (got != slice * n) memset(&buf[got], 0, slice * n - got);      // short file: zero tail (reported below)
        int r2 = vv_load_volume_stream_slices(c, buf.data(), VV_VOXEL_U8, z, n);
        if (r2) rc = r2;
        else if (got != slice * n) rc = VV_ERR_IO;
        // the staging copy has consumed buf when the call returns (unpinned source)
    }
    fclose(f);
    int r3 = vv_load_volume_stream_end(c);
    if (rc || r3) {
        // a partly filled volume must not be rendered: drop it, later calls report VV_ERR_NO_VOLUME
        drop_bricks(c);
        if (c->d_vol) { (void)hipFree(c->d_vol); c->d_vol = nullptr; }
        c->nx = c->ny = c->nz = 0; c->vol_bytes = 0; c->streaming = false;
        if (rc == VV_ERR_IO) return fail(c, VV_ERR_IO, "vv_load_volume_t3d: file shorter than its header says");
        return rc ? rc : r3;
    }
    return VV_OK;
}

int vv_volume_dims(const vv_context *c, int dims[3], int *vtype)
{
    if (!c || !c->d_vol) return VV_ERR_NO_VOLUME;
    if (dims) { dims[0] = c->nx; dims[1] = c->ny; dims[2] = c->nz; }
    if (vtype) *vtype = c->vtype;
    return VV_OK;
}

// Pitch of the linear layout.  A row pitch that is a multiple of 1 KiB (1024^3 f32: 4 KiB) puts the four
// rows a sample gathers from (y, y+1, z, z+1) on the same cache channels: the same frame takes 9 % longer
// on a 1024^3 volume than on 1016^3 or 1032^3 (tools/ab_size.sh).  Such volumes are re-pitched on the
// device after the upload: rows get 32 bytes of padding (swept 16...2048: 32, 48 and 96 are best, 128 is
// 4 % behind) and, if a slice would still be a multiple of 4 KiB, one extra row.  Needs the old and the new buffer side by side for a moment; if that does not
// fit, the dense layout stays.  Measured on C3 (A/B on one box): 1.19 -> 1.08 ms although the padded
// 1024^3 volume is 4.4 GB and so takes the 64-bit slice addressing (+3 % by itself); nothing on the Phong
// (+3 % by itself); C3 + Phong 2.52 -> 2.46 ms; 512^3 0.71 -> 0.66 ms and C2 0.245 -> 0.224 ms on the linear path;
// u8 1024^3 0.88 -> 0.77 ms on the linear path (= its z-pair copy), u8 + Phong -1 %; other edge lengths gain 1 %
// (not re-pitched).
// VV_PITCH_PAD=0 disables, VV_PITCH_PAD=<bytes> sets the padding, VV_PITCH_FORCE=1 pads any row length.
static int finalize_layout(vv_context *c, hipStream_t st)
{
    size_t pad_bytes = 32;
    if (const char *e = getenv("VV_PITCH_PAD")) { int t = atoi(e); if (t <= 0) return VV_OK; if (t >= 16 && t <= 4096 && t % 16 == 0) pad_bytes = (size_t)t; }
    const size_t dense_row = c->row_pitch;
    if (dense_row % 1024 != 0 && !getenv("VV_PITCH_FORCE")) return VV_OK;
    if (dense_row % 16 != 0) return VV_OK;
    const size_t row = dense_row + pad_bytes;
    size_t rows = (size_t)c->ny;
    if ((rows * row) % 4096 == 0) rows += 1;
    if (const char *e = getenv("VV_PITCH_ROWS")) { int t = atoi(e); if (t >= 0 && t <= 64) rows = (size_t)c->ny + (size_t)t; }
    const size_t slice = rows * row;
    if (slice > 0xFFFFFFF0ull || row >= (1u << 24)) return VV_OK;
    const size_t bytes = slice * (size_t)c->nz, pad = slice + 2 * row + 4096;
    size_t free_b = 0, total_b = 0;
    void *nv = nullptr;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < bytes + pad + (512ull << 20) ||
        hipMalloc(&nv, bytes + pad) != hipSuccess) { (void)hipGetLastError(); return VV_OK; }
    HIPCHK(c, hipMemsetAsync(nv, 0, bytes + pad, st));           // padding must be finite (weight-0 corners)
    launch_repitch(c->d_vol, nv, dense_row, (size_t)c->ny, (size_t)c->nz, row, slice, st);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(st));
    HIPCHK(c, hipFree(c->d_vol));
    c->d_vol = nv; c->row_pitch = row; c->slice_pitch = slice; c->alloc_bytes = bytes + pad;
    return VV_OK;
}

// Builds the bricked / z-pair copy of the loaded volume if it is missing and HBM has room.
// Returns true when the copy is usable afterwards.  
static bool ensure_bricks(vv_context *c, hipStream_t st)
{
    if (c->bricks_valid) return true;
    if (c->bricks_failed) return false;                                   // (until the next volume load)
    uint32_t sy = 0, sz64 = 0;
    const size_t bb = brick_copy_bytes(c->vtype, c->nx, c->ny, c->nz, &sy, &sz64);
    size_t free_b = 0, total_b = 0;
    if (!make_room(c, bb, 1u << CP_BRICKS)) { c->bricks_failed = true; return false; }
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < bb + (512ull << 20) || sz64 >= (1u << 24) || sy >= (1u << 24) ||
        hipMalloc(&c->d_bricks, bb + 16) != hipSuccess) {                  // (the sampler multiplies b_sy and b_sz64 as 24-bit values)
        (void)hipGetLastError(); c->d_bricks = nullptr; c->bricks_failed = true;
        return false;                                                     // no room: linear path
    }
    launch_build_bricks(c->vtype, c->d_vol, c->row_pitch, c->slice_pitch, c->d_bricks, c->nx, c->ny, c->nz, st);
    if (hipMemsetAsync((char *)c->d_bricks + bb, 0, 16, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) {                         // later frames may come on another stream
        (void)hipGetLastError(); (void)hipFree(c->d_bricks); c->d_bricks = nullptr;
        return false;
    }
    c->b_sy = sy; c->b_sz64 = sz64; c->bricks_bytes = bb; c->bricks_valid = true;
    return true;
}

static bool ensure_zpair(vv_context *c, hipStream_t st)
{
    if (c->zpair_valid) return true;
    uint32_t rb = 0, sb = 0;
    const size_t zb = zpair_copy_bytes(c->vtype, c->nx, c->ny, c->nz, &rb, &sb);
    size_t free_b = 0, total_b = 0;
    if (!make_room(c, zb, 1u << CP_ZPAIR)) return false;
    if ((size_t)(c->ny + 1) * ((size_t)c->nx + 1) * 8 >= (1ull << 32) || ((size_t)c->nx + 1) * 8 >= (1u << 24) ||
        (c->vtype == VV_VOXEL_U8 && zb >= (1ull << 32)) ||                // u8 sampler: 32-bit offsets
        hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < zb + (512ull << 20) ||
        hipMalloc(&c->d_zpair, zb + 32) != hipSuccess) {
        (void)hipGetLastError(); c->d_zpair = nullptr;
        return false;
    }
    launch_build_zpair(c->vtype, c->d_vol, c->row_pitch, c->slice_pitch, c->d_zpair, c->nx, c->ny, c->nz, st);
    if (hipMemsetAsync((char *)c->d_zpair + zb, 0, 32, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
        (void)hipGetLastError(); (void)hipFree(c->d_zpair); c->d_zpair = nullptr;
        return false;
    }
    c->zp_row = rb; c->zp_slab = sb; c->zpair_bytes = zb; c->zpair_valid = true;
    return true;
}

// The z-fastest copy of the volume (VolumeView::zfast): rows of nz voxels padded like the linear layout's rows (finalize_layout),
// ny rows per slice, nx slices + one slice, one row and 16 bytes of zeros behind them (the weight-0 corners of edge samples).
static bool ensure_zfast(vv_context *c, hipStream_t st)
{
    if (c->zfast_valid) return true;
    if (c->zfast_failed) return false;
    const size_t vsz = c->vtype == VV_VOXEL_F32 ? 4 : 1;
    size_t row = (size_t)c->nz * vsz;
    if (row % 1024 == 0) row += 32;
    else if (vsz == 1) row = (row + 3) & ~(size_t)3;                       // (u8 rows are read as aligned dwords)
    size_t rows = (size_t)c->ny;
    if (row % 1024 == 32 && (rows * row) % 4096 == 0) rows += 1;
    const size_t slice = rows * row, bytes = slice * ((size_t)c->nx + 1) + row + 16;
    size_t free_b = 0, total_b = 0;
    if (c->ny > 65535 || (c->nz + 31) / 32 > 65535 || !make_room(c, bytes, 1u << CP_ZFAST)) { c->zfast_failed = true; return false; }     // (launch_build_zfast: one grid layer per row)
    if (row >= (1u << 24) || slice >= (1ull << 32) || hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < bytes + (512ull << 20) ||
        hipMalloc(&c->d_zfast, bytes) != hipSuccess) {
        (void)hipGetLastError(); c->d_zfast = nullptr; c->zfast_failed = true;
        return false;                                                     // no room: the bricked copy serves the view
    }
    if (hipMemsetAsync(c->d_zfast, 0, bytes, st) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(c->d_zfast); c->d_zfast = nullptr; c->zfast_failed = true; return false; }
    launch_build_zfast(c->vtype, c->d_vol, (uint32_t)c->row_pitch, (uint64_t)c->slice_pitch, c->d_zfast, (uint32_t)row, (uint64_t)slice, c->nx, c->ny, c->nz, st);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {      // later frames may come on another stream
        (void)hipGetLastError(); (void)hipFree(c->d_zfast); c->d_zfast = nullptr; c->zfast_failed = true;
        return false;
    }
    c->zf_row = (uint32_t)row; c->zf_slice = (uint64_t)slice; c->zfast_bytes = bytes; c->zfast_valid = true;
    return true;
}

// The x-pair copy (the z-pair copy with x and z exchanged; built from the z-fastest copy): the limits of ensure_zpair with the roles swapped.
static bool ensure_xpair(vv_context *c, hipStream_t st)
{
    if (c->xpair_valid) return true;
    if (c->xpair_failed || !ensure_zfast(c, st)) return false;
    uint32_t rb = 0, sb = 0;
    const size_t xb = zpair_copy_bytes(c->vtype, c->nz, c->ny, c->nx, &rb, &sb);
    size_t free_b = 0, total_b = 0;
    if (!make_room(c, xb, (1u << CP_XPAIR) | (1u << CP_ZFAST))) { c->xpair_failed = true; return false; }
    if ((size_t)(c->ny + 1) * ((size_t)c->nz + 1) * 8 >= (1ull << 32) || ((size_t)c->nz + 1) * 8 >= (1u << 24) ||
        (c->vtype == VV_VOXEL_U8 && xb >= (1ull << 32)) ||                // u8 sampler: 32-bit offsets
        hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < xb + (512ull << 20) ||
        hipMalloc(&c->d_xpair, xb + 32) != hipSuccess) {
        (void)hipGetLastError(); c->d_xpair = nullptr; c->xpair_failed = true;
        return false;
    }
    launch_build_xpair(c->vtype, c->d_zfast, c->zf_row, c->zf_slice, c->d_xpair, c->nx, c->ny, c->nz, st);
    if (hipGetLastError() != hipSuccess ||
        hipMemsetAsync((char *)c->d_xpair + xb, 0, 32, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
        (void)hipGetLastError(); (void)hipFree(c->d_xpair); c->d_xpair = nullptr; c->xpair_failed = true;
        return false;
    }
    c->xp_row = rb; c->xp_slab = sb; c->xpair_bytes = xb; c->xpair_valid = true;
    return true;
}

static VolumeView view_of(const vv_context *c)
{
    VolumeView V;
    const uint32_t vsz = c->vtype == VV_VOXEL_F32 ? 4 : 1;
    V.data = c->d_vol; V.nx = c->nx; V.ny = c->ny; V.nz = c->nz;
    (void)vsz;
    V.row_bytes = (uint32_t)c->row_pitch;
    V.slice_bytes = (uint32_t)c->slice_pitch;
    V.big = c->slice_pitch * (size_t)c->nz > (1ull << 32) || V.slice_bytes >= (1u << 24) || c->knobs.force_big;
    V.bricks = nullptr; V.b_sy = 0; V.b_sz64 = 0;
    V.zpair = nullptr; V.zp_row_bytes = 0; V.zp_slab_bytes = 0;
    V.zfast = nullptr; V.zf_row_bytes = 0; V.zf_slice_bytes = 0;
    return V;
}

// ---- ray march: runCuda, kernel.cu:388-453 ----------------------------------------
static inline float vlen_h(float x, float y, float z) { return sqrtf(x * x + y * y + z * z); }

// camera basis of the analytic ray source (shared by vv_render and vv_first_pass)
static int camera_basis(vv_context *c, FrameParams &P, const camera_params *cam, const vv_ray_source *rays, int W, int H)
{
    // camera.cpp:78-91 look-at basis, float
    float lx = rays->look[0], ly = rays->look[1], lz = rays->look[2];
    float ll = vlen_h(lx, ly, lz);
    if (!(ll > 0.f)) return fail(c, VV_ERR_INVALID, "vv_render: look vector is zero");
    lx /= ll; ly /= ll; lz /= ll;
    float ux = rays->up[0], uy = rays->up[1], uz = rays->up[2];
    float sx = ly * uz - lz * uy, sy = lz * ux - lx * uz, sz = lx * uy - ly * ux;
    float sl = vlen_h(sx, sy, sz);
    if (!(sl > 0.f)) return fail(c, VV_ERR_INVALID, "vv_render: up vector is parallel to look");
    sx /= sl; sy /= sl; sz /= sl;
    float vx = sy * lz - sz * ly, vy = sz * lx - sx * lz, vz = sx * ly - sy * lx;
    float vl = vlen_h(vx, vy, vz);
    vx /= vl; vy /= vl; vz /= vl;
    P.look[0] = lx; P.look[1] = ly; P.look[2] = lz;
    P.side[0] = sx; P.side[1] = sy; P.side[2] = sz;
    P.up[0] = vx; P.up[1] = vy; P.up[2] = vz;
    float aspect = rays->aspect > 0.f ? rays->aspect : (float)W / (float)H;
    float th = (float)tan((double)cam->fovY * M_PI / 360.0);
    P.tan_half_x = th * aspect; P.tan_half_y = th;
    return VV_OK;
}


} // extern "C"

// ---- launch policy of vv_render (speed only: the pixels never depend on anything decided here) -----------------------------------------------
// From what is known about the view (`have_basis`: P.side holds the screen x direction in volume space; `density`: voxels of volume per sample, 1e9 =
// unknown) it picks the layout the frame samples (building a missing copy when the context allows it), the wave tile and block shape, the samples in
// flight per lane and the blocks per CU, and fills MarchArgs accordingly.  Every threshold below carries the measurement it comes from; the VV_* knobs
// override single decisions for A/Bs.  Returns whether the volume is beyond the caches (the callers pick the kernel build by it).
static bool choose_launch(vv_context *c, MarchArgs &A, const camera_params *cam, const vv_ray_source *rays, const shading_params *shading,
                          bool have_basis, float density, int H, hipStream_t st)
{
    FrameParams &P = A.P;
    A.V = view_of(c); A.V_type = c->vtype;
    // Wave tile shape (speed only): memory is contiguous along the volume's x axis.  When the
    // screen x direction maps (almost) onto it, a 32x2 tile lets the 32 lanes of a row read one
    // or two cache lines (measured C3, view along z: 1.6 ms vs 2.1 ms for 8x8); otherwise the
    // compact 8x8 tile touches the fewest lines.  VV_TILE_LOG2W overrides (3, 4 or 5).
    A.strips.tile_log2w = 3;
    if (have_basis) {
        const float ax = fabsf(P.side[0]) , ay = fabsf(P.side[1]), az = fabsf(P.side[2]);
        if (ax >= 0.97f * sqrtf(ax * ax + ay * ay + az * az)) A.strips.tile_log2w = 5;
    }
    const vv_knobs &K = c->knobs;
    // Sampling density: voxels of volume per sample = (voxels per step) x (voxels per pixel)^2 at the
    // cube centre.  Sparse rays (C3 at 1080p: 4.9) reuse little of a cache line and want few waves on a CU's
    // L1; the denser frames of the multi-GPU configurations (2716x1528: 2.5, 3840x2160: 1.2, and 0.6 at
    // step 1/1024) want more: the whole 3840x2160 / 1024-step frame takes 3.51 ms with 2 blocks per CU and
    // 2.71 ms with 4 (tools/sweep.sh --frame-of 8; bricked copy 4.71 -> 3.34 ms, Phong 9.0 -> 7.9 ms).
    if ((rays->mode == VV_RAYS_ANALYTIC || image_source_has_hint(rays)) && have_basis && H > 0) {
        const float dist = vlen_h(cam->origin[0], cam->origin[1], cam->origin[2]);
        const float vpw = cbrtf(((float)c->nx / (2.f * P.scale[0])) * ((float)c->ny / (2.f * P.scale[1])) * ((float)c->nz / (2.f * P.scale[2])));
        const float px_vox = 2.f * P.tan_half_y * dist / (float)H * vpw;
        const float step_vox = fmaxf(P.step[0] * c->nx, fmaxf(P.step[1] * c->ny, P.step[2] * c->nz));
        if (std::isfinite(px_vox) && std::isfinite(step_vox) && px_vox > 0.f && step_vox > 0.f) density = step_vox * px_vox * px_vox;
    }
    // z-fastest copy (speed only): when the screen x direction maps onto the volume's z axis (side views) the same 32 x 2 tile reads whole
    // lines of a copy whose rows run along z -- the front view's kernel instead of the bricked copy's (C3 1.33 -> 1.0 ms, C2 0.223 -> 0.186,
    // profiles/r04_side_view.txt).  Both voxel types, both kernels, every volume the bricked copy would serve; built on first use if HBM has
    // room (one more copy of the volume).  VV_ZFAST=0/1 overrides.
    bool use_zfast = false;
    if (have_basis && A.strips.tile_log2w == 3) {
        const float ax = fabsf(P.side[0]) , ay = fabsf(P.side[1]), az = fabsf(P.side[2]);
        use_zfast = az >= 0.97f * sqrtf(ax * ax + ay * ay + az * az) && (size_t)c->nx * c->ny * c->nz >= (1ull << 21);
        if (K.zfast >= 0) use_zfast = K.zfast != 0;
    }
    // Phong frames of f32 volumes beyond the caches sample the bricked copy from every direction: march_phong_kernel's blocks are 16 pixels wide whatever the
    // view, and 16 pixels of a row use 0.8 of the 1.8 lines they touch in a linear layout, while a 4 x 4 x 4 brick is used whole.  Measured:
    // C3 + Phong along z 1.750 -> 1.650 ms (-5.5 %), from the side (z-fastest copy) 1.758 -> 1.666; volumes in the caches lose 20 % (VALU-bound): not for those.
    // The 32 GiB volume of C5 (a pixel every 3.2 voxels: a brick serves 1.5 pixels a side) loses 9 %, the dense frame of the 8-GPU configuration 15 %, 768^3
    // (1.8 GiB, 2.2 voxels per sample) 12 %: for sparse frames (3.5 ... 8 voxels per sample) of volumes between 2 and 8 GiB.  VV_PHONG_BRICKS=0/1 overrides.
    const bool phong_bricks_fit = K.phong_bricks > 0 || (density > 3.5f && density <= 8.f && c->vol_bytes > (2ull << 30) && c->vol_bytes <= (8ull << 30));
    const bool phong_bricks = phong_bricks_fit && shading->phongShading && c->vol_bytes > (1ull << 30) && c->vtype == VV_VOXEL_F32 && K.bricked != 0 && K.phong_bricks != 0 &&
                              (c->bricks_valid || (c->build_in_render && !c->bricks_failed));
    if (phong_bricks && K.zfast < 0) use_zfast = false;
    ++c->frame_no;
    const unsigned long long builds_before = (unsigned long long)c->bricks_valid + c->zpair_valid + c->zfast_valid + c->xpair_valid;
    if (use_zfast) use_zfast = c->build_in_render ? ensure_zfast(c, st) : c->zfast_valid;
    if (use_zfast) {
        c->last_used[CP_ZFAST] = c->frame_no;
        A.strips.tile_log2w = 5;
        A.V.zfast = c->d_zfast; A.V.zf_row_bytes = c->zf_row; A.V.zf_slice_bytes = c->zf_slice;
    }
    if (K.tile_log2w >= 3 && K.tile_log2w <= 5 && !use_zfast) A.strips.tile_log2w = K.tile_log2w;
    // Occupancy cap + gathers in flight (speed only; measured on MI355X, profiles/EXPERIMENTS.md part B section 4):
    //   volume beyond the caches (> 1 GiB), aligned view : 2 blocks per CU, 3 samples per trip
    //   volume beyond the caches, rotated, linear layout : 1 block  per CU, 3 samples per trip
    //   smaller volumes                                  : 4 blocks per CU, 3 (aligned) / 2 samples per trip
    //   bricked copy (below)                             : 2 / 4 blocks per CU, 2 samples per trip
    // LDS per block = 4 KB table + reserve; 160 KB per CU.  VV_LDS_RESERVE / VV_UNROLL override.
    // XCD-aware block order: XCD k renders strips k, k+8, ... (each a full-width row of tiles), so
    // the tiles that share volume cache lines share an L2 (measured: -4 % on every workload)
    A.strips.xcd_band = 1;
    if (K.xcd_band >= 0 && K.xcd_band <= 64) A.strips.xcd_band = K.xcd_band;
    const bool beyond_caches = c->vol_bytes > (1ull << 30);
    const int big_reserve = density > 3.5f ? 76000 : (density > 1.8f ? 49000 : 36000);   // 2 / 3 / 4 blocks per CU
    // 3 samples per trip along the memory axis (A/B with repeats on MI355X: C3 -1.6 %, C2 -5.7 %, 512^3 -10 %,
    // u8 1024^3 -6.5 %; 4 per trip is no better), 2 on the bricked copy (3 there: +3.5 %)
    A.unroll = (A.strips.tile_log2w == 5 || beyond_caches) ? 3 : 2;
    // (re-swept with tools/ab_reserve.sh at the end of round 1: 4 blocks per CU for volumes up to 1 GiB)
    A.lds_reserve = !beyond_caches ? 36000 : (A.strips.tile_log2w == 5 ? big_reserve : 155000);
    // Block shape (speed only).  32 x 2 wave tiles: stacked (32 x 8 pixels), 2 x 2 (64 x 4) or side by side (128 x 2): the partial lines two x-adjacent wave
    // tiles share are then fetched within one block; strips get lower.  8 x 8 wave tiles: side by side (32 x 8), 2 x 2 (16 x 16) or stacked (8 x 32).  VV_BLOCK_W=8...128.
    //   measured (tools/ab_env.sh, profiles/r03_block_shape.txt): 64 x 4: C3 -2.4 % (EA bytes 1.474 -> 1.364 x algorithmic), u8 1024^3 -1.3 %, tilted views -1.7 %, but
    //   +4.5 % on C2, +1 % on C1 / 512^3 and +0.5 % on the dense frames of the multi-GPU configurations: used for sparse frames of volumes beyond the caches.
    //   16 x 16: a strip 16 pixels high re-reads fewer brick layers of its neighbours (which run on other XCDs): rotated C3 -2 % (EA bytes 2.00 -> 1.81 x), C2 -5.5 %,
    //   C1 -6.5 %, other orbits -2 ... -4 %, 512^3 and the 3840 x 2160 frame -0.5 %: used for every frame with 8 x 8 tiles.
    A.strips.blk_log2w = 5;
    A.strips.tail_batch = K.tail == 0 ? 0 : 1;
    const int rows_px_8 = A.strips.n_strips * 8;              // (the shard's pixel rows as strips of 8)
    int block_w = A.strips.tile_log2w == 3 ? 16 : ((A.strips.tile_log2w == 5 && c->vol_bytes >= (1ull << 30) && density > 3.5f) ? 64 : 32);      // (>=: u8 1024^3 -2 %, tools/policy_sweep.sh)
    if (K.block_w >= 8 && K.block_w <= 128 && (K.block_w & (K.block_w - 1)) == 0) block_w = K.block_w;
    {
        int lw = 3; while ((1 << lw) < block_w) ++lw;
        const int h = 256 >> lw;                               // strip height in pixels
        if (lw != 5 && lw >= A.strips.tile_log2w && h >= (64 >> A.strips.tile_log2w)) {     // (a block is at least one wave tile wide and high)
            A.strips.blk_log2w = lw;
            // rows past the end of a band or of the range belong to nobody or to another rank: not owned (row_owned), their lanes stay idle
            if (A.strips.strips_per_band < (1 << 26)) {
                const int band_px = A.strips.strips_per_band * 8, spb = (band_px + h - 1) / h;
                A.strips.n_strips = (rows_px_8 / band_px) * spb; A.strips.strips_per_band = spb;
            } else A.strips.n_strips = (rows_px_8 + h - 1) / h;
        }
    }
    const bool k_unroll = K.unroll >= 1 && K.unroll <= 3, k_reserve = K.lds_reserve >= 0 && K.lds_reserve <= 155 * 1024;
    if (k_unroll) A.unroll = K.unroll;
    if (k_reserve) A.lds_reserve = K.lds_reserve;
    // Bricked copy (speed only): off the memory axis the linear layout costs one cache line per lane
    // and gather; 4x4x4 bricks keep a wave's footprint in a few dozen lines.  both voxel types, both kernels;
    // built on first use if HBM has room (1.25x an f32 volume, 2x a u8 volume).  VV_BRICKED=0/1 overrides the policy.
    // Measured: 1024^3 rotated 3.85 -> 1.65 ms (f32), 3.17 -> 0.99 ms (u8); C2 (256^3) -19 %, C1 (128^3) -9 %;
    // only volumes far below the frame's sampling density lose (64^3 at 1080p, step 1/512: +10 %).
    bool use_bricks = (A.strips.tile_log2w == 3 || phong_bricks) && (size_t)c->nx * c->ny * c->nz >= (1ull << 21);
    if (K.bricked >= 0) use_bricks = K.bricked != 0;
    if (use_zfast) use_bricks = false;
    if (use_bricks) use_bricks = c->build_in_render ? ensure_bricks(c, st) : c->bricks_valid;
    if (use_bricks) {
        c->last_used[CP_BRICKS] = c->frame_no;
        A.V.bricks = c->d_bricks; A.V.b_sy = c->b_sy; A.V.b_sz64 = c->b_sz64;
        // measured (C3 rotated, 1024^3): 2 blocks per CU and 2 samples per trip: 3.64 -> 1.60 ms
        if (!k_unroll) A.unroll = 2;
        if (!k_reserve) A.lds_reserve = beyond_caches ? big_reserve : 36000;
    }
    // z-pair copy (speed only): along the memory axis the four corners (x..x+1, z..z+1) of a row come
    // from one gather (16 bytes for f32, 4 for u8), so a sample costs 2 gathers instead of 4 (f32) or 8
    // aligned dwords (u8).  2x the volume in HBM.  Measured on MI355X, unshaded frames:
    //   f32: -7 % on C1/C2, -10 % on 256^3 at C3's frame, -3 % on 512^3, but +6 % on 768^3 and +20 % on
    //        1024^3 (rays of different z phase stop sharing slices in L2): used up to 512 MiB;
    //   u8 : -8 % (256^3), -12 % (512^3), -4 % (1024^3): used whenever the copy stays below 4 GiB;
    //   Phong path: 0...-3 %: not used.                                  VV_ZPAIR=0/1 overrides.
    bool use_zpair = !use_bricks && A.strips.tile_log2w == 5 && !shading->phongShading &&
                     (c->vtype == VV_VOXEL_U8 || c->vol_bytes <= (512ull << 20));
    if (K.zpair >= 0) use_zpair = K.zpair != 0 && !use_bricks;
    if (use_zfast) use_zpair = false;
    if (use_zpair) use_zpair = c->build_in_render ? ensure_zpair(c, st) : c->zpair_valid;
    if (use_zpair) { c->last_used[CP_ZPAIR] = c->frame_no; A.V.zpair = c->d_zpair; A.V.zp_row_bytes = c->zp_row; A.V.zp_slab_bytes = c->zp_slab; }
    // x-pair copy (speed only): the same two-gather form for side views -- the z-pair copy with x and z exchanged, handed to the kernel in the
    // z-pair fields of the view -- under the z-pair copy's conditions (unshaded, u8 or f32 up to 512 MiB).  Follows VV_ZPAIR=0.
    A.xpair = false;
    if (use_zfast && !shading->phongShading && K.zpair != 0 && (c->vtype == VV_VOXEL_U8 || c->vol_bytes <= (512ull << 20)) &&
        (c->build_in_render ? ensure_xpair(c, st) : c->xpair_valid)) {
        A.xpair = true; c->last_used[CP_XPAIR] = c->frame_no;
        A.V.zpair = c->d_xpair; A.V.zp_row_bytes = c->xp_row; A.V.zp_slab_bytes = c->xp_slab;
    }
    c->builds_in_render += ((unsigned long long)c->bricks_valid + c->zpair_valid + c->zfast_valid + c->xpair_valid > builds_before)
                           ? ((unsigned long long)c->bricks_valid + c->zpair_valid + c->zfast_valid + c->xpair_valid - builds_before) : 0;
    // Phong kernel: 14.3 KB of LDS per block + this reserve.  Measured (tools/ab_phong.sh): volumes up to
    // 1 GiB like 5 blocks per CU (C2 0.54 -> 0.47 ms against no cap, u8 1024^3 1.88 -> 1.78), the 4 GiB
    // volume of C3 3 blocks (2.77 ms with 2, 2.48 with 3, 2.54 with 4), the 32 GiB volume of C5 2 (18.2 vs 20.2 ms)
    // (re-swept with the depth-limited refresh, tools/ab_phong_reserve.sh: C3 1.97 ms with 3, 4 or 5 blocks, 2.22 with 2; the
    // bricked copy likes 5: rotated C3 + Phong 2.25 -> 2.13 ms; C5 5.95 ms with 2, 6.25 with 3, 6.55 with 5)
    A.lds_reserve_phong = c->vol_bytes > (8ull << 30) ? 40000 : ((beyond_caches && density > 3.5f && !use_bricks) ? 30000 : 13000);
    if (K.lds_reserve_phong >= 0 && K.lds_reserve_phong <= 146 * 1024) A.lds_reserve_phong = K.lds_reserve_phong;
    c->last_density = density;
    return beyond_caches;
}

// The volume's screen rectangle (speed only in intent, but the frame relies on it: see below).  Analytic rays only.  Of C3's 8100 tiles 5150 lie beside the
// cube: blocks that set up 256 rays, find none alive and leave -- 2 us each, plus the ~4 us a slot stands empty between two blocks (per-block time stamps,
// tools/timeline.py): together 6 % of the frame's slot time.  march_kernel is therefore launched over the tiles under the cube's bounding rectangle only, and
// rad_kernel, which visits every slab anyway, writes the 0 of the pixels outside it and skips the radii nobody reads.
// The rectangle must contain every pixel whose ray meets the cube.  A ray that meets the cube passes through a point of it, so its pixel lies in the convex hull
// of the eight projected corners -- in exact arithmetic.  ray_endpoints decides in binary32: its direction carries an error of a few ulps (|d| ~ 1) and its slab
// test errors of a few ulps of (|o| + s) / |d_a|; a ray can be taken for a hit only if it passes the cube within ~1e-6 (|o| + s).  Seen from the eye that is an
// angle of at most 1e-6 (|o| + s) / depth_min, which the margin below covers a hundred times over, expressed in pixels (+ 1); a corner at or behind the eye's plane
// (depth <= 1e-3 (|o| + s)), a margin wider than the frame or anything non-finite: no rectangle, the whole frame is launched as before.
static bool screen_rect(const MarchArgs &A, int W, int H, double *xmin, double *xmax, double *ymin, double *ymax)
{
    const FrameParams &P = A.P;
    double lo[2] = {INFINITY, INFINITY}, hi[2] = {-INFINITY, -INFINITY}, dmin = INFINITY, ext = 0.0;
    for (int a = 0; a < 3; ++a) ext = std::max(ext, (double)fabsf(P.cam_pos[a]) + (double)P.scale[a]);
    for (int k = 0; k < 8; ++k) {
        const double v[3] = {((k & 1) ? P.scale[0] : -P.scale[0]) - (double)P.cam_pos[0], ((k & 2) ? P.scale[1] : -P.scale[1]) - (double)P.cam_pos[1],
                             ((k & 4) ? P.scale[2] : -P.scale[2]) - (double)P.cam_pos[2]};
        const double depth = v[0] * P.look[0] + v[1] * P.look[1] + v[2] * P.look[2];
        if (!(depth > 1e-3 * ext)) return false;
        dmin = std::min(dmin, depth);
        const double sx = (v[0] * P.side[0] + v[1] * P.side[1] + v[2] * P.side[2]) / depth, sy = (v[0] * P.up[0] + v[1] * P.up[1] + v[2] * P.up[2]) / depth;
        const double px = (sx / P.tan_half_x + 1.0) * 0.5 * W - 0.5, py = (sy / P.tan_half_y + 1.0) * 0.5 * H - 0.5;     // (ray_endpoints' pixel centres)
        lo[0] = std::min(lo[0], px); hi[0] = std::max(hi[0], px); lo[1] = std::min(lo[1], py); hi[1] = std::max(hi[1], py);
    }
    const double ang = 1e-4 * ext / dmin;                                                       // (100 x the bound above)
    const double mx = 1.0 + ang / (2.0 * P.tan_half_x / W), my = 1.0 + ang / (2.0 * P.tan_half_y / H);
    if (!std::isfinite(lo[0] + hi[0] + lo[1] + hi[1] + mx + my) || mx > W || my > H) return false;
    *xmin = lo[0] - mx; *xmax = hi[0] + mx; *ymin = lo[1] - my; *ymax = hi[1] + my;
    return true;
}

extern "C" {

int vv_render(vv_context *c, int W, int H, const slice_params *slice, const camera_params *cam,
              const shading_params *shading, const vv_ray_source *rays, const vv_render_options *opts,
              uint8_t *rgba_out, int out_on_device, void *stream)
{
    if (!c) return fail(nullptr, VV_ERR_INVALID, "vv_render: NULL context");
    if (!slice || !cam || !shading || !rays || !rgba_out) return fail(c, VV_ERR_INVALID, "vv_render: NULL argument");
    if (W < 1 || H < 1) return fail(c, VV_ERR_INVALID, "vv_render: width/height must be >= 1");
    if (!c->d_vol || !c->have_tf) return fail(c, VV_ERR_NO_VOLUME, "vv_render: no volume / transfer function loaded");
    if (slice->type != SLICE_NONE && slice->type != SLICE_PLANE && slice->type != SLICE_PLANE_CUT)
        return fail(c, VV_ERR_INVALID, "vv_render: slice type must be SLICE_NONE/PLANE/PLANE_CUT");
    if (rays->mode == VV_RAYS_IMAGES && (!rays->front || !rays->back || rays->img_w < 1 || rays->img_h < 1))
        return fail(c, VV_ERR_INVALID, "vv_render: image ray source needs front/back images");
    if (rays->mode != VV_RAYS_IMAGES && rays->mode != VV_RAYS_ANALYTIC)
        return fail(c, VV_ERR_INVALID, "vv_render: bad ray source mode");
    for (int a = 0; a < 3; ++a)
        if (!(cam->scale[a] > 0.f) || !std::isfinite(cam->scale[a]))
            return fail(c, VV_ERR_INVALID, "vv_render: camera scale must be finite and > 0");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = pick_stream(c, stream);

    MarchArgs A;
    memset(&A, 0, sizeof A);
    FrameParams &P = A.P;
    P.W = W; P.H = H;
    P.nbx = W / kSlab + ((W % kSlab) ? 1 : 0);                         // kernel.cu:418-425
    P.nby = H / kSlab + ((H % kSlab) ? 1 : 0);
    P.conflict_x = (W >= 2 && W - 1 == (P.nbx - 1) * kSlab);
    P.conflict_y = (H >= 2 && H - 1 == (P.nby - 1) * kSlab);
    int rb = 0, re = P.nby;
    int s_count = 1, s_index = 0, s_band = 4;
    P.step[0] = 1.f / (float)c->nx; P.step[1] = 1.f / (float)c->ny; P.step[2] = 1.f / (float)c->nz;   // :415
    P.ert_thr = .95f; P.ert_true = 0; P.alpha_unit = c->tf_alpha_unit;
    A.tex8 = true; A.instr = false;
    if (opts) {
        if (opts->step[0] > 0.f || opts->step[1] > 0.f || opts->step[2] > 0.f) {
            P.step[0] = opts->step[0]; P.step[1] = opts->step[1]; P.step[2] = opts->step[2];
        }
        if (opts->ert_threshold > 0.f) P.ert_thr = opts->ert_threshold;
        P.ert_true = opts->ert_mode == VV_ERT_TRUE;
        A.tex8 = opts->filter != VV_FILTER_EXACT;
        if (!(opts->slab_row_begin == 0 && opts->slab_row_end == 0)) { rb = opts->slab_row_begin; re = opts->slab_row_end; }
        if (opts->shard_count > 1) { s_count = opts->shard_count; s_index = opts->shard_index; s_band = opts->shard_band; }
        A.instr = opts->count_samples != 0 || opts->touched_bricks != nullptr || opts->touched_lines != nullptr || opts->touched_block_lines != nullptr;
        if (opts->touched_block_lines && (opts->touched_block_lines_log2 < 10 || opts->touched_block_lines_log2 > 34))
            return fail(c, VV_ERR_INVALID, "vv_render: touched_block_lines_log2 must lie in [10, 34]");
        A.I.pairs = opts->touched_block_lines; A.I.pairs_log2 = (uint32_t)opts->touched_block_lines_log2;
        A.I.bricks = opts->touched_bricks;
        A.I.lines = opts->touched_lines; A.I.line_bits = opts->touched_lines ? opts->touched_line_bits : 0; A.I.lines_all = opts->touched_lines_all;
    }
    float min_step = INFINITY;
    for (int a = 0; a < 3; ++a) {
        if (!(P.step[a] >= 1e-5f) || !std::isfinite(P.step[a]))
            return fail(c, VV_ERR_INVALID, "vv_render: step must be finite and >= 1e-5 per axis");
        min_step = fminf(min_step, P.step[a]);
    }
    if (rb < 0 || re > P.nby || rb > re) return fail(c, VV_ERR_INVALID, "vv_render: slab row range out of bounds");
    // chunks needed for the longest possible ray (upper <= sqrt 3, kernel.cu:350) + slack
    P.max_chunks = (int)(kSqrt3 / (min_step * kChunkSteps)) + 4;
    if (s_count > 1 && (s_band < 4 || s_band % 4 != 0 || s_index < 0 || s_index >= s_count))
        return fail(c, VV_ERR_INVALID, "vv_render: shard_band must be a positive multiple of 4, 0 <= shard_index < shard_count");
    P.rb = rb; P.re = re; P.band = s_band; P.count = s_count; P.index = s_index;
    // grid maps (see vv_kernels.h): which strips / slab rows this call launches
    const int last_written = H >= 2 ? H - 2 : 0;
    const int geo_rows = last_written / kSlab + 1;            // slab rows that own pixel rows geometrically
    if (s_count > 1) {
        const int nbands = (geo_rows + s_band - 1) / s_band;
        const int own = s_index < nbands ? (nbands - s_index + s_count - 1) / s_count : 0;
        A.strips.y0 = s_index * s_band * kSlab; A.strips.strips_per_band = s_band * kSlab / 8;
        A.strips.band_stride_px = s_count * s_band * kSlab;
        A.strips.n_strips = own * A.strips.strips_per_band;
        A.slabs.r0 = s_index * s_band; A.slabs.band = s_band; A.slabs.band_stride = s_count * s_band;
        A.slabs.n_regular = own * s_band;
    } else {
        const int r_lo = rb, r_hi = re < geo_rows ? re : geo_rows;
        const int y_lo = r_lo * kSlab, y_hi = (r_hi * kSlab < last_written + 1) ? r_hi * kSlab : last_written + 1;
        A.strips.y0 = y_lo; A.strips.strips_per_band = 1 << 28; A.strips.band_stride_px = 0;
        A.strips.n_strips = y_hi > y_lo ? (y_hi - y_lo + 7) / 8 : 0;
        A.slabs.r0 = r_lo; A.slabs.band = 1 << 28; A.slabs.band_stride = 0;
        A.slabs.n_regular = r_hi > r_lo ? r_hi - r_lo : 0;
    }
    P.slice_type = slice->type;
    for (int a = 0; a < 3; ++a) {
        P.slice_point[a] = slice->params[a]; P.slice_normal[a] = slice->params[3 + a];   // kernel.cu:224-225
        P.cam_pos[a] = cam->origin[a]; P.scale[a] = cam->scale[a];
        P.inv_scale[a] = 1.0f / cam->scale[a];
    }
    // kernel.cu:221-222: float * double / float -> double; tan in double; narrowed
    P.tan_fov_x = (float)tan((double)cam->fovX * M_PI / (double)(180.f * (float)(unsigned)W));
    P.tan_fov_y = (float)tan((double)cam->fovY * M_PI / (double)(180.f * (float)(unsigned)H));
    {
        const float lo = 0x1p-24f, hi = 0x1p8f;
        P.safe_div = P.tan_fov_x >= lo && P.tan_fov_x <= hi && P.tan_fov_y >= lo && P.tan_fov_y <= hi &&
                     P.step[0] <= 16.f && P.step[1] <= 16.f && P.step[2] <= 16.f;          // (>= 1e-5 and finite: checked above)
    }
    P.ray_mode = rays->mode; P.quantize8 = rays->quantize8;
    // What the launch policy below knows about the view (speed only): the camera basis -- given (analytic rays), hinted (an image
    // source whose look / up fields are filled in: the host that drew the first pass knows its camera) or estimated from the images.
    bool have_basis = false;
    float density = 1e9f;                           // voxels of volume per sample; unknown ray source: treat as sparse
    if (rays->mode == VV_RAYS_ANALYTIC) {
        int brc = camera_basis(c, P, cam, rays, W, H);
        if (brc) return brc;
        have_basis = true;
    } else {
        if (image_source_has_hint(rays)) have_basis = camera_basis(nullptr, P, cam, rays, W, H) == VV_OK;
        const size_t ib = (size_t)rays->img_w * rays->img_h * 4;
        P.img_w = rays->img_w; P.img_h = rays->img_h;
        if (rays->images_on_device) {
            if (((uintptr_t)rays->front & 3) || ((uintptr_t)rays->back & 3)) return fail(c, VV_ERR_INVALID, "vv_render: device images must be 4-byte aligned");
            P.front_img = rays->front; P.back_img = rays->back;
        }
        else {
            int rc = ensure(c, (void **)&c->d_img, &c->img_cap, 2 * ib);
            if (rc) return rc;
            HIPCHK(c, hipMemcpyAsync(c->d_img, rays->front, ib, hipMemcpyHostToDevice, st));
            HIPCHK(c, hipMemcpyAsync(c->d_img + ib, rays->back, ib, hipMemcpyHostToDevice, st));
            P.front_img = c->d_img; P.back_img = c->d_img + ib;
        }
        // no hint: the images themselves say how screen x runs through the volume and how dense the rays are.  Host images are
        // read in place; device images only by a synchronous call (an enqueue-only call must not wait for the stream), and the
        // estimate is kept while camera and image geometry stay the same.
        if (!have_basis) {
            bool est = false;
            float side[3] = {0, 0, 0}, px_cube = 0.f;
            if (!rays->images_on_device) est = estimate_view_from_images(rays->front, rays->back, rays->img_w, rays->img_h, W, H, side, &px_cube);
            else if (!stream) {
                const vv_view_key key = view_key_of(cam, rays, W, H);
                if (c->view_key_valid && memcmp(&c->view_key, &key, sizeof key) == 0) { est = c->view_est_ok; memcpy(side, c->view_side, sizeof side); px_cube = c->view_px_cube; }
                else {
                    const int yt = clamp_row((H / 2) * (long long)rays->img_h / H, rays->img_h);
                    const size_t rowb = (size_t)rays->img_w * 4;
                    c->row_buf.resize(2 * rowb);
                    HIPCHK(c, hipStreamSynchronize(st));
                    HIPCHK(c, hipMemcpy(c->row_buf.data(), rays->front + (size_t)yt * rowb, rowb, hipMemcpyDeviceToHost));
                    HIPCHK(c, hipMemcpy(c->row_buf.data() + rowb, rays->back + (size_t)yt * rowb, rowb, hipMemcpyDeviceToHost));
                    // (the centre row alone, presented as a one-row image pair)
                    est = estimate_view_from_images(c->row_buf.data(), c->row_buf.data() + rowb, rays->img_w, 1, W, 1, side, &px_cube);
                    c->view_key = key; c->view_key_valid = true; c->view_est_ok = est; memcpy(c->view_side, side, sizeof side); c->view_px_cube = px_cube;
                }
            }
            if (est) {
                P.side[0] = side[0]; P.side[1] = side[1]; P.side[2] = side[2];
                have_basis = true;
                // px_cube = cube-space distance between horizontally neighbouring pixels' rays at mid depth
                const float vpw = cbrtf((float)c->nx * (float)c->ny * (float)c->nz);
                const float px_vox = px_cube * vpw;
                const float step_vox = fmaxf(P.step[0] * c->nx, fmaxf(P.step[1] * c->ny, P.step[2] * c->nz));
                if (std::isfinite(px_vox) && px_vox > 0.f && step_vox > 0.f) density = step_vox * px_vox * px_vox;
            }
        }
    }
    const bool beyond_caches = choose_launch(c, A, cam, rays, shading, have_basis, density, H, st);
    density = c->last_density;
    A.gray = c->tf_gray; A.phong = shading->phongShading;
    // march_kernel's tiles: those under the volume's screen rectangle (screen_rect), or all of them
    {
        StripMap &M = A.strips;
        const int bw = 1 << M.blk_log2w, bh = 256 >> M.blk_log2w;
        M.tx0 = 0; M.wr = (W + bw - 1) >> M.blk_log2w; M.s0 = 0; M.s1 = M.n_strips;
        A.rect.x0 = 0; A.rect.y0 = 0; A.rect.x1 = INT_MAX; A.rect.y1 = INT_MAX;
        double xmin, xmax, ymin, ymax;
        SlabMap &S = A.slabs;
        S.gx0 = 0; S.wg = P.nbx; S.gs0 = 0; S.gs1 = S.n_regular;
        A.fill_outside = false;
        const bool have_rect = c->knobs.rect != 0 && rays->mode == VV_RAYS_ANALYTIC && W >= 2 && H >= 2 && screen_rect(A, W, H, &xmin, &xmax, &ymin, &ymax);
        if (have_rect && A.phong && S.n_regular > 0) {
            // march_phong_kernel: the slabs whose own 14 x 14 pixels meet the rectangle.  A pixel is written by the slab that owns it (pin 10): when W == 1 (mod 14)
            // pixel column W-2 lies in slab column nbx-2 and belongs to nbx-1, which therefore comes along; the extra slab row is always launched.
            auto slab_of = [](double v, int lo, int hi) { const double t = floor(v / kSlab); return t < lo ? lo : (t > hi ? hi : (int)t); };
            int gx0 = slab_of(xmin, 0, P.nbx), gx1 = slab_of(xmax, -1, P.nbx - 1) + 1;
            if (P.conflict_x && gx1 == P.nbx - 1) gx1 = P.nbx;
            if (gx1 < gx0) gx1 = gx0;
            S.gx0 = gx0; S.wg = gx1 - gx0;
            A.rect.x0 = gx0 * kSlab; A.rect.x1 = gx1 >= P.nbx ? INT_MAX : gx1 * kSlab;
            if (s_count <= 1) {
                S.gs0 = slab_of(ymin, S.r0, S.r0 + S.n_regular) - S.r0; S.gs1 = slab_of(ymax, S.r0 - 1, S.r0 + S.n_regular - 1) + 1 - S.r0;
                if (S.gs1 < S.gs0) S.gs1 = S.gs0;
                A.rect.y0 = (S.r0 + S.gs0) * kSlab; A.rect.y1 = (S.r0 + S.gs1) * kSlab;
            }
            if (S.wg == 0 || S.gs1 == S.gs0) { S.wg = 0; A.rect.x0 = A.rect.x1 = A.rect.y0 = A.rect.y1 = 0; }
            A.fill_outside = true;
        }
        if (have_rect && !A.phong && M.n_strips > 0) {
            auto tile_of = [](double v, int unit, int lo, int hi) { const double t = floor(v / unit); return t < lo ? lo : (t > hi ? hi : (int)t); };
            const int ntx = M.wr;
            M.tx0 = tile_of(xmin, bw, 0, ntx); M.wr = tile_of(xmax, bw, -1, ntx - 1) + 1 - M.tx0;
            if (s_count <= 1) { M.s0 = tile_of(ymin - M.y0, bh, 0, M.n_strips); M.s1 = tile_of(ymax - M.y0, bh, -1, M.n_strips - 1) + 1; }
            if (M.wr < 0) M.wr = 0;
            if (M.s1 < M.s0) M.s1 = M.s0;
            A.rect.x0 = M.tx0 * bw; A.rect.x1 = (M.tx0 + M.wr) * bw;
            if (s_count <= 1) { A.rect.y0 = M.y0 + M.s0 * bh; A.rect.y1 = M.y0 + M.s1 * bh; }
            if (M.wr == 0 || M.s1 == M.s0) { A.rect.x0 = A.rect.x1 = A.rect.y0 = A.rect.y1 = 0; }       // (the cube is off the screen: every pixel is rad_kernel's)
        }
    }
    A.tf = c->d_tf;
    int rc = ensure(c, (void **)&c->d_rad, &c->rad_cap, (size_t)P.nbx * P.nby * sizeof(float));
    if (rc) return rc;
    // Balanced, heaviest-first tile order (StripMap::order, built by rad_kernel's extra block): analytic rays (the weights are the centre rays' chords),
    // 8 ... 1024 units of at most 16 x-adjacent tiles.  VV_LPT=0 switches it off, VV_LPT_RUN sets the unit length.
    A.strips.order = nullptr; A.order_out = nullptr; A.strips.order_run = 1;
    {
        const int wr = A.strips.wr, ns = A.strips.s1 - A.strips.s0;
        int run = wr > 0 ? (wr + (wr + 15) / 16 - 1) / ((wr + 15) / 16) : 1;                  // the strip in ceil(wr / 16) equal runs
        if (c->knobs.lpt_run > 0 && c->knobs.lpt_run <= 256) run = c->knobs.lpt_run;
        const long long units = wr > 0 ? (long long)ns * ((wr + run - 1) / run) : 0;
        // Measured (tools/ab_rep.sh, profiles/EXPERIMENTS.md part A5): aligned views of volumes up to 1 GiB gain 4 % (C2, 512^3, u8 1024^3); volumes beyond the
        // caches lose 1-6 % (the heaviest tiles marching together move more bytes), oblique views gain or lose up to 15 % with the camera: there the strips stay.
        const bool want = c->knobs.lpt > 0 ? true : (A.strips.tile_log2w == 5 && !beyond_caches && (long long)wr * ns >= 1024);      // (C1's 576 tiles: +2 %, the sort outlasts the rad pre-pass)
        if (c->knobs.lpt != 0 && want && !A.phong && rays->mode == VV_RAYS_ANALYTIC && W >= 2 && H >= 2 && units >= 8 && units <= 1024 && (long long)(wr + 1) * (ns + 1) <= 24576) {
            const size_t words = (size_t)((units + 7) / 8 * 8) * run;
            rc = ensure(c, (void **)&c->d_order, &c->order_cap, std::max(words, (size_t)65536) * sizeof(uint32_t));
            if (rc) return rc;
            A.strips.order = c->d_order; A.order_out = c->d_order; A.strips.order_run = run;
        }
    }
    A.rad = c->d_rad; A.rad_out = c->d_rad;
    A.counter = c->d_counter;

    const size_t fb = (size_t)W * H * 4;
    uint8_t *d_out = rgba_out;
    // Pixels the frame does not write (column W-1, row H-1, rows of other shards) must keep the caller's bytes.  A whole
    // frame is read back as the (W-1) x (H-1) rectangle it writes; a sharded / row-limited frame goes through a staged
    // copy of the caller's buffer (rare path).
    const bool whole = s_count <= 1 && rb == 0 && re == P.nby && W >= 2 && H >= 2;
    if (!out_on_device) {
        rc = ensure(c, (void **)&c->d_frame, &c->frame_cap, fb);
        if (rc) return rc;
        d_out = c->d_frame;
        if (!whole) HIPCHK(c, hipMemcpyAsync(d_out, rgba_out, fb, hipMemcpyHostToDevice, st));
    }
    A.pixels = (uint32_t *)d_out;
    if (((uintptr_t)d_out & 3) != 0) return fail(c, VV_ERR_INVALID, "vv_render: output buffer must be 4-byte aligned");

    if (A.instr) HIPCHK(c, hipMemsetAsync(c->d_counter, 0, 16 * sizeof(unsigned long long), st));
#ifdef VV_TIMELINE
    { const char *e = getenv("VV_TIMELINE_PTR"); A.I.timeline = e ? (unsigned long long *)strtoull(e, nullptr, 0) : nullptr; }
#endif
    c->counter_valid = A.instr;
    {
        const int layout = A.xpair ? 5 : A.V.zfast ? 4 : (A.V.bricks ? 2 : (A.V.zpair ? 3 : ((A.V.big || (A.phong && beyond_caches)) ? 1 : 0)));
        const int v[8] = {A.strips.tile_log2w, A.strips.blk_log2w, A.unroll, A.phong ? A.lds_reserve_phong : A.lds_reserve, layout, have_basis ? 1 : 0,
                          (int)fminf(density * 1000.f, 2e9f), A.phong ? 1 : 0};
        memcpy(c->last_launch, v, sizeof v);
    }
    if (c->time_frames) HIPCHK(c, hipEventRecord(c->ev0, st));
    if (A.phong) {
        if (A.fill_outside) { A.rad_out = nullptr; launch_rad(A, st); }          // the pixels beside the volume's screen rectangle (rad_kernel writes them; no radii here)
        // (linear volumes beyond the caches take the 64-bit-addressing build even below 4 GiB: the other one is compiled for 5 waves per SIMD, which only
        //  cache-resident volumes want -- 1000^3 f32: 1.884 -> 1.817 ms, tools/ab_env.sh VV_FORCE_BIG=1)
        if (A.xpair) launch_raymarch_xpair(A, st); else if (A.V.zfast) launch_raymarch_zfast(A, st); else if (A.V.bricks) { if (beyond_caches) launch_raymarch_bricked(A, st); else launch_raymarch_bricked_cached(A, st); } else if (A.V.zpair) launch_raymarch_zpair(A, st); else if (A.V.big || beyond_caches) launch_raymarch_big(A, st); else launch_raymarch(A, st);
    } else if (A.strips.n_strips > 0) {
        if (W >= 2 && H >= 2) launch_rad(A, st);
        if (A.xpair) launch_raymarch_xpair(A, st);
        else if (A.V.zfast) launch_raymarch_zfast(A, st);
        else if (A.V.bricks) { if (beyond_caches) launch_raymarch_bricked(A, st); else launch_raymarch_bricked_cached(A, st); }
        else if (A.V.zpair) launch_raymarch_zpair(A, st);
        else if (A.V.big) launch_raymarch_big(A, st);
        else launch_raymarch(A, st);
    }
    if (c->time_frames) HIPCHK(c, hipEventRecord(c->ev1, st));
    HIPCHK(c, hipGetLastError());
    c->timed = c->time_frames;
    if (!out_on_device) {
        if (whole) HIPCHK(c, hipMemcpy2DAsync(rgba_out, (size_t)W * 4, d_out, (size_t)W * 4, (size_t)(W - 1) * 4, (size_t)(H - 1), hipMemcpyDeviceToHost, st));
        else HIPCHK(c, hipMemcpyAsync(rgba_out, d_out, fb, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
    } else if (!stream) {
        HIPCHK(c, hipStreamSynchronize(st));
    }
    return VV_OK;
}

// ---- first pass images ------------------------------------------------------------------
int vv_first_pass(vv_context *c, int W, int H, const camera_params *cam, const vv_ray_source *rays,
                  uint8_t *front, uint8_t *back, int out_on_device, void *stream)
{
    if (!c) return fail(nullptr, VV_ERR_INVALID, "vv_first_pass: NULL context");
    if (!cam || !rays || !front || !back || W < 1 || H < 1) return fail(c, VV_ERR_INVALID, "vv_first_pass: bad argument");
    if (rays->mode != VV_RAYS_ANALYTIC) return fail(c, VV_ERR_INVALID, "vv_first_pass: needs an analytic ray source");
    for (int a = 0; a < 3; ++a)
        if (!(cam->scale[a] > 0.f) || !std::isfinite(cam->scale[a])) return fail(c, VV_ERR_INVALID, "vv_first_pass: camera scale must be finite and > 0");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = pick_stream(c, stream);
    FrameParams P;
    memset(&P, 0, sizeof P);
    P.W = W; P.H = H;
    for (int a = 0; a < 3; ++a) { P.cam_pos[a] = cam->origin[a]; P.scale[a] = cam->scale[a]; }
    int rc = camera_basis(c, P, cam, rays, W, H);
    if (rc) return rc;
    const size_t ib = (size_t)W * H * 4;
    uint8_t *df = front, *db = back;
    if (!out_on_device) {
        rc = ensure(c, (void **)&c->d_img, &c->img_cap, 2 * ib);
        if (rc) return rc;
        df = c->d_img; db = c->d_img + ib;
    }
    if (((uintptr_t)df & 3) || ((uintptr_t)db & 3)) return fail(c, VV_ERR_INVALID, "vv_first_pass: images must be 4-byte aligned");
    launch_first_pass(P, (uint32_t *)df, (uint32_t *)db, st);
    HIPCHK(c, hipGetLastError());
    if (!out_on_device) {
        HIPCHK(c, hipMemcpyAsync(front, df, ib, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipMemcpyAsync(back, db, ib, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
    } else if (!stream) HIPCHK(c, hipStreamSynchronize(st));
    return VV_OK;
}

// what vv_render chose for its last frame (developer aid, tests): {wave tile log2 width, block log2 width, samples per trip, LDS reserve,
// layout (0 linear, 1 linear with 64-bit addressing, 2 bricked, 3 z-pair), 1 if the view was known to the policy, density x 1000, Phong}
int vv_debug_screen_rect(int W, int H, const camera_params *cam, const vv_ray_source *rays, double out[4])
{
    if (!cam || !rays || !out || W < 1 || H < 1) return fail(nullptr, VV_ERR_INVALID, "vv_debug_screen_rect: bad argument");
    if (rays->mode != VV_RAYS_ANALYTIC) return 0;
    MarchArgs A;
    memset(&A, 0, sizeof A);
    for (int a = 0; a < 3; ++a) { A.P.cam_pos[a] = cam->origin[a]; A.P.scale[a] = cam->scale[a]; }
    int rc = camera_basis(nullptr, A.P, cam, rays, W, H);
    if (rc) return rc;
    return screen_rect(A, W, H, &out[0], &out[1], &out[2], &out[3]) ? 1 : 0;
}

int vv_debug_last_launch(vv_context *c, int out[8])
{
    if (!c || !out) return VV_ERR_INVALID;
    memcpy(out, c->last_launch, sizeof c->last_launch);
    return VV_OK;
}

int vv_set_frame_timing(vv_context *c, int on)
{
    if (!c) return fail(nullptr, VV_ERR_INVALID, "vv_set_frame_timing: NULL context");
    c->time_frames = on != 0;
    if (!on) c->timed = false;
    return VV_OK;
}

float vv_last_frame_ms(const vv_context *c)
{
    if (!c || !c->timed) return -1.f;
    float ms = -1.f;
    if (hipEventSynchronize(c->ev1) != hipSuccess) return -1.f;
    if (hipEventElapsedTime(&ms, c->ev0, c->ev1) != hipSuccess) return -1.f;
    return ms;
}

unsigned long long vv_last_sample_count(vv_context *c)
{
    if (!c || !c->counter_valid) return 0;
    unsigned long long v = 0;
    hipSetDevice(c->device);
    if (hipEventSynchronize(c->ev1) != hipSuccess) return 0;
    if (hipMemcpy(&v, c->d_counter, sizeof v, hipMemcpyDeviceToHost) != hipSuccess) return 0;
    return v;
}

// ---- slice view: invoke_slice_kernel / invoke_advanced_slice_kernel, kernel.cu:506-541 ----
static int run_slice(vv_context *c, SliceArgs &S, float *buffer, int out_on_device, void *stream)
{
    if (!c) return fail(nullptr, VV_ERR_INVALID, "vv_slice: NULL context");
    if (!buffer) return fail(c, VV_ERR_INVALID, "vv_slice: NULL buffer");
    if (!c->d_vol) return fail(c, VV_ERR_NO_VOLUME, "vv_slice: no volume loaded");
    if (S.height < 1 || S.width < 1 || S.height > 65535u * 16 || S.width > 65535u * 16)
        return fail(c, VV_ERR_INVALID, "vv_slice: bad buffer size");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = pick_stream(c, stream);
    const size_t bytes = S.height * S.width * sizeof(float);
    S.V = view_of(c); S.V_type = c->vtype;
    float *d = buffer;
    if (!out_on_device) {
        // persistent scratch instead of the reference's cudaMalloc/cudaFree per call (kernel.cu:508-518)
        int rc = ensure(c, (void **)&c->d_slice, &c->slice_cap, bytes);
        if (rc) return rc;
        d = c->d_slice;
        HIPCHK(c, hipMemcpyAsync(d, buffer, bytes, hipMemcpyHostToDevice, st));   // elements the kernel skips keep caller bytes
    }
    S.buffer = d;
    launch_slice(S, st);
    HIPCHK(c, hipGetLastError());
    if (!out_on_device) {
        HIPCHK(c, hipMemcpyAsync(buffer, d, bytes, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
    } else if (!stream) HIPCHK(c, hipStreamSynchronize(st));
    return VV_OK;
}

int vv_slice(vv_context *c, float *buffer, size_t height, size_t width, float dx, float dy, float dz,
             int orientation, const float scale[3], int legacy, int filter, int out_on_device, void *stream)
{
    if (!scale) return fail(c, VV_ERR_INVALID, "vv_slice: NULL scale");
    SliceArgs S; memset(&S, 0, sizeof S);
    S.height = height; S.width = width; S.dx = dx; S.dy = dy; S.dz = dz;
    S.orientation = orientation; S.legacy = legacy; S.advanced = 0;
    S.tex8 = filter != VV_FILTER_EXACT;
    for (int a = 0; a < 3; ++a) S.scale[a] = scale[a];
    return run_slice(c, S, buffer, out_on_device, stream);
}

int vv_slice_advanced(vv_context *c, float *buffer, size_t height, size_t width, const float trans[16],
                      const float scale[3], int filter, int out_on_device, void *stream)
{
    if (!scale || !trans) return fail(c, VV_ERR_INVALID, "vv_slice_advanced: NULL argument");
    SliceArgs S; memset(&S, 0, sizeof S);
    S.height = height; S.width = width; S.advanced = 1;
    S.tex8 = filter != VV_FILTER_EXACT;
    for (int a = 0; a < 3; ++a) S.scale[a] = scale[a];
    memcpy(S.trans, trans, 16 * sizeof(float));
    return run_slice(c, S, buffer, out_on_device, stream);
}

// ---- generator: VolumeGenerator::drawEllipsoid / drawDefaultBrain ------------------------
static int generate_impl(vv_context *c, uint8_t *out, int out_on_device, int nx, int ny, int nz, int n,
                         const float *centers, const float *axes, const uint8_t *colors, int in_place, void *stream);

int vv_generate_ellipsoids(vv_context *c, uint8_t *out, int out_on_device, int nx, int ny, int nz, int n,
                           const float *centers, const float *axes, const uint8_t *colors, void *stream)
{
    return generate_impl(c, out, out_on_device, nx, ny, nz, n, centers, axes, colors, 0, stream);
}

int vv_draw_ellipsoid(vv_context *c, uint8_t *vol, int vol_on_device, int nx, int ny, int nz,
                      const float center[3], const float axes[3], uint8_t color, void *stream)
{
    return generate_impl(c, vol, vol_on_device, nx, ny, nz, 1, center, axes, &color, 1, stream);
}

static int generate_impl(vv_context *c, uint8_t *out, int out_on_device, int nx, int ny, int nz, int n,
                         const float *centers, const float *axes, const uint8_t *colors, int in_place, void *stream)
{
    if (!c) return fail(nullptr, VV_ERR_INVALID, "vv_generate_ellipsoids: NULL context");
    if (!out || nx < 1 || ny < 1 || nz < 1 || n < 0 || n > kMaxEllipsoids || (n > 0 && (!centers || !axes || !colors)))
        return fail(c, VV_ERR_INVALID, "vv_generate_ellipsoids: bad argument (n <= 64)");
    if ((((unsigned long long)nx + 15ull) / 16ull) * (unsigned long long)ny >= (1ull << 31))
        return fail(c, VV_ERR_INVALID, "vv_generate_ellipsoids: a slice must have fewer than 2^31 16-voxel chunks");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = pick_stream(c, stream);
    const size_t bytes = (size_t)nx * ny * nz;
    uint8_t *d = out;
    void *tmp = nullptr;
    if (!out_on_device) {
        HIPCHK(c, hipMalloc(&tmp, bytes)); d = (uint8_t *)tmp;
        if (in_place && hipMemcpyAsync(d, out, bytes, hipMemcpyHostToDevice, st) != hipSuccess) { hipFree(tmp); return fail(c, VV_ERR_DEVICE, "vv_draw_ellipsoid: upload failed"); }
    }
    if (((uintptr_t)d & 15) != 0) { if (tmp) hipFree(tmp); return fail(c, VV_ERR_INVALID, "vv_generate_ellipsoids: device buffer must be 16-byte aligned"); }
    {
        int rc = ensure(c, (void **)&c->d_gen, &c->gen_cap, generate_scratch_floats(nx, ny, nz, n) * sizeof(float));
        if (rc) { if (tmp) hipFree(tmp); return rc; }
    }
    launch_generate_ellipsoids(d, nx, ny, nz, n, centers, axes, colors, in_place, c->d_gen, st);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && !out_on_device) e = hipMemcpyAsync(out, d, bytes, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess && (!out_on_device || !stream)) e = hipStreamSynchronize(st);
    if (tmp) hipFree(tmp);
    if (e != hipSuccess) return fail(c, VV_ERR_DEVICE, std::string("vv_generate_ellipsoids: ") + hipGetErrorString(e));
    return VV_OK;
}

int vv_generate_default_brain(vv_context *c, uint8_t *out, int out_on_device, int nx, int ny, int nz, void *stream)
{
    // volumegenerator.cpp:100-119: centre-major, layer-minor; later ellipsoids overwrite earlier
    static const float centers2[2][3] = { {0.25f, 0.50f, 0.50f}, {0.75f, 0.50f, 0.50f} };
    static const float layers4[4][3]  = { {0.23f, 0.30f, 0.45f}, {0.18f, 0.27f, 0.40f},
                                          {0.10f, 0.23f, 0.30f}, {0.03f, 0.20f, 0.20f} };
    static const uint8_t shades4[4] = { 60, 80, 100, 120 };
    float centers[8 * 3], axes[8 * 3]; uint8_t colors[8];
    int e = 0;
    for (int ci = 0; ci < 2; ++ci) for (int li = 0; li < 4; ++li, ++e) {
        for (int a = 0; a < 3; ++a) { centers[3*e+a] = centers2[ci][a]; axes[3*e+a] = layers4[li][a]; }
        colors[e] = shades4[li];
    }
    return vv_generate_ellipsoids(c, out, out_on_device, nx, ny, nz, 8, centers, axes, colors, stream);
}

int vv_promote_u8_to_f32(vv_context *c, const uint8_t *dev_in, float *dev_out, size_t n, void *stream)
{
    if (!c || !dev_in || !dev_out) return fail(c, VV_ERR_INVALID, "vv_promote_u8_to_f32: NULL argument");
    if (((uintptr_t)dev_in & 15) || ((uintptr_t)dev_out & 15)) return fail(c, VV_ERR_INVALID, "vv_promote_u8_to_f32: buffers must be 16-byte aligned");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = pick_stream(c, stream);
    launch_promote_u8_f32(dev_in, dev_out, n, st);
    HIPCHK(c, hipGetLastError());
    if (!stream) HIPCHK(c, hipStreamSynchronize(st));
    return VV_OK;
}

int vv_generate_noise_u8(vv_context *c, uint8_t *dev_out, int nx, int ny, int nz, uint32_t seed, void *stream)
{
    if (!c || !dev_out || nx < 1 || ny < 1 || nz < 1) return fail(c, VV_ERR_INVALID, "vv_generate_noise_u8: bad argument");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = pick_stream(c, stream);
    launch_noise_u8(dev_out, nx, ny, nz, seed, st);
    HIPCHK(c, hipGetLastError());
    if (!stream) HIPCHK(c, hipStreamSynchronize(st));
    return VV_OK;
}

} // extern "C"
