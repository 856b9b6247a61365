// kernel_hip.cpp -- see kernel_hip.h.  Thin forwarding layer over include/volviz.h.
#include "kernel_hip.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace {
vv_context *g_ctx = nullptr;
struct { const unsigned char *front, *back; int w, h; unsigned char *pixels; } g_res = {nullptr, nullptr, 0, 0, nullptr};

// checkCudaErrors behaviour (include/helper_cuda.h:763-777): report and terminate
void check(int rc, const char *what)
{
    if (rc == VV_OK) return;
    fprintf(stderr, "volviz error at %s: code=%d \"%s\"\n", what, rc, vv_last_error(g_ctx));
    exit(EXIT_FAILURE);
}
vv_context *ctx()
{
    if (!g_ctx) check(vv_init(-1, &g_ctx), "initCuda");
    return g_ctx;
}
}

vv_context *volvizContext() { return ctx(); }

extern "C" void initCuda() { ctx(); }                                   // kernel.cu:369-373

extern "C" void registerCudaResources(unsigned, unsigned, unsigned)
{
    // GL interop is wired by the host application; this headless build has
    // no GL context to register against.
    fprintf(stderr, "registerCudaResources: this is the headless build; link libvolviz_host_gl.so (registerCudaResourcesGL / runCudaGL, "
                    "kernel_hip_gl.cpp) in a GL host, or use registerHostResources()\n");
    exit(EXIT_FAILURE);
}

void registerHostResources(const unsigned char *front, const unsigned char *back, int w, int h, unsigned char *pixels)
{
    g_res.front = front; g_res.back = back; g_res.w = w; g_res.h = h; g_res.pixels = pixels;
}

extern "C" void runCuda(int width, int height, struct slice_params slice, struct camera_params camera,
                        struct shading_params shading, void *)         // kernel.cu:388-453
{
    if (!g_res.front || !g_res.back || !g_res.pixels) { fprintf(stderr, "runCuda: no resources registered\n"); exit(EXIT_FAILURE); }
    vv_ray_source rs;
    memset(&rs, 0, sizeof rs);
    rs.mode = VV_RAYS_IMAGES; rs.front = g_res.front; rs.back = g_res.back; rs.img_w = g_res.w; rs.img_h = g_res.h;
    check(vv_render(ctx(), width, height, &slice, &camera, &shading, &rs, nullptr, g_res.pixels, 0, nullptr), "runCuda");
}

extern "C" void cudaLoadVolume(byte *texels, size_t size, Vector3 dims, float transferFunction[1024], void **)
{                                                                        // kernel.cu:456-498
    check(vv_load_volume_u8(ctx(), texels, size, (int)dims.x, (int)dims.y, (int)dims.z, transferFunction), "cudaLoadVolume");
    // The copies of the volume the launch policy can sample (include/volviz.h, "optional second layouts") are built here, at load time, so that no
    // frame of the paint loop pays for one (cudaMalloc3DArray hides the texture's layout in the reference; this is where its cost belongs).  Within the
    // context's HBM budget; a copy that does not fit is skipped and vv_render takes the next layout.  Frames never build afterwards.
    const int rc = vv_prepare_layouts(ctx(), VV_LAYOUT_POLICY, nullptr);
    if (rc < 0) check(rc, "cudaLoadVolume (vv_prepare_layouts)");
    check(vv_set_layout_policy(ctx(), 0, 0), "cudaLoadVolume (vv_set_layout_policy)");
}

void invoke_slice_kernel(float *buffer, BufferParameters bp, SliceParameters sp, canonicalOrientation c, float3 scale)
{                                                                        // kernel.cu:506-519
    const float s[3] = {scale.x, scale.y, scale.z};
    check(vv_slice(ctx(), buffer, bp.height, bp.width, sp.dx, sp.dy, sp.dz, (int)c, s, 0, VV_FILTER_TEX8, 0, nullptr), "invoke_slice_kernel");
}

void invoke_slice_kernel(float *buffer, BufferParameters bp, SliceParameters sp, canonicalOrientation c)
{                                                                        // slicekernel.cu:22-49
    const float s[3] = {1.f, 1.f, 1.f};
    check(vv_slice(ctx(), buffer, bp.height, bp.width, sp.dx, sp.dy, sp.dz, (int)c, s, 1, VV_FILTER_TEX8, 0, nullptr), "invoke_slice_kernel(legacy)");
}

void invoke_advanced_slice_kernel(float *buffer, BufferParameters bp, Matrix4x4 trans, float3 scale)
{                                                                        // kernel.cu:522-541
    const float s[3] = {scale.x, scale.y, scale.z};
    check(vv_slice_advanced(ctx(), buffer, bp.height, bp.width, trans.data, s, VV_FILTER_TEX8, 0, nullptr), "invoke_advanced_slice_kernel");
}

Matrix4x4 getTransformationMatrix(SliceParameters p)                     // slicewidget.cpp:147-165
{
    Matrix4x4 m;
    check(vv_slice_matrix(p.dx, p.dy, p.dz, p.theta, p.phi, p.psi, m.data), "getTransformationMatrix");
    return m;
}

// ---- VolumeGenerator ----------------------------------------------------------------------
VolumeGenerator::VolumeGenerator(int x, int y, int z) : m_volume(new byte[(size_t)x * y * z]()), m_x(x), m_y(y), m_z(z) {}
VolumeGenerator::~VolumeGenerator() { delete[] m_volume; m_volume = 0; }

void VolumeGenerator::drawEllipsoid(const Point3 &center, const Vector3 &axes, const byte &color)
{
    if ((size_t)m_x * m_y * m_z == 0) return;
    const float c[3] = {center.x, center.y, center.z}, a[3] = {axes.x, axes.y, axes.z};
    check(vv_draw_ellipsoid(ctx(), m_volume, 0, m_x, m_y, m_z, c, a, color, nullptr), "drawEllipsoid");
}

void VolumeGenerator::drawDefaultBrain()
{
    if ((size_t)m_x * m_y * m_z == 0) return;
    // drawEllipsoid never clears, so the brain is drawn over what is there: 8 in-place passes
    // fused would differ only if the volume was not blank; keep the reference's sequence.
    static const float centers[2][3] = {{0.25f, 0.50f, 0.50f}, {0.75f, 0.50f, 0.50f}};
    static const float layers[4][3] = {{0.23f, 0.30f, 0.45f}, {0.18f, 0.27f, 0.40f}, {0.10f, 0.23f, 0.30f}, {0.03f, 0.20f, 0.20f}};
    static const byte shades[4] = {60, 80, 100, 120};
    for (int ci = 0; ci < 2; ci++)
        for (int li = 0; li < 4; li++)
            drawEllipsoid(Point3(centers[ci][0], centers[ci][1], centers[ci][2]),
                          Vector3(layers[li][0], layers[li][1], layers[li][2]), shades[li]);
}

void VolumeGenerator::saveas_raw(char *dest, bool header)
{
    check(vv_t3d_write(dest, header ? 1 : 0, m_volume, m_x, m_y, m_z), "saveas_raw");
}

void VolumeGenerator::loadfrom_raw(const char *source, bool header)
{
    int x, y, z;
    check(vv_t3d_read_header(source, header ? 1 : 0, &x, &y, &z), "loadfrom_raw");
    delete[] m_volume;
    m_x = x; m_y = y; m_z = z;
    const size_t n = (size_t)x * y * z;
    m_volume = new byte[n]();
    int rc = vv_t3d_read(source, header ? 1 : 0, m_volume, n);
    if (rc != VV_OK && rc != VV_ERR_IO) check(rc, "loadfrom_raw");       // a short file is read as far as it goes
}

VolumeGenerator::byte *VolumeGenerator::getBytes(size_t &size) { size = getVolSize(); return m_volume; }
size_t VolumeGenerator::getVolSize() { return (size_t)m_x * m_y * m_z; }
Vector3 VolumeGenerator::getDims() { return Vector3((float)m_x, (float)m_y, (float)m_z); }
