// vv_host.cpp -- host-only parts of the C-ABI: transfer-function presets and the
// .t3d container.  No device code.
//
//   vv_transfer_preset : transfer_functions.h:4-9 (g_transferEngine/Head/Mri) as closed
//                        forms evaluated in double and narrowed to float; bit-identical
//                        to the header's tables (tests/golden/tf_*.f32).
//   vv_t3d_*           : VolumeGenerator::saveas_raw / loadfrom_raw
//                        (volumegenerator.cpp:147-220) and utils/{read,write}size.cpp:
//                        optional header of three native-endian 64-bit sizes x,y,z
//                        followed by x*y*z bytes, x fastest.
#include "../../include/volviz.h"

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>

namespace {
void tent(float tf[1024], double up, double dn)
{
    for (int i = 0; i < 256; ++i) {
        double v;
        if (i < 77) v = 0.0;
        else if (i <= 153) { v = 0.9 - up * (double)(153 - i) / 255.0; if (v < 0.1) v = 0.1; }
        else               { v = 0.9 - dn * (double)(i - 153) / 255.0; if (v < 0.1) v = 0.1; }
        tf[4*i] = tf[4*i+1] = tf[4*i+2] = (float)v;
        tf[4*i+3] = (float)(0.05 * v);
    }
}
}

extern "C" {

int vv_transfer_preset(int preset, float tf[1024])
{
    if (!tf) return VV_ERR_INVALID;
    switch (preset) {
    case VV_TF_ENGINE:
        for (int i = 0; i < 256; ++i) {
            double x = (double)i / 255.0, a = 2.0 * (x * x);
            if (a > 1.0) a = 1.0;
            tf[4*i] = tf[4*i+1] = tf[4*i+2] = (float)x;
            tf[4*i+3] = (float)a;
        }
        return VV_OK;
    case VV_TF_HEAD: tent(tf, 4.0, 2.0); return VV_OK;
    case VV_TF_MRI:  tent(tf, 4.6, 2.6); return VV_OK;
    default: return VV_ERR_INVALID;
    }
}

// glwidget.cpp:678-689: table and scale by the file name's ending (QString::endsWith, case-sensitive)
int vv_dataset_preset(const char *path, int *tf_preset, float scale[3])
{
    if (!path || !tf_preset || !scale) return VV_ERR_INVALID;
    static const struct { const char *suffix; int tf; float s[3]; } rules[] = {
        {"engine.t3d",  VV_TF_ENGINE, {1.f, 1.f, 1.f}},           // :678-681
        {"head.t3d",    VV_TF_ENGINE, {1.f, 1.f, 0.8f}},          // :682-685
        {"VisMale.t3d", VV_TF_HEAD,   {1.57f, 1.f, 1.f}},         // :686-689
    };
    const size_t n = strlen(path);
    for (const auto &r : rules) {
        const size_t m = strlen(r.suffix);
        if (n >= m && memcmp(path + n - m, r.suffix, m) == 0) {
            *tf_preset = r.tf; scale[0] = r.s[0]; scale[1] = r.s[1]; scale[2] = r.s[2];
            return 1;
        }
    }
    return 0;
}

// slicewidget.cpp:147-165 with the float algebra of cs123math (REAL == float)
int vv_slice_matrix(float dx, float dy, float dz, float theta, float phi, float psi, float out[16])
{
    if (!out) return VV_ERR_INVALID;
    const float ang[3] = {theta, phi, psi};
    for (float a : ang) if (!(a >= -3.2f && a < 3.2f)) return VV_ERR_INVALID;   // slicewidget.cpp:149-154
    auto mul = [](const float a[16], const float b[16], float r[16]) {
        float t[16];
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j)
                t[4*i+j] = a[4*i] * b[j] + a[4*i+1] * b[4+j] + a[4*i+2] * b[8+j] + a[4*i+3] * b[12+j];
        memcpy(r, t, sizeof t);
    };
    auto trans = [](float x, float y, float z, float m[16]) {
        const float t[16] = {1, 0, 0, x, 0, 1, 0, y, 0, 0, 1, z, 0, 0, 0, 1};
        memcpy(m, t, sizeof t);
    };
    const float ct = cosf(theta), st = sinf(theta), cp = cosf(phi), sp = sinf(phi), cs = cosf(psi), ss = sinf(psi);
    const float rx[16] = {1, 0, 0, 0, 0, ct, -st, 0, 0, st, ct, 0, 0, 0, 0, 1};
    const float ry[16] = {cp, 0, sp, 0, 0, 1, 0, 0, -sp, 0, cp, 0, 0, 0, 0, 1};
    const float rz[16] = {cs, -ss, 0, 0, ss, cs, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    float m[16], t[16];
    trans(0.5f, 0.5f, 0.5f, m);
    trans(dx, dy, dz, t); mul(m, t, m);
    mul(m, rx, m); mul(m, ry, m); mul(m, rz, m);
    trans(-0.5f, -0.5f, -0.5f, t); mul(m, t, m);
    memcpy(out, m, sizeof m);
    return VV_OK;
}

// Window::renderSlice, PRO_SLICING branch (window.cpp:425-443): the cutting plane the free-form slice sliders give the 3D view.
//   offset = Vector4(dx, dy, dz, 0) += Vector4(.5, .5, .5, 0)                                         (binary32 additions)
//   normal = getTransMat(.5) * getRotXMat(theta) * getRotYMat(phi) * getRotZMat(psi) * getTransMat(-.5) * Vector4(0, 0, 1, 0)
// evaluated as the reference's operators do: the matrix products left to right (mat4::operator*=, CS123Algebra.h:429-471: each element
// a sum of four products in index order), then mat4 * vec4 (CS123Algebra.h:340-345).  The translations drop out of the result in exact
// arithmetic; they are kept because they take part in the float sums.  Pinned bit for bit against the compiled reference
// (oracle/ref_shim.cpp ref_cut_plane_pro, tests/golden/cut_planes_pro.json).  No angle range is asserted on this path.
int vv_cut_plane_from_euler(float dx, float dy, float dz, float theta, float phi, float psi, float point[3], float normal[3])
{
    if (!point || !normal) return VV_ERR_INVALID;
    auto mul = [](const float a[16], const float b[16], float r[16]) {
        float t[16];
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j)
                t[4*i+j] = a[4*i] * b[j] + a[4*i+1] * b[4+j] + a[4*i+2] * b[8+j] + a[4*i+3] * b[12+j];
        memcpy(r, t, sizeof t);
    };
    const float ct = cosf(theta), st = sinf(theta), cp = cosf(phi), sp = sinf(phi), cs = cosf(psi), ss = sinf(psi);
    const float rx[16] = {1, 0, 0, 0, 0, ct, -st, 0, 0, st, ct, 0, 0, 0, 0, 1};
    const float ry[16] = {cp, 0, sp, 0, 0, 1, 0, 0, -sp, 0, cp, 0, 0, 0, 0, 1};
    const float rz[16] = {cs, -ss, 0, 0, ss, cs, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    float m[16] = {1, 0, 0, 0.5f, 0, 1, 0, 0.5f, 0, 0, 1, 0.5f, 0, 0, 0, 1};
    const float tb[16] = {1, 0, 0, -0.5f, 0, 1, 0, -0.5f, 0, 0, 1, -0.5f, 0, 0, 0, 1};
    mul(m, rx, m); mul(m, ry, m); mul(m, rz, m); mul(m, tb, m);
    const float v[4] = {0.f, 0.f, 1.f, 0.f};
    for (int i = 0; i < 3; ++i) normal[i] = m[4*i] * v[0] + m[4*i+1] * v[1] + m[4*i+2] * v[2] + m[4*i+3] * v[3];
    point[0] = dx + 0.5f; point[1] = dy + 0.5f; point[2] = dz + 0.5f;
    return VV_OK;
}

// GLWidget::setSliceCanonical, glwidget.cpp:757-776
int vv_cut_plane_canonical(int orientation, float displace, float point[3], float normal[3])
{
    if (!point || !normal) return VV_ERR_INVALID;
    point[0] = point[1] = point[2] = 0.f; normal[0] = normal[1] = normal[2] = 0.f;
    switch (orientation) {
    case VV_HORIZONTAL: normal[1] = 1.f; point[1] = displace; return VV_OK;
    case VV_SAGITTAL:   normal[2] = 1.f; point[2] = displace; return VV_OK;
    case VV_CORONAL:    normal[0] = 1.f; point[0] = displace; return VV_OK;
    default: return VV_ERR_INVALID;         // the reference's switch leaves the plane as it was
    }
}

// glwidget.cpp:232-258: slice_params marshalling
int vv_cut_plane_to_slice_params(int slice_type, const float point[3], const float normal[3],
                                 int flip_cross_section, struct slice_params *out)
{
    if (!out) return VV_ERR_INVALID;
    out->type = slice_type;
    for (int i = 0; i < 6; ++i) out->params[i] = 0.f;
    if (slice_type == SLICE_NONE) return VV_OK;
    if (slice_type != SLICE_PLANE && slice_type != SLICE_PLANE_CUT) return VV_ERR_INVALID;
    if (!point || !normal) return VV_ERR_INVALID;
    float n[3] = {normal[0], normal[1], normal[2]};
    const bool neg = flip_cross_section ? ((double)n[1] < -1e-6) : ((double)n[1] > 1e-6);
    if (neg) { n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2]; }
    for (int i = 0; i < 3; ++i) { out->params[i] = point[i]; out->params[3 + i] = n[i]; }
    return VV_OK;
}

// ---- camera / cutting-plane controls (glwidget.cpp:426-535, 607-620); double arithmetic ----
namespace {
struct V3 { double x, y, z; };
inline V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 normalized(V3 a) { double l = std::sqrt(dot(a, a)); return l > 0 ? V3{a.x / l, a.y / l, a.z / l} : a; }
inline V3 ld(const float *p) { return {p[0], p[1], p[2]}; }
inline void st(float *p, V3 v) { p[0] = (float)v.x; p[1] = (float)v.y; p[2] = (float)v.z; }
} // namespace

int vv_camera_orbit_drag(const float position[3], int dx, int dy, float out_position[3], float out_look[3])
{
    if (!position || !out_position) return VV_ERR_INVALID;
    const V3 p = ld(position);
    const float r = (float)std::sqrt(dot(p, p));
    if (!(r > 0.f)) return VV_ERR_INVALID;
    float theta = (float)(std::acos(p.y / r) - dy / 200.f);                 // :436-438
    const float phi = (float)(std::atan2(p.z, p.x) + dx / 200.f);
    if (theta < 0.1f) theta = 0.1f;                                          // :440-444
    if (theta > M_PI - 0.1f) theta = (float)(M_PI - 0.1f);
    const V3 q = {r * std::sin((double)theta) * std::cos((double)phi), r * std::cos((double)theta),
                  r * std::sin((double)theta) * std::sin((double)phi)};     // :446
    st(out_position, q);
    if (out_look) st(out_look, normalized(sub({0, 0, 0}, ld(out_position))));   // camera->lookAt(origin), camera.cpp:27
    return VV_OK;
}

int vv_camera_zoom(const float position[3], const float look[3], int delta, float out_position[3])
{
    if (!position || !look || !out_position) return VV_ERR_INVALID;
    const double k = delta / 200.f;                                          // :615
    for (int i = 0; i < 3; ++i) out_position[i] = delta ? (float)(position[i] + look[i] * k) : position[i];
    return VV_OK;
}

int vv_cut_plane_from_drag(const float position[3], const float look_[3], const float up_[3], float aspect,
                           const float press[2], const float release[2],
                           float point[3], float normal[3], float plane_up[3], float plane_right[3])
{
    if (!position || !look_ || !up_ || !press || !release || !point || !normal || !(aspect > 0.f)) return VV_ERR_INVALID;
    // camera frame of Camera::lazyComputeTransform (camera.cpp:78-91): rows side, up, -look
    const V3 eye = ld(position), look = normalized(ld(look_));
    const V3 side = normalized(cross(look, ld(up_))), up = normalized(cross(side, look));
    // inverse of perspective(45, aspect, .1, 100) * view applied to NDC points (x, y, z, 1):
    // eye-space point of NDC (x, y, z) is (x*a*t, y*t, -1) * w', w' = 1 / ((z*(n-f) + (f+n)) / (2fn))
    const double n = 0.1, f = 100.0, t = std::tan(45.0 * M_PI / 360.0);
    auto unproject = [&](double x, double y, double z) -> V3 {
        const double w = (2.0 * f * n) / ((f + n) - z * (f - n));           // distance along -z_eye
        const double ex = x * aspect * t * w, ey = y * t * w;
        return {eye.x + side.x * ex + up.x * ey + look.x * w,
                eye.y + side.y * ex + up.y * ey + look.y * w,
                eye.z + side.z * ex + up.z * ey + look.z * w};
    };
    auto glc = [](double v) { return v * 2.0 - 1.0; };                        // :85-88
    const V3 front = unproject(glc(release[0]), -glc(release[1]), -1.0);     // :491-495
    const V3 back  = unproject(glc(release[0]), -glc(release[1]),  1.0);
    const V3 sidep = unproject(glc(press[0]),   -glc(press[1]),   -1.0);
    const V3 a = normalized(sub(back, front)), b = normalized(sub(sidep, front));   // :503-504
    const V3 nrm_ = cross(a, b);                                             // :507-510 (not normalised)
    point[0] = (float)((front.x + 1.0) / 2.0);                               // :512-514, nrm() :90-93
    point[1] = (float)((front.y + 1.0) / 2.0);
    point[2] = (float)((front.z + 1.0) / 2.0);
    st(normal, nrm_);
    // inverse * (0,-1,0,0) and inverse * (1,0,0,0): directions; the projection scales them by
    // tan(fov/2) (and the aspect) before the camera rotation carries them to world space (:497-498)
    if (plane_up)    st(plane_up,    {-up.x * t, -up.y * t, -up.z * t});
    if (plane_right) st(plane_right, {side.x * aspect * t, side.y * aspect * t, side.z * aspect * t});
    return VV_OK;
}

int vv_cut_plane_drag(float point[3], const float plane_up[3], const float plane_right[3],
                      int dx, int dy, int width, int height)
{
    if (!point || !plane_up || !plane_right || width <= 0 || height <= 0) return VV_ERR_INVALID;
    for (int i = 0; i < 3; ++i) {                                            // :448-449
        double p = point[i] + (double)plane_right[i] * dx / width * 3.5;
        p = (float)p + (double)plane_up[i] * dy / height * 3.5;
        point[i] = (float)p;
    }
    return VV_OK;
}

// slicewidget.cpp:108-121
int vv_slice_to_bgra(const float *slice, size_t height, size_t width, uint8_t *bgra)
{
    if (!slice || !bgra || height == 0 || width == 0) return VV_ERR_INVALID;
    const size_t size = width * height;
    for (size_t j = 0; j < height; ++j)
        for (size_t i = 0; i < width; ++i) {
            const size_t offset = j * height + i;                  // :114 (height as stride)
            if (offset >= size) continue;                           // the reference reads out of bounds here
            const float f = slice[offset] * 255;
            const unsigned val = f > 0.f ? (f >= 4294967040.f ? 0xffffffffu : (unsigned)f) : 0u;   // (unsigned)(f*255)
            const size_t dst = size - offset;                       // :116 mirrored
            if (dst >= size) continue;                              // offset 0 -> one past the end in the reference
            uint8_t *p = bgra + 4 * dst;
            p[0] = p[1] = p[2] = (uint8_t)val; p[3] = 255;          // BGRA(val,val,val,255), CS123Common.h:24-27
        }
    return VV_OK;
}

// header-less files are 128 x 256 x 256 (volumegenerator.cpp:204-208)
int vv_t3d_read_header(const char *path, int header, int *nx, int *ny, int *nz)
{
    if (!path || !nx || !ny || !nz) return VV_ERR_INVALID;
    if (!header) { *nx = 128; *ny = 256; *nz = 256; return VV_OK; }
    FILE *f = fopen(path, "rb");
    if (!f) return VV_ERR_IO;
    uint64_t d[3];
    size_t got = fread(d, sizeof(uint64_t), 3, f);
    fclose(f);
    if (got != 3) return VV_ERR_IO;
    if (d[0] == 0 || d[1] == 0 || d[2] == 0 || d[0] > 0x7fffffffu || d[1] > 0x7fffffffu || d[2] > 0x7fffffffu)
        return VV_ERR_INVALID;
    *nx = (int)d[0]; *ny = (int)d[1]; *nz = (int)d[2];
    return VV_OK;
}

int vv_t3d_read(const char *path, int header, uint8_t *dst, size_t capacity)
{
    int nx, ny, nz;
    int rc = vv_t3d_read_header(path, header, &nx, &ny, &nz);
    if (rc) return rc;
    const size_t n = (size_t)nx * ny * nz;
    if (!dst || capacity < n) return VV_ERR_INVALID;
    FILE *f = fopen(path, "rb");
    if (!f) return VV_ERR_IO;
    if (header && fseek(f, 3 * sizeof(uint64_t), SEEK_SET) != 0) { fclose(f); return VV_ERR_IO; }
    size_t got = fread(dst, 1, n, f);
    fclose(f);
    // a short file leaves the tail as it was, like ifstream::read in the reference;
    // report it so callers can tell
    return got == n ? VV_OK : VV_ERR_IO;
}

int vv_t3d_write(const char *path, int header, const uint8_t *src, int nx, int ny, int nz)
{
    if (!path || !src || nx < 1 || ny < 1 || nz < 1) return VV_ERR_INVALID;
    FILE *f = fopen(path, "wb");
    if (!f) return VV_ERR_IO;
    bool ok = true;
    if (header) {
        uint64_t d[3] = {(uint64_t)nx, (uint64_t)ny, (uint64_t)nz};
        ok = fwrite(d, sizeof(uint64_t), 3, f) == 3;
    }
    const size_t n = (size_t)nx * ny * nz;
    ok = ok && fwrite(src, 1, n, f) == n;
    ok = (fclose(f) == 0) && ok;
    return ok ? VV_OK : VV_ERR_IO;
}

} // extern "C"
