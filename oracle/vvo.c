/*
 * vvo.c -- CPU oracle (TEST INFRASTRUCTURE ONLY; see vvo.h for the pinning status).
 *
 * Each function cites the reference file:line it restates.  Citations are relative
 * to the reference tree (jacobstern/volume-viz).  No reference text is copied; the
 * behaviour is restated.
 *
 * Build: gcc -O2 -std=gnu11 -ffp-contract=off -fno-fast-math -fopenmp  (oracle/Makefile)
 */
#include "vvo.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ============================================================================
 * Generator: VolumeGenerator::drawEllipsoid  volumegenerator.cpp:31-97
 *   REAL == float (cs123math/CS123Algebra.h:16), so every operation is binary32;
 *   the comparisons are against double literals 1.0 and 0.99 (float promoted).
 * ========================================================================== */
void vvo_draw_ellipsoid(uint8_t *vol, int nx, int ny, int nz,
                        const float c[3], const float a[3], uint8_t color)
{
    for (int k = 0; k < nz; k++) {
        for (int j = 0; j < ny; j++) {
            for (int i = 0; i < nx; i++) {
                /* volumegenerator.cpp:41  (int arithmetic, as the reference) */
                int offset = k * ny * nx + j * nx + i;
                float fi = ((float)i) / ((float)nx);          /* :44 */
                float fj = ((float)j) / ((float)ny);          /* :45 */
                float fk = ((float)k) / ((float)nz);          /* :46 */
                float ex = (c[0] - fi) / a[0];
                float ey = (c[1] - fj) / a[1];
                float ez = (c[2] - fk) / a[2];
                float q  = ((ex * ex) + (ey * ey)) + (ez * ez);   /* :57-59 */
                if ((double)q < 1.0) vol[offset] = color;       /* else keep, :60-62 */
                if ((double)fi >= 0.99) vol[offset] = (uint8_t)4; /* marker slab, :85-87 */
            }
        }
    }
}

/* VolumeGenerator::drawDefaultBrain  volumegenerator.cpp:100-119
 * (constructor zero-fill: volumegenerator.cpp:12-23) */
void vvo_draw_default_brain(uint8_t *vol, int nx, int ny, int nz)
{
    static const float centers[2][3] = { {0.25f, 0.50f, 0.50f}, {0.75f, 0.50f, 0.50f} };
    static const float layers[4][3]  = { {0.23f, 0.30f, 0.45f}, {0.18f, 0.27f, 0.40f},
                                         {0.10f, 0.23f, 0.30f}, {0.03f, 0.20f, 0.20f} };
    static const uint8_t shades[4] = { 60, 80, 100, 120 };
    memset(vol, 0, (size_t)nx * ny * nz);
    for (int ci = 0; ci < 2; ci++)
        for (int li = 0; li < 4; li++)
            vvo_draw_ellipsoid(vol, nx, ny, nz, centers[ci], layers[li], shades[li]);
}

/* ============================================================================
 * Transfer functions: transfer_functions.h:4-9 restated as closed forms in double,
 * narrowed to float (the header's literals are decimal doubles narrowed to float).
 * Checked entry-by-entry against tests/golden/tf_*.f32.
 * ========================================================================== */
static void tf_tent(float tf[1024], double up, double dn)
{
    for (int i = 0; i < 256; i++) {
        double v;
        if (i < 77) v = 0.0;
        else if (i <= 153) { v = 0.9 - up * (double)(153 - i) / 255.0; if (v < 0.1) v = 0.1; }
        else               { v = 0.9 - dn * (double)(i - 153) / 255.0; if (v < 0.1) v = 0.1; }
        tf[4*i+0] = tf[4*i+1] = tf[4*i+2] = (float)v;
        tf[4*i+3] = (float)(0.05 * v);
    }
}

void vvo_transfer_preset(int preset, float tf[1024])
{
    if (preset == VV_TF_ENGINE) {
        for (int i = 0; i < 256; i++) {
            double x = (double)i / 255.0, al = 2.0 * (x * x);
            if (al > 1.0) al = 1.0;
            tf[4*i+0] = tf[4*i+1] = tf[4*i+2] = (float)x;
            tf[4*i+3] = (float)al;
        }
    } else if (preset == VV_TF_HEAD) {
        tf_tent(tf, 4.0, 2.0);
    } else {
        tf_tent(tf, 4.6, 2.6);
    }
}

/* ============================================================================
 * helpers restating kernel.cu:53-71 and include/helper_math.h semantics
 * ========================================================================== */
/* ----------------------------------------------------------------------------
 * Arithmetic model.  VVO_MODEL 0 (the default, what every parity test uses) = the pins of
 * DESIGN.md section 3, which the product implements bit for bit.  The other models are built
 * into separate libraries (oracle/Makefile `models`) only to MEASURE how far frames move if
 * the choices nobody can read off the CUDA build (nvcc 5.5, -use_fast_math, sm_21) went the
 * other way -- tests/test_oracle_models.py; they are not parity targets:
 *   1 FMAD     a*b+c contracted wherever the compiler can (nvcc's default --fmad=true): the
 *              same source built with -ffp-contract=fast -mfma
 *   2 FAST     FMAD + what -use_fast_math adds: x/y as x * rcp(y), sqrt(x) as x * rsqrt(x),
 *              rsqrt(x) as rcp(sqrt(x)) (each step correctly rounded here; the hardware's
 *              approximations are within 1-2 ulp of these), denormals flushed (FTZ/DAZ)
 *   3 TEXTRUNC the texture unit's 8-bit interpolation weights truncated instead of rounded
 * -------------------------------------------------------------------------- */
#ifndef VVO_MODEL
#define VVO_MODEL 0
#endif
#if VVO_MODEL == 2
#include <xmmintrin.h>
static inline float vvo_rcp(float y) { return 1.0f / y; }
static inline float vvo_rsqrt(float x) { return vvo_rcp(sqrtf(x)); }
#define FDIV(a, b) ((a) * vvo_rcp(b))
#define FSQRT(x)   ((x) * vvo_rsqrt(x) == (x) * vvo_rsqrt(x) ? ((x) == 0.0f ? 0.0f : (x) * vvo_rsqrt(x)) : sqrtf(x))
#define FRSQRT(x)  vvo_rsqrt(x)
#else
#define FDIV(a, b) ((a) / (b))
#define FSQRT(x)   sqrtf(x)
#define FRSQRT(x)  (1.0f / sqrtf(x))
#endif
const char *vvo_model(void)
{
    static const char *const names[] = {"pins", "fmad", "fast", "textrunc"};
    return names[VVO_MODEL];
}

typedef struct { float x, y, z; } f3;

static inline f3 mk3(float x, float y, float z) { f3 r = {x, y, z}; return r; }
static inline f3 add3(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline f3 sub3(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline f3 mul3(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline f3 scl3(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
static inline float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; } /* helper_math.h dot */
/* kernel.cu:53-57 vectorLength */
static inline float vlen(f3 v) { return FSQRT(v.x * v.x + v.y * v.y + v.z * v.z); }
/* helper_math.h:1152-1155 clamp = fmaxf(a, fminf(f, b)) */
static inline float clampf(float f, float a, float b) { return fmaxf(a, fminf(f, b)); }
static inline int clampi(int f, int a, int b) { int m = f < b ? f : b; return a > m ? a : m; }

/* kernel.cu:65-71 boundsCheck: false for NaN */
static inline int bounds_check(f3 p)
{
    return p.x < 1.0f && p.x >= 0.0f && p.y < 1.0f && p.y >= 0.0f && p.z < 1.0f && p.z >= 0.0f;
}

/* ============================================================================
 * Texture unit model (hardware in the reference; ORACLE PIN):
 *   CUDA C Programming Guide, "Texture Fetching", linear filtering, normalised
 *   coordinates, clamp addressing (kernel.cu:485-489):
 *     xB = x*N - 0.5 ; i = floor(xB) ; alpha = frac(xB)
 *     TEX8 : alpha rounded to 8 fractional bits (1.8 fixed point), nearest, ties to even
 *     texel indices i, i+1 clamped to [0, N-1]
 *   lerp order x, then y, then z, each as fmaf(w, b - a, a).
 *   u8 texels (cudaReadModeNormalizedFloat, kernel.cu:46) are filtered as raw
 *   0..255 values and normalised after filtering: tex = L / 255.
 * ========================================================================== */
static inline void tex_axis(float x, int n, int filter, int *i0, int *i1, float *w)
{
    float xb = fmaf(x, (float)n, -0.5f);
    float fl = floorf(xb);
    float a = xb - fl;
#if VVO_MODEL == 3
    if (filter == VV_FILTER_TEX8) a = floorf(a * 256.0f) * (1.0f / 256.0f);  /* measurement model: truncated */
#else
    if (filter == VV_FILTER_TEX8) a = rintf(a * 256.0f) * (1.0f / 256.0f);   /* ties to even */
#endif
    int i = (int)fl;
    *i0 = clampi(i, 0, n - 1);
    *i1 = clampi(i + 1, 0, n - 1);
    *w = a;
}

static inline float texel_raw(const vvo_volume *v, int x, int y, int z)
{
    size_t o = ((size_t)z * v->ny + y) * v->nx + x;
    if (v->type == VV_VOXEL_U8) return (float)((const uint8_t *)v->data)[o];
    return ((const float *)v->data)[o];
}

/* filtered value in storage units: 0..255 for u8 volumes, as-is for f32 volumes */
static inline float tex3d_raw(const vvo_volume *v, float x, float y, float z, int filter)
{
    int x0, x1, y0, y1, z0, z1; float wx, wy, wz;
    tex_axis(x, v->nx, filter, &x0, &x1, &wx);
    tex_axis(y, v->ny, filter, &y0, &y1, &wy);
    tex_axis(z, v->nz, filter, &z0, &z1, &wz);
    float c000 = texel_raw(v, x0, y0, z0), c100 = texel_raw(v, x1, y0, z0);
    float c010 = texel_raw(v, x0, y1, z0), c110 = texel_raw(v, x1, y1, z0);
    float c001 = texel_raw(v, x0, y0, z1), c101 = texel_raw(v, x1, y0, z1);
    float c011 = texel_raw(v, x0, y1, z1), c111 = texel_raw(v, x1, y1, z1);
    float c00 = fmaf(wx, c100 - c000, c000);
    float c10 = fmaf(wx, c110 - c010, c010);
    float c01 = fmaf(wx, c101 - c001, c001);
    float c11 = fmaf(wx, c111 - c011, c011);
    float c0 = fmaf(wy, c10 - c00, c00);
    float c1 = fmaf(wy, c11 - c01, c01);
    return fmaf(wz, c1 - c0, c0);
}

float vvo_tex3d(const vvo_volume *v, float x, float y, float z, int filter)
{
    float L = tex3d_raw(v, x, y, z, filter);
    return v->type == VV_VOXEL_U8 ? FDIV(L, 255.0f) : L;
}

/* float -> unsigned char conversion as the GPU does it: truncate toward zero,
 * saturate to [0,255], NaN -> 0 (PTX cvt.rzi.u8.f32 / v_cvt_u32_f32). */
static inline uint8_t sat_u8(float f)
{
    if (!(f > 0.0f)) return 0;
    if (f >= 255.0f) return 255;
    return (uint8_t)(int)f;
}

/* kernel.cu:99-105 sample(): (uchar)(0xff * tex3D), 0 outside [0,1)^3.
 * ORACLE PIN 2: for u8 volumes tex3D = L/255 and 0xff * tex3D = L: the normalisation and
 * the multiplication cancel (255 * RN(k/255) == k for every k in 0..255, so a constant
 * region classifies exactly), i.e. the index is trunc(L).  f32 volumes: trunc(255 * L). */
static inline uint8_t sample_u8(const vvo_volume *v, f3 p, int filter)
{
    if (!bounds_check(p)) return 0;
    float L = tex3d_raw(v, p.x, p.y, p.z, filter);
    return sat_u8(v->type == VV_VOXEL_U8 ? L : 255.0f * L);
}

/* kernel.cu:136 (and :585): (pos - .5) / scale + .5.
 * ORACLE PIN 3: the reference is built with -use_fast_math (hellocudainterop.pro:52), which
 * turns float division into multiplication by the reciprocal and contracts a*b+c; the
 * expression is therefore evaluated as fma(pos - .5, 1/scale, .5), 1/scale correctly rounded.
 * With scale == 1 this equals the literal expression bit for bit. */
static inline f3 inv3(f3 s) { return mk3(1.0f / s.x, 1.0f / s.y, 1.0f / s.z); }
static inline f3 to_tex(f3 p, f3 inv_scale)
{
    return mk3(fmaf(p.x - 0.5f, inv_scale.x, 0.5f), fmaf(p.y - 0.5f, inv_scale.y, 0.5f),
               fmaf(p.z - 0.5f, inv_scale.z, 0.5f));
}

/* ============================================================================
 * Slice kernels
 * ========================================================================== */
/* kernel.cu:543-597 slice_kernel (5-arg); legacy: slicekernel.cu:51-82 (4-arg) */
void vvo_slice(const vvo_volume *v, float *buffer, size_t height, size_t width,
               float dx, float dy, float dz, int orientation, const float scale[3],
               int legacy, int filter)
{
    f3 sc = inv3(mk3(scale[0], scale[1], scale[2]));
    for (size_t j = 0; j < height; j++) {
        for (size_t i = 0; i < width; i++) {
            size_t offset = j * height + i;                 /* kernel.cu:550 (height as stride) */
            if (offset >= height * width) continue;         /* height > width: the reference writes out of bounds; skipped */
            float u = ((float)i) / ((float)width);
            float w = ((float)j) / ((float)height);
            f3 pos = mk3(0.f, 0.f, 0.f);
            if (legacy) {                                   /* slicekernel.cu:62-64 */
                pos.x = u; pos.y = w; pos.z = 0.f;
            } else {
                switch (orientation) {                      /* kernel.cu:559-579 */
                case VV_SAGITTAL:   pos.z += 0.f; pos.y += w;   pos.x += u;   break;
                case VV_HORIZONTAL: pos.z += u;   pos.y += 0.f; pos.x += w;   break;
                case VV_CORONAL:    pos.z += u;   pos.y += w;   pos.x += 0.f; break;
                default: break;                             /* FREE_FORM etc.: pos stays 0 */
                }
            }
            pos.x += dx; pos.y += dy; pos.z += dz;          /* :581-583 */
            float s;
            if (legacy) {
                /* slicekernel.cu:70: unconditional tex3D; clamp addressing does the rest */
                s = vvo_tex3d(v, pos.x, pos.y, pos.z, filter);
            } else {
                pos = to_tex(pos, sc);                      /* :585 */
                s = bounds_check(pos) ? vvo_tex3d(v, pos.x, pos.y, pos.z, filter) : 0.f; /* :587-593 */
            }
            buffer[offset] = s;
        }
    }
}

/* kernel.cu:599-644 advanced_slice_kernel */
void vvo_slice_advanced(const vvo_volume *v, float *buffer, size_t height, size_t width,
                        const float t[16], const float scale[3], int filter)
{
    for (size_t j = 0; j < height; j++) {
        for (size_t i = 0; i < width; i++) {
            size_t offset = j * height + i;                 /* :604 */
            if (offset >= height * width) continue;
            float rx = ((float)i) / ((float)width);
            float ry = ((float)j) / ((float)height);
            float rz = 0.5f, rw = 1.f;
            f3 p;
            p.x = t[0] * rx + t[1] * ry + t[2]  * rz + t[3]  * rw;   /* :616-618, row-major */
            p.y = t[4] * rx + t[5] * ry + t[6]  * rz + t[7]  * rw;
            p.z = t[8] * rx + t[9] * ry + t[10] * rz + t[11] * rw;
            f3 inv = inv3(mk3(scale[0], scale[1], scale[2]));
            p.x *= inv.x; p.y *= inv.y; p.z *= inv.z;                /* :620-622 (pin 3: reciprocal) */
            p = to_tex(p, inv);                                      /* :624 (second scale) */
            buffer[offset] = bounds_check(p) ? vvo_tex3d(v, p.x, p.y, p.z, filter) : 0.f;
        }
    }
}

/* slicewidget.cpp:147-165 getTransformationMatrix, with the float 4x4 algebra of
 * cs123math/CS123Algebra.h:429-447 (row-major product) and CS123Matrix.cpp:27-62. */
static void m4mul(const float a[16], const float b[16], float r[16])
{
    float t[16];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            t[4*i+j] = a[4*i+0] * b[0+j] + a[4*i+1] * b[4+j] + a[4*i+2] * b[8+j] + a[4*i+3] * b[12+j];
    memcpy(r, t, sizeof t);
}
static void m4trans(float x, float y, float z, float m[16])
{
    const float t[16] = {1,0,0,x, 0,1,0,y, 0,0,1,z, 0,0,0,1};
    memcpy(m, t, sizeof t);
}
void vvo_slice_matrix(float dx, float dy, float dz, float theta, float phi, float psi, float out[16])
{
    float c2o[16], o2c[16], tr[16], rx[16], ry[16], rz[16], m[16];
    m4trans(-0.5f, -0.5f, -0.5f, c2o);
    m4trans(0.5f, 0.5f, 0.5f, o2c);
    m4trans(dx, dy, dz, tr);
    /* cos/sin of a float argument: the float overloads (CS123Matrix.cpp:37-60, REAL=float) */
    float ct = cosf(theta), st = sinf(theta);
    float cp = cosf(phi),   sp = sinf(phi);
    float cs = cosf(psi),   ss = sinf(psi);
    const float RX[16] = {1,0,0,0, 0,ct,-st,0, 0,st,ct,0, 0,0,0,1};
    const float RY[16] = {cp,0,sp,0, 0,1,0,0, -sp,0,cp,0, 0,0,0,1};
    const float RZ[16] = {cs,-ss,0,0, ss,cs,0,0, 0,0,1,0, 0,0,0,1};
    memcpy(rx, RX, sizeof RX); memcpy(ry, RY, sizeof RY); memcpy(rz, RZ, sizeof RZ);
    /* origin2center * trans * rotX * rotY * rotZ * center2origin, left to right */
    m4mul(o2c, tr, m); m4mul(m, rx, m); m4mul(m, ry, m); m4mul(m, rz, m); m4mul(m, c2o, m);
    memcpy(out, m, sizeof m);
}

/* ============================================================================
 * First pass (firstpass.vert:6, firstpass.frag:4, glwidget.cpp:198-228)
 * ========================================================================== */
static inline f3 norm3(f3 v) { float l = vlen(v); return mk3(v.x / l, v.y / l, v.z / l); }
static inline f3 cross3(f3 a, f3 b)
{
    return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}

/* Analytic replacement of the GL pass (ORACLE PIN): the ray through the centre of
 * pixel (x,y) of the W x H frame under perspective(fovY, aspect) (glwidget.cpp:338)
 * and the look-at basis of camera.cpp:78-91, intersected with the box
 * [-scale, scale]; cubeSpace = world/2 + 0.5 (firstpass.vert:6 with the glScalef
 * of glwidget.cpp:198).  A face that is not visible leaves the clear colour 0. */
static void analytic_endpoints_vis(const vv_ray_source *rs, const struct camera_params *cam,
                                   int W, int H, int x, int y, f3 *front, f3 *back, int *vis_f, int *vis_b);
static void analytic_endpoints(const vv_ray_source *rs, const struct camera_params *cam,
                               int W, int H, int x, int y, f3 *front, f3 *back)
{
    int vf, vb;
    analytic_endpoints_vis(rs, cam, W, H, x, y, front, back, &vf, &vb);
}
static void analytic_endpoints_vis(const vv_ray_source *rs, const struct camera_params *cam,
                                   int W, int H, int x, int y, f3 *front, f3 *back, int *vis_f, int *vis_b)
{
    *vis_f = 0; *vis_b = 0;
    f3 look = norm3(mk3(rs->look[0], rs->look[1], rs->look[2]));
    f3 up0  = mk3(rs->up[0], rs->up[1], rs->up[2]);
    f3 side = norm3(cross3(look, up0));
    f3 up   = norm3(cross3(side, look));
    float aspect = rs->aspect > 0.f ? rs->aspect : (float)W / (float)H;
    float th = (float)tan((double)cam->fovY * M_PI / 360.0);
    float ndx = (2.0f * ((float)x + 0.5f)) / (float)W - 1.0f;
    float ndy = (2.0f * ((float)y + 0.5f)) / (float)H - 1.0f;
    float sx = ndx * (th * aspect), sy = ndy * th;
    f3 d = add3(add3(scl3(side, sx), scl3(up, sy)), look);
    f3 o = mk3(cam->origin[0], cam->origin[1], cam->origin[2]);
    float tmin = -INFINITY, tmax = INFINITY;
    int miss = 0;
    const float oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z};
    for (int a = 0; a < 3; a++) {
        float s = cam->scale[a];
        if (dd[a] != 0.0f) {
            float t1 = (-s - oo[a]) / dd[a], t2 = (s - oo[a]) / dd[a];
            float lo = fminf(t1, t2), hi = fmaxf(t1, t2);
            tmin = fmaxf(tmin, lo); tmax = fminf(tmax, hi);
        } else if (oo[a] < -s || oo[a] > s) miss = 1;
    }
    *front = mk3(0.f, 0.f, 0.f); *back = mk3(0.f, 0.f, 0.f);
    if (miss || !(tmin <= tmax) || !(tmax > 0.0f)) return;
    *vis_b = 1;
    f3 pb = add3(o, scl3(d, tmax));
    *back = mk3(pb.x * 0.5f + 0.5f, pb.y * 0.5f + 0.5f, pb.z * 0.5f + 0.5f);
    if (tmin > 0.0f) {
        *vis_f = 1;
        f3 pf = add3(o, scl3(d, tmin));
        *front = mk3(pf.x * 0.5f + 0.5f, pf.y * 0.5f + 0.5f, pf.z * 0.5f + 0.5f);
    }
    if (rs->quantize8) {       /* GL float -> UNORM8: round(clamp(c,0,1)*255), then kernel.cu:320-321 /255 */
        float *p[6] = {&front->x, &front->y, &front->z, &back->x, &back->y, &back->z};
        for (int i = 0; i < 6; i++) {
            float c = clampf(*p[i], 0.f, 1.f);
            float q = floorf(c * 255.0f + 0.5f);
            *p[i] = q / 255.f;
        }
    }
}

/* The two FBO images of the first pass (glwidget.cpp:200-228) for an analytic camera: UNORM8
 * cube-space positions, alpha 255 where the face is visible, clear colour 0 elsewhere. */
void vvo_first_pass(const vv_ray_source *rs, const struct camera_params *cam, int W, int H,
                    uint8_t *front_rgba, uint8_t *back_rgba)
{
    vv_ray_source q = *rs;
    q.quantize8 = 1;                                  /* end points come back as k/255 */
    for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) {
        f3 f, b;
        int vis_f, vis_b;
        analytic_endpoints_vis(&q, cam, W, H, x, y, &f, &b, &vis_f, &vis_b);
        uint8_t *pf = front_rgba + 4 * ((size_t)y * W + x), *pb = back_rgba + 4 * ((size_t)y * W + x);
        pf[0] = (uint8_t)(f.x * 255.f + 0.5f); pf[1] = (uint8_t)(f.y * 255.f + 0.5f); pf[2] = (uint8_t)(f.z * 255.f + 0.5f);
        pb[0] = (uint8_t)(b.x * 255.f + 0.5f); pb[1] = (uint8_t)(b.y * 255.f + 0.5f); pb[2] = (uint8_t)(b.z * 255.f + 0.5f);
        pf[3] = vis_f ? 255 : 0; pb[3] = vis_b ? 255 : 0;
    }
}

/* kernel.cu:317-321: tex2D point sampling at (x/W, y/H), normalised coordinates */
static void image_endpoints(const vv_ray_source *rs, int W, int H, int x, int y, f3 *front, f3 *back)
{
    float u = (float)x / (float)W, v = (float)y / (float)H;
    int tx = clampi((int)floorf(u * (float)rs->img_w), 0, rs->img_w - 1);
    int ty = clampi((int)floorf(v * (float)rs->img_h), 0, rs->img_h - 1);
    const uint8_t *f = rs->front + 4 * ((size_t)ty * rs->img_w + tx);
    const uint8_t *b = rs->back  + 4 * ((size_t)ty * rs->img_w + tx);
    *front = mk3(f[0] / 255.f, f[1] / 255.f, f[2] / 255.f);
    *back  = mk3(b[0] / 255.f, b[1] / 255.f, b[2] / 255.f);
}

void vvo_ray_endpoints(const vv_ray_source *rs, const struct camera_params *cam,
                       int W, int H, int x, int y, float front[3], float back[3])
{
    f3 f, b;
    if (rs->mode == VV_RAYS_IMAGES) image_endpoints(rs, W, H, x, y, &f, &b);
    else analytic_endpoints(rs, cam, W, H, x, y, &f, &b);
    front[0] = f.x; front[1] = f.y; front[2] = f.z;
    back[0] = b.x; back[1] = b.y; back[2] = b.z;
}

/* ============================================================================
 * implicit.cu
 * ========================================================================== */
/* implicit.cu:4-17 */
static int intersect_plane_ray(f3 p0, f3 n, f3 l0, f3 l, float *t)
{
    float denom = dot3(n, l);
    if ((double)denom > 1e-6) {
        f3 p0l0 = sub3(p0, l0);
        *t = FDIV(dot3(p0l0, n), denom);
        return *t >= 0;
    }
    return 0;
}
/* implicit.cu:20-35 */
static int intersect_sphere_ray(f3 p0, float r, f3 l0, f3 l, float *t)
{
    f3 l0p0 = sub3(l0, p0);
    float b = dot3(l, l0p0);
    float c = dot3(l0p0, l0p0) - r * r;
    float discrim = b * b - c;
    if (discrim >= 0.f) {
        *t = b * -1.f - FSQRT(discrim);
        return (double)*t > -1e-6;
    }
    return 0;
}
/* implicit.cu:38-41 */
static float signed_distance_plane(f3 p0, f3 n, f3 p) { return dot3(n, sub3(p, p0)); }
/* implicit.cu:44-47 */
static float distance_to_plane(f3 p0, f3 n, f3 p) { return fabsf(signed_distance_plane(p0, n, p)); }

/* ============================================================================
 * Ray march
 * ========================================================================== */
#define CACHE_DEPTH 32               /* kernel.cu:24 */
#define CACHE_DEPTH_MINUS_TWOF 30.f  /* kernel.cu:25 */
#define DIRECT_FACTOR 0.3f           /* kernel.cu:27 */
#define ONE_MINUS_DIRECT_FACTOR 0.7f /* kernel.cu:28 */
#define BLOCK_W 16                   /* kernel.cu:30-31 */
#define SQRT_3 1.73205081f           /* kernel.cu:33 */

typedef struct {
    f3 origin, dir;        /* aligned start, unit direction (kernel.cu:340-348) */
    f3 sdir;               /* scaledDirection (kernel.cu:228) */
    float sstep;           /* scaledStep (kernel.cu:229) */
    float upper;           /* after plane-cut clipping */
    float dist0;           /* first chunk's dist (0, or plane hit, kernel.cu:242) */
    int   cut_return;      /* kernel.cu:235-238 early return */
} ray_t;

typedef struct {
    const vvo_volume *vol; const float *tf;
    int W, H; int slice_type; f3 slice_point, slice_normal, scale, inv_scale, step, cam_pos;
    float tan_fov_x, tan_fov_y; int phong; int filter; int ert_true; float ert_thr;
    const vv_ray_source *rays; const struct camera_params *cam;
    int rb, re, band, count, index;   /* slab-row shard predicate, include/volviz.h */
} frame_t;

/* kernel.cu:107-118 blend */
static inline void blend(const float src[4], float dst[4])
{
    float bf = src[3] * (1.f - dst[3]);
    dst[0] = dst[0] + src[0] * bf;
    dst[1] = dst[1] + src[1] * bf;
    dst[2] = dst[2] + src[2] * bf;
    dst[3] = dst[3] + bf;
}

/* ray set-up: kernel.cu:331-350 and the head of mainLoop, kernel.cu:218-246 */
static void setup_ray(const frame_t *F, f3 front, f3 back, float rad, ray_t *r)
{
    f3 dist = sub3(back, front);
    float length = vlen(dist);
    f3 ray = mk3(FDIV(dist.x, length), FDIV(dist.y, length), FDIV(dist.z, length));  /* :340 (NaN if length==0) */
    f3 pos = front;
    float t;
    if (intersect_sphere_ray(F->cam_pos, rad, front, scl3(ray, -1.f), &t))  /* :344 */
        pos = sub3(pos, scl3(ray, t));                                  /* :347 */
    float upper = fminf(SQRT_3, vlen(sub3(back, pos)));                 /* :350 */
    r->origin = pos; r->dir = ray;
    r->sdir = mul3(ray, F->step);                                       /* :228 */
    r->sstep = vlen(r->sdir);                                           /* :229 */
    r->dist0 = 0.f; r->cut_return = 0;
    if (F->slice_type == SLICE_PLANE_CUT) {                             /* :234-246 */
        f3 fr = pos, bk = add3(pos, scl3(ray, upper));
        if ((double)signed_distance_plane(F->slice_point, F->slice_normal, fr) < 1e-6 &&
            (double)signed_distance_plane(F->slice_point, F->slice_normal, bk) < 1e-6) {
            r->cut_return = 1;
        } else {
            if (intersect_plane_ray(F->slice_point, F->slice_normal, pos, ray, &t)) r->dist0 = t;
            else if (intersect_plane_ray(F->slice_point, F->slice_normal, bk, scl3(ray, -1.f), &t)) upper -= t;
        }
    }
    r->upper = upper;
}

/* kernel.cu:125-145 rayMarch: 32 samples from pos, stepping by scaledDirection */
static void ray_march(const frame_t *F, const ray_t *r, float dist, uint8_t out[CACHE_DEPTH])
{
    f3 pos = add3(r->origin, scl3(r->dir, dist));                       /* :249 */
    for (int i = 0; i < CACHE_DEPTH; i++) {
        out[i] = sample_u8(F->vol, to_tex(pos, F->inv_scale), F->filter);   /* :136 */
        pos = add3(pos, r->sdir);                                       /* :141 */
    }
}

/* kernel.cu:73-78 getVoxel depth guard */
static inline uint8_t get_voxel(const uint8_t *cache, int offset)
{
    if (offset < 0 || offset + 1 > CACHE_DEPTH) return 0;
    return cache[offset];
}

/* kernel.cu:147-201 shadeVoxel.  nb[0..3] = left,right,top,bottom neighbour caches. */
static void shade_voxel(const frame_t *F, const uint8_t *self, const uint8_t *const nb[4],
                        int offset, f3 voxel_pos, f3 voxel_dim, float value[4])
{
    uint8_t s = get_voxel(self, offset);
    memcpy(value, F->tf + 4 * (int)s, 4 * sizeof(float));              /* :120-123 tex1Dfetch */
    if (F->phong && (double)value[3] > 1e-6) {                          /* :164 */
        float f = FDIV((float)get_voxel(self, offset - 1), 255.f);                 /* :167 */
        float a = FDIV((float)get_voxel(self, offset + 1), 255.f);                 /* :168 */
        float l = FDIV((float)get_voxel(nb[0], offset), 255.f);                    /* :170 */
        float r = FDIV((float)get_voxel(nb[1], offset), 255.f);                    /* :171 */
        float t = FDIV((float)get_voxel(nb[2], offset), 255.f);                    /* :172 */
        float b = FDIV((float)get_voxel(nb[3], offset), 255.f);                    /* :173 */
        f3 g = mk3(FDIV(r - l, voxel_dim.x), FDIV(t - b, voxel_dim.y), FDIV(a - f, voxel_dim.z)); /* :175-178 */
        if (g.x != 0.f && g.y != 0.f && g.z != 0.f) {                   /* :180-181 */
            float inv = FRSQRT(dot3(g, g));                             /* helper_math.h:1309-1312 rsqrtf */
            g = scl3(g, inv);
        }
        float direct = dot3(g, mk3(-1.f, -1.f, 1.f)) * DIRECT_FACTOR;   /* :183 */
        direct = clampf(direct, 0.f, DIRECT_FACTOR);                    /* :184 */
        value[0] = value[0] * ONE_MINUS_DIRECT_FACTOR + direct;         /* :186-190 */
        value[1] = value[1] * ONE_MINUS_DIRECT_FACTOR + direct;
        value[2] = value[2] * ONE_MINUS_DIRECT_FACTOR + direct;
    }
    if (F->slice_type == SLICE_PLANE) {                                 /* :193-198 */
        float d = distance_to_plane(F->slice_point, F->slice_normal, voxel_pos);
        if (d < .01f) value[0] = clampf(value[0] + (.01f - d) * 100.f, 0.f, 1.f);
    }
}

/* The body of the while loop of mainLoop for one chunk (kernel.cu:253-277) given the
 * caches.  Returns the number of executed samples.  *ert latches for VV_ERT_TRUE. */
static unsigned shade_chunk(const frame_t *F, const ray_t *r, float dist, const uint8_t *self,
                            const uint8_t *const nb[4], float result[4], int *ert)
{
    unsigned n = 0;
    for (int i = 1; i < CACHE_DEPTH - 1; ++i) {
        float voxel_dist = (float)i * r->sstep + dist;                  /* :254 */
        if (voxel_dist > r->upper) break;                               /* :255-257 */
        f3 voxel_dim = mk3(F->tan_fov_x * voxel_dist, F->tan_fov_y * voxel_dist, r->sstep * 2.f); /* :259-263 */
        f3 voxel_pos = add3(r->origin, scl3(r->dir, voxel_dist));       /* :264 */
        float shaded[4];
        shade_voxel(F, self, nb, i, voxel_pos, voxel_dim, shaded);      /* :266 */
        n++;
        if ((double)shaded[3] > 1e-6) blend(shaded, result);            /* :268-270 */
        if (result[3] > F->ert_thr) { *ert = 1; break; }                /* :272-274 */
    }
    return n;
}

/* One 16x16 thread block of kernel<> (kernel.cu:281-367), block (bx,by). */
static unsigned long long render_block(const frame_t *F, int bx, int by, uint8_t *rgba)
{
    const int W = F->W, H = F->H;
    /* kernel.cu:297-302 */
    int upx = (bx + 1) * (BLOCK_W - 2) + 1; if (upx > W - 1) upx = W - 1;
    int upy = (by + 1) * (BLOCK_W - 2) + 1; if (upy > H - 1) upy = H - 1;
    int lox = bx * (BLOCK_W - 2) - 1; if (lox < 0) lox = 0;
    int loy = by * (BLOCK_W - 2) - 1; if (loy < 0) loy = 0;
    int sw = upx - lox, sh = upy - loy;

    int px[BLOCK_W * BLOCK_W], py[BLOCK_W * BLOCK_W];
    f3 front[BLOCK_W * BLOCK_W], back[BLOCK_W * BLOCK_W];
    float cam_len[BLOCK_W * BLOCK_W];
    unsigned long long executed = 0;

    for (int ty = 0; ty < BLOCK_W; ty++) for (int tx = 0; tx < BLOCK_W; tx++) {
        int t = ty * BLOCK_W + tx;
        int x = bx * (BLOCK_W - 2) + (tx - 1), y = by * (BLOCK_W - 2) + (ty - 1);  /* :294-295 */
        x = clampi(x, lox, upx - 1); y = clampi(y, loy, upy - 1);                   /* :307-308 */
        px[t] = x; py[t] = y;
        float f[3], b[3];
        vvo_ray_endpoints(F->rays, F->cam, W, H, x, y, f, b);                     /* :317-321 */
        front[t] = mk3(f[0], f[1], f[2]); back[t] = mk3(b[0], b[1], b[2]);
        cam_len[t] = vlen(sub3(front[t], F->cam_pos));                             /* :323-325 */
    }
    /* kernel.cu:80-97,329 blockMin over slabUpper = sw*sh entries.  Every footprint
     * pixel is some thread's (x,y), so this is the min over the footprint; with a
     * degenerate footprint (sw*sh <= 0) the loop is empty and rad = own value. */
    int degenerate = (sw <= 0 || sh <= 0);
    float rad_all = INFINITY;
    if (!degenerate)
        for (int t = 0; t < BLOCK_W * BLOCK_W; t++) if (cam_len[t] < rad_all) rad_all = cam_len[t];

    ray_t rays[BLOCK_W * BLOCK_W];
    int   skip[BLOCK_W * BLOCK_W];      /* interior zero-length early-out (kernel.cu:334-338) */
    for (int t = 0; t < BLOCK_W * BLOCK_W; t++) {
        int tx = t % BLOCK_W, ty = t / BLOCK_W;
        int border = tx == 0 || ty == 0 || tx + 1 == BLOCK_W || ty + 1 == BLOCK_W;  /* :304-305 */
        float length = vlen(sub3(back[t], front[t]));
        skip[t] = (length < 0.001f && !border);
        setup_ray(F, front[t], back[t], degenerate ? cam_len[t] : rad_all, &rays[t]);
    }

    /* neighbour thread of thread t in direction d, clamped to the slab footprint
     * (ORACLE PIN 6: cacheIdx +-1 clamps to the footprint instead of wrapping). */
    #define THREAD_AT(fx, fy) (((fy) - (by * (BLOCK_W - 2) - 1)) * BLOCK_W + ((fx) - (bx * (BLOCK_W - 2) - 1)))

    /* ORACLE PIN 10 (write ownership): several interior threads can be clamped onto one
     * pixel (identical work, identical value), and when W or H == 1 (mod 14) the last
     * block's interior threads all clamp onto a pixel the previous block also writes
     * (a write race in the reference).  Pinned: the block with the higher index wins.
     * The oracle therefore lets a block write a pixel only if it is the pixel's owner,
     * and marches each owned pixel once. */
    int nbx = W / (BLOCK_W - 2) + ((W % (BLOCK_W - 2)) ? 1 : 0);
    int nby = H / (BLOCK_W - 2) + ((H % (BLOCK_W - 2)) ? 1 : 0);
    uint8_t seen[BLOCK_W * BLOCK_W];
    memset(seen, 0, sizeof seen);
    for (int ty = 1; ty < BLOCK_W - 1; ty++) for (int tx = 1; tx < BLOCK_W - 1; tx++) {
        int t = ty * BLOCK_W + tx;
        int ox = (W >= 2 && W - 1 == (nbx - 1) * (BLOCK_W - 2) && px[t] == W - 2) ? nbx - 1 : px[t] / (BLOCK_W - 2);
        int oy = (H >= 2 && H - 1 == (nby - 1) * (BLOCK_W - 2) && py[t] == H - 2) ? nby - 1 : py[t] / (BLOCK_W - 2);
        if (ox != bx || oy != by) continue;
        {   /* shard predicate on the pixel row's geometric slab row (include/volviz.h) */
            int r = py[t] / (BLOCK_W - 2);
            if (!(r >= F->rb && r < F->re && ((r / F->band) % F->count) == F->index)) continue;
        }
        int key = (py[t] - (by * (BLOCK_W - 2) - 1)) * BLOCK_W + (px[t] - (bx * (BLOCK_W - 2) - 1));
        if (seen[key]) continue;
        seen[key] = 1;
        uint8_t *pix = rgba + 4 * ((size_t)py[t] * W + px[t]);
        if (skip[t]) { pix[0] = pix[1] = pix[2] = pix[3] = 0; continue; }           /* :336 */
        const ray_t *r = &rays[t];
        float result[4] = {0.f, 0.f, 0.f, 0.f};                                    /* :219 */
        if (!r->cut_return) {
            /* neighbour rays (ORACLE PIN 5: neighbour cache entries are always the
             * neighbour ray's own chunk-c samples, re-sampled, never stale) */
            const ray_t *nbr[4] = {r, r, r, r};
            if (F->phong && !degenerate) {
                int x = px[t], y = py[t];
                int fxl = clampi(x - 1, lox, upx - 1), fxr = clampi(x + 1, lox, upx - 1);
                int fyt = clampi(y + 1, loy, upy - 1), fyb = clampi(y - 1, loy, upy - 1);
                nbr[0] = &rays[THREAD_AT(fxl, y)]; nbr[1] = &rays[THREAD_AT(fxr, y)];
                nbr[2] = &rays[THREAD_AT(x, fyt)]; nbr[3] = &rays[THREAD_AT(x, fyb)];
            }
            float dist = r->dist0;
            float nd[4] = {nbr[0]->dist0, nbr[1]->dist0, nbr[2]->dist0, nbr[3]->dist0};
            int ert = 0;
            while (dist < r->upper) {                                              /* :248 */
                uint8_t self[CACHE_DEPTH], nbc[4][CACHE_DEPTH];
                const uint8_t *nbp[4] = {self, self, self, self};
                ray_march(F, r, dist, self);                                       /* :251 */
                if (F->phong) for (int d = 0; d < 4; d++) {
                    ray_march(F, nbr[d], nd[d], nbc[d]); nbp[d] = nbc[d];
                    nd[d] += nbr[d]->sstep * CACHE_DEPTH_MINUS_TWOF;
                }
                executed += shade_chunk(F, r, dist, self, nbp, result, &ert);
                if (ert && F->ert_true) break;
                dist += r->sstep * CACHE_DEPTH_MINUS_TWOF;                         /* :277 */
            }
        }
        /* kernel.cu:359-366 */
        for (int c = 0; c < 4; c++) pix[c] = sat_u8(clampf(result[c], 0.f, 1.f) * 255.0f);
    }
    #undef THREAD_AT
    return executed;
}

unsigned long long vvo_render(const vvo_volume *v, const float tf[1024], int W, int H,
                              const struct slice_params *slice,
                              const struct camera_params *cam,
                              const struct shading_params *shading,
                              const vv_ray_source *rays,
                              const vv_render_options *opts,
                              uint8_t *rgba, int threads)
{
    frame_t F;
    memset(&F, 0, sizeof F);
    F.vol = v; F.tf = tf; F.W = W; F.H = H; F.rays = rays; F.cam = cam;
    F.slice_type = slice->type;
    F.slice_point  = mk3(slice->params[0], slice->params[1], slice->params[2]);   /* kernel.cu:224 */
    F.slice_normal = mk3(slice->params[3], slice->params[4], slice->params[5]);   /* :225 */
    F.scale = mk3(cam->scale[0], cam->scale[1], cam->scale[2]);                  /* :226 */
    F.inv_scale = inv3(F.scale);
    F.cam_pos = mk3(cam->origin[0], cam->origin[1], cam->origin[2]);             /* :323 */
    F.step = mk3(1.f / (float)v->nx, 1.f / (float)v->ny, 1.f / (float)v->nz);    /* :415 */
    F.filter = VV_FILTER_TEX8; F.ert_true = 0; F.ert_thr = .95f;
    int rb = 0, re = 0;
    F.band = 4; F.count = 1; F.index = 0;
    if (opts) {
        if (opts->shard_count > 1) { F.count = opts->shard_count; F.index = opts->shard_index; F.band = opts->shard_band; }
        if (opts->step[0] > 0.f || opts->step[1] > 0.f || opts->step[2] > 0.f)
            F.step = mk3(opts->step[0], opts->step[1], opts->step[2]);
        if (opts->ert_threshold > 0.f) F.ert_thr = opts->ert_threshold;
        F.filter = opts->filter; F.ert_true = opts->ert_mode == VV_ERT_TRUE;
        rb = opts->slab_row_begin; re = opts->slab_row_end;
    }
    /* kernel.cu:221-222: float * double / float -> double, tan in double, narrowed */
    F.tan_fov_x = (float)tan((double)cam->fovX * M_PI / (double)(180.f * (float)(unsigned)W));
    F.tan_fov_y = (float)tan((double)cam->fovY * M_PI / (double)(180.f * (float)(unsigned)H));
    F.phong = shading->phongShading ? 1 : 0;

    /* kernel.cu:418-425 launch geometry */
    int nbx = W / (BLOCK_W - 2) + ((W % (BLOCK_W - 2)) ? 1 : 0);
    int nby = H / (BLOCK_W - 2) + ((H % (BLOCK_W - 2)) ? 1 : 0);
    if (rb == 0 && re == 0) re = nby;
    if (re > nby) re = nby;
    F.rb = rb; F.re = re;
    unsigned long long executed = 0;
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#else
    (void)threads;
#endif
#ifdef _OPENMP
    #pragma omp parallel for schedule(dynamic, 4) num_threads(threads) reduction(+:executed)
#endif
    for (long b = 0; b < (long)nby * nbx; b++) {
#if VVO_MODEL == 2
        const unsigned csr = _mm_getcsr();
        _mm_setcsr(csr | 0x8040u);               /* FTZ + DAZ, as -use_fast_math's -ftz=true */
#endif
        executed += render_block(&F, (int)(b % nbx), (int)(b / nbx), rgba);
#if VVO_MODEL == 2
        _mm_setcsr(csr);
#endif
    }
    return executed;
}

/* ============================================================================
 * Synthetic noise volume V2 -- not from the reference.
 * raw(x,y,z) = mix32((x + nx*(y + ny*z)) ^ seed) >> 24 ; out = 3x3x3 box mean
 * with clamped neighbours, rounded to nearest.
 * ========================================================================== */
static inline uint32_t mix32(uint32_t h)
{
    h ^= h >> 16; h *= 0x7feb352dU; h ^= h >> 15; h *= 0x846ca68bU; h ^= h >> 16;
    return h;
}
void vvo_generate_noise_u8(uint8_t *out, int nx, int ny, int nz, uint32_t seed)
{
#ifdef _OPENMP
    #pragma omp parallel for schedule(static)
#endif
    for (int z = 0; z < nz; z++)
        for (int y = 0; y < ny; y++)
            for (int x = 0; x < nx; x++) {
                uint32_t sum = 0;
                for (int dz = -1; dz <= 1; dz++) for (int dy = -1; dy <= 1; dy++) for (int dx = -1; dx <= 1; dx++) {
                    int xx = clampi(x + dx, 0, nx - 1), yy = clampi(y + dy, 0, ny - 1), zz = clampi(z + dz, 0, nz - 1);
                    uint32_t idx = (uint32_t)xx + (uint32_t)nx * ((uint32_t)yy + (uint32_t)ny * (uint32_t)zz);
                    sum += mix32(idx ^ seed) >> 24;
                }
                out[((size_t)z * ny + y) * nx + x] = (uint8_t)((sum + 13u) / 27u);
            }
}
