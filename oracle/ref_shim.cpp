/*
 * ref_shim.cpp -- C entry points over the REAL reference sources (TEST INFRASTRUCTURE).
 *
 * This file contains no reference code.  oracle/Makefile compiles the reference's own
 * volumegenerator.cpp and cs123math/CS123Matrix.cpp in place (from $(REF), normally
 * /root/reference) and links them with this shim into oracle/_ref/libvvref.so, which
 * is git-ignored.  The shim only calls the reference's public interface
 * (volumegenerator.h:23-57, cs123math/CS123Algebra.h:496-506) so that tests can pin the
 * oracle's restatement to the reference's actual output and generate golden fixtures.
 */
#include "volumegenerator.h"
#include "cs123math/CS123Algebra.h"

#include <cstdio>
#include <cstring>
#include <unistd.h>
#include <fcntl.h>

namespace {
/* the reference prints progress to stdout; keep test logs quiet */
struct QuietStdout {
    int saved;
    QuietStdout() {
        fflush(stdout);
        saved = dup(1);
        int nul = open("/dev/null", O_WRONLY);
        dup2(nul, 1);
        close(nul);
    }
    ~QuietStdout() {
        fflush(stdout);
        std::cout.flush();
        dup2(saved, 1);
        close(saved);
    }
};
}

extern "C" {

/* VolumeGenerator(x,y,z) + drawDefaultBrain()  (volumegenerator.cpp:12-23,100-119) */
void ref_default_brain(unsigned char *out, int nx, int ny, int nz)
{
    QuietStdout q;
    VolumeGenerator g(nx, ny, nz);
    g.drawDefaultBrain();
    size_t size;
    byte *p = g.getBytes(size);
    memcpy(out, p, size);
}

/* VolumeGenerator(x,y,z) + n x drawEllipsoid()  (volumegenerator.cpp:31-97) */
void ref_draw_ellipsoids(unsigned char *out, int nx, int ny, int nz, int n,
                         const float *centers, const float *axes, const unsigned char *colors)
{
    QuietStdout q;
    VolumeGenerator g(nx, ny, nz);
    for (int e = 0; e < n; e++) {
        Point3 c(centers[3*e], centers[3*e+1], centers[3*e+2]);
        Vector3 a(axes[3*e], axes[3*e+1], axes[3*e+2]);
        g.drawEllipsoid(c, a, colors[e]);
    }
    size_t size;
    byte *p = g.getBytes(size);
    memcpy(out, p, size);
}

/* drawDefaultBrain + saveas_raw(path, header)  (volumegenerator.cpp:147-174) */
void ref_save_default_brain(const char *path, int header, int nx, int ny, int nz)
{
    QuietStdout q;
    VolumeGenerator g(nx, ny, nz);
    g.drawDefaultBrain();
    g.saveas_raw(const_cast<char *>(path), header != 0);
}

/* loadfrom_raw(path, header) -> dims + bytes  (volumegenerator.cpp:176-220) */
long ref_load_raw(const char *path, int header, unsigned char *out, long capacity, int dims[3])
{
    QuietStdout q;
    VolumeGenerator g(0, 0, 0);
    g.loadfrom_raw(path, header != 0);
    Vector3 d = g.getDims();
    dims[0] = (int)d.x; dims[1] = (int)d.y; dims[2] = (int)d.z;
    size_t size;
    byte *p = g.getBytes(size);
    if ((long)size <= capacity) memcpy(out, p, size);
    return (long)size;
}

/* The composition order of SliceWidget::getTransformationMatrix (slicewidget.cpp:156-162)
 * applied to the reference's own matrix builders and operator* (CS123Matrix.cpp:27-62,
 * CS123Algebra.h:429-471). */
void ref_slice_matrix(float dx, float dy, float dz, float theta, float phi, float psi, float out[16])
{
    /* M = T(+.5) * T(d) * Rx(theta) * Ry(phi) * Rz(psi) * T(-.5), multiplied left to right */
    Matrix4x4 m = getTransMat(Vector4(0.5, 0.5, 0.5, 0));
    m = m * getTransMat(Vector4(dx, dy, dz, 1.0));
    m = m * getRotXMat(theta);
    m = m * getRotYMat(phi);
    m = m * getRotZMat(psi);
    m = m * getTransMat(Vector4(-0.5, -0.5, -0.5, 0));
    memcpy(out, m.data, 16 * sizeof(float));
}

/* The cutting plane Window::renderSlice hands GLWidget::setSlicePro for the free-form slice view (window.cpp:425-441), composed with the
 * reference's own builders and operators exactly as that code writes it. */
void ref_cut_plane_pro(float dx, float dy, float dz, float theta, float phi, float psi, float point[3], float normal[3])
{
    Vector4 offset = Vector4(dx, dy, dz, 0);
    offset += Vector4(0.5, 0.5, 0.5, 0);
    Matrix4x4 trans = getTransMat(Vector4(0.5, 0.5, 0.5, 1.0));
    Matrix4x4 rotX = getRotXMat(theta);
    Matrix4x4 rotY = getRotYMat(phi);
    Matrix4x4 rotZ = getRotZMat(psi);
    Matrix4x4 transBack = getTransMat(Vector4(-0.5, -0.5, -0.5, 1.0));
    Vector4 n = trans * rotX * rotY * rotZ * transBack * Vector4(0, 0, 1, 0);
    point[0] = offset.x; point[1] = offset.y; point[2] = offset.z;
    normal[0] = n.x; normal[1] = n.y; normal[2] = n.z;
}

}
