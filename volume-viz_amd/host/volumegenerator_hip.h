// volumegenerator_hip.h -- VolumeGenerator with the reference's public interface
// (volumegenerator.h:23-57), drawing on the MI355X through the C-ABI generator kernels.
#pragma once
#include <cstddef>
#include <string>

struct Vector3 {                                    // cs123math vec3<float>: x, y, z
    Vector3() : x(0), y(0), z(0) {}
    Vector3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    float x, y, z;
};
typedef Vector3 Point3;                             // volumegenerator.h:19

class VolumeGenerator {
public:
    typedef unsigned char byte;
    VolumeGenerator(int x, int y, int z);           // zero-filled volume, volumegenerator.cpp:12-23
    ~VolumeGenerator();

    void drawEllipsoid(const Point3 &center, const Vector3 &axes, const byte &color);   // :31-97
    void drawDefaultBrain();                                                              // :100-119

    // (volume2csv / saveas_csv, volumegenerator.cpp:122-145: a debugging dump outside the hot path, not mirrored)
    void saveas_raw(char *dest, bool header = false);            // :147-174
    void loadfrom_raw(const char *source, bool header = false);  // :176-220

    byte *getBytes(size_t &size);                   // :222-226
    size_t getVolSize();                            // :228-231
    Vector3 getDims();                              // :233-236

private:
    byte *m_volume;
    int m_x, m_y, m_z;
};
