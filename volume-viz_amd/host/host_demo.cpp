// host_demo.cpp -- drives the mirror exactly the way the reference's host code drives
// kernel.cuh: GLWidget::loadVolume (glwidget.cpp:668-710), GLWidget::paintGL (:232-292),
// SliceWidget::renderSlice (slicewidget.cpp:77-106).  Writes raw outputs that
// tests/test_host_mirror.py compares with the oracle.
//   host_demo <out_dir> <front.rgba> <back.rgba> <fbo_w> <fbo_h> <w> <h>
#include "kernel_hip.h"
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

static std::vector<unsigned char> slurp(const char *p, size_t n)
{
    std::vector<unsigned char> v(n);
    FILE *f = fopen(p, "rb");
    if (!f || fread(v.data(), 1, n, f) != n) { fprintf(stderr, "cannot read %s\n", p); exit(2); }
    fclose(f);
    return v;
}
static void dump(const std::string &p, const void *d, size_t n)
{
    FILE *f = fopen(p.c_str(), "wb");
    fwrite(d, 1, n, f);
    fclose(f);
}

int main(int argc, char **argv)
{
    if (argc != 8) return 2;
    const std::string out = argv[1];
    const int fw = atoi(argv[4]), fh = atoi(argv[5]), w = atoi(argv[6]), h = atoi(argv[7]);
    std::vector<unsigned char> front = slurp(argv[2], (size_t)fw * fh * 4), back = slurp(argv[3], (size_t)fw * fh * 4);

    initCuda();                                                     // glwidget.cpp:182
    VolumeGenerator gen(48, 40, 56);
    gen.drawDefaultBrain();                                         // volumegenerator.cpp:100
    gen.saveas_raw(const_cast<char *>((out + "/demo.t3d").c_str()), true);
    VolumeGenerator *volgen = new VolumeGenerator(0, 0, 0);         // glwidget.cpp:674
    volgen->loadfrom_raw((out + "/demo.t3d").c_str(), true);        // :692
    size_t size;
    byte *texels = volgen->getBytes(size);
    float tf[1024];
    vv_transfer_preset(VV_TF_ENGINE, tf);                           // g_transferEngine, :679
    void *volumeArray = 0;
    cudaLoadVolume(texels, size, volgen->getDims(), tf, &volumeArray);   // :700
    dump(out + "/volume.u8", texels, size);
    delete volgen;                                                  // :703

    std::vector<unsigned char> pixels((size_t)w * h * 4, 0x5A);
    registerHostResources(front.data(), back.data(), fw, fh, pixels.data());   // :390 (GL names there)
    struct slice_params sp; sp.type = SLICE_PLANE;                  // :232-256
    sp.params[0] = 0.5f; sp.params[1] = 0.5f; sp.params[2] = 0.45f; sp.params[3] = 0.2f; sp.params[4] = -0.3f; sp.params[5] = 0.93f;
    struct camera_params cp;                                        // :262-276
    cp.scale[0] = 1.f; cp.scale[1] = 1.f; cp.scale[2] = 0.8f;
    cp.origin[0] = 3.2360680f * 0.8660254f; cp.origin[1] = 2.0f; cp.origin[2] = 2.3511410f * 0.8660254f;
    cp.fovY = 45.f; cp.fovX = 45.f * ((float)w / (float)h);
    struct shading_params sh; sh.transferPreset = TRANSFER_PRESET_DEFAULT; sh.phongShading = true;
    runCuda(w, h, sp, cp, sh, volumeArray);                         // :291
    dump(out + "/frame.rgba", pixels.data(), pixels.size());

    std::vector<float> sl(256 * 256);                               // slicewidget.cpp:77-106
    invoke_slice_kernel(sl.data(), BufferParameters(256, 256), SliceParameters(0.05f, 0.4f, 0.3f), CORONAL, make_float3(1.f, 1.f, 0.8f));
    dump(out + "/slice_coronal.f32", sl.data(), sl.size() * 4);
    SliceParameters fp(0.1f, -0.05f, 0.02f, 0.4f, -0.3f, 0.2f);
    invoke_advanced_slice_kernel(sl.data(), BufferParameters(256, 256), getTransformationMatrix(fp), make_float3(1.f, 1.f, 0.8f));
    dump(out + "/slice_free.f32", sl.data(), sl.size() * 4);
    printf("host_demo ok\n");
    return 0;
}
