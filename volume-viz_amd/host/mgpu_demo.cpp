// mgpu_demo.cpp -- renders one frame on every visible GPU through the native multi-GPU entry points
// (include/volviz_mgpu.h) and checks it against the same frame from one GPU.  Exit code 0 = identical.
//   mgpu_demo [n_gpus] [volume_edge] [width] [height]
#include "../../include/volviz_mgpu.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

int main(int argc, char **argv)
{
    int n = argc > 1 ? atoi(argv[1]) : 0;
    const int edge = argc > 2 ? atoi(argv[2]) : 128, W = argc > 3 ? atoi(argv[3]) : 640, H = argc > 4 ? atoi(argv[4]) : 360;
    vv_mgpu *m = nullptr;
    if (n <= 0) { n = 8; while (n > 1 && vv_mgpu_init(n, nullptr, &m) != VV_OK) --n; if (!m && vv_mgpu_init(1, nullptr, &m) != VV_OK) { fprintf(stderr, "no device: %s\n", vv_mgpu_last_error(nullptr)); return 2; } }
    else if (vv_mgpu_init(n, nullptr, &m) != VV_OK) { fprintf(stderr, "vv_mgpu_init(%d): %s\n", n, vv_mgpu_last_error(nullptr)); return 2; }
    n = vv_mgpu_size(m);
    float tf[1024];
    vv_transfer_preset(0 /* Engine */, tf);
    if (vv_mgpu_generate_default_brain(m, VV_VOXEL_F32, edge, edge, edge, tf) != VV_OK) { fprintf(stderr, "volume: %s\n", vv_mgpu_last_error(m)); return 2; }
    slice_params sp; memset(&sp, 0, sizeof sp); sp.type = SLICE_NONE;
    camera_params cp; memset(&cp, 0, sizeof cp);
    cp.origin[0] = 1.2f; cp.origin[1] = 0.8f; cp.origin[2] = -3.6f; cp.fovY = 45.f; cp.fovX = 45.f * W / H; cp.scale[0] = cp.scale[1] = cp.scale[2] = 1.f;
    shading_params sh; memset(&sh, 0, sizeof sh); sh.transferPreset = -1; sh.phongShading = false;
    vv_ray_source rs; memset(&rs, 0, sizeof rs);
    rs.mode = VV_RAYS_ANALYTIC; rs.look[0] = -cp.origin[0]; rs.look[1] = -cp.origin[1]; rs.look[2] = -cp.origin[2]; rs.up[1] = 1.f;
    std::vector<uint8_t> multi((size_t)W * H * 4, 0x5A), single((size_t)W * H * 4, 0x5A);
    for (int k = 0; k < 3; ++k)
        if (vv_mgpu_render(m, W, H, &sp, &cp, &sh, &rs, nullptr, multi.data(), 0) != VV_OK) { fprintf(stderr, "render: %s\n", vv_mgpu_last_error(m)); return 2; }
    std::vector<float> ms(n); float gms = 0.f;
    vv_mgpu_last_times(m, ms.data(), &gms);
    if (vv_render(vv_mgpu_context(m, 0), W, H, &sp, &cp, &sh, &rs, nullptr, single.data(), 0, nullptr) != VV_OK) { fprintf(stderr, "single: %s\n", vv_last_error(vv_mgpu_context(m, 0))); return 2; }
    size_t diff = 0, lit = 0;
    for (size_t i = 0; i < multi.size(); ++i) { diff += multi[i] != single[i]; lit += (i % 4 == 3) && multi[i] != 0 && multi[i] != 0x5A; }
    printf("mgpu_demo: %d GPU(s), %d^3 f32 volume, %dx%d: %zu differing bytes, %zu lit pixels; march ms per rank:", n, edge, W, H, diff, lit);
    for (int r = 0; r < n; ++r) printf(" %.3f", ms[r]);
    printf("; gather %.3f ms\n", gms);
    vv_mgpu_shutdown(m);
    return diff == 0 && lit > 0 ? 0 : 1;
}
