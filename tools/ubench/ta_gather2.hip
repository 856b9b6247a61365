// Micro-benchmark 2: what a wave-wide gather costs in the vector memory pipeline (TA/L1) as a
// function of (a) element width, (b) lane spacing, (c) how many lanes are active.
// Data is L1 resident (every block reads the same few dozen lines).  Build: hipcc -O3 --offload-arch=gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float __attribute__((ext_vector_type(2), aligned(4))) f2u;
typedef float __attribute__((ext_vector_type(4), aligned(4))) f4u;

// MASK: 0 all lanes, 1 even lanes, 2 lane%6==0, 3 lane 0 only, 4 lanes 0..15, 5 lanes with (lane&3)==0
template <int MODE, int MASK>
__global__ __launch_bounds__(256) void k(const char* __restrict__ buf, int iters, int rowstride, float spacing, int tw,
                                         float* out, int footprint)
{
    int lane = threadIdx.x & 63;
    int lx = lane & ((1 << tw) - 1), ly = lane >> tw;
    const char* base = buf;   // one small window for everybody: L1 resident
    float acc = 0.f;
    unsigned off0 = ((unsigned)(ly * rowstride) + (unsigned)(lx * spacing));
    bool active = MASK == 0 ? true : MASK == 1 ? !(lane & 1) : MASK == 2 ? (lane % 6 == 0) : MASK == 3 ? lane == 0
                : MASK == 4 ? lane < 16 : (lane & 3) == 0;
    // 8 independent loads in flight per lane and trip, so the figure is throughput, not latency
    for (int it = 0; it < iters; ++it) {
        unsigned e = (off0 + (unsigned)(it & 7) * 3u);
        float part[8];
        if (active) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                unsigned ej = e + (unsigned)j * 2048u;          // 8 KB apart: different lines, same pattern
                if (MODE == 0) { part[j] = *(const float*)(base + ej * 4u); }
                if (MODE == 1) { f2u v = *(const f2u*)(base + ej * 4u); part[j] = v.x + v.y; }
                if (MODE == 2) { f4u v = *(const f4u*)(base + ej * 4u); part[j] = v.x + v.y + v.z + v.w; }
                if (MODE == 3) { part[j] = *(const float*)(base + ej * 4u) + *(const float*)(base + ej * 4u + 4u); }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += part[j];
        }
        asm volatile("" : "+v"(acc));
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int MODE, int MASK>
static float run(const char* buf, int iters, int rs, float sp, int tw, float* out, int fp, int blocks)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(a);
        k<MODE, MASK><<<blocks, 256>>>(buf, iters, rs, sp, tw, out, fp);
        hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
    }
    hipEventDestroy(a); hipEventDestroy(b);
    return ms;
}

int main()
{
    int footprint = 64 * 1024;
    char* buf; hipMalloc(&buf, 64 * footprint + 4096); hipMemset(buf, 0, 64 * footprint + 4096);
    float* out; hipMalloc(&out, 2048 * 256 * 4);
    int iters = 1024, blocks = 2048;
    const char* mode_n[] = {"dword", "dwordx2", "dwordx4", "2 x dword"};
    const char* mask_n[] = {"all 64", "even lanes", "lane%6==0", "lane 0", "lanes 0-15", "lane%4==0"};
    struct Pat { const char* n; int tw; int rs; float sp; } pats[] = {
        {"64x1 s=1.00", 6, 0, 1.0f}, {"64x1 s=1.19", 6, 0, 1.19f}, {"32x2 s=1.00 rs=1024", 5, 1024, 1.0f},
        {"32x2 s=1.19 rs=1024", 5, 1024, 1.19f}, {"32x2 s=2.00 rs=1024", 5, 1024, 2.0f},
        {"16x4 s=1.19 rs=1024", 4, 1024, 1.19f}, {"8x8 s=1.19 rs=1024", 3, 1024, 1.19f}};
    double winstr = (double)blocks * 4 * iters * 8;
    auto cyc = [&](float ms) { return ms * 1e-3 * 2.4e9 / (winstr / 256.0); };
    printf("cycles per wave-instruction per CU at 2.4 GHz (loop overhead included)\n");
    for (auto& p : pats) {
        printf("pattern %s\n", p.n);
#define ROW(MODE) \
        printf("  %-10s", mode_n[MODE]); \
        printf(" all %5.1f", cyc(run<MODE, 0>(buf, iters, p.rs, p.sp, p.tw, out, footprint, blocks))); \
        printf(" | even %5.1f", cyc(run<MODE, 1>(buf, iters, p.rs, p.sp, p.tw, out, footprint, blocks))); \
        printf(" | %%6 %5.1f", cyc(run<MODE, 2>(buf, iters, p.rs, p.sp, p.tw, out, footprint, blocks))); \
        printf(" | lane0 %5.1f", cyc(run<MODE, 3>(buf, iters, p.rs, p.sp, p.tw, out, footprint, blocks))); \
        printf(" | 0-15 %5.1f", cyc(run<MODE, 4>(buf, iters, p.rs, p.sp, p.tw, out, footprint, blocks))); \
        printf(" | %%4 %5.1f\n", cyc(run<MODE, 5>(buf, iters, p.rs, p.sp, p.tw, out, footprint, blocks)));
        ROW(0) ROW(1) ROW(2) ROW(3)
    }
    (void)mask_n;
    return 0;
}
