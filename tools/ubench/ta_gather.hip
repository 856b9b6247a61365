// Micro-benchmark: cost of one wave-wide gather instruction in the vector memory pipeline (TA/L1)
// for the access shapes a trilinear sampler can use.  Data is L1/L2-resident (small footprint).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
typedef float __attribute__((ext_vector_type(2), aligned(4))) f2u;
typedef float __attribute__((ext_vector_type(2))) f2a;
typedef float __attribute__((ext_vector_type(4))) f4a;
typedef float __attribute__((ext_vector_type(4), aligned(4))) f4u;
typedef unsigned short __attribute__((aligned(1))) us_u;
typedef unsigned __attribute__((aligned(1))) u32_u;

template <int MODE>
__global__ __launch_bounds__(256) void k(const char* __restrict__ buf, int iters, int rowstride, float spacing, int tw, float* out, int footprint)
{
    int lane = threadIdx.x & 63;
    int lx = lane & ((1 << tw) - 1), ly = lane >> tw;
    // per-block private window to stay cache resident
    const char* base = buf + (size_t)(blockIdx.x % 64) * footprint;
    float acc = 0.f;
    unsigned off0 = ((unsigned)(ly * rowstride) + (unsigned)(lx * spacing)) ;
    for (int it = 0; it < iters; ++it) {
        unsigned e = (off0 + (unsigned)(it & 7) * 3u);   // element index, drifts a little
        if (MODE == 0) { acc += *(const float*)(base + e * 4u); }
        if (MODE == 1) { f2u v = *(const f2u*)(base + e * 4u); acc += v.x + v.y; }                 // unaligned pair
        if (MODE == 2) { f2a v = *(const f2a*)(base + (e & ~1u) * 4u); acc += v.x + v.y; }       // aligned pair
        if (MODE == 3) { f4a v = *(const f4a*)(base + (e & ~3u) * 4u); acc += v.x + v.y + v.z + v.w; }   // aligned quad
        if (MODE == 4) { f4u v = *(const f4u*)(base + e * 4u); acc += v.x + v.y + v.z + v.w; }   // unaligned quad
        if (MODE == 5) { acc += (float)*(const us_u*)(base + e); }                                 // u8 pair (ushort)
        if (MODE == 6) { acc += (float)*(const unsigned char*)(base + e); }
        if (MODE == 7) { acc += (float)*(const unsigned*)(base + (e & ~3u)); }                     // aligned dword of u8 (4 voxels)
        if (MODE == 8) { acc += (float)*(const u32_u*)(base + e); }                                // byte-aligned dword (u8 x..x+3)
        asm volatile("" : "+v"(acc));
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main(int argc, char** argv)
{
    int footprint = 64 * 1024;
    char* buf; hipMalloc(&buf, 64 * footprint + 4096); hipMemset(buf, 0, 64 * footprint + 4096);
    float* out; hipMalloc(&out, 2048 * 256 * 4);
    const char* names[] = {"dword", "dwordx2 unaligned", "dwordx2 aligned", "dwordx4 aligned", "dwordx4 unaligned", "ushort", "ubyte", "dword(u8x4) aligned", "dword byte-aligned"};
    int iters = 4096, blocks = 2048;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    struct Pat { const char* n; int tw; int rowstride; float spacing; } pats[] = {
        {"coalesced 64x1 s=1", 6, 0, 1.0f}, {"tile 8x8 s=1.5 rs=1024", 3, 1024, 1.5f}, {"tile 32x2 s=1.5 rs=1024", 5, 1024, 1.5f},
        {"tile 8x8 s=1.5 rs=160 (LDS-like box)", 3, 160, 1.5f}, {"scattered: 8x8 s=40 rs=1024", 3, 1024, 40.f}};
    for (auto& p : pats) {
        printf("pattern %s\n", p.n);
        for (int m = 0; m < 9; ++m) {
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(a);
                switch (m) {
                case 0: k<0><<<blocks, 256>>>(buf, iters, p.rowstride, p.spacing, p.tw, out, footprint); break;
                case 1: k<1><<<blocks, 256>>>(buf, iters, p.rowstride, p.spacing, p.tw, out, footprint); break;
                case 2: k<2><<<blocks, 256>>>(buf, iters, p.rowstride, p.spacing, p.tw, out, footprint); break;
                case 3: k<3><<<blocks, 256>>>(buf, iters, p.rowstride, p.spacing, p.tw, out, footprint); break;
                case 4: k<4><<<blocks, 256>>>(buf, iters, p.rowstride, p.spacing, p.tw, out, footprint); break;
                case 5: k<5><<<blocks, 256>>>(buf, iters, p.rowstride, p.spacing, p.tw, out, footprint); break;
                case 6: k<6><<<blocks, 256>>>(buf, iters, p.rowstride, p.spacing, p.tw, out, footprint); break;
                case 7: k<7><<<blocks, 256>>>(buf, iters, p.rowstride, p.spacing, p.tw, out, footprint); break;
                case 8: k<8><<<blocks, 256>>>(buf, iters, p.rowstride, p.spacing, p.tw, out, footprint); break;
                }
                hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
            }
            double winstr = (double)blocks * 4 * iters;           // wave instructions
            double cyc_per_cu = ms * 1e-3 * 2.4e9 / (winstr / 256.0);
            printf("  %-22s %8.3f ms  %6.1f cycles per wave-instr per CU (at 2.4 GHz)\n", names[m], ms, cyc_per_cu);
        }
    }
    return 0;
}
