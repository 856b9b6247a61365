// lds_pairs.hip -- how should the sweep kernel read an x-pair (8 bytes at a 4-byte aligned LDS address)?
//   variant 0: ds_read2_b32 offset1:1   (what hipcc emits for an align-4 float2)
//   variant 1: ds_read_b64 at the unaligned address (inline asm)
//   variant 2: ds_read_b64 at the address rounded down to 8 bytes (what an aligned layout would cost; values differ)
// 64 lanes read pairs at x = floor(spacing * lane) of two rows (like a 32x2 pixel wave: lanes 32..63 one row up),
// 4 reads per "sample" (2 rows x 2 slices).  Prints cycles per wave-sample per CU and checks the values.
//   hipcc --offload-arch=gfx950 -O3 -o bin/lds_pairs lds_pairs.hip && bin/lds_pairs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float __attribute__((ext_vector_type(2), aligned(4))) float2u;
typedef float __attribute__((ext_vector_type(2))) f2;

template <int VAR>
__global__ __launch_bounds__(1024) void k(float *out, unsigned long long *cyc, int iters, float spacing, int pitch, int slot)
{
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    float *f = (float *)lds;
    for (int i = threadIdx.x; i < 36 * 1024; i += blockDim.x) f[i] = (float)i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int x = (int)(spacing * (float)(lane & 31)) + 1;        // odd/even mix: 4-byte aligned only
    uint32_t a0 = (uint32_t)((lane >> 5) * 2 * pitch + x * 4);
    float acc = 0.f;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        uint32_t a = a0 + (uint32_t)((it & 7) * 4);
        if (VAR == 0 || VAR == 2) {
            f2 p, q, r, s;
            uint32_t a_ = VAR == 2 ? (a & ~7u) : a;            // variant 2: the same reads 8-byte aligned (reference)
            uint32_t b = a_ + pitch, c = a_ + slot, d = a_ + slot + pitch;
            if (VAR == 0)
                asm volatile("ds_read2_b32 %0, %4 offset1:1\n\tds_read2_b32 %1, %5 offset1:1\n\tds_read2_b32 %2, %6 offset1:1\n\tds_read2_b32 %3, %7 offset1:1\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(p), "=&v"(q), "=&v"(r), "=&v"(s) : "v"(a_), "v"(b), "v"(c), "v"(d) : "memory");
            else
                asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %5\n\tds_read_b64 %2, %6\n\tds_read_b64 %3, %7\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(p), "=&v"(q), "=&v"(r), "=&v"(s) : "v"(a_), "v"(b), "v"(c), "v"(d) : "memory");
            acc += (p.x + p.y) + (q.x + q.y) + (r.x + r.y) + (s.x + s.y);
        } else {
            f2 p, q, r, s;
            uint32_t b = a + pitch, c = a + slot, d = a + slot + pitch;
            asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %5\n\tds_read_b64 %2, %6\n\tds_read_b64 %3, %7\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(p), "=&v"(q), "=&v"(r), "=&v"(s) : "v"(a), "v"(b), "v"(c), "v"(d) : "memory");
            acc += (p.x + p.y) + (q.x + q.y) + (r.x + r.y) + (s.x + s.y);
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main()
{
    const int blocks = 256, threads = 1024, iters = 4096;
    float *out; unsigned long long *cyc;
    hipMalloc(&out, blocks * threads * 4); hipMalloc(&cyc, blocks * 8);
    for (float spacing : {1.0f, 1.2f, 1.6f, 1.96f}) for (int var = 0; var < 3; ++var) {
        const int pitch = 896, slot = 17 * 896;
        auto kern = var == 0 ? k<0> : (var == 1 ? k<1> : k<2>);
        hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 36 * 1024 * 4, 0, out, cyc, iters, spacing, pitch, slot);
        hipDeviceSynchronize();
        std::vector<unsigned long long> c(blocks); std::vector<float> o(threads);
        hipMemcpy(c.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
        hipMemcpy(o.data(), out, threads * 4, hipMemcpyDeviceToHost);
        // expected value for thread 0 .. check lane 5 of wave 0
        double want = 0; { int lane = 5; int x = (int)(spacing * lane) + 1;
            for (int it = 0; it < iters; ++it) { int a = (x * 4 + (it & 7) * 4) / 4; int P = pitch / 4, S = slot / 4;
                want += (double)((a) + (a + 1)) + (double)((a + P) + (a + P + 1)) + (double)((a + S) + (a + S + 1)) + (double)((a + S + P) + (a + S + P + 1)); } }
        double mean = 0; for (auto v : c) mean += v; mean /= blocks;
        printf("spacing %.2f variant %d: %.1f cycles per wave-sample per CU (16 waves, 4 reads), lane5 got %.0f want %.0f rel.err %.2e\n",
               spacing, var, mean / iters / 16.0, (double)o[5], want, (o[5] - want) / want);
    }
    return 0;
}
