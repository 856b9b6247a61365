// microbenchmark (MI355X): issue rate of v_pk_fma_f32 against v_fma_f32 on wave64 -- the same number of INSTRUCTIONS per wave in both
// kernels (8 independent chains), so equal times mean a packed instruction issues like a plain one (2 x the flops), double time means
// it takes two issue slots.   hipcc --offload-arch=gfx950 -O3 -o bin/pk_rate pk_rate.hip && bin/pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float __attribute__((ext_vector_type(2))) f2;
__global__ __launch_bounds__(256) void plain(float *out, float a, float b, int iters)
{
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = (float)(threadIdx.x + i);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
        }
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void packed(float *out, float a, float b, int iters)
{
    f2 x[8];
    for (int i = 0; i < 8; ++i) x[i] = f2{(float)(threadIdx.x + i), (float)i};
    const f2 aa = {a, a}, bb = {b, b};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(aa), "v"(bb));
        }
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += x[i].x + x[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main()
{
    float *d; hipMalloc(&d, 256 * 1024 * 256 * sizeof(float));
    const int iters = 2000, blocks = 256 * 8;           // 8 blocks of 4 waves per CU: 8 waves per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int which = 0; which < 2; ++which) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (which == 0) hipLaunchKernelGGL(plain, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f, iters);
            else hipLaunchKernelGGL(packed, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double inst = (double)blocks * 4 * iters * 128;           // wave instructions
            if (rep == 2) printf("%s: %.3f ms, %.1f G wave-instructions/s, %.2f cycles per instruction per SIMD at 2.4 GHz\n", which ? "v_pk_fma_f32" : "v_fma_f32   ", ms, inst / ms / 1e6, 2.4e9 * 1024 * (ms * 1e-3) / inst);
        }
    }
    return 0;
}
