// microbenchmark (MI355X): issue cost of the integer instructions of the samplers' address arithmetic, relative to v_add_u32
// (8 waves per SIMD, 8 independent chains per lane).   hipcc --offload-arch=gfx950 -O3 -o bin/int_rate int_rate.hip && bin/int_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define KERNEL(name, decl, init, body, fin) \
__global__ __launch_bounds__(256) void name(unsigned *out, unsigned a, unsigned b, int iters) { \
    decl; for (int i = 0; i < 8; ++i) { init; } \
    for (int it = 0; it < iters; ++it) { _Pragma("unroll") for (int r = 0; r < 16; ++r) { _Pragma("unroll") for (int i = 0; i < 8; ++i) { body; } } } \
    unsigned s = 0; for (int i = 0; i < 8; ++i) { fin; } out[blockIdx.x * blockDim.x + threadIdx.x] = s; }
KERNEL(k_add, unsigned x[8], x[i] = threadIdx.x + i, asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[i]) : "v"(a)), s += x[i])
KERNEL(k_mad24, unsigned x[8], x[i] = threadIdx.x + i, asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b)), s += x[i])
KERNEL(k_mullo, unsigned x[8], x[i] = threadIdx.x + i, asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[i]) : "v"(a)), s += x[i])
KERNEL(k_mad64, unsigned long long x[8], x[i] = threadIdx.x + i, asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x[i]) : "v"(a), "v"(b) : "vcc"), s += (unsigned)x[i])
KERNEL(k_lshladd64, unsigned long long x[8], x[i] = threadIdx.x + i, asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(x[i]) : "v"((unsigned long long)a)), s += (unsigned)x[i])
KERNEL(k_fma, float x[8], x[i] = (float)(threadIdx.x + i), asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(1.0001f), "v"(0.5f)), s += (unsigned)x[i])
KERNEL(k_med3, float x[8], x[i] = (float)(threadIdx.x + i), asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(1.0001f), "v"(0.5f)), s += (unsigned)x[i])
KERNEL(k_cvt, float x[8], x[i] = (float)(threadIdx.x + i), asm volatile("v_cvt_u32_f32 %0, %0" : "+v"(x[i])), s += (unsigned)x[i])
KERNEL(k_fract, float x[8], x[i] = (float)(threadIdx.x + i), asm volatile("v_fract_f32 %0, %0" : "+v"(x[i])), s += (unsigned)x[i])
KERNEL(k_mulf, float x[8], x[i] = (float)(threadIdx.x + i), asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(1.0001f)), s += (unsigned)x[i])
KERNEL(k_subf, float x[8], x[i] = (float)(threadIdx.x + i), asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x[i]) : "v"(1.0001f)), s += (unsigned)x[i])
KERNEL(k_lshl, unsigned x[8], x[i] = threadIdx.x + i, asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(x[i])), s += x[i])
KERNEL(k_and, unsigned x[8], x[i] = threadIdx.x + i, asm volatile("v_and_b32 %0, %0, %1" : "+v"(x[i]) : "v"(a)), s += x[i])
KERNEL(k_cndmask, unsigned x[8], x[i] = threadIdx.x + i, asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "s"(0x5555555555555555ull)), s += x[i])
KERNEL(k_or, unsigned x[8], x[i] = threadIdx.x + i, asm volatile("v_or_b32 %0, %0, %1" : "+v"(x[i]) : "v"(a)), s += x[i])
KERNEL(k_min, unsigned x[8], x[i] = threadIdx.x + i, asm volatile("v_min_u32 %0, %0, %1" : "+v"(x[i]) : "v"(a)), s += x[i])
KERNEL(k_subu, unsigned x[8], x[i] = threadIdx.x + i, asm volatile("v_sub_u32 %0, %0, %1" : "+v"(x[i]) : "v"(a)), s += x[i])
KERNEL(k_addf, float x[8], x[i] = (float)(threadIdx.x + i), asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(1.0001f)), s += (unsigned)x[i])
KERNEL(k_pkadd, unsigned long long x[8], x[i] = threadIdx.x + i, asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"((unsigned long long)a)), s += (unsigned)x[i])
KERNEL(k_mul24, unsigned x[8], x[i] = threadIdx.x + i, asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x[i]) : "v"(a)), s += x[i])
KERNEL(k_fmascalar, float x[8], x[i] = (float)(threadIdx.x + i), asm volatile("v_fma_f32 %0, %0, %1, 0.5" : "+v"(x[i]) : "s"(1.0001f)), s += (unsigned)x[i])
KERNEL(k_lshladd, unsigned x[8], x[i] = threadIdx.x + i, asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(x[i]) : "v"(a)), s += x[i])
KERNEL(k_bfe, unsigned x[8], x[i] = threadIdx.x + i, asm volatile("v_bfe_u32 %0, %0, 4, 8" : "+v"(x[i])), s += x[i])
KERNEL(k_add3, unsigned x[8], x[i] = threadIdx.x + i, asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b)), s += x[i])
KERNEL(k_ubyte, float x[8], x[i] = (float)(threadIdx.x + i), asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(x[i])), s += (unsigned)x[i])
KERNEL(k_perm, unsigned x[8], x[i] = threadIdx.x + i, asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b)), s += x[i])
KERNEL(k_cmp, unsigned x[8], x[i] = threadIdx.x + i, asm volatile("v_cmp_lt_u32 vcc, %0, %1" :: "v"(x[i]), "v"(a) : "vcc"), s += x[i])
KERNEL(k_max, float x[8], x[i] = (float)(threadIdx.x + i), asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[i]) : "v"(1.0001f)), s += (unsigned)x[i])
KERNEL(k_fmac, float x[8], x[i] = (float)(threadIdx.x + i), asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x[i]) : "v"(1.0001f), "v"(0.5f)), s += (unsigned)x[i])
KERNEL(k_floor, float x[8], x[i] = (float)(threadIdx.x + i), asm volatile("v_floor_f32 %0, %0" : "+v"(x[i])), s += (unsigned)x[i])
KERNEL(k_bfi, unsigned x[8], x[i] = threadIdx.x + i, asm volatile("v_bfi_b32 %0, %1, %2, %0" : "+v"(x[i]) : "v"(a), "v"(b)), s += x[i])
KERNEL(k_mov, unsigned x[8], x[i] = threadIdx.x + i, asm volatile("v_mov_b32 %0, %1" : "+v"(x[i]) : "v"(a)), s += x[i])
KERNEL(k_add_inl, float x[8], x[i] = (float)(threadIdx.x + i), asm volatile("v_add_f32 %0, -0.5, %0" : "+v"(x[i])), s += (unsigned)x[i])
KERNEL(k_add_lit, float x[8], x[i] = (float)(threadIdx.x + i), asm volatile("v_add_f32 %0, 0x47400000, %0" : "+v"(x[i])), s += (unsigned)x[i])
KERNEL(k_add_sgpr, float x[8], x[i] = (float)(threadIdx.x + i), asm volatile("v_add_f32 %0, %1, %0" : "+v"(x[i]) : "s"(1.0001f)), s += (unsigned)x[i])
KERNEL(k_fma_inl, float x[8], x[i] = (float)(threadIdx.x + i), asm volatile("v_fma_f32 %0, %0, %1, 0.5" : "+v"(x[i]) : "v"(1.0001f)), s += (unsigned)x[i])
KERNEL(k_fma_sgpr, float x[8], x[i] = (float)(threadIdx.x + i), asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "s"(1.0001f), "v"(0.5f)), s += (unsigned)x[i])
KERNEL(k_mul_lit, float x[8], x[i] = (float)(threadIdx.x + i), asm volatile("v_mul_f32 %0, 0x437f0000, %0" : "+v"(x[i])), s += (unsigned)x[i])
KERNEL(k_addu_inl, unsigned x[8], x[i] = threadIdx.x + i, asm volatile("v_add_u32 %0, 3, %0" : "+v"(x[i])), s += x[i])
KERNEL(k_addu_sgpr, unsigned x[8], x[i] = threadIdx.x + i, asm volatile("v_add_u32 %0, %1, %0" : "+v"(x[i]) : "s"(a)), s += x[i])
KERNEL(k_fma_2dst, float x[8]; float y = 0, x[i] = (float)(threadIdx.x + i), asm volatile("v_fma_f32 %0, %1, %2, %3" : "+v"(y) : "v"(x[i]), "v"(1.0001f), "v"(0.5f)), s += (unsigned)x[i] + (unsigned)y)
KERNEL(k_rcp, float x[8], x[i] = (float)(threadIdx.x + i), asm volatile("v_rcp_f32 %0, %0" : "+v"(x[i])), s += (unsigned)x[i])
int main()
{
    unsigned *d; (void)hipMalloc(&d, 2048 * 256 * sizeof(unsigned));
    const int iters = 1000, blocks = 256 * 8;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    typedef void (*K)(unsigned *, unsigned, unsigned, int);
    struct { const char *n; K k; } ks[] = {{"v_add_u32", k_add}, {"v_mad_u32_u24", k_mad24}, {"v_mul_lo_u32", k_mullo}, {"v_mad_u64_u32", k_mad64}, {"v_lshl_add_u64", k_lshladd64},
                                          {"v_fma_f32", k_fma}, {"v_med3_f32", k_med3}, {"v_cvt_u32_f32", k_cvt}, {"v_fract_f32", k_fract}, {"v_rcp_f32", k_rcp},
                                          {"v_mul_f32", k_mulf}, {"v_sub_f32", k_subf}, {"v_lshlrev_b32", k_lshl}, {"v_and_b32", k_and}, {"v_cndmask_b32", k_cndmask}, {"v_lshl_add_u32", k_lshladd},
                                          {"v_bfe_u32", k_bfe}, {"v_add3_u32", k_add3}, {"v_cvt_f32_ubyte1", k_ubyte}, {"v_perm_b32", k_perm}, {"v_cmp_lt_u32", k_cmp}, {"v_max_f32", k_max},
                                          {"v_fmac_f32", k_fmac}, {"v_floor_f32", k_floor}, {"v_bfi_b32", k_bfi}, {"v_mov_b32", k_mov}, {"v_or_b32", k_or}, {"v_min_u32", k_min}, {"v_sub_u32", k_subu}, {"v_add_f32", k_addf}, {"v_pk_add_f32", k_pkadd}, {"v_mul_u32_u24", k_mul24}, {"v_fma_f32 v,s,c", k_fmascalar}, {"v_add_f32 inline", k_add_inl}, {"v_add_f32 literal", k_add_lit}, {"v_add_f32 sgpr", k_add_sgpr}, {"v_fma_f32 v,v,inl", k_fma_inl}, {"v_fma_f32 v,s,v", k_fma_sgpr}, {"v_mul_f32 literal", k_mul_lit}, {"v_add_u32 inline", k_addu_inl}, {"v_add_u32 sgpr", k_addu_sgpr}, {"v_fma dst!=src", k_fma_2dst}};
    float base = 0;
    for (auto &q : ks) {
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(q.k, dim3(blocks), dim3(256), 0, 0, d, 3u, 5u, iters);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
        }
        if (base == 0) base = ms;
        printf("%-16s %.3f ms  = %.2f x v_add_u32\n", q.n, ms, ms / base);
    }
    return 0;
}
