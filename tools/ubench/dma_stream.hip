// dma_stream.hip -- what does a CU's LDS-DMA stream deliver in the sweep kernel's shape?
// One 1024-thread block per CU.  NL loader waves copy "slices" (ROWS rows of RB bytes, row pitch 4128 B, slice pitch
// 4.3 MB: the C3 volume) round-robin into an LDS ring of RING slots with global_load_lds_dwordx4, one row per
// instruction, keeping DEPTH slices per wave in flight (vmcnt).  The other waves (a) idle, (b) spin on VALU, or
// (c) read LDS pairs + VALU like the march.  Prints GB/s per CU and TB/s chip-wide.
//   hipcc --offload-arch=gfx950 -O3 -o bin/dma_stream dma_stream.hip && bin/dma_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
__device__ __forceinline__ void wait_vm_n(int n)
{
    n = n > 63 ? 63 : n;
    switch (n) {
#define W(i) case i: wait_vm<i>(); break;
#define W8(b) W(b) W(b + 1) W(b + 2) W(b + 3) W(b + 4) W(b + 5) W(b + 6) W(b + 7)
    W8(0) W8(8) W8(16) W8(24) W8(32) W8(40) W8(48) W8(56)
    default: wait_vm<0>(); break;
    }
}

__global__ __launch_bounds__(1024) void k(const char *vol, size_t row_pitch, size_t slice_pitch, int nslices, int rows, int row_bytes,
                                          int nl, int ring, int depth, int mode, int prio, float *out, unsigned long long *cyc)
{
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int slot_bytes = rows * 1024;
    volatile int *done = (volatile int *)(lds + ring * slot_bytes);      // loaders still running
    if (threadIdx.x == 0) *done = nl;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (wave < nl) {
        if (prio) __builtin_amdgcn_s_setprio(3);
        // tile (blockIdx) walks its own column of the volume: x offset per block, rows from y0
        const char *base = vol + (size_t)(blockIdx.x % 9) * 896 + (size_t)(blockIdx.x / 9 % 60) * 17 * row_pitch + (size_t)lane * 16;
        int pend = 0;                                     // slices in flight
        for (int s = wave; s < nslices; s += nl) {
            const char *g = base + (size_t)s * slice_pitch;
            int lb = (s % ring) * slot_bytes;
            const bool mine = lane * 16 < row_bytes;
            for (int r = 0; r < rows; ++r) {
                if (mine)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g, (__attribute__((address_space(3))) void *)(lds + lb), 16, 0, 0);
                g += row_pitch; lb += 1024;
            }
            if (++pend > depth) { wait_vm_n(depth * rows); --pend; }
        }
        wait_vm<0>();
        if (lane == 0) atomicSub((int *)done, 1);
    } else {
        float acc = (float)lane;
        uint32_t a = (uint32_t)(lane * 8 + wave * 1024);
        while (*done > 0) {
            if (mode == 1) { for (int i = 0; i < 64; ++i) acc = __builtin_fmaf(acc, 1.0001f, 0.5f); }
            else if (mode == 2) {
                for (int i = 0; i < 8; ++i) {
                    const volatile float *pp = (const volatile float *)(lds + a), *qq = (const volatile float *)(lds + a + 1024);
                    acc += pp[0] + pp[1] + qq[0] * qq[1];
                    for (int j = 0; j < 24; ++j) acc = __builtin_fmaf(acc, 1.0001f, 0.5f);
                    a = (a + 64) & 0xffff;
                }
            } else __builtin_amdgcn_s_sleep(8);
        }
        out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    }
    if (threadIdx.x == 0) cyc[blockIdx.x] = __builtin_readcyclecounter() - t0;
}

int main()
{
    const size_t row_pitch = 4128, slice_pitch = row_pitch * 1025;
    const int nz = 1024;
    char *vol; hipMalloc(&vol, slice_pitch * (nz + 2)); hipMemset(vol, 1, slice_pitch * (nz + 2));
    float *out; unsigned long long *cyc; hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 8);
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("rows 17 x 896 B per slice, 1024 slices per block, 256 blocks\n");
    for (int mode = 0; mode < 3; ++mode)
        for (int nl : {1, 2, 4})
            for (int depth : {1, 2, 3})
                for (int prio : {0, 1}) {
                    const int rows = 17, rb = 896, ring = 8;
                    if (mode != 2 && prio) continue;
                    if ((depth + 1) * nl > ring + 4) continue;
                    hipEventRecord(e0);
                    hipLaunchKernelGGL(k, dim3(256), dim3(1024), ring * rows * 1024 + 64, 0, vol, row_pitch, slice_pitch, nz, rows, rb, nl, ring, depth, mode, prio, out, cyc);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    const double bytes = 256.0 * nz * rows * rb;
                    printf("others %s, %d loaders, %d slices in flight each, prio %d: %.3f ms  %.1f GB/s per CU  %.2f TB/s\n",
                           mode == 0 ? "idle" : (mode == 1 ? "VALU" : "LDS+VALU"), nl, depth, prio, ms, bytes / 256 / ms / 1e6, bytes / ms / 1e9);
                }
    return 0;
}
