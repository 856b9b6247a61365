// ubench / check: can buffer_load_dwordx2 in index mode (stride 4 in the descriptor, `idxen`) address more than 4 GiB with a
// 32-bit dword index, where global_load saddr + 32-bit byte offset stops?  Measured on MI355X (profiles/r02_ubench_buffer_idx.txt):
// NO -- index * stride + offset is a 32-bit quantity, every x pair beyond the 4 GiB mark of a 5 GiB buffer reads the wrong
// address (20 % of uniformly drawn indices), so the linear layout keeps its 64-bit variant above 4 GiB.
// build: hipcc -O3 --offload-arch=gfx950 -o bin/buffer_idx buffer_idx.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int   i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ f32x2 struct_buffer_load_v2f32(i32x4 rsrc, int vindex, int voffset, int soffset, int aux) __asm("llvm.amdgcn.struct.buffer.load.v2f32");

__global__ void fill(float *p, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (float)(i % 16777213ull); }

__global__ void check(const float *p, unsigned nrec, unsigned row_b, unsigned slice_b, const unsigned *idx, int m, unsigned long long *bad, float *oob)
{
    const unsigned long long b = (unsigned long long)p;
    i32x4 r;
    r.x = (int)(unsigned)b; r.y = (int)(((unsigned)(b >> 32) & 0xffffu) | (4u << 16)); r.z = (int)nrec; r.w = 0x00020000;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m) return;
    const unsigned i = idx[t];
    const f32x2 a = struct_buffer_load_v2f32(r, (int)i, 0, 0, 0), c = struct_buffer_load_v2f32(r, (int)i, 0, (int)row_b, 0), d = struct_buffer_load_v2f32(r, (int)i, 0, (int)slice_b, 0);
    const unsigned long long i0 = i, ir = i0 + row_b / 4, is = i0 + slice_b / 4;
    bool ok = a.x == (float)(i0 % 16777213ull) && a.y == (float)((i0 + 1) % 16777213ull) && c.x == (float)(ir % 16777213ull) && c.y == (float)((ir + 1) % 16777213ull) &&
              d.x == (float)(is % 16777213ull) && d.y == (float)((is + 1) % 16777213ull);
    if (!ok) atomicAdd(bad, 1ull);
    if (t == 0) { const f32x2 o = struct_buffer_load_v2f32(r, (int)nrec, 0, 0, 0); oob[0] = o.x; oob[1] = o.y; }
}

int main()
{
    const size_t n = (5ull << 30) / 4;                     // 5 GiB of floats
    float *p; unsigned *d_idx; unsigned long long *d_bad; float *d_oob;
    if (hipMalloc(&p, n * 4) != hipSuccess) { printf("no memory\n"); return 1; }
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, p, n);
    const int m = 1 << 20;
    unsigned *h = new unsigned[m];
    uint64_t s = 88172645463325252ull;
    const unsigned row_b = 4128, slice_b = 4231200;
    const unsigned nrec = (unsigned)(n - 2 - slice_b / 4 - row_b / 4);
    for (int k = 0; k < m; ++k) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[k] = (unsigned)(s % nrec); }
    h[1] = nrec - 1; h[2] = (1u << 30) + 1; h[3] = (1u << 30) - 1; h[4] = 0;          // around the 4 GiB mark, odd dword addresses
    hipMalloc(&d_idx, m * 4); hipMalloc(&d_bad, 8); hipMalloc(&d_oob, 8);
    hipMemcpy(d_idx, h, m * 4, hipMemcpyHostToDevice); hipMemset(d_bad, 0, 8);
    hipLaunchKernelGGL(check, dim3(m / 256), dim3(256), 0, 0, p, nrec, row_b, slice_b, d_idx, m, d_bad, d_oob);
    unsigned long long bad = 1; float oob[2];
    hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost); hipMemcpy(oob, d_oob, 8, hipMemcpyDeviceToHost);
    printf("index-mode buffer loads over a 5 GiB buffer: %d x-pairs x 3 corners checked, %llu wrong; record == num_records reads (%g, %g)\n", m, bad, oob[0], oob[1]);
    return bad ? 2 : 0;
}
