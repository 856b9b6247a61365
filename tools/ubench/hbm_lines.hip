// Micro-benchmark 3: HBM read bandwidth of a 4 GiB buffer as a function of the contiguous run each
// request stream touches: sequential streaming vs. pseudo-random runs of 64 B ... 4 KB.
// Gives the practical ceiling for a gather kernel whose unit of locality is a cache line or a brick.
// Build: hipcc -O3 --offload-arch=gfx950 -o hbm_lines hbm_lines.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ uint32_t mix32(uint32_t h)
{
    h ^= h >> 16; h *= 0x7feb352dU; h ^= h >> 15; h *= 0x846ca68bU; h ^= h >> 16;
    return h;
}

// Every group of (run/16) consecutive lanes reads one contiguous run of `run` bytes (16 B per lane);
// runs are visited in hashed order.  run == 0: plain streaming (consecutive lanes, consecutive runs).
__global__ __launch_bounds__(256) void reader(const uint4* __restrict__ buf, size_t n16, uint32_t run_log2, int iters, uint32_t* out)
{
    const size_t gtid = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t nthreads = (size_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    const uint32_t lanes_per_run = run_log2 ? (1u << (run_log2 - 4)) : 1u;
    const size_t nruns = run_log2 ? (n16 >> (run_log2 - 4)) : n16;
    for (int it = 0; it < iters; ++it) {
        size_t idx;
        if (!run_log2) idx = (gtid + (size_t)it * nthreads) % n16;
        else {
            const size_t r = (gtid / lanes_per_run) + (size_t)it * (nthreads / lanes_per_run);
            const size_t hr = ((size_t)mix32((uint32_t)r) | ((size_t)mix32((uint32_t)(r >> 32) + 17u) << 32)) % nruns;
            idx = hr * lanes_per_run + (gtid % lanes_per_run);
        }
        uint4 v = buf[idx];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main()
{
    const size_t bytes = 4ull << 30, n16 = bytes / 16;
    uint4* buf; if (hipMalloc(&buf, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 1, bytes);
    uint32_t* out; hipMalloc(&out, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = 256 * 32, iters = 64;           // 2 M threads x 64 x 16 B = 2 GiB per launch
    const uint32_t runs[] = {0, 6, 7, 8, 9, 10, 12};
    printf("4 GiB buffer, %d blocks x 256 threads, %d x 16 B per thread\n", blocks, iters);
    for (uint32_t rl : runs) {
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(a);
            reader<<<blocks, 256>>>(buf, n16, rl, iters, out);
            hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
        }
        const double gb = (double)blocks * 256 * iters * 16 / 1e9;
        if (rl) printf("random runs of %5u B : %7.3f ms  %7.1f GB/s\n", 1u << rl, ms, gb / (ms * 1e-3));
        else    printf("sequential streaming   : %7.3f ms  %7.1f GB/s\n", ms, gb / (ms * 1e-3));
    }
    return 0;
}
