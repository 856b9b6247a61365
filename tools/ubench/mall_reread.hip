// Micro-benchmark (round 5): what a RE-READ of a 128-byte line costs as a function of how long ago the line was first read.
//
// Question behind it (VERDICT round 4, weak #2a): march_kernel moves 1.35 x its algorithmic bytes through the L2's memory-side (EA)
// interface; the lines it fetches twice are fetched "tens of microseconds" apart.  MI355X has a 256 MiB Infinity Cache (MALL) between the
// L2s and HBM, and FETCH_SIZE / TCC_EA0_RDREQ count its hits.  Are those re-fetches DRAM reads or MALL hits, and does a MALL hit relieve the
// stream?
//
// Method: a 4 GiB buffer is streamed once in 4 KiB chunks by a persistent grid (block b takes chunks b, b + G, ...; G = 2048 blocks keep
// 8 MiB in flight).  Beside chunk s a block also reads chunk s - lag ("re-read"; for `num` of every 4 chunks).  An odd lag puts the re-read
// on another XCD than the first read (block b runs on XCD b % 8), so the XCD's own L2 never holds it: what serves it is the MALL or DRAM.
// A lag that is a multiple of 8 re-reads on the same XCD (L2 hits while lag x 4 KiB / 8 < 4 MiB).
// Reported per lag: total bytes / time, and (under rocprofv3 --pmc, tools/r05_mall.sh) TCC_EA0_RDREQ, TCC_EA0_RDREQ_DRAM, TCC_HIT, TCC_MISS.
// The time between first read and re-read is lag x 4 KiB / (rate of NEW bytes).
//
// Build: hipcc -O3 --offload-arch=gfx950 -o bin/mall_reread mall_reread.hip        Run: bin/mall_reread [reps]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

__global__ __launch_bounds__(256) void reread(const uint4 *__restrict__ buf, size_t nchunks, size_t lag, int num, uint32_t *out)
{
    uint32_t acc = 0;
    const size_t G = gridDim.x;
#pragma unroll 4
    for (size_t s = blockIdx.x; s < nchunks; s += G) {
        const uint4 v = buf[s * 256 + threadIdx.x];
        acc += v.x ^ v.y ^ v.z ^ v.w;
        if (lag && s >= lag && (int)(s & 3) < num) {
            const uint4 w = buf[(s - lag) * 256 + threadIdx.x];
            acc += w.x ^ w.y ^ w.z ^ w.w;
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

// the same line set read over and over (a table that fits / does not fit the MALL): the MALL's own bandwidth
__global__ __launch_bounds__(256) void loop_read(const uint4 *__restrict__ buf, size_t nchunks, int passes, uint32_t *out)
{
    uint32_t acc = 0;
    const size_t G = gridDim.x;
    for (int p = 0; p < passes; ++p) {
        // every pass starts at another block offset: chunk s is read by a different XCD in every pass (no L2 hits)
#pragma unroll 4
        for (size_t s = (blockIdx.x + (size_t)p * 3) % G; s < nchunks; s += G) {
            const uint4 v = buf[s * 256 + threadIdx.x];
            acc += v.x ^ v.y ^ v.z ^ v.w;
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main(int argc, char **argv)
{
    const int reps = argc > 1 ? atoi(argv[1]) : 3;
    const size_t bytes = 4ull << 30, nchunks = bytes / 4096;
    uint4 *buf; if (hipMalloc(&buf, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 1, bytes);
    uint32_t *out; hipMalloc(&out, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int G = 2048;
    printf("# 4 GiB buffer, %d blocks x 256 threads, 4 KiB chunks, %d timed repetitions (minimum reported)\n", G, reps);
    printf("# lag_MiB xcd      num/4  ms       total_GB/s  new_GB/s  lag_us\n");
    const size_t lags_mib[] = {0, 2, 8, 16, 32, 64, 96, 128, 160, 192, 224, 256, 320, 384, 512, 1024};
    for (int num = 4; num >= 2; num -= 2)
        for (int same = 0; same < 2; ++same)
            for (size_t lm : lags_mib) {
                if (lm == 0 && (same || num != 4)) continue;
                if (same && lm > 64) continue;
                size_t lag = lm * 256;                                   // chunks
                if (lag) lag = same ? (lag & ~(size_t)7) : (lag | 1);
                float best = 1e30f;
                for (int r = 0; r < reps + 1; ++r) {
                    float ms = 0;
                    hipEventRecord(a);
                    reread<<<G, 256>>>(buf, nchunks, lag, num, out);
                    hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
                    if (r > 0 || reps == 0) best = ms < best ? ms : best;
                }
                size_t re = 0;
                if (lag) for (size_t s = lag; s < nchunks; ++s) re += (int)(s & 3) < num;
                const double gb = (double)(nchunks + re) * 4096 / 1e9, gbn = (double)nchunks * 4096 / 1e9;
                printf("%7zu %-8s %d/4    %7.3f  %9.1f  %8.1f  %7.1f\n", lm, lag ? (same ? "same" : "other") : "-", num, best,
                       gb / (best * 1e-3), gbn / (best * 1e-3), lag ? (double)lag * 4096 / (gbn * 1e9 / (best * 1e-3)) * 1e6 : 0.0);
                fflush(stdout);
            }
    printf("# the same table read 8 times over (every pass on other XCDs): size_MiB ms GB/s\n");
    const size_t tabs_mib[] = {16, 64, 128, 192, 224, 256, 320, 512, 2048};
    for (size_t tm : tabs_mib) {
        const size_t nc = tm * 256;
        float best = 1e30f;
        for (int r = 0; r < reps + 1; ++r) {
            float ms = 0;
            hipEventRecord(a);
            loop_read<<<G, 256>>>(buf, nc, 8, out);
            hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
            if (r > 0 || reps == 0) best = ms < best ? ms : best;
        }
        printf("table %5zu MiB  %7.3f ms  %8.1f GB/s\n", tm, best, (double)nc * 4096 * 8 / 1e9 / (best * 1e-3));
        fflush(stdout);
    }
    return 0;
}
