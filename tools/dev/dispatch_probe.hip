// How long does the GPU take to START the workgroups of a launch shaped like k_pretok's (≈1250 workgroups of 256 lanes,
// ~80 VGPRs, ~28 KB of LDS, all resident at once), and what does a private (scratch) segment do to that?
// Development probe, not part of the product.  Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/dispatch_probe dispatch_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int SCRATCH_DWORDS, int LDS_BYTES>
__global__ __launch_bounds__(256) void k_probe(unsigned long long* start, unsigned long long* endt, const uint32_t* idx, uint32_t* sink,
                                              unsigned hold_ticks) {
    __shared__ uint32_t s[LDS_BYTES / 4];
    const unsigned long long t0 = wall_clock64();
    uint32_t acc = 0;
    if constexpr (SCRATCH_DWORDS > 0) {
        volatile uint32_t priv[SCRATCH_DWORDS];            // dynamically indexed: stays in scratch
        for (int i = 0; i < SCRATCH_DWORDS; i++) priv[i] = threadIdx.x + i;
        acc = priv[idx[threadIdx.x & 31] % SCRATCH_DWORDS];
    }
    s[threadIdx.x] = acc + threadIdx.x;
    __syncthreads();
    acc += s[(threadIdx.x + 1) & 255];
    asm volatile("; keep 80 VGPRs allocated" ::: "v79");
    if (threadIdx.x == 0) start[blockIdx.x] = t0;
    while (wall_clock64() - t0 < hold_ticks) { }          // stay resident, as the tiles of the real kernel do
    if (threadIdx.x == 0) endt[blockIdx.x] = wall_clock64();
    if (acc == 0xDEADBEEFu) sink[0] = acc;
}

template <int SD, int LDS> static void run(const char* name, int grid, unsigned hold) {
    unsigned long long *d_s, *d_e; uint32_t *d_idx, *d_sink;
    hipMalloc(&d_s, grid * 8); hipMalloc(&d_e, grid * 8); hipMalloc(&d_idx, 128); hipMalloc(&d_sink, 4);
    hipMemset(d_idx, 0, 128);
    std::vector<unsigned long long> hs(grid), he(grid);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    double p50 = 0, p90 = 0, mx = 0, last_end = 0, ev = 0;
    const int reps = 20;
    for (int r = -3; r < reps; r++) {
        hipEventRecord(e0);
        k_probe<SD, LDS><<<grid, 256>>>(d_s, d_e, d_idx, d_sink, hold);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        if (r < 0) continue;
        float ms; hipEventElapsedTime(&ms, e0, e1); ev += ms * 1000.0;
        hipMemcpy(hs.data(), d_s, grid * 8, hipMemcpyDeviceToHost);
        hipMemcpy(he.data(), d_e, grid * 8, hipMemcpyDeviceToHost);
        const unsigned long long t0 = *std::min_element(hs.begin(), hs.end());
        std::vector<double> st(grid);
        for (int i = 0; i < grid; i++) st[i] = (hs[i] - t0) * 0.01;          // 100 MHz ticks -> us
        std::sort(st.begin(), st.end());
        p50 += st[grid / 2]; p90 += st[grid * 9 / 10]; mx += st[grid - 1];
        last_end += (*std::max_element(he.begin(), he.end()) - t0) * 0.01;
    }
    printf("%-34s grid %5d  start p50 %5.2f  p90 %5.2f  max %5.2f us   last end %6.2f us   events %6.2f us\n", name, grid,
           p50 / reps, p90 / reps, mx / reps, last_end / reps, ev / reps);
    hipFree(d_s); hipFree(d_e); hipFree(d_idx); hipFree(d_sink);
}

int main(int argc, char** argv) {
    const int grid = argc > 1 ? atoi(argv[1]) : 1248;
    const unsigned hold = argc > 2 ? atoi(argv[2]) : 2000;   // 20 us
    run<0, 1024>("no scratch, 1 KB LDS", grid, hold);
    run<0, 28672>("no scratch, 28 KB LDS", grid, hold);
    run<20, 28672>("80 B/lane scratch, 28 KB LDS", grid, hold);
    run<84, 28672>("336 B/lane scratch, 28 KB LDS", grid, hold);
    run<0, 28672>("no scratch, 28 KB LDS (again)", grid, hold);
    run<0, 28672>("no scratch, 28 KB, 320 wgs", 320, hold);
    run<20, 28672>("80 B/lane scratch, 28 KB, 320 wgs", 320, hold);
    return 0;
}
