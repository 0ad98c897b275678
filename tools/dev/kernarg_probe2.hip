// Kernel-argument latency, part 2: is it the FRESHNESS of the argument buffer (written by the host for every launch)
// or just a cold fetch?  The same kernel launched (a) normally, (b) as an instantiated hipGraph whose argument buffer
// is written once.  Every workgroup records the wall clock once its last argument (end of a 448-byte struct) is there;
// reported relative to a timestamp the PREVIOUS tiny kernel on the stream left (so the absolute start-up cost of a
// launch shows, not only the spread between workgroups).  Development probe, not part of the product.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
struct Big { unsigned long long w[56]; };
__global__ void k_mark(unsigned long long* t) { if (threadIdx.x == 0) t[0] = wall_clock64(); }
__global__ __launch_bounds__(256) void k_args(unsigned long long* out, unsigned hold, Big big) {
    const unsigned long long tail = big.w[55];
    asm volatile("" :: "s"(tail));
    const unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 + (tail == 77);
    while (wall_clock64() - t1 < hold) { }
}
int main(int argc, char** argv) {
    const int grid = argc > 1 ? atoi(argv[1]) : 1248;
    unsigned long long *d, *dm; hipMalloc(&d, grid * 8); hipMalloc(&dm, 8);
    std::vector<unsigned long long> h(grid); unsigned long long hm;
    Big big{}; big.w[55] = 5;
    hipStream_t s; hipStreamCreate(&s);
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    k_mark<<<1, 64, 0, s>>>(dm);
    k_args<<<grid, 256, 0, s>>>(d, 1500, big);
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int mode = 0; mode < 2; mode++) {
        double a[3] = {}, first = 0;
        const int reps = 20;
        for (int r = -3; r < reps; r++) {
            if (mode == 0) { k_mark<<<1, 64, 0, s>>>(dm); k_args<<<grid, 256, 0, s>>>(d, 1500, big); }
            else hipGraphLaunch(ge, s);
            hipStreamSynchronize(s);
            if (r < 0) continue;
            hipMemcpy(h.data(), d, grid * 8, hipMemcpyDeviceToHost); hipMemcpy(&hm, dm, 8, hipMemcpyDeviceToHost);
            std::vector<double> v(grid);
            for (int i = 0; i < grid; i++) v[i] = ((long long)h[i] - (long long)hm) * 0.01;
            std::sort(v.begin(), v.end());
            first += v[0]; a[0] += v[grid / 2]; a[1] += v[grid * 9 / 10]; a[2] += v[grid - 1];
        }
        printf("grid %d, %s: arguments there, us after the previous kernel's stamp: first %5.2f  p50 %5.2f  p90 %5.2f  max %5.2f\n", grid,
               mode == 0 ? "plain launches " : "graph launches ", first / reps, a[0] / reps, a[1] / reps, a[2] / reps);
    }
    return 0;
}
