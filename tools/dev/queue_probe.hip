// Dev aid: which HIP streams of a process run side by side?  Pairwise: a ~150 us spin kernel on stream A, four empty kernels on stream B; how long B's take.
//   hipcc --offload-arch=gfx950 -O2 -o queue_probe tools/dev/queue_probe.hip && ./queue_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void k_spin(unsigned long long ticks) { const unsigned long long t0 = wall_clock64(); while (wall_clock64() - t0 < ticks) { } }
__global__ void k_nop() { }
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const int G = argc > 1 ? atoi(argv[1]) : 8, B = argc > 2 ? atoi(argv[2]) : 64, NOPS = argc > 3 ? atoi(argv[3]) : 4;
    printf("spinner %d x %d, %d empty kernels\n", G, B, NOPS);
    int lo, hi;
    hipDeviceGetStreamPriorityRange(&lo, &hi);
    printf("priority range: lowest %d highest %d\n", lo, hi);
    std::vector<hipStream_t> st; std::vector<int> pr;
    for (int p : {0, 0, 0, 0, 0, 0, hi, hi, hi, lo, lo, lo}) { hipStream_t s; hipStreamCreateWithPriority(&s, hipStreamNonBlocking, p); st.push_back(s); pr.push_back(p); }
    for (auto s : st) { hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, s); }
    hipDeviceSynchronize();
    const int n = (int)st.size();
    printf("rows: the stream that spins (150 us, 8 workgroups); columns: the stream that runs four empty kernels; us until they are through (min of 3)\n      ");
    for (int b = 0; b < n; b++) printf(" %2d(%+d)", b, pr[b]);
    printf("\n");
    for (int a = 0; a < n; a++) {
        printf("%2d(%+d)", a, pr[a]);
        for (int b = 0; b < n; b++) {
            if (a == b) { printf("    -  "); continue; }
            double best = 1e9;
            for (int rep = 0; rep < 3; rep++) {
                hipStreamSynchronize(st[a]); hipStreamSynchronize(st[b]);
                hipLaunchKernelGGL(k_spin, dim3(G), dim3(B), 0, st[a], 15000ull);
                const double t0 = now_us();
                for (int k = 0; k < NOPS; k++) hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, st[b]);
                hipStreamSynchronize(st[b]);
                const double t1 = now_us();
                hipStreamSynchronize(st[a]);
                if (t1 - t0 < best) best = t1 - t0;
            }
            printf(" %6.0f", best);
        }
        printf("\n");
    }
    return 0;
}
