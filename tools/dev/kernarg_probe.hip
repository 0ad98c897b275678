// How long does a wavefront wait for its kernel arguments?  (They are written by the host for every launch, so the
// first s_load of a launch can hit in no cache.)  Each workgroup records the wall clock before it touches any argument
// and after it has read one -- a leading scalar, and a word at the end of a 448-byte struct passed by value.
// Built twice: plain, and with -mllvm -amdgpu-kernarg-preload-count=16 (leading scalars arrive in SGPRs).
// Development probe, not part of the product.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
struct Big { unsigned long long w[56]; };
__global__ __launch_bounds__(256) void k_args(unsigned long long* out, unsigned hold, unsigned lead, Big big) {
    const unsigned long long t0 = wall_clock64();
    unsigned long long* o = out + 4 * blockIdx.x;            // needs `out`: the first argument
    const unsigned long long t1 = wall_clock64();
    asm volatile("" :: "s"(lead));
    const unsigned long long t2 = wall_clock64();
    const unsigned long long tail = big.w[55];
    asm volatile("" :: "s"(tail));
    const unsigned long long t3 = wall_clock64();
    if (threadIdx.x == 0) { o[0] = t0; o[1] = t1 + (out == nullptr); o[2] = t2; o[3] = t3 + (tail == 77); }
    while (wall_clock64() - t0 < hold) { }
}
int main(int argc, char** argv) {
    const int grid = argc > 1 ? atoi(argv[1]) : 1248;
    unsigned long long* d; hipMalloc(&d, grid * 32);
    std::vector<unsigned long long> h(grid * 4);
    Big big{}; big.w[55] = 5;
    double a[4][3] = {};
    const int reps = 20;
    for (int r = -3; r < reps; r++) {
        k_args<<<grid, 256>>>(d, 1500, 3, big);
        hipDeviceSynchronize();
        if (r < 0) continue;
        hipMemcpy(h.data(), d, grid * 32, hipMemcpyDeviceToHost);
        unsigned long long m = ~0ull;
        for (int i = 0; i < grid; i++) m = std::min(m, h[4 * i]);
        for (int k = 0; k < 4; k++) {
            std::vector<double> v(grid);
            for (int i = 0; i < grid; i++) v[i] = (h[4 * i + k] - m) * 0.01;
            std::sort(v.begin(), v.end());
            a[k][0] += v[grid / 2]; a[k][1] += v[grid * 9 / 10]; a[k][2] += v[grid - 1];
        }
    }
    const char* nm[4] = {"first instruction", "pointer argument used", "leading scalar read", "struct tail (byte 440+) read"};
    printf("grid %d: wall clock since the earliest workgroup's first instruction, us (p50 / p90 / max)\n", grid);
    for (int k = 0; k < 4; k++) printf("  %-30s %5.2f / %5.2f / %5.2f\n", nm[k], a[k][0] / reps, a[k][1] / reps, a[k][2] / reps);
    return 0;
}
