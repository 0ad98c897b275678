// Does an XCD's L2 keep read-only lines from one kernel launch to the next?  One wavefront per XCD chases a random
// permutation through a 2 MB buffer twice per launch (the second pass inside the launch is the L2-hit reference);
// three launches back to back.  Development probe, not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <numeric>
#include <random>
#include <algorithm>

__global__ void k_chase(const uint32_t* next, int steps, unsigned long long* out) {
    if (threadIdx.x != 0) return;
    uint32_t p = 0;
    for (int pass = 0; pass < 2; pass++) {
        const unsigned long long t0 = wall_clock64();
        for (int i = 0; i < steps; i++) p = next[p];
        const unsigned long long t1 = wall_clock64();
        out[blockIdx.x * 4 + pass] = t1 - t0;
        p = p == 0xFFFFFFFFu ? 1u : 0u;
    }
    out[blockIdx.x * 4 + 2] = p;
}
__global__ void k_other(uint32_t* x) { x[threadIdx.x + blockIdx.x * 256] += 1; }

int main() {
    const int lines = 16384, stride = 32;                  // 16 K lines of 128 B = 2 MB
    std::vector<uint32_t> perm(lines); std::iota(perm.begin(), perm.end(), 0u);
    std::mt19937 rng(7); std::shuffle(perm.begin() + 1, perm.end(), rng);
    std::vector<uint32_t> next((size_t)lines * stride, 0u);
    for (int i = 0; i < lines; i++) next[(size_t)perm[i] * stride] = perm[(i + 1) % lines] * stride;
    uint32_t *d_next, *d_x; unsigned long long* d_out;
    hipMalloc(&d_next, next.size() * 4); hipMalloc(&d_out, 8 * 4 * 8); hipMalloc(&d_x, 256 * 256 * 4);
    hipMemcpy(d_next, next.data(), next.size() * 4, hipMemcpyHostToDevice);
    hipMemset(d_x, 0, 256 * 256 * 4);
    const int steps = 2048;
    for (int mode = 0; mode < 2; mode++) {
        printf(mode == 0 ? "launches back to back:\n" : "with a small unrelated kernel between the launches:\n");
        for (int l = 0; l < 4; l++) {
            if (mode == 1) k_other<<<256, 256>>>(d_x);
            k_chase<<<8, 64>>>(d_next, steps, d_out);
            hipDeviceSynchronize();
            unsigned long long h[32]; hipMemcpy(h, d_out, sizeof h, hipMemcpyDeviceToHost);
            double a = 0, b = 0;
            for (int x = 0; x < 8; x++) { a += h[x * 4] * 10.0 / steps; b += h[x * 4 + 1] * 10.0 / steps; }
            printf("  launch %d: first pass %6.0f ns per dependent load, second pass (same launch) %6.0f ns   (mean of 8 workgroups)\n", l, a / 8, b / 8);
        }
    }
    return 0;
}
