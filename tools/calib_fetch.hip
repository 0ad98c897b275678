// calib_fetch.hip -- what do FETCH_SIZE / WRITE_SIZE count on gfx950?  Known access patterns over a
// buffer far larger than L2 + Infinity Cache, to be run under rocprofv3 --pmc FETCH_SIZE (and
// WRITE_SIZE, TCC_MISS_sum) and compared with the bytes each kernel is known to touch:
//   k_stream      every lane 16 B, fully coalesced: n_bytes read once
//   k_scatter<G>  groups of G lanes read G*16 contiguous bytes at a random 128-byte-aligned place (+ a
//                 random 32-byte sub-offset for G = 2): n_groups * G*16 bytes asked for -- the shape of a
//                 32-byte (tiny table) or 48/64-byte (t8 / short table) bucket probe
//   k_write       every lane writes 4 B, coalesced
// Build: hipcc --offload-arch=gfx950 -O2 -o gpurun_out/calib_fetch tools/calib_fetch.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

__global__ void k_stream(const uint4* __restrict__ p, size_t n16, uint32_t* sink) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (; i < n16; i += (size_t)gridDim.x * blockDim.x) { const uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345u) *sink = acc;
}
template <int G> __global__ void k_scatter(const uint4* __restrict__ p, size_t n128, uint32_t ngroups, uint32_t* sink) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t g = t / G, l = t % G;
    if (g >= ngroups) return;
    uint64_t h = (uint64_t)g * 0x9E3779B97F4A7C15ull;
    h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
    const size_t line = (size_t)(h % n128);
    const uint32_t sub = G == 2 ? (uint32_t)((h >> 40) & 3u) * 2u : G == 4 ? (uint32_t)((h >> 40) & 1u) * 4u : 0u;   // in 16-byte units
    const uint4 v = p[line * 8 + sub + l];
    if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345u) *sink = v.x;
}
__global__ void k_write(uint32_t* p, size_t n4) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n4; i += (size_t)gridDim.x * blockDim.x) p[i] = (uint32_t)i;
}

int main() {
    const size_t bytes = 4ull << 30;                       // 4 GiB: beyond L2 (32 MiB) and Infinity Cache (256 MiB)
    uint8_t* d; uint32_t* sink;
    if (hipMalloc(&d, bytes) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(d, 1, bytes);
    hipDeviceSynchronize();
    const uint32_t ngroups = 1u << 22;                     // 4 Mi random probes per scatter launch
    printf("known bytes per launch: k_stream %zu, k_scatter<2> %u, k_scatter<4> %u, k_scatter<8> %u, k_write %zu\n",
           (size_t)1 << 30, ngroups * 32, ngroups * 64, ngroups * 128, (size_t)1 << 28);
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(k_stream, dim3(4096), dim3(256), 0, 0, (const uint4*)d, ((size_t)1 << 30) / 16, sink);
        hipLaunchKernelGGL(k_scatter<2>, dim3(ngroups * 2 / 256), dim3(256), 0, 0, (const uint4*)d, bytes / 128, ngroups, sink);
        hipLaunchKernelGGL(k_scatter<4>, dim3(ngroups * 4 / 256), dim3(256), 0, 0, (const uint4*)d, bytes / 128, ngroups, sink);
        hipLaunchKernelGGL(k_scatter<8>, dim3(ngroups / 32), dim3(256), 0, 0, (const uint4*)d, bytes / 128, ngroups, sink);
        hipLaunchKernelGGL(k_write, dim3(4096), dim3(256), 0, 0, (uint32_t*)d, ((size_t)1 << 28) / 4);
    }
    hipDeviceSynchronize();
    printf("done\n");
    return 0;
}
