// Dev probe (gfx950): does global_load_lds_dwordx4 place lane l's 16 bytes at M0 + 16 l, for a partial wavefront too?
//   hipcc --offload-arch=gfx950 -O2 tools/dev/lds_dma_probe.hip -o tools/dev/lds_dma_probe.bin && tools/dev/lds_dma_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
__global__ void k(const uint8_t* src, uint32_t* out) {
    __shared__ __attribute__((aligned(16))) uint32_t buf[2][264];
    const int tid = threadIdx.x;
    for (int i = tid; i < 528; i += 256) (&buf[0][0])[i] = 0xDEADBEEFu;
    __syncthreads();
    if (tid < 65) {
        const uint8_t* g = src + (size_t)blockIdx.x * 1040 + tid * 16;
        uint32_t* l = buf[1] + (tid & ~63) * 4;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    out[blockIdx.x * 264 + tid] = buf[1][tid];
    if (tid < 8) out[blockIdx.x * 264 + 256 + tid] = buf[1][256 + tid];
}
int main() {
    const int nb = 4;
    std::vector<uint8_t> h(nb * 1040 + 64);
    for (size_t i = 0; i < h.size(); i++) h[i] = (uint8_t)(i * 7 + 3);
    uint8_t* d; uint32_t* o;
    hipMalloc(&d, h.size()); hipMalloc(&o, nb * 264 * 4);
    hipMemcpy(d, h.data(), h.size(), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(nb), dim3(256), 0, 0, d, o);
    std::vector<uint32_t> r(nb * 264);
    hipMemcpy(r.data(), o, r.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int b = 0; b < nb; b++)
        for (int w = 0; w < 260; w++) {
            uint32_t want; memcpy(&want, &h[(size_t)b * 1040 + w * 4], 4);
            if (r[b * 264 + w] != want) { if (bad < 5) printf("block %d word %d: got %08x want %08x\n", b, w, r[b * 264 + w], want); bad++; }
        }
    printf("lds dma probe: %d mismatches of %d words\n", bad, nb * 260);
    return bad != 0;
}
