// What wave_shr:1 / wave_shl:1 DPP moves on gfx950: hipcc --offload-arch=gfx950 tools/probes/dpp_probe.hip -o /tmp/dpp_probe && /tmp/dpp_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* p) {
    unsigned v = threadIdx.x + 100;
    unsigned a = __builtin_amdgcn_update_dpp(7u, v, 0x138, 0xf, 0xf, false);  // wave_shr:1
    unsigned b = __builtin_amdgcn_update_dpp(9u, v, 0x130, 0xf, 0xf, false);  // wave_shl:1
    p[threadIdx.x] = a;
    p[64 + threadIdx.x] = b;
}
int main() {
    unsigned* d; unsigned h[128];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("wave_shr:1 :"); for (int i = 0; i < 64; ++i) printf(" %u", h[i]); printf("\n");
    printf("wave_shl:1 :"); for (int i = 0; i < 64; ++i) printf(" %u", h[64 + i]); printf("\n");
    return 0;
}
