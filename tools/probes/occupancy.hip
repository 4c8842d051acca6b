// How many wavefronts of a register-light kernel a CU really holds: every wavefront spins for N ticks; blocks = 256 CUs x 8 x 4.
// hipcc -O3 --offload-arch=gfx950 tools/probes/occupancy.hip -o /tmp/occupancy && /tmp/occupancy
#include <hip/hip_runtime.h>
#include <cstdio>
struct Big { int v[160]; };   // 640 bytes of kernel arguments, like k_blur7_strips
__global__ __launch_bounds__(256) void spin(unsigned long long n, unsigned* out) {
    const unsigned long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < n) {}
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = 1;
}
__global__ __launch_bounds__(256) void spin_big(Big b, unsigned long long n, unsigned* out) {
    const unsigned long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < n) {}
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = b.v[blockIdx.y & 127];
}
__global__ __launch_bounds__(256) void clock_probe(unsigned long long* out) {
    if (threadIdx.x == 0) { const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64(); while (wall_clock64() - w0 < 100000) {} out[0] = __builtin_readcyclecounter() - t0; out[1] = wall_clock64() - w0; }
}
int main() {
    unsigned* d; hipMalloc(&d, 64); unsigned long long* dc; hipMalloc(&dc, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(clock_probe, dim3(1), dim3(64), 0, 0, dc); unsigned long long hc[2]; hipMemcpy(hc, dc, 16, hipMemcpyDeviceToHost);
    printf("s_memtime ticks per wall_clock64 tick (100 MHz): %.2f -> s_memtime runs at %.0f MHz\n", (double)hc[0] / hc[1], 100.0 * hc[0] / hc[1]);
    for (unsigned long long n : {20000ull, 200000ull}) for (int wg_per_cu : {1, 2, 4, 8, 16}) {
        const int blocks = 256 * wg_per_cu;
        hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, 0, n, d);
        hipEventRecord(e0);
        hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, 0, n, d);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        Big b{};
        hipEventRecord(e0);
        hipLaunchKernelGGL(spin_big, dim3(blocks), dim3(256), 0, 0, b, n, d);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms2; hipEventElapsedTime(&ms2, e0, e1);
        printf("spin %7llu ticks, %2d workgroups of 4 waves per CU (%5d blocks): %.3f ms; with 640 B of arguments %.3f ms\n", n, wg_per_cu, blocks, ms, ms2);
    }
    return 0;
}
