// Issue rate of the integer vector instructions the ORB kernels are made of, on gfx950, 8 waves per SIMD (everything resident at once):
// hipcc -O3 --offload-arch=gfx950 tools/probes/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
template <int OP>
__global__ __launch_bounds__(512) void k(unsigned* out, int iters, unsigned seed) {
    unsigned a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 2654435761u + i * 40503u + seed;
    unsigned b = seed | 0x01020304u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) a[i] = __builtin_amdgcn_udot4(a[i], b, a[(i + 3) & 7], false);
                if (OP == 1) a[i] = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, a[i]), __builtin_bit_cast(us2, b), a[(i + 3) & 7], false);
                if (OP == 2) a[i] = __builtin_amdgcn_perm(a[i], a[(i + 3) & 7], b);
                if (OP == 3) a[i] = __builtin_amdgcn_alignbyte(a[i], a[(i + 3) & 7], 1);
                if (OP == 4) a[i] = a[i] * 18u + a[(i + 3) & 7];                       // v_mad_u32_u24 or v_mul_lo + add
                if (OP == 5) a[i] = (a[i] >> 3) | (a[(i + 3) & 7] << 16);                 // v_lshl_or / alignbit
                if (OP == 6) a[i] = (unsigned)__builtin_amdgcn_update_dpp((int)a[(i + 3) & 7], (int)a[i], 0x138, 0xf, 0xf, false);
                if (OP == 7) a[i] = a[i] + a[(i + 3) & 7];
                if (OP == 8) a[i] = min(a[i], a[(i + 3) & 7]) ^ b;                         // 2 ops
                if (OP == 9) { float f = __builtin_bit_cast(float, a[i]); f = __builtin_fmaf(f, 1.0001f, 0.5f); a[i] = __builtin_bit_cast(unsigned, f); }
                if (OP == 10) a[i] = __shfl_up(a[i], 1, 64);
            }
    }
    unsigned s = 0;
    for (int i = 0; i < 8; ++i) s ^= a[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int OP> void run(const char* name, unsigned* d, int per_iter) {
    const int blocks = 256 * 4, iters = 2000;   // 4 blocks of 8 waves per CU: 8 waves per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(512), 0, 0, d, 10, 1u);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(512), 0, 0, d, iters, 1u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double wave_instr = (double)blocks * 8 * iters * 32 * per_iter;     // per launch
    const double per_simd = wave_instr / (256.0 * 4);
    printf("%-28s %8.3f ms  -> %.2f ns per wave-instruction per SIMD (%.2f cycles at 2.4 GHz)\n", name, ms, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
}
int main() {
    unsigned* d; hipMalloc(&d, 256 * 4 * 512 * 4);
    run<9>("v_fma_f32", d, 1); run<7>("v_add_u32", d, 1); run<0>("v_dot4_u32_u8", d, 1); run<1>("v_dot2_u32_u16", d, 1); run<2>("v_perm_b32", d, 1);
    run<3>("v_alignbyte_b32", d, 1); run<4>("mul + add (u32)", d, 1); run<5>("shift | shift", d, 1); run<6>("v_mov_dpp wave_shr:1", d, 1); run<8>("min, xor (2 ops)", d, 2);
    run<10>("__shfl_up(., 1)", d, 1);
    return 0;
}
