// Measurement helpers behind tc2li_diag_* (include/tc2li_hip.h): the peaks this GPU actually reaches, which bench.py prices the
// Schur GEMM (f64 matrix pipe) and the streaming kernels (HBM copy) against.  SURVEY.md section 8d asks for the f64 peak to be
// measured: MI355X_MICROARCH.md lists no f64 MFMA rate.  Not part of the hot path.
#include <hip/hip_runtime.h>

#include "common.hpp"

namespace tc2li {

typedef double v4d __attribute__((ext_vector_type(4)));

// Back-to-back v_mfma_f64_16x16x4_f64 on 8 independent accumulators per wavefront (no memory traffic inside the loop).
__global__ __launch_bounds__(256) void k_diag_mfma_f64(int iters, double* __restrict__ sink) {
    const double a = 1.0 + 1e-9 * threadIdx.x, b = 1.0 - 1e-9 * threadIdx.x;
    v4d acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = v4d{0.0, 0.0, 0.0, 0.0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678) sink[0] = s;  // never true: keeps the loop alive
}

// The same loops with the shader clock (s_memtime) and the constant 100 MHz counter (s_memrealtime) read around them by one lane:
// cycles per microsecond = the clock the chip holds under this load (the f64 matrix peak of the data sheet assumes 2.4 GHz).
template <bool MFMA>
__global__ __launch_bounds__(256) void k_diag_clock(int iters, long long* __restrict__ out, double* __restrict__ sink) {
    const long long c0 = clock64(), w0 = wall_clock64();
    double s = 0;
    if (MFMA) {
        const double a = 1.0 + 1e-9 * threadIdx.x, b = 1.0 - 1e-9 * threadIdx.x;
        v4d acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = v4d{0.0, 0.0, 0.0, 0.0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else {
        double x[8];
        const double m = 1.0 + 1e-12 * threadIdx.x, c = 1e-9;
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = 1.0 + i;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = fma(x[i], m, c);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) s += x[i];
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    if (s == 12345.678) sink[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; }
}

// Plain f64 FMA on the vector pipe, 8 independent chains per lane (the rate the non-matrix BA kernels are bounded by).
__global__ __launch_bounds__(256) void k_diag_fma_f64(int iters, double* __restrict__ sink) {
    double x[8];
    const double m = 1.0 + 1e-12 * threadIdx.x, c = 1e-9;
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = 1.0 + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = fma(x[i], m, c);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i];
    if (s == 12345.678) sink[0] = s;
}

__global__ __launch_bounds__(256) void k_diag_copy(const float4* __restrict__ src, float4* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

}  // namespace tc2li

using namespace tc2li;

extern "C" int tc2li_diag_clocks(double* mfma_loop_ghz, double* fma_loop_ghz) {
    if (!device_ready()) return TC2LI_ERR_NO_DEVICE;
    hipStream_t st = private_stream();
    DevBuf<double> sink;
    DevBuf<long long> d_out;
    TC2LI_HIP_CHECK(sink.alloc(8));
    TC2LI_HIP_CHECK(d_out.alloc(2));
    hipDeviceProp_t prop;
    int dev = 0;
    TC2LI_HIP_CHECK(hipGetDevice(&dev));
    TC2LI_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    const int cus = prop.multiProcessorCount;
    long long h[2];
    for (int which = 0; which < 2; ++which) {
        double* dst = which == 0 ? mfma_loop_ghz : fma_loop_ghz;
        if (!dst) continue;
        for (int rep = 0; rep < 2; ++rep) {  // the second run: clocks settled under the load
            if (which == 0) TC2LI_LAUNCH(k_diag_clock<true>, dim3(cus * 4), dim3(256), 0, st, 4096, d_out.p, sink.p);
            else TC2LI_LAUNCH(k_diag_clock<false>, dim3(cus * 8), dim3(256), 0, st, 8192, d_out.p, sink.p);
            TC2LI_HIP_CHECK(hipGetLastError());
            TC2LI_HIP_CHECK(copy_sync(h, d_out.p, sizeof(h), hipMemcpyDeviceToHost, st));
        }
        *dst = h[1] > 0 ? (double)h[0] / ((double)h[1] * 10.0) : 0.0;  // shader cycles per nanosecond (s_memrealtime ticks are 10 ns)
    }
    return TC2LI_OK;
}

extern "C" int tc2li_diag_peaks(double* mfma_f64_tflops, double* fma_f64_tflops, double* hbm_copy_gbps) {
    if (!device_ready()) return TC2LI_ERR_NO_DEVICE;
    hipStream_t st = private_stream();
    hipEvent_t e0, e1;
    TC2LI_HIP_CHECK(hipEventCreate(&e0));
    TC2LI_HIP_CHECK(hipEventCreate(&e1));
    DevBuf<double> sink;
    TC2LI_HIP_CHECK(sink.alloc(8));
    hipDeviceProp_t prop;
    int dev = 0;
    TC2LI_HIP_CHECK(hipGetDevice(&dev));
    TC2LI_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    const int cus = prop.multiProcessorCount;
    auto timed = [&](auto&& launch, float& best) -> int {
        best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {  // first repetition warms up
            TC2LI_HIP_CHECK(hipEventRecord(e0, st));
            launch();
            TC2LI_HIP_CHECK(hipEventRecord(e1, st));
            TC2LI_HIP_CHECK(hipEventSynchronize(e1));
            float ms = 0;
            TC2LI_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best) best = ms;
        }
        return TC2LI_OK;
    };
    float ms = 0;
    if (mfma_f64_tflops) {
        const int iters = 4096, blocks = cus * 4;  // 4 blocks of 4 wavefronts per CU: 4 waves per SIMD
        int rc = timed([&] { TC2LI_LAUNCH(k_diag_mfma_f64, dim3(blocks), dim3(256), 0, st, iters, sink.p); }, ms);
        if (rc != TC2LI_OK) return rc;
        *mfma_f64_tflops = (double)blocks * 4 * iters * 8 * 2048.0 / (ms * 1e-3) / 1e12;  // 16 x 16 x 4 x 2 FLOP per instruction
    }
    if (fma_f64_tflops) {
        const int iters = 8192, blocks = cus * 8;
        int rc = timed([&] { TC2LI_LAUNCH(k_diag_fma_f64, dim3(blocks), dim3(256), 0, st, iters, sink.p); }, ms);
        if (rc != TC2LI_OK) return rc;
        *fma_f64_tflops = (double)blocks * 256 * iters * 8 * 2.0 / (ms * 1e-3) / 1e12;
    }
    if (hbm_copy_gbps) {
        const size_t n = (size_t)1 << 26;  // 1 GiB in + 1 GiB out: far beyond the 256 MiB Infinity Cache
        DevBuf<float4> a, b;
        TC2LI_HIP_CHECK(a.alloc(n));
        TC2LI_HIP_CHECK(b.alloc(n));
        TC2LI_HIP_CHECK(hipMemsetAsync(a.p, 1, n * sizeof(float4), st));
        int rc = timed([&] { TC2LI_LAUNCH(k_diag_copy, dim3(cus * 16), dim3(256), 0, st, a.p, b.p, n); }, ms);
        if (rc != TC2LI_OK) return rc;
        *hbm_copy_gbps = 2.0 * n * sizeof(float4) / (ms * 1e-3) / 1e9;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return TC2LI_OK;
}
