// Checks the fragment layout of v_mfma_f64_16x16x4_f64 on gfx950: D = A(16x4) * B(4x16) + C.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ void k(const double* A, const double* B, double* D) {
    const int l = threadIdx.x;
    const double a = A[(l % 16) * 4 + (l / 16)];  // A[i][k], i = l%16, k = l/16
    const double b = B[(l / 16) * 16 + (l % 16)];  // B[k][j], k = l/16, j = l%16
    v4d c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[l * 4 + r] = c[r];
}
int main() {
    std::vector<double> A(64), B(64), D(256), R(256, 0.0);
    for (int i = 0; i < 64; ++i) { A[i] = 1 + i * 0.37; B[i] = 2 - i * 0.11; }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 4; ++k) R[i * 16 + j] += A[i * 4 + k] * B[k * 16 + j];
    double *dA, *dB, *dD;
    hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, 2048);
    hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost);
    // hypothesis 1: D[lane][r] = R[4*(lane/16) + r][lane%16]
    int bad1 = 0, bad2 = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
        if (fabs(D[l * 4 + r] - R[(4 * (l / 16) + r) * 16 + (l % 16)]) > 1e-9) bad1++;
        if (fabs(D[l * 4 + r] - R[((l / 16) + 4 * r) * 16 + (l % 16)]) > 1e-9) bad2++;
    }
    printf("layout rows=4*(l/16)+r: bad=%d ; rows=(l/16)+4*r: bad=%d\n", bad1, bad2);
    return 0;
}
