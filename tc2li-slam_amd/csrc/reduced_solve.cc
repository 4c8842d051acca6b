// reduced_solve.hpp's factorisation and substitutions: plain host C++ (not a HIP translation unit: the inner loops exist twice, for the x86-64
// baseline and for AVX2, and the one the CPU supports is chosen once -- function multiversioning is not available under -x hip).  Both builds
// do the same IEEE operations on the same eight partial sums in the same order (mul, then add: no fused multiply-add), so the result does not
// depend on which one runs.
#include "reduced_solve.hpp"

namespace tc2li {
namespace {

typedef double v4d __attribute__((vector_size(32), aligned(8)));  // four lanes; on a baseline x86-64 the compiler splits it into two SSE2 registers
// the eight partial sums s0..s7 as two vectors (s0..s3 | s4..s7); closed as ((s0 + s4) + (s2 + s6)) + ((s1 + s5) + (s3 + s7)), then the tail's own sum
#define TC2LI_DOT8_BODY                                                                                                             \
    v4d lo = {0, 0, 0, 0}, hi = {0, 0, 0, 0};                                                                                       \
    int k = 0;                                                                                                                      \
    for (; k + 8 <= len; k += 8) {                                                                                                  \
        lo += *reinterpret_cast<const v4d*>(a + k) * *reinterpret_cast<const v4d*>(b + k);                                          \
        hi += *reinterpret_cast<const v4d*>(a + k + 4) * *reinterpret_cast<const v4d*>(b + k + 4);                                  \
    }                                                                                                                               \
    const v4d s = lo + hi;                                                                                                          \
    double t = 0;                                                                                                                   \
    for (; k < len; ++k) t += a[k] * b[k];                                                                                          \
    return ((s[0] + s[2]) + (s[1] + s[3])) + t;

struct Base { static inline double dot8(const double* a, const double* b, int len) { TC2LI_DOT8_BODY } };
struct Avx2 { __attribute__((target("avx2"))) static inline double dot8(const double* a, const double* b, int len) { TC2LI_DOT8_BODY } };

// rows 0 .. n - 1 of the LDL^T inside the envelope; M holds the matrix on entry, L below the diagonal on return, D the pivots
#define TC2LI_FACTOR_BODY(K)                                                                                                        \
    for (int i = 0; i < n; ++i) {                                                                                                   \
        double* ri = M + (size_t)i * n;                                                                                             \
        const int fi = first[i];                                                                                                    \
        /* y_j = M_ij - sum_k y_k L_jk  (y = L_i D): row i against every row before it, inside both envelopes */                  \
        for (int j = fi; j < i; ++j) {                                                                                              \
            const int k0 = first[j] > fi ? first[j] : fi;                                                                           \
            if (k0 < j) ri[j] -= K::dot8(ri + k0, M + (size_t)j * n + k0, j - k0);                                                  \
        }                                                                                                                           \
        double d = ri[i];                                                                                                           \
        for (int j = fi; j < i; ++j) {                                                                                              \
            const double y = ri[j], l = y / D[j];                                                                                   \
            d -= y * l;                                                                                                             \
            ri[j] = l;                                                                                                              \
        }                                                                                                                           \
        if (!(d == d) || d == 0.0 || d - d != 0.0) return false;                                                                    \
        D[i] = d;                                                                                                                   \
    }                                                                                                                               \
    return true;

bool factor_base(double* M, double* D, const int* first, int n) { TC2LI_FACTOR_BODY(Base) }
__attribute__((target("avx2"))) bool factor_avx2(double* M, double* D, const int* first, int n) { TC2LI_FACTOR_BODY(Avx2) }

const bool kHasAvx2 = __builtin_cpu_supports("avx2");

}  // namespace

bool ReducedSolver::factorise(const double* Hi, const double* S, double lambda) {
    for (int r = 0; r < ni; ++r) {
        const double* src = Hi + (size_t)(np + r) * n + np;
        double* dst = M.data() + (size_t)r * n;
        for (int c = first[r]; c <= r; ++c) dst[c] = src[c];
        dst[r] += lambda;
    }
    for (int r = 0; r < np; ++r) {
        double* dst = M.data() + (size_t)(ni + r) * n;
        for (int c = first[ni + r]; c < ni; ++c) dst[c] = Hi[(size_t)(np + c) * n + r];
        const double* hp = Hi + (size_t)r * n;
        const double* sp = S + (size_t)r * np;
        for (int c = 0; c <= r; ++c) dst[ni + c] = hp[c] + sp[c];
    }
    return kHasAvx2 ? factor_avx2(M.data(), D.data(), first.data(), n) : factor_base(M.data(), D.data(), first.data(), n);
}

void ReducedSolver::solve(const double* rhs, double* x) {
    for (int j = 0; j < n; ++j) z[to_solver(j)] = rhs[j];
    for (int i = 0; i < n; ++i) {
        const int fi = first[i];
        if (fi < i) z[i] -= Base::dot8(M.data() + (size_t)i * n + fi, z.data() + fi, i - fi);
    }
    for (int i = 0; i < n; ++i) z[i] /= D[i];
    for (int i = n - 1; i >= 0; --i) {       // column sweep: x_i is final, every row of its envelope takes its term
        const double xi = z[i];
        const double* ri = M.data() + (size_t)i * n;
        for (int k = first[i]; k < i; ++k) z[k] -= ri[k] * xi;
    }
    for (int j = 0; j < n; ++j) x[j] = z[to_solver(j)];
}

}  // namespace tc2li
