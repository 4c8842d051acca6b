// Host algebra of the iterated error-state Kalman filter of the LiDAR-inertial front end:
//   esekf::predict / update_iterated_dyn_share_modified   SF/include/IKFoM_toolkit/esekfom/esekfom.hpp:281-392, 1621-1932
//   get_f / df_dx / df_dw                                 SF/src/use-ikfom.cpp:45-91
//   MTK::S2 / SO3 charts, A_matrix                        SF/include/IKFoM_toolkit/mtk/types/{S2,SOn}.hpp, mtk/src/mtkmath.hpp
// 23 x 23 matrices, once per IMU sample / filter iteration: host work by design.  The data-parallel part of the update (the
// measurement rows and their normal equations over ~10^4 points) is k_eskf_normal / k_eskf_refit in lidar_kernels.hip.
// State order of the error vector: pos 0, rot 3, offset_R_L_I 6, offset_T_L_I 9, vel 12, bg 15, ba 18, grav 21 (2 dof).
// The reference's integer division `scalar(1/2)` (rotation blocks of F_x1, exponential inside S2_Mx) is kept: those factors are
// the identity.
#pragma once
#include <array>
#include <cmath>
#include <cstring>

#include "../../include/tc2li_hip.h"

namespace tc2li {
namespace eskf {

constexpr int kN = 23, kW = 12;
constexpr double kTol = 1e-11;               // MTK::tolerance<double>()
constexpr double kGravLen = 98090.0 / 10000.0;  // S2<double, 98090, 10000, 1>

template <int R, int C>
struct Mat {
    double a[R * C];
    double& operator()(int r, int c) { return a[r * C + c]; }
    double operator()(int r, int c) const { return a[r * C + c]; }
    static Mat zero() { Mat m; for (double& v : m.a) v = 0.0; return m; }
    static Mat identity() { Mat m = zero(); for (int i = 0; i < (R < C ? R : C); ++i) m(i, i) = 1.0; return m; }
};
template <int R, int K, int C>
inline Mat<R, C> mul(const Mat<R, K>& x, const Mat<K, C>& y) {
    Mat<R, C> o;
    for (int r = 0; r < R; ++r)
        for (int c = 0; c < C; ++c) { double s = 0; for (int k = 0; k < K; ++k) s += x(r, k) * y(k, c); o(r, c) = s; }
    return o;
}
template <int R, int K, int C>
inline Mat<R, C> mul_t(const Mat<R, K>& x, const Mat<C, K>& y) {  // x * y^T
    Mat<R, C> o;
    for (int r = 0; r < R; ++r)
        for (int c = 0; c < C; ++c) { double s = 0; for (int k = 0; k < K; ++k) s += x(r, k) * y(c, k); o(r, c) = s; }
    return o;
}
template <int R, int C>
inline Mat<C, R> transpose(const Mat<R, C>& x) { Mat<C, R> o; for (int r = 0; r < R; ++r) for (int c = 0; c < C; ++c) o(c, r) = x(r, c); return o; }
using M3 = Mat<3, 3>;
using Cov = Mat<kN, kN>;

inline M3 m3_from(const double* p) { M3 m; std::memcpy(m.a, p, sizeof(m.a)); return m; }
inline M3 skew(const double v[3]) { M3 m = M3::zero(); m(0, 1) = -v[2]; m(0, 2) = v[1]; m(1, 0) = v[2]; m(1, 2) = -v[0]; m(2, 0) = -v[1]; m(2, 1) = v[0]; return m; }

inline M3 rodrigues(const double v[3]) {  // rotation by |v| about v: what SO3::exp(v, 1) / S2::boxplus build
    const double n2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2], n = std::sqrt(n2);
    const M3 K = skew(v), K2 = mul(K, K);
    double a, b;
    if (n < 1e-7) { a = 1.0 - n2 / 6.0; b = 0.5 - n2 / 24.0; }
    else { a = std::sin(n) / n; b = (1.0 - std::cos(n)) / n2; }
    M3 R = M3::identity();
    for (int i = 0; i < 9; ++i) R.a[i] += a * K.a[i] + b * K2.a[i];
    return R;
}

inline M3 A_matrix(const double v[3]) {  // mtkmath.hpp
    const double sq = v[0] * v[0] + v[1] * v[1] + v[2] * v[2], n = std::sqrt(sq);
    M3 A = M3::identity();
    if (n < kTol) return A;
    const M3 K = skew(v), K2 = mul(K, K);
    const double a = (1 - std::cos(n)) / sq, b = (1 - std::sin(n) / n) / sq;
    for (int i = 0; i < 9; ++i) A.a[i] += a * K.a[i] + b * K2.a[i];
    return A;
}

// log of (other^T * self) through the quaternion of that matrix (SO3::boxminus -> MTK::log(w, vec, 2, periodic))
inline void so3_minus(const double* self, const double* other, double out[3]) {
    const M3 R = mul(transpose(m3_from(other)), m3_from(self));
    double q[4];
    const double t = R(0, 0) + R(1, 1) + R(2, 2);
    if (t > 0) {
        double s = std::sqrt(t + 1.0);
        q[3] = 0.5 * s;
        s = 0.5 / s;
        q[0] = (R(2, 1) - R(1, 2)) * s; q[1] = (R(0, 2) - R(2, 0)) * s; q[2] = (R(1, 0) - R(0, 1)) * s;
    } else {
        int i = 0;
        if (R(1, 1) > R(0, 0)) i = 1;
        if (R(2, 2) > R(i, i)) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        double s = std::sqrt(R(i, i) - R(j, j) - R(k, k) + 1.0);
        q[i] = 0.5 * s;
        s = 0.5 / s;
        q[3] = (R(k, j) - R(j, k)) * s;
        q[j] = (R(j, i) + R(i, j)) * s;
        q[k] = (R(k, i) + R(i, k)) * s;
    }
    double nv = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
    if (nv < kTol) nv = kTol;
    const double s = 2.0 / nv * std::atan(nv / q[3]);
    for (int c = 0; c < 3; ++c) out[c] = s * q[c];
}

// ---- S2 chart (S2.hpp, S2_typ == 1) -------------------------------------------------------------------------------------------
inline Mat<3, 2> s2_Bx(const double g[3]) {
    Mat<3, 2> B = Mat<3, 2>::zero();
    if (g[0] + kGravLen > kTol) {
        const double d = kGravLen + g[0];
        B(0, 0) = -g[1]; B(0, 1) = -g[2];
        B(1, 0) = kGravLen - g[1] * g[1] / d; B(1, 1) = -g[2] * g[1] / d;
        B(2, 0) = -g[2] * g[1] / d; B(2, 1) = kGravLen - g[2] * g[2] / d;
        for (double& v : B.a) v /= kGravLen;
    } else {
        B(1, 1) = -1; B(2, 0) = 1;
    }
    return B;
}
inline Mat<2, 3> s2_Nx_yy(const double g[3]) {
    Mat<2, 3> N = mul(transpose(s2_Bx(g)), skew(g));
    for (double& v : N.a) v = 1 / kGravLen / kGravLen * v;
    return N;
}
inline Mat<3, 2> s2_Mx(const double g[3], const double delta[2]) {
    const Mat<3, 2> B = s2_Bx(g);
    Mat<3, 2> M;
    if (std::sqrt(delta[0] * delta[0] + delta[1] * delta[1]) < kTol) {
        M = mul(skew(g), B);
    } else {
        double Bu[3];
        for (int r = 0; r < 3; ++r) Bu[r] = B(r, 0) * delta[0] + B(r, 1) * delta[1];
        M = mul(mul_t(skew(g), A_matrix(Bu)), B);  // exp(Bu, scalar(1/2)) = identity
    }
    for (double& v : M.a) v = -v;
    return M;
}

inline void boxplus(tc2li_imu_state& x, const double d[kN]) {
    for (int k = 0; k < 3; ++k) { x.pos[k] += d[k]; x.offset_T_L_I[k] += d[9 + k]; x.vel[k] += d[12 + k]; x.bg[k] += d[15 + k]; x.ba[k] += d[18 + k]; }
    const M3 R = mul(m3_from(x.rot), rodrigues(d + 3)), Ro = mul(m3_from(x.offset_R_L_I), rodrigues(d + 6));
    std::memcpy(x.rot, R.a, sizeof(R.a));
    std::memcpy(x.offset_R_L_I, Ro.a, sizeof(Ro.a));
    const Mat<3, 2> B = s2_Bx(x.grav);
    double Bu[3], g[3];
    for (int r = 0; r < 3; ++r) Bu[r] = B(r, 0) * d[21] + B(r, 1) * d[22];
    const M3 E = rodrigues(Bu);
    for (int r = 0; r < 3; ++r) g[r] = E(r, 0) * x.grav[0] + E(r, 1) * x.grav[1] + E(r, 2) * x.grav[2];
    std::memcpy(x.grav, g, sizeof(g));
}

inline void boxminus(const tc2li_imu_state& x, const tc2li_imu_state& o, double d[kN]) {
    for (int k = 0; k < 3; ++k) { d[k] = x.pos[k] - o.pos[k]; d[9 + k] = x.offset_T_L_I[k] - o.offset_T_L_I[k]; d[12 + k] = x.vel[k] - o.vel[k]; d[15 + k] = x.bg[k] - o.bg[k]; d[18 + k] = x.ba[k] - o.ba[k]; }
    so3_minus(x.rot, o.rot, d + 3);
    so3_minus(x.offset_R_L_I, o.offset_R_L_I, d + 6);
    const M3 Hx = skew(x.grav);
    double hv[3];
    for (int r = 0; r < 3; ++r) hv[r] = Hx(r, 0) * o.grav[0] + Hx(r, 1) * o.grav[1] + Hx(r, 2) * o.grav[2];
    const double v_sin = std::sqrt(hv[0] * hv[0] + hv[1] * hv[1] + hv[2] * hv[2]);
    const double v_cos = x.grav[0] * o.grav[0] + x.grav[1] * o.grav[1] + x.grav[2] * o.grav[2];
    const double theta = std::atan2(v_sin, v_cos);
    if (v_sin < kTol) {
        d[21] = std::fabs(theta) > kTol ? 3.1415926 : 0.0;
        d[22] = 0.0;
        return;
    }
    const Mat<3, 2> B = s2_Bx(o.grav);
    const M3 Ho = skew(o.grav);
    double t[3];
    for (int r = 0; r < 3; ++r) t[r] = Ho(r, 0) * x.grav[0] + Ho(r, 1) * x.grav[1] + Ho(r, 2) * x.grav[2];
    for (int c = 0; c < 2; ++c) d[21 + c] = theta / v_sin * (B(0, c) * t[0] + B(1, c) * t[1] + B(2, c) * t[2]);
}

// rows / columns [idx, idx + D) of a 23 x 23 matrix under a D x D chart correction T:  M <- T M,  M <- M T^T
template <int D>
inline void rows_apply(Cov& M, int idx, const Mat<D, D>& T) {
    for (int c = 0; c < kN; ++c) {
        double in[D], out[D];
        for (int r = 0; r < D; ++r) in[r] = M(idx + r, c);
        for (int r = 0; r < D; ++r) { double s = 0; for (int k = 0; k < D; ++k) s += T(r, k) * in[k]; out[r] = s; }
        for (int r = 0; r < D; ++r) M(idx + r, c) = out[r];
    }
}
template <int D>
inline void cols_apply(Cov& M, int idx, const Mat<D, D>& T) {
    for (int r = 0; r < kN; ++r) {
        double in[D], out[D];
        for (int c = 0; c < D; ++c) in[c] = M(r, idx + c);
        for (int c = 0; c < D; ++c) { double s = 0; for (int k = 0; k < D; ++k) s += in[k] * T(c, k); out[c] = s; }
        for (int c = 0; c < D; ++c) M(r, idx + c) = out[c];
    }
}

// general inverse by partial-pivot LU (what Eigen's fixed-size inverse() does above 4 x 4); n x n row-major in / out
inline bool lu_inverse(const double* A_, int n, double* inv) {
    std::array<double, kN * kN> A;
    std::array<int, kN> perm;
    std::array<double, kN> y;
    if (n > kN) return false;
    std::memcpy(A.data(), A_, sizeof(double) * n * n);
    for (int i = 0; i < n; ++i) perm[i] = i;
    for (int k = 0; k < n; ++k) {
        int p = k;
        double best = std::fabs(A[k * n + k]);
        for (int r = k + 1; r < n; ++r) if (std::fabs(A[r * n + k]) > best) { best = std::fabs(A[r * n + k]); p = r; }
        if (best == 0) return false;
        if (p != k) { for (int c = 0; c < n; ++c) std::swap(A[k * n + c], A[p * n + c]); std::swap(perm[k], perm[p]); }
        for (int r = k + 1; r < n; ++r) {
            const double f = A[r * n + k] / A[k * n + k];
            A[r * n + k] = f;
            for (int c = k + 1; c < n; ++c) A[r * n + c] -= f * A[k * n + c];
        }
    }
    for (int col = 0; col < n; ++col) {
        for (int r = 0; r < n; ++r) {
            double s = perm[r] == col ? 1.0 : 0.0;
            for (int c = 0; c < r; ++c) s -= A[r * n + c] * y[c];
            y[r] = s;
        }
        for (int r = n - 1; r >= 0; --r) {
            double s = y[r];
            for (int c = r + 1; c < n; ++c) s -= A[r * n + c] * inv[c * n + col];
            inv[r * n + col] = s / A[r * n + r];
        }
    }
    return true;
}

// esekf::predict: one IMU step of state and covariance
inline void predict(tc2li_imu_state& x, Cov& P, const Mat<kW, kW>& Q, const double acc[3], const double gyr[3], double dt) {
    double omega[3], acc_[3], a_in[3];
    for (int k = 0; k < 3; ++k) { omega[k] = gyr[k] - x.bg[k]; acc_[k] = acc[k] - x.ba[k]; }
    const M3 R = m3_from(x.rot);
    for (int r = 0; r < 3; ++r) a_in[r] = R(r, 0) * acc_[0] + R(r, 1) * acc_[1] + R(r, 2) * acc_[2];
    const double zero2[2] = {0, 0};
    const M3 RHa = mul(R, skew(acc_));
    const Mat<3, 2> grav_matrix = s2_Mx(x.grav, zero2);
    const tc2li_imu_state before = x;
    // x oplus f dt
    for (int k = 0; k < 3; ++k) x.pos[k] += x.vel[k] * dt;
    const double wdt[3] = {omega[0] * dt, omega[1] * dt, omega[2] * dt};
    const M3 Rn = mul(R, rodrigues(wdt));
    for (int k = 0; k < 3; ++k) x.vel[k] += (a_in[k] + x.grav[k]) * dt;
    std::memcpy(x.rot, Rn.a, sizeof(Rn.a));
    // f_x_final, f_w_final in error-state rows (the rows of offset_R, offset_T, bg, ba and grav are zero: their f is constant)
    Cov fx = Cov::zero();
    Mat<kN, kW> fw = Mat<kN, kW>::zero();
    for (int r = 0; r < 3; ++r) {
        fx(r, 12 + r) = 1.0;                                   // d pos / d vel
        for (int c = 0; c < 3; ++c) { fx(12 + r, 3 + c) = -RHa(r, c); fx(12 + r, 18 + c) = -R(r, c); fw(12 + r, 3 + c) = -R(r, c); }
        for (int c = 0; c < 2; ++c) fx(12 + r, 21 + c) = grav_matrix(r, c);
        fw(15 + r, 6 + r) = 1.0;
        fw(18 + r, 9 + r) = 1.0;
    }
    const double seg[3] = {-1 * omega[0] * dt, -1 * omega[1] * dt, -1 * omega[2] * dt};
    const M3 A = A_matrix(seg);  // rows of rot: A * (-I) in the bg columns of f_x and the ng columns of f_w
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { fx(3 + r, 15 + c) = -A(r, c); fw(3 + r, c) = -A(r, c); }
    Cov F = Cov::identity();
    const Mat<2, 2> G2 = mul(s2_Nx_yy(x.grav), s2_Mx(before.grav, zero2));  // Nx * identity * Mx
    for (int r = 0; r < 2; ++r) for (int c = 0; c < 2; ++c) F(21 + r, 21 + c) = G2(r, c);
    for (int i = 0; i < kN * kN; ++i) F.a[i] += fx.a[i] * dt;
    Mat<kN, kW> G = fw;
    for (double& v : G.a) v = dt * v;
    const Cov FPFt = mul_t(mul(F, P), F), GQGt = mul_t(mul(G, Q), G);
    for (int i = 0; i < kN * kN; ++i) P.a[i] = FPFt.a[i] + GQGt.a[i];
}

}  // namespace eskf
}  // namespace tc2li
