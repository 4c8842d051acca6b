// Host side of the inertial edges of Optimizer::LocalInertialBA (a few tens of edges per window: SURVEY.md section 8a row c6
// keeps their arithmetic on the host; the projection edges run on the GPU):
//   EdgeInertial (information, computeError, linearizeOplus)   SF/src/G2oTypes.cc:499-601
//   EdgeGyroRW / EdgeAccRW                                     SF/include/G2oTypes.h:645-714
//   LogSO3, RightJacobianSO3, InverseRightJacobianSO3          SF/src/G2oTypes.cc:807-858
#pragma once
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/tc2li_hip.h"
#include "inertial_math.hpp"

namespace tc2li {

struct ImuVertexState { double v[3], bg[3], ba[3]; };  // VertexVelocity, VertexGyroBias, VertexAccBias of one keyframe

namespace inertial_detail {
inline void hat3(const double* v, double* o) { o[0] = 0; o[1] = -v[2]; o[2] = v[1]; o[3] = v[2]; o[4] = 0; o[5] = -v[0]; o[6] = -v[1]; o[7] = v[0]; o[8] = 0; }
inline void tr3(const double* a, double* o) { for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) o[3 * r + c] = a[3 * c + r]; }
inline void log_so3(const double* R, double* w) {
    const double trc = R[0] + R[4] + R[8];
    w[0] = (R[7] - R[5]) / 2; w[1] = (R[2] - R[6]) / 2; w[2] = (R[3] - R[1]) / 2;
    const double costheta = (trc - 1.0) * 0.5f;
    if (costheta > 1 || costheta < -1) return;
    const double theta = std::acos(costheta), s = std::sin(theta);
    if (std::fabs(s) < 1e-5) return;
    for (int k = 0; k < 3; ++k) w[k] = theta * w[k] / s;
}
inline void jr_so3(const double* v, bool inverse, double* J) {  // RightJacobianSO3 / InverseRightJacobianSO3
    const double d2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2], d = std::sqrt(d2);
    double W[9], W2[9];
    hat3(v, W);
    r3_mul(W, W, W2);
    for (int k = 0; k < 9; ++k) J[k] = k % 4 == 0 ? 1.0 : 0.0;
    if (d < 1e-5) return;
    if (inverse) for (int k = 0; k < 9; ++k) J[k] = J[k] + W[k] / 2 + W2[k] * (1.0 / d2 - (1.0 + std::cos(d)) / (2.0 * d * std::sin(d)));
    else for (int k = 0; k < 9; ++k) J[k] = J[k] - W[k] * (1.0 - std::cos(d)) / d2 + W2[k] * (d - std::sin(d)) / (d2 * d);
}
// inverse of a symmetric positive definite matrix through its Cholesky factor; false if it is not positive definite
inline bool spd_inverse(const double* A, int n, double* inv) {
    std::vector<double> L((size_t)n * n, 0.0), Li((size_t)n * n, 0.0);
    for (int j = 0; j < n; ++j) {
        double d = A[(size_t)j * n + j];
        for (int k = 0; k < j; ++k) d -= L[(size_t)j * n + k] * L[(size_t)j * n + k];
        if (!(d > 0)) return false;
        L[(size_t)j * n + j] = std::sqrt(d);
        for (int i = j + 1; i < n; ++i) {
            double s = A[(size_t)i * n + j];
            for (int k = 0; k < j; ++k) s -= L[(size_t)i * n + k] * L[(size_t)j * n + k];
            L[(size_t)i * n + j] = s / L[(size_t)j * n + j];
        }
    }
    for (int c = 0; c < n; ++c)  // L^-1 by forward substitution, column by column
        for (int i = c; i < n; ++i) {
            double s = i == c ? 1.0 : 0.0;
            for (int k = c; k < i; ++k) s -= L[(size_t)i * n + k] * Li[(size_t)k * n + c];
            Li[(size_t)i * n + c] = s / L[(size_t)i * n + i];
        }
    for (int r = 0; r < n; ++r)
        for (int c = 0; c < n; ++c) {
            double s = 0;
            for (int k = std::max(r, c); k < n; ++k) s += Li[(size_t)k * n + r] * Li[(size_t)k * n + c];
            inv[(size_t)r * n + c] = s;
        }
    return true;
}
// eigenvalues / eigenvectors (columns of V) of a symmetric matrix by cyclic Jacobi rotations
inline void sym_eigen(std::vector<double> A, int n, std::vector<double>& w, std::vector<double>& V) {
    V.assign((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i) V[(size_t)i * n + i] = 1.0;
    for (int sweep = 0; sweep < 100; ++sweep) {
        double off = 0, dg = 0;
        for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) (i == j ? dg : off) += A[(size_t)i * n + j] * A[(size_t)i * n + j];
        if (off <= 1e-30 * dg || off == 0.0) break;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                const double apq = A[(size_t)p * n + q];
                if (apq == 0.0) continue;
                const double theta = (A[(size_t)q * n + q] - A[(size_t)p * n + p]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < n; ++k) { const double x = A[(size_t)k * n + p], y = A[(size_t)k * n + q]; A[(size_t)k * n + p] = c * x - s * y; A[(size_t)k * n + q] = s * x + c * y; }
                for (int k = 0; k < n; ++k) { const double x = A[(size_t)p * n + k], y = A[(size_t)q * n + k]; A[(size_t)p * n + k] = c * x - s * y; A[(size_t)q * n + k] = s * x + c * y; }
                for (int k = 0; k < n; ++k) { const double x = V[(size_t)k * n + p], y = V[(size_t)k * n + q]; V[(size_t)k * n + p] = c * x - s * y; V[(size_t)k * n + q] = s * x + c * y; }
            }
    }
    w.resize(n);
    for (int i = 0; i < n; ++i) w[i] = A[(size_t)i * n + i];
}
}  // namespace inertial_detail

struct InertialLinkHost {
    int kf1 = 0, kf2 = 0;
    bool robust = false;
    const tc2li_preintegrated* pre = nullptr;
    double info[81], infoG[9], infoA[9];

    // EdgeInertial ctor (information = inverse of C(0:9, 0:9), symmetrised, eigenvalues below 1e-12 cleared) and the
    // random-walk informations (inverses of the bias-walk blocks of C), OptimizerWithLidar / Optimizer.cc:1767-1795
    bool prepare(double info_scale) {
        using namespace inertial_detail;
        double C9[81], G[9], A[9], inv[81];
        for (int r = 0; r < 9; ++r) for (int c = 0; c < 9; ++c) C9[9 * r + c] = pre->C[15 * r + c];
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { G[3 * r + c] = pre->C[15 * (9 + r) + 9 + c]; A[3 * r + c] = pre->C[15 * (12 + r) + 12 + c]; }
        if (!spd_inverse(C9, 9, inv) || !spd_inverse(G, 3, infoG) || !spd_inverse(A, 3, infoA)) return false;
        std::vector<double> S(inv, inv + 81), w, V;
        for (int r = 0; r < 9; ++r) for (int c = r + 1; c < 9; ++c) S[9 * r + c] = S[9 * c + r] = (inv[9 * r + c] + inv[9 * c + r]) / 2;
        sym_eigen(S, 9, w, V);
        for (double& x : w) if (x < 1e-12) x = 0;
        for (int r = 0; r < 9; ++r)
            for (int c = 0; c < 9; ++c) {
                double s = 0;
                for (int k = 0; k < 9; ++k) s += V[9 * r + k] * w[k] * V[9 * c + k];
                info[9 * r + c] = s * info_scale;
            }
        return true;
    }

    // EdgeInertial::computeError / linearizeOplus: err (er, ev, ep); J (9 x 24, columns P1 6 | V1 3 | G1 3 | A1 3 | P2 6 | V2 3) or NULL
    void evaluate(const ImuPose& P1, const ImuVertexState& s1, const ImuPose& P2, const ImuVertexState& s2, double err[9], double* J) const {
        using namespace inertial_detail;
        tc2li_imu_bias b1{(float)s1.ba[0], (float)s1.ba[1], (float)s1.ba[2], (float)s1.bg[0], (float)s1.bg[1], (float)s1.bg[2]};
        float dRf[9], dVf[3], dPf[3];
        tc2li_imu_delta(pre, &b1, dRf, dVf, dPf);
        double dR[9], dV[3], dP[3];
        for (int k = 0; k < 9; ++k) dR[k] = dRf[k];
        for (int k = 0; k < 3; ++k) { dV[k] = dVf[k]; dP[k] = dPf[k]; }
        const double dt = pre->dT, g[3] = {0, 0, -(double)9.81f};
        double Rbw1[9], dRt[9], t1[9], eR[9], er[3], dv[3], dp[3], rv[3], rp[3];
        tr3(P1.Rwb, Rbw1); tr3(dR, dRt);
        r3_mul(dRt, Rbw1, t1); r3_mul(t1, P2.Rwb, eR);
        log_so3(eR, er);
        for (int k = 0; k < 3; ++k) { dv[k] = s2.v[k] - s1.v[k] - g[k] * dt; dp[k] = P2.twb[k] - P1.twb[k] - s1.v[k] * dt - g[k] * dt * dt / 2; }
        r3_vec(Rbw1, dv, rv); r3_vec(Rbw1, dp, rp);
        for (int k = 0; k < 3; ++k) { err[k] = er[k]; err[3 + k] = rv[k] - dV[k]; err[6 + k] = rp[k] - dP[k]; }
        if (!J) return;
        memset(J, 0, 9 * 24 * sizeof(double));
        auto put = [&](int r0, int c0, const double* m, double s) { for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) J[24 * (r0 + r) + c0 + c] = s * m[3 * r + c]; };
        const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        double invJr[9], R2t[9], m1[9], m2[9], hv[9], hp[9], dp2[3];
        jr_so3(er, true, invJr);
        tr3(P2.Rwb, R2t);
        r3_mul(invJr, R2t, m1); r3_mul(m1, P1.Rwb, m2);
        put(0, 0, m2, -1.0);
        for (int k = 0; k < 3; ++k) dp2[k] = P2.twb[k] - P1.twb[k] - s1.v[k] * dt - 0.5 * g[k] * dt * dt;
        r3_vec(Rbw1, dp2, rp);
        hat3(rv, hv); hat3(rp, hp);
        put(3, 0, hv, 1.0); put(6, 0, hp, 1.0); put(6, 3, I, -1.0);
        put(3, 6, Rbw1, -1.0); put(6, 6, Rbw1, -dt);
        double JRg[9], JVg[9], JPg[9], JVa[9], JPa[9];
        for (int k = 0; k < 9; ++k) { JRg[k] = pre->JRg[k]; JVg[k] = pre->JVg[k]; JPg[k] = pre->JPg[k]; JVa[k] = pre->JVa[k]; JPa[k] = pre->JPa[k]; }
        const double dbg[3] = {(double)(b1.bwx - pre->bias.bwx), (double)(b1.bwy - pre->bias.bwy), (double)(b1.bwz - pre->bias.bwz)};
        double Jd[3], RJ[9], eRt[9], a1[9], a2[9], a3[9], R12[9];
        r3_vec(JRg, dbg, Jd);
        jr_so3(Jd, false, RJ);
        tr3(eR, eRt);
        r3_mul(invJr, eRt, a1); r3_mul(a1, RJ, a2); r3_mul(a2, JRg, a3);
        put(0, 9, a3, -1.0); put(3, 9, JVg, -1.0); put(6, 9, JPg, -1.0);
        put(3, 12, JVa, -1.0); put(6, 12, JPa, -1.0);
        put(0, 15, invJr, 1.0);
        r3_mul(Rbw1, P2.Rwb, R12);
        put(6, 18, R12, 1.0);
        put(3, 21, Rbw1, 1.0);
    }
};

}  // namespace tc2li
