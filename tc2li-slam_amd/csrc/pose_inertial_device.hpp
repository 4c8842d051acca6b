// Optimizer::PoseInertialOptimizationLastKeyFrame / LastFrame (SF/src/Optimizer.cc:2469-2852, 2854-3270): structures shared by the
// host entry point and the kernel, and the edge arithmetic both sides evaluate (the kernel inside the Gauss-Newton iterations, the
// host once at the final estimate for the Hessian of the new prior):
//   EdgeInertial::computeError / linearizeOplus           SF/src/G2oTypes.cc:517-601
//   IMU::Preintegrated::GetDelta{Rotation,Velocity,Position} (float)   SF/src/ImuTypes.cc:292-316
//   EdgePriorPoseImu::computeError / linearizeOplus       SF/src/G2oTypes.cc:738-767
//   LogSO3, RightJacobianSO3, InverseRightJacobianSO3     SF/src/G2oTypes.cc:807-858
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/tc2li_hip.h"
#include "inertial_math.hpp"

namespace tc2li {

struct PiState { ImuPose P; double v[3], bg[3], ba[3]; };  // VertexPose + VertexVelocity + VertexGyroBias + VertexAccBias

// What the kernel reads of IMU::Preintegrated (floats, as the reference keeps them) + the edge informations the host prepared
struct PiPreint {
    float dT, dR[9], dV[3], dP[3], JRg[9], JVg[9], JVa[9], JPg[9], JPa[9];
    float bias[6];                       // bax bay baz bwx bwy bwz: the bias of the integration
    double info[81], infoG[9], infoA[9];  // EdgeInertial information (eigenvalue clamp applied), random-walk informations
};
struct PiPrior { double Rwb[9], twb[3], vwb[3], bg[3], ba[3], H[225]; };

struct PiProblem {
    PiState cur, other;     // in: the frame / the last keyframe or previous frame
    PiPreint pre;
    PiPrior prior;          // last-frame form
    int32_t edge_off, n_edges, last_frame, rec_init;
};
struct PiResult {
    PiState cur, other;
    double Hv[21];          // sum over the inlier edges of B^T Omega B (upper triangle, row-major)
    int32_t n_bad, n_inliers, solver_failed, pad_;
};

void launch_pose_inertial(const PiProblem* probs, int n, const double* Xw, const BaEdge* edges, const uint8_t* close, const ImuCalib& cal,
                          const CameraD& cam, uint8_t* outlier, double* chi2_scratch, PiResult* results, hipStream_t st);

// ---- float side: the bias-corrected pre-integration -------------------------------------------------------------------------------
__host__ __device__ inline void pi_mul3f(const float* a, const float* b, float* o) {
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) o[3 * r + c] = a[3 * r] * b[c] + a[3 * r + 1] * b[3 + c] + a[3 * r + 2] * b[6 + c];
}
// IMU::NormalizeRotation (float JacobiSVD in the reference): one-sided Jacobi rotations, as csrc/imu_host.cpp
__host__ __device__ inline void pi_normalize_rotation_f(const float* R, float* out) {
    // (round 5: dR exp(JRg dbg) is orthogonal to float precision already -- its polar factor by two Newton steps X <- (X + X^-T) / 2 instead of
    // Jacobi sweeps on one lane; the sweeps remain for anything further from a rotation)
    {
        float e = 0;
        for (int a = 0; a < 3; ++a)
            for (int b = a; b < 3; ++b)
                e = fmaxf(e, fabsf(R[a] * R[b] + R[3 + a] * R[3 + b] + R[6 + a] * R[6 + b] - (a == b ? 1.0f : 0.0f)));
        if (e < 1e-3f) {
            float X[9];
            for (int k = 0; k < 9; ++k) X[k] = R[k];
            for (int it = 0; it < 2; ++it) {
                const float c[9] = {X[4] * X[8] - X[5] * X[7], X[5] * X[6] - X[3] * X[8], X[3] * X[7] - X[4] * X[6],
                                    X[2] * X[7] - X[1] * X[8], X[0] * X[8] - X[2] * X[6], X[1] * X[6] - X[0] * X[7],
                                    X[1] * X[5] - X[2] * X[4], X[2] * X[3] - X[0] * X[5], X[0] * X[4] - X[1] * X[3]};
                const float inv = 1.0f / (X[0] * c[0] + X[1] * c[1] + X[2] * c[2]);
                for (int k = 0; k < 9; ++k) X[k] = 0.5f * (X[k] + c[k] * inv);
            }
            for (int k = 0; k < 9; ++k) out[k] = X[k];
            return;
        }
    }
    float A[9], V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int k = 0; k < 9; ++k) A[k] = R[k];
    for (int sweep = 0; sweep < 30; ++sweep) {
        float off = 0;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                float app = 0, aqq = 0, apq = 0;
                for (int k = 0; k < 3; ++k) { app += A[3 * k + p] * A[3 * k + p]; aqq += A[3 * k + q] * A[3 * k + q]; apq += A[3 * k + p] * A[3 * k + q]; }
                off = fmaxf(off, fabsf(apq) / sqrtf(fmaxf(app * aqq, 1e-30f)));
                if (fabsf(apq) <= 1e-12f * sqrtf(app * aqq)) continue;
                const float tau = (aqq - app) / (2.0f * apq);
                const float t = (tau >= 0 ? 1.0f : -1.0f) / (fabsf(tau) + sqrtf(1.0f + tau * tau));
                const float c = 1.0f / sqrtf(1.0f + t * t), s = t * c;
                for (int k = 0; k < 3; ++k) {
                    const float x = A[3 * k + p], y = A[3 * k + q];
                    A[3 * k + p] = c * x - s * y; A[3 * k + q] = s * x + c * y;
                    const float vx = V[3 * k + p], vy = V[3 * k + q];
                    V[3 * k + p] = c * vx - s * vy; V[3 * k + q] = s * vx + c * vy;
                }
            }
        if (off < 1e-7f) break;
    }
    float U[9];
    for (int c = 0; c < 3; ++c) {
        float n = 0;
        for (int k = 0; k < 3; ++k) n += A[3 * k + c] * A[3 * k + c];
        n = sqrtf(n);
        for (int k = 0; k < 3; ++k) U[3 * k + c] = n > 0 ? A[3 * k + c] / n : (k == c ? 1.0f : 0.0f);
    }
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) out[3 * r + c] = U[3 * r] * V[3 * c] + U[3 * r + 1] * V[3 * c + 1] + U[3 * r + 2] * V[3 * c + 2];
}
__host__ __device__ inline void pi_so3_exp_f(const float v[3], float R[9]) {  // Sophus::SO3f::exp(v).matrix()
    const float th2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    float imag, real;
    if (th2 < 1e-10f * 1e-10f) {
        const float th4 = th2 * th2;
        imag = 0.5f - (1.0f / 48.0f) * th2 + (1.0f / 3840.0f) * th4;
        real = 1.0f - (1.0f / 8.0f) * th2 + (1.0f / 384.0f) * th4;
    } else {
        const float th = sqrtf(th2), half = 0.5f * th;
        imag = sinf(half) / th;
        real = cosf(half);
    }
    const float q[4] = {imag * v[0], imag * v[1], imag * v[2], real};
    const float tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
    const float twx = tx * q[3], twy = ty * q[3], twz = tz * q[3], txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
    const float tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
// GetDeltaRotation / GetDeltaVelocity / GetDeltaPosition at the bias (bg, ba), widened to double as EdgeInertial does
__host__ __device__ inline void pi_delta(const PiPreint& p, const double bg[3], const double ba[3], double dR[9], double dV[3], double dP[3], float dbg_out[3]) {
    const float b[6] = {(float)ba[0], (float)ba[1], (float)ba[2], (float)bg[0], (float)bg[1], (float)bg[2]};
    const float dbg[3] = {b[3] - p.bias[3], b[4] - p.bias[4], b[5] - p.bias[5]};
    const float dba[3] = {b[0] - p.bias[0], b[1] - p.bias[1], b[2] - p.bias[2]};
    float v[3], E[9], RE[9], Rn[9];
    for (int r = 0; r < 3; ++r) v[r] = p.JRg[3 * r] * dbg[0] + p.JRg[3 * r + 1] * dbg[1] + p.JRg[3 * r + 2] * dbg[2];
    pi_so3_exp_f(v, E);
    pi_mul3f(p.dR, E, RE);
    pi_normalize_rotation_f(RE, Rn);
    for (int k = 0; k < 9; ++k) dR[k] = (double)Rn[k];
    for (int r = 0; r < 3; ++r) {
        dV[r] = (double)(p.dV[r] + (p.JVg[3 * r] * dbg[0] + p.JVg[3 * r + 1] * dbg[1] + p.JVg[3 * r + 2] * dbg[2]) +
                         (p.JVa[3 * r] * dba[0] + p.JVa[3 * r + 1] * dba[1] + p.JVa[3 * r + 2] * dba[2]));
        dP[r] = (double)(p.dP[r] + (p.JPg[3 * r] * dbg[0] + p.JPg[3 * r + 1] * dbg[1] + p.JPg[3 * r + 2] * dbg[2]) +
                         (p.JPa[3 * r] * dba[0] + p.JPa[3 * r + 1] * dba[1] + p.JPa[3 * r + 2] * dba[2]));
        dbg_out[r] = dbg[r];
    }
}

// ---- double side ---------------------------------------------------------------------------------------------------------------------
__host__ __device__ inline void pi_hat(const double* v, double* o) { o[0] = 0; o[1] = -v[2]; o[2] = v[1]; o[3] = v[2]; o[4] = 0; o[5] = -v[0]; o[6] = -v[1]; o[7] = v[0]; o[8] = 0; }
__host__ __device__ inline void pi_tr(const double* a, double* o) { for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) o[3 * r + c] = a[3 * c + r]; }
__host__ __device__ inline void pi_log_so3(const double* R, double* w) {
    const double trc = R[0] + R[4] + R[8];
    w[0] = (R[7] - R[5]) / 2; w[1] = (R[2] - R[6]) / 2; w[2] = (R[3] - R[1]) / 2;
    const double costheta = (trc - 1.0) * 0.5f;
    if (costheta > 1 || costheta < -1) return;
    const double theta = acos(costheta), s = sin(theta);
    if (fabs(s) < 1e-5) return;
    for (int k = 0; k < 3; ++k) w[k] = theta * w[k] / s;
}
__host__ __device__ inline void pi_jr_so3(const double* v, bool inverse, double* J) {  // RightJacobianSO3 / InverseRightJacobianSO3
    const double d2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2], d = sqrt(d2);
    double W[9], W2[9];
    pi_hat(v, W);
    r3_mul(W, W, W2);
    for (int k = 0; k < 9; ++k) J[k] = k % 4 == 0 ? 1.0 : 0.0;
    if (d < 1e-5) return;
    if (inverse) for (int k = 0; k < 9; ++k) J[k] = J[k] + W[k] / 2 + W2[k] * (1.0 / d2 - (1.0 + cos(d)) / (2.0 * d * sin(d)));
    else for (int k = 0; k < 9; ++k) J[k] = J[k] - W[k] * (1.0 - cos(d)) / d2 + W2[k] * (d - sin(d)) / (d2 * d);
}

// EdgeInertial between state 1 (P, V, G, A) and state 2 (P, V): err (er, ev, ep); J (9 x 24, columns P1 6 | V1 3 | G1 3 | A1 3 | P2 6 | V2 3)
// or NULL
__host__ __device__ inline void pi_inertial_edge(const PiPreint& pre, const PiState& s1, const PiState& s2, double err[9], double* J) {
    double dR[9], dV[3], dP[3];
    float dbgf[3];
    pi_delta(pre, s1.bg, s1.ba, dR, dV, dP, dbgf);
    const double dt = (double)pre.dT, g[3] = {0, 0, -(double)9.81f};
    double Rbw1[9], dRt[9], t1[9], eR[9], er[3], dv[3], dp[3], rv[3], rp[3];
    pi_tr(s1.P.Rwb, Rbw1); pi_tr(dR, dRt);
    r3_mul(dRt, Rbw1, t1); r3_mul(t1, s2.P.Rwb, eR);
    pi_log_so3(eR, er);
    for (int k = 0; k < 3; ++k) { dv[k] = s2.v[k] - s1.v[k] - g[k] * dt; dp[k] = s2.P.twb[k] - s1.P.twb[k] - s1.v[k] * dt - g[k] * dt * dt / 2; }
    r3_vec(Rbw1, dv, rv); r3_vec(Rbw1, dp, rp);
    for (int k = 0; k < 3; ++k) { err[k] = er[k]; err[3 + k] = rv[k] - dV[k]; err[6 + k] = rp[k] - dP[k]; }
    if (!J) return;
    for (int k = 0; k < 9 * 24; ++k) J[k] = 0;
    const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    double invJr[9], R2t[9], m1[9], m2[9], hv[9], hp[9], dp2[3];
    pi_jr_so3(er, true, invJr);
    pi_tr(s2.P.Rwb, R2t);
    r3_mul(invJr, R2t, m1); r3_mul(m1, s1.P.Rwb, m2);
#define TC2LI_PI_PUT(r0, c0, m, s) for (int r_ = 0; r_ < 3; ++r_) for (int c_ = 0; c_ < 3; ++c_) J[24 * ((r0) + r_) + (c0) + c_] = (s) * (m)[3 * r_ + c_]
    TC2LI_PI_PUT(0, 0, m2, -1.0);
    for (int k = 0; k < 3; ++k) dp2[k] = s2.P.twb[k] - s1.P.twb[k] - s1.v[k] * dt - 0.5 * g[k] * dt * dt;
    r3_vec(Rbw1, dp2, rp);
    pi_hat(rv, hv); pi_hat(rp, hp);
    TC2LI_PI_PUT(3, 0, hv, 1.0); TC2LI_PI_PUT(6, 0, hp, 1.0); TC2LI_PI_PUT(6, 3, I, -1.0);
    TC2LI_PI_PUT(3, 6, Rbw1, -1.0); TC2LI_PI_PUT(6, 6, Rbw1, -dt);
    double JRg[9], JVg[9], JPg[9], JVa[9], JPa[9];
    for (int k = 0; k < 9; ++k) { JRg[k] = pre.JRg[k]; JVg[k] = pre.JVg[k]; JPg[k] = pre.JPg[k]; JVa[k] = pre.JVa[k]; JPa[k] = pre.JPa[k]; }
    const double dbg[3] = {(double)dbgf[0], (double)dbgf[1], (double)dbgf[2]};
    double Jd[3], RJ[9], eRt[9], a1[9], a2[9], a3[9], R12[9];
    r3_vec(JRg, dbg, Jd);
    pi_jr_so3(Jd, false, RJ);
    pi_tr(eR, eRt);
    r3_mul(invJr, eRt, a1); r3_mul(a1, RJ, a2); r3_mul(a2, JRg, a3);
    TC2LI_PI_PUT(0, 9, a3, -1.0); TC2LI_PI_PUT(3, 9, JVg, -1.0); TC2LI_PI_PUT(6, 9, JPg, -1.0);
    TC2LI_PI_PUT(3, 12, JVa, -1.0); TC2LI_PI_PUT(6, 12, JPa, -1.0);
    TC2LI_PI_PUT(0, 15, invJr, 1.0);
    r3_mul(Rbw1, s2.P.Rwb, R12);
    TC2LI_PI_PUT(6, 18, R12, 1.0);
    TC2LI_PI_PUT(3, 21, Rbw1, 1.0);
#undef TC2LI_PI_PUT
}

// EdgePriorPoseImu on state s: error 15 (er, et, ev, ebg, eba); the two non-trivial 3 x 3 Jacobian blocks: Jr = d er / d rotation
// (InverseRightJacobianSO3(er)) and Jt = d et / d translation (Rwb_prior^T Rwb); every other block is the identity on its diagonal
__host__ __device__ inline void pi_prior_edge(const PiPrior& c, const PiState& s, double e15[15], double Jr[9], double Jt[9]) {
    double Rt[9], dR[9], er[3], dt[3], et[3];
    pi_tr(c.Rwb, Rt);
    r3_mul(Rt, s.P.Rwb, dR);
    pi_log_so3(dR, er);
    for (int k = 0; k < 3; ++k) dt[k] = s.P.twb[k] - c.twb[k];
    r3_vec(Rt, dt, et);
    for (int k = 0; k < 3; ++k) { e15[k] = er[k]; e15[3 + k] = et[k]; e15[6 + k] = s.v[k] - c.vwb[k]; e15[9 + k] = s.bg[k] - c.bg[k]; e15[12 + k] = s.ba[k] - c.ba[k]; }
    if (Jr) pi_jr_so3(er, true, Jr);
    if (Jt) for (int k = 0; k < 9; ++k) Jt[k] = dR[k];
}

}  // namespace tc2li
