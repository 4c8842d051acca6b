// Visual-inertial vertex / edge arithmetic shared by the BA kernels and the host side of the inertial local BA:
//   ImuCamPose::{Project, ProjectStereo, isDepthPositive, Update}   SF/src/G2oTypes.cc:178-232
//   EdgeMono / EdgeStereo::{computeError, linearizeOplus}           SF/include/G2oTypes.h:378-460, SF/src/G2oTypes.cc:358-436
//   ExpSO3, NormalizeRotation                                       SF/src/G2oTypes.cc:783-805, SF/include/G2oTypes.h:76-81
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "ba_math.hpp"

namespace tc2li {

struct ImuPose {  // the parts of ImuCamPose the optimisation moves (single camera)
    double Rcw[9], tcw[3], Rwb[9], twb[3];
    int32_t its, pad_;
};
struct ImuCalib { double Rcb[9], tcb[3], Rbc[9], tbc[3]; };  // tc2li_imu_calib

__host__ __device__ inline void r3_mul(const double* a, const double* b, double* o) {
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) o[3 * r + c] = a[3 * r] * b[c] + a[3 * r + 1] * b[3 + c] + a[3 * r + 2] * b[6 + c];
}
__host__ __device__ inline void r3_vec(const double* a, const double* v, double* o) {
    for (int r = 0; r < 3; ++r) o[r] = a[3 * r] * v[0] + a[3 * r + 1] * v[1] + a[3 * r + 2] * v[2];
}

// NormalizeRotation: U V^T of the SVD (Eigen::JacobiSVD in the reference), here by one-sided Jacobi rotations on the
// columns: R V = U S, hence U V^T = (R V) S^-1 V^T.
// X <- (X + X^-T) / 2: one Newton step towards the orthogonal polar factor of X (= U V^T of its SVD); X^-T = cofactors / determinant
__host__ __device__ inline void polar_newton_step_d(double* X) {
    const double c[9] = {X[4] * X[8] - X[5] * X[7], X[5] * X[6] - X[3] * X[8], X[3] * X[7] - X[4] * X[6],
                         X[2] * X[7] - X[1] * X[8], X[0] * X[8] - X[2] * X[6], X[1] * X[6] - X[0] * X[7],
                         X[1] * X[5] - X[2] * X[4], X[2] * X[3] - X[0] * X[5], X[0] * X[4] - X[1] * X[3]};
    const double inv = 1.0 / (X[0] * c[0] + X[1] * c[1] + X[2] * c[2]);
    for (int k = 0; k < 9; ++k) X[k] = 0.5 * (X[k] + c[k] * inv);
}
__host__ __device__ inline void normalize_rotation_d(double* R) {
    // Round 5: a matrix that is orthogonal to 1e-8 already -- every caller's: exp_so3's Rodrigues sum, a product of rotations every third update -- gets
    // its polar factor U V^T by two Newton steps (error e -> e^2 / 2 per step: exact to the last bits after the first, the second is for the
    // margin) instead of Jacobi sweeps to 1e-16: on one lane of k_pose_inertial the sweeps were ~9 k of an iteration's 17 k cycles of state update.
    {
        double e = 0;
        for (int a = 0; a < 3; ++a)
            for (int b = a; b < 3; ++b)
                e = fmax(e, fabs(R[a] * R[b] + R[3 + a] * R[3 + b] + R[6 + a] * R[6 + b] - (a == b ? 1.0 : 0.0)));
        if (e < 1e-8) { polar_newton_step_d(R); polar_newton_step_d(R); return; }
    }
    double A[9], V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int k = 0; k < 9; ++k) A[k] = R[k];
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0;
#pragma unroll
        for (int pq = 0; pq < 3; ++pq) {
            const int p = pq == 2 ? 1 : 0, q = pq == 0 ? 1 : 2;
            const double app = A[p] * A[p] + A[3 + p] * A[3 + p] + A[6 + p] * A[6 + p];
            const double aqq = A[q] * A[q] + A[3 + q] * A[3 + q] + A[6 + q] * A[6 + q];
            const double apq = A[p] * A[q] + A[3 + p] * A[3 + q] + A[6 + p] * A[6 + q];
            const double rel = fabs(apq) / sqrt(fmax(app * aqq, 1e-300));
            off = fmax(off, rel);
            if (rel <= 1e-17) continue;
            const double tau = (aqq - app) / (2.0 * apq);
            const double t = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
            const double c = 1.0 / sqrt(1.0 + t * t), s = t * c;
            for (int k = 0; k < 3; ++k) {
                const double x = A[3 * k + p], y = A[3 * k + q];
                A[3 * k + p] = c * x - s * y; A[3 * k + q] = s * x + c * y;
                const double vx = V[3 * k + p], vy = V[3 * k + q];
                V[3 * k + p] = c * vx - s * vy; V[3 * k + q] = s * vx + c * vy;
            }
        }
        if (off < 1e-16) break;
    }
    double U[9];
    for (int c = 0; c < 3; ++c) {
        const double n = sqrt(A[c] * A[c] + A[3 + c] * A[3 + c] + A[6 + c] * A[6 + c]);
        for (int k = 0; k < 3; ++k) U[3 * k + c] = A[3 * k + c] / n;
    }
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) R[3 * r + c] = U[3 * r] * V[3 * c] + U[3 * r + 1] * V[3 * c + 1] + U[3 * r + 2] * V[3 * c + 2];
}

__host__ __device__ inline void exp_so3(const double w[3], double R[9]) {
    const double d2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], d = sqrt(d2);
    const double W[9] = {0.0, -w[2], w[1], w[2], 0.0, -w[0], -w[1], w[0], 0.0};
    double W2[9];
    r3_mul(W, W, W2);
    const double a = d < 1e-5 ? 1.0 : sin(d) / d, b = d < 1e-5 ? 0.5 : (1.0 - cos(d)) / d2;
    for (int k = 0; k < 9; ++k) R[k] = (k % 4 == 0 ? 1.0 : 0.0) + W[k] * a + W2[k] * b;
    normalize_rotation_d(R);
}

// ImuCamPose::Update: body pose moved by (rotation, translation) increments in the body frame, camera pose re-derived
__host__ __device__ inline void imu_pose_update(ImuPose& T, const ImuCalib& cal, const double u[6]) {
    double Rut[3], E[9], Rn[9];
    r3_vec(T.Rwb, u + 3, Rut);
    for (int k = 0; k < 3; ++k) T.twb[k] += Rut[k];
    exp_so3(u, E);
    r3_mul(T.Rwb, E, Rn);
    for (int k = 0; k < 9; ++k) T.Rwb[k] = Rn[k];
    T.its++;
    if (T.its >= 3) { normalize_rotation_d(T.Rwb); T.its = 0; }
    double Rbw[9], tbw[3], t[3];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Rbw[3 * r + c] = T.Rwb[3 * c + r];
    r3_vec(Rbw, T.twb, t);
    for (int k = 0; k < 3; ++k) tbw[k] = -t[k];
    r3_mul(cal.Rcb, Rbw, T.Rcw);
    r3_vec(cal.Rcb, tbw, t);
    for (int k = 0; k < 3; ++k) T.tcw[k] = t[k] + cal.tcb[k];
}

// EdgeMono / EdgeStereo::computeError with a pinhole camera whose parameters are floats: returns the dimension
__host__ __device__ inline int imu_edge_error(const ImuPose& T, const double X[3], const BaEdge& e, const CameraD& cam, double Xc[3], double err[3]) {
    r3_vec(T.Rcw, X, Xc);
    for (int k = 0; k < 3; ++k) Xc[k] += T.tcw[k];
    const double u = cam.fx * Xc[0] / Xc[2] + cam.cx, v = cam.fy * Xc[1] / Xc[2] + cam.cy;
    err[0] = e.u - u; err[1] = e.v - v; err[2] = 0;
    if (e.ur >= 0) { const double invZ = 1 / Xc[2]; err[2] = e.ur - (u - cam.bf * invZ); return 3; }
    return 2;
}
// linearizeOplus: A = -proj_jac Rcw (dim x 3), B = proj_jac Rcb SE3deriv(Xb) (dim x 6); rows beyond dim are zero
__host__ __device__ inline void imu_edge_jacobians(const ImuPose& T, const ImuCalib& cal, const double Xc[3], bool stereo, const CameraD& cam,
                                                   double A[9], double B[18]) {
    double Xb[3];
    r3_vec(cal.Rbc, Xc, Xb);
    for (int k = 0; k < 3; ++k) Xb[k] += cal.tbc[k];
    double pj[9] = {cam.fx / Xc[2], 0.0, -cam.fx * Xc[0] / (Xc[2] * Xc[2]), 0.0, cam.fy / Xc[2], -cam.fy * Xc[1] / (Xc[2] * Xc[2]), 0.0, 0.0, 0.0};
    if (stereo) { pj[6] = pj[0]; pj[7] = pj[1]; pj[8] = pj[2] + cam.bf * (1.0 / (Xc[2] * Xc[2])); }
    double PR[9];
    r3_mul(pj, T.Rcw, A);
    for (int k = 0; k < 9; ++k) A[k] = -A[k];
    r3_mul(pj, cal.Rcb, PR);
    const double x = Xb[0], y = Xb[1], z = Xb[2];
    const double S[18] = {0.0, z, -y, 1.0, 0.0, 0.0, -z, 0.0, x, 0.0, 1.0, 0.0, y, -x, 0.0, 0.0, 0.0, 1.0};
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 6; ++c) B[6 * r + c] = PR[3 * r] * S[c] + PR[3 * r + 1] * S[6 + c] + PR[3 * r + 2] * S[12 + c];
    if (!stereo) { for (int c = 0; c < 3; ++c) A[6 + c] = 0; for (int c = 0; c < 6; ++c) B[12 + c] = 0; }
}

}  // namespace tc2li
