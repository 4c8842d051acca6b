// Arithmetic of the LiDAR plane term of the local bundle adjustment, shared by the host plane builder and the gfx950
// kernels: point clusters (SF/include/tools.h:163-214), the LiDAR pose of a window keyframe through the float SE3 of
// LidarCovisRes::UpdatePose (SF/src/LidarRes.cc:221-235) and the symmetric 3x3 eigen decomposition that stands in for
// Eigen::SelfAdjointEigenSolver<Matrix3d> in VOX_HESS (SF/include/bavoxel.h:80-196,276-315).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "ba_math.hpp"

namespace tc2li {

// Sufficient statistics of the points one keyframe contributes to one plane, in that keyframe's LiDAR frame:
// P = sum x x^T (packed 00 01 02 11 12 22), v = sum x, n = count.
struct PlaneCluster { double P[6], v[3], n; };
struct LidarPose { double R[9], p[3]; };  // T_world_lidar
struct SE3f { float q[4], t[3]; };        // Sophus::SE3f: unit quaternion (x, y, z, w) + translation

__host__ __device__ inline void quat_rotate_f(const float q[4], const float v[3], float o[3]) {
    float uv[3] = {q[1] * v[2] - q[2] * v[1], q[2] * v[0] - q[0] * v[2], q[0] * v[1] - q[1] * v[0]};
    uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
    o[0] = v[0] + q[3] * uv[0] + (q[1] * uv[2] - q[2] * uv[1]);
    o[1] = v[1] + q[3] * uv[1] + (q[2] * uv[0] - q[0] * uv[2]);
    o[2] = v[2] + q[3] * uv[2] + (q[0] * uv[1] - q[1] * uv[0]);
}
__host__ __device__ inline void matrix_to_quat_f(const float R[9], float q[4]) {  // Eigen::Quaternionf(Matrix3f), normalised
    float t = R[0] + R[4] + R[8];
    if (t > 0) {
        t = sqrtf(t + 1.0f);
        q[3] = 0.5f * t;
        t = 0.5f / t;
        q[0] = (R[7] - R[5]) * t; q[1] = (R[2] - R[6]) * t; q[2] = (R[3] - R[1]) * t;
    } else {
        // the largest diagonal element decides the case; each case with constant indices (the same operations in the same order as the
        // indexed form -- which kept R and q in scratch memory on the device: every access a trip to memory beside the other stages' kernels)
        int i = 0;
        if (R[4] > R[0]) i = 1;
        if (R[8] > (i == 1 ? R[4] : R[0])) i = 2;
#define TC2LI_QUAT_CASE(I, J, K)                                          \
    {                                                                     \
        t = sqrtf(R[4 * I] - R[4 * J] - R[4 * K] + 1.0f);                     \
        const float qi = 0.5f * t;                                        \
        t = 0.5f / t;                                                     \
        q[3] = (R[3 * K + J] - R[3 * J + K]) * t;                         \
        q[J] = (R[3 * J + I] + R[3 * I + J]) * t;                         \
        q[K] = (R[3 * K + I] + R[3 * I + K]) * t;                         \
        q[I] = qi;                                                        \
    }
        if (i == 0) TC2LI_QUAT_CASE(0, 1, 2)
        else if (i == 1) TC2LI_QUAT_CASE(1, 2, 0)
        else TC2LI_QUAT_CASE(2, 0, 1)
#undef TC2LI_QUAT_CASE
    }
    const float n = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}
__host__ __device__ inline void quat_to_matrix_f(const float q[4], double R[9]) {  // Eigen::Quaternionf::toRotationMatrix, widened
    const float tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
    const float twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
    const float txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
    const float tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
    const float Rf[9] = {1 - (tyy + tzz), txy - twz, txz + twy, txy + twz, 1 - (txx + tzz), tyz - twx, txz - twy, tyz + twx, 1 - (txx + tyy)};
    for (int k = 0; k < 9; ++k) R[k] = (double)Rf[k];
}
// (Tcw^-1 * Tcl) evaluated in float like Sophus::SE3f, widened to double afterwards
__host__ __device__ inline LidarPose lidar_pose_from(const SE3f& Tcw, const SE3f& Tcl) {
    const float qi[4] = {-Tcw.q[0], -Tcw.q[1], -Tcw.q[2], Tcw.q[3]};
    const float nt[3] = {Tcw.t[0] * -1.f, Tcw.t[1] * -1.f, Tcw.t[2] * -1.f};
    float twc[3], rt[3], q[4];
    quat_rotate_f(qi, nt, twc);
    q[3] = qi[3] * Tcl.q[3] - qi[0] * Tcl.q[0] - qi[1] * Tcl.q[1] - qi[2] * Tcl.q[2];
    q[0] = qi[3] * Tcl.q[0] + qi[0] * Tcl.q[3] + qi[1] * Tcl.q[2] - qi[2] * Tcl.q[1];
    q[1] = qi[3] * Tcl.q[1] + qi[1] * Tcl.q[3] + qi[2] * Tcl.q[0] - qi[0] * Tcl.q[2];
    q[2] = qi[3] * Tcl.q[2] + qi[2] * Tcl.q[3] + qi[0] * Tcl.q[1] - qi[1] * Tcl.q[0];
    quat_rotate_f(qi, Tcl.t, rt);
    LidarPose L;
    quat_to_matrix_f(q, L.R);
    for (int k = 0; k < 3; ++k) L.p[k] = (double)(twc[k] + rt[k]);
    return L;
}
// the double-precision vertex estimate (g2o::SE3Quat) as the Sophus::SE3f that UpdatePose builds from R and t
__host__ __device__ inline SE3f se3f_from_vertex(const Se3& T) {
    double Rd[9];
    quat_to_matrix(T.q, Rd);
    float Rf[9];
    for (int k = 0; k < 9; ++k) Rf[k] = (float)Rd[k];
    SE3f o;
    matrix_to_quat_f(Rf, o.q);
    for (int k = 0; k < 3; ++k) o.t[k] = (float)T.t[k];
    return o;
}
// VertexPose's estimate().Rcw[0] / tcw[0] (doubles) as the Sophus::SE3f of the same UpdatePose (SF/src/G2oTypesWithLidar.cc:35-39)
__host__ __device__ inline SE3f se3f_from_rt(const double* Rcw, const double* tcw) {
    float Rf[9];
    for (int k = 0; k < 9; ++k) Rf[k] = (float)Rcw[k];
    SE3f o;
    matrix_to_quat_f(Rf, o.q);
    for (int k = 0; k < 3; ++k) o.t[k] = (float)tcw[k];
    return o;
}

// ---- 3x3 helpers (row-major) ------------------------------------------------------------------------------------------
__host__ __device__ inline void m3_mul(const double* a, const double* b, double* o) {
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) o[3 * r + c] = a[3 * r] * b[c] + a[3 * r + 1] * b[3 + c] + a[3 * r + 2] * b[6 + c];
}
__host__ __device__ inline void m3_mul_bt(const double* a, const double* b, double* o) {  // a * b^T
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) o[3 * r + c] = a[3 * r] * b[3 * c] + a[3 * r + 1] * b[3 * c + 1] + a[3 * r + 2] * b[3 * c + 2];
}
__host__ __device__ inline void m3_vec(const double* a, const double* v, double* o) {
    for (int r = 0; r < 3; ++r) o[r] = a[3 * r] * v[0] + a[3 * r + 1] * v[1] + a[3 * r + 2] * v[2];
}
__host__ __device__ inline void m3_tvec(const double* a, const double* v, double* o) {  // a^T v
    for (int r = 0; r < 3; ++r) o[r] = a[r] * v[0] + a[3 + r] * v[1] + a[6 + r] * v[2];
}
__host__ __device__ inline void m3_hat(const double* v, double* o) {
    o[0] = 0; o[1] = -v[2]; o[2] = v[1]; o[3] = v[2]; o[4] = 0; o[5] = -v[0]; o[6] = -v[1]; o[7] = v[0]; o[8] = 0;
}
__host__ __device__ inline void sym_unpack(const double* s, double* m) {
    m[0] = s[0]; m[1] = s[1]; m[2] = s[2]; m[3] = s[1]; m[4] = s[3]; m[5] = s[4]; m[6] = s[2]; m[7] = s[4]; m[8] = s[5];
}

// PointCluster::transform: statistics of the same points after x -> R x + p (full 3x3 P, v, n)
struct ClusterW { double P[9], v[3], n; };
__host__ __device__ inline void cluster_transform(const PlaneCluster& s, const LidarPose& T, ClusterW& o) {
    double P[9], RP[9], Rv[3];
    sym_unpack(s.P, P);
    o.n = s.n;
    m3_vec(T.R, s.v, Rv);
    for (int k = 0; k < 3; ++k) o.v[k] = Rv[k] + s.n * T.p[k];
    m3_mul(T.R, P, RP);
    m3_mul_bt(RP, T.R, o.P);
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) o.P[3 * r + c] = o.P[3 * r + c] + Rv[r] * T.p[c] + Rv[c] * T.p[r] + s.n * (T.p[r] * T.p[c]);
}

// Symmetric 3x3 eigen decomposition by cyclic Jacobi rotations; eigenvalues ascending, eigenvectors in the columns of U.
__host__ __device__ inline void eig_sym3(const double* Ain, double lambda[3], double U[9]) {
    double A[9], V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) A[3 * r + c] = 0.5 * (Ain[3 * r + c] + Ain[3 * c + r]);
    for (int sweep = 0; sweep < 60; ++sweep) {
        const double off = A[1] * A[1] + A[2] * A[2] + A[5] * A[5];
        const double dg = A[0] * A[0] + A[4] * A[4] + A[8] * A[8];
        if (off <= 1e-32 * dg || off == 0.0) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                const double apq = A[3 * p + q];
                if (apq == 0.0) continue;
                const double theta = (A[4 * q] - A[4 * p]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; ++k) { const double x = A[3 * k + p], y = A[3 * k + q]; A[3 * k + p] = c * x - s * y; A[3 * k + q] = s * x + c * y; }
                for (int k = 0; k < 3; ++k) { const double x = A[3 * p + k], y = A[3 * q + k]; A[3 * p + k] = c * x - s * y; A[3 * q + k] = s * x + c * y; }
                for (int k = 0; k < 3; ++k) { const double x = V[3 * k + p], y = V[3 * k + q]; V[3 * k + p] = c * x - s * y; V[3 * k + q] = s * x + c * y; }
            }
    }
    int o0 = 0, o1 = 1, o2 = 2;
    const double dgn[3] = {A[0], A[4], A[8]};
    auto diag = [&](int o) { return o == 0 ? dgn[0] : o == 1 ? dgn[1] : dgn[2]; };
    if (diag(o1) < diag(o0)) { const int t = o0; o0 = o1; o1 = t; }
    if (diag(o2) < diag(o0)) { const int t = o0; o0 = o2; o2 = t; }
    if (diag(o2) < diag(o1)) { const int t = o1; o1 = o2; o2 = t; }
    // (selected with comparisons, not indexed: an index known only at run time would put A and V into scratch memory on the device)
    const int ord[3] = {o0, o1, o2};
    for (int k = 0; k < 3; ++k) {
        const int o = ord[k];
        lambda[k] = o == 0 ? A[0] : o == 1 ? A[4] : A[8];
        for (int r = 0; r < 3; ++r) U[3 * r + k] = o == 0 ? V[3 * r] : o == 1 ? V[3 * r + 1] : V[3 * r + 2];
    }
}

}  // namespace tc2li
