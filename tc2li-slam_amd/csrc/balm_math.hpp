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

// ---- sin / cos / atanf as fixed sequences of IEEE operations --------------------------------------------------------------------------
// The change of variables of the plane term (LidarCovisRes::ComputeJandHSE3: Sophus::SO3f::log -> atan, InverseRightJacobianSO3 -> sin, cos)
// runs on the host for the one-window entry points and on the device for a lock-step batch whose Levenberg-Marquardt loop stays on the GPU
// (k_ba_lm_begin_b, round 6).  libm's and the device library's functions differ in the last bit now and then; these do not: the classic
// argument reduction + minimax polynomials (the published fdlibm kernels), every step a plain multiply / add / divide (the build has
// -ffp-contract=off on both sides).  Accuracy below 1 ulp over the ranges used: the rotation angle in [0, 3.3] for sin / cos, any float for atanf.
__host__ __device__ inline double det_kernel_sin(double x, double y) {  // |x| <= pi / 4, y: tail of x
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double z = x * x, v = z * x;
    const double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}
__host__ __device__ inline double det_kernel_cos(double x, double y) {
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double z = x * x;
    const double r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    const double hz = 0.5 * z, w = 1.0 - hz;
    return w + (((1.0 - w) - hz) + (z * r - x * y));
}
// x in [0, ~3.4]: n = nearest multiple of pi / 2 (0, 1 or 2), remainder in two pieces
__host__ __device__ inline void det_sincos(double x, double* s, double* c) {
    const double pio2_1 = 1.57079632673412561417e+00, pio2_1t = 6.07710050650619224932e-11;  // pi / 2 = pio2_1 + pio2_1t (33 + 53 bits)
    const int n = x < 0.78539816339744830962 ? 0 : x < 2.35619449019234492885 ? 1 : 2;
    const double r = x - n * pio2_1, w = n * pio2_1t;
    const double y0 = r - w, y1 = (r - y0) - w;
    const double ks = det_kernel_sin(y0, y1), kc = det_kernel_cos(y0, y1);
    if (n == 0) { *s = ks; *c = kc; }
    else if (n == 1) { *s = kc; *c = -ks; }
    else { *s = -ks; *c = -kc; }
}
__host__ __device__ inline float det_atanf(float x) {
    const float hi[4] = {4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f};
    const float lo[4] = {5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f};
    const float aT[11] = {3.3333334327e-01f, -2.0000000298e-01f, 1.4285714924e-01f, -1.1111110449e-01f, 9.0908870101e-02f, -7.6918758452e-02f,
                          6.6610731184e-02f, -5.8335702866e-02f, 4.9768779427e-02f, -3.6531571299e-02f, 1.6285819933e-02f};
    const bool neg = x < 0;
    float a = neg ? -x : x;
    if (!(a == a)) return x;
    if (a >= 1.7179869184e10f) { const float z = hi[3] + lo[3]; return neg ? -z : z; }  // 2^34
    int id = -1;
    if (a < 0.4375f) {
        if (a < 1.862645149e-09f) return x;  // 2^-29
    } else if (a < 1.1875f) {
        if (a < 0.6875f) { id = 0; a = (2.0f * a - 1.0f) / (2.0f + a); }
        else { id = 1; a = (a - 1.0f) / (a + 1.0f); }
    } else {
        if (a < 2.4375f) { id = 2; a = (a - 1.5f) / (1.0f + 1.5f * a); }
        else { id = 3; a = -1.0f / a; }
    }
    const float z = a * a, w = z * z;
    const float s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
    const float s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
    if (id < 0) { const float r = a - a * (s1 + s2); return neg ? -r : r; }
    const float r = (id == 0 ? hi[0] : id == 1 ? hi[1] : id == 2 ? hi[2] : hi[3]) -
                    ((a * (s1 + s2) - (id == 0 ? lo[0] : id == 1 ? lo[1] : id == 2 ? lo[2] : lo[3])) - a);
    return neg ? -r : r;
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

// ---- change of variables of the plane term's Jacobian / Hessian: LiDAR-pose increments -> the optimiser's vertex increments ----------------
// (host: balm_to_camera_se3 / balm_to_body, balm_host.cpp; device: k_ba_lm_begin_b -- the same functions, the same bits)
__host__ __device__ inline void so3_log_f(const double* Rd, double out[3]) {  // Sophus::SO3f(R.cast<float>()).log()
    float R[9], q[4];
    for (int i = 0; i < 9; ++i) R[i] = (float)Rd[i];
    matrix_to_quat_f(R, q);
    const float sq = q[0] * q[0] + q[1] * q[1] + q[2] * q[2], w = q[3];
    float two_atan;
    const float eps = 1e-10f;
    if (sq < eps * eps) {
        two_atan = 2.0f / w - (2.0f / 3.0f) * sq / (w * w * w);
    } else {
        const float n = sqrtf(sq);
        if (fabsf(w) < eps) two_atan = (w > 0 ? 3.14159265358979323846f : -3.14159265358979323846f) / n;
        else two_atan = 2.0f * det_atanf(n / w) / n;
    }
    for (int k = 0; k < 3; ++k) out[k] = (double)(two_atan * q[k]);
}
__host__ __device__ inline void inverse_right_jacobian_so3(const double v[3], double J[9]) {  // SF/src/G2oTypes.cc:823-839
    const double d2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2], d = sqrt(d2);
    for (int k = 0; k < 9; ++k) J[k] = k % 4 == 0 ? 1.0 : 0.0;
    if (d < 1e-5) return;
    double Wm[9], W2[9], sd, cd;
    m3_hat(v, Wm);
    m3_mul(Wm, Wm, W2);
    det_sincos(d, &sd, &cd);  // (d = |log R| <= pi)
    const double k2 = 1.0 / d2 - (1.0 + cd) / (2.0 * d * sd);
    for (int k = 0; k < 9; ++k) J[k] = J[k] + 0.5 * Wm[k] + k2 * W2[k];
}
// Tlc = Tcl^-1 in float, widened (LidarCovisRes::ComputeJandHSE3, SF/src/LidarRes.cc:136-186)
struct BalmCameraFrame { double Rlc[9], tlc[3], tcl[3]; };
__host__ __device__ inline BalmCameraFrame balm_camera_frame(const SE3f& Tcl) {
    BalmCameraFrame F;
    const float qi[4] = {-Tcl.q[0], -Tcl.q[1], -Tcl.q[2], Tcl.q[3]};
    const float nt[3] = {Tcl.t[0] * -1.f, Tcl.t[1] * -1.f, Tcl.t[2] * -1.f};
    float tlc_f[3];
    quat_rotate_f(qi, nt, tlc_f);
    quat_to_matrix_f(qi, F.Rlc);
    for (int k = 0; k < 3; ++k) { F.tlc[k] = (double)tlc_f[k]; F.tcl[k] = (double)Tcl.t[k]; }
    return F;
}
// keyframe i of the window: its six entries of JacT (in place) and D_i^T (6 x 6: rows = camera increment (rotation, translation), columns =
// LiDAR-pose increment)
__host__ __device__ inline void balm_camera_se3_D(const LidarPose& T, const BalmCameraFrame& F, double* J6, double* DT) {
    const double* Rwl = T.R;
    double Rwc[9], Rcw[9], twc[3], tcw[3], tmp[3];
    m3_mul(Rwl, F.Rlc, Rwc);
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Rcw[3 * r + c] = Rwc[3 * c + r];
    m3_vec(Rwl, F.tlc, tmp);
    for (int k = 0; k < 3; ++k) twc[k] = tmp[k] + T.p[k];
    m3_vec(Rcw, twc, tmp);
    for (int k = 0; k < 3; ++k) tcw[k] = -1.0 * tmp[k];
    const double Jw[3] = {J6[0], J6[1], J6[2]}, Jt[3] = {J6[3], J6[4], J6[5]};
    double rwl[3], Jr[9], JrRlc[9], A[9] /* (Jr^-1 Rlc)^T */, dt[3], dth[9], RwcH[9], B[9] /* (Rwc [tcl - tcw]x)^T */;
    so3_log_f(Rwl, rwl);
    inverse_right_jacobian_so3(rwl, Jr);
    m3_mul(Jr, F.Rlc, JrRlc);
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) A[3 * r + c] = JrRlc[3 * c + r];
    for (int k = 0; k < 3; ++k) dt[k] = F.tcl[k] - tcw[k];
    m3_hat(dt, dth);
    m3_mul(Rwc, dth, RwcH);
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) B[3 * r + c] = RwcH[3 * c + r];
    double AJw[3], BJt[3], Jw2[3], Jt2[3], tcwh[9], tcwhT[9], t1[3], t2[3];
    m3_vec(A, Jw, AJw);
    m3_vec(B, Jt, BJt);
    for (int k = 0; k < 3; ++k) Jw2[k] = -1.0 * AJw[k] + BJt[k];
    m3_vec(Rcw, Jt, tmp);
    for (int k = 0; k < 3; ++k) Jt2[k] = -1.0 * tmp[k];
    m3_hat(tcw, tcwh);
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) tcwhT[3 * r + c] = tcwh[3 * c + r];
    m3_vec(Rcw, Jw2, t1);
    m3_vec(tcwhT, Jt2, t2);
    double m1[9], m2[9], m3[9];
    m3_mul(Rcw, A, m1);
    m3_mul(Rcw, B, m2);
    m3_mul(tcwhT, Rcw, m3);
    for (int k = 0; k < 36; ++k) DT[k] = 0;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            DT[6 * r + c] = -1.0 * m1[3 * r + c];
            DT[6 * r + 3 + c] = m2[3 * r + c] + m3[3 * r + c];
            DT[6 * (3 + r) + 3 + c] = -1.0 * Rcw[3 * r + c];
        }
    for (int k = 0; k < 3; ++k) { J6[k] = t1[k] - t2[k]; J6[3 + k] = Jt2[k]; }
}
// the same for the body-frame increment of VertexPose (LidarCovisRes::ComputeJandH, SF/src/LidarRes.cc:89-128); Rlb of mTlb = mTbl.inverse()
__host__ __device__ inline void balm_body_D(const LidarPose& T, const double Rlb[9], const double tbl[3], double* J6, double* DT) {
    const double* Rwl = T.R;
    double Rwb[9], RwbT[9], rwl[3], Jr[9], JrRlb[9], A[9] /* (Jr^-1 Rlb)^T */, th[9], RwbH[9], B[9] /* (Rwb [tbl]x)^T */;
    m3_mul(Rwl, Rlb, Rwb);
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) RwbT[3 * r + c] = Rwb[3 * c + r];
    so3_log_f(Rwl, rwl);
    inverse_right_jacobian_so3(rwl, Jr);
    m3_mul(Jr, Rlb, JrRlb);
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) A[3 * r + c] = JrRlb[3 * c + r];
    m3_hat(tbl, th);
    m3_mul(Rwb, th, RwbH);
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) B[3 * r + c] = RwbH[3 * c + r];
    const double Jw[3] = {J6[0], J6[1], J6[2]}, Jt[3] = {J6[3], J6[4], J6[5]};
    double AJw[3], BJt[3], RJt[3];
    m3_vec(A, Jw, AJw);
    m3_vec(B, Jt, BJt);
    m3_vec(RwbT, Jt, RJt);
    for (int k = 0; k < 3; ++k) { J6[k] = AJw[k] - BJt[k]; J6[3 + k] = RJt[k]; }
    for (int k = 0; k < 36; ++k) DT[k] = 0;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            DT[6 * r + c] = A[3 * r + c];
            DT[6 * r + 3 + c] = -1.0 * B[3 * r + c];
            DT[6 * (3 + r) + 3 + c] = RwbT[3 * r + c];
        }
}
// Block (a, b) of the (6W)^2 Hessian: D_a^T H_ab D_b.  The reference walks i = 0 .. W - 1 and, for every j, replaces row block (i, j) by
// D_i^T (i, j) and then column block (j, i) by (j, i) D_i, in place: block (a, b) is multiplied from the left first when a <= b and from
// the right first when a > b -- the order the two roundings happen in, kept here so that the blocks can be formed independently.
__host__ __device__ inline void balm_change_block(double* H, int n, int a, int b, const double* DTa, const double* DTb) {
    double blk[36], o[36];
    for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) blk[6 * r + c] = H[(size_t)(6 * a + r) * n + 6 * b + c];
    for (int pass = 0; pass < 2; ++pass) {
        if ((pass == 0) == (a <= b)) {  // D_a^T blk
            for (int r = 0; r < 6; ++r)
                for (int c = 0; c < 6; ++c) { double s = 0; for (int k = 0; k < 6; ++k) s += DTa[6 * r + k] * blk[6 * k + c]; o[6 * r + c] = s; }
        } else {                        // blk D_b (D_b[k][c] = DTb[c][k])
            for (int r = 0; r < 6; ++r)
                for (int c = 0; c < 6; ++c) { double s = 0; for (int k = 0; k < 6; ++k) s += blk[6 * r + k] * DTb[6 * c + k]; o[6 * r + c] = s; }
        }
        for (int k = 0; k < 36; ++k) blk[k] = o[k];
    }
    for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) H[(size_t)(6 * a + r) * n + 6 * b + c] = blk[6 * r + c];
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
