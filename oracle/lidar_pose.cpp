// TEST INFRASTRUCTURE ONLY -- CPU oracle, never linked into or called from the product path.  See lidar_pose.hpp.
#include "lidar_pose.hpp"

#include <cmath>
#include <cstring>

namespace oracle {

namespace {
const float kEps = 1e-5f;  // Sophus::Constants<float>::epsilon()
const float kPi = 3.141592653589793238462643383279502884f;

void normalize_q(float q[4]) {  // SO3::normalize(): coeffs /= norm
    const float len = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int k = 0; k < 4; ++k) q[k] = q[k] / len;
}
// Eigen::QuaternionBase::_transformVector: v + w * (2 q x v) + q x (2 q x v)
void rotate(const float q[4], const float v[3], float o[3]) {
    float uv[3] = {q[1] * v[2] - q[2] * v[1], q[2] * v[0] - q[0] * v[2], q[0] * v[1] - q[1] * v[0]};
    for (int k = 0; k < 3; ++k) uv[k] = uv[k] + uv[k];
    const float c[3] = {q[1] * uv[2] - q[2] * uv[1], q[2] * uv[0] - q[0] * uv[2], q[0] * uv[1] - q[1] * uv[0]};
    for (int k = 0; k < 3; ++k) o[k] = v[k] + q[3] * uv[k] + c[k];
}
void hat(const float w[3], float O[9]) { O[0] = 0; O[1] = -w[2]; O[2] = w[1]; O[3] = w[2]; O[4] = 0; O[5] = -w[0]; O[6] = -w[1]; O[7] = w[0]; O[8] = 0; }
void mul3(const float* a, const float* b, float* o) {
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) o[3 * r + c] = a[3 * r] * b[c] + a[3 * r + 1] * b[3 + c] + a[3 * r + 2] * b[6 + c];
}
void mulv3(const float* a, const float* v, float* o) { for (int r = 0; r < 3; ++r) o[r] = a[3 * r] * v[0] + a[3 * r + 1] * v[1] + a[3 * r + 2] * v[2]; }
// Eigen: Quaternion from a rotation matrix (Geometry/Quaternion.h, quaternionbase_assign_impl<Other, 3, 3>)
void quat_from_matrix(const float m[9], float q[4]) {
    float t = m[0] + m[4] + m[8];
    if (t > 0.0f) {
        t = std::sqrt(t + 1.0f);
        q[3] = 0.5f * t;
        t = 0.5f / t;
        q[0] = (m[7] - m[5]) * t; q[1] = (m[2] - m[6]) * t; q[2] = (m[3] - m[1]) * t;
    } else {
        int i = 0;
        if (m[4] > m[0]) i = 1;
        if (m[8] > m[4 * i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(m[4 * i] - m[4 * j] - m[4 * k] + 1.0f);
        q[i] = 0.5f * t;
        t = 0.5f / t;
        q[3] = (m[3 * k + j] - m[3 * j + k]) * t;
        q[j] = (m[3 * j + i] + m[3 * i + j]) * t;
        q[k] = (m[3 * k + i] + m[3 * i + k]) * t;
    }
}
}  // namespace

void se3f_rotation_matrix(const SE3F& T, float R[9]) {  // Eigen toRotationMatrix
    const float* q = T.q;
    const float tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
    const float twx = tx * q[3], twy = ty * q[3], twz = tz * q[3], txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
    const float tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}

SE3F se3f_inverse(const SE3F& T) {  // se3.hpp:208-211
    SE3F o;
    o.q[0] = -T.q[0]; o.q[1] = -T.q[1]; o.q[2] = -T.q[2]; o.q[3] = T.q[3];
    normalize_q(o.q);  // SO3(quaternion) normalises
    const float nt[3] = {T.t[0] * -1.0f, T.t[1] * -1.0f, T.t[2] * -1.0f};
    rotate(o.q, nt, o.t);
    return o;
}

SE3F se3f_mul(const SE3F& A, const SE3F& B) {  // se3.hpp:304-310, so3.hpp:325-340
    SE3F o;
    const float *a = A.q, *b = B.q;  // (x, y, z, w)
    o.q[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
    o.q[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    o.q[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
    o.q[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
    normalize_q(o.q);
    float r[3];
    rotate(A.q, B.t, r);
    for (int k = 0; k < 3; ++k) o.t[k] = A.t[k] + r[k];
    return o;
}

void se3f_log(const SE3F& T, float out[6]) {  // se3.hpp:223-260, so3.hpp:247-290
    const float* q = T.q;
    const float squared_n = q[0] * q[0] + q[1] * q[1] + q[2] * q[2], w = q[3];
    float two_atan_nbyw_by_n, theta;
    if (squared_n < kEps * kEps) {
        const float squared_w = w * w;
        two_atan_nbyw_by_n = 2.0f / w - (float)(2.0 / 3.0) * squared_n / (w * squared_w);
        theta = 2.0f * squared_n / w;
    } else {
        const float n = std::sqrt(squared_n);
        if (std::fabs(w) < kEps) two_atan_nbyw_by_n = w > 0.0f ? kPi / n : -kPi / n;
        else two_atan_nbyw_by_n = 2.0f * std::atan(n / w) / n;
        theta = two_atan_nbyw_by_n * n;
    }
    float om[3] = {two_atan_nbyw_by_n * q[0], two_atan_nbyw_by_n * q[1], two_atan_nbyw_by_n * q[2]};
    float O[9], O2[9], V[9];
    hat(om, O);
    mul3(O, O, O2);
    if (std::fabs(theta) < kEps) {
        for (int k = 0; k < 9; ++k) V[k] = (k % 4 == 0 ? 1.0f : 0.0f) - 0.5f * O[k] + (float)(1. / 12.) * O2[k];
    } else {
        const float half = 0.5f * theta;
        const float c = (1.0f - theta * std::cos(half) / (2.0f * std::sin(half))) / (theta * theta);
        for (int k = 0; k < 9; ++k) V[k] = (k % 4 == 0 ? 1.0f : 0.0f) - 0.5f * O[k] + c * O2[k];
    }
    mulv3(V, T.t, out);
    out[3] = om[0]; out[4] = om[1]; out[5] = om[2];
}

SE3F se3f_exp(const float a[6]) {  // se3.hpp:761-782, so3.hpp:583-618
    SE3F o;
    const float* om = a + 3;
    const float theta_sq = om[0] * om[0] + om[1] * om[1] + om[2] * om[2];
    float theta, imag, real;
    if (theta_sq < kEps * kEps) {
        theta = 0;
        const float th4 = theta_sq * theta_sq;
        imag = 0.5f - (float)(1.0 / 48.0) * theta_sq + (float)(1.0 / 3840.0) * th4;
        real = 1.0f - (float)(1.0 / 8.0) * theta_sq + (float)(1.0 / 384.0) * th4;
    } else {
        theta = std::sqrt(theta_sq);
        const float half = 0.5f * theta;
        imag = std::sin(half) / theta;
        real = std::cos(half);
    }
    o.q[3] = real; o.q[0] = imag * om[0]; o.q[1] = imag * om[1]; o.q[2] = imag * om[2];
    float O[9], O2[9], V[9];
    hat(om, O);
    mul3(O, O, O2);
    if (theta < kEps) {
        se3f_rotation_matrix(o, V);
    } else {
        const float th2 = theta * theta;
        const float c1 = (1.0f - std::cos(theta)) / th2, c2 = (theta - std::sin(theta)) / (th2 * theta);
        for (int k = 0; k < 9; ++k) V[k] = (k % 4 == 0 ? 1.0f : 0.0f) + c1 * O[k] + c2 * O2[k];
    }
    mulv3(V, a, o.t);
    return o;
}

SE3F InterpolateSE3(const SE3F& A, const SE3F& B, float t) {  // Tracking.cc:1552-1563
    float Ra[9], Rb[9], q1[4], q2[4];
    se3f_rotation_matrix(A, Ra); se3f_rotation_matrix(B, Rb);
    quat_from_matrix(Ra, q1); quat_from_matrix(Rb, q2);
    // Eigen::QuaternionBase::slerp
    const float one = 1.0f - 1.1920928955078125e-07f;
    const float d = q1[0] * q2[0] + q1[1] * q2[1] + q1[2] * q2[2] + q1[3] * q2[3];
    const float absD = std::fabs(d);
    float scale0, scale1;
    if (absD >= one) {
        scale0 = 1.0f - t; scale1 = t;
    } else {
        const float theta = std::acos(absD), sinTheta = std::sin(theta);
        scale0 = std::sin((1.0f - t) * theta) / sinTheta;
        scale1 = std::sin(t * theta) / sinTheta;
    }
    if (d < 0.0f) scale1 = -scale1;
    SE3F o;
    for (int k = 0; k < 4; ++k) o.q[k] = scale0 * q1[k] + scale1 * q2[k];
    normalize_q(o.q);  // Sophus::SE3f(Quaternionf, Vector3f)
    for (int k = 0; k < 3; ++k) o.t[k] = A.t[k] + t * (B.t[k] - A.t[k]);
    return o;
}

void UpdateLidarPose(const SE3F& Tcw_last, const SE3F& velocity, double timeFromLastFrame, const SE3F& Tcl, LidarState& st, double pos_lid[3]) {
    float lg[6], a[6];
    se3f_log(se3f_inverse(velocity), lg);
    const float s = (float)timeFromLastFrame;  // double scalar times a float vector: the scalar is converted
    for (int k = 0; k < 6; ++k) a[k] = s * lg[k];
    const SE3F Twc = se3f_mul(se3f_inverse(Tcw_last), se3f_exp(a));
    // Mwl = Twc.matrix() * Tcl.matrix() (4 x 4 floats, sums over k = 0..3 left to right; the last rows are 0 0 0 1)
    float Rwc[9], Rcl[9], Mwl[12];
    se3f_rotation_matrix(Twc, Rwc); se3f_rotation_matrix(Tcl, Rcl);
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) Mwl[4 * r + c] = ((Rwc[3 * r] * Rcl[c] + Rwc[3 * r + 1] * Rcl[3 + c]) + Rwc[3 * r + 2] * Rcl[6 + c]) + Twc.t[r] * 0.0f;
        Mwl[4 * r + 3] = ((Rwc[3 * r] * Tcl.t[0] + Rwc[3 * r + 1] * Tcl.t[1]) + Rwc[3 * r + 2] * Tcl.t[2]) + Twc.t[r] * 1.0f;
    }
    // Rw2_w1 = [0 0 1; -1 0 0; 0 -1 0]: row 0 = row 2, row 1 = -row 0, row 2 = -row 1 (products with 0 and 1 are exact)
    const int src[3] = {2, 0, 1};
    const float sgn[3] = {1.0f, -1.0f, -1.0f};
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) st.rot[3 * r + c] = (double)(sgn[r] * Mwl[4 * src[r] + c]);
        st.pos[r] = (double)(sgn[r] * Mwl[4 * src[r] + 3]);
    }
    for (int r = 0; r < 3; ++r) pos_lid[r] = st.pos[r] + (st.rot[3 * r] * st.offset_T_L_I[0] + st.rot[3 * r + 1] * st.offset_T_L_I[1] + st.rot[3 * r + 2] * st.offset_T_L_I[2]);
}

PointVector transformPointCloud(const PointVector& in, const SE3F& T) {  // LidarTypes.cc:42-65
    float R[9];
    se3f_rotation_matrix(T, R);
    PointVector out(in.size());
    for (size_t i = 0; i < in.size(); ++i) {
        PointXYZINormal p;
        std::memset(&p, 0, sizeof(p));
        p.pad0 = 1.0f;  // cloudOut->resize(): default-constructed points
        const float v[3] = {in[i].x, in[i].y, in[i].z};
        p.x = (R[0] * v[0] + R[1] * v[1] + R[2] * v[2]) + T.t[0];
        p.y = (R[3] * v[0] + R[4] * v[1] + R[5] * v[2]) + T.t[1];
        p.z = (R[6] * v[0] + R[7] * v[1] + R[8] * v[2]) + T.t[2];
        p.intensity = in[i].intensity;
        out[i] = p;
    }
    return out;
}

SE3F sync_transform(const SE3F& Tcw_frame, const SE3F& Tcw_last, const SE3F& Tcw_cur, float ratio, const SE3F& Tlc, const SE3F& Tcl) {
    const SE3F cloudTwc = InterpolateSE3(se3f_inverse(Tcw_last), se3f_inverse(Tcw_cur), ratio);
    return se3f_mul(se3f_mul(se3f_mul(Tlc, Tcw_frame), cloudTwc), Tcl);  // left to right, as the expression is written
}
SE3F keyframe_transform(const SE3F& Tcw_cur, const SE3F& rel, const SE3F& Tcw_refkf, const SE3F& Tlc, const SE3F& Tcl) {
    const SE3F poseLidar_cw = se3f_mul(rel, Tcw_refkf);
    return se3f_mul(se3f_mul(se3f_mul(Tlc, Tcw_cur), se3f_inverse(poseLidar_cw)), Tcl);
}

}  // namespace oracle
