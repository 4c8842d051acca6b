// Pose plumbing between the camera thread and the LiDAR front end, behind tc2li_lidar_update_pose,
// tc2li_se3_interpolate, tc2li_lidar_sync_transform, tc2li_lidar_keyframe_transform and tc2li_transform_point_cloud(_batch):
//   UpdateLidarPose                         SF/include/lidar_front_end/LidarFrontEnd.cpp:786-800
//   InterpolateSE3                          SF/src/Tracking.cc:1552-1563
//   Tracking::SyncWithLidar                 SF/src/Tracking.cc:1565-1630  (which frame a scan pairs with stays with the caller; this is the chain
//                                                                          Tlc * Tcw(frame) * cloudTwc * Tcl it applies to the cloud)
//   Tracking::BuildLidarFeat4KeyFrame       SF/src/Tracking.cc:1510-1550  (Tlc * Tcw(cur) * (rel * Tcw(refKF))^-1 * Tcl)
//   LidarFrontEndTools::transformPointCloud SF/src/LidarTypes.cc:42-65
// The pose algebra is a handful of Sophus::SE3f operations per frame (host, float, in the order Sophus / Eigen evaluate them); the
// clouds are streamed through one kernel for all scans of a batch, reading the front end's device-resident selections in place.
#include <cmath>
#include <cstring>
#include <vector>

#include "common.hpp"
#include "lidar_device.hpp"

using namespace tc2li;

namespace {

constexpr float kSophusEps = 1e-5f;  // Sophus::Constants<float>::epsilon()
constexpr float kPiF = 3.141592653589793238462643383279502884f;

struct Quat { float x, y, z, w; };
struct Vec3 { float v[3]; };
struct Pose {  // Sophus::SE3f
    Quat q{0, 0, 0, 1};
    Vec3 t{{0, 0, 0}};
    static Pose from7(const float* p) { Pose T; T.q = Quat{p[0], p[1], p[2], p[3]}; memcpy(T.t.v, p + 4, 12); return T; }
    void to7(float* p) const { p[0] = q.x; p[1] = q.y; p[2] = q.z; p[3] = q.w; memcpy(p + 4, t.v, 12); }
};

Quat unit(Quat q) {  // SO3::normalize
    const float len = std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    return Quat{q.x / len, q.y / len, q.z / len, q.w / len};
}
Vec3 spin(const Quat& q, const Vec3& p) {  // QuaternionBase::_transformVector
    float a[3] = {q.y * p.v[2] - q.z * p.v[1], q.z * p.v[0] - q.x * p.v[2], q.x * p.v[1] - q.y * p.v[0]};
    for (float& c : a) c = c + c;
    const float b[3] = {q.y * a[2] - q.z * a[1], q.z * a[0] - q.x * a[2], q.x * a[1] - q.y * a[0]};
    return Vec3{{p.v[0] + q.w * a[0] + b[0], p.v[1] + q.w * a[1] + b[1], p.v[2] + q.w * a[2] + b[2]}};
}
struct Mat3 { float m[9]; };
Mat3 rotation_of(const Quat& q) {  // toRotationMatrix
    const float tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
    const float twx = tx * q.w, twy = ty * q.w, twz = tz * q.w, txx = tx * q.x, txy = ty * q.x, txz = tz * q.x, tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    return Mat3{{1 - (tyy + tzz), txy - twz, txz + twy, txy + twz, 1 - (txx + tzz), tyz - twx, txz - twy, tyz + twx, 1 - (txx + tyy)}};
}
Mat3 skew(const float w[3]) { return Mat3{{0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0}}; }
Mat3 mm(const Mat3& a, const Mat3& b) {
    Mat3 o;
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) o.m[3 * r + c] = a.m[3 * r] * b.m[c] + a.m[3 * r + 1] * b.m[3 + c] + a.m[3 * r + 2] * b.m[6 + c];
    return o;
}
Vec3 mv(const Mat3& a, const float* v) {
    return Vec3{{a.m[0] * v[0] + a.m[1] * v[1] + a.m[2] * v[2], a.m[3] * v[0] + a.m[4] * v[1] + a.m[5] * v[2], a.m[6] * v[0] + a.m[7] * v[1] + a.m[8] * v[2]}};
}
Pose inverse(const Pose& T) {
    Pose o;
    o.q = unit(Quat{-T.q.x, -T.q.y, -T.q.z, T.q.w});
    o.t = spin(o.q, Vec3{{T.t.v[0] * -1.0f, T.t.v[1] * -1.0f, T.t.v[2] * -1.0f}});
    return o;
}
Pose compose(const Pose& A, const Pose& B) {
    Pose o;
    const Quat &a = A.q, &b = B.q;
    o.q = unit(Quat{a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
                    a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z});
    const Vec3 r = spin(A.q, B.t);
    for (int k = 0; k < 3; ++k) o.t.v[k] = A.t.v[k] + r.v[k];
    return o;
}
void log6(const Pose& T, float out[6]) {
    const Quat& q = T.q;
    const float n2 = q.x * q.x + q.y * q.y + q.z * q.z;
    float k, theta;
    if (n2 < kSophusEps * kSophusEps) {
        k = 2.0f / q.w - (float)(2.0 / 3.0) * n2 / (q.w * (q.w * q.w));
        theta = 2.0f * n2 / q.w;
    } else {
        const float n = std::sqrt(n2);
        if (std::fabs(q.w) < kSophusEps) k = q.w > 0.0f ? kPiF / n : -kPiF / n;
        else k = 2.0f * std::atan(n / q.w) / n;
        theta = k * n;
    }
    const float om[3] = {k * q.x, k * q.y, k * q.z};
    const Mat3 O = skew(om), O2 = mm(O, O);
    Mat3 Vinv;
    float c;
    if (std::fabs(theta) < kSophusEps) c = (float)(1. / 12.);
    else { const float h = 0.5f * theta; c = (1.0f - theta * std::cos(h) / (2.0f * std::sin(h))) / (theta * theta); }
    for (int i = 0; i < 9; ++i) Vinv.m[i] = (i % 4 == 0 ? 1.0f : 0.0f) - 0.5f * O.m[i] + c * O2.m[i];
    const Vec3 u = mv(Vinv, T.t.v);
    memcpy(out, u.v, 12); memcpy(out + 3, om, 12);
}
Pose exp6(const float a[6]) {
    Pose o;
    const float* om = a + 3;
    const float th2 = om[0] * om[0] + om[1] * om[1] + om[2] * om[2];
    float theta, im, re;
    if (th2 < kSophusEps * kSophusEps) {
        theta = 0;
        const float th4 = th2 * th2;
        im = 0.5f - (float)(1.0 / 48.0) * th2 + (float)(1.0 / 3840.0) * th4;
        re = 1.0f - (float)(1.0 / 8.0) * th2 + (float)(1.0 / 384.0) * th4;
    } else {
        theta = std::sqrt(th2);
        const float h = 0.5f * theta;
        im = std::sin(h) / theta;
        re = std::cos(h);
    }
    o.q = Quat{im * om[0], im * om[1], im * om[2], re};
    Mat3 V;
    if (theta < kSophusEps) {
        V = rotation_of(o.q);
    } else {
        const Mat3 O = skew(om), O2 = mm(O, O);
        const float t2 = theta * theta, c1 = (1.0f - std::cos(theta)) / t2, c2 = (theta - std::sin(theta)) / (t2 * theta);
        for (int i = 0; i < 9; ++i) V.m[i] = (i % 4 == 0 ? 1.0f : 0.0f) + c1 * O.m[i] + c2 * O2.m[i];
    }
    o.t = mv(V, a);
    return o;
}
Quat quat_of(const Mat3& M) {  // Eigen: Quaternionf(Matrix3f)
    const float* m = M.m;
    float q[4];
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
    return Quat{q[0], q[1], q[2], q[3]};
}
Pose interpolate(const Pose& A, const Pose& B, float t) {
    const Quat p = quat_of(rotation_of(A.q)), r = quat_of(rotation_of(B.q));
    const float one = 1.0f - 1.1920928955078125e-07f;  // 1 - NumTraits<float>::epsilon()
    const float d = p.x * r.x + p.y * r.y + p.z * r.z + p.w * r.w, ad = std::fabs(d);
    float s0, s1;
    if (ad >= one) { s0 = 1.0f - t; s1 = t; }
    else { const float th = std::acos(ad), sn = std::sin(th); s0 = std::sin((1.0f - t) * th) / sn; s1 = std::sin(t * th) / sn; }
    if (d < 0.0f) s1 = -s1;
    Pose o;
    o.q = unit(Quat{s0 * p.x + s1 * r.x, s0 * p.y + s1 * r.y, s0 * p.z + s1 * r.z, s0 * p.w + s1 * r.w});
    for (int k = 0; k < 3; ++k) o.t.v[k] = A.t.v[k] + t * (B.t.v[k] - A.t.v[k]);
    return o;
}

struct XformWorkspace {
    DevBuf<PointXYZINormal> d_in, d_out;
    DevBuf<TransformTask> d_tasks;
    std::mutex mu;
};
XformWorkspace& xws() { static thread_local XformWorkspace w; return w; }

}  // namespace

namespace tc2li {
void fill_transform_task(TransformTask& t, const float T7[7]) {  // transformIn.rotationMatrix() / translation()
    const Pose T = Pose::from7(T7);
    const Mat3 R = rotation_of(T.q);
    memcpy(t.R, R.m, 36); memcpy(t.t, T.t.v, 12);
}
}  // namespace tc2li

extern "C" {

int tc2li_lidar_update_pose(const float Tcw_last7[7], const float velocity7[7], double time_from_last_frame, const float Tcl7[7], tc2li_lidar_state* state,
                            double pos_lid[3]) {
    if (!Tcw_last7 || !velocity7 || !Tcl7 || !state) { set_error("tc2li_lidar_update_pose: invalid argument"); return TC2LI_ERR_INVALID; }
    float lg[6], a[6];
    log6(inverse(Pose::from7(velocity7)), lg);
    const float s = (float)time_from_last_frame;
    for (int k = 0; k < 6; ++k) a[k] = s * lg[k];
    const Pose Twc = compose(inverse(Pose::from7(Tcw_last7)), exp6(a)), Tcl = Pose::from7(Tcl7);
    const Mat3 Rwc = rotation_of(Twc.q), Rcl = rotation_of(Tcl.q);
    float M[12];  // the upper three rows of Twc.matrix() * Tcl.matrix()
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) M[4 * r + c] = ((Rwc.m[3 * r] * Rcl.m[c] + Rwc.m[3 * r + 1] * Rcl.m[3 + c]) + Rwc.m[3 * r + 2] * Rcl.m[6 + c]) + Twc.t.v[r] * 0.0f;
        M[4 * r + 3] = ((Rwc.m[3 * r] * Tcl.t.v[0] + Rwc.m[3 * r + 1] * Tcl.t.v[1]) + Rwc.m[3 * r + 2] * Tcl.t.v[2]) + Twc.t.v[r] * 1.0f;
    }
    // Rw2_w1 = [0 0 1; -1 0 0; 0 -1 0] picks rows 2, 0, 1 with signs +, -, -
    static const int row[3] = {2, 0, 1};
    static const float sign[3] = {1.0f, -1.0f, -1.0f};
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) state->rot[3 * r + c] = (double)(sign[r] * M[4 * row[r] + c]);
        state->pos[r] = (double)(sign[r] * M[4 * row[r] + 3]);
    }
    if (pos_lid)
        for (int r = 0; r < 3; ++r)
            pos_lid[r] = state->pos[r] + (state->rot[3 * r] * state->offset_T_L_I[0] + state->rot[3 * r + 1] * state->offset_T_L_I[1] + state->rot[3 * r + 2] * state->offset_T_L_I[2]);
    return TC2LI_OK;
}

int tc2li_se3_interpolate(const float a7[7], const float b7[7], float t, float out7[7]) {
    if (!a7 || !b7 || !out7) { set_error("tc2li_se3_interpolate: invalid argument"); return TC2LI_ERR_INVALID; }
    interpolate(Pose::from7(a7), Pose::from7(b7), t).to7(out7);
    return TC2LI_OK;
}

int tc2li_lidar_sync_transform(const float Tcw_frame7[7], const float Tcw_last7[7], const float Tcw_cur7[7], float ratio, const float Tlc7[7],
                               const float Tcl7[7], float out7[7]) {
    if (!Tcw_frame7 || !Tcw_last7 || !Tcw_cur7 || !Tlc7 || !Tcl7 || !out7) { set_error("tc2li_lidar_sync_transform: invalid argument"); return TC2LI_ERR_INVALID; }
    const Pose cloudTwc = interpolate(inverse(Pose::from7(Tcw_last7)), inverse(Pose::from7(Tcw_cur7)), ratio);
    compose(compose(compose(Pose::from7(Tlc7), Pose::from7(Tcw_frame7)), cloudTwc), Pose::from7(Tcl7)).to7(out7);
    return TC2LI_OK;
}

int tc2li_lidar_keyframe_transform(const float Tcw_cur7[7], const float rel7[7], const float Tcw_refkf7[7], const float Tlc7[7], const float Tcl7[7],
                                   float out7[7]) {
    if (!Tcw_cur7 || !rel7 || !Tcw_refkf7 || !Tlc7 || !Tcl7 || !out7) { set_error("tc2li_lidar_keyframe_transform: invalid argument"); return TC2LI_ERR_INVALID; }
    const Pose scan_cw = compose(Pose::from7(rel7), Pose::from7(Tcw_refkf7));
    compose(compose(compose(Pose::from7(Tlc7), Pose::from7(Tcw_cur7)), inverse(scan_cw)), Pose::from7(Tcl7)).to7(out7);
    return TC2LI_OK;
}

int tc2li_transform_point_cloud(const tc2li_point* in, int n, const float T7[7], tc2li_point* out, void* stream_) {
    if (n < 0 || (n > 0 && (!in || !out)) || !T7) { set_error("tc2li_transform_point_cloud: invalid argument"); return TC2LI_ERR_INVALID; }
    if (n == 0) return 0;
    if (!device_ready()) return TC2LI_ERR_NO_DEVICE;
    hipStream_t st = stream_ ? (hipStream_t)stream_ : private_stream();
    XformWorkspace& w = xws();
    std::lock_guard<std::mutex> lk(w.mu);
    TC2LI_HIP_CHECK(w.d_in.ensure(n)); TC2LI_HIP_CHECK(w.d_out.ensure(n)); TC2LI_HIP_CHECK(w.d_tasks.ensure(1));
    TransformTask t{};
    t.in = w.d_in.p; t.out = w.d_out.p; t.n = n;
    fill_transform_task(t, T7);
    TC2LI_HIP_CHECK(hipMemcpyAsync(w.d_in.p, in, (size_t)n * sizeof(PointXYZINormal), hipMemcpyHostToDevice, st));
    TC2LI_HIP_CHECK(hipMemcpyAsync(w.d_tasks.p, &t, sizeof(t), hipMemcpyHostToDevice, st));
    launch_transform_points(w.d_tasks.p, 1, n, st);
    TC2LI_HIP_CHECK(hipGetLastError());
    TC2LI_HIP_CHECK(hipMemcpyAsync(out, w.d_out.p, (size_t)n * sizeof(PointXYZINormal), hipMemcpyDeviceToHost, st));
    TC2LI_HIP_CHECK(stream_wait_blocking(st));
    return n;
}

}  // extern "C"
