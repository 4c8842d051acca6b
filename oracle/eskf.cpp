// TEST INFRASTRUCTURE ONLY -- see eskf.hpp.
#include "eskf.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace oracle {
namespace {

constexpr int N = kEskfN;
constexpr double kTol = 1e-11;     // MTK::tolerance<double>()
constexpr double kLen = 98090.0 / 10000.0;  // S2<double, 98090, 10000, 1>::length

inline void hat(const double v[3], double H[9]) { H[0] = 0; H[1] = -v[2]; H[2] = v[1]; H[3] = v[2]; H[4] = 0; H[5] = -v[0]; H[6] = -v[1]; H[7] = v[0]; H[8] = 0; }
inline void mm3(const double* a, const double* b, double* o) { for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) o[3 * r + c] = a[3 * r] * b[c] + a[3 * r + 1] * b[3 + c] + a[3 * r + 2] * b[6 + c]; }
inline void mv3(const double* a, const double* v, double* o) { for (int r = 0; r < 3; ++r) o[r] = a[3 * r] * v[0] + a[3 * r + 1] * v[1] + a[3 * r + 2] * v[2]; }
inline void mtv3(const double* a, const double* v, double* o) { for (int r = 0; r < 3; ++r) o[r] = a[r] * v[0] + a[3 + r] * v[1] + a[6 + r] * v[2]; }

// rotation by |v| about v (the quaternion MTK::SO3::exp(v, 1) builds, as a matrix)
void so3_exp(const double v[3], double R[9]) {
    const double n2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2], n = std::sqrt(n2);
    double K[9], K2[9];
    hat(v, K);
    mm3(K, K, K2);
    double a, b;
    if (n < 1e-7) { a = 1.0 - n2 / 6.0; b = 0.5 - n2 / 24.0; }
    else { a = std::sin(n) / n; b = (1.0 - std::cos(n)) / n2; }
    for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0 ? 1.0 : 0.0) + a * K[i] + b * K2[i];
}

// MTK::SO3::log(other.conjugate() * this): quaternion of the relative rotation, then MTK::log(w, vec, scale 2, periodic)
void so3_log_rel(const double Rthis[9], const double Rother[9], double out[3]) {
    double Rt[9], R[9];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Rt[3 * r + c] = Rother[3 * c + r];
    mm3(Rt, Rthis, R);
    double q[4];  // x y z w, Eigen::Quaternion(Matrix3)
    const double t = R[0] + R[4] + R[8];
    if (t > 0) {
        double s = std::sqrt(t + 1.0);
        q[3] = 0.5 * s;
        s = 0.5 / s;
        q[0] = (R[7] - R[5]) * s; q[1] = (R[2] - R[6]) * s; q[2] = (R[3] - R[1]) * s;
    } else {
        int i = 0;
        if (R[4] > R[0]) i = 1;
        if (R[8] > R[4 * i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        double s = std::sqrt(R[4 * i] - R[4 * j] - R[4 * k] + 1.0);
        q[i] = 0.5 * s;
        s = 0.5 / s;
        q[3] = (R[3 * k + j] - R[3 * j + k]) * s;
        q[j] = (R[3 * j + i] + R[3 * i + j]) * s;
        q[k] = (R[3 * k + i] + R[3 * i + k]) * s;
    }
    double nv = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
    if (nv < kTol) nv = kTol;
    const double s = 2.0 / nv * std::atan(nv / q[3]);
    for (int k = 0; k < 3; ++k) out[k] = s * q[k];
}

// Eigen::Matrix<double, n, n>::inverse() for n > 4: partial-pivot LU, then the solve against the identity
bool invert(const std::vector<double>& A_, int n, std::vector<double>& inv) {
    std::vector<double> A(A_);
    std::vector<int> perm(n);
    for (int i = 0; i < n; ++i) perm[i] = i;
    for (int k = 0; k < n; ++k) {
        int p = k;
        double best = std::fabs(A[(size_t)k * n + k]);
        for (int r = k + 1; r < n; ++r) if (std::fabs(A[(size_t)r * n + k]) > best) { best = std::fabs(A[(size_t)r * n + k]); p = r; }
        if (best == 0) return false;
        if (p != k) { for (int c = 0; c < n; ++c) std::swap(A[(size_t)k * n + c], A[(size_t)p * n + c]); std::swap(perm[k], perm[p]); }
        for (int r = k + 1; r < n; ++r) {
            const double f = A[(size_t)r * n + k] / A[(size_t)k * n + k];
            A[(size_t)r * n + k] = f;
            for (int c = k + 1; c < n; ++c) A[(size_t)r * n + c] -= f * A[(size_t)k * n + c];
        }
    }
    inv.assign((size_t)n * n, 0.0);
    std::vector<double> y(n);
    for (int col = 0; col < n; ++col) {
        for (int r = 0; r < n; ++r) {
            double s = perm[r] == col ? 1.0 : 0.0;
            for (int c = 0; c < r; ++c) s -= A[(size_t)r * n + c] * y[c];
            y[r] = s;
        }
        for (int r = n - 1; r >= 0; --r) {
            double s = y[r];
            for (int c = r + 1; c < n; ++c) s -= A[(size_t)r * n + c] * inv[(size_t)c * n + col];
            inv[(size_t)r * n + col] = s / A[(size_t)r * n + r];
        }
    }
    return true;
}

// rows [idx, idx + d) of M (N columns) <- T (d x d) * those rows
void left_rows(double* M, int idx, int d, const double* T) {
    for (int c = 0; c < N; ++c) {
        double in[3], out[3];
        for (int r = 0; r < d; ++r) in[r] = M[(size_t)(idx + r) * N + c];
        for (int r = 0; r < d; ++r) { double s = 0; for (int k = 0; k < d; ++k) s += T[d * r + k] * in[k]; out[r] = s; }
        for (int r = 0; r < d; ++r) M[(size_t)(idx + r) * N + c] = out[r];
    }
}
// columns [idx, idx + d) of M <- those columns * T^T
void right_cols(double* M, int idx, int d, const double* T) {
    for (int r = 0; r < N; ++r) {
        double in[3], out[3];
        for (int c = 0; c < d; ++c) in[c] = M[(size_t)r * N + idx + c];
        for (int c = 0; c < d; ++c) { double s = 0; for (int k = 0; k < d; ++k) s += in[k] * T[d * c + k]; out[c] = s; }
        for (int c = 0; c < d; ++c) M[(size_t)r * N + idx + c] = out[c];
    }
}

}  // namespace

void mtk_A_matrix(const double v[3], double A[9]) {  // mtkmath.hpp A_matrix
    const double sq = v[0] * v[0] + v[1] * v[1] + v[2] * v[2], n = std::sqrt(sq);
    for (int i = 0; i < 9; ++i) A[i] = i % 4 == 0 ? 1.0 : 0.0;
    if (n < kTol) return;
    double K[9], K2[9];
    hat(v, K);
    mm3(K, K, K2);
    const double a = (1 - std::cos(n)) / sq, b = (1 - std::sin(n) / n) / sq;
    for (int i = 0; i < 9; ++i) A[i] += a * K[i] + b * K2[i];
}

void s2_Bx(const double g[3], double B[6]) {  // S2.hpp S2_Bx, S2_typ == 1
    if (g[0] + kLen > kTol) {
        const double d = kLen + g[0];
        B[0] = -g[1]; B[1] = -g[2];
        B[2] = kLen - g[1] * g[1] / d; B[3] = -g[2] * g[1] / d;
        B[4] = -g[2] * g[1] / d; B[5] = kLen - g[2] * g[2] / d;
        for (int i = 0; i < 6; ++i) B[i] /= kLen;
    } else {
        for (int i = 0; i < 6; ++i) B[i] = 0;
        B[3] = -1;  // (1, 1)
        B[4] = 1;   // (2, 0)
    }
}

void s2_Nx_yy(const double g[3], double Nx[6]) {  // 1 / length / length * Bx^T * hat(vec)
    double B[6], H[9];
    s2_Bx(g, B);
    hat(g, H);
    for (int r = 0; r < 2; ++r)
        for (int c = 0; c < 3; ++c) Nx[3 * r + c] = 1 / kLen / kLen * (B[r] * H[c] + B[2 + r] * H[3 + c] + B[4 + r] * H[6 + c]);
}

void s2_Mx(const double g[3], const double delta[2], double Mx[6]) {
    double B[6], H[9];
    s2_Bx(g, B);
    hat(g, H);
    if (std::sqrt(delta[0] * delta[0] + delta[1] * delta[1]) < kTol) {
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 2; ++c) Mx[2 * r + c] = -(H[3 * r] * B[c] + H[3 * r + 1] * B[2 + c] + H[3 * r + 2] * B[4 + c]);
        return;
    }
    // exp(Bu, scalar(1/2)) is the identity (integer division): res = -hat(vec) * A_matrix(Bu)^T * Bx
    double Bu[3], A[9], HA[9];
    for (int r = 0; r < 3; ++r) Bu[r] = B[2 * r] * delta[0] + B[2 * r + 1] * delta[1];
    mtk_A_matrix(Bu, A);
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) HA[3 * r + c] = H[3 * r] * A[3 * c] + H[3 * r + 1] * A[3 * c + 1] + H[3 * r + 2] * A[3 * c + 2];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 2; ++c) Mx[2 * r + c] = -(HA[3 * r] * B[c] + HA[3 * r + 1] * B[2 + c] + HA[3 * r + 2] * B[4 + c]);
}

void eskf_boxplus(ImuState& x, const double d[N]) {
    double E[9], Rn[9];
    for (int k = 0; k < 3; ++k) x.pos[k] += d[k];
    so3_exp(d + 3, E); mm3(x.rot, E, Rn); std::memcpy(x.rot, Rn, sizeof(Rn));
    so3_exp(d + 6, E); mm3(x.offset_R_L_I, E, Rn); std::memcpy(x.offset_R_L_I, Rn, sizeof(Rn));
    for (int k = 0; k < 3; ++k) { x.offset_T_L_I[k] += d[9 + k]; x.vel[k] += d[12 + k]; x.bg[k] += d[15 + k]; x.ba[k] += d[18 + k]; }
    double B[6], Bu[3], g[3];  // S2::boxplus
    s2_Bx(x.grav, B);
    for (int r = 0; r < 3; ++r) Bu[r] = B[2 * r] * d[21] + B[2 * r + 1] * d[22];
    so3_exp(Bu, E);
    mv3(E, x.grav, g);
    std::memcpy(x.grav, g, sizeof(g));
}

void eskf_boxminus(const ImuState& x, const ImuState& o, double d[N]) {
    for (int k = 0; k < 3; ++k) { d[k] = x.pos[k] - o.pos[k]; d[9 + k] = x.offset_T_L_I[k] - o.offset_T_L_I[k]; d[12 + k] = x.vel[k] - o.vel[k]; d[15 + k] = x.bg[k] - o.bg[k]; d[18 + k] = x.ba[k] - o.ba[k]; }
    so3_log_rel(x.rot, o.rot, d + 3);
    so3_log_rel(x.offset_R_L_I, o.offset_R_L_I, d + 6);
    // S2::boxminus
    double H[9], hv[3];
    hat(x.grav, H);
    mv3(H, o.grav, hv);
    const double v_sin = std::sqrt(hv[0] * hv[0] + hv[1] * hv[1] + hv[2] * hv[2]);
    const double v_cos = x.grav[0] * o.grav[0] + x.grav[1] * o.grav[1] + x.grav[2] * o.grav[2];
    const double theta = std::atan2(v_sin, v_cos);
    if (v_sin < kTol) {
        d[21] = std::fabs(theta) > kTol ? 3.1415926 : 0.0;
        d[22] = 0;
    } else {
        double B[6], Ho[9], t[3];
        s2_Bx(o.grav, B);
        hat(o.grav, Ho);
        mv3(Ho, x.grav, t);
        for (int r = 0; r < 2; ++r) d[21 + r] = theta / v_sin * (B[r] * t[0] + B[2 + r] * t[1] + B[4 + r] * t[2]);
    }
}

void eskf_predict(ImuState& x, double* P, const double* Q, const double acc[3], const double gyr[3], double dt) {
    // f, f_x (24 x 23), f_w (24 x 12) at the state before the step (use-ikfom.cpp:45-91)
    double omega[3], acc_[3], a_inertial[3];
    for (int k = 0; k < 3; ++k) { omega[k] = gyr[k] - x.bg[k]; acc_[k] = acc[k] - x.ba[k]; }
    mv3(x.rot, acc_, a_inertial);
    double f[24] = {0};
    for (int k = 0; k < 3; ++k) { f[k] = x.vel[k]; f[3 + k] = omega[k]; f[12 + k] = a_inertial[k] + x.grav[k]; }
    std::vector<double> fx(24 * N, 0.0), fw(24 * 12, 0.0);
    double Ha[9], RH[9], zero2[2] = {0, 0}, Mx0[6];
    hat(acc_, Ha);
    mm3(x.rot, Ha, RH);
    s2_Mx(x.grav, zero2, Mx0);
    for (int r = 0; r < 3; ++r) {
        fx[(size_t)r * N + 12 + r] = 1.0;
        fx[(size_t)(3 + r) * N + 15 + r] = -1.0;
        for (int c = 0; c < 3; ++c) { fx[(size_t)(12 + r) * N + 3 + c] = -RH[3 * r + c]; fx[(size_t)(12 + r) * N + 18 + c] = -x.rot[3 * r + c]; fw[(size_t)(12 + r) * 12 + 3 + c] = -x.rot[3 * r + c]; }
        for (int c = 0; c < 2; ++c) fx[(size_t)(12 + r) * N + 21 + c] = Mx0[2 * r + c];
        fw[(size_t)(3 + r) * 12 + r] = -1.0;
        fw[(size_t)(15 + r) * 12 + 6 + r] = 1.0;
        fw[(size_t)(18 + r) * 12 + 9 + r] = 1.0;
    }
    const ImuState before = x;
    // x_.oplus(f_, dt)
    {
        double E[9], Rn[9];
        for (int k = 0; k < 3; ++k) x.pos[k] += x.vel[k] * dt;
        const double wdt[3] = {omega[0] * dt, omega[1] * dt, omega[2] * dt};
        so3_exp(wdt, E);
        mm3(x.rot, E, Rn);
        for (int k = 0; k < 3; ++k) x.vel[k] += (a_inertial[k] + x.grav[k]) * dt;
        std::memcpy(x.rot, Rn, sizeof(Rn));
        // offset_R_L_I, offset_T_L_I, bg, ba, grav: f = 0
    }
    std::vector<double> F((size_t)N * N, 0.0), fxf((size_t)N * N, 0.0), fwf((size_t)N * 12, 0.0);
    for (int i = 0; i < N; ++i) F[(size_t)i * N + i] = 1.0;
    const int vect_idx[5] = {0, 9, 12, 15, 18};
    for (int v : vect_idx)
        for (int j = 0; j < 3; ++j) {
            for (int i = 0; i < N; ++i) fxf[(size_t)(v + j) * N + i] = fx[(size_t)(v + j) * N + i];
            for (int i = 0; i < 12; ++i) fwf[(size_t)(v + j) * 12 + i] = fw[(size_t)(v + j) * 12 + i];
        }
    const int so3_idx[2] = {3, 6};
    for (int s : so3_idx) {
        double seg[3], A[9];
        for (int i = 0; i < 3; ++i) seg[i] = -1 * f[s + i] * dt;
        // res = exp(seg, scalar(1/2)) = identity: the diagonal block of F_x1 stays the identity
        mtk_A_matrix(seg, A);
        for (int i = 0; i < N; ++i) {
            const double c[3] = {fx[(size_t)s * N + i], fx[(size_t)(s + 1) * N + i], fx[(size_t)(s + 2) * N + i]};
            for (int r = 0; r < 3; ++r) fxf[(size_t)(s + r) * N + i] = A[3 * r] * c[0] + A[3 * r + 1] * c[1] + A[3 * r + 2] * c[2];
        }
        for (int i = 0; i < 12; ++i) {
            const double c[3] = {fw[(size_t)s * 12 + i], fw[(size_t)(s + 1) * 12 + i], fw[(size_t)(s + 2) * 12 + i]};
            for (int r = 0; r < 3; ++r) fwf[(size_t)(s + r) * 12 + i] = A[3 * r] * c[0] + A[3 * r + 1] * c[1] + A[3 * r + 2] * c[2];
        }
    }
    {   // S2 state (grav): seg = f(21..23) dt = 0
        double seg[3], Nx[6], Mx[6], Hb[9], A[9], NH[6], T[6];
        for (int i = 0; i < 3; ++i) seg[i] = f[21 + i] * dt;
        s2_Nx_yy(x.grav, Nx);
        s2_Mx(before.grav, zero2, Mx);
        for (int r = 0; r < 2; ++r)
            for (int c = 0; c < 2; ++c) F[(size_t)(21 + r) * N + 21 + c] = Nx[3 * r] * Mx[c] + Nx[3 * r + 1] * Mx[2 + c] + Nx[3 * r + 2] * Mx[4 + c];
        hat(before.grav, Hb);
        mtk_A_matrix(seg, A);
        for (int r = 0; r < 2; ++r) for (int c = 0; c < 3; ++c) NH[3 * r + c] = -(Nx[3 * r] * Hb[c] + Nx[3 * r + 1] * Hb[3 + c] + Nx[3 * r + 2] * Hb[6 + c]);
        for (int r = 0; r < 2; ++r) for (int c = 0; c < 3; ++c) T[3 * r + c] = NH[3 * r] * A[3 * c] + NH[3 * r + 1] * A[3 * c + 1] + NH[3 * r + 2] * A[3 * c + 2];
        for (int i = 0; i < N; ++i)
            for (int r = 0; r < 2; ++r) fxf[(size_t)(21 + r) * N + i] = T[3 * r] * fx[(size_t)21 * N + i] + T[3 * r + 1] * fx[(size_t)22 * N + i] + T[3 * r + 2] * fx[(size_t)23 * N + i];
        for (int i = 0; i < 12; ++i)
            for (int r = 0; r < 2; ++r) fwf[(size_t)(21 + r) * 12 + i] = T[3 * r] * fw[(size_t)21 * 12 + i] + T[3 * r + 1] * fw[(size_t)22 * 12 + i] + T[3 * r + 2] * fw[(size_t)23 * 12 + i];
    }
    for (size_t i = 0; i < F.size(); ++i) F[i] += fxf[i] * dt;
    // P = F P F^T + (dt G) Q (dt G)^T
    std::vector<double> FP((size_t)N * N), Pn((size_t)N * N), G((size_t)N * 12), GQ((size_t)N * 12);
    for (int r = 0; r < N; ++r) for (int c = 0; c < N; ++c) { double s = 0; for (int k = 0; k < N; ++k) s += F[(size_t)r * N + k] * P[(size_t)k * N + c]; FP[(size_t)r * N + c] = s; }
    for (int r = 0; r < N; ++r) for (int c = 0; c < N; ++c) { double s = 0; for (int k = 0; k < N; ++k) s += FP[(size_t)r * N + k] * F[(size_t)c * N + k]; Pn[(size_t)r * N + c] = s; }
    for (size_t i = 0; i < G.size(); ++i) G[i] = dt * fwf[i];
    for (int r = 0; r < N; ++r) for (int c = 0; c < 12; ++c) { double s = 0; for (int k = 0; k < 12; ++k) s += G[(size_t)r * 12 + k] * Q[k * 12 + c]; GQ[(size_t)r * 12 + c] = s; }
    for (int r = 0; r < N; ++r) for (int c = 0; c < N; ++c) { double s = 0; for (int k = 0; k < 12; ++k) s += GQ[(size_t)r * 12 + k] * G[(size_t)c * 12 + k]; P[(size_t)r * N + c] = Pn[(size_t)r * N + c] + s; }
}

namespace {
struct DynShare { bool valid = true, converge = true; std::vector<double> h_x, h; };

// h_share_model (LidarFrontEnd.cpp:485-602); Nearest_Points / point_selected_surf persist between the calls of one update
struct MeasurementModel {
    const KdTree& tree;
    const PointVector& body;
    bool extrinsic_est_en;
    std::vector<PointVector> Nearest_Points;
    std::vector<uint8_t> point_selected_surf;
    PointVector normvec;
    int effct_feat_num = 0, searches = 0;
    double res_mean_last = 0;
    MeasurementModel(const KdTree& t, const PointVector& b, bool ext) : tree(t), body(b), extrinsic_est_en(ext) {
        Nearest_Points.resize(b.size());
        point_selected_surf.assign(b.size(), 0);
        PointXYZINormal blank;
        std::memset(&blank, 0, sizeof(blank));
        normvec.assign(b.size(), blank);
    }
    void operator()(const ImuState& s, DynShare& d) {
        const int n = (int)body.size();
        LidarState ls;
        std::memcpy(ls.rot, s.rot, 72); std::memcpy(ls.pos, s.pos, 24); std::memcpy(ls.offset_R_L_I, s.offset_R_L_I, 72); std::memcpy(ls.offset_T_L_I, s.offset_T_L_I, 24);
        std::vector<float> res_last(n, 0.f);
        if (d.converge) ++searches;
        for (int i = 0; i < n; ++i) {
            const PointXYZINormal& pb = body[i];
            const PointXYZINormal pw = pointBodyToWorld(pb, ls);
            if (d.converge) {
                std::vector<float> sq(5);
                tree.Nearest_Search(pw, 5, Nearest_Points[i], sq);
                point_selected_surf[i] = Nearest_Points[i].size() < 5 ? 0 : (sq[4] > 5 ? 0 : 1);
            }
            if (!point_selected_surf[i]) continue;
            float pabcd[4];
            point_selected_surf[i] = 0;
            if (EstiPlane(pabcd, Nearest_Points[i], 0.1f)) {
                const float pd2 = pabcd[0] * pw.x + pabcd[1] * pw.y + pabcd[2] * pw.z + pabcd[3];
                const double p[3] = {pb.x, pb.y, pb.z};
                const float sc = (float)(1 - 0.9 * std::fabs(pd2) / std::sqrt(std::sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2])));
                if (sc > 0.9) {
                    point_selected_surf[i] = 1;
                    normvec[i].x = pabcd[0]; normvec[i].y = pabcd[1]; normvec[i].z = pabcd[2]; normvec[i].intensity = pd2;
                    res_last[i] = std::fabs(pd2);
                }
            }
        }
        effct_feat_num = 0;
        double total_residual = 0;
        std::vector<int> sel;
        for (int i = 0; i < n; ++i) if (point_selected_surf[i]) { sel.push_back(i); total_residual += res_last[i]; effct_feat_num++; }
        if (effct_feat_num < 1) { d.valid = false; return; }
        res_mean_last = total_residual / effct_feat_num;
        d.h_x.assign((size_t)effct_feat_num * 12, 0.0);
        d.h.assign(effct_feat_num, 0.0);
        for (int k = 0; k < effct_feat_num; ++k) {
            const PointXYZINormal& lp = body[sel[k]];
            const PointXYZINormal& np = normvec[sel[k]];
            const double pbe[3] = {lp.x, lp.y, lp.z}, nv[3] = {np.x, np.y, np.z};
            double pt[3], Hbe[9], Hp[9], C[3], A[3];
            hat(pbe, Hbe);
            mv3(s.offset_R_L_I, pbe, pt);
            for (int c = 0; c < 3; ++c) pt[c] += s.offset_T_L_I[c];
            hat(pt, Hp);
            mtv3(s.rot, nv, C);  // s.rot.conjugate() * norm_vec
            mv3(Hp, C, A);
            double* row = &d.h_x[(size_t)k * 12];
            row[0] = np.x; row[1] = np.y; row[2] = np.z;
            for (int c = 0; c < 3; ++c) row[3 + c] = A[c];
            if (extrinsic_est_en) {
                double RtC[3], B[3];
                mtv3(s.offset_R_L_I, C, RtC);
                mv3(Hbe, RtC, B);
                for (int c = 0; c < 3; ++c) { row[6 + c] = B[c]; row[9 + c] = C[c]; }
            }
            d.h[k] = -np.intensity;
        }
    }
};
}  // namespace

EskfUpdate eskf_update(ImuState& x, double* P_, const KdTree& tree, const PointVector& body, double R, int maximum_iter, const double* limit,
                       bool extrinsic_est_en) {
    EskfUpdate out;
    MeasurementModel model(tree, body, extrinsic_est_en);
    DynShare dyn;
    int t = 0;
    const ImuState x_propagated = x;
    const std::vector<double> P_propagated(P_, P_ + N * N);
    std::vector<double> P(P_propagated), K_h(N, 0.0), K_x((size_t)N * N, 0.0), dx_new(N, 0.0);
    for (int i = -1; i < maximum_iter; i++) {
        dyn.valid = true;
        model(x, dyn);
        ++out.calls;
        out.effct_feat_num = model.effct_feat_num; out.res_mean_last = model.res_mean_last; out.searches = model.searches;
        if (!dyn.valid) continue;
        const int M = (int)dyn.h.size();
        const std::vector<double>& H = dyn.h_x;
        double dx[N];
        eskf_boxminus(x, x_propagated, dx);
        for (int k = 0; k < N; ++k) dx_new[k] = dx[k];
        P = P_propagated;
        const int so3_idx[2] = {3, 6};
        for (int s : so3_idx) {
            double A[9], At[9], v[3];
            mtk_A_matrix(dx + s, A);
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) At[3 * r + c] = A[3 * c + r];
            mv3(At, &dx_new[s], v);
            for (int k = 0; k < 3; ++k) dx_new[s + k] = v[k];
            left_rows(P.data(), s, 3, At);
            right_cols(P.data(), s, 3, At);
        }
        {
            double Nx[6], Mx[6], T[4], v[2];
            s2_Nx_yy(x.grav, Nx);
            s2_Mx(x_propagated.grav, dx + 21, Mx);
            for (int r = 0; r < 2; ++r) for (int c = 0; c < 2; ++c) T[2 * r + c] = Nx[3 * r] * Mx[c] + Nx[3 * r + 1] * Mx[2 + c] + Nx[3 * r + 2] * Mx[4 + c];
            for (int r = 0; r < 2; ++r) v[r] = T[2 * r] * dx_new[21] + T[2 * r + 1] * dx_new[22];
            dx_new[21] = v[0]; dx_new[22] = v[1];
            left_rows(P.data(), 21, 2, T);
            right_cols(P.data(), 21, 2, T);
        }
        if (N > M) {
            // K = P Hc^T (Hc P Hc^T / R + I)^-1 / R with Hc = [H 0]
            std::vector<double> PHt((size_t)N * M), S((size_t)M * M), Si, K((size_t)N * M);
            for (int r = 0; r < N; ++r) for (int c = 0; c < M; ++c) { double s = 0; for (int k = 0; k < 12; ++k) s += P[(size_t)r * N + k] * H[(size_t)c * 12 + k]; PHt[(size_t)r * M + c] = s; }
            for (int r = 0; r < M; ++r) for (int c = 0; c < M; ++c) { double s = 0; for (int k = 0; k < 12; ++k) s += H[(size_t)r * 12 + k] * PHt[(size_t)k * M + c]; S[(size_t)r * M + c] = s / R + (r == c ? 1.0 : 0.0); }
            invert(S, M, Si);
            for (int r = 0; r < N; ++r) for (int c = 0; c < M; ++c) { double s = 0; for (int k = 0; k < M; ++k) s += PHt[(size_t)r * M + k] * Si[(size_t)k * M + c]; K[(size_t)r * M + c] = s / R; }
            for (int r = 0; r < N; ++r) { double s = 0; for (int k = 0; k < M; ++k) s += K[(size_t)r * M + k] * dyn.h[k]; K_h[r] = s; }
            std::fill(K_x.begin(), K_x.end(), 0.0);
            for (int r = 0; r < N; ++r) for (int c = 0; c < 12; ++c) { double s = 0; for (int k = 0; k < M; ++k) s += K[(size_t)r * M + k] * H[(size_t)k * 12 + c]; K_x[(size_t)r * N + c] = s; }
        } else {
            std::vector<double> PR((size_t)N * N), P_temp, P_inv;
            for (size_t k = 0; k < PR.size(); ++k) PR[k] = P[k] / R;
            invert(PR, N, P_temp);
            double HTH[144], HTh[12];
            for (int r = 0; r < 12; ++r) {
                for (int c = 0; c < 12; ++c) { double s = 0; for (int k = 0; k < M; ++k) s += H[(size_t)k * 12 + r] * H[(size_t)k * 12 + c]; HTH[12 * r + c] = s; }
                double s = 0;
                for (int k = 0; k < M; ++k) s += H[(size_t)k * 12 + r] * dyn.h[k];
                HTh[r] = s;
            }
            for (int r = 0; r < 12; ++r) for (int c = 0; c < 12; ++c) P_temp[(size_t)r * N + c] += HTH[12 * r + c];
            invert(P_temp, N, P_inv);
            for (int r = 0; r < N; ++r) { double s = 0; for (int k = 0; k < 12; ++k) s += P_inv[(size_t)r * N + k] * HTh[k]; K_h[r] = s; }
            std::fill(K_x.begin(), K_x.end(), 0.0);
            for (int r = 0; r < N; ++r) for (int c = 0; c < 12; ++c) { double s = 0; for (int k = 0; k < 12; ++k) s += P_inv[(size_t)r * N + k] * HTH[12 * k + c]; K_x[(size_t)r * N + c] = s; }
        }
        double dx_[N];
        for (int r = 0; r < N; ++r) {
            double s = K_h[r];
            for (int c = 0; c < N; ++c) s += (K_x[(size_t)r * N + c] - (r == c ? 1.0 : 0.0)) * dx_new[c];
            dx_[r] = s;
        }
        eskf_boxplus(x, dx_);
        dyn.converge = true;
        for (int k = 0; k < N; ++k) if (std::fabs(dx_[k]) > limit[k]) { dyn.converge = false; break; }
        if (dyn.converge) t++;
        if (!t && i == maximum_iter - 2) dyn.converge = true;
        if (t > 1 || i == maximum_iter - 1) {
            std::vector<double> L(P);
            for (int s : so3_idx) {
                double A[9], At[9];
                mtk_A_matrix(dx_ + s, A);
                for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) At[3 * r + c] = A[3 * c + r];
                for (int c = 0; c < N; ++c)  // L rows <- At * P rows
                    for (int r = 0; r < 3; ++r) L[(size_t)(s + r) * N + c] = At[3 * r] * P[(size_t)s * N + c] + At[3 * r + 1] * P[(size_t)(s + 1) * N + c] + At[3 * r + 2] * P[(size_t)(s + 2) * N + c];
                for (int c = 0; c < 12; ++c) {
                    const double k3[3] = {K_x[(size_t)s * N + c], K_x[(size_t)(s + 1) * N + c], K_x[(size_t)(s + 2) * N + c]};
                    for (int r = 0; r < 3; ++r) K_x[(size_t)(s + r) * N + c] = At[3 * r] * k3[0] + At[3 * r + 1] * k3[1] + At[3 * r + 2] * k3[2];
                }
                right_cols(L.data(), s, 3, At);
                right_cols(P.data(), s, 3, At);
            }
            {
                double Nx[6], Mx[6], T[4];
                s2_Nx_yy(x.grav, Nx);
                s2_Mx(x_propagated.grav, dx_ + 21, Mx);
                for (int r = 0; r < 2; ++r) for (int c = 0; c < 2; ++c) T[2 * r + c] = Nx[3 * r] * Mx[c] + Nx[3 * r + 1] * Mx[2 + c] + Nx[3 * r + 2] * Mx[4 + c];
                for (int c = 0; c < N; ++c)
                    for (int r = 0; r < 2; ++r) L[(size_t)(21 + r) * N + c] = T[2 * r] * P[(size_t)21 * N + c] + T[2 * r + 1] * P[(size_t)22 * N + c];
                for (int c = 0; c < 12; ++c) {
                    const double k2[2] = {K_x[(size_t)21 * N + c], K_x[(size_t)22 * N + c]};
                    for (int r = 0; r < 2; ++r) K_x[(size_t)(21 + r) * N + c] = T[2 * r] * k2[0] + T[2 * r + 1] * k2[1];
                }
                right_cols(L.data(), 21, 2, T);
                right_cols(P.data(), 21, 2, T);
            }
            for (int r = 0; r < N; ++r)
                for (int c = 0; c < N; ++c) {
                    double s = 0;
                    for (int k = 0; k < 12; ++k) s += K_x[(size_t)r * N + k] * P[(size_t)k * N + c];
                    P_[(size_t)r * N + c] = L[(size_t)r * N + c] - s;
                }
            out.converged = t;
            out.finished = true;
            return out;
        }
    }
    out.converged = t;
    std::memcpy(P_, P.data(), sizeof(double) * N * N);  // P_ as the last iteration left it (P_propagated with the chart corrections)
    return out;
}

std::vector<Pose6D> ForwardPropagateCov(ImuState& st, double* P, const double cov[12], const std::vector<ImuMeas>& v_imu, double pcl_beg_time,
                                        double pcl_end_time, double last_lidar_end_time, double acc_scale, const double acc_s_last[3],
                                        const double angvel_last[3]) {
    std::vector<Pose6D> out;
    auto save = [&](double t, const double* acc, const double* gyr) {
        Pose6D p;
        p.offset_time = t;
        std::memcpy(p.acc, acc, 24); std::memcpy(p.gyr, gyr, 24); std::memcpy(p.vel, st.vel, 24); std::memcpy(p.pos, st.pos, 24); std::memcpy(p.rot, st.rot, 72);
        out.push_back(p);
    };
    save(0.0, acc_s_last, angvel_last);
    double Q[144] = {0};
    for (int k = 0; k < 12; ++k) Q[13 * k] = cov[k];
    double acc_avr[3] = {0, 0, 0}, angvel_avr[3] = {0, 0, 0};
    for (size_t i = 0; i + 1 < v_imu.size(); ++i) {
        const ImuMeas& head = v_imu[i];
        const ImuMeas& tail = v_imu[i + 1];
        if (tail.t < last_lidar_end_time) continue;
        for (int k = 0; k < 3; ++k) { angvel_avr[k] = 0.5 * (head.gyr[k] + tail.gyr[k]); acc_avr[k] = 0.5 * (head.acc[k] + tail.acc[k]) * acc_scale; }
        const double dt = head.t < last_lidar_end_time ? tail.t - last_lidar_end_time : tail.t - head.t;
        eskf_predict(st, P, Q, acc_avr, angvel_avr, dt);
        double gl[3], al[3], am[3];
        for (int k = 0; k < 3; ++k) { gl[k] = angvel_avr[k] - st.bg[k]; am[k] = acc_avr[k] - st.ba[k]; }
        mv3(st.rot, am, al);
        for (int k = 0; k < 3; ++k) al[k] += st.grav[k];
        save(tail.t - pcl_beg_time, al, gl);
    }
    const double imu_end_time = v_imu.back().t;
    const double note = pcl_end_time > imu_end_time ? 1.0 : -1.0;
    eskf_predict(st, P, Q, acc_avr, angvel_avr, note * (pcl_end_time - imu_end_time));
    return out;
}

}  // namespace oracle
