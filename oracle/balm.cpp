// TEST INFRASTRUCTURE ONLY -- CPU oracle, never linked into or called from the product path.
// See balm.hpp for the reference locations.
#include "balm.hpp"

#include <array>
#include <cmath>
#include <cstring>

namespace oracle {

// ---- small algebra ------------------------------------------------------------------------------------------------
static inline V3 operator+(const V3& a, const V3& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline V3 operator-(const V3& a, const V3& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline V3 operator*(double s, const V3& a) { return {s * a.x, s * a.y, s * a.z}; }
static inline double dot(const V3& a, const V3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline M3 operator+(const M3& a, const M3& b) { M3 r; for (int i = 0; i < 9; ++i) r.m[i] = a.m[i] + b.m[i]; return r; }
static inline M3 operator-(const M3& a, const M3& b) { M3 r; for (int i = 0; i < 9; ++i) r.m[i] = a.m[i] - b.m[i]; return r; }
static inline M3 operator*(double s, const M3& a) { M3 r; for (int i = 0; i < 9; ++i) r.m[i] = s * a.m[i]; return r; }
static inline M3 operator*(const M3& a, const M3& b) {
    M3 r;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r(i, j) = a(i, 0) * b(0, j) + a(i, 1) * b(1, j) + a(i, 2) * b(2, j);
    return r;
}
static inline V3 operator*(const M3& a, const V3& v) { return {a(0, 0) * v.x + a(0, 1) * v.y + a(0, 2) * v.z, a(1, 0) * v.x + a(1, 1) * v.y + a(1, 2) * v.z, a(2, 0) * v.x + a(2, 1) * v.y + a(2, 2) * v.z}; }
static inline M3 T(const M3& a) { M3 r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r(i, j) = a(j, i); return r; }
static inline M3 outer(const V3& a, const V3& b) { M3 r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r(i, j) = a[i] * b[j]; return r; }
static inline M3 hat(const V3& v) { M3 r; r(0, 1) = -v.z; r(0, 2) = v.y; r(1, 0) = v.z; r(1, 2) = -v.x; r(2, 0) = -v.y; r(2, 1) = v.x; return r; }
static inline M3 I3() { M3 r; r(0, 0) = r(1, 1) = r(2, 2) = 1; return r; }

void PointCluster::push(const V3& vec) { N++; P = P + outer(vec, vec); v = v + vec; }
M3 PointCluster::cov() const { const V3 c = (1.0 / N) * v; return (1.0 / N) * P - outer(c, c); }
PointCluster& PointCluster::operator+=(const PointCluster& o) { P = P + o.P; v = v + o.v; N += o.N; return *this; }
void PointCluster::transform(const PointCluster& s, const IMUST& st) {
    N = s.N;
    v = st.R * s.v + (double)N * st.p;
    const M3 rp = outer(st.R * s.v, st.p);
    P = st.R * s.P * T(st.R) + rp + T(rp) + (double)N * outer(st.p, st.p);
}

void eig3(const M3& Ain, double lambda[3], M3& U) {
    double A[3][3], V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) A[i][j] = 0.5 * (Ain(i, j) + Ain(j, i));
    for (int sweep = 0; sweep < 60; ++sweep) {
        const double off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
        const double diag = A[0][0] * A[0][0] + A[1][1] * A[1][1] + A[2][2] * A[2][2];
        if (off <= 1e-32 * diag || off == 0.0) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (A[p][q] == 0.0) continue;
                const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; ++k) { const double akp = A[k][p], akq = A[k][q]; A[k][p] = c * akp - s * akq; A[k][q] = s * akp + c * akq; }
                for (int k = 0; k < 3; ++k) { const double apk = A[p][k], aqk = A[q][k]; A[p][k] = c * apk - s * aqk; A[q][k] = s * apk + c * aqk; }
                for (int k = 0; k < 3; ++k) { const double vkp = V[k][p], vkq = V[k][q]; V[k][p] = c * vkp - s * vkq; V[k][q] = s * vkp + c * vkq; }
            }
    }
    int idx[3] = {0, 1, 2};
    for (int a = 0; a < 2; ++a) for (int b = a + 1; b < 3; ++b) if (A[idx[b]][idx[b]] < A[idx[a]][idx[a]]) std::swap(idx[a], idx[b]);
    for (int k = 0; k < 3; ++k) { lambda[k] = A[idx[k]][idx[k]]; for (int r = 0; r < 3; ++r) U(r, k) = V[r][idx[k]]; }
}

// ---- Sophus::SE3f pieces used by LidarRes.cc (float arithmetic) ------------------------------------------------------
static void quat_rot_f(const float q[4], const float v[3], float o[3]) {
    float uv[3] = {q[1] * v[2] - q[2] * v[1], q[2] * v[0] - q[0] * v[2], q[0] * v[1] - q[1] * v[0]};
    uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
    o[0] = v[0] + q[3] * uv[0] + (q[1] * uv[2] - q[2] * uv[1]);
    o[1] = v[1] + q[3] * uv[1] + (q[2] * uv[0] - q[0] * uv[2]);
    o[2] = v[2] + q[3] * uv[2] + (q[0] * uv[1] - q[1] * uv[0]);
}
static void quat_mul_f(const float a[4], const float b[4], float o[4]) {
    o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
    o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    o[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
    o[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
}
static void quat_to_mat_f(const float q[4], float R[9]) {
    const float tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
    const float twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
    const float txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
    const float tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
static void mat_to_quat_f(const float R[9], float q[4]) {
    float t = R[0] + R[4] + R[8];
    if (t > 0) {
        t = std::sqrt(t + 1.0f);
        q[3] = 0.5f * t;
        t = 0.5f / t;
        q[0] = (R[7] - R[5]) * t; q[1] = (R[2] - R[6]) * t; q[2] = (R[3] - R[1]) * t;
    } else {
        int i = 0;
        if (R[4] > R[0]) i = 1;
        if (R[8] > R[4 * i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(R[4 * i] - R[4 * j] - R[4 * k] + 1.0f);
        q[i] = 0.5f * t;
        t = 0.5f / t;
        q[3] = (R[3 * k + j] - R[3 * j + k]) * t;
        q[j] = (R[3 * j + i] + R[3 * i + j]) * t;
        q[k] = (R[3 * k + i] + R[3 * i + k]) * t;
    }
    const float n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int c = 0; c < 4; ++c) q[c] /= n;
}
// Twl = Tcw.inverse() * Tcl as rotation matrix / translation in double (cast from the float result)
static IMUST twl_from(const SE3fQ& Tcw, const SE3fQ& Tcl) {
    const float qi[4] = {-Tcw.q[0], -Tcw.q[1], -Tcw.q[2], Tcw.q[3]};
    const float nt[3] = {Tcw.t[0] * -1.f, Tcw.t[1] * -1.f, Tcw.t[2] * -1.f};
    float ti[3], q[4], rt[3], R[9];
    quat_rot_f(qi, nt, ti);          // inverse translation
    quat_mul_f(qi, Tcl.q, q);        // rotation of the product
    quat_rot_f(qi, Tcl.t, rt);
    IMUST s;
    quat_to_mat_f(q, R);
    for (int i = 0; i < 9; ++i) s.R.m[i] = (double)R[i];
    for (int i = 0; i < 3; ++i) s.p[i] = (double)(ti[i] + rt[i]);
    return s;
}
// Sophus::SO3f(R).log()
static V3 so3f_log(const M3& Rd) {
    float R[9], q[4];
    for (int i = 0; i < 9; ++i) R[i] = (float)Rd.m[i];
    mat_to_quat_f(R, q);
    const float sq = q[0] * q[0] + q[1] * q[1] + q[2] * q[2], w = q[3];
    float two_atan;
    const float eps = 1e-10f;
    if (sq < eps * eps) {
        two_atan = 2.0f / w - (2.0f / 3.0f) * sq / (w * w * w);
    } else {
        const float n = std::sqrt(sq);
        if (std::fabs(w) < eps) two_atan = (w > 0 ? 3.14159265358979323846f : -3.14159265358979323846f) / n;
        else two_atan = 2.0f * std::atan(n / w) / n;
    }
    return {(double)(two_atan * q[0]), (double)(two_atan * q[1]), (double)(two_atan * q[2])};
}
static M3 InverseRightJacobianSO3(const V3& v) {  // G2oTypes.cc:828-839
    const double d2 = v.x * v.x + v.y * v.y + v.z * v.z, d = std::sqrt(d2);
    const M3 W = hat(v);
    if (d < 1e-5) return I3();
    return I3() + 0.5 * W + (1.0 / d2 - (1.0 + std::cos(d)) / (2.0 * d * std::sin(d))) * (W * W);
}

// ---- octree ---------------------------------------------------------------------------------------------------------
static const int layer_limit = 2, min_ps = 15;
static const int layer_size[4] = {30, 30, 30, 30};
static const float eigen_value_array[4] = {1.0f / 36, 1.0f / 25, 1.0f / 25, 1.0f / 25};
static const double voxel_size = 1;

struct LidarCovisRes::Node {
    int octo_state = 0, push_state = 0, layer = 0;
    std::vector<std::vector<V3>> vec_orig, vec_tran;
    std::vector<PointCluster> sig_orig, sig_tran;
    std::shared_ptr<Node> leaves[8];
    float voxel_center[3] = {0, 0, 0}, quater_length = 0;
    double decision = 0;
    explicit Node(int win) : vec_orig(win), vec_tran(win), sig_orig(win), sig_tran(win) {}

    bool judge_eigen(int win_count) {  // bavoxel.h:492-536 (the child statistics there do not influence the result)
        PointCluster covMat;
        for (int i = 0; i < win_count; i++) covMat += sig_tran[i];
        double ev[3];
        M3 U;
        eig3(covMat.cov(), ev, U);
        decision = ev[0] / ev[1];
        return decision < eigen_value_array[layer];
    }
    void cut_func(int ci, int win_size) {  // :538-570
        auto& po = vec_orig[ci];
        auto& pt = vec_tran[ci];
        for (size_t j = 0; j < pt.size(); j++) {
            int xyz[3] = {0, 0, 0};
            for (int k = 0; k < 3; k++) if (pt[j][k] > voxel_center[k]) xyz[k] = 1;
            const int leafnum = 4 * xyz[0] + 2 * xyz[1] + xyz[2];
            if (!leaves[leafnum]) {
                leaves[leafnum] = std::make_shared<Node>(win_size);
                for (int k = 0; k < 3; ++k) leaves[leafnum]->voxel_center[k] = voxel_center[k] + (2 * xyz[k] - 1) * quater_length;
                leaves[leafnum]->quater_length = quater_length / 2;
                leaves[leafnum]->layer = layer + 1;
            }
            leaves[leafnum]->vec_orig[ci].push_back(po[j]);
            leaves[leafnum]->vec_tran[ci].push_back(pt[j]);
            if (leaves[leafnum]->octo_state != 1) {
                leaves[leafnum]->sig_orig[ci].push(po[j]);
                leaves[leafnum]->sig_tran[ci].push(pt[j]);
            }
        }
        po.clear(); pt.clear();
    }
    void recut(int win_count, int win_size) {  // :572-602
        if (octo_state != 1) {
            int point_size = 0;
            for (int i = 0; i < win_count; i++) point_size += sig_orig[i].N;
            push_state = 0;
            if (point_size <= min_ps) return;
            if (judge_eigen(win_count)) {
                if (octo_state == 0 && point_size > layer_size[layer]) octo_state = 2;
                if (point_size > min_ps) push_state = 1;
                return;
            } else if (layer == layer_limit) {
                octo_state = 2;
                return;
            }
            octo_state = 1;
            sig_orig.clear(); sig_tran.clear();
            for (int i = 0; i < win_count; i++) cut_func(i, win_size);
        } else {
            cut_func(win_count - 1, win_size);
        }
        for (auto& l : leaves) if (l) l->recut(win_count, win_size);
    }
    void tras_opt(std::vector<PlaneVoxel>& out, int win_count, int win_size) {  // :723-740 with VOX_HESS::push_voxel :57-78
        if (octo_state != 1) {
            int points_size = 0;
            for (int i = 0; i < win_count; i++) points_size += sig_orig[i].N;
            if (points_size < min_ps) return;
            if (push_state == 1) {
                int process_size = 0;
                for (int i = 0; i < win_size; i++) if (sig_orig[i].N != 0) process_size++;
                if (process_size < 2) return;
                PlaneVoxel pv;
                pv.sig_orig = sig_orig;
                for (int j = 0; j < win_size; j++) pv.coe += sig_orig[j].N;
                out.push_back(pv);
            }
        } else {
            for (auto& l : leaves) if (l) l->tras_opt(out, win_count, win_size);
        }
    }
};

size_t LidarCovisRes::LocHash::operator()(const std::array<int64_t, 3>& s) const {  // tools.h:66-79
    const size_t P = 116101, MAXN = 10000000000ull;
    return (((std::hash<int64_t>()(s[2]) * P) % MAXN + std::hash<int64_t>()(s[1])) * P) % MAXN + std::hash<int64_t>()(s[0]);
}

void LidarCovisRes::AddFromKeyFrame(const SE3fQ& Tcw, const std::vector<float>& cloud) {  // LidarRes.cc:32-63 + bavoxel.cc:42-91
    if (cloud.empty()) return;
    IMUST curr = twl_from(Tcw, mTcl);
    if (mPoseBuf.empty()) mPose0 = curr;
    const M3 R0t = T(mPose0.R);
    curr.p = R0t * (curr.p - mPose0.p);
    curr.R = R0t * curr.R;
    mPoseBuf.push_back(curr);
    const int fnum = mCurrPosId;
    for (size_t k = 0; k + 2 < cloud.size(); k += 3) {
        const V3 po{cloud[k], cloud[k + 1], cloud[k + 2]};
        const V3 pt = curr.R * po + curr.p;
        float loc[3];
        for (int j = 0; j < 3; j++) { loc[j] = (float)(pt[j] / voxel_size); if (loc[j] < 0) loc[j] -= 1.0f; }
        const std::array<int64_t, 3> key{(int64_t)loc[0], (int64_t)loc[1], (int64_t)loc[2]};
        auto it = mSurfMap.find(key);
        if (it != mSurfMap.end()) {
            Node& n = *it->second;
            if (n.octo_state != 2) { n.vec_orig[fnum].push_back(po); n.vec_tran[fnum].push_back(pt); }
            if (n.octo_state != 1) { n.sig_orig[fnum].push(po); n.sig_tran[fnum].push(pt); }
        } else {
            auto n = std::make_shared<Node>(win_size_);
            n->vec_orig[fnum].push_back(po); n->vec_tran[fnum].push_back(pt);
            n->sig_orig[fnum].push(po); n->sig_tran[fnum].push(pt);
            for (int j = 0; j < 3; ++j) n->voxel_center[j] = (float)((0.5 + key[j]) * voxel_size);
            n->quater_length = (float)(voxel_size / 4.0);
            mSurfMap[key] = n;
        }
    }
    mCurrPosId++;
}

void LidarCovisRes::BuildVoxHess() {  // LidarRes.cc:64-80
    for (auto& kv : mSurfMap) {
        kv.second->recut(mCurrPosId, win_size_);
        kv.second->tras_opt(mVoxHess, mCurrPosId, win_size_);
    }
}

SE3fQ se3f_from_rt(const double R[9], const double t[3]) {
    float Rf[9];
    SE3fQ T;
    for (int k = 0; k < 9; ++k) Rf[k] = (float)R[k];
    mat_to_quat_f(Rf, T.q);
    for (int k = 0; k < 3; ++k) T.t[k] = (float)t[k];
    return T;
}

void LidarCovisRes::UpdatePose(int i, const double Rcw[9], const double tcw[3]) {  // LidarRes.cc:221-235
    if (i >= win_size_) return;
    mPoseBuf[i] = twl_from(se3f_from_rt(Rcw, tcw), mTcl);
}

double LidarCovisRes::ComputeError() const {  // evaluate_only_residual, bavoxel.h:276-315
    double residual = 0;
    std::vector<PointCluster> sig_tran(win_size_);
    for (const PlaneVoxel& pv : mVoxHess) {
        PointCluster sig;
        for (int i = 0; i < win_size_; i++) { sig_tran[i].transform(pv.sig_orig[i], mPoseBuf[i]); sig += sig_tran[i]; }
        const V3 vBar = (1.0 / sig.N) * sig.v;
        const M3 cmt = (1.0 / sig.N) * sig.P - outer(vBar, vBar);
        double ev[3];
        M3 U;
        eig3(cmt, ev, U);
        residual += pv.coe * ev[0];
    }
    return residual;
}

static inline void add_block(std::vector<double>& H, int n, int r0, int c0, const double* B /*6x6*/, double s) {
    for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) H[(size_t)(r0 + r) * n + c0 + c] += s * B[6 * r + c];
}

void LidarCovisRes::acc_evaluate2(int head, int end, std::vector<double>& Hess, std::vector<double>& JacT, double& residual) const {
    const int W = win_size_, n = 6 * W;
    residual = 0;
    std::vector<PointCluster> sig_tran(W);
    std::vector<V3> viRiTuk(W);
    std::vector<M3> viRiTukukT(W);
    std::vector<std::array<double, 18>> Auk(W);  // 3 x 6 row-major
    for (int a = head; a < end; a++) {
        const std::vector<PointCluster>& sig_orig = mVoxHess[a].sig_orig;
        const double coe = mVoxHess[a].coe;
        PointCluster sig;
        for (int i = 0; i < W; i++) if (sig_orig[i].N != 0) { sig_tran[i].transform(sig_orig[i], mPoseBuf[i]); sig += sig_tran[i]; }
        const V3 vBar = (1.0 / sig.N) * sig.v;
        double lmbd[3];
        M3 U;
        eig3((1.0 / sig.N) * sig.P - outer(vBar, vBar), lmbd, U);
        const int NN = sig.N;
        const V3 u[3] = {{U(0, 0), U(1, 0), U(2, 0)}, {U(0, 1), U(1, 1), U(2, 1)}, {U(0, 2), U(1, 2), U(2, 2)}};
        const V3& uk = u[0];
        const M3 ukukT = outer(uk, uk);
        M3 umumT;
        for (int i = 1; i < 3; i++) umumT = umumT + (2.0 / (lmbd[0] - lmbd[i])) * outer(u[i], u[i]);
        auto AtMB = [&](const std::array<double, 18>& A, const std::array<double, 18>& B, double out[36]) {  // A^T umumT B
            double MB[18];
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 6; ++c) MB[6 * r + c] = umumT(r, 0) * B[c] + umumT(r, 1) * B[6 + c] + umumT(r, 2) * B[12 + c];
            for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) out[6 * r + c] = A[r] * MB[c] + A[6 + r] * MB[6 + c] + A[12 + r] * MB[12 + c];
        };
        auto add33 = [](double Hb[36], int r0, int c0, const M3& m, double s) { for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Hb[6 * (r0 + r) + c0 + c] += s * m(r, c); };
        for (int i = 0; i < W; i++) {
            if (sig_orig[i].N == 0) continue;
            const M3& Pi = sig_orig[i].P;
            const V3& vi = sig_orig[i].v;
            const M3& Ri = mPoseBuf[i].R;
            const double ni = sig_orig[i].N;
            const M3 vihat = hat(vi);
            const V3 RiTuk = T(Ri) * uk;
            const M3 RiTukhat = hat(RiTuk);
            const V3 PiRiTuk = Pi * RiTuk;
            viRiTuk[i] = vihat * RiTuk;
            viRiTukukT[i] = outer(viRiTuk[i], uk);
            const V3 ti_v = mPoseBuf[i].p - vBar;
            const double ukTti_v = dot(uk, ti_v);
            const M3 combo1 = hat(PiRiTuk) + ukTti_v * vihat;
            const V3 combo2 = Ri * vi + ni * ti_v;
            const M3 left = (Ri * Pi + outer(ti_v, vi)) * RiTukhat - Ri * combo1;
            const M3 right = outer(combo2, uk) + dot(combo2, uk) * I3();
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { Auk[i][6 * r + c] = left(r, c) / NN; Auk[i][6 * r + 3 + c] = right(r, c) / NN; }
            double jjt[6];
            for (int c = 0; c < 6; ++c) jjt[c] = Auk[i][c] * uk.x + Auk[i][6 + c] * uk.y + Auk[i][12 + c] * uk.z;
            for (int c = 0; c < 6; ++c) JacT[6 * i + c] += coe * jjt[c];
            const M3 HRt = (2.0 / NN * (1.0 - ni / NN)) * viRiTukukT[i];
            double Hb[36];
            AtMB(Auk[i], Auk[i], Hb);
            add33(Hb, 0, 0, (combo1 - RiTukhat * Pi) * RiTukhat, 2.0 / NN);
            add33(Hb, 0, 0, outer(viRiTuk[i], viRiTuk[i]), -2.0 / NN / NN);
            add33(Hb, 0, 0, hat(V3{jjt[0], jjt[1], jjt[2]}), -0.5);
            add33(Hb, 0, 3, HRt, 1.0);
            add33(Hb, 3, 0, T(HRt), 1.0);
            add33(Hb, 3, 3, ukukT, 2.0 / NN * (ni - ni * ni / NN));
            add_block(Hess, n, 6 * i, 6 * i, Hb, coe);
        }
        for (int i = 0; i < W - 1; i++) {
            if (sig_orig[i].N == 0) continue;
            const double ni = sig_orig[i].N;
            for (int j = i + 1; j < W; j++) {
                if (sig_orig[j].N == 0) continue;
                const double nj = sig_orig[j].N;
                double Hb[36];
                AtMB(Auk[i], Auk[j], Hb);
                add33(Hb, 0, 0, outer(viRiTuk[i], viRiTuk[j]), -2.0 / NN / NN);
                add33(Hb, 0, 3, viRiTukukT[i], -2.0 * nj / NN / NN);
                add33(Hb, 3, 0, T(viRiTukukT[j]), -2.0 * ni / NN / NN);
                add33(Hb, 3, 3, ukukT, -2.0 * ni * nj / NN / NN);
                add_block(Hess, n, 6 * i, 6 * j, Hb, coe);
            }
        }
        residual += coe * lmbd[0];
    }
    for (int i = 1; i < W; i++)
        for (int j = 0; j < i; j++)
            for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) Hess[(size_t)(6 * i + r) * n + 6 * j + c] = Hess[(size_t)(6 * j + c) * n + 6 * i + r];
}

double LidarCovisRes::divide_thread(std::vector<double>& Hess, std::vector<double>& JacT) const {  // bavoxel.h:778-817
    const int n = 6 * win_size_, g_size = (int)mVoxHess.size();
    Hess.assign((size_t)n * n, 0.0);
    JacT.assign(n, 0.0);
    int tthd = 4;
    if (g_size < tthd) tthd = 1;
    const double part = 1.0 * g_size / tthd;
    double residual = 0;
    for (int i = 0; i < tthd; i++) {
        std::vector<double> h((size_t)n * n, 0.0), j(n, 0.0);
        double r = 0;
        acc_evaluate2((int)(part * i), (int)(part * (i + 1)), h, j, r);
        for (size_t k = 0; k < h.size(); ++k) Hess[k] += h[k];
        for (int k = 0; k < n; ++k) JacT[k] += j[k];
        residual += r;
    }
    return residual;
}

void LidarCovisRes::ComputeJandHSE3(std::vector<double>& JacT, std::vector<double>& Hess) const {  // LidarRes.cc:136-186
    const int W = win_size_, n = 6 * W;
    divide_thread(Hess, JacT);
    // mTlc = mTcl.inverse(): rotation / translation in double (cast from float)
    SE3fQ I;
    const IMUST Tlc_st = [&] {  // Tcl^-1 = (Tcl.inverse() * identity)
        const float qi[4] = {-mTcl.q[0], -mTcl.q[1], -mTcl.q[2], mTcl.q[3]};
        const float nt[3] = {mTcl.t[0] * -1.f, mTcl.t[1] * -1.f, mTcl.t[2] * -1.f};
        float ti[3], R[9];
        quat_rot_f(qi, nt, ti);
        quat_to_mat_f(qi, R);
        IMUST s;
        for (int i = 0; i < 9; ++i) s.R.m[i] = (double)R[i];
        for (int i = 0; i < 3; ++i) s.p[i] = (double)ti[i];
        return s;
    }();
    (void)I;
    const M3 Rlc = Tlc_st.R;
    const V3 tlc = Tlc_st.p;
    const V3 tcl{(double)mTcl.t[0], (double)mTcl.t[1], (double)mTcl.t[2]};
    for (int i = 0; i < W; ++i) {
        const M3 Rwci = mPoseBuf[i].R * Rlc;
        const M3 Rcwi = T(Rwci);
        const V3 twci = mPoseBuf[i].R * tlc + mPoseBuf[i].p;
        const V3 tcwi = -1.0 * (T(Rwci) * twci);
        const V3 JacwT{JacT[6 * i], JacT[6 * i + 1], JacT[6 * i + 2]}, JactT{JacT[6 * i + 3], JacT[6 * i + 4], JacT[6 * i + 5]};
        const V3 rwl = so3f_log(mPoseBuf[i].R);
        const M3 inverseJr_Rlc_T = T(InverseRightJacobianSO3(rwl) * Rlc);
        const M3 Rwc_tcl_tcw_T = T(Rwci * hat(tcl - tcwi));
        const V3 JacwT_ = -1.0 * (inverseJr_Rlc_T * JacwT) + Rwc_tcl_tcw_T * JactT;
        const V3 JactT_ = -1.0 * (Rcwi * JactT);
        const V3 jw = Rcwi * JacwT_ - T(hat(tcwi)) * JactT_;
        for (int c = 0; c < 3; ++c) { JacT[6 * i + c] = jw[c]; JacT[6 * i + 3 + c] = JactT_[c]; }
        double DiT[36] = {0}, Di[36];
        auto set33 = [&](int r0, int c0, const M3& m, double s) { for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) DiT[6 * (r0 + r) + c0 + c] += s * m(r, c); };
        set33(0, 0, Rcwi * inverseJr_Rlc_T, -1.0);
        set33(0, 3, Rcwi * Rwc_tcl_tcw_T, 1.0);
        set33(0, 3, T(hat(tcwi)) * Rcwi, 1.0);
        set33(3, 3, Rcwi, -1.0);
        for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) Di[6 * r + c] = DiT[6 * c + r];
        for (int j = 0; j < W; ++j) {
            double Hij[36], out[36];
            for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) Hij[6 * r + c] = Hess[(size_t)(6 * i + r) * n + 6 * j + c];
            for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) { double s = 0; for (int k = 0; k < 6; ++k) s += DiT[6 * r + k] * Hij[6 * k + c]; out[6 * r + c] = s; }
            for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) Hess[(size_t)(6 * i + r) * n + 6 * j + c] = out[6 * r + c];
            double Hji[36];
            for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) Hji[6 * r + c] = Hess[(size_t)(6 * j + r) * n + 6 * i + c];
            for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) { double s = 0; for (int k = 0; k < 6; ++k) s += Hji[6 * r + k] * Di[6 * k + c]; out[6 * r + c] = s; }
            for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) Hess[(size_t)(6 * j + r) * n + 6 * i + c] = out[6 * r + c];
        }
    }
}

void LidarCovisRes::ComputeJandH(std::vector<double>& JacT, std::vector<double>& Hess) const {  // LidarRes.cc:89-128
    const int W = win_size_, n = 6 * W;
    divide_thread(Hess, JacT);
    // mTlb = mTbl.inverse() (Sophus::SE3f), its rotation cast to double; tbl = mTbl.translation()
    const float qi[4] = {-mTbl.q[0], -mTbl.q[1], -mTbl.q[2], mTbl.q[3]};
    float Rf[9];
    quat_to_mat_f(qi, Rf);
    M3 Rlb;
    for (int i = 0; i < 9; ++i) Rlb.m[i] = (double)Rf[i];
    const V3 tbl{(double)mTbl.t[0], (double)mTbl.t[1], (double)mTbl.t[2]};
    for (int i = 0; i < W; ++i) {
        const M3 Rwbi = mPoseBuf[i].R * Rlb;
        const V3 JacwT{JacT[6 * i], JacT[6 * i + 1], JacT[6 * i + 2]}, JactT{JacT[6 * i + 3], JacT[6 * i + 4], JacT[6 * i + 5]};
        const V3 rwl = so3f_log(mPoseBuf[i].R);
        const M3 inverseJr_Rlb_T = T(InverseRightJacobianSO3(rwl) * Rlb);
        const M3 Rwb_tbl_skew_T = T(Rwbi * hat(tbl));
        const V3 jw = inverseJr_Rlb_T * JacwT - Rwb_tbl_skew_T * JactT;
        const V3 jt = T(Rwbi) * JactT;
        for (int c = 0; c < 3; ++c) { JacT[6 * i + c] = jw[c]; JacT[6 * i + 3 + c] = jt[c]; }
        double DiT[36] = {0}, Di[36];
        auto set33 = [&](int r0, int c0, const M3& m, double s) { for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) DiT[6 * (r0 + r) + c0 + c] += s * m(r, c); };
        set33(0, 0, inverseJr_Rlb_T, 1.0);
        set33(0, 3, Rwb_tbl_skew_T, -1.0);
        set33(3, 3, T(Rwbi), 1.0);
        for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) Di[6 * r + c] = DiT[6 * c + r];
        for (int j = 0; j < W; ++j) {
            double Hij[36], out[36];
            for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) Hij[6 * r + c] = Hess[(size_t)(6 * i + r) * n + 6 * j + c];
            for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) { double s = 0; for (int k = 0; k < 6; ++k) s += DiT[6 * r + k] * Hij[6 * k + c]; out[6 * r + c] = s; }
            for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) Hess[(size_t)(6 * i + r) * n + 6 * j + c] = out[6 * r + c];
            double Hji[36];
            for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) Hji[6 * r + c] = Hess[(size_t)(6 * j + r) * n + 6 * i + c];
            for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) { double s = 0; for (int k = 0; k < 6; ++k) s += Hji[6 * r + k] * Di[6 * k + c]; out[6 * r + c] = s; }
            for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) Hess[(size_t)(6 * j + r) * n + 6 * i + c] = out[6 * r + c];
        }
    }
}

// ---- EdgeLidarSE3 / EdgeLidar -----------------------------------------------------------------------------------------
void EdgeLidar::computeError(const double* R, const double* t, int W) {  // G2oTypesWithLidar.h:124-140
    for (int i = 0; i < W; ++i) lio->UpdatePose(i, R + 9 * i, t + 3 * i);
    const double r = lio->ComputeError();
    error = body ? std::sqrt(r) : r;  // G2oTypesWithLidar.cc:41-42 / G2oTypesWithLidar.h:131
    r1 = r2;
    r2 = r;
    is_calc_hess = !(r1 - r2 < 0);
}

void EdgeLidar::linearizeOplus(const double* R, const double* t, int W) {  // :148-166
    for (int i = 0; i < W; ++i) lio->UpdatePose(i, R + 9 * i, t + 3 * i);
    if (is_calc_hess) { if (body) lio->ComputeJandH(JacT, Hessian); else lio->ComputeJandHSE3(JacT, Hessian); }
}

}  // namespace oracle
