// TEST INFRASTRUCTURE ONLY -- CPU oracle, never linked into or called from the product path.  See inertial_ba.hpp.
#include "inertial_ba.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>

namespace oracle {

namespace {

// ---- 3x3 algebra (row-major) ---------------------------------------------------------------------------------------------
void mul(const double* a, const double* b, double* o) { for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) o[3 * r + c] = a[3 * r] * b[c] + a[3 * r + 1] * b[3 + c] + a[3 * r + 2] * b[6 + c]; }
void mulv(const double* a, const double* v, double* o) { for (int r = 0; r < 3; ++r) o[r] = a[3 * r] * v[0] + a[3 * r + 1] * v[1] + a[3 * r + 2] * v[2]; }
void tr(const double* a, double* o) { for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) o[3 * r + c] = a[3 * c + r]; }
void hat(const double* v, double* o) { o[0] = 0; o[1] = -v[2]; o[2] = v[1]; o[3] = v[2]; o[4] = 0; o[5] = -v[0]; o[6] = -v[1]; o[7] = v[0]; o[8] = 0; }
bool inv3(const double* a, double* o) {
    const double c0 = a[4] * a[8] - a[5] * a[7], c1 = a[5] * a[6] - a[3] * a[8], c2 = a[3] * a[7] - a[4] * a[6];
    const double det = a[0] * c0 + a[1] * c1 + a[2] * c2;
    if (det == 0) return false;
    const double id = 1.0 / det;
    o[0] = c0 * id; o[1] = (a[2] * a[7] - a[1] * a[8]) * id; o[2] = (a[1] * a[5] - a[2] * a[4]) * id;
    o[3] = c1 * id; o[4] = (a[0] * a[8] - a[2] * a[6]) * id; o[5] = (a[2] * a[3] - a[0] * a[5]) * id;
    o[6] = c2 * id; o[7] = (a[1] * a[6] - a[0] * a[7]) * id; o[8] = (a[0] * a[4] - a[1] * a[3]) * id;
    return true;
}
void normalize_rotation(double* R) {  // U V^T of the SVD = orthogonal polar factor (Newton iteration)
    for (int it = 0; it < 50; ++it) {
        double Xi[9], XiT[9], N[9], diff = 0;
        if (!inv3(R, Xi)) return;
        tr(Xi, XiT);
        for (int k = 0; k < 9; ++k) { N[k] = 0.5 * (R[k] + XiT[k]); diff = std::fmax(diff, std::fabs(N[k] - R[k])); }
        std::memcpy(R, N, sizeof(N));
        if (diff < 1e-16) break;
    }
}
void right_jacobian(const double v[3], double J[9]) {
    const double d2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2], d = std::sqrt(d2);
    double W[9], W2[9];
    hat(v, W); mul(W, W, W2);
    for (int k = 0; k < 9; ++k) J[k] = k % 4 == 0 ? 1.0 : 0.0;
    if (d < 1e-5) return;
    for (int k = 0; k < 9; ++k) J[k] = J[k] - W[k] * (1.0 - std::cos(d)) / d2 + W2[k] * (d - std::sin(d)) / (d2 * d);
}
void inv_right_jacobian(const double v[3], double J[9]) {
    const double d2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2], d = std::sqrt(d2);
    double W[9], W2[9];
    hat(v, W); mul(W, W, W2);
    for (int k = 0; k < 9; ++k) J[k] = k % 4 == 0 ? 1.0 : 0.0;
    if (d < 1e-5) return;
    for (int k = 0; k < 9; ++k) J[k] = J[k] + W[k] / 2 + W2[k] * (1.0 / d2 - (1.0 + std::cos(d)) / (2.0 * d * std::sin(d)));
}

// ---- dense helpers --------------------------------------------------------------------------------------------------------
bool invert(std::vector<double> A, int n, std::vector<double>& inv) {  // Gauss-Jordan with partial pivoting
    inv.assign((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i) inv[(size_t)i * n + i] = 1.0;
    for (int c = 0; c < n; ++c) {
        int p = c;
        for (int r = c + 1; r < n; ++r) if (std::fabs(A[(size_t)r * n + c]) > std::fabs(A[(size_t)p * n + c])) p = r;
        if (A[(size_t)p * n + c] == 0.0) return false;
        if (p != c) for (int k = 0; k < n; ++k) { std::swap(A[(size_t)p * n + k], A[(size_t)c * n + k]); std::swap(inv[(size_t)p * n + k], inv[(size_t)c * n + k]); }
        const double d = 1.0 / A[(size_t)c * n + c];
        for (int k = 0; k < n; ++k) { A[(size_t)c * n + k] *= d; inv[(size_t)c * n + k] *= d; }
        for (int r = 0; r < n; ++r) {
            if (r == c) continue;
            const double f = A[(size_t)r * n + c];
            if (f == 0.0) continue;
            for (int k = 0; k < n; ++k) { A[(size_t)r * n + k] -= f * A[(size_t)c * n + k]; inv[(size_t)r * n + k] -= f * inv[(size_t)c * n + k]; }
        }
    }
    return true;
}
void eig_sym(std::vector<double> A, int n, std::vector<double>& w, std::vector<double>& V) {  // cyclic Jacobi
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
bool ldlt(std::vector<double>& H, int n, const double* b, double* x) {
    std::vector<double> D(n);
    for (int j = 0; j < n; ++j) {
        double d = H[(size_t)j * n + j];
        for (int k = 0; k < j; ++k) d -= H[(size_t)j * n + k] * H[(size_t)j * n + k] * D[k];
        if (!std::isfinite(d) || d == 0.0) return false;
        D[j] = d;
        for (int i = j + 1; i < n; ++i) {
            double s = H[(size_t)i * n + j];
            for (int k = 0; k < j; ++k) s -= H[(size_t)i * n + k] * H[(size_t)j * n + k] * D[k];
            H[(size_t)i * n + j] = s / d;
        }
    }
    for (int i = 0; i < n; ++i) { double s = b[i]; for (int k = 0; k < i; ++k) s -= H[(size_t)i * n + k] * x[k]; x[i] = s; }
    for (int i = 0; i < n; ++i) x[i] /= D[i];
    for (int i = n - 1; i >= 0; --i) { double s = x[i]; for (int k = i + 1; k < n; ++k) s -= H[(size_t)k * n + i] * x[k]; x[i] = s; }
    return true;
}
// LinearSolverEigen (SF/Thirdparty/g2o/g2o/solvers/linear_solver_eigen.h: Eigen::SimplicialLDLT under a fill-reducing ordering) is a SPARSE
// factorisation; its ordering cannot be restated without Eigen's AMD code.  This stand-in keeps the property that matters for the cost -- the
// zeros of the velocity / bias part are not multiplied: the unknowns from `first_imu` on (9 per keyframe, coupled only to their neighbours in
// time and to the poses) are taken first, the dense pose block after them, and every row is factorised from its first non-zero on (a profile
// LDL^T: fill stays inside the profile).  Same solution as ldlt() up to rounding; a fifth to a ninth of its multiplications.
bool ldlt_profile(const std::vector<double>& H, int n, int first_imu, const double* b, double* x) {
    std::vector<int> order(n);
    for (int k = 0; k < n; ++k) order[k] = k < n - first_imu ? first_imu + k : k - (n - first_imu);
    std::vector<double> A((size_t)n * n), D(n), y(n);
    std::vector<int> lo(n);
    for (int r = 0; r < n; ++r) {
        lo[r] = r;
        for (int c = 0; c <= r; ++c) {
            const int orr = order[r], oc = order[c];
            const double v = orr >= oc ? H[(size_t)orr * n + oc] : H[(size_t)oc * n + orr];  // the lower triangle of H, as ldlt() reads it
            A[(size_t)r * n + c] = v;
            if (v != 0.0 && c < lo[r]) lo[r] = c;
        }
    }
    for (int i = 0; i < n; ++i) {
        double* ri = &A[(size_t)i * n];
        for (int j = lo[i]; j < i; ++j) {
            const double* rj = &A[(size_t)j * n];
            double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
            int k = std::max(lo[i], lo[j]);
            for (; k + 4 <= j; k += 4) { s0 += ri[k] * rj[k]; s1 += ri[k + 1] * rj[k + 1]; s2 += ri[k + 2] * rj[k + 2]; s3 += ri[k + 3] * rj[k + 3]; }
            for (; k < j; ++k) s0 += ri[k] * rj[k];
            ri[j] -= (s0 + s1) + (s2 + s3);
        }
        double d = ri[i];
        for (int j = lo[i]; j < i; ++j) { const double l = ri[j] / D[j]; d -= ri[j] * l; ri[j] = l; }
        if (!std::isfinite(d) || d == 0.0) return false;
        D[i] = d;
    }
    for (int i = 0; i < n; ++i) { double s = b[order[i]]; for (int k = lo[i]; k < i; ++k) s -= A[(size_t)i * n + k] * y[k]; y[i] = s; }
    for (int i = 0; i < n; ++i) y[i] /= D[i];
    for (int i = n - 1; i >= 0; --i) for (int k = lo[i]; k < i; ++k) y[k] -= A[(size_t)i * n + k] * y[i];
    for (int i = 0; i < n; ++i) x[order[i]] = y[i];
    return true;
}
struct HuberD {
    double delta; float dsqr;
    explicit HuberD(float d) : delta(d), dsqr((float)((double)d * (double)d)) {}
    void rho(double e, double& r0, double& r1) const {
        if (e <= dsqr) { r0 = e; r1 = 1.0; } else { const double s = std::sqrt(e); r0 = 2 * s * delta - dsqr; r1 = delta / s; }
    }
};

}  // namespace

void ExpSO3(const double w[3], double R[9]) {
    const double d2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], d = std::sqrt(d2);
    double W[9], W2[9];
    hat(w, W); mul(W, W, W2);
    for (int k = 0; k < 9; ++k) {
        const double I = k % 4 == 0 ? 1.0 : 0.0;
        R[k] = d < 1e-5 ? I + W[k] + 0.5 * W2[k] : I + W[k] * std::sin(d) / d + W2[k] * (1.0 - std::cos(d)) / d2;
    }
    normalize_rotation(R);
}
void LogSO3(const double R[9], double w[3]) {
    const double trc = R[0] + R[4] + R[8];
    w[0] = (R[7] - R[5]) / 2; w[1] = (R[2] - R[6]) / 2; w[2] = (R[3] - R[1]) / 2;
    const double costheta = (trc - 1.0) * 0.5f;
    if (costheta > 1 || costheta < -1) return;
    const double theta = std::acos(costheta), s = std::sin(theta);
    if (std::fabs(s) < 1e-5) return;
    for (int k = 0; k < 3; ++k) w[k] = theta * w[k] / s;
}

void imu_pose_update(InertialKeyFrame& kf, int& its, const ImuCalibD& cal, const double u[6]) {  // ImuCamPose::Update
    double Rut[3], E[9], Rn[9];
    mulv(kf.Rwb, u + 3, Rut);
    for (int k = 0; k < 3; ++k) kf.twb[k] += Rut[k];
    ExpSO3(u, E);
    mul(kf.Rwb, E, Rn);
    std::memcpy(kf.Rwb, Rn, sizeof(Rn));
    its++;
    if (its >= 3) { normalize_rotation(kf.Rwb); its = 0; }
    double Rbw[9], tbw[3], t[3];
    tr(kf.Rwb, Rbw);
    mulv(Rbw, kf.twb, t);
    for (int k = 0; k < 3; ++k) tbw[k] = -t[k];
    mul(cal.Rcb, Rbw, kf.Rcw);
    mulv(cal.Rcb, tbw, t);
    for (int k = 0; k < 3; ++k) kf.tcw[k] = t[k] + cal.tcb[k];
}

int inertial_visual_edge(const InertialKeyFrame& kf, const ImuCalibD& cal, const double X[3], const BAEdge& e, const Camera& cam, double err[3],
                         double A[9], double B[18]) {
    double Xc[3], Xb[3];
    mulv(kf.Rcw, X, Xc);
    for (int k = 0; k < 3; ++k) Xc[k] += kf.tcw[k];
    mulv(cal.Rbc, Xc, Xb);
    for (int k = 0; k < 3; ++k) Xb[k] += cal.tbc[k];
    const bool stereo = e.obs[2] >= 0;
    const int dim = stereo ? 3 : 2;
    const double fx = (double)(float)cam.fx, fy = (double)(float)cam.fy, cx = (double)(float)cam.cx, cy = (double)(float)cam.cy;  // mvParameters are floats
    const double u = fx * Xc[0] / Xc[2] + cx, v = fy * Xc[1] / Xc[2] + cy;
    err[0] = e.obs[0] - u; err[1] = e.obs[1] - v; err[2] = 0;
    if (stereo) { const double invZ = 1 / Xc[2]; err[2] = e.obs[2] - (u - cam.bf * invZ); }
    double pj[9] = {fx / Xc[2], 0.0, -fx * Xc[0] / (Xc[2] * Xc[2]), 0.0, fy / Xc[2], -fy * Xc[1] / (Xc[2] * Xc[2]), 0, 0, 0};
    if (stereo) { pj[6] = pj[0]; pj[7] = pj[1]; pj[8] = pj[2] + cam.bf * (1.0 / (Xc[2] * Xc[2])); }
    std::memset(A, 0, 9 * sizeof(double));
    std::memset(B, 0, 18 * sizeof(double));
    for (int r = 0; r < dim; ++r)
        for (int c = 0; c < 3; ++c) A[3 * r + c] = -(pj[3 * r] * kf.Rcw[c] + pj[3 * r + 1] * kf.Rcw[3 + c] + pj[3 * r + 2] * kf.Rcw[6 + c]);
    const double x = Xb[0], y = Xb[1], z = Xb[2];
    const double S[18] = {0.0, z, -y, 1.0, 0.0, 0.0, -z, 0.0, x, 0.0, 1.0, 0.0, y, -x, 0.0, 0.0, 0.0, 1.0};
    double PR[9];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) PR[3 * r + c] = pj[3 * r] * cal.Rcb[c] + pj[3 * r + 1] * cal.Rcb[3 + c] + pj[3 * r + 2] * cal.Rcb[6 + c];
    for (int r = 0; r < dim; ++r)
        for (int c = 0; c < 6; ++c) B[6 * r + c] = PR[3 * r] * S[c] + PR[3 * r + 1] * S[6 + c] + PR[3 * r + 2] * S[12 + c];
    return dim;
}

void inertial_edge(const InertialKeyFrame& k1, const InertialKeyFrame& k2, const Preintegrated& pint, double err[9], double J[9 * 24]) {
    ImuBias b1;
    b1.bax = (float)k1.ba[0]; b1.bay = (float)k1.ba[1]; b1.baz = (float)k1.ba[2];
    b1.bwx = (float)k1.bg[0]; b1.bwy = (float)k1.bg[1]; b1.bwz = (float)k1.bg[2];
    float dRf[9], dVf[3], dPf[3];
    pint.GetDeltaRotation(b1, dRf); pint.GetDeltaVelocity(b1, dVf); pint.GetDeltaPosition(b1, dPf);
    double dR[9], dV[3], dP[3];
    for (int k = 0; k < 9; ++k) dR[k] = dRf[k];
    for (int k = 0; k < 3; ++k) { dV[k] = dVf[k]; dP[k] = dPf[k]; }
    const double dt = pint.dT;
    const double g[3] = {0, 0, -(double)9.81f};
    double Rbw1[9], dRt[9], t1[9], eR[9], er[3];
    tr(k1.Rwb, Rbw1); tr(dR, dRt);
    mul(dRt, Rbw1, t1); mul(t1, k2.Rwb, eR);
    LogSO3(eR, er);
    double dv[3], dp[3], rv[3], rp[3];
    for (int k = 0; k < 3; ++k) { dv[k] = k2.v[k] - k1.v[k] - g[k] * dt; dp[k] = k2.twb[k] - k1.twb[k] - k1.v[k] * dt - g[k] * dt * dt / 2; }
    mulv(Rbw1, dv, rv); mulv(Rbw1, dp, rp);
    for (int k = 0; k < 3; ++k) { err[k] = er[k]; err[3 + k] = rv[k] - dV[k]; err[6 + k] = rp[k] - dP[k]; }
    if (!J) return;
    std::memset(J, 0, 9 * 24 * sizeof(double));
    auto put = [&](int r0, int c0, const double* m, double s) { for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) J[24 * (r0 + r) + c0 + c] = s * m[3 * r + c]; };
    double invJr[9], Rwb2t[9], m1[9], m2[9], hv[9], hp[9], I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    inv_right_jacobian(er, invJr);
    tr(k2.Rwb, Rwb2t);
    mul(invJr, Rwb2t, m1); mul(m1, k1.Rwb, m2);
    put(0, 0, m2, -1.0);                                  // pose 1: rotation
    double dp2[3];
    for (int k = 0; k < 3; ++k) dp2[k] = k2.twb[k] - k1.twb[k] - k1.v[k] * dt - 0.5 * g[k] * dt * dt;
    mulv(Rbw1, dp2, rp);
    hat(rv, hv); hat(rp, hp);
    put(3, 0, hv, 1.0); put(6, 0, hp, 1.0);
    put(6, 3, I, -1.0);                                   // pose 1: translation
    put(3, 6, Rbw1, -1.0); put(6, 6, Rbw1, -dt);          // velocity 1
    // gyro bias 1: -invJr * eR^T * RightJacobian(JRg * dbg) * JRg, -JVg, -JPg
    double JRg[9], JVg[9], JPg[9], JVa[9], JPa[9];
    for (int k = 0; k < 9; ++k) { JRg[k] = pint.JRg[k]; JVg[k] = pint.JVg[k]; JPg[k] = pint.JPg[k]; JVa[k] = pint.JVa[k]; JPa[k] = pint.JPa[k]; }
    const double dbg[3] = {(double)(b1.bwx - pint.b.bwx), (double)(b1.bwy - pint.b.bwy), (double)(b1.bwz - pint.b.bwz)};  // GetDeltaBias is float
    double Jd[3], RJ[9], eRt[9], a1[9], a2[9], a3[9];
    mulv(JRg, dbg, Jd);
    right_jacobian(Jd, RJ);
    tr(eR, eRt);
    mul(invJr, eRt, a1); mul(a1, RJ, a2); mul(a2, JRg, a3);
    put(0, 9, a3, -1.0); put(3, 9, JVg, -1.0); put(6, 9, JPg, -1.0);
    put(3, 12, JVa, -1.0); put(6, 12, JPa, -1.0);         // accelerometer bias 1
    put(0, 15, invJr, 1.0);                               // pose 2: rotation
    double R12[9];
    mul(Rbw1, k2.Rwb, R12);
    put(6, 18, R12, 1.0);                                 // pose 2: translation
    put(3, 21, Rbw1, 1.0);                                // velocity 2
}

InertialBAResult LocalInertialBA(std::vector<InertialKeyFrame>& kfs, const ImuCalibD& cal, std::vector<double>& points,
                                 const std::vector<BAEdge>& edges, const std::vector<InertialLink>& links, const Camera& cam,
                                 int iterations, double lambda_init, EdgeLidar* lidar, const std::vector<int>* lidar_kf) {
    const int K = (int)kfs.size(), P = (int)points.size() / 3, E = (int)edges.size(), Lk = (int)links.size();
    InertialBAResult res;
    res.chi2.assign(E, 0.0); res.depth_pos.assign(E, 0);
    std::vector<int> pose_var(K, -1), imu_var(K, -1), its(K, 0);
    int n_pose = 0, n_imu = 0;
    for (int k = 0; k < K; ++k) if (!kfs[k].fixed) pose_var[k] = n_pose++;
    for (int k = 0; k < K; ++k) if (!kfs[k].fixed && kfs[k].has_imu) imu_var[k] = n_imu++;
    const int np = 6 * n_pose, n = np + 9 * n_imu;
    // edge informations
    std::vector<std::vector<double>> infoI(Lk), infoG(Lk), infoA(Lk);
    for (int l = 0; l < Lk; ++l) {
        const Preintegrated& p = *links[l].pint;
        std::vector<double> C9(81), inv, w, V;
        for (int r = 0; r < 9; ++r) for (int c = 0; c < 9; ++c) C9[9 * r + c] = p.C[15 * r + c];
        invert(C9, 9, inv);
        for (int r = 0; r < 9; ++r) for (int c = r + 1; c < 9; ++c) { const double m = (inv[9 * r + c] + inv[9 * c + r]) / 2; inv[9 * r + c] = inv[9 * c + r] = m; }
        eig_sym(inv, 9, w, V);
        for (double& x : w) if (x < 1e-12) x = 0;
        infoI[l].assign(81, 0.0);
        for (int r = 0; r < 9; ++r) for (int c = 0; c < 9; ++c) { double s = 0; for (int k = 0; k < 9; ++k) s += V[9 * r + k] * w[k] * V[9 * c + k]; infoI[l][9 * r + c] = s * links[l].info_scale; }
        std::vector<double> G(9), A(9);
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { G[3 * r + c] = p.C[15 * (9 + r) + 9 + c]; A[3 * r + c] = p.C[15 * (12 + r) + 12 + c]; }
        invert(G, 3, infoG[l]); invert(A, 3, infoA[l]);
    }
    const HuberD hub_mono((float)std::sqrt(5.991)), hub_stereo((float)std::sqrt(7.815)), hub_imu((float)std::sqrt(16.92));

    std::vector<double> err_v(3 * (size_t)E), err_i(9 * (size_t)Lk);
    std::vector<double> chi_i(Lk), chi_g(Lk), chi_a(Lk);
    const int Wl = lidar && lidar_kf ? (int)lidar_kf->size() : 0;
    std::vector<double> lR(9 * (size_t)Wl), lt(3 * (size_t)Wl);
    auto lidar_vertices = [&]() {  // VPi->estimate().Rcw[0] / tcw[0]
        for (int i = 0; i < Wl; ++i) { std::memcpy(&lR[9 * i], kfs[(*lidar_kf)[i]].Rcw, 72); std::memcpy(&lt[3 * i], kfs[(*lidar_kf)[i]].tcw, 24); }
    };
    auto compute_errors = [&]() {
        if (Wl) { lidar_vertices(); lidar->computeError(lR.data(), lt.data(), Wl); }
        for (int e = 0; e < E; ++e) {
            double A[9], B[18];
            const int dim = inertial_visual_edge(kfs[edges[e].pose], cal, &points[3 * edges[e].point], edges[e], cam, &err_v[3 * e], A, B);
            double s = 0;
            for (int d = 0; d < dim; ++d) s += err_v[3 * e + d] * edges[e].info * err_v[3 * e + d];
            res.chi2[e] = s;
        }
        for (int l = 0; l < Lk; ++l) {
            inertial_edge(kfs[links[l].kf1], kfs[links[l].kf2], *links[l].pint, &err_i[9 * l], nullptr);
            double s = 0;
            for (int r = 0; r < 9; ++r) { double t = 0; for (int c = 0; c < 9; ++c) t += infoI[l][9 * r + c] * err_i[9 * l + c]; s += err_i[9 * l + r] * t; }
            chi_i[l] = s;
            double sg = 0, sa = 0;
            for (int r = 0; r < 3; ++r) {
                double tg = 0, ta = 0;
                for (int c = 0; c < 3; ++c) {
                    tg += infoG[l][3 * r + c] * (kfs[links[l].kf2].bg[c] - kfs[links[l].kf1].bg[c]);
                    ta += infoA[l][3 * r + c] * (kfs[links[l].kf2].ba[c] - kfs[links[l].kf1].ba[c]);
                }
                sg += (kfs[links[l].kf2].bg[r] - kfs[links[l].kf1].bg[r]) * tg;
                sa += (kfs[links[l].kf2].ba[r] - kfs[links[l].kf1].ba[r]) * ta;
            }
            chi_g[l] = sg; chi_a[l] = sa;
        }
    };
    auto robust_chi2 = [&]() {
        double chi = Wl ? lidar->chi2() : 0.0;
        for (int l = 0; l < Lk; ++l) {
            double r0 = chi_i[l], r1;
            if (links[l].robust) hub_imu.rho(chi_i[l], r0, r1);
            chi += r0 + chi_g[l] + chi_a[l];
        }
        for (int e = 0; e < E; ++e) {
            double r0, r1;
            (edges[e].obs[2] >= 0 ? hub_stereo : hub_mono).rho(res.chi2[e], r0, r1);
            chi += r0;
        }
        return chi;
    };
    compute_errors();
    res.err = robust_chi2();

    std::vector<double> H((size_t)n * n), b(n), Hll(9 * (size_t)P), bl(3 * (size_t)P), Hpl(18 * (size_t)E), S, bs(n), x(n), xl(3 * (size_t)P), Dinv(9 * (size_t)P);
    double lambda = lambda_init, ni = 2;
    int n_bad = 0;
    bool ok = true;
    for (int it = 0; it < iterations && ok; ++it) {
        compute_errors();
        double currentChi = robust_chi2(), tempChi = currentChi;
        const double iniChi = currentChi;
        std::fill(H.begin(), H.end(), 0.0); std::fill(b.begin(), b.end(), 0.0); std::fill(Hll.begin(), Hll.end(), 0.0);
        std::fill(bl.begin(), bl.end(), 0.0); std::fill(Hpl.begin(), Hpl.end(), 0.0);
        for (int e = 0; e < E; ++e) {
            const BAEdge& ed = edges[e];
            double er[3], A[9], B[18];
            const int dim = inertial_visual_edge(kfs[ed.pose], cal, &points[3 * ed.point], ed, cam, er, A, B);
            double r0, r1;
            (ed.obs[2] >= 0 ? hub_stereo : hub_mono).rho(res.chi2[e], r0, r1);
            const double w = r1 * ed.info;
            const int pv = pose_var[ed.pose], l = ed.point;
            for (int r = 0; r < 3; ++r) {
                double s = 0;
                for (int d = 0; d < dim; ++d) s += A[3 * d + r] * (-ed.info * er[d] * r1);
                bl[3 * l + r] += s;
                for (int c = 0; c < 3; ++c) { double h = 0; for (int d = 0; d < dim; ++d) h += A[3 * d + r] * w * A[3 * d + c]; Hll[9 * l + 3 * r + c] += h; }
            }
            if (pv >= 0) {
                for (int r = 0; r < 6; ++r) {
                    double s = 0;
                    for (int d = 0; d < dim; ++d) s += B[6 * d + r] * (-ed.info * er[d] * r1);
                    b[6 * pv + r] += s;
                    for (int c = 0; c < 6; ++c) { double h = 0; for (int d = 0; d < dim; ++d) h += B[6 * d + r] * w * B[6 * d + c]; H[(size_t)(6 * pv + r) * n + 6 * pv + c] += h; }
                    for (int c = 0; c < 3; ++c) { double h = 0; for (int d = 0; d < dim; ++d) h += B[6 * d + r] * w * A[3 * d + c]; Hpl[18 * (size_t)e + 3 * r + c] += h; }
                }
            }
        }
        for (int l = 0; l < Lk; ++l) {
            const InertialLink& lk = links[l];
            double er[9], J[9 * 24];
            inertial_edge(kfs[lk.kf1], kfs[lk.kf2], *lk.pint, er, J);
            double r0, r1 = 1.0;
            if (lk.robust) hub_imu.rho(chi_i[l], r0, r1);
            // vertex blocks of the edge: offset in the state vector (-1: fixed), column offset in J, size
            const int off[6] = {pose_var[lk.kf1] >= 0 ? 6 * pose_var[lk.kf1] : -1, imu_var[lk.kf1] >= 0 ? np + 9 * imu_var[lk.kf1] : -1,
                                imu_var[lk.kf1] >= 0 ? np + 9 * imu_var[lk.kf1] + 3 : -1, imu_var[lk.kf1] >= 0 ? np + 9 * imu_var[lk.kf1] + 6 : -1,
                                pose_var[lk.kf2] >= 0 ? 6 * pose_var[lk.kf2] : -1, imu_var[lk.kf2] >= 0 ? np + 9 * imu_var[lk.kf2] : -1};
            const int col[6] = {0, 6, 9, 12, 15, 21}, sz[6] = {6, 3, 3, 3, 6, 3};
            double OJ[9 * 24], Oe[9];  // (w Omega) J and (w Omega) e
            for (int r = 0; r < 9; ++r) {
                double s = 0;
                for (int k = 0; k < 9; ++k) s += r1 * infoI[l][9 * r + k] * er[k];
                Oe[r] = s;
                for (int c = 0; c < 24; ++c) { double t = 0; for (int k = 0; k < 9; ++k) t += r1 * infoI[l][9 * r + k] * J[24 * k + c]; OJ[24 * r + c] = t; }
            }
            for (int a = 0; a < 6; ++a) {
                if (off[a] < 0) continue;
                for (int r = 0; r < sz[a]; ++r) {
                    double s = 0;
                    for (int k = 0; k < 9; ++k) s += J[24 * k + col[a] + r] * Oe[k];
                    b[off[a] + r] -= s;
                    for (int bb = 0; bb < 6; ++bb) {
                        if (off[bb] < 0) continue;
                        for (int c = 0; c < sz[bb]; ++c) { double h = 0; for (int k = 0; k < 9; ++k) h += J[24 * k + col[a] + r] * OJ[24 * k + col[bb] + c]; H[(size_t)(off[a] + r) * n + off[bb] + c] += h; }
                    }
                }
            }
            // random walks of the two biases
            for (int which = 0; which < 2; ++which) {
                const std::vector<double>& Om = which == 0 ? infoG[l] : infoA[l];
                const int o1 = imu_var[lk.kf1] >= 0 ? np + 9 * imu_var[lk.kf1] + 3 + 3 * which : -1, o2 = imu_var[lk.kf2] >= 0 ? np + 9 * imu_var[lk.kf2] + 3 + 3 * which : -1;
                double e3[3], Oe3[3];
                for (int k = 0; k < 3; ++k) e3[k] = which == 0 ? kfs[lk.kf2].bg[k] - kfs[lk.kf1].bg[k] : kfs[lk.kf2].ba[k] - kfs[lk.kf1].ba[k];
                for (int r = 0; r < 3; ++r) Oe3[r] = Om[3 * r] * e3[0] + Om[3 * r + 1] * e3[1] + Om[3 * r + 2] * e3[2];
                for (int r = 0; r < 3; ++r) {
                    if (o1 >= 0) b[o1 + r] += Oe3[r];   // J1 = -I: b += -J1^T Omega e
                    if (o2 >= 0) b[o2 + r] -= Oe3[r];
                    for (int c = 0; c < 3; ++c) {
                        if (o1 >= 0) H[(size_t)(o1 + r) * n + o1 + c] += Om[3 * r + c];
                        if (o2 >= 0) H[(size_t)(o2 + r) * n + o2 + c] += Om[3 * r + c];
                        if (o1 >= 0 && o2 >= 0) { H[(size_t)(o1 + r) * n + o2 + c] -= Om[3 * r + c]; H[(size_t)(o2 + r) * n + o1 + c] -= Om[3 * r + c]; }
                    }
                }
            }
        }
        if (Wl) {
            // EdgeLidar::linearizeOplus + computeQuadraticFormLidarRes (G2oTypesWithLidar.cc:55-140), quirks as in ba.cpp: the
            // 6x6 blocks are read at element offsets (i, i) / (i, j), b takes -info * JacT without the residual
            lidar_vertices();
            lidar->linearizeOplus(lR.data(), lt.data(), Wl);
            const int nl = 6 * Wl;
            const double info = lidar->information;
            for (int i = 0; i < Wl; ++i) {
                const int vi = pose_var[(*lidar_kf)[i]];
                if (vi < 0) continue;
                for (int r = 0; r < 6; ++r) {
                    b[6 * vi + r] -= info * lidar->JacT[6 * i + r];
                    for (int c = 0; c < 6; ++c) H[(size_t)(6 * vi + r) * n + 6 * vi + c] += lidar->Hessian[(size_t)(i + r) * nl + i + c] * info;
                }
                for (int j = i + 1; j < Wl; ++j) {
                    const int vj = pose_var[(*lidar_kf)[j]];
                    if (vj < 0) continue;
                    // the upper block (smaller variable index first) receives Hessian.block(i, j), or block(j, i) when the
                    // helper is transposed; the matrix is kept symmetric here
                    const bool transposed = vi > vj;
                    for (int r = 0; r < 6; ++r)
                        for (int c = 0; c < 6; ++c) {
                            const double h = (transposed ? lidar->Hessian[(size_t)(j + r) * nl + i + c] : lidar->Hessian[(size_t)(i + r) * nl + j + c]) * info;
                            const int ur = transposed ? 6 * vj + r : 6 * vi + r, uc = transposed ? 6 * vi + c : 6 * vj + c;
                            H[(size_t)ur * n + uc] += h;
                            H[(size_t)uc * n + ur] += h;
                        }
                }
            }
        }
        if (it == 0 && !(lambda_init > 0)) {
            double mx = 0;
            for (int i = 0; i < n; ++i) mx = std::max(mx, std::fabs(H[(size_t)i * n + i]));
            for (int l = 0; l < P; ++l) for (int j = 0; j < 3; ++j) mx = std::max(mx, std::fabs(Hll[9 * (size_t)l + 4 * j]));
            lambda = 1e-5 * mx;
        }
        double rho = 0;
        int qmax = 0;
        do {
            const std::vector<InertialKeyFrame> backup_kf = kfs;
            const std::vector<int> backup_its = its;
            const std::vector<double> backup_pts = points;
            S = H;
            for (int i = 0; i < n; ++i) S[(size_t)i * n + i] += lambda;
            std::vector<double> coeff(n, 0.0);
            for (int l = 0; l < P; ++l) {
                double D[9];
                std::memcpy(D, &Hll[9 * (size_t)l], sizeof(D));
                D[0] += lambda; D[4] += lambda; D[8] += lambda;
                inv3(D, &Dinv[9 * (size_t)l]);
            }
            std::vector<std::vector<int>> by_point(P);
            for (int e = 0; e < E; ++e) if (pose_var[edges[e].pose] >= 0) by_point[edges[e].point].push_back(e);
            for (int l = 0; l < P; ++l) {
                const double* Di = &Dinv[9 * (size_t)l];
                double db[3];
                mulv(Di, &bl[3 * l], db);
                for (int e1 : by_point[l]) {
                    const int i1 = pose_var[edges[e1].pose];
                    const double* Bi = &Hpl[18 * (size_t)e1];
                    double BD[18];
                    for (int r = 0; r < 6; ++r) for (int c = 0; c < 3; ++c) BD[3 * r + c] = Bi[3 * r] * Di[c] + Bi[3 * r + 1] * Di[3 + c] + Bi[3 * r + 2] * Di[6 + c];
                    for (int r = 0; r < 6; ++r) coeff[6 * i1 + r] += Bi[3 * r] * db[0] + Bi[3 * r + 1] * db[1] + Bi[3 * r + 2] * db[2];
                    for (int e2 : by_point[l]) {
                        const int i2 = pose_var[edges[e2].pose];
                        const double* Bj = &Hpl[18 * (size_t)e2];
                        for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c)
                            S[(size_t)(6 * i1 + r) * n + 6 * i2 + c] -= BD[3 * r] * Bj[3 * c] + BD[3 * r + 1] * Bj[3 * c + 1] + BD[3 * r + 2] * Bj[3 * c + 2];
                    }
                }
            }
            for (int i = 0; i < n; ++i) bs[i] = b[i] - coeff[i];
            const bool ok2 = n == 0 ? true : (n > np ? ldlt_profile(S, n, np, bs.data(), x.data()) : ldlt(S, n, bs.data(), x.data()));
            if (ok2) {
                std::vector<double> cl(bl);
                for (int e = 0; e < E; ++e) {
                    const int pv = pose_var[edges[e].pose];
                    if (pv < 0) continue;
                    const double* Bi = &Hpl[18 * (size_t)e];
                    for (int c = 0; c < 3; ++c) { double s = 0; for (int r = 0; r < 6; ++r) s += Bi[3 * r + c] * x[6 * pv + r]; cl[3 * edges[e].point + c] -= s; }
                }
                for (int l = 0; l < P; ++l) mulv(&Dinv[9 * (size_t)l], &cl[3 * l], &xl[3 * l]);
            }
            for (int k = 0; k < K; ++k) {
                if (pose_var[k] >= 0) imu_pose_update(kfs[k], its[k], cal, &x[6 * pose_var[k]]);
                if (imu_var[k] >= 0) {
                    const double* u = &x[np + 9 * imu_var[k]];
                    for (int c = 0; c < 3; ++c) { kfs[k].v[c] += u[c]; kfs[k].bg[c] += u[3 + c]; kfs[k].ba[c] += u[6 + c]; }
                }
            }
            for (int i = 0; i < 3 * P; ++i) points[i] += xl[i];
            compute_errors();
            tempChi = robust_chi2();
            if (!ok2) tempChi = std::numeric_limits<double>::max();
            rho = currentChi - tempChi;
            double scale = 0;
            for (int j = 0; j < n; ++j) scale += x[j] * (lambda * x[j] + b[j]);
            for (int j = 0; j < 3 * P; ++j) scale += xl[j] * (lambda * xl[j] + bl[j]);
            scale += 1e-3;
            rho /= scale;
            if (rho > 0 && std::isfinite(tempChi)) {
                double alpha = 1. - std::pow((2 * rho - 1), 3);
                alpha = std::min(alpha, 2. / 3.);
                lambda *= std::max(1. / 3., alpha);
                ni = 2;
                currentChi = tempChi;
            } else {
                lambda *= ni;
                ni *= 2;
                kfs = backup_kf; its = backup_its; points = backup_pts;
            }
            qmax++;
        } while (rho < 0 && qmax < 10);
        ++res.iterations;
        res.trace.chi2.push_back(currentChi); res.trace.lambda.push_back(lambda); res.trace.trials.push_back(qmax);
        if (qmax == 10 || rho == 0) { ok = false; continue; }
        if ((iniChi - currentChi) * 1e3 < iniChi) n_bad++; else n_bad = 0;
        if (n_bad >= 3) ok = false;
    }
    res.err_end = robust_chi2();  // activeRobustChi2 with the errors of the last computeActiveErrors
    for (int e = 0; e < E; ++e) {
        const InertialKeyFrame& kf = kfs[edges[e].pose];
        const double* X = &points[3 * edges[e].point];
        res.depth_pos[e] = (kf.Rcw[6] * X[0] + kf.Rcw[7] * X[1] + kf.Rcw[8] * X[2] + kf.tcw[2]) > 0.0;
    }
    return res;
}


// ---- PoseInertialOptimizationLastKeyFrame / LastFrame -----------------------------------------------------------------------------------
namespace {
// EdgeInertial ctor (G2oTypes.cc:499-515): inverse of C(0:9, 0:9), symmetrised, eigenvalues below 1e-12 cleared
void edge_inertial_information(const Preintegrated& p, std::vector<double>& info) {
    std::vector<double> C9(81), inv, w, V;
    for (int r = 0; r < 9; ++r) for (int c = 0; c < 9; ++c) C9[9 * r + c] = p.C[15 * r + c];
    invert(C9, 9, inv);
    for (int r = 0; r < 9; ++r) for (int c = r + 1; c < 9; ++c) { const double m = (inv[9 * r + c] + inv[9 * c + r]) / 2; inv[9 * r + c] = inv[9 * c + r] = m; }
    eig_sym(inv, 9, w, V);
    for (double& x : w) if (x < 1e-12) x = 0;
    info.assign(81, 0.0);
    for (int r = 0; r < 9; ++r) for (int c = 0; c < 9; ++c) { double s = 0; for (int k = 0; k < 9; ++k) s += V[9 * r + k] * w[k] * V[9 * c + k]; info[9 * r + c] = s; }
}
// ConstraintPoseImu ctor (G2oTypes.h:721-732)
void constraint_clamp(double H[225]) {
    std::vector<double> A(H, H + 225), w, V;
    for (int r = 0; r < 15; ++r) for (int c = 0; c < 15; ++c) A[15 * r + c] = (H[15 * r + c] + H[15 * r + c]) / 2;  // "(H + H) / 2", as written
    eig_sym(A, 15, w, V);
    for (double& x : w) if (x < 1e-12) x = 0;
    for (int r = 0; r < 15; ++r) for (int c = 0; c < 15; ++c) { double s = 0; for (int k = 0; k < 15; ++k) s += V[15 * r + k] * w[k] * V[15 * c + k]; H[15 * r + c] = s; }
}
}  // namespace

PoseInertialResult PoseInertialOptimization(InertialKeyFrame& cur, InertialKeyFrame& other, bool last_frame, const PoseImuPrior* prior_prev,
                                            const ImuCalibD& cal, const Preintegrated& pint, const Preintegrated& pint_rw,
                                            const std::vector<double>& Xw, const std::vector<BAEdge>& edges, const std::vector<uint8_t>& close,
                                            const Camera& cam, bool bRecInit) {
    const int E = (int)edges.size();
    PoseInertialResult res;
    res.outlier.assign(E, 0);
    res.n_initial = E;
    // unknowns: the frame's pose 6, velocity 3, gyro bias 3, accelerometer bias 3 (vertex ids 0..3), then the previous frame's (ids 4..7)
    const int n = last_frame ? 30 : 15;
    std::vector<double> infoI, infoG, infoA;
    edge_inertial_information(pint, infoI);
    {
        std::vector<double> G(9), A(9);
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { G[3 * r + c] = pint_rw.C[15 * (9 + r) + 9 + c]; A[3 * r + c] = pint_rw.C[15 * (12 + r) + 12 + c]; }
        invert(G, 3, infoG); invert(A, 3, infoA);
    }
    const HuberD hub_mono((float)std::sqrt(5.991)), hub_stereo((float)std::sqrt(7.815)), hub_prior(5.0f);
    std::vector<uint8_t> level(E, 0), robust(E, 1);
    std::vector<double> err(3 * (size_t)E), chi2(E, 0.0);
    int its_cur = 0, its_other = 0;
    auto visual_error = [&](int e, double* A, double* B) {
        const int dim = inertial_visual_edge(cur, cal, &Xw[3 * edges[e].point], edges[e], cam, &err[3 * e], A, B);
        double s = 0;
        for (int d = 0; d < dim; ++d) s += err[3 * e + d] * edges[e].info * err[3 * e + d];
        chi2[e] = s;
        return dim;
    };
    // EdgePriorPoseImu (G2oTypes.cc:738-767): error 15, Jacobian 15 x 15 (block diagonal), information = prior H
    auto prior_edge = [&](double e15[15], double J[225]) {
        double Rt[9], dR[9], er[3], dt[3], et[3];
        tr(prior_prev->Rwb, Rt);
        mul(Rt, other.Rwb, dR);
        LogSO3(dR, er);
        for (int k = 0; k < 3; ++k) dt[k] = other.twb[k] - prior_prev->twb[k];
        mulv(Rt, dt, et);
        for (int k = 0; k < 3; ++k) { e15[k] = er[k]; e15[3 + k] = et[k]; e15[6 + k] = other.v[k] - prior_prev->vwb[k]; e15[9 + k] = other.bg[k] - prior_prev->bg[k]; e15[12 + k] = other.ba[k] - prior_prev->ba[k]; }
        if (!J) return;
        std::memset(J, 0, 225 * sizeof(double));
        double iJr[9];
        inv_right_jacobian(er, iJr);
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { J[15 * r + c] = iJr[3 * r + c]; J[15 * (3 + r) + 3 + c] = dR[3 * r + c]; }
        for (int k = 6; k < 15; ++k) J[15 * k + k] = 1.0;
    };
    std::vector<double> H((size_t)n * n), b(n), x(n, 0.0);
    const float chi2Mono_kf[4] = {12, 7.5, 5.991, 5.991}, chi2Mono_f[4] = {5.991, 5.991, 5.991, 5.991}, chi2Stereo[4] = {15.6f, 9.8f, 7.815f, 7.815f};
    const int n_graph_edges = E + 3 + (last_frame ? 1 : 0);
    int nBad = 0, nInliers = 0;
    for (int round = 0; round < 4; ++round) {
        bool ok = true;
        for (int it = 0; it < 10 && ok; ++it) {
            std::fill(H.begin(), H.end(), 0.0);
            std::fill(b.begin(), b.end(), 0.0);
            // visual edges of level 0 -> pose block of the frame
            for (int e = 0; e < E; ++e) {
                if (level[e]) continue;
                double A[9], B[18];
                const int dim = visual_error(e, A, B);
                double r0 = chi2[e], r1 = 1.0;
                if (robust[e]) (dim == 3 ? hub_stereo : hub_mono).rho(chi2[e], r0, r1);
                const double w = edges[e].info;
                for (int i = 0; i < 6; ++i) {
                    double g = 0;
                    for (int d = 0; d < dim; ++d) g += B[6 * d + i] * w * err[3 * e + d];
                    b[i] -= r1 * g;
                    for (int j = 0; j < 6; ++j) {
                        double h = 0;
                        for (int d = 0; d < dim; ++d) h += B[6 * d + i] * (r1 * w) * B[6 * d + j];
                        H[(size_t)i * n + j] += h;
                    }
                }
            }
            // EdgeInertial: vertices (other P, V, G, A | cur P, V); columns of J: P1 6 | V1 3 | G1 3 | A1 3 | P2 6 | V2 3
            {
                double e9[9], J[9 * 24];
                inertial_edge(other, cur, pint, e9, J);
                int col[24];
                for (int k = 0; k < 15; ++k) col[k] = last_frame ? 15 + k : -1;  // the other state: free only in the last-frame form
                for (int k = 0; k < 9; ++k) col[15 + k] = k;                     // the frame's pose and velocity
                double Oe[9];
                for (int r = 0; r < 9; ++r) { double s = 0; for (int c = 0; c < 9; ++c) s += infoI[9 * r + c] * e9[c]; Oe[r] = s; }
                for (int i = 0; i < 24; ++i) {
                    if (col[i] < 0) continue;
                    double g = 0;
                    for (int r = 0; r < 9; ++r) g += J[24 * r + i] * Oe[r];
                    b[col[i]] -= g;
                    for (int j = 0; j < 24; ++j) {
                        if (col[j] < 0) continue;
                        double h = 0;
                        for (int r = 0; r < 9; ++r) { double t = 0; for (int c = 0; c < 9; ++c) t += infoI[9 * r + c] * J[24 * c + j]; h += J[24 * r + i] * t; }
                        H[(size_t)col[i] * n + col[j]] += h;
                    }
                }
            }
            // EdgeGyroRW / EdgeAccRW: error = cur - other, Jacobians -I (other) and +I (cur)
            for (int which = 0; which < 2; ++which) {
                const std::vector<double>& info = which ? infoA : infoG;
                const double* c2 = which ? cur.ba : cur.bg;
                const double* c1 = which ? other.ba : other.bg;
                const int ic = which ? 12 : 9, io = last_frame ? 15 + ic : -1;
                double e3[3], Oe[3];
                for (int k = 0; k < 3; ++k) e3[k] = c2[k] - c1[k];
                for (int r = 0; r < 3; ++r) Oe[r] = info[3 * r] * e3[0] + info[3 * r + 1] * e3[1] + info[3 * r + 2] * e3[2];
                for (int r = 0; r < 3; ++r) {
                    b[ic + r] -= Oe[r];
                    if (io >= 0) b[io + r] += Oe[r];
                    for (int c = 0; c < 3; ++c) {
                        H[(size_t)(ic + r) * n + ic + c] += info[3 * r + c];
                        if (io >= 0) {
                            H[(size_t)(io + r) * n + io + c] += info[3 * r + c];
                            H[(size_t)(ic + r) * n + io + c] -= info[3 * r + c];
                            H[(size_t)(io + r) * n + ic + c] -= info[3 * r + c];
                        }
                    }
                }
            }
            // EdgePriorPoseImu on the previous frame (Huber 5)
            if (last_frame) {
                double e15[15], J[225], Oe[15];
                prior_edge(e15, J);
                double c = 0;
                for (int r = 0; r < 15; ++r) { double s = 0; for (int k = 0; k < 15; ++k) s += prior_prev->H[15 * r + k] * e15[k]; Oe[r] = s; c += e15[r] * s; }
                double r0, r1;
                hub_prior.rho(c, r0, r1);
                for (int i = 0; i < 15; ++i) {
                    double g = 0;
                    for (int r = 0; r < 15; ++r) g += J[15 * r + i] * Oe[r];
                    b[15 + i] -= r1 * g;
                    for (int j = 0; j < 15; ++j) {
                        double h = 0;
                        for (int r = 0; r < 15; ++r) { double t = 0; for (int k = 0; k < 15; ++k) t += prior_prev->H[15 * r + k] * J[15 * k + j]; h += J[15 * r + i] * t; }
                        H[(size_t)(15 + i) * n + 15 + j] += r1 * h;
                    }
                }
            }
            // LinearSolverDense: LDLT, usable only when no pivot is negative (Eigen::LDLT::isPositive)
            std::vector<double> Hw = H;
            std::vector<double> xn(n);
            bool pos = ldlt(Hw, n, b.data(), xn.data());
            if (pos) {  // the sign of D: recompute the pivots (ldlt() leaves the unit lower factor in Hw, D is not kept)
                std::vector<double> D(n);
                for (int j = 0; j < n && pos; ++j) {
                    double d = H[(size_t)j * n + j];
                    for (int k = 0; k < j; ++k) d -= Hw[(size_t)j * n + k] * Hw[(size_t)j * n + k] * D[k];
                    D[j] = d;
                    if (d < 0) pos = false;
                }
            }
            if (pos) x = xn; else { ok = false; res.solver_failed = true; }
            // SparseOptimizer::update: the increment is applied whether or not the solve succeeded (GaussNewton::solve :84-91)
            imu_pose_update(cur, its_cur, cal, &x[0]);
            for (int k = 0; k < 3; ++k) { cur.v[k] += x[6 + k]; cur.bg[k] += x[9 + k]; cur.ba[k] += x[12 + k]; }
            if (last_frame) {
                imu_pose_update(other, its_other, cal, &x[15]);
                for (int k = 0; k < 3; ++k) { other.v[k] += x[21 + k]; other.bg[k] += x[24 + k]; other.ba[k] += x[27 + k]; }
            }
        }
        // Classification.  e->chi2() reads the error stored by the LAST computeActiveErrors(), which Gauss-Newton runs at the start of
        // an iteration: for the edges that were active that is the estimate BEFORE the round's last update; only the edges that were
        // outliers are recomputed, at the final estimate (:2680-2683, :3097-3100).
        nBad = 0; nInliers = 0;
        const float* chiM = last_frame ? chi2Mono_f : chi2Mono_kf;
        const float chi2close = 1.5f * chiM[round];
        for (int e = 0; e < E; ++e) {
            const bool stereo = edges[e].obs[2] >= 0;
            if (res.outlier[e]) { double A[9], B[18]; visual_error(e, A, B); }
            const float c = (float)chi2[e];
            bool bad;
            if (stereo) {
                bad = c > chi2Stereo[round];
            } else {
                double Xc[3];
                mulv(cur.Rcw, &Xw[3 * edges[e].point], Xc);
                const bool depth_pos = Xc[2] + cur.tcw[2] > 0.0;
                bad = (c > chiM[round] && !close[e]) || (close[e] && c > chi2close) || !depth_pos;
            }
            res.outlier[e] = bad;
            level[e] = bad;
            if (bad) ++nBad; else ++nInliers;
            if (round == 2) robust[e] = 0;
        }
        if (n_graph_edges < 10) break;
    }
    // "If not too much tracks, recover not too bad points" (:2738-2765, :3160-3188)
    if (nInliers < 30 && !bRecInit) {
        nBad = 0;
        for (int pass = 0; pass < 2; ++pass)  // monocular edges first, then the stereo ones (the order only matters for nBad's sum)
            for (int e = 0; e < E; ++e) {
                const bool stereo = edges[e].obs[2] >= 0;
                if (stereo != (pass == 1)) continue;
                double A[9], B[18];
                visual_error(e, A, B);
                if ((float)chi2[e] < (stereo ? 24.f : 18.f)) res.outlier[e] = 0; else ++nBad;
            }
    }
    res.n_bad = nBad;
    res.n_inliers = nInliers;
    // ---- the new prior: Hessian at the final estimate (GetHessian* call linearizeOplus(), no robust weights) ----
    double e9[9], J[9 * 24];
    inertial_edge(other, cur, pint, e9, J);
    auto JtOJ = [&](const double* Jm, int rows, int cols, const double* Om, std::vector<double>& out) {  // out[cols x cols] = J^T Omega J
        out.assign((size_t)cols * cols, 0.0);
        for (int i = 0; i < cols; ++i)
            for (int j = 0; j < cols; ++j) {
                double h = 0;
                for (int r = 0; r < rows; ++r) { double t = 0; for (int c = 0; c < rows; ++c) t += Om[rows * r + c] * Jm[cols * c + j]; h += Jm[cols * r + i] * t; }
                out[(size_t)i * cols + j] = h;
            }
    };
    std::vector<double> Hv(36, 0.0);  // inlier visual edges: sum of B^T Omega B
    for (int e = 0; e < E; ++e) {
        if (res.outlier[e]) continue;
        double A[9], B[18], ebuf[3];
        const int dim = inertial_visual_edge(cur, cal, &Xw[3 * edges[e].point], edges[e], cam, ebuf, A, B);
        for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) { double h = 0; for (int d = 0; d < dim; ++d) h += B[6 * d + i] * edges[e].info * B[6 * d + j]; Hv[6 * i + j] += h; }
    }
    double Hn[225];
    std::memset(Hn, 0, sizeof(Hn));
    if (!last_frame) {
        // H(0:9, 0:9) += ei->GetHessian2() (pose and velocity of the frame), H(9:12) += InfoG, H(12:15) += InfoA, pose block += visual
        double J2[9 * 9];
        for (int r = 0; r < 9; ++r) for (int c = 0; c < 9; ++c) J2[9 * r + c] = J[24 * r + 15 + c];
        std::vector<double> H2;
        JtOJ(J2, 9, 9, infoI.data(), H2);
        for (int r = 0; r < 9; ++r) for (int c = 0; c < 9; ++c) Hn[15 * r + c] += H2[9 * r + c];
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { Hn[15 * (9 + r) + 9 + c] += infoG[3 * r + c]; Hn[15 * (12 + r) + 12 + c] += infoA[3 * r + c]; }
        for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) Hn[15 * r + c] += Hv[6 * r + c];
    } else {
        // 30 x 30: [previous frame P V G A | frame P V G A] (:3200-3262), then Marginalize(H, 0, 14)
        std::vector<double> H30(900, 0.0), H24;
        JtOJ(J, 9, 24, infoI.data(), H24);
        for (int r = 0; r < 24; ++r) for (int c = 0; c < 24; ++c) H30[30 * r + c] += H24[24 * r + c];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) {
                const double g = infoG[3 * r + c], a = infoA[3 * r + c];  // J = [-I, I]: blocks +, -, -, +
                H30[30 * (9 + r) + 9 + c] += g; H30[30 * (9 + r) + 24 + c] -= g; H30[30 * (24 + r) + 9 + c] -= g; H30[30 * (24 + r) + 24 + c] += g;
                H30[30 * (12 + r) + 12 + c] += a; H30[30 * (12 + r) + 27 + c] -= a; H30[30 * (27 + r) + 12 + c] -= a; H30[30 * (27 + r) + 27 + c] += a;
            }
        double e15[15], Jp[225];
        prior_edge(e15, Jp);
        std::vector<double> Hp;
        JtOJ(Jp, 15, 15, prior_prev->H, Hp);
        for (int r = 0; r < 15; ++r) for (int c = 0; c < 15; ++c) H30[30 * r + c] += Hp[15 * r + c];
        for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) H30[30 * (15 + r) + 15 + c] += Hv[6 * r + c];
        // Marginalize(H, 0, 14): c* = c - cb pinv(b) bc with b = H(0:15, 0:15); pinv through the SVD, singular values <= 1e-6 dropped.
        // b is symmetric: its SVD is the eigen decomposition (singular value = |eigenvalue|, U = V sign).
        std::vector<double> Bm(225), w, V;
        for (int r = 0; r < 15; ++r) for (int c = 0; c < 15; ++c) Bm[15 * r + c] = H30[30 * r + c];
        eig_sym(Bm, 15, w, V);
        std::vector<double> pinv(225, 0.0);
        for (int k = 0; k < 15; ++k) {
            if (!(std::fabs(w[k]) > 1e-6)) continue;
            const double iw = 1.0 / w[k];
            for (int r = 0; r < 15; ++r) for (int c = 0; c < 15; ++c) pinv[15 * r + c] += V[15 * r + k] * iw * V[15 * c + k];
        }
        for (int r = 0; r < 15; ++r)
            for (int c = 0; c < 15; ++c) {
                double s = 0;
                for (int i = 0; i < 15; ++i) { double t = 0; for (int j = 0; j < 15; ++j) t += pinv[15 * i + j] * H30[30 * j + 15 + c]; s += H30[30 * (15 + r) + i] * t; }
                Hn[15 * r + c] = H30[30 * (15 + r) + 15 + c] - s;
            }
    }
    constraint_clamp(Hn);
    std::memcpy(res.prior.Rwb, cur.Rwb, 72); std::memcpy(res.prior.twb, cur.twb, 24); std::memcpy(res.prior.vwb, cur.v, 24);
    std::memcpy(res.prior.bg, cur.bg, 24); std::memcpy(res.prior.ba, cur.ba, 24); std::memcpy(res.prior.H, Hn, sizeof(Hn));
    return res;
}


// ---- IMU initialisation: Optimizer::InertialOptimization (SF/src/Optimizer.cc:2169-2356, 2359-2466) --------------------------------------
// EdgeInertialGS (SF/src/G2oTypes.cc:603-724): the inertial residual with the gravity direction Rwg and the scale s as variables; err (er, ev,
// ep) and the Jacobian blocks of the non-fixed vertices, 9 x 15 row-major: V1 3 | gyro bias 3 | acc bias 3 | V2 3 | gravity direction 2 | scale 1
// (the two VertexPose are fixed in both optimisations).
void inertial_gs_edge(const InertialKeyFrame& k1, const InertialKeyFrame& k2, const double bg[3], const double ba[3], const double Rwg[9], double s,
                      const Preintegrated& pint, double err[9], double* J) {
    ImuBias b;
    b.bax = (float)ba[0]; b.bay = (float)ba[1]; b.baz = (float)ba[2];
    b.bwx = (float)bg[0]; b.bwy = (float)bg[1]; b.bwz = (float)bg[2];
    float dRf[9], dVf[3], dPf[3];
    pint.GetDeltaRotation(b, dRf); pint.GetDeltaVelocity(b, dVf); pint.GetDeltaPosition(b, dPf);
    double dR[9], dV[3], dP[3];
    for (int k = 0; k < 9; ++k) dR[k] = dRf[k];
    for (int k = 0; k < 3; ++k) { dV[k] = dVf[k]; dP[k] = dPf[k]; }
    const double dt = pint.dT, G = (double)9.81f;
    const double gI[3] = {0, 0, -G};
    double g[3];
    mulv(Rwg, gI, g);
    double Rbw1[9], dRt[9], t1[9], eR[9], er[3], dv[3], dp[3], rv[3], rp[3];
    tr(k1.Rwb, Rbw1); tr(dR, dRt);
    mul(dRt, Rbw1, t1); mul(t1, k2.Rwb, eR);
    LogSO3(eR, er);
    for (int k = 0; k < 3; ++k) {
        dv[k] = s * (k2.v[k] - k1.v[k]) - g[k] * dt;
        dp[k] = s * (k2.twb[k] - k1.twb[k] - k1.v[k] * dt) - g[k] * dt * dt / 2;
    }
    mulv(Rbw1, dv, rv); mulv(Rbw1, dp, rp);
    for (int k = 0; k < 3; ++k) { err[k] = er[k]; err[3 + k] = rv[k] - dV[k]; err[6 + k] = rp[k] - dP[k]; }
    if (!J) return;
    std::memset(J, 0, 9 * 15 * sizeof(double));
    auto put = [&](int r0, int c0, const double* m, double f) { for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) J[15 * (r0 + r) + c0 + c] = f * m[3 * r + c]; };
    put(3, 0, Rbw1, -s); put(6, 0, Rbw1, -s * dt);                                 // velocity 1
    double invJr[9], JRg[9], JVg[9], JPg[9], JVa[9], JPa[9], Jd[3], RJ[9], eRt[9], a1[9], a2[9], a3[9];
    inv_right_jacobian(er, invJr);
    for (int k = 0; k < 9; ++k) { JRg[k] = pint.JRg[k]; JVg[k] = pint.JVg[k]; JPg[k] = pint.JPg[k]; JVa[k] = pint.JVa[k]; JPa[k] = pint.JPa[k]; }
    const double dbg[3] = {(double)(b.bwx - pint.b.bwx), (double)(b.bwy - pint.b.bwy), (double)(b.bwz - pint.b.bwz)};  // GetDeltaBias is float
    mulv(JRg, dbg, Jd);
    right_jacobian(Jd, RJ);
    tr(eR, eRt);
    mul(invJr, eRt, a1); mul(a1, RJ, a2); mul(a2, JRg, a3);
    put(0, 3, a3, -1.0); put(3, 3, JVg, -1.0); put(6, 3, JPg, -1.0);               // gyro bias
    put(3, 6, JVa, -1.0); put(6, 6, JPa, -1.0);                                    // accelerometer bias
    put(3, 9, Rbw1, s);                                                            // velocity 2
    // gravity direction: dGdTheta = Rwg * Gm, Gm = [0 -G; G 0; 0 0]
    double dG[6];
    for (int r = 0; r < 3; ++r) { dG[2 * r] = Rwg[3 * r + 1] * G; dG[2 * r + 1] = Rwg[3 * r] * -G; }
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 2; ++c) {
            double v = 0;
            for (int k = 0; k < 3; ++k) v += Rbw1[3 * r + k] * dG[2 * k + c];
            J[15 * (3 + r) + 12 + c] = -v * dt;
            J[15 * (6 + r) + 12 + c] = -0.5 * v * dt * dt;
        }
    // scale
    double d1[3], d2[3], s1[3], s2[3];
    for (int k = 0; k < 3; ++k) { d1[k] = k2.v[k] - k1.v[k]; d2[k] = k2.twb[k] - k1.twb[k] - k1.v[k] * dt; }
    mulv(Rbw1, d1, s1); mulv(Rbw1, d2, s2);
    for (int r = 0; r < 3; ++r) { J[15 * (3 + r) + 14] = s1[r]; J[15 * (6 + r) + 14] = s2[r]; }
}

namespace {
// the variables of the two optimisations and where their increments sit in the solution vector (-1: fixed)
struct InitState {
    std::vector<InertialKeyFrame> kfs;
    double bg[3], ba[3], Rwg[9], s;
};
void gdir_update(double Rwg[9], double u0, double u1) {  // GDirection::Update: Rwg = Rwg * ExpSO3(u0, u1, 0)
    const double w[3] = {u0, u1, 0.0};
    double E[9], R[9];
    ExpSO3(w, E);
    mul(Rwg, E, R);
    std::memcpy(Rwg, R, sizeof(R));
}
}  // namespace

InertialInitResult InertialOptimization(std::vector<InertialKeyFrame>& kfs, const std::vector<const Preintegrated*>& pints, double Rwg[9], double& scale,
                                        double bg[3], double ba[3], bool mono, bool fixed_vel, float priorG, float priorA, int its) {
    const int N = (int)kfs.size();
    InertialInitResult res;
    // unknowns: velocities | gyro bias | acc bias | gravity direction | scale
    const int o_v = 0, n_v = fixed_vel ? 0 : 3 * N, o_bg = fixed_vel ? -1 : n_v, o_ba = fixed_vel ? -1 : n_v + 3, o_g = fixed_vel ? 0 : n_v + 6,
              o_s = mono ? o_g + 2 : -1, n = o_g + 2 + (mono ? 1 : 0);
    std::vector<std::vector<double>> infos(N);
    for (int i = 1; i < N; ++i) if (pints[i]) edge_inertial_information(*pints[i], infos[i]);
    InitState st;
    st.kfs = kfs; std::memcpy(st.bg, bg, 24); std::memcpy(st.ba, ba, 24); std::memcpy(st.Rwg, Rwg, 72); st.s = scale;
    const double infoA = priorA, infoG = priorG;
    auto chi2_of = [&](const InitState& x) {  // computeActiveErrors + activeRobustChi2 (no robust kernels here)
        double chi = 0;
        for (int i = 1; i < N; ++i) {
            if (!pints[i]) continue;
            double e[9];
            inertial_gs_edge(x.kfs[i - 1], x.kfs[i], x.bg, x.ba, x.Rwg, x.s, *pints[i], e, nullptr);
            for (int r = 0; r < 9; ++r) for (int c = 0; c < 9; ++c) chi += e[r] * infos[i][9 * r + c] * e[c];
        }
        for (int k = 0; k < 3; ++k) chi += infoA * x.ba[k] * x.ba[k] + infoG * x.bg[k] * x.bg[k];  // EdgePriorAcc / EdgePriorGyro: e = 0 - b
        return chi;
    };
    std::vector<double> H((size_t)n * n), b(n), x(n);
    auto build = [&](const InitState& xs) {
        std::fill(H.begin(), H.end(), 0.0); std::fill(b.begin(), b.end(), 0.0);
        for (int i = 1; i < N; ++i) {
            if (!pints[i]) continue;
            double e[9], J[9 * 15];
            inertial_gs_edge(xs.kfs[i - 1], xs.kfs[i], xs.bg, xs.ba, xs.Rwg, xs.s, *pints[i], e, J);
            // column -> unknown
            int col[15];
            for (int k = 0; k < 3; ++k) {
                col[k] = fixed_vel ? -1 : o_v + 3 * (i - 1) + k; col[3 + k] = o_bg < 0 ? -1 : o_bg + k; col[6 + k] = o_ba < 0 ? -1 : o_ba + k;
                col[9 + k] = fixed_vel ? -1 : o_v + 3 * i + k;
            }
            col[12] = o_g; col[13] = o_g + 1; col[14] = o_s;
            double OJ[9 * 15], Oe[9];
            for (int r = 0; r < 9; ++r) {
                for (int c = 0; c < 15; ++c) { double v = 0; for (int k = 0; k < 9; ++k) v += infos[i][9 * r + k] * J[15 * k + c]; OJ[15 * r + c] = v; }
                double v = 0; for (int k = 0; k < 9; ++k) v += infos[i][9 * r + k] * e[k]; Oe[r] = v;
            }
            for (int a = 0; a < 15; ++a) {
                if (col[a] < 0) continue;
                double g = 0;
                for (int r = 0; r < 9; ++r) g += J[15 * r + a] * Oe[r];
                b[col[a]] -= g;
                for (int c = 0; c < 15; ++c) {
                    if (col[c] < 0) continue;
                    double h = 0;
                    for (int r = 0; r < 9; ++r) h += J[15 * r + a] * OJ[15 * r + c];
                    H[(size_t)col[a] * n + col[c]] += h;
                }
            }
        }
        // the bias priors: error bprior - b with bprior = 0 and the Jacobian the reference declares, +I (SF/src/G2oTypes.cc:769-781)
        if (o_ba >= 0) for (int k = 0; k < 3; ++k) { H[(size_t)(o_ba + k) * n + o_ba + k] += infoA; b[o_ba + k] -= infoA * (0.0 - xs.ba[k]); }
        if (o_bg >= 0) for (int k = 0; k < 3; ++k) { H[(size_t)(o_bg + k) * n + o_bg + k] += infoG; b[o_bg + k] -= infoG * (0.0 - xs.bg[k]); }
    };
    auto apply = [&](InitState& xs, const std::vector<double>& u) {
        if (!fixed_vel) {
            for (int i = 0; i < N; ++i) for (int k = 0; k < 3; ++k) xs.kfs[i].v[k] += u[o_v + 3 * i + k];
            for (int k = 0; k < 3; ++k) { xs.bg[k] += u[o_bg + k]; xs.ba[k] += u[o_ba + k]; }
        }
        gdir_update(xs.Rwg, u[o_g], u[o_g + 1]);
        if (mono) xs.s *= std::exp(u[o_s]);
    };
    // OptimizationAlgorithmLevenberg::solve (core/optimization_algorithm_levenberg.cpp:61-169), dense system (all vertices in one block set:
    // BlockSolverX without marginalised vertices; LinearSolverEigen = a Cholesky factorisation of it)
    double lambda = 0, ni = 2;
    int n_bad = 0;
    res.err = chi2_of(st);
    for (int it = 0; it < its; ++it) {
        double currentChi = chi2_of(st), tempChi = currentChi;
        const double iniChi = currentChi;
        build(st);
        if (it == 0) {
            if (priorG != 0.f) lambda = 1e3;  // setUserLambdaInit(1e3)
            else { double md = 0; for (int j = 0; j < n; ++j) md = std::fmax(md, std::fabs(H[(size_t)j * n + j])); lambda = 1e-5 * md; }
            ni = 2; n_bad = 0;
        }
        double rho = 0;
        int qmax = 0;
        do {
            InitState backup = st;
            std::vector<double> Hl = H;
            for (int j = 0; j < n; ++j) Hl[(size_t)j * n + j] += lambda;
            const bool ok2 = ldlt(Hl, n, b.data(), x.data());
            apply(st, x);
            tempChi = chi2_of(st);
            if (!ok2) tempChi = std::numeric_limits<double>::max();
            rho = currentChi - tempChi;
            double sc = 0;
            for (int j = 0; j < n; ++j) sc += x[j] * (lambda * x[j] + b[j]);
            sc += 1e-3;
            rho /= sc;
            if (rho > 0 && std::isfinite(tempChi)) {
                double alpha = 1. - std::pow((2 * rho - 1), 3);
                alpha = std::min(alpha, 2. / 3.);
                lambda *= std::max(1. / 3., alpha);
                ni = 2;
                currentChi = tempChi;
            } else {
                lambda *= ni;
                ni *= 2;
                st = backup;
            }
            qmax++;
            res.trials++;
        } while (rho < 0 && qmax < 10);
        res.iterations++;
        res.trace.chi2.push_back(currentChi); res.trace.lambda.push_back(lambda); res.trace.trials.push_back(qmax);
        if (qmax == 10 || rho == 0) break;
        if ((iniChi - currentChi) * 1e3 < iniChi) n_bad++; else n_bad = 0;
        if (n_bad >= 3) break;
    }
    res.err_end = chi2_of(st);
    kfs = st.kfs; std::memcpy(bg, st.bg, 24); std::memcpy(ba, st.ba, 24); std::memcpy(Rwg, st.Rwg, 72); scale = st.s;
    return res;
}

// Optimizer::InertialOptimization(pMap, Rwg, scale) (SF/src/Optimizer.cc:2359-2466; LocalMapping::ScaleRefinement): Gauss-Newton
// (core/optimization_algorithm_gauss_newton.cpp:49-93: build, solve, update -- no step control), 10 iterations, only the gravity direction
// and the scale are variables; every EdgeInertialGS takes the biases of its earlier keyframe and carries a Huber kernel with delta 1
// (robustInformation = rho'(chi2) * information, omega_r = -rho' * information * e; core/base_multi_edge.hpp).  Returns the iterations.
int InertialScaleRefinement(const std::vector<InertialKeyFrame>& kfs, const std::vector<const Preintegrated*>& pints, double Rwg[9], double& scale, int its,
                            double err2[2]) {
    const int N = (int)kfs.size();
    std::vector<std::vector<double>> infos(N);
    for (int i = 1; i < N; ++i) if (pints[i]) edge_inertial_information(*pints[i], infos[i]);
    const HuberD huber(1.f);
    auto robust_chi2 = [&](const double R[9], double s) {
        double tot = 0;
        for (int i = 1; i < N; ++i) {
            if (!pints[i]) continue;
            double e[9], c = 0, r0, r1;
            inertial_gs_edge(kfs[i - 1], kfs[i], kfs[i - 1].bg, kfs[i - 1].ba, R, s, *pints[i], e, nullptr);
            for (int r = 0; r < 9; ++r) for (int q = 0; q < 9; ++q) c += e[r] * infos[i][9 * r + q] * e[q];
            huber.rho(c, r0, r1);
            tot += r0;
        }
        return tot;
    };
    if (err2) err2[0] = robust_chi2(Rwg, scale);
    int done = 0;
    for (int it = 0; it < its; ++it) {
        std::vector<double> H(9, 0.0);
        double b[3] = {0, 0, 0}, x[3];
        for (int i = 1; i < N; ++i) {
            if (!pints[i]) continue;
            double e[9], J[135], c = 0, r0, r1;
            inertial_gs_edge(kfs[i - 1], kfs[i], kfs[i - 1].bg, kfs[i - 1].ba, Rwg, scale, *pints[i], e, J);
            double Oe[9];
            for (int r = 0; r < 9; ++r) { double v = 0; for (int q = 0; q < 9; ++q) v += infos[i][9 * r + q] * e[q]; Oe[r] = v; c += e[r] * v; }
            huber.rho(c, r0, r1);
            for (int a = 0; a < 3; ++a) {
                double g = 0;
                for (int r = 0; r < 9; ++r) g += J[15 * r + 12 + a] * Oe[r];
                b[a] -= r1 * g;
                for (int q = 0; q < 3; ++q) {
                    double h = 0;
                    for (int r = 0; r < 9; ++r) { double oj = 0; for (int k = 0; k < 9; ++k) oj += infos[i][9 * r + k] * J[15 * k + 12 + q]; h += J[15 * r + 12 + a] * oj; }
                    H[3 * a + q] += r1 * h;
                }
            }
        }
        if (!ldlt(H, 3, b, x)) break;
        gdir_update(Rwg, x[0], x[1]);
        scale *= std::exp(x[2]);
        ++done;
    }
    if (err2) err2[1] = robust_chi2(Rwg, scale);
    return done;
}

namespace {
void so3f_exp_matrix(const float v[3], float R[9]) {  // Sophus::SO3f::exp(v).matrix(), evaluated in double and rounded
    const double w[3] = {v[0], v[1], v[2]};
    const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], th = std::sqrt(th2);
    double W[9], W2[9];
    hat(w, W); mul(W, W, W2);
    const double a = th < 1e-8 ? 1.0 - th2 / 6 : std::sin(th) / th, bq = th < 1e-8 ? 0.5 - th2 / 24 : (1 - std::cos(th)) / th2;
    for (int k = 0; k < 9; ++k) R[k] = (float)((k % 4 == 0 ? 1.0 : 0.0) + a * W[k] + bq * W2[k]);
}
}  // namespace

// LocalMapping::InitializeIMU, the first estimate (SF/src/LocalMapping.cc:1241-1270): dirG = -sum Rwb_prev * dV, the keyframe velocities
// from the position differences, Rwg = exp(v * ang / |v|) with v = gI x dirG -- all in float (Eigen::Vector3f, Sophus::SO3f).
void InitialGravityDirection(const std::vector<InertialKeyFrame>& kfs, const std::vector<const Preintegrated*>& pints, float vel[], float Rwg[9]) {
    const int N = (int)kfs.size();
    float dirG[3] = {0, 0, 0};
    for (int i = 0; i < N; ++i) for (int k = 0; k < 3; ++k) vel[3 * i + k] = (float)kfs[i].v[k];
    for (int i = 1; i < N; ++i) {
        if (!pints[i]) continue;
        float dV[3];
        pints[i]->GetDeltaVelocity(pints[i]->b, dV);  // GetUpdatedDeltaVelocity at the bias of the integration: dV itself
        float R[9], p1[3], p0[3];
        for (int k = 0; k < 9; ++k) R[k] = (float)kfs[i - 1].Rwb[k];
        for (int k = 0; k < 3; ++k) { p1[k] = (float)kfs[i].twb[k]; p0[k] = (float)kfs[i - 1].twb[k]; }
        for (int r = 0; r < 3; ++r) dirG[r] -= (R[3 * r] * dV[0] + R[3 * r + 1] * dV[1]) + R[3 * r + 2] * dV[2];
        for (int k = 0; k < 3; ++k) { const float v = (p1[k] - p0[k]) / pints[i]->dT; vel[3 * i + k] = v; vel[3 * (i - 1) + k] = v; }
    }
    const float nrm = std::sqrt((dirG[0] * dirG[0] + dirG[1] * dirG[1]) + dirG[2] * dirG[2]);
    for (int k = 0; k < 3; ++k) dirG[k] = dirG[k] / nrm;
    const float gI[3] = {0.0f, 0.0f, -1.0f};
    const float v[3] = {gI[1] * dirG[2] - gI[2] * dirG[1], gI[2] * dirG[0] - gI[0] * dirG[2], gI[0] * dirG[1] - gI[1] * dirG[0]};
    const float nv = std::sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
    const float cosg = (gI[0] * dirG[0] + gI[1] * dirG[1]) + gI[2] * dirG[2];
    const float ang = std::acos(cosg);
    const float vzg[3] = {v[0] * ang / nv, v[1] * ang / nv, v[2] * ang / nv};
    so3f_exp_matrix(vzg, Rwg);
}

}  // namespace oracle
