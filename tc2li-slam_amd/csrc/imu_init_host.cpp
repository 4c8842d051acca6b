// IMU initialisation behind tc2li_inertial_optimization / tc2li_imu_init_gravity (include/tc2li_hip.h), SURVEY.md section 8f item 4:
//   Optimizer::InertialOptimization, first overload          SF/src/Optimizer.cc:2169-2356
//   EdgeInertialGS, EdgePriorAcc / EdgePriorGyro             SF/src/G2oTypes.cc:603-724, 769-781
//   VertexGDir (GDirection::Update), VertexScale             SF/include/G2oTypes.h:267-330
//   LocalMapping::InitializeIMU, first gravity estimate      SF/src/LocalMapping.cc:1241-1270
//   g2o Levenberg-Marquardt control                          Thirdparty/g2o/g2o/core/optimization_algorithm_levenberg.cpp:61-169
// Host code by design, like the pre-integration (row a11) and the inertial edges of the local BA (row c6): a few tens of keyframes, one
// 9-dimensional edge per consecutive pair.  The normal equations have an arrow shape -- the velocities only couple to their neighbours in
// time and to the nine shared unknowns (biases, gravity direction, scale) -- and are solved as such: the block-tridiagonal velocity part
// is factorised in O(N) 3 x 3 steps, the shared part through its Schur complement (g2o hands the reference the same system as a
// sparse Cholesky; the results agree to rounding).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

#include "common.hpp"
#include "inertial_host.hpp"

using namespace tc2li;
using namespace tc2li::inertial_detail;

namespace {

constexpr double kG = (double)9.81f;  // IMU::GRAVITY_VALUE is a float constant

struct InitVars {
    std::vector<double> v;  // [3 N]
    double bg[3], ba[3], Rwg[9], s;
};

// EdgeInertialGS between keyframes i - 1 and i: error and the Jacobian columns V1 3 | bg 3 | ba 3 | V2 3 | gdir 2 | scale 1 (9 x 15, row-major)
void gs_edge(const double* Rwb1, const double* twb1, const double* v1, const double* Rwb2, const double* twb2, const double* v2, const InitVars& x,
             const tc2li_preintegrated* pre, double err[9], double* J) {
    const tc2li_imu_bias b{(float)x.ba[0], (float)x.ba[1], (float)x.ba[2], (float)x.bg[0], (float)x.bg[1], (float)x.bg[2]};
    float dRf[9], dVf[3], dPf[3];
    tc2li_imu_delta(pre, &b, dRf, dVf, dPf);
    double dR[9], g[3], Rbw1[9], dRt[9], t1[9], eR[9], er[3], dv[3], dp[3], rv[3], rp[3];
    for (int k = 0; k < 9; ++k) dR[k] = dRf[k];
    const double gI[3] = {0, 0, -kG}, dt = pre->dT, s = x.s;
    r3_vec(x.Rwg, gI, g);
    tr3(Rwb1, Rbw1); tr3(dR, dRt);
    r3_mul(dRt, Rbw1, t1); r3_mul(t1, Rwb2, eR);
    log_so3(eR, er);
    for (int k = 0; k < 3; ++k) {
        dv[k] = s * (v2[k] - v1[k]) - g[k] * dt;
        dp[k] = s * (twb2[k] - twb1[k] - v1[k] * dt) - g[k] * dt * dt / 2;
    }
    r3_vec(Rbw1, dv, rv); r3_vec(Rbw1, dp, rp);
    for (int k = 0; k < 3; ++k) { err[k] = er[k]; err[3 + k] = rv[k] - (double)dVf[k]; err[6 + k] = rp[k] - (double)dPf[k]; }
    if (!J) return;
    memset(J, 0, 9 * 15 * sizeof(double));
    auto put = [&](int r0, int c0, const double* m, double f) { for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) J[15 * (r0 + r) + c0 + c] = f * m[3 * r + c]; };
    put(3, 0, Rbw1, -s); put(6, 0, Rbw1, -s * dt);
    double invJr[9], JRg[9], JVg[9], JPg[9], JVa[9], JPa[9], Jd[3], RJ[9], eRt[9], a1[9], a2[9], a3[9];
    jr_so3(er, true, invJr);
    for (int k = 0; k < 9; ++k) { JRg[k] = pre->JRg[k]; JVg[k] = pre->JVg[k]; JPg[k] = pre->JPg[k]; JVa[k] = pre->JVa[k]; JPa[k] = pre->JPa[k]; }
    const double dbg[3] = {(double)(b.bwx - pre->bias.bwx), (double)(b.bwy - pre->bias.bwy), (double)(b.bwz - pre->bias.bwz)};
    r3_vec(JRg, dbg, Jd);
    jr_so3(Jd, false, RJ);
    tr3(eR, eRt);
    r3_mul(invJr, eRt, a1); r3_mul(a1, RJ, a2); r3_mul(a2, JRg, a3);
    put(0, 3, a3, -1.0); put(3, 3, JVg, -1.0); put(6, 3, JPg, -1.0);
    put(3, 6, JVa, -1.0); put(6, 6, JPa, -1.0);
    put(3, 9, Rbw1, s);
    // dGdTheta = Rwg * [0 -G; G 0; 0 0]
    for (int r = 0; r < 3; ++r) {
        double c0 = 0, c1 = 0;
        for (int k = 0; k < 3; ++k) { c0 += Rbw1[3 * r + k] * (x.Rwg[3 * k + 1] * kG); c1 += Rbw1[3 * r + k] * (x.Rwg[3 * k] * -kG); }
        J[15 * (3 + r) + 12] = -c0 * dt; J[15 * (3 + r) + 13] = -c1 * dt;
        J[15 * (6 + r) + 12] = -0.5 * c0 * dt * dt; J[15 * (6 + r) + 13] = -0.5 * c1 * dt * dt;
    }
    double d1[3], d2[3], s1[3], s2[3];
    for (int k = 0; k < 3; ++k) { d1[k] = v2[k] - v1[k]; d2[k] = twb2[k] - twb1[k] - v1[k] * dt; }
    r3_vec(Rbw1, d1, s1); r3_vec(Rbw1, d2, s2);
    for (int r = 0; r < 3; ++r) { J[15 * (3 + r) + 14] = s1[r]; J[15 * (6 + r) + 14] = s2[r]; }
}

// 3 x 3 helpers of the arrow solver
inline bool inv3(const double* a, double* o) {
    const double c0 = a[4] * a[8] - a[5] * a[7], c1 = a[5] * a[6] - a[3] * a[8], c2 = a[3] * a[7] - a[4] * a[6];
    const double det = a[0] * c0 + a[1] * c1 + a[2] * c2;
    if (!(std::fabs(det) > 0) || !std::isfinite(det)) return false;
    const double id = 1.0 / det;
    o[0] = c0 * id; o[1] = (a[2] * a[7] - a[1] * a[8]) * id; o[2] = (a[1] * a[5] - a[2] * a[4]) * id;
    o[3] = c1 * id; o[4] = (a[0] * a[8] - a[2] * a[6]) * id; o[5] = (a[2] * a[3] - a[0] * a[5]) * id;
    o[6] = c2 * id; o[7] = (a[1] * a[6] - a[0] * a[7]) * id; o[8] = (a[0] * a[4] - a[1] * a[3]) * id;
    return true;
}

// The system  [ T  B ] [xv]   [bv]     T: block tridiagonal (N blocks of 3 x 3: diagonal D_i, sub-diagonal L_i = T(i, i-1)),
//             [ B' C ] [xc] = [bc]     B: 3N x m border, C: m x m (m <= 9); symmetric.
// Block LDL' of T (S_i = D_i - L_i S_{i-1}^-1 L_i'), then the Schur complement of C.  false: a pivot block is singular.
struct ArrowSystem {
    int N = 0, m = 0;
    std::vector<double> D, L, B, C, bv, bc;  // D [N][9], L [N][9] (L[0] unused), B [3N][m], C [m][m]
    void reset(int N_, int m_) {
        N = N_; m = m_;
        D.assign((size_t)9 * N, 0.0); L.assign((size_t)9 * N, 0.0); B.assign((size_t)3 * N * m, 0.0); C.assign((size_t)m * m, 0.0);
        bv.assign((size_t)3 * N, 0.0); bc.assign(m, 0.0);
    }
    bool solve(double lambda, std::vector<double>& xv, std::vector<double>& xc) const {
        // forward: W_i = S_i^-1 applied to [B_i | bv_i] after eliminating the block above
        const int w = m + 1;
        std::vector<double> Sinv((size_t)9 * N), Y((size_t)3 * N * w);  // Y_i = rows of (B | bv) after the elimination
        for (int i = 0; i < N; ++i) {
            double S[9];
            for (int k = 0; k < 9; ++k) S[k] = D[9 * (size_t)i + k];
            S[0] += lambda; S[4] += lambda; S[8] += lambda;
            for (int r = 0; r < 3; ++r) { for (int c = 0; c < m; ++c) Y[(size_t)(3 * i + r) * w + c] = B[(size_t)(3 * i + r) * m + c]; Y[(size_t)(3 * i + r) * w + m] = bv[3 * (size_t)i + r]; }
            if (i > 0) {
                // G = L_i S_{i-1}^-1;  S -= G L_i';  Y_i -= G Y_{i-1}
                double G[9];
                r3_mul(&L[9 * (size_t)i], &Sinv[9 * (size_t)(i - 1)], G);
                for (int r = 0; r < 3; ++r)
                    for (int c = 0; c < 3; ++c) { double t = 0; for (int k = 0; k < 3; ++k) t += G[3 * r + k] * L[9 * (size_t)i + 3 * c + k]; S[3 * r + c] -= t; }
                for (int r = 0; r < 3; ++r)
                    for (int c = 0; c < w; ++c) { double t = 0; for (int k = 0; k < 3; ++k) t += G[3 * r + k] * Y[(size_t)(3 * (i - 1) + k) * w + c]; Y[(size_t)(3 * i + r) * w + c] -= t; }
            }
            if (!inv3(S, &Sinv[9 * (size_t)i])) return false;
        }
        // Schur complement of the border: Cs = C + lambda I - sum_i Y_i' S_i^-1 Y_i(:, :m),  rhs = bc - sum_i Y_i(:, :m)' S_i^-1 Y_i(:, m)
        std::vector<double> Cs((size_t)m * m), rhs(m), Z((size_t)3 * N * w);
        for (int i = 0; i < N; ++i)
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < w; ++c) { double t = 0; for (int k = 0; k < 3; ++k) t += Sinv[9 * (size_t)i + 3 * r + k] * Y[(size_t)(3 * i + k) * w + c]; Z[(size_t)(3 * i + r) * w + c] = t; }
        for (int a = 0; a < m; ++a) {
            for (int c = 0; c < m; ++c) {
                double t = C[(size_t)a * m + c] + (a == c ? lambda : 0.0);
                for (int q = 0; q < 3 * N; ++q) t -= Y[(size_t)q * w + a] * Z[(size_t)q * w + c];
                Cs[(size_t)a * m + c] = t;
            }
            double t = bc[a];
            for (int q = 0; q < 3 * N; ++q) t -= Y[(size_t)q * w + a] * Z[(size_t)q * w + m];
            rhs[a] = t;
        }
        xc.assign(m, 0.0);
        if (m > 0 && !ldlt_solve_small(Cs.data(), m, rhs.data(), xc.data(), false)) return false;
        // back substitution: x_i = S_i^-1 (Y_i(:, m) - Y_i(:, :m) xc) - S_i^-1 L_{i+1}' x_{i+1}
        xv.assign((size_t)3 * N, 0.0);
        for (int i = N - 1; i >= 0; --i) {
            double r3[3];
            for (int r = 0; r < 3; ++r) {
                double t = Y[(size_t)(3 * i + r) * w + m];
                for (int c = 0; c < m; ++c) t -= Y[(size_t)(3 * i + r) * w + c] * xc[c];
                if (i + 1 < N) for (int k = 0; k < 3; ++k) t -= L[9 * (size_t)(i + 1) + 3 * k + r] * xv[3 * (size_t)(i + 1) + k];
                r3[r] = t;
            }
            r3_vec(&Sinv[9 * (size_t)i], r3, &xv[3 * (size_t)i]);
        }
        return true;
    }
};

}  // namespace

extern "C" {

int tc2li_inertial_optimization(int n_kfs, const double* Rwb9, const double* twb3, double* vel3, const tc2li_preintegrated* const* pre, double Rwg9[9],
                                double* scale, double bg[3], double ba[3], int mono, int fixed_vel, float prior_g, float prior_a, int iterations,
                                tc2li_inertial_init_stats* stats) {
    if (n_kfs < 2 || !Rwb9 || !twb3 || !vel3 || !pre || !Rwg9 || !scale || !bg || !ba || iterations < 0) {
        set_error("tc2li_inertial_optimization: invalid argument");
        return TC2LI_ERR_INVALID;
    }
    const int N = n_kfs;
    std::vector<InertialLinkHost> links(N);
    for (int i = 1; i < N; ++i) {
        if (!pre[i]) continue;  // the reference prints "Not preintegrated measurement" and dereferences; a missing link is skipped here
        links[i].pre = pre[i];
        if (!links[i].prepare(1.0)) { set_error("tc2li_inertial_optimization: covariance of link %d is not positive definite", i); return TC2LI_ERR_INVALID; }
    }
    InitVars x;
    x.v.assign(vel3, vel3 + 3 * (size_t)N);
    memcpy(x.bg, bg, 24); memcpy(x.ba, ba, 24); memcpy(x.Rwg, Rwg9, 72); x.s = *scale;
    // shared unknowns: [bg 3 | ba 3 |] gdir 2 [| scale 1]
    const int o_bg = fixed_vel ? -1 : 0, o_ba = fixed_vel ? -1 : 3, o_g = fixed_vel ? 0 : 6, o_s = mono ? o_g + 2 : -1, m = o_g + 2 + (mono ? 1 : 0);
    const int Nv = fixed_vel ? 0 : N;
    auto chi2_of = [&](const InitVars& y) {
        double chi = 0;
        for (int i = 1; i < N; ++i) {
            if (!links[i].pre) continue;
            double e[9];
            gs_edge(Rwb9 + 9 * (size_t)(i - 1), twb3 + 3 * (size_t)(i - 1), &y.v[3 * (size_t)(i - 1)], Rwb9 + 9 * (size_t)i, twb3 + 3 * (size_t)i, &y.v[3 * (size_t)i], y,
                    links[i].pre, e, nullptr);
            for (int r = 0; r < 9; ++r) for (int c = 0; c < 9; ++c) chi += e[r] * links[i].info[9 * r + c] * e[c];
        }
        for (int k = 0; k < 3; ++k) chi += (double)prior_a * y.ba[k] * y.ba[k] + (double)prior_g * y.bg[k] * y.bg[k];
        return chi;
    };
    ArrowSystem A;
    auto build = [&](const InitVars& y) {
        A.reset(Nv, m);
        for (int i = 1; i < N; ++i) {
            if (!links[i].pre) continue;
            double e[9], J[135], OJ[135], Oe[9];
            gs_edge(Rwb9 + 9 * (size_t)(i - 1), twb3 + 3 * (size_t)(i - 1), &y.v[3 * (size_t)(i - 1)], Rwb9 + 9 * (size_t)i, twb3 + 3 * (size_t)i, &y.v[3 * (size_t)i], y,
                    links[i].pre, e, J);
            const double* O = links[i].info;
            for (int r = 0; r < 9; ++r) {
                for (int c = 0; c < 15; ++c) { double t = 0; for (int k = 0; k < 9; ++k) t += O[9 * r + k] * J[15 * k + c]; OJ[15 * r + c] = t; }
                double t = 0; for (int k = 0; k < 9; ++k) t += O[9 * r + k] * e[k]; Oe[r] = t;
            }
            // column -> (velocity block, component) or shared index
            int vb[15], sh[15];
            for (int k = 0; k < 15; ++k) { vb[k] = -1; sh[k] = -1; }
            for (int k = 0; k < 3; ++k) {
                if (!fixed_vel) { vb[k] = i - 1; vb[9 + k] = i; }
                sh[3 + k] = o_bg < 0 ? -1 : o_bg + k; sh[6 + k] = o_ba < 0 ? -1 : o_ba + k;
            }
            sh[12] = o_g; sh[13] = o_g + 1; sh[14] = o_s;
            auto comp = [](int col) { return col < 3 ? col : col - 9; };
            for (int a = 0; a < 15; ++a) {
                double grad = 0;
                for (int r = 0; r < 9; ++r) grad += J[15 * r + a] * Oe[r];
                if (vb[a] >= 0) A.bv[3 * (size_t)vb[a] + comp(a)] -= grad;
                else if (sh[a] >= 0) A.bc[sh[a]] -= grad;
                else continue;
                for (int c = 0; c < 15; ++c) {
                    if (vb[c] < 0 && sh[c] < 0) continue;
                    double h = 0;
                    for (int r = 0; r < 9; ++r) h += J[15 * r + a] * OJ[15 * r + c];
                    if (vb[a] >= 0 && vb[c] >= 0) {
                        if (vb[a] == vb[c]) A.D[9 * (size_t)vb[a] + 3 * comp(a) + comp(c)] += h;
                        else if (vb[a] == vb[c] + 1) A.L[9 * (size_t)vb[a] + 3 * comp(a) + comp(c)] += h;  // T(i, i-1)
                    } else if (vb[a] >= 0 && sh[c] >= 0) {
                        A.B[(size_t)(3 * vb[a] + comp(a)) * m + sh[c]] += h;
                    } else if (sh[a] >= 0 && sh[c] >= 0) {
                        A.C[(size_t)sh[a] * m + sh[c]] += h;
                    }
                }
            }
        }
        // EdgePriorAcc / EdgePriorGyro: error = 0 - b, Jacobian as the reference declares it: +I (SF/src/G2oTypes.cc:769-781)
        if (o_ba >= 0) for (int k = 0; k < 3; ++k) { A.C[(size_t)(o_ba + k) * m + o_ba + k] += (double)prior_a; A.bc[o_ba + k] -= (double)prior_a * (0.0 - y.ba[k]); }
        if (o_bg >= 0) for (int k = 0; k < 3; ++k) { A.C[(size_t)(o_bg + k) * m + o_bg + k] += (double)prior_g; A.bc[o_bg + k] -= (double)prior_g * (0.0 - y.bg[k]); }
    };
    auto apply = [&](InitVars& y, const std::vector<double>& xv, const std::vector<double>& xc) {
        if (!fixed_vel) {
            for (size_t k = 0; k < y.v.size(); ++k) y.v[k] += xv[k];
            for (int k = 0; k < 3; ++k) { y.bg[k] += xc[o_bg + k]; y.ba[k] += xc[o_ba + k]; }
        }
        const double w[3] = {xc[o_g], xc[o_g + 1], 0.0};  // GDirection::Update
        double E[9], R[9];
        exp_so3(w, E);
        r3_mul(y.Rwg, E, R);
        memcpy(y.Rwg, R, sizeof(R));
        if (mono) y.s *= std::exp(xc[o_s]);
    };
    double lambda = 0, ni = 2;
    int n_bad = 0, done = 0, trials = 0;
    const double err0 = chi2_of(x);
    std::vector<double> xv, xc;
    for (int it = 0; it < iterations; ++it) {
        double currentChi = chi2_of(x), tempChi = currentChi;
        const double iniChi = currentChi;
        build(x);
        if (it == 0) {
            if (prior_g != 0.f) lambda = 1e3;
            else {
                double md = 0;
                for (int i = 0; i < Nv; ++i) for (int k = 0; k < 3; ++k) md = std::max(md, std::fabs(A.D[9 * (size_t)i + 4 * k]));
                for (int k = 0; k < m; ++k) md = std::max(md, std::fabs(A.C[(size_t)k * m + k]));
                lambda = 1e-5 * md;
            }
            ni = 2; n_bad = 0;
        }
        double rho = 0;
        int qmax = 0;
        do {
            const InitVars backup = x;
            const bool ok2 = A.solve(lambda, xv, xc);
            if (ok2) apply(x, xv, xc);
            tempChi = ok2 ? chi2_of(x) : std::numeric_limits<double>::max();
            rho = currentChi - tempChi;
            double sc = 0;
            if (ok2) {
                for (size_t k = 0; k < xv.size(); ++k) sc += xv[k] * (lambda * xv[k] + A.bv[k]);
                for (int k = 0; k < m; ++k) sc += xc[k] * (lambda * xc[k] + A.bc[k]);
            }
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
                x = backup;
            }
            ++qmax; ++trials;
        } while (rho < 0 && qmax < 10);
        ++done;
        if (qmax == 10 || rho == 0) break;
        if ((iniChi - currentChi) * 1e3 < iniChi) n_bad++; else n_bad = 0;
        if (n_bad >= 3) break;
    }
    memcpy(vel3, x.v.data(), 3 * (size_t)N * sizeof(double));
    memcpy(bg, x.bg, 24); memcpy(ba, x.ba, 24); memcpy(Rwg9, x.Rwg, 72); *scale = x.s;
    if (stats) { stats->iterations = done; stats->trials = trials; stats->initial_chi2 = err0; stats->final_chi2 = chi2_of(x); stats->final_lambda = lambda; }
    return done;
}

int tc2li_inertial_scale_refinement(int n_kfs, const double* Rwb9, const double* twb3, const double* vel3, const double* bg3, const double* ba3,
                                    const tc2li_preintegrated* const* pre, double Rwg9[9], double* scale, int iterations, double chi2[2]) {
    if (n_kfs < 2 || !Rwb9 || !twb3 || !vel3 || !bg3 || !ba3 || !pre || !Rwg9 || !scale || iterations < 0) {
        set_error("tc2li_inertial_scale_refinement: invalid argument");
        return TC2LI_ERR_INVALID;
    }
    const int N = n_kfs;
    std::vector<InertialLinkHost> links(N);
    for (int i = 1; i < N; ++i) {
        if (!pre[i]) continue;
        links[i].pre = pre[i];
        if (!links[i].prepare(1.0)) { set_error("tc2li_inertial_scale_refinement: covariance of link %d is not positive definite", i); return TC2LI_ERR_INVALID; }
    }
    InitVars x;
    x.v.assign(vel3, vel3 + 3 * (size_t)N);
    memcpy(x.Rwg, Rwg9, 72); x.s = *scale;
    const float dsqr = 1.f;  // RobustKernelHuber, delta 1
    auto huber = [&](double c, double& r0, double& r1) { if (c <= dsqr) { r0 = c; r1 = 1.0; } else { const double q = std::sqrt(c); r0 = 2 * q * 1.0 - dsqr; r1 = 1.0 / q; } };
    auto edge = [&](int i, double e[9], double* J) {
        memcpy(x.bg, bg3 + 3 * (size_t)(i - 1), 24); memcpy(x.ba, ba3 + 3 * (size_t)(i - 1), 24);  // the earlier keyframe's (fixed) biases
        gs_edge(Rwb9 + 9 * (size_t)(i - 1), twb3 + 3 * (size_t)(i - 1), &x.v[3 * (size_t)(i - 1)], Rwb9 + 9 * (size_t)i, twb3 + 3 * (size_t)i, &x.v[3 * (size_t)i], x,
                links[i].pre, e, J);
    };
    auto robust_chi2 = [&] {
        double tot = 0;
        for (int i = 1; i < N; ++i) {
            if (!links[i].pre) continue;
            double e[9], c = 0, r0, r1;
            edge(i, e, nullptr);
            for (int r = 0; r < 9; ++r) for (int q = 0; q < 9; ++q) c += e[r] * links[i].info[9 * r + q] * e[q];
            huber(c, r0, r1);
            tot += r0;
        }
        return tot;
    };
    if (chi2) chi2[0] = robust_chi2();
    int done = 0;
    for (int it = 0; it < iterations; ++it) {
        double H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, b[3] = {0, 0, 0}, u[3];
        for (int i = 1; i < N; ++i) {
            if (!links[i].pre) continue;
            double e[9], J[135], Oe[9], OJ[27], c = 0, r0, r1;
            edge(i, e, J);
            const double* O = links[i].info;
            for (int r = 0; r < 9; ++r) {
                double t = 0; for (int k = 0; k < 9; ++k) t += O[9 * r + k] * e[k];
                Oe[r] = t; c += e[r] * t;
                for (int q = 0; q < 3; ++q) { double w = 0; for (int k = 0; k < 9; ++k) w += O[9 * r + k] * J[15 * k + 12 + q]; OJ[3 * r + q] = w; }
            }
            huber(c, r0, r1);
            for (int a = 0; a < 3; ++a) {
                double g = 0;
                for (int r = 0; r < 9; ++r) g += J[15 * r + 12 + a] * Oe[r];
                b[a] -= r1 * g;
                for (int q = 0; q < 3; ++q) { double h = 0; for (int r = 0; r < 9; ++r) h += J[15 * r + 12 + a] * OJ[3 * r + q]; H[3 * a + q] += r1 * h; }
            }
        }
        if (!ldlt_solve_small(H, 3, b, u, false)) break;
        const double w[3] = {u[0], u[1], 0.0};
        double E[9], R[9];
        exp_so3(w, E);
        r3_mul(x.Rwg, E, R);
        memcpy(x.Rwg, R, sizeof(R));
        x.s *= std::exp(u[2]);
        ++done;
    }
    if (chi2) chi2[1] = robust_chi2();
    memcpy(Rwg9, x.Rwg, 72); *scale = x.s;
    return done;
}

int tc2li_imu_init_gravity(int n_kfs, const float* Rwb9, const float* twb3, const tc2li_preintegrated* const* pre, float* vel3, float Rwg9[9]) {
    if (n_kfs < 2 || !Rwb9 || !twb3 || !pre || !vel3 || !Rwg9) { set_error("tc2li_imu_init_gravity: invalid argument"); return TC2LI_ERR_INVALID; }
    float dirG[3] = {0, 0, 0};
    int used = 0;
    for (int i = 1; i < n_kfs; ++i) {
        if (!pre[i]) continue;
        float dV[3];
        tc2li_imu_delta(pre[i], &pre[i]->bias, nullptr, dV, nullptr);  // GetUpdatedDeltaVelocity before any bias update: dV
        const float* R = Rwb9 + 9 * (size_t)(i - 1);
        for (int r = 0; r < 3; ++r) dirG[r] -= (R[3 * r] * dV[0] + R[3 * r + 1] * dV[1]) + R[3 * r + 2] * dV[2];
        for (int k = 0; k < 3; ++k) {
            const float v = (twb3[3 * (size_t)i + k] - twb3[3 * (size_t)(i - 1) + k]) / pre[i]->dT;
            vel3[3 * (size_t)i + k] = v; vel3[3 * (size_t)(i - 1) + k] = v;
        }
        ++used;
    }
    if (!used) { set_error("tc2li_imu_init_gravity: no pre-integrated link"); return TC2LI_ERR_INVALID; }
    const float nrm = std::sqrt((dirG[0] * dirG[0] + dirG[1] * dirG[1]) + dirG[2] * dirG[2]);
    for (int k = 0; k < 3; ++k) dirG[k] = dirG[k] / nrm;
    const float gI[3] = {0.0f, 0.0f, -1.0f};
    const float v[3] = {gI[1] * dirG[2] - gI[2] * dirG[1], gI[2] * dirG[0] - gI[0] * dirG[2], gI[0] * dirG[1] - gI[1] * dirG[0]};
    const float nv = std::sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
    const float cosg = (gI[0] * dirG[0] + gI[1] * dirG[1]) + gI[2] * dirG[2];
    const float ang = std::acos(cosg);
    // Sophus::SO3f::exp(v * ang / nv).matrix(): Rodrigues in double, rounded (the pre-integration's convention, imu_host.cpp)
    const double w[3] = {(double)(v[0] * ang / nv), (double)(v[1] * ang / nv), (double)(v[2] * ang / nv)};
    const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], th = std::sqrt(th2);
    double W[9], W2[9];
    hat3(w, W);
    r3_mul(W, W, W2);
    const double a = th < 1e-8 ? 1.0 - th2 / 6 : std::sin(th) / th, bq = th < 1e-8 ? 0.5 - th2 / 24 : (1 - std::cos(th)) / th2;
    for (int k = 0; k < 9; ++k) Rwg9[k] = (float)((k % 4 == 0 ? 1.0 : 0.0) + a * W[k] + bq * W2[k]);
    return used;
}

}  // extern "C"
