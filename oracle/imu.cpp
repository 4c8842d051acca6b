// TEST INFRASTRUCTURE ONLY -- CPU oracle, never linked into or called from the product path.  See imu.hpp.
#include "imu.hpp"

#include <cmath>
#include <cstring>

namespace oracle {

namespace {
typedef double M3[9];
void mul(const double* a, const double* b, double* o) { for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) o[3 * r + c] = a[3 * r] * b[c] + a[3 * r + 1] * b[3 + c] + a[3 * r + 2] * b[6 + c]; }
void mulv(const double* a, const double* v, double* o) { for (int r = 0; r < 3; ++r) o[r] = a[3 * r] * v[0] + a[3 * r + 1] * v[1] + a[3 * r + 2] * v[2]; }
void tr(const double* a, double* o) { for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) o[3 * r + c] = a[3 * c + r]; }
void hat(const double* v, double* o) { o[0] = 0; o[1] = -v[2]; o[2] = v[1]; o[3] = v[2]; o[4] = 0; o[5] = -v[0]; o[6] = -v[1]; o[7] = v[0]; o[8] = 0; }
void to_d(const float* f, double* d, int n) { for (int i = 0; i < n; ++i) d[i] = f[i]; }
void to_f(const double* d, float* f, int n) { for (int i = 0; i < n; ++i) f[i] = (float)d[i]; }
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
// U V^T of the SVD of R = the orthogonal polar factor (Higham's Newton iteration X <- (X + X^-T) / 2)
void polar(const double* R, double* X) {
    std::memcpy(X, R, 9 * sizeof(double));
    for (int it = 0; it < 50; ++it) {
        double Xi[9], XiT[9], N[9];
        if (!inv3(X, Xi)) break;
        tr(Xi, XiT);
        double diff = 0;
        for (int k = 0; k < 9; ++k) { N[k] = 0.5 * (X[k] + XiT[k]); diff = std::fmax(diff, std::fabs(N[k] - X[k])); }
        std::memcpy(X, N, sizeof(N));
        if (diff < 1e-15) break;
    }
}
// IntegratedRotation (ImuTypes.cc:95-116): deltaR and rightJ of (w - bg) * dt, float thresholds
void integrated_rotation(const float w[3], const ImuBias& b, float dt, double* deltaR, double* rightJ) {
    const float x = (w[0] - b.bwx) * dt, y = (w[1] - b.bwy) * dt, z = (w[2] - b.bwz) * dt;
    const float d2 = x * x + y * y + z * z, d = std::sqrt(d2);
    const double v[3] = {x, y, z};
    double W[9], W2[9];
    hat(v, W);
    mul(W, W, W2);
    const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (d < 1e-4f) {
        for (int k = 0; k < 9; ++k) { deltaR[k] = I[k] + W[k]; rightJ[k] = I[k]; }
    } else {
        const double dd = d, dd2 = d2;
        for (int k = 0; k < 9; ++k) {
            deltaR[k] = I[k] + W[k] * std::sin(dd) / dd + W2[k] * (1.0 - std::cos(dd)) / dd2;
            rightJ[k] = I[k] - W[k] * (1.0 - std::cos(dd)) / dd2 + W2[k] * (dd - std::sin(dd)) / (dd2 * dd);
        }
    }
}
void so3_exp(const double* v, double* R) {  // Sophus::SO3f::exp(v).matrix()
    const double th2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2], th = std::sqrt(th2);
    double W[9], W2[9];
    hat(v, W);
    mul(W, W, W2);
    const double a = th < 1e-8 ? 1.0 - th2 / 6 : std::sin(th) / th, bq = th < 1e-8 ? 0.5 - th2 / 24 : (1 - std::cos(th)) / th2;
    for (int k = 0; k < 9; ++k) R[k] = (k % 4 == 0 ? 1.0 : 0.0) + a * W[k] + bq * W2[k];
}
}  // namespace

void NormalizeRotation(const float R[9], float out[9]) {
    double Rd[9], X[9];
    to_d(R, Rd, 9);
    polar(Rd, X);
    to_f(X, out, 9);
}

Preintegrated::Preintegrated(const ImuBias& b_, float ng, float na, float ngw, float naw) : b(b_) {
    const float ng2 = ng * ng, na2 = na * na, ngw2 = ngw * ngw, naw2 = naw * naw;
    for (int k = 0; k < 3; ++k) { Nga[k] = ng2; Nga[3 + k] = na2; NgaWalk[k] = ngw2; NgaWalk[3 + k] = naw2; }
    const float I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    std::memcpy(dR, I, sizeof(I));
    std::memset(dV, 0, sizeof(dV)); std::memset(dP, 0, sizeof(dP));
    std::memset(JRg, 0, sizeof(JRg)); std::memset(JVg, 0, sizeof(JVg)); std::memset(JVa, 0, sizeof(JVa));
    std::memset(JPg, 0, sizeof(JPg)); std::memset(JPa, 0, sizeof(JPa));
    std::memset(avgA, 0, sizeof(avgA)); std::memset(avgW, 0, sizeof(avgW));
    std::memset(C, 0, sizeof(C));
}

void Preintegrated::IntegrateNewMeasurement(const float acceleration[3], const float angVel[3], float dt_) {
    ++n_measurements;
    const double dt = dt_;
    const float accf[3] = {acceleration[0] - b.bax, acceleration[1] - b.bay, acceleration[2] - b.baz};
    const float accWf[3] = {angVel[0] - b.bwx, angVel[1] - b.bwy, angVel[2] - b.bwz};
    double acc[3] = {accf[0], accf[1], accf[2]}, accW[3] = {accWf[0], accWf[1], accWf[2]};
    double R[9], V[3], P[3], jrg[9], jvg[9], jva[9], jpg[9], jpa[9];
    to_d(dR, R, 9); to_d(dV, V, 3); to_d(dP, P, 3); to_d(JRg, jrg, 9); to_d(JVg, jvg, 9); to_d(JVa, jva, 9); to_d(JPg, jpg, 9); to_d(JPa, jpa, 9);
    double Ra[3];
    mulv(R, acc, Ra);
    const double T = dT;
    for (int k = 0; k < 3; ++k) {
        avgA[k] = (float)((T * avgA[k] + Ra[k] * dt) / (T + dt));
        avgW[k] = (float)((T * avgW[k] + accW[k] * dt) / (T + dt));
    }
    for (int k = 0; k < 3; ++k) { P[k] = P[k] + V[k] * dt + 0.5 * Ra[k] * dt * dt; }
    for (int k = 0; k < 3; ++k) { V[k] = V[k] + Ra[k] * dt; }
    double Wacc[9], RW[9], RWJ[9];
    hat(acc, Wacc);
    mul(R, Wacc, RW);
    mul(RW, jrg, RWJ);
    double A[81] = {0}, B[54] = {0};
    for (int k = 0; k < 9; ++k) A[10 * k] = 1;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            A[9 * (3 + r) + c] = -RW[3 * r + c] * dt;
            A[9 * (6 + r) + c] = -0.5 * RW[3 * r + c] * dt * dt;
            A[9 * (6 + r) + 3 + c] = r == c ? dt : 0.0;
            B[6 * (3 + r) + 3 + c] = R[3 * r + c] * dt;
            B[6 * (6 + r) + 3 + c] = 0.5 * R[3 * r + c] * dt * dt;
        }
    for (int k = 0; k < 9; ++k) {
        jpa[k] = jpa[k] + jva[k] * dt - 0.5 * R[k] * dt * dt;
        jpg[k] = jpg[k] + jvg[k] * dt - 0.5 * RWJ[k] * dt * dt;
    }
    for (int k = 0; k < 9; ++k) {
        jva[k] = jva[k] - R[k] * dt;
        jvg[k] = jvg[k] - RWJ[k] * dt;
    }
    double dRi[9], rJ[9], Rn[9], Rp[9], dRiT[9], t1[9];
    integrated_rotation(angVel, b, dt_, dRi, rJ);
    mul(R, dRi, Rn);
    polar(Rn, Rp);
    tr(dRi, dRiT);
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) { A[9 * r + c] = dRiT[3 * r + c]; B[6 * r + c] = rJ[3 * r + c] * dt; }
    // C.block<9,9>(0,0) = A C A^T + B Nga B^T ; C.block<6,6>(9,9) += NgaWalk
    double Cd[81], AC[81], Cn[81];
    for (int r = 0; r < 9; ++r) for (int c = 0; c < 9; ++c) Cd[9 * r + c] = C[15 * r + c];
    for (int r = 0; r < 9; ++r) for (int c = 0; c < 9; ++c) { double s = 0; for (int k = 0; k < 9; ++k) s += A[9 * r + k] * Cd[9 * k + c]; AC[9 * r + c] = s; }
    for (int r = 0; r < 9; ++r)
        for (int c = 0; c < 9; ++c) {
            double s = 0;
            for (int k = 0; k < 9; ++k) s += AC[9 * r + k] * A[9 * c + k];
            for (int k = 0; k < 6; ++k) s += B[6 * r + k] * (double)Nga[k] * B[6 * c + k];
            Cn[9 * r + c] = s;
        }
    for (int r = 0; r < 9; ++r) for (int c = 0; c < 9; ++c) C[15 * r + c] = (float)Cn[9 * r + c];
    for (int k = 0; k < 6; ++k) C[15 * (9 + k) + 9 + k] += NgaWalk[k];
    mul(dRiT, jrg, t1);
    for (int k = 0; k < 9; ++k) jrg[k] = t1[k] - rJ[k] * dt;
    to_f(Rp, dR, 9); to_f(V, dV, 3); to_f(P, dP, 3); to_f(jrg, JRg, 9); to_f(jvg, JVg, 9); to_f(jva, JVa, 9); to_f(jpg, JPg, 9); to_f(jpa, JPa, 9);
    dT += dt_;
}

void Preintegrated::GetDeltaRotation(const ImuBias& b_, float out[9]) const {
    const double dbg[3] = {(double)(b_.bwx - b.bwx), (double)(b_.bwy - b.bwy), (double)(b_.bwz - b.bwz)};
    double J[9], v[3], E[9], R[9], RE[9], X[9];
    to_d(JRg, J, 9); to_d(dR, R, 9);
    mulv(J, dbg, v);
    so3_exp(v, E);
    mul(R, E, RE);
    polar(RE, X);
    to_f(X, out, 9);
}
void Preintegrated::GetDeltaVelocity(const ImuBias& b_, float out[3]) const {
    const double dbg[3] = {(double)(b_.bwx - b.bwx), (double)(b_.bwy - b.bwy), (double)(b_.bwz - b.bwz)};
    const double dba[3] = {(double)(b_.bax - b.bax), (double)(b_.bay - b.bay), (double)(b_.baz - b.baz)};
    for (int r = 0; r < 3; ++r) {
        double s = dV[r];
        for (int c = 0; c < 3; ++c) s += (double)JVg[3 * r + c] * dbg[c] + (double)JVa[3 * r + c] * dba[c];
        out[r] = (float)s;
    }
}
void Preintegrated::GetDeltaPosition(const ImuBias& b_, float out[3]) const {
    const double dbg[3] = {(double)(b_.bwx - b.bwx), (double)(b_.bwy - b.bwy), (double)(b_.bwz - b.bwz)};
    const double dba[3] = {(double)(b_.bax - b.bax), (double)(b_.bay - b.bay), (double)(b_.baz - b.baz)};
    for (int r = 0; r < 3; ++r) {
        double s = dP[r];
        for (int c = 0; c < 3; ++c) s += (double)JPg[3 * r + c] * dbg[c] + (double)JPa[3 * r + c] * dba[c];
        out[r] = (float)s;
    }
}

int PreintegrateIMU(const std::vector<ImuSample>& m, double t_prev, double t_cur, Preintegrated& p) {
    const int n = (int)m.size() - 1;
    if (n <= 0) return 0;
    for (int i = 0; i < n; i++) {
        float tstep = 0, acc[3] = {0, 0, 0}, angVel[3] = {0, 0, 0};
        if (i == 0 && i < n - 1) {
            const float tab = (float)(m[i + 1].t - m[i].t), tini = (float)(m[i].t - t_prev);
            for (int k = 0; k < 3; ++k) {
                acc[k] = (m[i].a[k] + m[i + 1].a[k] - (m[i + 1].a[k] - m[i].a[k]) * (tini / tab)) * 0.5f;
                angVel[k] = (m[i].w[k] + m[i + 1].w[k] - (m[i + 1].w[k] - m[i].w[k]) * (tini / tab)) * 0.5f;
            }
            tstep = (float)(m[i + 1].t - t_prev);
        } else if (i < n - 1) {
            for (int k = 0; k < 3; ++k) { acc[k] = (m[i].a[k] + m[i + 1].a[k]) * 0.5f; angVel[k] = (m[i].w[k] + m[i + 1].w[k]) * 0.5f; }
            tstep = (float)(m[i + 1].t - m[i].t);
        } else if (i > 0 && i == n - 1) {
            const float tab = (float)(m[i + 1].t - m[i].t), tend = (float)(m[i + 1].t - t_cur);
            for (int k = 0; k < 3; ++k) {
                acc[k] = (m[i].a[k] + m[i + 1].a[k] - (m[i + 1].a[k] - m[i].a[k]) * (tend / tab)) * 0.5f;
                angVel[k] = (m[i].w[k] + m[i + 1].w[k] - (m[i + 1].w[k] - m[i].w[k]) * (tend / tab)) * 0.5f;
            }
            tstep = (float)(t_cur - m[i].t);
        } else if (i == 0 && i == n - 1) {
            for (int k = 0; k < 3; ++k) { acc[k] = m[i].a[k]; angVel[k] = m[i].w[k]; }
            tstep = (float)(t_cur - t_prev);
        }
        p.IntegrateNewMeasurement(acc, angVel, tstep);
    }
    return n;
}

void PredictStateIMU(const Preintegrated& p, const ImuBias& b, const float Rwb1[9], const float twb1[3], const float Vwb1[3],
                     float Rwb2[9], float twb2[3], float Vwb2[3]) {
    const double Gz[3] = {0, 0, -(double)9.81f};
    const double t12 = p.dT;
    float dRf[9], dVf[3], dPf[3];
    p.GetDeltaRotation(b, dRf); p.GetDeltaVelocity(b, dVf); p.GetDeltaPosition(b, dPf);
    double R1[9], dRd[9], RR[9], X[9], dVd[3], dPd[3], RdP[3], RdV[3];
    to_d(Rwb1, R1, 9); to_d(dRf, dRd, 9); to_d(dVf, dVd, 3); to_d(dPf, dPd, 3);
    mul(R1, dRd, RR);
    polar(RR, X);
    to_f(X, Rwb2, 9);
    mulv(R1, dPd, RdP);
    mulv(R1, dVd, RdV);
    for (int k = 0; k < 3; ++k) {
        twb2[k] = (float)((double)twb1[k] + (double)Vwb1[k] * t12 + 0.5 * t12 * t12 * Gz[k] + RdP[k]);
        Vwb2[k] = (float)((double)Vwb1[k] + t12 * Gz[k] + RdV[k]);
    }
}

}  // namespace oracle
