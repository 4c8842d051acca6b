// TEST INFRASTRUCTURE ONLY -- CPU oracle, never linked into or called from the product path.  See imu.hpp.
#include "imu.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>

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

// ---- float evaluation: the reference's expressions in float, in Eigen's order of evaluation ------------------
// Every product / scaling below is one Eigen expression node evaluated left to right as the source text associates it
// (`W*sin(d)/d` is (W * sin d) / d; `0.5f*dR*dt*dt*Wacc*JRg` is ((((0.5f dR) dt) dt) Wacc) JRg); a 3 x 3 product's coefficient is
// (a0 b0 + a1 b1) + a2 b2.  Unknowable from here and fixed by choice: the summation order inside Eigen's 9 x 9 products (sequential
// in k), and the SVD of NormalizeRotation, restated from Eigen's Jacobi/SVD sources (two-sided Jacobi sweeps; JacobiSVD.h, Jacobi.h
// of Eigen 3.3) -- "parity unpinned" applies as everywhere in this oracle.
namespace f32 {
struct Mf { float m[9]; float& operator()(int r, int c) { return m[3 * r + c]; } float operator()(int r, int c) const { return m[3 * r + c]; } };
Mf I3() { return Mf{{1, 0, 0, 0, 1, 0, 0, 0, 1}}; }
Mf mulm(const Mf& a, const Mf& b) { Mf o; for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) o(r, c) = (a(r, 0) * b(0, c) + a(r, 1) * b(1, c)) + a(r, 2) * b(2, c); return o; }
Mf scale(const Mf& a, float s) { Mf o; for (int k = 0; k < 9; ++k) o.m[k] = a.m[k] * s; return o; }
Mf div(const Mf& a, float s) { Mf o; for (int k = 0; k < 9; ++k) o.m[k] = a.m[k] / s; return o; }
Mf add(const Mf& a, const Mf& b) { Mf o; for (int k = 0; k < 9; ++k) o.m[k] = a.m[k] + b.m[k]; return o; }
Mf sub(const Mf& a, const Mf& b) { Mf o; for (int k = 0; k < 9; ++k) o.m[k] = a.m[k] - b.m[k]; return o; }
Mf neg(const Mf& a) { Mf o; for (int k = 0; k < 9; ++k) o.m[k] = -a.m[k]; return o; }
Mf trm(const Mf& a) { Mf o; for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) o(r, c) = a(c, r); return o; }
Mf hatm(const float v[3]) { return Mf{{0, -v[2], v[1], v[2], 0, -v[0], -v[1], v[0], 0}}; }
void mulvf(const Mf& a, const float v[3], float o[3]) { for (int r = 0; r < 3; ++r) o[r] = (a(r, 0) * v[0] + a(r, 1) * v[1]) + a(r, 2) * v[2]; }

// Eigen::JacobiRotation<float>
struct Rot { float c, s; };
Rot rot_mul(const Rot& a, const Rot& b) { return Rot{a.c * b.c - a.s * b.s, a.c * b.s + a.s * b.c}; }  // operator* for real scalars
Rot rot_T(const Rot& a) { return Rot{a.c, -a.s}; }
// makeJacobi(x, y, z): the rotation that diagonalises the symmetric 2 x 2 [[x, y], [y, z]]
Rot make_jacobi(float x, float y, float z) {
    const float deno = 2.0f * std::fabs(y);
    if (deno < std::numeric_limits<float>::min()) return Rot{1.0f, 0.0f};
    const float tau = (x - z) / deno, w = std::sqrt(tau * tau + 1.0f);
    const float t = tau > 0.0f ? 1.0f / (tau + w) : 1.0f / (tau - w);
    const float sign_t = t > 0.0f ? 1.0f : -1.0f;
    const float n = 1.0f / std::sqrt(t * t + 1.0f);
    return Rot{n, -sign_t * (y / std::fabs(y)) * std::fabs(t) * n};
}
// apply_rotation_in_the_plane on rows p, q (applyOnTheLeft) / columns p, q with j^T (applyOnTheRight)
void rows(Mf& m, int p, int q, const Rot& j) {
    if (j.c == 1.0f && j.s == 0.0f) return;
    for (int i = 0; i < 3; ++i) { const float x = m(p, i), y = m(q, i); m(p, i) = j.c * x + j.s * y; m(q, i) = -j.s * x + j.c * y; }
}
void cols(Mf& m, int p, int q, const Rot& j_) {
    const Rot j = rot_T(j_);
    if (j.c == 1.0f && j.s == 0.0f) return;
    for (int i = 0; i < 3; ++i) { const float x = m(i, p), y = m(i, q); m(i, p) = j.c * x + j.s * y; m(i, q) = -j.s * x + j.c * y; }
}
// JacobiSVD<Matrix3f>(R, ComputeFullU | ComputeFullV): U V^T
Mf svd_uvT(const Mf& R) {
    const float precision = 2.0f * std::numeric_limits<float>::epsilon(), tiny = std::numeric_limits<float>::min();
    float scl = 0;
    for (int k = 0; k < 9; ++k) scl = std::fmax(scl, std::fabs(R.m[k]));
    if (scl == 0.0f) scl = 1.0f;
    Mf W = div(R, scl), U = I3(), V = I3();
    float max_diag = std::fmax(std::fabs(W(0, 0)), std::fmax(std::fabs(W(1, 1)), std::fabs(W(2, 2))));
    for (bool finished = false; !finished;) {
        finished = true;
        for (int p = 1; p < 3; ++p)
            for (int q = 0; q < p; ++q) {
                const float threshold = std::fmax(tiny, precision * max_diag);
                if (std::fabs(W(p, q)) > threshold || std::fabs(W(q, p)) > threshold) {
                    finished = false;
                    // real_2x2_jacobi_svd: a rotation that makes the 2 x 2 block symmetric, then the Jacobi rotation that diagonalises it
                    float m00 = W(p, p), m01 = W(p, q), m10 = W(q, p), m11 = W(q, q);
                    Rot rot1;
                    const float t = m00 + m11, d = m10 - m01;
                    if (std::fabs(d) < tiny) rot1 = Rot{1.0f, 0.0f};
                    else { const float u = t / d, tmp = std::sqrt(1.0f + u * u); rot1 = Rot{u / tmp, 1.0f / tmp}; }
                    if (!(rot1.c == 1.0f && rot1.s == 0.0f)) {  // m.applyOnTheLeft(0, 1, rot1)
                        const float a0 = m00, a1 = m01, b0 = m10, b1 = m11;
                        m00 = rot1.c * a0 + rot1.s * b0; m01 = rot1.c * a1 + rot1.s * b1;
                        m10 = -rot1.s * a0 + rot1.c * b0; m11 = -rot1.s * a1 + rot1.c * b1;
                    }
                    const Rot j_right = make_jacobi(m00, m01, m11);
                    const Rot j_left = rot_mul(rot1, rot_T(j_right));
                    rows(W, p, q, j_left);
                    cols(U, p, q, rot_T(j_left));
                    cols(W, p, q, j_right);
                    cols(V, p, q, j_right);
                    max_diag = std::fmax(max_diag, std::fmax(std::fabs(W(p, p)), std::fabs(W(q, q))));
                }
            }
    }
    float sv[3];
    for (int i = 0; i < 3; ++i) {
        const float a = W(i, i);
        sv[i] = std::fabs(a);
        if (a < 0.0f) for (int r = 0; r < 3; ++r) U(r, i) = -U(r, i);
    }
    for (int i = 0; i < 3; ++i) {  // singular values in descending order, columns follow
        int pos = i;
        for (int k = i + 1; k < 3; ++k) if (sv[k] > sv[pos]) pos = k;
        if (sv[pos] == 0.0f) break;
        if (pos != i) {
            std::swap(sv[i], sv[pos]);
            for (int r = 0; r < 3; ++r) { std::swap(U(r, i), U(r, pos)); std::swap(V(r, i), V(r, pos)); }
        }
    }
    return mulm(U, trm(V));
}
}  // namespace f32

void NormalizeRotationFloat(const float R[9], float out[9]) {
    f32::Mf m;
    std::memcpy(m.m, R, 36);
    const f32::Mf o = f32::svd_uvT(m);
    std::memcpy(out, o.m, 36);
}

void Preintegrated::IntegrateNewMeasurementFloat(const float acceleration[3], const float angVel[3], float dt) {
    using namespace f32;
    ++n_measurements;
    Mf R, jrg, jvg, jva, jpg, jpa;
    std::memcpy(R.m, dR, 36); std::memcpy(jrg.m, JRg, 36); std::memcpy(jvg.m, JVg, 36); std::memcpy(jva.m, JVa, 36);
    std::memcpy(jpg.m, JPg, 36); std::memcpy(jpa.m, JPa, 36);
    const float acc[3] = {acceleration[0] - b.bax, acceleration[1] - b.bay, acceleration[2] - b.baz};
    const float accW[3] = {angVel[0] - b.bwx, angVel[1] - b.bwy, angVel[2] - b.bwz};
    float Ra[3];
    mulvf(R, acc, Ra);                                              // dR*acc
    for (int k = 0; k < 3; ++k) {                                  // (dT*avg + dR*acc*dt) / (dT + dt)
        avgA[k] = (dT * avgA[k] + Ra[k] * dt) / (dT + dt);
        avgW[k] = (dT * avgW[k] + accW[k] * dt) / (dT + dt);
    }
    const Mf hR = scale(R, 0.5f);                                  // 0.5f*dR
    float hRa[3];
    mulvf(hR, acc, hRa);                                            // 0.5f*dR*acc
    for (int k = 0; k < 3; ++k) dP[k] = (dP[k] + dV[k] * dt) + (hRa[k] * dt) * dt;   // dP + dV*dt + 0.5f*dR*acc*dt*dt
    for (int k = 0; k < 3; ++k) dV[k] = dV[k] + Ra[k] * dt;        // dV + dR*acc*dt
    const Mf Wacc = hatm(acc);
    float A[81] = {0}, B[54] = {0};
    for (int k = 0; k < 9; ++k) A[10 * k] = 1;
    const Mf A30 = mulm(scale(neg(R), dt), Wacc);                   // -dR*dt*Wacc
    const Mf A60 = mulm(scale(scale(scale(R, -0.5f), dt), dt), Wacc);  // -0.5f*dR*dt*dt*Wacc
    const Mf B33 = scale(R, dt);                                   // dR*dt
    const Mf B63 = scale(scale(hR, dt), dt);                       // 0.5f*dR*dt*dt
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            A[9 * (3 + r) + c] = A30(r, c);
            A[9 * (6 + r) + c] = A60(r, c);
            A[9 * (6 + r) + 3 + c] = r == c ? dt : 0.0f;
            B[6 * (3 + r) + 3 + c] = B33(r, c);
            B[6 * (6 + r) + 3 + c] = B63(r, c);
        }
    jpa = sub(add(jpa, scale(jva, dt)), B63);                                      // JPa + JVa*dt - 0.5f*dR*dt*dt
    jpg = sub(add(jpg, scale(jvg, dt)), mulm(mulm(B63, Wacc), jrg));                 // JPg + JVg*dt - 0.5f*dR*dt*dt*Wacc*JRg
    jva = sub(jva, B33);                                                           // JVa - dR*dt
    jvg = sub(jvg, mulm(mulm(B33, Wacc), jrg));                                      // JVg - dR*dt*Wacc*JRg
    // IntegratedRotation dRi(angVel, b, dt)
    const float x = (angVel[0] - b.bwx) * dt, y = (angVel[1] - b.bwy) * dt, z = (angVel[2] - b.bwz) * dt;
    const float d2 = x * x + y * y + z * z, d = std::sqrt(d2);
    const float v[3] = {x, y, z};
    const Mf W = hatm(v);
    Mf deltaR, rightJ;
    if (d < 1e-4f) { deltaR = add(I3(), W); rightJ = I3(); }
    else {
        const Mf WW = mulm(W, W);
        deltaR = add(add(I3(), div(scale(W, std::sin(d)), d)), div(scale(WW, 1.0f - std::cos(d)), d2));      // I + W*sin(d)/d + W*W*(1-cos(d))/d2
        rightJ = add(sub(I3(), div(scale(W, 1.0f - std::cos(d)), d2)), div(scale(WW, d - std::sin(d)), d2 * d));  // I - W*(1-cos(d))/d2 + W*W*(d-sin(d))/(d2*d)
    }
    R = svd_uvT(mulm(R, deltaR));                                   // NormalizeRotation(dR*dRi.deltaR)
    const Mf dRiT = trm(deltaR), rJdt = scale(rightJ, dt);
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) { A[9 * r + c] = dRiT(r, c); B[6 * r + c] = rJdt(r, c); }
    float AC[81], Cn[81];
    for (int r = 0; r < 9; ++r) for (int c = 0; c < 9; ++c) { float s = 0; for (int k = 0; k < 9; ++k) s += A[9 * r + k] * C[15 * k + c]; AC[9 * r + c] = s; }
    for (int r = 0; r < 9; ++r)
        for (int c = 0; c < 9; ++c) {
            float s = 0, n = 0;
            for (int k = 0; k < 9; ++k) s += AC[9 * r + k] * A[9 * c + k];
            for (int k = 0; k < 6; ++k) n += B[6 * r + k] * Nga[k] * B[6 * c + k];
            Cn[9 * r + c] = s + n;
        }
    for (int r = 0; r < 9; ++r) for (int c = 0; c < 9; ++c) C[15 * r + c] = Cn[9 * r + c];
    for (int k = 0; k < 6; ++k) C[15 * (9 + k) + 9 + k] += NgaWalk[k];
    jrg = sub(mulm(dRiT, jrg), rJdt);                               // dRi.deltaR.transpose()*JRg - dRi.rightJ*dt
    std::memcpy(dR, R.m, 36); std::memcpy(JRg, jrg.m, 36); std::memcpy(JVg, jvg.m, 36); std::memcpy(JVa, jva.m, 36);
    std::memcpy(JPg, jpg.m, 36); std::memcpy(JPa, jpa.m, 36);
    dT += dt;
}

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

int PreintegrateIMU(const std::vector<ImuSample>& m, double t_prev, double t_cur, Preintegrated& p, bool float_eval) {
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
        if (float_eval) p.IntegrateNewMeasurementFloat(acc, angVel, tstep);
        else p.IntegrateNewMeasurement(acc, angVel, tstep);
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
