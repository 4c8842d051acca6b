// Host side of tc2li_imu_* (include/tc2li_hip.h): IMU pre-integration between frames / keyframes and the IMU state
// prediction of the tracking thread -- IMU::Preintegrated (SF/src/ImuTypes.cc:152-316), Tracking::PreintegrateIMU
// (SF/src/Tracking.cc:1710-1822) and Tracking::PredictStateIMU (:1825-1875).  About ten samples per frame in float:
// SURVEY.md section 8a row a11 keeps it on the host; the result feeds the inertial edges of the local BA.
#include <atomic>
#include <cmath>
#include <cstring>

#include "common.hpp"

using namespace tc2li;

namespace {

struct M3f {
    float m[9];
    float& operator()(int r, int c) { return m[3 * r + c]; }
    float operator()(int r, int c) const { return m[3 * r + c]; }
};
inline M3f ident() { return M3f{{1, 0, 0, 0, 1, 0, 0, 0, 1}}; }
inline M3f operator*(const M3f& a, const M3f& b) {
    M3f o;
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) o(r, c) = a(r, 0) * b(0, c) + a(r, 1) * b(1, c) + a(r, 2) * b(2, c);
    return o;
}
inline M3f operator*(const M3f& a, float s) { M3f o; for (int k = 0; k < 9; ++k) o.m[k] = a.m[k] * s; return o; }
inline M3f operator/(const M3f& a, float s) { M3f o; for (int k = 0; k < 9; ++k) o.m[k] = a.m[k] / s; return o; }
inline M3f operator+(const M3f& a, const M3f& b) { M3f o; for (int k = 0; k < 9; ++k) o.m[k] = a.m[k] + b.m[k]; return o; }
inline M3f operator-(const M3f& a, const M3f& b) { M3f o; for (int k = 0; k < 9; ++k) o.m[k] = a.m[k] - b.m[k]; return o; }
inline M3f transpose(const M3f& a) { M3f o; for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) o(r, c) = a(c, r); return o; }
inline M3f hat(const float v[3]) { return M3f{{0, -v[2], v[1], v[2], 0, -v[0], -v[1], v[0], 0}}; }
inline void mulv(const M3f& a, const float v[3], float o[3]) { for (int r = 0; r < 3; ++r) o[r] = a(r, 0) * v[0] + a(r, 1) * v[1] + a(r, 2) * v[2]; }

// IMU::NormalizeRotation (ImuTypes.cc:41-44): U V^T of Eigen::JacobiSVD<Matrix3f>(R, ComputeFullU | ComputeFullV).  Eigen's algorithm
// for a square real matrix, step for step (JacobiSVD.h compute(), real_2x2_jacobi_svd; Jacobi.h makeJacobi / apply_rotation_in_the_plane):
// the matrix divided by its largest |entry|; sweeps over the pairs (p, q), q < p, until every off-diagonal pair is below
// 2 eps x the largest diagonal entry met so far; a pair is diagonalised by a left rotation that symmetrises its 2 x 2 block times a
// Jacobi rotation, U and V accumulate the rotations; negative diagonal entries flip their column of U; the singular values are
// brought into descending order with their columns.  (Rounds 1-3 used one-sided Jacobi rotations: the same matrix to a few ulps.)
struct JRot { float c, s; };
inline JRot jrot_T(const JRot& j) { return JRot{j.c, -j.s}; }
inline void rotate_rows(M3f& m, int p, int q, const JRot& j) {  // applyOnTheLeft(p, q, j)
    if (j.c == 1.0f && j.s == 0.0f) return;
    for (int i = 0; i < 3; ++i) {
        const float x = m(p, i), y = m(q, i);
        m(p, i) = j.c * x + j.s * y;
        m(q, i) = -j.s * x + j.c * y;
    }
}
inline void rotate_cols(M3f& m, int p, int q, const JRot& j_right) {  // applyOnTheRight(p, q, j): the plane rotation j^T on the columns
    const JRot j = jrot_T(j_right);
    if (j.c == 1.0f && j.s == 0.0f) return;
    for (int i = 0; i < 3; ++i) {
        const float x = m(i, p), y = m(i, q);
        m(i, p) = j.c * x + j.s * y;
        m(i, q) = -j.s * x + j.c * y;
    }
}
M3f normalize_rotation(const M3f& R) {
    const float eps2 = 2.0f * 1.1920929e-7f, tiny = 1.17549435e-38f;  // 2 * NumTraits<float>::epsilon(), numeric_limits<float>::min()
    float scale = 0;
    for (int k = 0; k < 9; ++k) scale = fmaxf(scale, fabsf(R.m[k]));
    if (scale == 0.0f) scale = 1.0f;
    M3f W, U = ident(), V = ident();
    for (int k = 0; k < 9; ++k) W.m[k] = R.m[k] / scale;
    float max_diag = fmaxf(fabsf(W(0, 0)), fmaxf(fabsf(W(1, 1)), fabsf(W(2, 2))));
    bool finished = false;
    while (!finished) {
        finished = true;
        for (int p = 1; p < 3; ++p)
            for (int q = 0; q < p; ++q) {
                const float threshold = fmaxf(tiny, eps2 * max_diag);
                if (!(fabsf(W(p, q)) > threshold || fabsf(W(q, p)) > threshold)) continue;
                finished = false;
                float a = W(p, p), b = W(p, q), c = W(q, p), d = W(q, q);
                JRot r1{1.0f, 0.0f};
                const float t = a + d, df = c - b;
                if (!(fabsf(df) < tiny)) {
                    const float u = t / df, tmp = sqrtf(1.0f + u * u);
                    r1.s = 1.0f / tmp;
                    r1.c = u / tmp;
                    const float a0 = a, b0 = b, c0 = c, d0 = d;
                    a = r1.c * a0 + r1.s * c0; b = r1.c * b0 + r1.s * d0;
                    c = -r1.s * a0 + r1.c * c0; d = -r1.s * b0 + r1.c * d0;
                }
                JRot jr{1.0f, 0.0f};  // makeJacobi(a, b, d)
                const float deno = 2.0f * fabsf(b);
                if (!(deno < tiny)) {
                    const float tau = (a - d) / deno, w = sqrtf(tau * tau + 1.0f);
                    const float tt = tau > 0.0f ? 1.0f / (tau + w) : 1.0f / (tau - w);
                    const float sign_t = tt > 0.0f ? 1.0f : -1.0f, n = 1.0f / sqrtf(tt * tt + 1.0f);
                    jr.s = -sign_t * (b / fabsf(b)) * fabsf(tt) * n;
                    jr.c = n;
                }
                const JRot jrt = jrot_T(jr);
                const JRot jl{r1.c * jrt.c - r1.s * jrt.s, r1.c * jrt.s + r1.s * jrt.c};  // rot1 * j_right^T
                rotate_rows(W, p, q, jl);
                rotate_cols(U, p, q, jrot_T(jl));
                rotate_cols(W, p, q, jr);
                rotate_cols(V, p, q, jr);
                max_diag = fmaxf(max_diag, fmaxf(fabsf(W(p, p)), fabsf(W(q, q))));
            }
    }
    float sv[3];
    for (int i = 0; i < 3; ++i) {
        const float a = W(i, i);
        sv[i] = fabsf(a);
        if (a < 0.0f) for (int r = 0; r < 3; ++r) U(r, i) = -U(r, i);
    }
    for (int i = 0; i < 3; ++i) {
        int pos = i;
        for (int k = i + 1; k < 3; ++k) if (sv[k] > sv[pos]) pos = k;
        if (sv[pos] == 0.0f) break;
        if (pos != i) {
            const float ts = sv[i]; sv[i] = sv[pos]; sv[pos] = ts;
            for (int r = 0; r < 3; ++r) {
                const float tu = U(r, i); U(r, i) = U(r, pos); U(r, pos) = tu;
                const float tv = V(r, i); V(r, i) = V(r, pos); V(r, pos) = tv;
            }
        }
    }
    return U * transpose(V);
}

M3f so3_exp(const float v[3]) {  // Sophus::SO3f::exp(v).matrix(): unit quaternion, then its rotation matrix
    const float th2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    float imag, real;
    if (th2 < 1e-10f * 1e-10f) {
        const float th4 = th2 * th2;
        imag = 0.5f - (1.0f / 48.0f) * th2 + (1.0f / 3840.0f) * th4;
        real = 1.0f - (1.0f / 8.0f) * th2 + (1.0f / 384.0f) * th4;
    } else {
        const float th = sqrtf(th2), half = 0.5f * th;
        imag = sinf(half) / th;
        real = cosf(half);
    }
    const float q[4] = {imag * v[0], imag * v[1], imag * v[2], real};
    const float tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
    const float twx = tx * q[3], twy = ty * q[3], twz = tz * q[3], txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
    const float tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
    return M3f{{1 - (tyy + tzz), txy - twz, txz + twy, txy + twz, 1 - (txx + tzz), tyz - twx, txz - twy, tyz + twx, 1 - (txx + tyy)}};
}

void integrate(tc2li_preintegrated& p, const float acceleration[3], const float ang_vel[3], float dt) {
    const tc2li_imu_bias& b = p.bias;
    const float acc[3] = {acceleration[0] - b.bax, acceleration[1] - b.bay, acceleration[2] - b.baz};
    const float accW[3] = {ang_vel[0] - b.bwx, ang_vel[1] - b.bwy, ang_vel[2] - b.bwz};
    M3f dR, JRg, JVg, JVa, JPg, JPa;
    memcpy(dR.m, p.dR, 36); memcpy(JRg.m, p.JRg, 36); memcpy(JVg.m, p.JVg, 36); memcpy(JVa.m, p.JVa, 36);
    memcpy(JPg.m, p.JPg, 36); memcpy(JPa.m, p.JPa, 36);
    float Ra[3];
    mulv(dR, acc, Ra);
    for (int k = 0; k < 3; ++k) {
        p.avgA[k] = (p.dT * p.avgA[k] + Ra[k] * dt) / (p.dT + dt);
        p.avgW[k] = (p.dT * p.avgW[k] + accW[k] * dt) / (p.dT + dt);
    }
    // position first (old velocity and rotation), then velocity (old rotation), rotation last
    for (int k = 0; k < 3; ++k) p.dP[k] = p.dP[k] + p.dV[k] * dt + 0.5f * Ra[k] * dt * dt;
    for (int k = 0; k < 3; ++k) p.dV[k] = p.dV[k] + Ra[k] * dt;
    // The reference's expressions in Eigen's order of evaluation (ImuTypes.cc:213-225): a scalar factor scales the matrix it stands next to,
    // element by element, before the next product -- `-dR*dt*Wacc` is ((-dR) dt) Wacc, `0.5f*dR*dt*dt*Wacc*JRg` is ((((0.5 dR) dt) dt) Wacc) JRg.
    // (Round 2 formed (dR Wacc) first and scaled afterwards: the same numbers to the last bit or two of a float.)
    const M3f Wacc = hat(acc);
    const M3f dRdt = dR * dt, dRdt2h = (dRdt * dt) * 0.5f;  // the factor 0.5 is exact wherever it is applied
    const M3f RWdt = dRdt * Wacc, RWdt2h = dRdt2h * Wacc;
    float A[81] = {0}, B[54] = {0};
    for (int k = 0; k < 9; ++k) A[10 * k] = 1;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            A[9 * (3 + r) + c] = -RWdt(r, c);
            A[9 * (6 + r) + c] = -RWdt2h(r, c);
            A[9 * (6 + r) + 3 + c] = r == c ? dt : 0.0f;
            B[6 * (3 + r) + 3 + c] = dRdt(r, c);
            B[6 * (6 + r) + 3 + c] = dRdt2h(r, c);
        }
    JPa = JPa + JVa * dt - dRdt2h;
    JPg = JPg + JVg * dt - RWdt2h * JRg;
    JVa = JVa - dRdt;
    JVg = JVg - RWdt * JRg;
    // IntegratedRotation (ImuTypes.cc:95-116)
    const float v[3] = {accW[0] * dt, accW[1] * dt, accW[2] * dt};
    const float d2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2], d = sqrtf(d2);
    const M3f W = hat(v);
    M3f deltaR, rightJ;
    if (d < 1e-4f) {
        deltaR = ident() + W;
        rightJ = ident();
    } else {
        // Eigen evaluates `W*sin(d)/d` as (W * sin d) / d and `W*W*(1.0f-cos(d))/d2` as ((W W) (1 - cos d)) / d2 (ImuTypes.cc:111-112): the
        // scalar multiplies the matrix before the division (round 3 scaled by the quotient: the last bit of an entry could differ)
        const M3f W2 = W * W;
        deltaR = ident() + (W * sinf(d)) / d + (W2 * (1.0f - cosf(d))) / d2;
        rightJ = ident() - (W * (1.0f - cosf(d))) / d2 + (W2 * (d - sinf(d))) / (d2 * d);
    }
    dR = normalize_rotation(dR * deltaR);
    const M3f dRiT = transpose(deltaR);
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) { A[9 * r + c] = dRiT(r, c); B[6 * r + c] = rightJ(r, c) * dt; }
    // covariance: C(0:9, 0:9) = A C A^T + B Nga B^T, C(9:15, 9:15) += NgaWalk
    float AC[81], Cn[81];
    for (int r = 0; r < 9; ++r)
        for (int c = 0; c < 9; ++c) { float s = 0; for (int k = 0; k < 9; ++k) s += A[9 * r + k] * p.C[15 * k + c]; AC[9 * r + c] = s; }
    for (int r = 0; r < 9; ++r)
        for (int c = 0; c < 9; ++c) {
            float s = 0;
            for (int k = 0; k < 9; ++k) s += AC[9 * r + k] * A[9 * c + k];
            float n = 0;
            for (int k = 0; k < 6; ++k) n += B[6 * r + k] * p.noise[k] * B[6 * c + k];
            Cn[9 * r + c] = s + n;
        }
    for (int r = 0; r < 9; ++r) for (int c = 0; c < 9; ++c) p.C[15 * r + c] = Cn[9 * r + c];
    for (int k = 0; k < 6; ++k) p.C[15 * (9 + k) + 9 + k] += p.noise_walk[k];
    JRg = dRiT * JRg - rightJ * dt;
    memcpy(p.dR, dR.m, 36); memcpy(p.JRg, JRg.m, 36); memcpy(p.JVg, JVg.m, 36); memcpy(p.JVa, JVa.m, 36);
    memcpy(p.JPg, JPg.m, 36); memcpy(p.JPa, JPa.m, 36);
    p.dT += dt;
    p.n_measurements++;
}

void delta_rotation(const tc2li_preintegrated& p, const tc2li_imu_bias& b_, M3f& out) {
    const float dbg[3] = {b_.bwx - p.bias.bwx, b_.bwy - p.bias.bwy, b_.bwz - p.bias.bwz};
    M3f J, R;
    memcpy(J.m, p.JRg, 36); memcpy(R.m, p.dR, 36);
    float v[3];
    mulv(J, dbg, v);
    out = normalize_rotation(R * so3_exp(v));
}

}  // namespace

extern "C" {

int tc2li_imu_preintegrated_init(tc2li_preintegrated* p, const tc2li_imu_bias* bias, float ng, float na, float ngw, float naw) {
    if (!p || !bias) { set_error("tc2li_imu_preintegrated_init: invalid argument"); return TC2LI_ERR_INVALID; }
    memset(p, 0, sizeof(*p));
    p->dR[0] = p->dR[4] = p->dR[8] = 1.0f;
    p->bias = *bias;
    const float ng2 = ng * ng, na2 = na * na, ngw2 = ngw * ngw, naw2 = naw * naw;
    for (int k = 0; k < 3; ++k) { p->noise[k] = ng2; p->noise[3 + k] = na2; p->noise_walk[k] = ngw2; p->noise_walk[3 + k] = naw2; }
    return TC2LI_OK;
}

int tc2li_imu_integrate(tc2li_preintegrated* p, const float acc[3], const float ang_vel[3], float dt) {
    if (!p || !acc || !ang_vel) { set_error("tc2li_imu_integrate: invalid argument"); return TC2LI_ERR_INVALID; }
    integrate(*p, acc, ang_vel, dt);
    return TC2LI_OK;
}

int tc2li_imu_preintegrate(tc2li_preintegrated* p, const tc2li_imu_sample* m, int n_samples, double t_prev, double t_cur) {
    if (!p || n_samples < 0 || (n_samples > 0 && !m)) { set_error("tc2li_imu_preintegrate: invalid argument"); return TC2LI_ERR_INVALID; }
    const int n = n_samples - 1;
    if (n <= 0) return 0;
    for (int i = 0; i < n; i++) {
        float tstep = 0, acc[3] = {0, 0, 0}, w[3] = {0, 0, 0};
        if (i == 0 && i < n - 1) {
            const float tab = (float)(m[i + 1].t - m[i].t), tini = (float)(m[i].t - t_prev);
            for (int k = 0; k < 3; ++k) {
                acc[k] = (m[i].a[k] + m[i + 1].a[k] - (m[i + 1].a[k] - m[i].a[k]) * (tini / tab)) * 0.5f;
                w[k] = (m[i].w[k] + m[i + 1].w[k] - (m[i + 1].w[k] - m[i].w[k]) * (tini / tab)) * 0.5f;
            }
            tstep = (float)(m[i + 1].t - t_prev);
        } else if (i < n - 1) {
            for (int k = 0; k < 3; ++k) { acc[k] = (m[i].a[k] + m[i + 1].a[k]) * 0.5f; w[k] = (m[i].w[k] + m[i + 1].w[k]) * 0.5f; }
            tstep = (float)(m[i + 1].t - m[i].t);
        } else if (i > 0 && i == n - 1) {
            const float tab = (float)(m[i + 1].t - m[i].t), tend = (float)(m[i + 1].t - t_cur);
            for (int k = 0; k < 3; ++k) {
                acc[k] = (m[i].a[k] + m[i + 1].a[k] - (m[i + 1].a[k] - m[i].a[k]) * (tend / tab)) * 0.5f;
                w[k] = (m[i].w[k] + m[i + 1].w[k] - (m[i + 1].w[k] - m[i].w[k]) * (tend / tab)) * 0.5f;
            }
            tstep = (float)(t_cur - m[i].t);
        } else if (i == 0 && i == n - 1) {
            for (int k = 0; k < 3; ++k) { acc[k] = m[i].a[k]; w[k] = m[i].w[k]; }
            tstep = (float)(t_cur - t_prev);
        }
        integrate(*p, acc, w, tstep);
    }
    return n;
}

int tc2li_imu_delta(const tc2li_preintegrated* p, const tc2li_imu_bias* bias, float dR[9], float dV[3], float dP[3]) {
    if (!p || !bias) { set_error("tc2li_imu_delta: invalid argument"); return TC2LI_ERR_INVALID; }
    const float dbg[3] = {bias->bwx - p->bias.bwx, bias->bwy - p->bias.bwy, bias->bwz - p->bias.bwz};
    const float dba[3] = {bias->bax - p->bias.bax, bias->bay - p->bias.bay, bias->baz - p->bias.baz};
    if (dR) { M3f R; delta_rotation(*p, *bias, R); memcpy(dR, R.m, 36); }
    for (int r = 0; r < 3; ++r) {
        if (dV) dV[r] = p->dV[r] + (p->JVg[3 * r] * dbg[0] + p->JVg[3 * r + 1] * dbg[1] + p->JVg[3 * r + 2] * dbg[2]) +
                        (p->JVa[3 * r] * dba[0] + p->JVa[3 * r + 1] * dba[1] + p->JVa[3 * r + 2] * dba[2]);
        if (dP) dP[r] = p->dP[r] + (p->JPg[3 * r] * dbg[0] + p->JPg[3 * r + 1] * dbg[1] + p->JPg[3 * r + 2] * dbg[2]) +
                        (p->JPa[3 * r] * dba[0] + p->JPa[3 * r + 1] * dba[1] + p->JPa[3 * r + 2] * dba[2]);
    }
    return TC2LI_OK;
}

int tc2li_imu_predict_state(const tc2li_preintegrated* p, const tc2li_imu_bias* bias, const float Rwb1[9], const float twb1[3],
                            const float Vwb1[3], float Rwb2[9], float twb2[3], float Vwb2[3]) {
    if (!p || !bias || !Rwb1 || !twb1 || !Vwb1 || !Rwb2 || !twb2 || !Vwb2) { set_error("tc2li_imu_predict_state: invalid argument"); return TC2LI_ERR_INVALID; }
    const float Gz[3] = {0, 0, -9.81f};  // IMU::GRAVITY_VALUE
    const float t12 = p->dT;
    float dR[9], dV[3], dP[3], RdP[3], RdV[3];
    tc2li_imu_delta(p, bias, dR, dV, dP);
    M3f R1, D;
    memcpy(R1.m, Rwb1, 36); memcpy(D.m, dR, 36);
    const M3f R2 = normalize_rotation(R1 * D);
    memcpy(Rwb2, R2.m, 36);
    mulv(R1, dP, RdP);
    mulv(R1, dV, RdV);
    for (int k = 0; k < 3; ++k) {
        twb2[k] = twb1[k] + Vwb1[k] * t12 + 0.5f * t12 * t12 * Gz[k] + RdP[k];
        Vwb2[k] = Vwb1[k] + t12 * Gz[k] + RdV[k];
    }
    return TC2LI_OK;
}

// Tracking::PreintegrateIMU for the frames of a batch of sequences: per frame a fresh Preintegrated at the frame's bias, then the loop
// over its samples (tc2li_imu_preintegrated_init + tc2li_imu_preintegrate), the frames dealt over host threads.
int tc2li_imu_preintegrate_frames(int n_frames, tc2li_preintegrated* pre, const tc2li_imu_bias* bias, float ng, float na, float ngw, float naw,
                                  const tc2li_imu_sample* samples, const int32_t* sample_offsets, const double* t_prev, const double* t_cur) {
    if (n_frames < 0 || (n_frames > 0 && (!pre || !bias || !samples || !sample_offsets || !t_prev || !t_cur))) {
        set_error("tc2li_imu_preintegrate_frames: invalid argument");
        return TC2LI_ERR_INVALID;
    }
    for (int f = 0; f < n_frames; ++f)
        if (sample_offsets[f + 1] < sample_offsets[f]) { set_error("tc2li_imu_preintegrate_frames: sample offsets must be non-decreasing"); return TC2LI_ERR_INVALID; }
    std::atomic<int> bad{0};
    tracking_pool().parallel_for(n_frames, [&](int f) {
        if (tc2li_imu_preintegrated_init(&pre[f], &bias[f], ng, na, ngw, naw) < 0 ||
            tc2li_imu_preintegrate(&pre[f], samples + sample_offsets[f], sample_offsets[f + 1] - sample_offsets[f], t_prev[f], t_cur[f]) < 0) bad++;
    });
    return bad.load() ? TC2LI_ERR_INVALID : n_frames;
}

}  // extern "C"
