// Double-precision SE(3) / projection-edge arithmetic shared by the optimisation kernels (gfx950) -- the device
// statement of g2o::SE3Quat (Thirdparty/g2o/g2o/types/se3quat.h), the stereo / monocular projection edges
// (types/types_six_dof_expmap.cpp:190-274,339-404; SF/src/OptimizableTypes.cpp:58-72,148-169) and the Huber kernel
// (core/robust_kernel_impl.cpp:65-91).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tc2li {

struct Se3 { double q[4], t[3]; };  // q = (x, y, z, w); Tcw
struct CameraD { double fx, fy, cx, cy, bf; };
struct BaEdge {  // tc2li_ba_edge
    int32_t point, pose;
    double u, v, ur, info;
};

__host__ __device__ inline void quat_rotate(const double q[4], const double v[3], double out[3]) {
    double uv[3] = {q[1] * v[2] - q[2] * v[1], q[2] * v[0] - q[0] * v[2], q[0] * v[1] - q[1] * v[0]};
    uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
    out[0] = v[0] + q[3] * uv[0] + (q[1] * uv[2] - q[2] * uv[1]);
    out[1] = v[1] + q[3] * uv[1] + (q[2] * uv[0] - q[0] * uv[2]);
    out[2] = v[2] + q[3] * uv[2] + (q[0] * uv[1] - q[1] * uv[0]);
}
__host__ __device__ inline void se3_map(const Se3& T, const double X[3], double out[3]) {
    quat_rotate(T.q, X, out);
    out[0] += T.t[0]; out[1] += T.t[1]; out[2] += T.t[2];
}
__host__ __device__ inline void normalize_rotation(double q[4]) {
    if (q[3] < 0) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}
__host__ __device__ inline void quat_to_matrix(const double q[4], double R[9]) {
    const double tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
    const double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
    const double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
    const double tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
__host__ __device__ inline void matrix_to_quat(const double R[9], double q[4]) {
    double t = R[0] + R[4] + R[8];
    if (t > 0) {
        t = sqrt(t + 1.0);
        q[3] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (R[7] - R[5]) * t; q[1] = (R[2] - R[6]) * t; q[2] = (R[3] - R[1]) * t;
    } else {
        // the largest diagonal element decides the case; each case with constant indices (the same operations in the same order as the
        // indexed form -- which kept R and q in scratch memory on the device: every access a trip to memory beside the other stages' kernels)
        int i = 0;
        if (R[4] > R[0]) i = 1;
        if (R[8] > (i == 1 ? R[4] : R[0])) i = 2;
#define TC2LI_QUAT_CASE(I, J, K)                                          \
    {                                                                     \
        t = sqrt(R[4 * I] - R[4 * J] - R[4 * K] + 1.0);                     \
        const double qi = 0.5 * t;                                        \
        t = 0.5 / t;                                                     \
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
}
// SE3Quat::exp(update) * T  (VertexSE3Expmap::oplusImpl, types_six_dof_expmap.h:73-76)
__host__ __device__ inline Se3 se3_exp_mul(const double u[6], const Se3& T) {
    const double w0 = u[0], w1 = u[1], w2 = u[2];
    const double theta = sqrt(w0 * w0 + w1 * w1 + w2 * w2);
    const double O[9] = {0, -w2, w1, w2, 0, -w0, -w1, w0, 0};
    double O2[9], R[9], V[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) O2[3 * r + c] = O[3 * r] * O[c] + O[3 * r + 1] * O[3 + c] + O[3 * r + 2] * O[6 + c];
    if (theta < 0.00001) {
        for (int i = 0; i < 9; ++i) { R[i] = (i % 4 == 0 ? 1.0 : 0.0) + O[i] + O2[i]; V[i] = R[i]; }
    } else {
        const double a = sin(theta) / theta, b = (1 - cos(theta)) / (theta * theta), c = (theta - sin(theta)) / (theta * theta * theta);
        for (int i = 0; i < 9; ++i) {
            R[i] = (i % 4 == 0 ? 1.0 : 0.0) + a * O[i] + b * O2[i];
            V[i] = (i % 4 == 0 ? 1.0 : 0.0) + b * O[i] + c * O2[i];
        }
    }
    Se3 E;
    matrix_to_quat(R, E.q);
    for (int r = 0; r < 3; ++r) E.t[r] = V[3 * r] * u[3] + V[3 * r + 1] * u[4] + V[3 * r + 2] * u[5];
    normalize_rotation(E.q);
    // E * T
    Se3 out;
    double rt[3];
    quat_rotate(E.q, T.t, rt);
    out.t[0] = E.t[0] + rt[0]; out.t[1] = E.t[1] + rt[1]; out.t[2] = E.t[2] + rt[2];
    out.q[3] = E.q[3] * T.q[3] - E.q[0] * T.q[0] - E.q[1] * T.q[1] - E.q[2] * T.q[2];
    out.q[0] = E.q[3] * T.q[0] + E.q[0] * T.q[3] + E.q[1] * T.q[2] - E.q[2] * T.q[1];
    out.q[1] = E.q[3] * T.q[1] + E.q[1] * T.q[3] + E.q[2] * T.q[0] - E.q[0] * T.q[2];
    out.q[2] = E.q[3] * T.q[2] + E.q[2] * T.q[3] + E.q[0] * T.q[1] - E.q[1] * T.q[0];
    normalize_rotation(out.q);
    return out;
}

// error of a projection edge at camera-frame point p; returns the dimension (3 stereo, 2 mono); invz is a float in the
// stereo edges (types_six_dof_expmap.cpp:191,340)
__host__ __device__ inline int edge_error(const double p[3], const BaEdge& e, const CameraD& cam, double err[3]) {
    if (e.ur >= 0) {
        const float invz = (float)(1.0 / p[2]);
        const double u = p[0] * (double)invz * cam.fx + cam.cx, v = p[1] * (double)invz * cam.fy + cam.cy;
        err[0] = e.u - u; err[1] = e.v - v; err[2] = e.ur - (u - cam.bf * (double)invz);
        return 3;
    }
    err[0] = e.u - (cam.fx * p[0] / p[2] + cam.cx);
    err[1] = e.v - (cam.fy * p[1] / p[2] + cam.cy);
    err[2] = 0;
    return 2;
}

// d(error)/d(pose increment), 3 x 6 row-major (rotation columns first); rows beyond `dim` are zero
__host__ __device__ inline void pose_jacobian(const double p[3], bool stereo, bool only_pose, const CameraD& cam, double B[18]) {
    const double x = p[0], y = p[1], z = p[2];
    for (int i = 0; i < 18; ++i) B[i] = 0;
    if (stereo) {
        // the only-pose edge multiplies by 1/z, the binary edge divides by z (types_six_dof_expmap.cpp:254-273 vs :384-403)
        const double iz = 1.0 / z, iz2 = only_pose ? iz * iz : 1.0 / (z * z), fx = cam.fx, fy = cam.fy, bf = cam.bf;
        if (only_pose) {
            B[0] = x * y * iz2 * fx; B[1] = -(1 + (x * x * iz2)) * fx; B[2] = y * iz * fx; B[3] = -iz * fx; B[5] = x * iz2 * fx;
            B[6] = (1 + y * y * iz2) * fy; B[7] = -x * y * iz2 * fy; B[8] = -x * iz * fy; B[10] = -iz * fy; B[11] = y * iz2 * fy;
            B[12] = B[0] - bf * y * iz2; B[13] = B[1] + bf * x * iz2; B[14] = B[2]; B[15] = B[3]; B[17] = B[5] - bf * iz2;
        } else {
            const double z2 = z * z;
            B[0] = x * y / z2 * fx; B[1] = -(1 + (x * x / z2)) * fx; B[2] = y / z * fx; B[3] = -1. / z * fx; B[5] = x / z2 * fx;
            B[6] = (1 + y * y / z2) * fy; B[7] = -x * y / z2 * fy; B[8] = -x / z * fy; B[10] = -1. / z * fy; B[11] = y / z2 * fy;
            B[12] = B[0] - bf * y / z2; B[13] = B[1] + bf * x / z2; B[14] = B[2]; B[15] = B[3]; B[17] = B[5] - bf / z2;
        }
    } else {
        const double J0 = -(cam.fx / z), J2 = cam.fx * x / (z * z), J4 = -(cam.fy / z), J5 = cam.fy * y / (z * z);  // -projectJac
        // SE3deriv = [0 z -y 1 0 0; -z 0 x 0 1 0; y -x 0 0 0 1]
        B[0] = J2 * y;  B[1] = J0 * z + J2 * (-x); B[2] = J0 * (-y); B[3] = J0; B[4] = 0;  B[5] = J2;
        B[6] = J4 * (-z) + J5 * y; B[7] = J5 * (-x); B[8] = J4 * x; B[9] = 0; B[10] = J4; B[11] = J5;
    }
}

// d(error)/d(point), 3 x 3 row-major, of the binary edges
__host__ __device__ inline void point_jacobian(const double p[3], const double R[9], bool stereo, const CameraD& cam, double A[9]) {
    const double x = p[0], y = p[1], z = p[2], z2 = z * z;
    for (int i = 0; i < 9; ++i) A[i] = 0;
    if (stereo) {
        for (int c = 0; c < 3; ++c) {
            A[c] = -cam.fx * R[c] / z + cam.fx * x * R[6 + c] / z2;
            A[3 + c] = -cam.fy * R[3 + c] / z + cam.fy * y * R[6 + c] / z2;
            A[6 + c] = A[c] - cam.bf * R[6 + c] / z2;
        }
    } else {
        const double J0 = -(cam.fx / z), J2 = cam.fx * x / z2, J4 = -(cam.fy / z), J5 = cam.fy * y / z2;
        for (int c = 0; c < 3; ++c) {
            A[c] = J0 * R[c] + J2 * R[6 + c];
            A[3 + c] = J4 * R[3 + c] + J5 * R[6 + c];
        }
    }
}

// RobustKernelHuber::robustify; dsqr is a float member (robust_kernel_impl.h:84)
__host__ __device__ inline void huber(double e, double delta, float dsqr, double& rho0, double& rho1) {
    if (e <= (double)dsqr) { rho0 = e; rho1 = 1.0; }
    else {
        const double s = sqrt(e);
        rho0 = 2 * s * delta - (double)dsqr;
        rho1 = delta / s;
    }
}

// (2 rho - 1)^3 of the Levenberg-Marquardt damping update (optimization_algorithm_levenberg.cpp:133: pow((2 * rho - 1), 3)) as one
// correctly rounded cube: the square and the product carried in double-double (two error-free fma steps), rounded once.  The LM decision
// runs on the host (one-window entry points) and on the device (k_ba_lm_decide_b, round 6); libm's pow and the device library's differ in the
// last bit now and then, this form is the same IEEE operations on both sides -- a window gives the same lambda wherever its loop runs.
__host__ __device__ inline double lm_cube(double x) {
    if (!(x > -1e100 && x < 1e100)) return x * x * x;  // (overflow, infinities, NaN: no error terms to carry)
    const double p = x * x;
    const double pe = __builtin_fma(x, x, -p);   // p + pe = x^2 exactly
    const double h = p * x;
    const double he = __builtin_fma(p, x, -h);   // h + he = p x exactly
    return h + (he + pe * x);
}
// lambda after an accepted step with gain ratio rho (:133-137)
__host__ __device__ inline double lm_lambda_accepted(double lambda, double rho) {
    double alpha = 1. - lm_cube(2 * rho - 1);
    alpha = alpha < 2. / 3. ? alpha : 2. / 3.;
    const double scale = 1. / 3. > alpha ? 1. / 3. : alpha;
    return lambda * scale;
}

// dense LDL^T without pivoting on an n x n row-major matrix (lower triangle used, overwritten); returns false on a
// zero / non-finite pivot, or a negative one when need_positive (Eigen::LDLT::isPositive of LinearSolverDense)
__host__ __device__ inline bool ldlt_solve_small(double* H, int n, const double* b, double* x, bool need_positive) {
    for (int j = 0; j < n; ++j) {
        double d = H[j * n + j];
        for (int k = 0; k < j; ++k) d -= H[j * n + k] * H[j * n + k] * H[k * n + k];
        if (!(d == d) || d == 0.0 || d - d != 0.0 || (need_positive && d < 0.0)) return false;
        H[j * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double s = H[i * n + j];
            for (int k = 0; k < j; ++k) s -= H[i * n + k] * H[j * n + k] * H[k * n + k];
            H[i * n + j] = s / d;
        }
    }
    for (int i = 0; i < n; ++i) {
        double s = b[i];
        for (int k = 0; k < i; ++k) s -= H[i * n + k] * x[k];
        x[i] = s;
    }
    for (int i = 0; i < n; ++i) x[i] /= H[i * n + i];
    // (a row's terms with k DESCENDING since round 5: the order in which a column sweep on the device -- x_k final, every row above takes its
    // term -- adds them, k_ba_solve_b / pi_ldlt_solve_wave; rounds 1-4 added them ascending, which chains the rows on the device)
    for (int i = n - 1; i >= 0; --i) {
        double s = x[i];
        for (int k = n - 1; k > i; --k) s -= H[k * n + i] * x[k];
        x[i] = s;
    }
    return true;
}

}  // namespace tc2li
