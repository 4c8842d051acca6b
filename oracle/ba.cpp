// TEST INFRASTRUCTURE ONLY -- CPU oracle, never linked into or called from the product path.
// See ba.hpp for the reference locations each part follows.
#include "ba.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>

namespace oracle {

// ---- SE3Quat (types/se3quat.h) ----------------------------------------------------------------------------------
static void quat_rotate(const double q[4], const double v[3], double out[3]) {
    // Eigen: uv = 2 * (q.vec x v); v + w * uv + q.vec x uv
    double uv[3] = {q[1] * v[2] - q[2] * v[1], q[2] * v[0] - q[0] * v[2], q[0] * v[1] - q[1] * v[0]};
    for (double& c : uv) c += c;
    out[0] = v[0] + q[3] * uv[0] + (q[1] * uv[2] - q[2] * uv[1]);
    out[1] = v[1] + q[3] * uv[1] + (q[2] * uv[0] - q[0] * uv[2]);
    out[2] = v[2] + q[3] * uv[2] + (q[0] * uv[1] - q[1] * uv[0]);
}
static void quat_mul(const double a[4], const double b[4], double o[4]) {
    o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
    o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    o[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
    o[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
}
static void normalize_rotation(double q[4]) {  // se3quat.h:270-275
    if (q[3] < 0) for (int i = 0; i < 4; ++i) q[i] = -q[i];
    const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int i = 0; i < 4; ++i) q[i] /= n;
}
static void quat_to_matrix(const double q[4], double R[9]) {
    const double tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
    const double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
    const double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
    const double tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
void quat_to_matrix_public(const double q[4], double R[9]) { quat_to_matrix(q, R); }
static void matrix_to_quat(const double R[9], double q[4]) {  // Eigen::Quaterniond(Matrix3d)
    double t = R[0] + R[4] + R[8];
    if (t > 0) {
        t = std::sqrt(t + 1.0);
        q[3] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (R[7] - R[5]) * t; q[1] = (R[2] - R[6]) * t; q[2] = (R[3] - R[1]) * t;
    } else {
        int i = 0;
        if (R[4] > R[0]) i = 1;
        if (R[8] > R[4 * i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(R[4 * i] - R[4 * j] - R[4 * k] + 1.0);
        q[i] = 0.5 * t;
        t = 0.5 / t;
        q[3] = (R[3 * k + j] - R[3 * j + k]) * t;
        q[j] = (R[3 * j + i] + R[3 * i + j]) * t;
        q[k] = (R[3 * k + i] + R[3 * i + k]) * t;
    }
}

void se3_map(const SE3Quat& T, const double X[3], double out[3]) {
    quat_rotate(T.q, X, out);
    for (int i = 0; i < 3; ++i) out[i] += T.t[i];
}

SE3Quat se3_mul(const SE3Quat& a, const SE3Quat& b) {  // se3quat.h:107-113
    SE3Quat r;
    double rt[3];
    quat_rotate(a.q, b.t, rt);
    for (int i = 0; i < 3; ++i) r.t[i] = a.t[i] + rt[i];
    quat_mul(a.q, b.q, r.q);
    normalize_rotation(r.q);
    return r;
}

SE3Quat se3_exp(const double u[6]) {  // se3quat.h:217-257
    const double w[3] = {u[0], u[1], u[2]}, ups[3] = {u[3], u[4], u[5]};
    const double theta = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    const double O[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
    double O2[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) O2[3 * r + c] = O[3 * r] * O[c] + O[3 * r + 1] * O[3 + c] + O[3 * r + 2] * O[6 + c];
    double R[9], V[9];
    if (theta < 0.00001) {
        for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0 ? 1.0 : 0.0) + O[i] + O2[i];
        std::memcpy(V, R, sizeof(R));
    } else {
        const double a = std::sin(theta) / theta, b = (1 - std::cos(theta)) / (theta * theta),
                     c = (theta - std::sin(theta)) / std::pow(theta, 3);
        for (int i = 0; i < 9; ++i) {
            R[i] = (i % 4 == 0 ? 1.0 : 0.0) + a * O[i] + b * O2[i];
            V[i] = (i % 4 == 0 ? 1.0 : 0.0) + b * O[i] + c * O2[i];
        }
    }
    SE3Quat T;
    matrix_to_quat(R, T.q);
    for (int r = 0; r < 3; ++r) T.t[r] = V[3 * r] * ups[0] + V[3 * r + 1] * ups[1] + V[3 * r + 2] * ups[2];
    normalize_rotation(T.q);
    return T;
}

// ---- edges --------------------------------------------------------------------------------------------------------
// error of the projection edges: stereo (types_six_dof_expmap.cpp:190-197, 339-346; invz in float) and monocular
// (Pinhole::project, SF/src/CameraModels/Pinhole.cpp:44-50)
static int edge_error(const double Xc[3], const BAEdge& e, const Camera& cam, double err[3]) {
    if (e.obs[2] >= 0) {
        const float invz = (float)(1.0f / Xc[2]);
        const double u = Xc[0] * invz * cam.fx + cam.cx, v = Xc[1] * invz * cam.fy + cam.cy;
        err[0] = e.obs[0] - u;
        err[1] = e.obs[1] - v;
        err[2] = e.obs[2] - (u - cam.bf * invz);
        return 3;
    }
    err[0] = e.obs[0] - (cam.fx * Xc[0] / Xc[2] + cam.cx);
    err[1] = e.obs[1] - (cam.fy * Xc[1] / Xc[2] + cam.cy);
    err[2] = 0;
    return 2;
}

int edge_linearize(const SE3Quat& T, const double X[3], const BAEdge& e, const Camera& cam, double err[3], double A[9], double B[18]) {
    double p[3];
    se3_map(T, X, p);
    const int dim = edge_error(p, e, cam, err);
    const double x = p[0], y = p[1], z = p[2], z_2 = z * z;
    double R[9];
    quat_to_matrix(T.q, R);
    std::memset(A, 0, 9 * sizeof(double));
    std::memset(B, 0, 18 * sizeof(double));
    if (dim == 3) {  // EdgeStereoSE3ProjectXYZ::linearizeOplus, types_six_dof_expmap.cpp:228-274
        const double fx = cam.fx, fy = cam.fy, bf = cam.bf;
        for (int c = 0; c < 3; ++c) {
            A[c] = -fx * R[c] / z + fx * x * R[6 + c] / z_2;
            A[3 + c] = -fy * R[3 + c] / z + fy * y * R[6 + c] / z_2;
            A[6 + c] = A[c] - bf * R[6 + c] / z_2;
        }
        B[0] = x * y / z_2 * fx; B[1] = -(1 + (x * x / z_2)) * fx; B[2] = y / z * fx; B[3] = -1. / z * fx; B[4] = 0; B[5] = x / z_2 * fx;
        B[6] = (1 + y * y / z_2) * fy; B[7] = -x * y / z_2 * fy; B[8] = -x / z * fy; B[9] = 0; B[10] = -1. / z * fy; B[11] = y / z_2 * fy;
        B[12] = B[0] - bf * y / z_2; B[13] = B[1] + bf * x / z_2; B[14] = B[2]; B[15] = B[3]; B[16] = 0; B[17] = B[5] - bf / z_2;
    } else {  // TC2LI_SLAM::EdgeSE3ProjectXYZ::linearizeOplus, SF/src/OptimizableTypes.cpp:148-169
        const double J[6] = {-(cam.fx / z), -0.0, -(-cam.fx * x / (z * z)), -0.0, -(cam.fy / z), -(-cam.fy * y / (z * z))};  // -projectJac
        for (int r = 0; r < 2; ++r)
            for (int c = 0; c < 3; ++c) A[3 * r + c] = J[3 * r] * R[c] + J[3 * r + 1] * R[3 + c] + J[3 * r + 2] * R[6 + c];
        const double D[18] = {0, z, -y, 1, 0, 0, -z, 0, x, 0, 1, 0, y, -x, 0, 0, 0, 1};  // SE3deriv
        for (int r = 0; r < 2; ++r)
            for (int c = 0; c < 6; ++c) B[6 * r + c] = J[3 * r] * D[c] + J[3 * r + 1] * D[6 + c] + J[3 * r + 2] * D[12 + c];
    }
    return dim;
}

// pose-only Jacobian (EdgeStereoSE3ProjectXYZOnlyPose::linearizeOplus, types_six_dof_expmap.cpp:376-404;
// EdgeSE3ProjectXYZOnlyPose::linearizeOplus, OptimizableTypes.cpp:58-72)
static void pose_only_jacobian(const double p[3], bool stereo, const Camera& cam, double B[18]) {
    const double x = p[0], y = p[1];
    std::memset(B, 0, 18 * sizeof(double));
    if (stereo) {
        const double invz = 1.0 / p[2], invz_2 = invz * invz, fx = cam.fx, fy = cam.fy, bf = cam.bf;
        B[0] = x * y * invz_2 * fx; B[1] = -(1 + (x * x * invz_2)) * fx; B[2] = y * invz * fx; B[3] = -invz * fx; B[4] = 0; B[5] = x * invz_2 * fx;
        B[6] = (1 + y * y * invz_2) * fy; B[7] = -x * y * invz_2 * fy; B[8] = -x * invz * fy; B[9] = 0; B[10] = -invz * fy; B[11] = y * invz_2 * fy;
        B[12] = B[0] - bf * y * invz_2; B[13] = B[1] + bf * x * invz_2; B[14] = B[2]; B[15] = B[3]; B[16] = 0; B[17] = B[5] - bf * invz_2;
    } else {
        const double z = p[2];
        const double J[6] = {-(cam.fx / z), -0.0, -(-cam.fx * x / (z * z)), -0.0, -(cam.fy / z), -(-cam.fy * y / (z * z))};
        const double D[18] = {0, z, -y, 1, 0, 0, -z, 0, x, 0, 1, 0, y, -x, 0, 0, 0, 1};
        for (int r = 0; r < 2; ++r)
            for (int c = 0; c < 6; ++c) B[6 * r + c] = J[3 * r] * D[c] + J[3 * r + 1] * D[6 + c] + J[3 * r + 2] * D[12 + c];
    }
}

// RobustKernelHuber (robust_kernel_impl.cpp:65-91; dsqr is stored in a float, robust_kernel_impl.h:84)
struct Huber {
    double delta;
    float dsqr;
    explicit Huber(float d) : delta(d), dsqr((float)((double)d * (double)d)) {}
    void robustify(double e, double rho[3]) const {
        if (e <= dsqr) { rho[0] = e; rho[1] = 1.; rho[2] = 0.; }
        else {
            const double sqrte = std::sqrt(e);
            rho[0] = 2 * sqrte * delta - dsqr;
            rho[1] = delta / sqrte;
            rho[2] = -0.5 * rho[1] / e;
        }
    }
};

// dense LDL^T without pivoting.  `need_positive`: Eigen::LDLT::isPositive() of LinearSolverDense (linear_solver_dense.h:105-110);
// otherwise only a zero pivot fails, like SimplicialLDLT in LinearSolverEigen (linear_solver_eigen.h:100-111).
static bool ldlt_solve(std::vector<double>& H, int n, const double* b, double* x, bool need_positive) {
    std::vector<double> D(n);
    for (int j = 0; j < n; ++j) {
        double d = H[(size_t)j * n + j];
        for (int k = 0; k < j; ++k) d -= H[(size_t)j * n + k] * H[(size_t)j * n + k] * D[k];
        if (!(std::isfinite(d)) || d == 0.0 || (need_positive && d < 0.0)) return false;
        D[j] = d;
        for (int i = j + 1; i < n; ++i) {
            double s = H[(size_t)i * n + j];
            for (int k = 0; k < j; ++k) s -= H[(size_t)i * n + k] * H[(size_t)j * n + k] * D[k];
            H[(size_t)i * n + j] = s / d;
        }
    }
    for (int i = 0; i < n; ++i) {
        double s = b[i];
        for (int k = 0; k < i; ++k) s -= H[(size_t)i * n + k] * x[k];
        x[i] = s;
    }
    for (int i = 0; i < n; ++i) x[i] /= D[i];
    for (int i = n - 1; i >= 0; --i) {
        double s = x[i];
        for (int k = i + 1; k < n; ++k) s -= H[(size_t)k * n + i] * x[k];
        x[i] = s;
    }
    return true;
}

// ---- the graph + Levenberg-Marquardt ------------------------------------------------------------------------------
namespace {

struct Problem {
    std::vector<SE3Quat>* poses;
    std::vector<uint8_t> fixed;     // per pose
    std::vector<double>* points;    // 3 per point; nullptr: points are constants (pose-only problem, Xw)
    const std::vector<double>* Xw;  // constants for the pose-only problem
    const std::vector<BAEdge>* edges;
    std::vector<int> level;         // per edge (0 = active)
    std::vector<uint8_t> robust;    // per edge
    Camera cam;
    Huber huber_mono{(float)std::sqrt(5.991)}, huber_stereo{(float)std::sqrt(7.815)};
    const bool* stop = nullptr;
    EdgeLidar* lidar = nullptr;      // EdgeLidarSE3 over the window vertices lidar_pose
    std::vector<int> lidar_pose;
    // per-edge state kept between calls, like g2o's _error
    std::vector<double> err;   // 3 per edge
    std::vector<double> chi2;  // per edge

    // active structure
    std::vector<int> active;           // edge ids
    std::vector<int> pose_var, point_var;  // -1 = not a variable
    int n_pose_vars = 0, n_point_vars = 0;
    // LM state
    double lambda = -1, ni = 2;
    int n_bad = 0;
    double user_lambda = 0;
    LMTrace* trace = nullptr;

    const double* X(int e) const { return points ? &(*points)[3 * (*edges)[e].point] : &(*Xw)[3 * (*edges)[e].point]; }
    bool stereo(int e) const { return (*edges)[e].obs[2] >= 0; }
    const Huber& huber(int e) const { return stereo(e) ? huber_stereo : huber_mono; }

    void compute_error(int e) {
        double p[3];
        se3_map((*poses)[(*edges)[e].pose], X(e), p);
        const int dim = edge_error(p, (*edges)[e], cam, &err[3 * e]);
        double s = 0;
        for (int i = 0; i < dim; ++i) s += err[3 * e + i] * (*edges)[e].info * err[3 * e + i];
        chi2[e] = s;
    }
    void lidar_vertex_matrices(std::vector<double>& R, std::vector<double>& t) const {
        R.resize(9 * lidar_pose.size()); t.resize(3 * lidar_pose.size());
        for (size_t i = 0; i < lidar_pose.size(); ++i) {
            quat_to_matrix((*poses)[lidar_pose[i]].q, &R[9 * i]);
            std::memcpy(&t[3 * i], (*poses)[lidar_pose[i]].t, 3 * sizeof(double));
        }
    }
    void compute_active_errors() {
        for (int e : active) compute_error(e);
        if (lidar) {
            std::vector<double> R, t;
            lidar_vertex_matrices(R, t);
            lidar->computeError(R.data(), t.data(), (int)lidar_pose.size());
        }
    }
    double active_robust_chi2() const {
        double chi = lidar ? lidar->chi2() : 0.0;
        for (int e : active) {
            if (robust[e]) { double rho[3]; huber(e).robustify(chi2[e], rho); chi += rho[0]; }
            else chi += chi2[e];
        }
        return chi;
    }
    bool terminate() const { return stop && *stop; }

    void initialize(int lvl) {  // SparseOptimizer::initializeOptimization(level)
        active.clear();
        for (int e = 0; e < (int)edges->size(); ++e) if (level[e] == lvl) active.push_back(e);
        pose_var.assign(poses->size(), -1);
        std::vector<uint8_t> used_pose(poses->size(), 0), used_point(points ? points->size() / 3 : 0, 0);
        for (int e : active) { used_pose[(*edges)[e].pose] = 1; if (points) used_point[(*edges)[e].point] = 1; }
        for (int k : lidar_pose) used_pose[k] = 1;
        n_pose_vars = 0;
        for (size_t i = 0; i < poses->size(); ++i) if (used_pose[i] && !fixed[i]) pose_var[i] = n_pose_vars++;
        point_var.assign(used_point.size(), -1);
        n_point_vars = 0;
        for (size_t i = 0; i < used_point.size(); ++i) if (used_point[i]) point_var[i] = n_point_vars++;
    }

    int optimize(int iterations);
};

int Problem::optimize(int iterations) {
    const int np = 6 * n_pose_vars, nl = 3 * n_point_vars, nvar = np + nl;
    if (nvar == 0) return -1;
    std::vector<double> Hpp((size_t)np * np), Hll((size_t)9 * n_point_vars), b(nvar), x(nvar), bs(np), coeff(nvar);
    // Hpl blocks per active edge with both ends variable (pose rows x point cols)
    std::vector<double> Hpl((size_t)18 * active.size());
    std::vector<double> S, Dinv((size_t)9 * n_point_vars);
    std::vector<SE3Quat> backup_pose;
    std::vector<double> backup_points;
    int done = 0;
    bool ok = true;
    for (int it = 0; it < iterations && !terminate() && ok; ++it) {
        // ---- OptimizationAlgorithmLevenberg::solve(it) ----
        compute_active_errors();
        double currentChi = active_robust_chi2(), tempChi = currentChi;
        const double iniChi = currentChi;
        // buildSystem
        std::fill(Hpp.begin(), Hpp.end(), 0.0); std::fill(Hll.begin(), Hll.end(), 0.0);
        std::fill(b.begin(), b.end(), 0.0); std::fill(Hpl.begin(), Hpl.end(), 0.0);
        for (size_t k = 0; k < active.size(); ++k) {
            const int e = active[k];
            const BAEdge& ed = (*edges)[e];
            const int pv = pose_var[ed.pose];
            double er[3], A[9], B[18];
            int dim;
            if (points) dim = edge_linearize((*poses)[ed.pose], X(e), ed, cam, er, A, B);
            else {
                double p[3];
                se3_map((*poses)[ed.pose], X(e), p);
                dim = edge_error(p, ed, cam, er);
                pose_only_jacobian(p, stereo(e), cam, B);
            }
            // g2o keeps _error from computeActiveErrors; it is the same value
            double rho1 = 1.0;
            if (robust[e]) { double rho[3]; huber(e).robustify(chi2[e], rho); rho1 = rho[1]; }
            const double w = rho1 * ed.info;  // weightedOmega = rho'[1] * information
            double omega_r[3];
            for (int i = 0; i < dim; ++i) omega_r[i] = -ed.info * err[3 * e + i] * rho1;
            if (points) {
                const int lv = point_var[ed.point];
                double* hl = &Hll[(size_t)9 * lv];
                for (int r = 0; r < 3; ++r) {
                    double s = 0;
                    for (int i = 0; i < dim; ++i) s += A[3 * i + r] * omega_r[i];
                    b[np + 3 * lv + r] += s;
                    for (int c = 0; c < 3; ++c) {
                        double h = 0;
                        for (int i = 0; i < dim; ++i) h += A[3 * i + r] * w * A[3 * i + c];
                        hl[3 * r + c] += h;
                    }
                }
                if (pv >= 0) {
                    double* hp = &Hpl[(size_t)18 * k];
                    for (int r = 0; r < 6; ++r)
                        for (int c = 0; c < 3; ++c) {
                            double h = 0;
                            for (int i = 0; i < dim; ++i) h += B[6 * i + r] * w * A[3 * i + c];
                            hp[3 * r + c] += h;
                        }
                }
            }
            if (pv >= 0) {
                for (int r = 0; r < 6; ++r) {
                    double s = 0;
                    for (int i = 0; i < dim; ++i) s += B[6 * i + r] * omega_r[i];
                    b[6 * pv + r] += s;
                    for (int c = 0; c < 6; ++c) {
                        double h = 0;
                        for (int i = 0; i < dim; ++i) h += B[6 * i + r] * w * B[6 * i + c];
                        Hpp[(size_t)(6 * pv + r) * np + 6 * pv + c] += h;
                    }
                }
            }
        }
        if (lidar) {
            // EdgeLidarSE3::linearizeOplus + computeQuadraticFormLidarRes (G2oTypesWithLidar.h:148-236), quirks included:
            // the 6x6 blocks are read at ELEMENT offsets (i, i) / (i, j) of the 6W x 6W Hessian, and b -= info * J^T
            std::vector<double> R, t;
            lidar_vertex_matrices(R, t);
            const int W = (int)lidar_pose.size(), n = 6 * W;
            lidar->linearizeOplus(R.data(), t.data(), W);
            const double info = lidar->information;
            for (int i = 0; i < W; ++i) {
                const int vi = pose_var[lidar_pose[i]];
                if (vi < 0) continue;
                for (int r = 0; r < 6; ++r) {
                    b[6 * vi + r] -= info * lidar->JacT[6 * i + r];
                    for (int c = 0; c < 6; ++c) Hpp[(size_t)(6 * vi + r) * np + 6 * vi + c] += lidar->Hessian[(size_t)(i + r) * n + i + c] * info;
                }
                for (int j = i + 1; j < W; ++j) {
                    const int vj = pose_var[lidar_pose[j]];
                    if (vj < 0) continue;
                    for (int r = 0; r < 6; ++r)
                        for (int c = 0; c < 6; ++c) {
                            const double h = lidar->Hessian[(size_t)(i + r) * n + j + c] * info;
                            Hpp[(size_t)(6 * vi + r) * np + 6 * vj + c] += h;
                            Hpp[(size_t)(6 * vj + c) * np + 6 * vi + r] += h;
                        }
                }
            }
        }
        if (it == 0) {  // computeLambdaInit, optimization_algorithm_levenberg.cpp:171-185
            if (user_lambda > 0) lambda = user_lambda;
            else {
                double mx = 0;
                for (int i = 0; i < np; ++i) mx = std::max(std::fabs(Hpp[(size_t)i * np + i]), mx);
                for (int l = 0; l < n_point_vars; ++l) for (int j = 0; j < 3; ++j) mx = std::max(std::fabs(Hll[(size_t)9 * l + 4 * j]), mx);
                lambda = 1e-5 * mx;
            }
            ni = 2;
            n_bad = 0;
        }
        double rho = 0;
        int qmax = 0;
        do {
            backup_pose = *poses;
            if (points) backup_points = *points;
            // setLambda + solve (block_solver.hpp:353-486)
            bool ok2;
            if (!points) {
                std::vector<double> H = Hpp;
                for (int i = 0; i < np; ++i) H[(size_t)i * np + i] += lambda;
                ok2 = ldlt_solve(H, np, b.data(), x.data(), true);
            } else {
                S = Hpp;
                for (int i = 0; i < np; ++i) S[(size_t)i * np + i] += lambda;
                std::fill(coeff.begin(), coeff.end(), 0.0);
                for (int l = 0; l < n_point_vars; ++l) {
                    double D[9];
                    std::memcpy(D, &Hll[(size_t)9 * l], sizeof(D));
                    D[0] += lambda; D[4] += lambda; D[8] += lambda;
                    // 3x3 inverse by cofactors (Eigen's fixed-size inverse)
                    const double c00 = D[4] * D[8] - D[5] * D[7], c01 = D[5] * D[6] - D[3] * D[8], c02 = D[3] * D[7] - D[4] * D[6];
                    const double det = D[0] * c00 + D[1] * c01 + D[2] * c02, id = 1.0 / det;
                    double* Di = &Dinv[(size_t)9 * l];
                    Di[0] = c00 * id; Di[1] = (D[2] * D[7] - D[1] * D[8]) * id; Di[2] = (D[1] * D[5] - D[2] * D[4]) * id;
                    Di[3] = c01 * id; Di[4] = (D[0] * D[8] - D[2] * D[6]) * id; Di[5] = (D[2] * D[3] - D[0] * D[5]) * id;
                    Di[6] = c02 * id; Di[7] = (D[1] * D[6] - D[0] * D[7]) * id; Di[8] = (D[0] * D[4] - D[1] * D[3]) * id;
                }
                // per landmark: the edges with a variable pose, in active order
                std::vector<std::vector<int>> by_point(n_point_vars);
                for (size_t k = 0; k < active.size(); ++k) {
                    const BAEdge& ed = (*edges)[active[k]];
                    if (pose_var[ed.pose] >= 0) by_point[point_var[ed.point]].push_back((int)k);
                }
                for (int l = 0; l < n_point_vars; ++l) {
                    const double* Di = &Dinv[(size_t)9 * l];
                    double db[3];
                    for (int r = 0; r < 3; ++r) db[r] = Di[3 * r] * b[np + 3 * l] + Di[3 * r + 1] * b[np + 3 * l + 1] + Di[3 * r + 2] * b[np + 3 * l + 2];
                    for (int k1 : by_point[l]) {
                        const int i1 = pose_var[(*edges)[active[k1]].pose];
                        const double* Bi = &Hpl[(size_t)18 * k1];
                        double BDinv[18];
                        for (int r = 0; r < 6; ++r)
                            for (int c = 0; c < 3; ++c) BDinv[3 * r + c] = Bi[3 * r] * Di[c] + Bi[3 * r + 1] * Di[3 + c] + Bi[3 * r + 2] * Di[6 + c];
                        for (int r = 0; r < 6; ++r) coeff[6 * i1 + r] += Bi[3 * r] * db[0] + Bi[3 * r + 1] * db[1] + Bi[3 * r + 2] * db[2];
                        for (int k2 : by_point[l]) {
                            const int i2 = pose_var[(*edges)[active[k2]].pose];
                            const double* Bj = &Hpl[(size_t)18 * k2];
                            for (int r = 0; r < 6; ++r)
                                for (int c = 0; c < 6; ++c)
                                    S[(size_t)(6 * i1 + r) * np + 6 * i2 + c] -= BDinv[3 * r] * Bj[3 * c] + BDinv[3 * r + 1] * Bj[3 * c + 1] + BDinv[3 * r + 2] * Bj[3 * c + 2];
                        }
                    }
                }
                for (int i = 0; i < np; ++i) bs[i] = b[i] - coeff[i];
                ok2 = np == 0 ? true : ldlt_solve(S, np, bs.data(), x.data(), false);
                if (ok2) {
                    // xl = Dinv * (bl - Hpl^T xp)
                    std::vector<double> cl(b.begin() + np, b.end());
                    for (size_t k = 0; k < active.size(); ++k) {
                        const BAEdge& ed = (*edges)[active[k]];
                        const int pv = pose_var[ed.pose];
                        if (pv < 0) continue;
                        const int lv = point_var[ed.point];
                        const double* Bi = &Hpl[(size_t)18 * k];
                        for (int c = 0; c < 3; ++c) {
                            double s = 0;
                            for (int r = 0; r < 6; ++r) s += Bi[3 * r + c] * x[6 * pv + r];
                            cl[3 * lv + c] -= s;
                        }
                    }
                    for (int l = 0; l < n_point_vars; ++l) {
                        const double* Di = &Dinv[(size_t)9 * l];
                        for (int r = 0; r < 3; ++r)
                            x[np + 3 * l + r] = Di[3 * r] * cl[3 * l] + Di[3 * r + 1] * cl[3 * l + 1] + Di[3 * r + 2] * cl[3 * l + 2];
                    }
                }
            }
            // update (SparseOptimizer::update; VertexSE3Expmap::oplusImpl, VertexSBAPointXYZ::oplusImpl)
            for (size_t i = 0; i < poses->size(); ++i)
                if (pose_var[i] >= 0) (*poses)[i] = se3_mul(se3_exp(&x[6 * pose_var[i]]), (*poses)[i]);
            if (points)
                for (size_t i = 0; i < point_var.size(); ++i)
                    if (point_var[i] >= 0) for (int c = 0; c < 3; ++c) (*points)[3 * i + c] += x[np + 3 * point_var[i] + c];
            compute_active_errors();
            tempChi = active_robust_chi2();
            if (!ok2) tempChi = std::numeric_limits<double>::max();
            rho = currentChi - tempChi;
            double scale = 0;
            for (int j = 0; j < nvar; ++j) scale += x[j] * (lambda * x[j] + b[j]);
            scale += 1e-3;
            rho /= scale;
            if (rho > 0 && std::isfinite(tempChi)) {
                double alpha = 1. - std::pow((2 * rho - 1), 3);
                alpha = std::min(alpha, 2. / 3.);
                const double scaleFactor = std::max(1. / 3., alpha);
                lambda *= scaleFactor;
                ni = 2;
                currentChi = tempChi;
            } else {
                lambda *= ni;
                ni *= 2;
                *poses = backup_pose;
                if (points) *points = backup_points;
            }
            qmax++;
        } while (rho < 0 && qmax < 10 && !terminate());
        ++done;
        if (trace) { trace->chi2.push_back(currentChi); trace->lambda.push_back(lambda); trace->trials.push_back(qmax); }
        if (qmax == 10 || rho == 0) { ok = false; continue; }
        if ((iniChi - currentChi) * 1e3 < iniChi) n_bad++; else n_bad = 0;
        if (n_bad >= 3) ok = false;
    }
    return done;
}

}  // namespace

int PoseOptimization(SE3Quat& pose, const std::vector<double>& Xw, const std::vector<BAEdge>& edges, const Camera& cam,
                     std::vector<uint8_t>& outlier, LMTrace* trace) {
    const int N = (int)edges.size();
    outlier.assign(N, 0);
    if (N < 3) return 0;  // nInitialCorrespondences < 3, Optimizer.cc:999-1000
    std::vector<SE3Quat> poses(1);
    Problem P;
    P.poses = &poses; P.fixed = {0}; P.points = nullptr; P.Xw = &Xw; P.edges = &edges; P.cam = cam;
    P.level.assign(N, 0); P.robust.assign(N, 1); P.err.assign(3 * N, 0); P.chi2.assign(N, 0); P.trace = trace;
    const SE3Quat initial = pose;
    int nBad = 0;
    for (int it = 0; it < 4; it++) {
        poses[0] = initial;  // every round restarts from the frame's pose (Optimizer.cc:1012-1013)
        P.initialize(0);
        P.optimize(10);
        nBad = 0;
        for (int i = 0; i < N; ++i) {
            if (outlier[i]) P.compute_error(i);
            const float chi2 = (float)P.chi2[i];
            const float th = edges[i].obs[2] >= 0 ? 7.815f : 5.991f;
            if (chi2 > th) { outlier[i] = 1; P.level[i] = 1; nBad++; }
            else { outlier[i] = 0; P.level[i] = 0; }
            if (it == 2) P.robust[i] = 0;
        }
        if (N < 10) break;
    }
    // Frame::SetPose(Sophus::SE3<float>(q.cast<float>(), t.cast<float>()))
    for (int i = 0; i < 4; ++i) { volatile float f = (float)poses[0].q[i]; pose.q[i] = (double)f; }
    for (int i = 0; i < 3; ++i) { volatile float f = (float)poses[0].t[i]; pose.t[i] = (double)f; }
    return N - nBad;
}

BAResult LocalBundleAdjustment(std::vector<SE3Quat>& poses, const std::vector<uint8_t>& fixed, std::vector<double>& points,
                               const std::vector<BAEdge>& edges, const Camera& cam, int iterations, double lambda_init,
                               const bool* stop, EdgeLidar* lidar, const std::vector<int>* lidar_pose) {
    BAResult res;
    const int E = (int)edges.size();
    Problem P;
    P.poses = &poses; P.fixed = fixed; P.points = &points; P.Xw = nullptr; P.edges = &edges; P.cam = cam;
    P.level.assign(E, 0); P.robust.assign(E, 1); P.err.assign(3 * E, 0); P.chi2.assign(E, 0);
    P.user_lambda = lambda_init; P.stop = stop; P.trace = &res.trace;
    if (lidar && lidar_pose) { P.lidar = lidar; P.lidar_pose = *lidar_pose; }
    P.initialize(0);
    res.iterations = (stop && *stop) ? 0 : P.optimize(iterations);
    res.chi2 = P.chi2;  // e->chi2() as the optimiser left it (OptimizerWithLidar.cc:406-449)
    res.depth_pos.resize(E);
    for (int e = 0; e < E; ++e) {
        double p[3];
        se3_map(poses[edges[e].pose], &points[3 * edges[e].point], p);
        res.depth_pos[e] = p[2] > 0.0;
    }
    return res;
}

}  // namespace oracle
