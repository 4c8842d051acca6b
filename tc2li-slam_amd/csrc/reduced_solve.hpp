// The reduced system of LocalInertialBA / LocalLVIBA on the host (SF/src/Optimizer.cc:1635-1638, OptimizerWithLidar.cc:613-616: g2o's
// BlockSolverX over LinearSolverEigen, i.e. a SPARSE LDL^T of the system left after the landmarks are marginalised): unknowns are 6 per free
// keyframe pose and 9 per keyframe with velocity / gyro bias / accelerometer bias vertices -- 375 for the 25-keyframe bLarge window.  The pose
// block is dense (the Schur complement couples every pair of covisible keyframes), the rest is not: an inertial edge joins the states of two
// CONSECUTIVE keyframes, so with the velocity / bias unknowns ordered first the matrix is a narrow band followed by the pose rows, and the
// pose row of keyframe p starts at the states of keyframe p - 1.  A row-wise LDL^T inside that envelope (fill never leaves it) does a fifth of
// the dense factorisation's work at 150 unknowns and a ninth at 375; the dense, scalar ldlt_solve_small of rounds 1-4 took 6.4 ms per bLarge
// window and trial on a host core -- two thirds of such a batch's wall time, and most of the sixteen CPUs of a one-GPU box in configs[3].
// The arithmetic is plain IEEE double in a fixed order (inner products as eight interleaved partial sums, closed in a fixed tree): a window
// gives the same bits alone and in a lock-step batch, on every x86-64 host (no FMA contraction: the library is built with -ffp-contract=off).
#pragma once
#include <algorithm>
#include <cstddef>
#include <vector>

namespace tc2li {

struct ReducedSolver {
    int n = 0, np = 0, ni = 0;        // unknowns, of them pose unknowns (first in the caller's numbering) and velocity / bias unknowns (after them)
    std::vector<double> M;             // [n][n] row-major in the solver's order [velocity / bias | poses]; lower triangle used, L below D after factorise
    std::vector<int> first;            // the envelope: first[i] = column of row i's first entry
    std::vector<double> D, z;

    static inline double dot8(const double* a, const double* b, int len) {
        double s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0, s6 = 0, s7 = 0;
        int k = 0;
        for (; k + 8 <= len; k += 8) {
            s0 += a[k] * b[k]; s1 += a[k + 1] * b[k + 1]; s2 += a[k + 2] * b[k + 2]; s3 += a[k + 3] * b[k + 3];
            s4 += a[k + 4] * b[k + 4]; s5 += a[k + 5] * b[k + 5]; s6 += a[k + 6] * b[k + 6]; s7 += a[k + 7] * b[k + 7];
        }
        double t = 0;
        for (; k < len; ++k) t += a[k] * b[k];
        return (((s0 + s4) + (s2 + s6)) + ((s1 + s5) + (s3 + s7))) + t;
    }
    inline int to_solver(int j) const { return j < np ? ni + j : j - np; }   // the caller's index -> the solver's

    // Once per linearisation: the envelope from the entries of Hi (the inertial / LiDAR part, [n][n] in the caller's numbering, lower triangle
    // as ldlt_solve_small read it); the pose block counts as dense.
    void set_pattern(const double* Hi, int n_, int np_) {
        n = n_; np = np_; ni = n - np;
        M.resize((size_t)std::max(n * n, 1)); first.resize(std::max(n, 1)); D.resize(std::max(n, 1)); z.resize(std::max(n, 1));
        for (int r = 0; r < ni; ++r) {           // a velocity / bias row: against the velocity / bias columns before it
            const double* row = Hi + (size_t)(np + r) * n + np;
            int f = r;
            for (int c = 0; c < r; ++c) if (row[c] != 0.0) { f = c; break; }
            first[r] = f;
        }
        for (int r = 0; r < np; ++r) {           // a pose row: its entries against the velocity / bias columns are Hi[velocity / bias row][pose column]
            int f = ni;
            for (int c = 0; c < ni; ++c) if (Hi[(size_t)(np + c) * n + r] != 0.0) { f = c; break; }
            first[ni + r] = f;
        }
    }
    // Per trial: M = [S + Hi(poses) | Hi(poses, imu); . | Hi(imu) + lambda I] in the solver's order, factorised; false as ldlt_solve_small (a zero or
    // non-finite pivot).  S: [np][np] with the damping already on its diagonal (k_ba_schur_finish).
    bool factorise(const double* Hi, const double* S, double lambda) {
        for (int r = 0; r < ni; ++r) {
            const double* src = Hi + (size_t)(np + r) * n + np;
            double* dst = M.data() + (size_t)r * n;
            for (int c = first[r]; c <= r; ++c) dst[c] = src[c];
            dst[r] += lambda;
        }
        for (int r = 0; r < np; ++r) {
            double* dst = M.data() + (size_t)(ni + r) * n;
            for (int c = first[ni + r]; c < ni; ++c) dst[c] = Hi[(size_t)(np + c) * n + r];
            const double* hp = Hi + (size_t)r * n;
            const double* sp = S + (size_t)r * np;
            for (int c = 0; c <= r; ++c) dst[ni + c] = hp[c] + sp[c];
        }
        for (int i = 0; i < n; ++i) {
            double* ri = M.data() + (size_t)i * n;
            const int fi = first[i];
            // y_j = M_ij - sum_k y_k L_jk  (y = L_i D): row i against every row before it, inside both envelopes
            for (int j = fi; j < i; ++j) {
                const int k0 = std::max(fi, first[j]);
                if (k0 < j) ri[j] -= dot8(ri + k0, M.data() + (size_t)j * n + k0, j - k0);
            }
            double d = ri[i];
            for (int j = fi; j < i; ++j) {
                const double y = ri[j], l = y / D[j];
                d -= y * l;
                ri[j] = l;
            }
            if (!(d == d) || d == 0.0 || d - d != 0.0) return false;
            D[i] = d;
        }
        return true;
    }
    // x (the caller's numbering) from rhs (the caller's numbering)
    void solve(const double* rhs, double* x) {
        for (int j = 0; j < n; ++j) z[to_solver(j)] = rhs[j];
        for (int i = 0; i < n; ++i) {
            const int fi = first[i];
            if (fi < i) z[i] -= dot8(M.data() + (size_t)i * n + fi, z.data() + fi, i - fi);
        }
        for (int i = 0; i < n; ++i) z[i] /= D[i];
        for (int i = n - 1; i >= 0; --i) {       // column sweep: x_i is final, every row of its envelope takes its term
            const double xi = z[i];
            const double* ri = M.data() + (size_t)i * n;
            for (int k = first[i]; k < i; ++k) z[k] -= ri[k] * xi;
        }
        for (int j = 0; j < n; ++j) x[j] = z[to_solver(j)];
    }
};

}  // namespace tc2li
