// The reduced system of LocalInertialBA / LocalLVIBA on the host (SF/src/Optimizer.cc:1635-1638, OptimizerWithLidar.cc:613-616: g2o's
// BlockSolverX over LinearSolverEigen, i.e. a SPARSE LDL^T of the system left after the landmarks are marginalised): unknowns are 6 per free
// keyframe pose and 9 per keyframe with velocity / gyro bias / accelerometer bias vertices -- 375 for the 25-keyframe bLarge window.  The pose
// block is dense (the Schur complement couples every pair of covisible keyframes), the rest is not: an inertial edge joins the states of two
// CONSECUTIVE keyframes, so with the velocity / bias unknowns ordered first the matrix is a narrow band followed by the pose rows, and the
// pose row of keyframe p starts at the states of keyframe p - 1.  A row-wise LDL^T inside that envelope (fill never leaves it) does a fifth of
// the dense factorisation's work at 150 unknowns and a ninth at 375; the dense, scalar ldlt_solve_small of rounds 1-4 took 6.4 ms per bLarge
// window and trial on a host core -- two thirds of such a batch's wall time, and most of the sixteen CPUs of a one-GPU box in configs[3].
// The arithmetic is plain IEEE double in a fixed order (inner products as eight interleaved partial sums, closed in a fixed tree): a window
// gives the same bits alone and in a lock-step batch, on every x86-64 host (no FMA contraction: the library is built with -ffp-contract=off; the
// AVX2 build of the inner products -- reduced_solve.cc, chosen at run time -- multiplies and adds the same eight lanes in the same order).
#pragma once
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <vector>

namespace tc2li {

struct ReducedSolver {
    int n = 0, np = 0, ni = 0;        // unknowns, of them pose unknowns (first in the caller's numbering) and velocity / bias unknowns (after them)
    std::vector<double> M;             // [n][n] row-major in the solver's order [velocity / bias | poses]; lower triangle used, L below D after factorise
    std::vector<int> first;            // the envelope: first[i] = column of row i's first entry
    std::vector<int> span_end;         // [np] pose row r: one past its last entry against the velocity / bias columns (what follows up to the pose block is fill)
    std::vector<double> D, z;

    inline int to_solver(int j) const { return j < np ? ni + j : j - np; }   // the caller's index -> the solver's

    // Once per linearisation: the envelope from the entries of Hi (the inertial / LiDAR part, [n][n] in the caller's numbering, lower triangle
    // as ldlt_solve_small read it); the pose block counts as dense.
    // need_matrix = false: the system is solved on the device (only the envelope and pack_for_device are wanted).
    void set_pattern(const double* Hi, int n_, int np_, bool need_matrix = true) {
        n = n_; np = np_; ni = n - np;
        if (need_matrix) { M.resize((size_t)std::max(n * n, 1)); D.resize(std::max(n, 1)); z.resize(std::max(n, 1)); }
        first.resize(std::max(n, 1)); span_end.resize(std::max(np, 1));
        for (int r = 0; r < ni; ++r) {           // a velocity / bias row: against the velocity / bias columns before it
            const double* row = Hi + (size_t)(np + r) * n + np;
            int f = r;
            for (int c = 0; c < r; ++c) if (row[c] != 0.0) { f = c; break; }
            first[r] = f;
        }
        // a pose row's entries against the velocity / bias columns are Hi[velocity / bias row][pose column]: the rows of Hi are walked along, not across
        for (int r = 0; r < np; ++r) { first[ni + r] = ni; span_end[r] = 0; }
        for (int c = 0; c < ni; ++c) {
            const double* row = Hi + (size_t)(np + c) * n;
            for (int r = 0; r < np; ++r)
                if (row[r] != 0.0) { if (c < first[ni + r]) first[ni + r] = c; span_end[r] = c + 1; }
        }
        for (int r = 0; r < np; ++r) span_end[r] = std::max(span_end[r], first[ni + r]);
    }
    // For the solve on the device (k_lvi_solve*, ba_kernels.hip): the inertial / LiDAR part of the matrix as the kernel reads it --
    //   hband [ni][32]: the velocity / bias rows at a fixed width, entry (i, c) at 32 i + (c - (i - 31)), zero outside the envelope;
    //   per pose row r: span_first / span_end [np] = the columns of its entries against the velocity / bias unknowns (what lies between the span's end
    //   and the pose block is fill: zero in this part), rowoff [np + 1], and at hpose + rowoff[r] the span's entries followed by the row's r + 1
    //   entries of the pose block.
    // band(): the widest velocity / bias row (i - first[i]).  Returns the number of doubles written to hpose.
    int band() const { int b = 0; for (int i = 0; i < ni; ++i) b = std::max(b, i - first[i]); return b; }
    size_t pack_for_device(const double* Hi, int32_t* span_first, int32_t* span_end_out, int32_t* rowoff, double* hband, double* hpose) const {
        for (int i = 0; i < ni; ++i) {
            const double* src = Hi + (size_t)(np + i) * n + np;
            for (int t = 0; t < 32; ++t) {
                const int c = i - 31 + t;
                hband[(size_t)32 * i + t] = c >= first[i] && c >= 0 ? src[c] : 0.0;
            }
        }
        size_t at = 0;
        for (int r = 0; r < np; ++r) {
            span_first[r] = first[ni + r]; span_end_out[r] = span_end[r]; rowoff[r] = (int32_t)at;
            at += (size_t)(span_end[r] - first[ni + r]);
            const double* hp = Hi + (size_t)r * n;
            for (int c = 0; c <= r; ++c) hpose[at++] = hp[c];
        }
        rowoff[np] = (int32_t)at;
        for (int c = 0; c < ni; ++c) {           // the spans' entries, the rows of Hi walked along
            const double* row = Hi + (size_t)(np + c) * n;
            for (int r = 0; r < np; ++r)
                if (c >= first[ni + r] && c < span_end[r]) hpose[rowoff[r] + (c - first[ni + r])] = row[r];
        }
        return at;
    }
    // Per trial: M = [S + Hi(poses) | Hi(poses, imu); . | Hi(imu) + lambda I] in the solver's order, factorised; false as ldlt_solve_small (a zero or
    // non-finite pivot).  S: [np][np] with the damping already on its diagonal (k_ba_schur_finish).
    bool factorise(const double* Hi, const double* S, double lambda);
    // x (the caller's numbering) from rhs (the caller's numbering)
    void solve(const double* rhs, double* x);
};

}  // namespace tc2li
