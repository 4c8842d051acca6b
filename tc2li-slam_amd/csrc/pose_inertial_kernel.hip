// Optimizer::PoseInertialOptimizationLastKeyFrame / LastFrame (SF/src/Optimizer.cc:2469-2852, 2854-3270) as ONE workgroup per frame:
// the four classification rounds of ten Gauss-Newton iterations each (g2o OptimizationAlgorithmGaussNewton + LinearSolverDense,
// core/optimization_algorithm_gauss_newton.cpp:49-93, solvers/linear_solver_dense.h:65-113) run inside the kernel.  Per iteration the
// unary visual edges (EdgeMonoOnlyPose / EdgeStereoOnlyPose, G2oTypes.cc:384-463) are spread over the 256 threads and their 6 x 6
// normal equations reduced through LDS in a fixed order; the inertial edge, the two random-walk edges and (previous-frame form) the
// prior edge are evaluated by single lanes and expanded into the dense 15 / 30-unknown system by all threads; the LDL^T runs column
// by column with one lane per row.  No host round trip until the frame is done; the Hessian of the new prior is assembled by the
// host from the final states (pose_inertial_host.cpp).
#include <hip/hip_runtime.h>

#include "launch.hpp"
#pragma clang fp contract(off)
#include <stdint.h>

#include <cstdlib>

#include "ba_math.hpp"
#include "pose_inertial_device.hpp"
#include "wave_reduce.hpp"

namespace tc2li {

constexpr int kPiThreads = 256;
constexpr int kPiRed = 28;  // 21 upper-triangular H entries + 6 b entries + one counter

__device__ __forceinline__ void pi_block_reduce(double (&v)[kPiRed], double* s_red /*[4][kPiRed]*/, double* s_out /*[kPiRed]*/) {
    const int lane = threadIdx.x & 63, wave = wave_in_block();
    {   // (wave_reduce.hpp: the xor butterfly's sums with a sixth of its shuffles)
        const double x = wave_reduce_32(v);
        const int k = wave_reduce_index(lane);
        if ((lane & 1) == 0 && k < kPiRed) s_red[wave * kPiRed + k] = x;
    }
    __syncthreads();
    if (threadIdx.x < kPiRed) s_out[threadIdx.x] = (s_red[threadIdx.x] + s_red[kPiRed + threadIdx.x]) + (s_red[2 * kPiRed + threadIdx.x] + s_red[3 * kPiRed + threadIdx.x]);
    __syncthreads();
}

// chi2 of one visual edge at pose T; B = d error / d pose increment when wanted
__device__ __forceinline__ double pi_visual(const ImuPose& T, const ImuCalib& cal, const double* X, const BaEdge& e, const CameraD& cam, double err[3], double* B) {
    double Xc[3], A[9];
    const int dim = imu_edge_error(T, X, e, cam, Xc, err);
    double c2 = 0;
    for (int d = 0; d < dim; ++d) c2 += err[d] * e.info * err[d];
    if (B) imu_edge_jacobians(T, cal, Xc, dim == 3, cam, A, B);
    return c2;
}

// LinearSolverDense's LDL^T and its two substitutions for the N x N system in LDS, by ONE wavefront with the matrix in registers: lane r
// holds row r of the lower triangle, an entry of another row comes through v_readlane at compile-time (row, column) -- no LDS round trip and
// no barrier per dependent step (the LDS form, one lane for the pivots and the substitutions, was 125 k + 115 k of an iteration's 290 k cycles
// at N = 30: every step of its 4 x 435-step chains waited for an LDS read).  The operations and their order are those of the LDS form --
// d_j = H_jj - sum_k (L_jk L_jk) D_k, L_ij = (H_ij - sum_k (L_ik L_jk) D_k) / D_j, k ascending; forward, division, backward substitution with
// k ascending -- so the bits are the same.  Returns whether every pivot was positive (then x holds the solution), the same in every lane.
__device__ __forceinline__ double pi_readlane(double v, int lane /* compile-time constant after unrolling */) {
    union { double d; int i[2]; } u;
    u.d = v;
    u.i[0] = __builtin_amdgcn_readlane(u.i[0], lane);
    u.i[1] = __builtin_amdgcn_readlane(u.i[1], lane);
    return u.d;
}
template <int N>
__device__ __forceinline__ bool pi_ldlt_solve_wave(const double* __restrict__ s_H, const double* __restrict__ s_b, double* __restrict__ s_x) {
    const int lane = threadIdx.x & 63;
    double row[N];
#pragma unroll
    for (int k = 0; k < N; ++k) row[k] = lane < N ? s_H[lane * N + k] : 0.0;
    // Factorisation, right-looking: after column k has its pivot and its entries L_ik = a_ik / D_k, every remaining entry (i, j), j > k, takes
    // its term of column k -- a_ij -= (L_ik L_jk) D_k -- so an entry collects its terms for k = 0, 1, ... in the order the row-by-row sums of
    // the LDS form took them (same operands, same order: same bits), but the (N - 1 - k) updates of a step are independent instructions.
    bool ok = true;
    double my_d = 1.0;  // lane i: D_i
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const double dk = pi_readlane(row[k], k);
        if (!(dk > 0.0) || !(dk - dk == 0.0)) ok = false;
        if (lane == k) my_d = dk;
        row[k] = row[k] / dk;  // lanes above k: L_ik
#pragma unroll
        for (int j = k + 1; j < N; ++j) row[j] -= row[k] * pi_readlane(row[k], j) * dk;  // (lanes below j update entries nothing reads)
    }
    if (!ok) return false;
    // Forward substitution, column by column over all rows at once: z_k is final when the columns before k have been applied; row i then
    // takes its term L_ik z_k -- k ascending, as the row-by-row sum did.  Then the division by D.
    double z = lane < N ? s_b[lane] : 0.0;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const double zk = pi_readlane(z, k);
        if (lane > k) z -= row[k] * zk;
    }
    const double y_mine = z / my_d;
    // Backward substitution x_i = y_i - sum_{k > i} L_ki x_k.  Rounds 3-4 kept the LDS form's order of the terms (k ascending), which makes
    // row i wait for x_{i + 1} first: a chain of N (N - 1) / 2 dependent steps, 25 k of an iteration's 105 k cycles at N = 30.  Round 5: the
    // column sweep -- once x_k is final every row above takes its term L_ki x_k (k DESCENDING per row: the sums' order changes, the results
    // agree with the oracle at its tolerance as before) -- N steps.  Lane i needs column i of L below the diagonal, which lives in the lanes
    // below as row entries: transposed once through the matrix's LDS block (this wavefront owns it until the workgroup's next barrier).
    double* s_Lt = const_cast<double*>(s_H);
#pragma unroll
    for (int i = 0; i < N; ++i) if (lane < N) s_Lt[i * N + lane] = row[i];  // lane k, entry (k, i): read by lane i as its coefficient of x_k
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    double col[N];
#pragma unroll
    for (int k = 0; k < N; ++k) col[k] = lane < N ? s_Lt[lane * N + k] : 0.0;
    double x = lane < N ? y_mine : 0.0;
#pragma unroll
    for (int k = N - 1; k >= 1; --k) {
        const double xk = pi_readlane(x, k);
        if (lane < k) x -= col[k] * xk;
    }
    if (lane < N) s_x[lane] = x;
    return true;
}

__device__ __forceinline__ void pose_inertial_body(const PiProblem* __restrict__ probs, const double* __restrict__ Xw, const BaEdge* __restrict__ edges,
                                                   const uint8_t* __restrict__ close_flags, const ImuCalib& cal, const CameraD& cam, uint8_t* __restrict__ outlier,
                                                   double* __restrict__ chi2_scratch, PiResult* __restrict__ results) {
    __shared__ double s_red[4 * kPiRed], s_sum[kPiRed];
    __shared__ double s_H[900], s_b[30], s_x[30];
    __shared__ double s_J[216], s_T[216], s_e[9], s_Oe[9];        // inertial edge: J, Omega J, error, Omega error
    __shared__ double s_pJ[225], s_pT[225], s_pe[15], s_pOe[15];  // prior edge
    __shared__ double s_r1;                                        // Huber weight of the prior edge
    __shared__ PiState s_cur, s_oth;
    __shared__ int s_flag[2];  // [0] the factorisation found only positive pivots, [1] solver failed at least once
    const PiProblem& pr = probs[blockIdx.x];
    const int tid = threadIdx.x, N = pr.n_edges, last = pr.last_frame, n = last ? 30 : 15;
    const BaEdge* E = edges + pr.edge_off;
    const double* X = Xw + 3 * (size_t)pr.edge_off;
    const uint8_t* cl = close_flags + pr.edge_off;
    uint8_t* out = outlier + pr.edge_off;
    double* chi2 = chi2_scratch + pr.edge_off;

    // The pre-integration record and the prior are read in every one of the 40 iterations -- by the one lane that forms the inertial edge,
    // by the lanes that multiply with the information matrices -- and came from global memory each time (a chain of L2 round trips on
    // single lanes): staged once, 3 KB.  Same values, same operations.
    __shared__ PiPreint s_pre;
    __shared__ PiPrior s_prior;
    static_assert(sizeof(PiPreint) % 4 == 0 && sizeof(PiPrior) % 4 == 0, "copied as dwords");
    for (int k = tid; k < (int)(sizeof(PiPreint) / 4); k += kPiThreads) reinterpret_cast<uint32_t*>(&s_pre)[k] = reinterpret_cast<const uint32_t*>(&pr.pre)[k];
    if (last) for (int k = tid; k < (int)(sizeof(PiPrior) / 4); k += kPiThreads) reinterpret_cast<uint32_t*>(&s_prior)[k] = reinterpret_cast<const uint32_t*>(&pr.prior)[k];
    for (int i = tid; i < N; i += kPiThreads) out[i] = 0;
    if (tid == 0) { s_cur = pr.cur; s_oth = pr.other; s_flag[1] = 0; }
    if (tid < 30) s_x[tid] = 0;
    const double d_mono = (double)sqrtf(5.991f), d_stereo = (double)sqrtf(7.815f);
    const float dsqr_mono = (float)(d_mono * d_mono), dsqr_stereo = (float)(d_stereo * d_stereo);
    const float chi2Mono_kf[4] = {12.f, 7.5f, 5.991f, 5.991f}, chi2Stereo[4] = {15.6f, 9.8f, 7.815f, 7.815f};
    __syncthreads();

    bool robust = true;
    int n_bad = 0, n_inl = 0;
    const int n_graph_edges = N + 3 + (last ? 1 : 0);
    for (int round = 0; round < 4; ++round) {
        bool ok = true;
        for (int it = 0; it < 10 && ok; ++it) {
            // ---- computeActiveErrors + the visual part of buildSystem ----
            const ImuPose T = s_cur.P;
            double acc[kPiRed];
#pragma unroll
            for (int k = 0; k < kPiRed; ++k) acc[k] = 0;
            // Round 5: the inertial edge (one lane, ~21 k cycles: pre-integration update, two SO3 logarithms / Jacobians) and the prior edge do
            // not depend on the visual edges -- wavefront 3 and wavefront 2 form them WHILE wavefronts 0 and 1 walk the correspondences (about
            // the same time at ~1200 correspondences), instead of after them.
            if (tid == 192) pi_inertial_edge(s_pre, s_oth, s_cur, s_e, s_J);
            if (tid == 128 && last) {
                double Jr[9], Jt[9];
                pi_prior_edge(s_prior, s_oth, s_pe, Jr, Jt);
                for (int k = 0; k < 225; ++k) s_pJ[k] = 0;
                for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { s_pJ[15 * r + c] = Jr[3 * r + c]; s_pJ[15 * (3 + r) + 3 + c] = Jt[3 * r + c]; }
                for (int k = 6; k < 15; ++k) s_pJ[15 * k + k] = 1.0;
            }
            for (int i = tid; i < N && tid < 128; i += 128) {
                if (out[i]) continue;  // level 1
                const BaEdge e = E[i];
                double err[3], B[18];
                const double c2 = pi_visual(T, cal, X + 3 * i, e, cam, err, B);
                chi2[i] = c2;
                const bool stereo = e.ur >= 0;
                double rho0 = c2, rho1 = 1.0;
                if (robust) huber(c2, stereo ? d_stereo : d_mono, stereo ? dsqr_stereo : dsqr_mono, rho0, rho1);
                const double w = rho1 * e.info;
                int h = 0;
#pragma unroll
                for (int r = 0; r < 6; ++r) {
#pragma unroll
                    for (int c = r; c < 6; ++c) {
                        double s = 0;
#pragma unroll
                        for (int d = 0; d < 3; ++d) s += B[6 * d + r] * w * B[6 * d + c];  // rows beyond the edge's dimension are zero
                        acc[h++] += s;
                    }
                }
#pragma unroll
                for (int r = 0; r < 6; ++r) {
                    double s = 0;
#pragma unroll
                    for (int d = 0; d < 3; ++d) s += B[6 * d + r] * (e.info * err[d]);
                    acc[21 + r] -= rho1 * s;
                }
            }
            pi_block_reduce(acc, s_red, s_sum);
            for (int k = tid; k < n * n; k += kPiThreads) s_H[k] = 0;
            if (tid < n) s_b[tid] = 0;
            __syncthreads();
            if (tid < 36) {
                const int r = tid / 6, c = tid % 6, lo = r < c ? r : c, hi = r < c ? c : r;
                s_H[r * n + c] = s_sum[lo * 6 - lo * (lo - 1) / 2 + (hi - lo)];
            }
            if (tid < 6) s_b[tid] = s_sum[21 + tid];
            // (the inertial and the prior edge were formed beside the visual edges above)
            __syncthreads();
            if (tid < 216) {  // T = Omega J
                const int r = tid / 24, c = tid % 24;
                double s = 0;
                for (int k = 0; k < 9; ++k) s += s_pre.info[9 * r + k] * s_J[24 * k + c];
                s_T[tid] = s;
            } else if (tid < 225) {
                const int r = tid - 216;
                double s = 0;
                for (int k = 0; k < 9; ++k) s += s_pre.info[9 * r + k] * s_e[k];
                s_Oe[r] = s;
            }
            if (last && tid < 225) {  // prior: T2 = H_prior J
                const int r = tid / 15, c = tid % 15;
                double s = 0;
                for (int k = 0; k < 15; ++k) s += s_prior.H[15 * r + k] * s_pJ[15 * k + c];
                s_pT[tid] = s;
            }
            if (last && tid >= 225 && tid < 240) {
                const int r = tid - 225;
                double s = 0;
                for (int k = 0; k < 15; ++k) s += s_prior.H[15 * r + k] * s_pe[k];
                s_pOe[r] = s;
            }
            __syncthreads();
            // inertial edge into the system: columns P1 V1 G1 A1 (the other state: unknowns 15.. only in the previous-frame form) | P2 V2
            for (int k = tid; k < 576; k += kPiThreads) {
                const int i = k / 24, j = k % 24;
                const int ci = i < 15 ? (last ? 15 + i : -1) : i - 15, cj = j < 15 ? (last ? 15 + j : -1) : j - 15;
                if (ci < 0 || cj < 0) continue;
                double h = 0;
                for (int r = 0; r < 9; ++r) h += s_J[24 * r + i] * s_T[24 * r + j];
                s_H[ci * n + cj] += h;
            }
            if (tid < 24) {
                const int ci = tid < 15 ? (last ? 15 + tid : -1) : tid - 15;
                if (ci >= 0) {
                    double g = 0;
                    for (int r = 0; r < 9; ++r) g += s_J[24 * r + tid] * s_Oe[r];
                    s_b[ci] -= g;
                }
            }
            if (tid == 0 && last) {  // Huber weight of the prior edge (delta 5)
                double c = 0;
                for (int k = 0; k < 15; ++k) c += s_pe[k] * s_pOe[k];
                double r0, r1;
                huber(c, 5.0, 25.0f, r0, r1);
                s_r1 = r1;
            }
            __syncthreads();
            // EdgeGyroRW / EdgeAccRW: error = frame - other, Jacobians -I / +I
            if (tid < 18) {
                const int which = tid / 9, r = (tid % 9) / 3, c = tid % 3;
                const double* info = which ? s_pre.infoA : s_pre.infoG;
                const int ic = which ? 12 : 9, io = last ? 15 + ic : -1;
                const double v = info[3 * r + c];
                s_H[(ic + r) * n + ic + c] += v;
                if (io >= 0) { s_H[(io + r) * n + io + c] += v; s_H[(ic + r) * n + io + c] -= v; s_H[(io + r) * n + ic + c] -= v; }
                if (c == 0) {
                    const double* c2 = which ? s_cur.ba : s_cur.bg;
                    const double* c1 = which ? s_oth.ba : s_oth.bg;
                    const double Oe = info[3 * r] * (c2[0] - c1[0]) + info[3 * r + 1] * (c2[1] - c1[1]) + info[3 * r + 2] * (c2[2] - c1[2]);
                    s_b[ic + r] -= Oe;
                    if (io >= 0) s_b[io + r] += Oe;
                }
            }
            __syncthreads();
            if (last) {
                if (tid < 225) {
                    const int i = tid / 15, j = tid % 15;
                    double h = 0;
                    for (int r = 0; r < 15; ++r) h += s_pJ[15 * r + i] * s_pT[15 * r + j];
                    s_H[(15 + i) * n + 15 + j] += s_r1 * h;
                } else if (tid < 240) {
                    const int i = tid - 225;
                    double g = 0;
                    for (int r = 0; r < 15; ++r) g += s_pJ[15 * r + i] * s_pOe[r];
                    s_b[15 + i] -= s_r1 * g;
                }
                __syncthreads();
            }
            // ---- LinearSolverDense: LDL^T, usable only when every pivot is positive (pi_ldlt_solve_wave: wavefront 0, matrix in registers) ----
            if (tid < 64) {
                const bool pivots_ok = last ? pi_ldlt_solve_wave<30>(s_H, s_b, s_x) : pi_ldlt_solve_wave<15>(s_H, s_b, s_x);
                if (tid == 0) {
                    s_flag[0] = pivots_ok ? 1 : 0;
                    if (!pivots_ok) s_flag[1] = 1;  // the increment of the previous iteration is applied once more and the round ends (GaussNewton::solve)
                }
            }
            __syncthreads();
            if (tid == 0 || (last && tid == 64)) {  // the two states by two wavefronts at the same time, each on a private copy (not through LDS references)
                const int o = tid == 0 ? 0 : 15;
                PiState st = tid == 0 ? s_cur : s_oth;
                double x15[15];
                for (int k = 0; k < 15; ++k) x15[k] = s_x[o + k];
                imu_pose_update(st.P, cal, x15);
                for (int k = 0; k < 3; ++k) { st.v[k] += x15[6 + k]; st.bg[k] += x15[9 + k]; st.ba[k] += x15[12 + k]; }
                if (tid == 0) s_cur = st; else s_oth = st;
            }
            __syncthreads();
            ok = s_flag[0] != 0;
            __syncthreads();
        }
        // ---- classification: active edges keep the chi2 of the last computeActiveErrors (the estimate BEFORE the round's last update),
        // outliers are recomputed at the final estimate (:2680-2683, :3097-3100) ----
        const ImuPose T = s_cur.P;
        const float chiM = last ? 5.991f : chi2Mono_kf[round];
        const float chi2close = 1.5f * chiM;
        double cnt[kPiRed];
#pragma unroll
        for (int k = 0; k < kPiRed; ++k) cnt[k] = 0;
        for (int i = tid; i < N; i += kPiThreads) {
            const BaEdge e = E[i];
            if (out[i]) { double err[3]; chi2[i] = pi_visual(T, cal, X + 3 * i, e, cam, err, nullptr); }
            const float c = (float)chi2[i];
            bool bad;
            if (e.ur >= 0) {
                bad = c > chi2Stereo[round];
            } else {
                const double* Xp = X + 3 * i;
                const bool depth_pos = (T.Rcw[6] * Xp[0] + T.Rcw[7] * Xp[1] + T.Rcw[8] * Xp[2]) + T.tcw[2] > 0.0;
                bad = (c > chiM && !cl[i]) || (cl[i] && c > chi2close) || !depth_pos;
            }
            out[i] = bad ? 1 : 0;
            cnt[bad ? 0 : 1] += 1.0;
        }
        pi_block_reduce(cnt, s_red, s_sum);
        n_bad = (int)s_sum[0];
        n_inl = (int)s_sum[1];
        if (round == 2) robust = false;
        if (n_graph_edges < 10) break;
    }
    // ---- "recover not too bad points" (:2738-2765, :3160-3188) ----
    const ImuPose T = s_cur.P;
    if (n_inl < 30 && !pr.rec_init) {
        double cnt[kPiRed];
#pragma unroll
        for (int k = 0; k < kPiRed; ++k) cnt[k] = 0;
        for (int i = tid; i < N; i += kPiThreads) {
            const BaEdge e = E[i];
            double err[3];
            const double c2 = pi_visual(T, cal, X + 3 * i, e, cam, err, nullptr);
            chi2[i] = c2;
            if ((float)c2 < (e.ur >= 0 ? 24.f : 18.f)) out[i] = 0; else cnt[0] += 1.0;
        }
        pi_block_reduce(cnt, s_red, s_sum);
        n_bad = (int)s_sum[0];
    }
    // ---- sum over the inlier edges of B^T Omega B at the final estimate (GetHessian(): no robust weight) ----
    {
        double acc[kPiRed];
#pragma unroll
        for (int k = 0; k < kPiRed; ++k) acc[k] = 0;
        for (int i = tid; i < N; i += kPiThreads) {
            if (out[i]) continue;
            const BaEdge e = E[i];
            double err[3], B[18];
            pi_visual(T, cal, X + 3 * i, e, cam, err, B);
            int h = 0;
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int c = r; c < 6; ++c) {
                    double s = 0;
#pragma unroll
                    for (int d = 0; d < 3; ++d) s += B[6 * d + r] * e.info * B[6 * d + c];
                    acc[h++] += s;
                }
        }
        pi_block_reduce(acc, s_red, s_sum);
    }
    PiResult& R = results[blockIdx.x];
    if (tid < 21) R.Hv[tid] = s_sum[tid];
    if (tid == 0) {
        R.cur = s_cur; R.other = s_oth;
        R.n_bad = n_bad; R.n_inliers = n_inl; R.solver_failed = s_flag[1]; R.pad_ = 0;
    }
}

// Two kernels of the same body.  k_pose_inertial: the registers the body wants (256 + 28): one wavefront per SIMD, the shortest chain for a
// frame -- what a single sequence sees.  k_pose_inertial_batch: held to two wavefronts per SIMD (amdgpu_waves_per_eu: 256 registers, 100 B of
// scratch per lane), so that two workgroups share a CU and the 512 frames of a batch are ONE round of 2.6 ms instead of two of 2.3
// (4.6 -> 2.7 ms per 512 frames with the tracking stages alone; a lone frame is ~8 % slower in this form).
__global__ __launch_bounds__(kPiThreads) void k_pose_inertial(const PiProblem* __restrict__ probs, const double* __restrict__ Xw, const BaEdge* __restrict__ edges,
                                                            const uint8_t* __restrict__ close_flags, ImuCalib cal, CameraD cam, uint8_t* __restrict__ outlier,
                                                            double* __restrict__ chi2_scratch, PiResult* __restrict__ results) {
    pose_inertial_body(probs, Xw, edges, close_flags, cal, cam, outlier, chi2_scratch, results);
}
__global__ __launch_bounds__(kPiThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_pose_inertial_batch(
    const PiProblem* __restrict__ probs, const double* __restrict__ Xw, const BaEdge* __restrict__ edges, const uint8_t* __restrict__ close_flags, ImuCalib cal,
    CameraD cam, uint8_t* __restrict__ outlier, double* __restrict__ chi2_scratch, PiResult* __restrict__ results) {
    pose_inertial_body(probs, Xw, edges, close_flags, cal, cam, outlier, chi2_scratch, results);
}

void launch_pose_inertial(const PiProblem* probs, int n, const double* Xw, const BaEdge* edges, const uint8_t* close, const ImuCalib& cal,
                          const CameraD& cam, uint8_t* outlier, double* chi2_scratch, PiResult* results, hipStream_t st) {
    if (n <= 0) return;
    static const bool kNoBatchForm = getenv("TC2LI_PI_BATCH_KERNEL") && atoi(getenv("TC2LI_PI_BATCH_KERNEL")) == 0;  // A/B
    if (n > 256 && !kNoBatchForm) TC2LI_LAUNCH(k_pose_inertial_batch, dim3(n), dim3(kPiThreads), 0, st, probs, Xw, edges, close, cal, cam, outlier, chi2_scratch, results);
    else TC2LI_LAUNCH(k_pose_inertial, dim3(n), dim3(kPiThreads), 0, st, probs, Xw, edges, close, cal, cam, outlier, chi2_scratch, results);
}

}  // namespace tc2li
