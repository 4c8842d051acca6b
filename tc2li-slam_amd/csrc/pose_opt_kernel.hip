// Optimizer::PoseOptimization (SF/src/Optimizer.cc:816-1116) as ONE persistent workgroup per frame: the four
// outlier-classification rounds, each a g2o Levenberg-Marquardt run of up to 10 iterations (optimization_algorithm_
// levenberg.cpp:61-169) on a single 6-dof vertex, execute inside the kernel -- no host round trip per iteration.
// Edges are spread over the 256 threads; the 6x6 normal equations and the robust chi2 are reduced through LDS in a
// fixed order (deterministic), thread 0 does the 6x6 LDL^T and the accept/reject logic and broadcasts the decision.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "launch.hpp"
#pragma clang fp contract(off)
#include <stdint.h>

#include "ba_math.hpp"
#include "pose_opt_device.hpp"
#include "wave_reduce.hpp"

namespace tc2li {

constexpr int kPoThreads = 256;
constexpr int kRed = 28;  // 21 upper-triangular H entries + 6 b entries + chi

__device__ __forceinline__ void block_reduce(double (&v)[kRed], double* s_red /*[4][kRed]*/, double* s_out /*[kRed]*/) {
    const int lane = threadIdx.x & 63, wave = wave_in_block();
    {   // the wavefront's sums: the xor butterfly's, a lane keeping only the values it answers for (wave_reduce.hpp: same bits, a sixth of the shuffles)
        const double x = wave_reduce_32(v);
        const int k = wave_reduce_index(lane);
        if ((lane & 1) == 0 && k < kRed) s_red[wave * kRed + k] = x;
    }
    __syncthreads();
    if (threadIdx.x < kRed) s_out[threadIdx.x] = (s_red[threadIdx.x] + s_red[kRed + threadIdx.x]) + (s_red[2 * kRed + threadIdx.x] + s_red[3 * kRed + threadIdx.x]);
    __syncthreads();
}

// sum of ONE value over the block (the trial pass and the outlier count need nothing else: a 28-value reduction there is wasted latency)
__device__ __forceinline__ double block_sum(double x, double* s_red, double* s_out) {
    const int lane = threadIdx.x & 63, wave = wave_in_block();
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) x += __shfl_xor(x, o, 64);
    if (lane == 0) s_red[wave] = x;
    __syncthreads();
    if (threadIdx.x == 0) s_out[0] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
    __syncthreads();
    return s_out[0];
}

// E, X, chi2, out: the frame's N correspondences, their map points, one chi2 and one outlier flag per edge -- in global memory, or staged in
// LDS by the caller (k_pose_optimization_lds)
// (edge_at(i): the i-th correspondence as a BaEdge -- out of global memory, or rebuilt from the four doubles the LDS form keeps of it; Chi2T: double
// in global memory, float in LDS -- the only reader of the stored chi2 is the classification, which compares it as a float: Optimizer.cc:1018-1102)
template <typename EdgeAt, typename Chi2T>
__device__ __forceinline__ void pose_optimization_body(const int N, const EdgeAt edge_at, const double* __restrict__ X, Chi2T* __restrict__ chi2,
                                                       uint8_t* __restrict__ out, const CameraD& cam, double* __restrict__ pose_io, int* __restrict__ inliers) {
    __shared__ double s_red[4 * kRed], s_sum[kRed];
    __shared__ Se3 s_pose, s_trial;
    __shared__ int s_flag[4];  // [0] continue trial loop, [1] accepted, [2] iteration result ok, [3] solve ok
    const int tid = threadIdx.x;

    for (int i = tid; i < N; i += kPoThreads) out[i] = 0;
    if (N < 3) {  // nInitialCorrespondences < 3 (Optimizer.cc:999-1000)
        if (tid == 0) inliers[0] = 0;
        return;
    }
    Se3 initial;
    for (int k = 0; k < 4; ++k) initial.q[k] = pose_io[k];
    for (int k = 0; k < 3; ++k) initial.t[k] = pose_io[4 + k];
    const double d_mono = (double)sqrtf(5.991f), d_stereo = (double)sqrtf(7.815f);
    const float dsqr_mono = (float)(d_mono * d_mono), dsqr_stereo = (float)(d_stereo * d_stereo);
    __syncthreads();

    int n_bad_total = 0;
    bool robust = true;
    for (int round = 0; round < 4; ++round) {
        if (tid == 0) s_pose = initial;  // every round restarts from the frame pose (Optimizer.cc:1012-1013)
        __syncthreads();
        double lambda = 0, ni = 2;  // thread 0 only
        int n_bad_lm = 0;
        bool ok = true;
        for (int it = 0; it < 10 && ok; ++it) {
            // computeActiveErrors + buildSystem over the level-0 edges
            const Se3 T = s_pose;
            double acc[kRed];
#pragma unroll
            for (int k = 0; k < kRed; ++k) acc[k] = 0;
            for (int i = tid; i < N; i += kPoThreads) {
                if (out[i]) continue;  // level 1
                const BaEdge e = edge_at(i);
                double p[3], err[3], B[18];
                se3_map(T, X + 3 * i, p);
                const bool stereo = e.ur >= 0;
                const int dim = edge_error(p, e, cam, err);
                double c2 = 0;
                for (int d = 0; d < dim; ++d) c2 += err[d] * e.info * err[d];
                chi2[i] = (Chi2T)c2;
                double rho0 = c2, rho1 = 1.0;
                if (robust) huber(c2, stereo ? d_stereo : d_mono, stereo ? dsqr_stereo : dsqr_mono, rho0, rho1);
                pose_jacobian(p, stereo, true, cam, B);
                const double w = rho1 * e.info;
                int h = 0;
#pragma unroll
                for (int r = 0; r < 6; ++r) {
#pragma unroll
                    for (int c = r; c < 6; ++c) {
                        double s = 0;
#pragma unroll
                        for (int d = 0; d < 3; ++d) s += B[6 * d + r] * w * B[6 * d + c];  // rows beyond dim are zero
                        acc[h++] += s;
                    }
                }
#pragma unroll
                for (int r = 0; r < 6; ++r) {
                    double s = 0;
#pragma unroll
                    for (int d = 0; d < 3; ++d) s += B[6 * d + r] * (e.info * err[d]);  // err[2] = 0 for a monocular edge
                    acc[21 + r] -= rho1 * s;
                }
                acc[27] += rho0;
            }
            block_reduce(acc, s_red, s_sum);
            double H[36], b[6], currentChi = s_sum[27];
            {
                int h = 0;
                for (int r = 0; r < 6; ++r)
                    for (int c = r; c < 6; ++c) { H[6 * r + c] = s_sum[h]; H[6 * c + r] = s_sum[h]; ++h; }
                for (int r = 0; r < 6; ++r) b[r] = s_sum[21 + r];
            }
            const double iniChi = currentChi;
            if (tid == 0 && it == 0) {  // computeLambdaInit: tau * max diagonal
                double mx = 0;
                for (int r = 0; r < 6; ++r) mx = fmax(fabs(H[7 * r]), mx);
                lambda = 1e-5 * mx;
                ni = 2;
                n_bad_lm = 0;
            }
            double rho = 0;
            int qmax = 0;
            for (;;) {
                double x[6] = {0, 0, 0, 0, 0, 0};
                if (tid == 0) {
                    double Hl[36];
                    for (int k = 0; k < 36; ++k) Hl[k] = H[k];
                    for (int r = 0; r < 6; ++r) Hl[7 * r] += lambda;
                    const bool ok2 = ldlt_solve_small(Hl, 6, b, x, true);
                    s_flag[3] = ok2 ? 1 : 0;
                    s_trial = se3_exp_mul(x, s_pose);
                }
                __syncthreads();
                const Se3 Tt = s_trial;
                double chi_trial = 0;
                for (int i = tid; i < N; i += kPoThreads) {
                    if (out[i]) continue;
                    const BaEdge e = edge_at(i);
                    double p[3], err[3];
                    se3_map(Tt, X + 3 * i, p);
                    const bool stereo = e.ur >= 0;
                    const int dim = edge_error(p, e, cam, err);
                    double c2 = 0;
                    for (int d = 0; d < dim; ++d) c2 += err[d] * e.info * err[d];
                    chi2[i] = (Chi2T)c2;
                    double rho0 = c2, rho1 = 1.0;
                    if (robust) huber(c2, stereo ? d_stereo : d_mono, stereo ? dsqr_stereo : dsqr_mono, rho0, rho1);
                    chi_trial += rho0;
                }
                block_sum(chi_trial, s_red, s_sum + 27);
                if (tid == 0) {
                    double tempChi = s_sum[27];
                    if (!s_flag[3]) tempChi = 1.7976931348623157e308;
                    rho = currentChi - tempChi;
                    double scale = 0;
                    for (int r = 0; r < 6; ++r) scale += x[r] * (lambda * x[r] + b[r]);
                    scale += 1e-3;
                    rho /= scale;
                    const bool finite = tempChi - tempChi == 0.0;
                    if (rho > 0 && finite) {
                        double alpha = 1. - pow((2 * rho - 1), 3.0);
                        alpha = fmin(alpha, 2. / 3.);
                        lambda *= fmax(1. / 3., alpha);
                        ni = 2;
                        currentChi = tempChi;
                        s_pose = s_trial;
                    } else {
                        lambda *= ni;
                        ni *= 2;
                    }
                    qmax++;
                    s_flag[0] = (rho < 0 && qmax < 10) ? 1 : 0;
                }
                __syncthreads();
                if (!s_flag[0]) break;
            }
            if (tid == 0) {
                int res_ok = 1;
                if (qmax == 10 || rho == 0) res_ok = 0;
                else {
                    if ((iniChi - currentChi) * 1e3 < iniChi) n_bad_lm++; else n_bad_lm = 0;
                    if (n_bad_lm >= 3) res_ok = 0;
                }
                s_flag[2] = res_ok;
            }
            __syncthreads();
            ok = s_flag[2] != 0;
            __syncthreads();
        }
        // classification (Optimizer.cc:1018-1102): outliers get their error recomputed at the final pose, inliers keep
        // the chi2 the optimiser left behind
        const Se3 T = s_pose;
        int bad = 0;
        for (int i = tid; i < N; i += kPoThreads) {
            const BaEdge e = edge_at(i);
            if (out[i]) {
                double p[3], err[3];
                se3_map(T, X + 3 * i, p);
                const int dim = edge_error(p, e, cam, err);
                double c2 = 0;
                for (int d = 0; d < dim; ++d) c2 += err[d] * e.info * err[d];
                chi2[i] = (Chi2T)c2;
            }
            const float c = (float)chi2[i];
            const float th = e.ur >= 0 ? 7.815f : 5.991f;
            if (c > th) { out[i] = 1; bad++; } else out[i] = 0;
        }
        __syncthreads();
        {
            n_bad_total = (int)block_sum((double)bad, s_red, s_sum);
        }
        if (round == 2) robust = false;
        if (N < 10) break;  // optimizer.edges().size() < 10
    }
    if (tid == 0) {
        // Frame::SetPose(Sophus::SE3<float>): the result is stored in float
        for (int k = 0; k < 4; ++k) pose_io[k] = (double)(float)s_pose.q[k];
        for (int k = 0; k < 3; ++k) pose_io[4 + k] = (double)(float)s_pose.t[k];
        inliers[0] = N - n_bad_total;
    }
}

__global__ __launch_bounds__(kPoThreads) void k_pose_optimization(const PoseProblem* __restrict__ probs, const double* __restrict__ Xw,
                                                                 const BaEdge* __restrict__ edges, CameraD cam,
                                                                 double* __restrict__ poses7, uint8_t* __restrict__ outlier,
                                                                 double* __restrict__ chi2_scratch, int* __restrict__ inliers) {
    const PoseProblem pr = probs[blockIdx.x];
    const BaEdge* E = edges + pr.edge_off;
    pose_optimization_body(pr.n, [E](int i) { return E[i]; }, Xw + 3 * (size_t)pr.edge_off, chi2_scratch + pr.edge_off, outlier + pr.edge_off, cam,
                           poses7 + 7 * (size_t)blockIdx.x, inliers + blockIdx.x);
}

// The same with the frame's correspondences staged in LDS for the whole optimisation: HBM sees each edge and map point once and one outlier
// flag per edge, instead of a re-read of edge, point and chi2 on each of the ~40 linearisations and trial evaluations (26 x the algorithmic
// bytes in round 2's counters).  Same loops, same order of the sums: the same bits.  cap = edges the dynamic LDS block holds (>= every n).
// Round 5: 61 bytes per correspondence instead of 73 -- observation and information as four doubles (the edge's two indices are not used here),
// the chi2 as the float the classification compares -- so that a frame of up to 1 311 correspondences leaves room for a second workgroup on its CU
// (the batched frames hold ~1 200: one workgroup per CU at 73 bytes, i.e. two rounds of 256 for 512 frames).
constexpr int kPoLdsPerEdge = 4 * sizeof(double) + 3 * sizeof(double) + sizeof(float) + 1;  // 61 B
__global__ __launch_bounds__(kPoThreads) void k_pose_optimization_lds(const PoseProblem* __restrict__ probs, const double* __restrict__ Xw,
                                                                     const BaEdge* __restrict__ edges, CameraD cam,
                                                                     double* __restrict__ poses7, uint8_t* __restrict__ outlier,
                                                                     int* __restrict__ inliers, int cap) {
    extern __shared__ double s_po[];
    const PoseProblem pr = probs[blockIdx.x];
    const int N = pr.n, tid = threadIdx.x;
    double* const s_X = s_po;                                  // [cap][3]
    double* const s_E = s_X + 3 * (size_t)cap;                 // [cap][4]: u, v, ur, info
    float* const s_chi2 = reinterpret_cast<float*>(s_E + 4 * (size_t)cap);  // [cap]
    uint8_t* const s_out = reinterpret_cast<uint8_t*>(s_chi2 + cap);
    {
        const double* gx = Xw + 3 * (size_t)pr.edge_off;
        for (int k = tid; k < 3 * N; k += kPoThreads) s_X[k] = gx[k];
        const double* ge = reinterpret_cast<const double*>(edges + pr.edge_off);   // 5 doubles per edge: (point, pose), u, v, ur, info
        for (int k = tid; k < 4 * N; k += kPoThreads) s_E[k] = ge[5 * (k >> 2) + 1 + (k & 3)];
    }
    __syncthreads();
    pose_optimization_body(N, [s_E](int i) { return BaEdge{0, 0, s_E[4 * i], s_E[4 * i + 1], s_E[4 * i + 2], s_E[4 * i + 3]}; }, s_X, s_chi2, s_out, cam,
                           poses7 + 7 * (size_t)blockIdx.x, inliers + blockIdx.x);
    __syncthreads();
    uint8_t* out = outlier + pr.edge_off;
    for (int i = tid; i < N; i += kPoThreads) out[i] = s_out[i];
}

void launch_pose_optimization(const PoseProblem* probs, int nprobs, const double* Xw, const BaEdge* edges, const CameraD& cam,
                              double* poses7, uint8_t* outlier, double* chi2_scratch, int* inliers, int max_edges, hipStream_t st) {
    if (nprobs <= 0) return;
    // the correspondences of a frame in LDS when they fit (2048: 125 KB).  The block is sized from the batch's largest frame, not in two
    // classes: a workgroup lives for the whole optimisation (milliseconds), and what it does not take of its CU's 160 KB the other stages'
    // kernels can (1200 correspondences: 73 KB; <= 1311: two workgroups per CU).  TC2LI_PO_LDS_CLASSES=1: the two classes (A/B).
    static const bool kClasses = getenv("TC2LI_PO_LDS_CLASSES") && atoi(getenv("TC2LI_PO_LDS_CLASSES")) != 0;
    const int cap = kClasses ? (max_edges <= 1024 ? 1024 : 2048) : std::max(64, (max_edges + 63) / 64 * 64);
    if (max_edges <= 2048 && ensure_dynamic_lds((const void*)k_pose_optimization_lds, 2048 * kPoLdsPerEdge + 64)) {
        TC2LI_LAUNCH(k_pose_optimization_lds, dim3(nprobs), dim3(kPoThreads), (size_t)cap * kPoLdsPerEdge + 64, st, probs, Xw, edges, cam, poses7, outlier,
                     inliers, cap);
    } else {
        TC2LI_LAUNCH(k_pose_optimization, dim3(nprobs), dim3(kPoThreads), 0, st, probs, Xw, edges, cam, poses7, outlier, chi2_scratch, inliers);
    }
}

}  // namespace tc2li
