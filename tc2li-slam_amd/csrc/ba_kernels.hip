// gfx950 kernels of the local bundle adjustment (visual part of Optimizer::LocalBundleAdjustment /
// OptimizerWithLidar::LocalLVBundleAdjustment, SF/src/Optimizer.cc:1118, SF/src/OptimizerWithLidar.cc:60), i.e. the
// numerical work of g2o's BlockSolver_6_3 with Schur complement (Thirdparty/g2o/g2o/core/block_solver.hpp:353-607):
//   k_ba_linearize      one thread per edge: error, Huber weight, Jacobians, the edge's blocks of J^T W J and J^T W r
//   k_ba_reduce_all     by workgroup role: Hll, b_l per landmark (CSR, fixed order) | Hpp, b_p per free pose (LDS tree) | robust cost
//   k_ba_maxdiag        largest diagonal entries for computeLambdaInit (first iteration only)
//   k_ba_schur_lean     windows of <= 21 free keyframes: S_part = (W D^-1) W^T block by block on the f64 vector unit
//   k_ba_schur_coef     larger windows, per edge with a free pose: W D^-1 b_l; k_ba_reduce_coef sums them per free pose
//   k_ba_schur_units    larger windows: S_part = (W D^-1) W^T as a block-sparse product with v_mfma_f64_16x16x4_f64, operand panels in LDS
//   k_ba_schur_finish   S = Hpp + lambda I - sum S_part, b_s = b_p - coefficients
//   k_ba_trial_update   by role: x_l = D^-1 (b_l - W^T x_p), trial points, landmark part of the gain-ratio scale | trial poses exp(x_p) * T
//   k_ba_errors         robust chi2 at the trial estimate
//   k_ba_trial_reduce   scale and cost sums
// Launches that only depend on the same earlier results share one grid (a dependent launch costs ~10 us on its own).
// All arithmetic is double precision; reductions run in a fixed order, so results are reproducible run to run.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <cstdlib>

#include "launch.hpp"
#pragma clang fp contract(off)
#include <stdint.h>

#include "ba_device.hpp"
#include "balm_residual_device.hpp"

namespace tc2li {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

// N doubles (N even) from / to a 16-byte aligned address as double2
template <int N>
__device__ __forceinline__ void store_d2(double* __restrict__ dst, const double* v) {
    v2d* o = reinterpret_cast<v2d*>(dst);
#pragma unroll
    for (int q = 0; q < N / 2; ++q) o[q] = v2d{v[2 * q], v[2 * q + 1]};
}
template <int N>
__device__ __forceinline__ void load_d2(const double* __restrict__ src, double* v) {
    const v2d* i = reinterpret_cast<const v2d*>(src);
#pragma unroll
    for (int q = 0; q < N / 2; ++q) { const v2d t = i[q]; v[2 * q] = t.x; v[2 * q + 1] = t.y; }
}

// error of one projection edge at the given estimate: g2o's SE3 vertices (local BA) or ImuCamPose vertices (inertial BA)
// (inertial: pb.inertial, as a compile-time value where a kernel exists per vertex type -- the linearisation -- and read from the record elsewhere)
__device__ __forceinline__ void edge_state(const BaProblemDev& pb, bool inertial, bool trial, const BaEdge& e, double p[3], double err[3], int& dim, double& chi2) {
    const double* X = (trial ? pb.points_trial : pb.points) + 3 * (size_t)e.point;
    if (inertial) {
        dim = imu_edge_error((trial ? pb.iposes_trial : pb.iposes)[e.pose], X, e, pb.cam, p, err);
    } else {
        se3_map((trial ? pb.poses_trial : pb.poses)[e.pose], X, p);
        dim = edge_error(p, e, pb.cam, err);
    }
    chi2 = 0;
    for (int d = 0; d < dim; ++d) chi2 += err[d] * e.info * err[d];
}
__device__ __forceinline__ void edge_state(const BaProblemDev& pb, bool trial, const BaEdge& e, double p[3], double err[3], int& dim, double& chi2) {
    edge_state(pb, pb.inertial != 0, trial, e, p, err, dim, chi2);
}

// sum of v over the workgroup's 256 threads in a fixed order -> out[0]
template <bool DEVICE_SCOPE = false>
__device__ __forceinline__ void block_sum_256(double v, double* s, double* __restrict__ out) {
    s[threadIdx.x] = v;
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {
        if ((int)threadIdx.x < st) s[threadIdx.x] += s[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (DEVICE_SCOPE) __hip_atomic_store(out, s[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else out[0] = s[0];
    }
}

// Partial sums that another workgroup of the SAME launch adds up (ba_last_of below) are written and read at device scope -- the store goes
// through the XCD's L2 to memory (block_sum_256<true>), the load does not take what that L2 holds -- so that no cache-wide write-back /
// invalidate (__threadfence: buffer_wbl2 + buffer_inv by every wavefront of the launch, on everybody's lines) is needed to see them:
// measured, with the fences the BA stage took twice as long (35 against 17 ms per 128 windows).
template <bool DEVICE_SCOPE>
__device__ __forceinline__ double load_partial(const double* p) {
    return DEVICE_SCOPE ? __hip_atomic_load(const_cast<double*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p;
}

// Workgroups [0, nbe): every edge -- error, Huber weight, robust cost and the landmark block A^T W A, A^T omega_r.  Workgroups from
// nbe on: the edges with a free pose in landmark-major order (fl_edge) -- the pose block B^T W B, B^T omega_r and W = B^T W A.  Two
// roles instead of one thread doing both: four edges in ten have a free pose, and the pose part (most of the arithmetic and of the
// bytes) ran with that share of its lanes; its W blocks now leave in slot order.  The error and the weight are simply formed again.
// LDS of the two roles of the linearisation, declared by the kernel (one launch runs both: the roles share it)
struct LinearizeLds { double big[256 * 9]; double small[256]; };
// a partial sum for another workgroup of the same launch (ba_last_of) leaves at device scope; for the next launch a plain store will do
template <bool DEVICE_SCOPE>
__device__ __forceinline__ void store_partial(double* p, double v) {
    if (DEVICE_SCOPE) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
template <bool INERTIAL, bool FUSED = false>
__device__ __forceinline__ void d_ba_linearize_pose(const BaProblemDev& pb, const int bx, LinearizeLds& lds) {
    double* const s_cp = lds.big;
    uint8_t* const s_rows = reinterpret_cast<uint8_t*>(lds.small);
    const int s = bx * 256 + threadIdx.x;
    const bool valid = s < pb.n_free_edges;
    // the per-pose sums below: the block's rows sorted by pose and this thread's range of them, requested with the edge's own loads
    const int* off = pb.blk_off + (size_t)bx * (pb.n_free + 1);
    s_rows[threadIdx.x] = pb.blk_rows[(size_t)bx * 256 + threadIdx.x];
    int my_r0 = 0, my_r1 = 0;
    if ((int)threadIdx.x < 9 * pb.n_free) { my_r0 = off[threadIdx.x / 9]; my_r1 = off[threadIdx.x / 9 + 1]; }
    double B[18], wr[3], w = 0;  // what the pose block is formed from, nine values at a time (below): 28 doubles less to hold
#pragma unroll
    for (int i = 0; i < 18; ++i) B[i] = 0;
    wr[0] = wr[1] = wr[2] = 0;
    if (valid) {
        const int e = pb.fl_edge[s];
        const BaEdge ed = pb.edges[e];
        double p[3], err[3], c2, rho0, rho1;
        int dim;
        edge_state(pb, INERTIAL, false, ed, p, err, dim, c2);
        const bool stereo = ed.ur >= 0;
        huber(c2, stereo ? pb.delta_stereo : pb.delta_mono, stereo ? pb.dsqr_stereo : pb.dsqr_mono, rho0, rho1);
        double A[9];
        if (INERTIAL) {
            imu_edge_jacobians(pb.iposes[ed.pose], pb.calib, p, stereo, pb.cam, A, B);
        } else {
            double R[9];
            quat_to_matrix(pb.poses[ed.pose].q, R);
            point_jacobian(p, R, stereo, pb.cam, A);
            pose_jacobian(p, stereo, false, pb.cam, B);
        }
        w = rho1 * ed.info;
        // omega_r = -rho' * Omega * e
#pragma unroll
        for (int d = 0; d < 3; ++d) wr[d] = d < dim ? -(ed.info * err[d]) * rho1 : 0.0;
        double W[18];  // Hpl block: B^T W A (6 x 3)
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                double sum = 0;
#pragma unroll
                for (int d = 0; d < 3; ++d) sum += B[6 * d + r] * w * A[3 * d + c];
                W[3 * r + c] = sum;
            }
        store_d2<18>(pb.W + 18 * (size_t)s, W);
    }
    // The pose blocks of the workgroup's 256 edges, added per pose in slot order (blk_rows: the block's rows sorted by pose, blk_off
    // the poses' ranges; nine of the 27 values at a time through LDS): what leaves the kernel is one 27-vector per (block, pose)
    // instead of one per edge (224 B written here and read back by the reduction).  Value h < 21 of the 27: entry (r, c >= r) of
    // B^T W B in row-major order of the upper triangle; 21 + r: (B^T omega_r)_r.  The row sums run over all three rows with compile-time
    // indices (everything stays in registers); the third row of A and B is zero for a monocular edge, so its terms add exact zeros.
    const int nf = pb.n_free;
#pragma unroll
    for (int c0 = 0; c0 < 27; c0 += 9) {
        if (valid) {
            int h = 0;
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int c = r; c < 6; ++c) {
                    if (h >= c0 && h < c0 + 9) {
                        double sum = 0;
#pragma unroll
                        for (int d = 0; d < 3; ++d) sum += B[6 * d + r] * w * B[6 * d + c];
                        s_cp[9 * threadIdx.x + (h - c0)] = sum;
                    }
                    ++h;
                }
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                if (21 + r >= c0 && 21 + r < c0 + 9) {
                    double sum = 0;
#pragma unroll
                    for (int d = 0; d < 3; ++d) sum += B[6 * d + r] * wr[d];
                    s_cp[9 * threadIdx.x + (21 + r - c0)] = sum;
                }
            }
        }
        __syncthreads();
        for (int t = threadIdx.x; t < 9 * nf; t += 256) {
            const int i = t / 9, c = t - 9 * i;
            const int r0 = t < 256 ? my_r0 : off[i], r1 = t < 256 ? my_r1 : off[i + 1];
            double acc = 0;
            for (int r = r0; r < r1; ++r) acc += s_cp[9 * s_rows[r] + c];
            store_partial<FUSED>(pb.cp_part + ((size_t)bx * nf + i) * kContribP + c0 + c, acc);
        }
        __syncthreads();
    }
}
template <bool INERTIAL, bool FUSED = false>
__device__ __forceinline__ void d_ba_linearize(const BaProblemDev& pb, const int g, LinearizeLds& lds) {
    double* const s_sum = lds.small;
    double* const s_cl = lds.big;
    const int k0 = pb.grp_k0[g], k1 = pb.grp_k0[g + 1], l0 = pb.grp_l0[g], l1 = pb.grp_l0[g + 1];
    const int k = k0 + (int)threadIdx.x;
    double rho0 = 0;
    if (k < k1) {
        const int e = pb.pt_edges[k];
        const BaEdge ed = pb.edges[e];
        double p[3], err[3], c2;
        int dim;
        edge_state(pb, INERTIAL, false, ed, p, err, dim, c2);
        const bool stereo = ed.ur >= 0;
        double rho1;
        huber(c2, stereo ? pb.delta_stereo : pb.delta_mono, stereo ? pb.dsqr_stereo : pb.dsqr_mono, rho0, rho1);
        pb.chi2[e] = c2;
        pb.rho0[e] = rho0;
        double A[9];
        if (INERTIAL) {
            double B[18];
            imu_edge_jacobians(pb.iposes[ed.pose], pb.calib, p, stereo, pb.cam, A, B);
        } else {
            double R[9];
            quat_to_matrix(pb.poses[ed.pose].q, R);
            point_jacobian(p, R, stereo, pb.cam, A);
        }
        const double w = rho1 * ed.info;
        double wr[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) wr[d] = d < dim ? -(ed.info * err[d]) * rho1 : 0.0;
        // landmark block: A^T W A (upper 6) and A^T omega_r; the row sums run over all three rows with compile-time indices (the third row
        // of A is zero for a monocular edge, so its terms add exact zeros)
        double* cl = s_cl + 9 * threadIdx.x;
        int h = 0;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = r; c < 3; ++c) {
                double sum = 0;
#pragma unroll
                for (int d = 0; d < 3; ++d) sum += A[3 * d + r] * w * A[3 * d + c];
                cl[h++] = sum;
            }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            double sum = 0;
#pragma unroll
            for (int d = 0; d < 3; ++d) sum += A[3 * d + r] * wr[d];
            cl[6 + r] = sum;
        }
    }
    __syncthreads();
    const int l = l0 + (int)threadIdx.x;
    if (l < l1) {
        double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int j = pb.pt_off[l] - k0; j < pb.pt_off[l + 1] - k0; ++j) {
#pragma unroll
            for (int i = 0; i < 9; ++i) acc[i] += s_cl[9 * j + i];
        }
        for (int i = 0; i < 6; ++i) pb.Hll[6 * (size_t)l + i] = acc[i];
        for (int i = 0; i < 3; ++i) pb.bl[3 * (size_t)l + i] = acc[6 + i];
        store_partial<FUSED>(pb.diag_l + l, fmax(fabs(acc[0]), fmax(fabs(acc[3]), fabs(acc[5]))));
    }
    block_sum_256<FUSED>(rho0, s_sum, pb.chi_part + g);  // the robust cost is summed per workgroup here, finished by the closing sums
}
// one launch, both roles: workgroups [0, n_groups) the landmark role, the rest the pose role (they do not depend on each other)
__global__ __launch_bounds__(256) void k_ba_linearize(BaProblemDev pb) {
    __shared__ LinearizeLds lds;
    if (pb.inertial) {
        if ((int)blockIdx.x < pb.n_groups) d_ba_linearize<true>(pb, blockIdx.x, lds);
        else d_ba_linearize_pose<true>(pb, (int)blockIdx.x - pb.n_groups, lds);
    } else {
        if ((int)blockIdx.x < pb.n_groups) d_ba_linearize<false>(pb, blockIdx.x, lds);
        else d_ba_linearize_pose<false>(pb, (int)blockIdx.x - pb.n_groups, lds);
    }
}

// Fixed-order block sum of `width` values per item over the items [begin, end) of an index list.  The tree is the one a 256-entry
// LDS array would be folded with (t += t + 128, t += t + 64, ... , t += t + 1), so the bits do not depend on how it is carried out:
// the two folds that cross wavefronts go through LDS in chunks of kSumChunk values (128 x kSumChunk doubles instead of 256 x WIDTH:
// a 55 KB workgroup had to wait for half a CU's LDS on a busy GPU), the six inside wavefront 0 are shuffles.
constexpr int kSumChunk = 9;
template <int WIDTH, int STRIDE = WIDTH>
__device__ __forceinline__ void block_sum_items(const double* __restrict__ items, const int* __restrict__ index, int begin, int end,
                                                double* s_part /*[128][min(WIDTH, kSumChunk)]*/, double* out) {
    constexpr int CH = WIDTH < kSumChunk ? WIDTH : kSumChunk;
    const int tid = threadIdx.x;
    double acc[WIDTH];
#pragma unroll
    for (int i = 0; i < WIDTH; ++i) acc[i] = 0;
    for (int k = begin + tid; k < end; k += 256) {
        double c[STRIDE];
        load_d2<STRIDE>(items + STRIDE * (size_t)(index ? index[k] : k), c);  // STRIDE is even: records are 16-byte aligned
#pragma unroll
        for (int i = 0; i < WIDTH; ++i) acc[i] += c[i];
    }
#pragma unroll
    for (int c0 = 0; c0 < WIDTH; c0 += CH) {
        __syncthreads();
        if (tid >= 128) {
#pragma unroll
            for (int i = 0; i < CH; ++i) if (c0 + i < WIDTH) s_part[(tid - 128) * CH + i] = acc[c0 + i];
        }
        __syncthreads();
        if (tid < 128) {
#pragma unroll
            for (int i = 0; i < CH; ++i) if (c0 + i < WIDTH) acc[c0 + i] += s_part[tid * CH + i];
        }
        __syncthreads();
        if (tid >= 64 && tid < 128) {
#pragma unroll
            for (int i = 0; i < CH; ++i) if (c0 + i < WIDTH) s_part[(tid - 64) * CH + i] = acc[c0 + i];
        }
        __syncthreads();
        if (tid < 64) {
#pragma unroll
            for (int i = 0; i < CH; ++i) if (c0 + i < WIDTH) acc[c0 + i] += s_part[tid * CH + i];
        }
    }
    if (tid < 64) {
#pragma unroll
        for (int sft = 32; sft >= 1; sft >>= 1) {
#pragma unroll
            for (int i = 0; i < WIDTH; ++i) acc[i] += __shfl_down(acc[i], sft, 64);  // lanes >= sft compute values nobody reads
        }
        if (tid == 0) {
#pragma unroll
            for (int i = 0; i < WIDTH; ++i) out[i] = acc[i];
        }
    }
    __syncthreads();
}

// Hpp, b_p of free pose i: the blocks' partial sums, eight interleaved series of blocks (lanes 32 q + c take the blocks q, q + 8, ...)
// added in series order
__device__ __forceinline__ void reduce_poses_body(const BaProblemDev& pb, int i /* free pose */, double* s_part, double* __restrict__ hpp_out) {
    const int q = threadIdx.x >> 5, c = threadIdx.x & 31, nb = (pb.n_free_edges + 255) / 256;
    double acc = 0;
    if (c < 27)
        for (int b = q; b < nb; b += 8) acc += pb.cp_part[((size_t)b * pb.n_free + i) * kContribP + c];
    s_part[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x < 27) {
        double sum = 0;
#pragma unroll
        for (int u = 0; u < 8; ++u) sum += s_part[32 * u + threadIdx.x];
        pb.Hpp[27 * (size_t)i + threadIdx.x] = sum;
        if (hpp_out) hpp_out[27 * (size_t)i + threadIdx.x] = sum;  // the host's copy (pinned): computeLambdaInit with a LiDAR term reads the diagonal
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double* h = pb.Hpp + 27 * (size_t)i;
        // diagonal entries of the packed upper triangle: 0, 6, 11, 15, 18, 20
        pb.diag_p[i] = fmax(fmax(fabs(h[0]), fabs(h[6])), fmax(fmax(fabs(h[11]), fabs(h[15])), fmax(fabs(h[18]), fabs(h[20]))));
    }
}

// out[0] = in[0] + in[1] + ... (or the maximum) by one 256-thread workgroup, fixed order
template <bool MAX, bool DEVICE_SCOPE = false>
__device__ __forceinline__ void block_reduce_256(const double* __restrict__ in, int n, double* s, double* out) {
    double a = 0;
    for (int k = threadIdx.x; k < n; k += 256) { const double v = load_partial<DEVICE_SCOPE>(in + k); a = MAX ? fmax(a, v) : a + v; }
    s[threadIdx.x] = a;
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {
        if ((int)threadIdx.x < st) s[threadIdx.x] = MAX ? fmax(s[threadIdx.x], s[threadIdx.x + st]) : s[threadIdx.x] + s[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = s[0];
    __syncthreads();
}

// The closing sums of a linearisation by ONE workgroup -- the window's last to deliver (k_ba_linearize*_b, ba_last_of): Hpp / b_p of every free
// pose as reduce_poses_body adds them (eight interleaved series of blocks, the series in order: the same bits), four poses at a time so that
// a lane has up to twenty independent loads in flight instead of one dependent chain per pose; then the robust cost over the groups'
// partials and, when the phase asks for them, the largest diagonal entries.  All partials come from other workgroups of the same launch: read
// at device scope.  s_part: 4 x 256 doubles.
__device__ __forceinline__ void d_ba_linearize_close(const BaProblemDev& pb, double* s_part, double* __restrict__ chi_out, double* __restrict__ hpp_out,
                                                     double* __restrict__ maxdiag_out) {
    const int q = threadIdx.x >> 5, c = threadIdx.x & 31, nb = (pb.n_free_edges + 255) / 256, nf = pb.n_free;
    for (int i0 = 0; i0 < nf; i0 += 4) {
        double a[4] = {0, 0, 0, 0};
        if (c < 27)
            for (int b = q; b < nb; b += 8) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (i0 + u < nf) a[u] += load_partial<true>(pb.cp_part + ((size_t)b * nf + i0 + u) * kContribP + c);
            }
#pragma unroll
        for (int u = 0; u < 4; ++u) s_part[256 * u + threadIdx.x] = a[u];
        __syncthreads();
        if (threadIdx.x < 128) {
            const int u = threadIdx.x >> 5, v = threadIdx.x & 31, i = i0 + u;
            if (v < 27 && i < nf) {
                double sum = 0;
#pragma unroll
                for (int w = 0; w < 8; ++w) sum += s_part[256 * u + 32 * w + v];
                pb.Hpp[27 * (size_t)i + v] = sum;
                if (hpp_out) hpp_out[27 * (size_t)i + v] = sum;
                s_part[256 * u + v] = sum;  // (own column of series 0: read above by this lane only)
            }
        }
        __syncthreads();
        if (threadIdx.x < 4 && i0 + (int)threadIdx.x < nf) {
            const double* h = s_part + 256 * threadIdx.x;
            pb.diag_p[i0 + threadIdx.x] = fmax(fmax(fabs(h[0]), fabs(h[6])), fmax(fmax(fabs(h[11]), fabs(h[15])), fmax(fabs(h[18]), fabs(h[20]))));
        }
        __syncthreads();
    }
    block_reduce_256<false, true>(pb.chi_part, pb.n_groups, s_part, chi_out);
    if (maxdiag_out) {
        block_reduce_256<true, true>(pb.diag_l, pb.n_points, s_part, maxdiag_out);
        block_reduce_256<true, true>(pb.diag_p, pb.n_free, s_part, maxdiag_out + 1);  // written by this workgroup above (behind barriers)
    }
}

// One launch after the linearisation: workgroups [0, n_free) sum the pose blocks, the last one the robust cost
__device__ __forceinline__ void d_ba_reduce_all(const BaProblemDev& pb, const int bx, double* __restrict__ chi_out, double* __restrict__ hpp_out) {
    __shared__ double s_part[128 * kSumChunk];  // block_sum_items' two cross-wavefront folds (also >= the 256 of block_reduce_256)
    if (bx < pb.n_free) reduce_poses_body(pb, bx, s_part, hpp_out);
    else block_reduce_256<false>(pb.chi_part, pb.n_groups, s_part, chi_out);
}
__global__ __launch_bounds__(256) void k_ba_reduce_all(BaProblemDev pb, double* __restrict__ chi_out) { d_ba_reduce_all(pb, blockIdx.x, chi_out, nullptr); }

// computeLambdaInit needs the largest diagonal entries: [0] landmarks, [1] poses (first iteration only)
__device__ __forceinline__ void d_ba_maxdiag(const BaProblemDev& pb, const int bx, double* __restrict__ out) {
    __shared__ double s[256];
    if (bx == 0) block_reduce_256<true>(pb.diag_l, pb.n_points, s, out);
    else block_reduce_256<true>(pb.diag_p, pb.n_free, s, out + 1);
}
__global__ __launch_bounds__(256) void k_ba_maxdiag(BaProblemDev pb, double* __restrict__ out) { d_ba_maxdiag(pb, blockIdx.x, out); }

// Duplicate edges of a (point, free pose) pair (BaProblemDev::dup_*): one thread per free pose adds its duplicates' blocks, in edge order, to
// what the linearisation and its sums left -- W of the pair's slot += B^T W A, Hpp / b_p of the pose += B^T W B / B^T omega_r (the pose role's
// expressions) -- and renews the pose's largest diagonal entry.  g2o adds the edges' blocks to the same Hpl / Hpp entries
// (base_binary_edge.hpp:55-137); a window without duplicates never launches this.
template <bool INERTIAL>
__device__ __forceinline__ void d_ba_dups(const BaProblemDev& pb, double* __restrict__ hpp_out) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= pb.n_free || !pb.n_dups) return;
    const int k0 = pb.dup_off[i], k1 = pb.dup_off[i + 1];
    if (k0 == k1) return;
    double* h = pb.Hpp + 27 * (size_t)i;
    for (int k = k0; k < k1; ++k) {
        const BaEdge ed = pb.edges[pb.dup_edge[k]];
        double p[3], err[3], c2, rho0, rho1, A[9], B[18];
        int dim;
        edge_state(pb, INERTIAL, false, ed, p, err, dim, c2);
        const bool stereo = ed.ur >= 0;
        huber(c2, stereo ? pb.delta_stereo : pb.delta_mono, stereo ? pb.dsqr_stereo : pb.dsqr_mono, rho0, rho1);
        if (INERTIAL) {
            imu_edge_jacobians(pb.iposes[ed.pose], pb.calib, p, stereo, pb.cam, A, B);
        } else {
            double R[9];
            quat_to_matrix(pb.poses[ed.pose].q, R);
            point_jacobian(p, R, stereo, pb.cam, A);
            pose_jacobian(p, stereo, false, pb.cam, B);
        }
        const double w = rho1 * ed.info;
        double wr[3];
        for (int d = 0; d < 3; ++d) wr[d] = d < dim ? -(ed.info * err[d]) * rho1 : 0.0;
        double* Ws = pb.W + 18 * (size_t)pb.dup_slot[k];
        for (int r = 0; r < 6; ++r)
            for (int c = 0; c < 3; ++c) {
                double sum = 0;
                for (int d = 0; d < 3; ++d) sum += B[6 * d + r] * w * A[3 * d + c];
                Ws[3 * r + c] += sum;
            }
        int hh = 0;
        for (int r = 0; r < 6; ++r)
            for (int c = r; c < 6; ++c) {
                double sum = 0;
                for (int d = 0; d < 3; ++d) sum += B[6 * d + r] * w * B[6 * d + c];
                h[hh++] += sum;
            }
        for (int r = 0; r < 6; ++r) {
            double sum = 0;
            for (int d = 0; d < 3; ++d) sum += B[6 * d + r] * wr[d];
            h[21 + r] += sum;
        }
    }
    if (hpp_out) for (int v = 0; v < 27; ++v) hpp_out[27 * (size_t)i + v] = h[v];
    pb.diag_p[i] = fmax(fmax(fabs(h[0]), fabs(h[6])), fmax(fmax(fabs(h[11]), fabs(h[15])), fmax(fabs(h[18]), fabs(h[20]))));
}
__global__ __launch_bounds__(64) void k_ba_dups(BaProblemDev pb) {
    if (pb.inertial) d_ba_dups<true>(pb, nullptr); else d_ba_dups<false>(pb, nullptr);
}

// (Hll + lambda I)^-1 and its product with b_l for landmark l
__device__ __forceinline__ void point_dinv(const BaProblemDev& pb, int l, double lambda, double Di[9], double db[3]) {
    const double* h = pb.Hll + 6 * (size_t)l;
    // D = Hll + lambda I (symmetric: h = [00 01 02 11 12 22]); inverse by cofactors like Eigen's fixed 3x3 inverse
    const double d00 = h[0] + lambda, d01 = h[1], d02 = h[2], d11 = h[3] + lambda, d12 = h[4], d22 = h[5] + lambda;
    const double c00 = d11 * d22 - d12 * d12, c01 = d12 * d02 - d01 * d22, c02 = d01 * d12 - d11 * d02;
    const double det = d00 * c00 + d01 * c01 + d02 * c02, id = 1.0 / det;
    Di[0] = c00 * id; Di[1] = (d02 * d12 - d01 * d22) * id; Di[2] = (d01 * d12 - d02 * d11) * id;
    Di[3] = c01 * id; Di[4] = (d00 * d22 - d02 * d02) * id; Di[5] = (d02 * d01 - d00 * d12) * id;
    Di[6] = c02 * id; Di[7] = (d01 * d02 - d00 * d12) * id; Di[8] = (d00 * d11 - d01 * d01) * id;
    const double* b = pb.bl + 3 * (size_t)l;
    for (int r = 0; r < 3; ++r) db[r] = Di[3 * r] * b[0] + Di[3 * r + 1] * b[1] + Di[3 * r + 2] * b[2];
}

// Dense path, per edge with a free pose: W D^-1 and W (6x3) scattered into the two k-major GEMM operands, W D^-1 b_l.  An edge
// recomputes its landmark's 3x3 inverse (same arithmetic, same value as the back substitution's).
// Dense-window Schur path (windows of more than 21 free keyframes: LocalInertialBA's bLarge window, Optimizer.cc:1516-1523), per slot =
// edge with a free pose: the edge's part of sum_l W D^-1 b_l (k_ba_reduce_coef adds a pose's edges in its edge order).  Rounds 1-4 also
// scattered W D^-1 and W into two dense k-major operands [3 P][np_pad] here (11.5 MB per 25-keyframe window, 29 % non-zero, written
// every trial and read back by the GEMM); round 5 builds the operand panels in LDS instead (d_ba_schur_units).
__device__ __forceinline__ void d_ba_schur_coef(const BaProblemDev& pb, const int bx, double lambda, const bool write_Y) {
    const int s = bx * 256 + threadIdx.x;
    if (s >= pb.n_free_edges) return;
    const int e = pb.fl_edge[s], l = pb.fl_lm[s];
    double W[18];
    load_d2<18>(pb.W + 18 * (size_t)s, W);
    double Di[9], db[3];
    point_dinv(pb, l, lambda, Di, db);
    double ce[6], Y[18];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        ce[r] = W[3 * r] * db[0] + W[3 * r + 1] * db[1] + W[3 * r + 2] * db[2];
#pragma unroll
        for (int k = 0; k < 3; ++k) Y[3 * r + k] = W[3 * r] * Di[k] + W[3 * r + 1] * Di[3 + k] + W[3 * r + 2] * Di[6 + k];  // (W D^-1)[r][k]: the expression of rounds 1-4's prepare kernel
    }
    store_d2<6>(pb.coef_e + 6 * (size_t)e, ce);
    // the product's first operand, slot by slot, for the 64 x 64 units (144 B per slot: they load it instead of inverting again); the full-width form
    // rebuilds it from W and the landmark's block while it stages the slot (d_ba_schur_full): a third of that kernel's bytes less, and none here
    if (write_Y) store_d2<18>(pb.Y + 18 * (size_t)s, Y);
}
__global__ __launch_bounds__(256) void k_ba_schur_coef(BaProblemDev pb, double lambda, int write_Y) { d_ba_schur_coef(pb, blockIdx.x, lambda, write_Y != 0); }

__device__ __forceinline__ void d_ba_reduce_coef(const BaProblemDev& pb, const int bx) {
    __shared__ double s_part[128 * 6];
    const int i = bx;
    block_sum_items<6>(pb.coef_e, pb.pv_edges, pb.pv_off[i], pb.pv_off[i + 1], s_part, pb.coef + 6 * (size_t)i);
}
__global__ __launch_bounds__(256) void k_ba_reduce_coef(BaProblemDev pb) { d_ba_reduce_coef(pb, blockIdx.x); }

// S_part[slice] (np_pad x np_pad, row-major, lower tiles) = sum over the slice's landmarks of (W D^-1) W^T as a block-sparse f64 MFMA
// product (v_mfma_f64_16x16x4_f64).  Fragment layout (checked on gfx950, tools/dbg/mfma_f64_test.hip): A: lane -> A[i = lane % 16][k = lane / 16];
// B: lane -> B[k = lane / 16][j = lane % 16]; D: lane, r -> D[i = lane / 16 + 4 r][j = lane % 16].
//
// Round 5.  One workgroup of four wavefronts per UNIT = 32 x 32 block of S on or below the diagonal (2 x 2 tiles) and
// per slice of landmark CHUNKS (16 landmarks = 48 operand rows).  Per chunk the workgroup builds the two operand panels in LDS straight
// from the landmark-major W blocks -- (W D^-1)^T for the unit's 32 rows, W^T for its 32 columns, [48][32] doubles each, stored as two
// [48][16] halves so that a fragment read (16 consecutive doubles of four consecutive rows) touches every LDS bank once -- and multiplies
// them: the chunk's twelve k-steps are dealt to the four wavefronts (w, w + 4, w + 8), each holding the unit's 2 x 2 accumulator tiles,
// i.e. four independent MFMA chains per wavefront and two operand reads per MFMA pair.  A chunk in which no landmark is seen from the
// unit's rows or from its columns is skipped (chunk_mask, one bit per 16 columns, built by the host), tile by tile inside a unit: the
// product is block-sparse -- a landmark touches the poses that see it -- and the dense form multiplied the zeros (29 % useful).  The
// wavefronts' tiles meet in LDS and are added in wavefront order; k_ba_schur_finish adds the slices in slice order: the bits are a
// function of the window alone.  No dense operands in HBM (11.5 MB per window and trial before), no prepare scatter, 8 partial
// sums per window instead of 64.
constexpr int kUnitChunk = kUnitChunkHost;      // landmarks per chunk (= per slice of slice_off on this path)
constexpr int kUnitRows = 3 * kUnitChunk;       // operand rows per chunk
constexpr int kUnitW = 64;                      // rows = columns of a unit (4 x 4 tiles)
struct UnitPanels { double A[4][kUnitRows][16], B[4][kUnitRows][16]; };   // [16-column quarter][operand row][column]: 48 KB
constexpr int kUnitIdx = 2048;                  // slots of a slice whose (pose, rank in chunk) the workgroup keeps in LDS (beyond: read where they are)
constexpr int kUnitMaxChunks = 64;              // chunks per slice (the host cuts more slices for larger windows)
struct UnitsLds {
    UnitPanels P;
    unsigned short idx[kUnitIdx];
    int off[kUnitMaxChunks + 1];
};
// What a lane holds of ONE slot of the next chunk while the current one is multiplied: the loads of chunk c + 1 are in flight under the
// product of chunk c (without it every chunk was three dependent trips to memory -- offsets, indices, blocks -- with the matrix pipe idle).
struct UnitSlot { int p, ca, cb; bool inA, inB; double Y[18], W[18]; };
// Unit = 64 x 64 block of S (second form of the round: 32 x 32 units re-read a chunk's blocks once per unit that touches it and gave the matrix
// pipe twelve MFMAs per wavefront and chunk between two trips to memory -- 0.20 of the MFMA peak alone; a 64 x 64 unit multiplies four times
// the tiles per block loaded).  Wavefront w owns tile ROW w of the unit -- its four accumulator tiles, all twelve k-steps of a chunk: one A
// fragment and four B fragments per four MFMAs, and no sum across wavefronts at the end.
__device__ __forceinline__ void d_ba_schur_units(const BaProblemDev& pb, const int unit, const int slice, const int chunks_per_slice, const double lambda,
                                                 UnitsLds& L) {
    (void)lambda;  // (the operands W D^-1 come from k_ba_schur_coef of the same trial)
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_in_block();
    const int tiles = pb.np_pad / 16, n_chunks = pb.n_schur_slices;
    int bi = 0;
    while ((bi + 1) * (bi + 2) / 2 <= unit) ++bi;
    const int bj = unit - bi * (bi + 1) / 2;
    const int ti = 4 * bi + wave, tj0 = 4 * bj;   // this wavefront's tile row; the unit's first tile column
    v4d acc[4];
#pragma unroll
    for (int y = 0; y < 4; ++y) acc[y] = v4d{0, 0, 0, 0};
    bool want[4];                                  // which of the row's four tiles exist and lie on or below the diagonal
#pragma unroll
    for (int y = 0; y < 4; ++y) want[y] = ti < tiles && tj0 + y <= ti;
    const int c_begin = slice * chunks_per_slice, c_end = min(min(c_begin + chunks_per_slice, c_begin + kUnitMaxChunks), n_chunks);
    const int nc = max(c_end - c_begin, 0);
    // ---- once per workgroup: the slice's chunk masks (a lane each), its chunks' slot ranges, the slots' (pose, rank) pairs; the panels clear ----
    const unsigned my_mask = lane < nc ? pb.chunk_mask[c_begin + lane] : 0u;
    for (int k = tid; k <= nc; k += 256) L.off[k] = pb.slice_off[c_begin + k];
    auto clear_panels = [&] {
        v2d* z = reinterpret_cast<v2d*>(&L.P);
        for (int k = tid; k < (int)(sizeof(UnitPanels) / sizeof(v2d)); k += 256) z[k] = v2d{0, 0};
    };
    clear_panels();
    __syncthreads();
    const int S0 = L.off[0], S1 = L.off[nc];
    for (int s = S0 + tid; s < min(S1, S0 + kUnitIdx); s += 256) L.idx[s - S0] = (unsigned short)(pb.fl_pose[s] | pb.fl_place[s] << 8);
    __syncthreads();
    const unsigned rmask = 15u << (4 * bi), cmask = 15u << (4 * bj);
    auto slot_index = [&](int s) -> unsigned { return s - S0 < kUnitIdx ? (unsigned)L.idx[s - S0] : (unsigned)(pb.fl_pose[s] | pb.fl_place[s] << 8); };
    auto next_chunk = [&](int c) {  // the next chunk from c on that reaches the unit (uniform)
        for (; c < c_end; ++c) {
            const unsigned m = __builtin_amdgcn_readlane(my_mask, c - c_begin);
            if ((m & rmask) && (m & cmask)) break;
        }
        return c;
    };
    auto load_slot = [&](int s, UnitSlot& u) {  // requests the slot's blocks; nothing waits for them here
        const unsigned ix = slot_index(s);
        u.p = (int)(ix >> 8);
        const int col0 = 6 * (int)(ix & 255u);
        u.ca = col0 - kUnitW * bi; u.cb = col0 - kUnitW * bj;  // first column of the pose inside the unit's row / column range
        u.inA = u.ca > -6 && u.ca < kUnitW; u.inB = u.cb > -6 && u.cb < kUnitW;
        if (u.inA) load_d2<18>(pb.Y + 18 * (size_t)s, u.Y);  // W D^-1 of the slot (k_ba_schur_coef)
        if (u.inB) load_d2<18>(pb.W + 18 * (size_t)s, u.W);
    };
    auto put_slot = [&](const UnitSlot& u) {
        if (u.inA) {
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                const int col = u.ca + r;
                if (col < 0 || col >= kUnitW) continue;
#pragma unroll
                for (int k = 0; k < 3; ++k) L.P.A[col >> 4][3 * u.p + k][col & 15] = u.Y[3 * r + k];
            }
        }
        if (u.inB) {
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                const int col = u.cb + r;
                if (col < 0 || col >= kUnitW) continue;
#pragma unroll
                for (int k = 0; k < 3; ++k) L.P.B[col >> 4][3 * u.p + k][col & 15] = u.W[3 * r + k];
            }
        }
    };
    UnitSlot pre;
    pre.inA = pre.inB = false;
    auto preload = [&](int c) {  // this lane's slot of chunk c (the first 256 slots of a chunk are prefetched: 16 landmarks x 16 poses)
        const int s = L.off[c - c_begin] + tid;
        pre.inA = pre.inB = false;
        if (s < L.off[c - c_begin + 1]) load_slot(s, pre);
    };
    int c = next_chunk(c_begin);
    if (c < c_end) preload(c);
    while (c < c_end) {
        const unsigned m = __builtin_amdgcn_readlane(my_mask, c - c_begin);
        const bool row_on = ((m >> ti) & 1u) != 0;
        const unsigned cb = (m >> tj0) & 15u;
        // ---- the panels (clear on entry): every slot of the chunk whose pose has columns in the unit's row / column range writes its part ----
        put_slot(pre);
        for (int s = L.off[c - c_begin] + 256 + tid; s < L.off[c - c_begin + 1]; s += 256) {  // (a chunk of more than 256 slots: the rest, not prefetched)
            load_slot(s, pre);
            put_slot(pre);
        }
        const int cn = next_chunk(c + 1);
        if (cn < c_end) preload(cn);  // in flight under the product below
        __syncthreads();
        // ---- the product: this wavefront's tile row over the chunk's twelve k-steps ----
        if (row_on && ti < tiles) {   // uniform over the wavefront
#pragma unroll 4
            for (int q = 0; q < kUnitRows / 4; ++q) {
                const int krow = 4 * q + (lane >> 4);
                const double a = L.P.A[wave][krow][lane & 15];
#pragma unroll
                for (int y = 0; y < 4; ++y)
                    if (want[y] && ((cb >> y) & 1u))
                        acc[y] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, L.P.B[y][krow][lane & 15], acc[y], 0, 0, 0);
            }
        }
        __syncthreads();  // every wavefront has read the panels
        clear_panels();
        __syncthreads();
        c = cn;
    }
    if (ti >= tiles) return;
    double* out = pb.S_part + (size_t)slice * pb.np_pad * pb.np_pad;
#pragma unroll
    for (int y = 0; y < 4; ++y) {
        if (!want[y]) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) out[(size_t)(16 * ti + (lane >> 4) + 4 * r) * pb.np_pad + 16 * (tj0 + y) + (lane & 15)] = acc[y][r];
    }
}
// The same product for reduced systems of at most 176 columns (29 free keyframes: LocalInertialBA's 25-keyframe bLarge window) with the chunk's
// panels at FULL width in LDS -- [tile][48][16] for (W D^-1)^T and W^T, 135 KB -- and ONE workgroup of eight wavefronts per (window, slice):
// every slot of a chunk is loaded once per window (a 64 x 64 unit re-reads it for each of up to six units), all lower tiles the chunk's band
// touches are multiplied from the one pair of panels (tile n of the lower triangle belongs to wavefront n % 8: up to nine accumulator tiles
// each, independent MFMA chains), two wavefronts per SIMD.  Same skipping, same chunk order per
// tile: the bits are those of the unit form.
constexpr int kFullTilesMax = 11, kFullThreads = 512, kFullTilesPerWave = 9;   // 11 x 12 / 2 = 66 lower tiles over 8 wavefronts (1024 lanes would cap a lane at 128 registers: 146 spilled)
struct FullLds {
    double A[kFullTilesMax][kUnitRows][16], B[kFullTilesMax][kUnitRows][16];
    unsigned short idx[kUnitIdx];
    int lm[kUnitIdx];            // the slots' landmarks
    int off[kUnitMaxChunks + 1];
};
__device__ __forceinline__ void d_ba_schur_full(const BaProblemDev& pb, const int slice, const int chunks_per_slice, const double lambda, FullLds& L) {
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_in_block();
    const int tiles = pb.np_pad / 16, n_lower = tiles * (tiles + 1) / 2, n_chunks = pb.n_schur_slices;
    int my_ti[kFullTilesPerWave], my_tj[kFullTilesPerWave];
    v4d acc[kFullTilesPerWave];
#pragma unroll
    for (int q = 0; q < kFullTilesPerWave; ++q) {
        const int n = wave + (kFullThreads / 64) * q;
        int ti = 0;
        while ((ti + 1) * (ti + 2) / 2 <= n) ++ti;
        my_ti[q] = n < n_lower ? ti : -1;
        my_tj[q] = n - ti * (ti + 1) / 2;
        acc[q] = v4d{0, 0, 0, 0};
    }
    const int c_begin = slice * chunks_per_slice, c_end = min(min(c_begin + chunks_per_slice, c_begin + kUnitMaxChunks), n_chunks);
    const int nc = max(c_end - c_begin, 0);
    const unsigned my_mask = lane < nc ? pb.chunk_mask[c_begin + lane] : 0u;
    for (int k = tid; k <= nc; k += kFullThreads) L.off[k] = pb.slice_off[c_begin + k];
    const int panel_v2 = tiles * kUnitRows * 16 / 2;  // v2d per panel actually used
    auto clear_panels = [&] {
        v2d* za = reinterpret_cast<v2d*>(&L.A[0][0][0]);
        v2d* zb = reinterpret_cast<v2d*>(&L.B[0][0][0]);
        for (int k = tid; k < panel_v2; k += kFullThreads) { za[k] = v2d{0, 0}; zb[k] = v2d{0, 0}; }
    };
    clear_panels();
    __syncthreads();
    const int S0 = L.off[0], S1 = L.off[nc];
    for (int s = S0 + tid; s < min(S1, S0 + kUnitIdx); s += kFullThreads) { L.idx[s - S0] = (unsigned short)(pb.fl_pose[s] | pb.fl_place[s] << 8); L.lm[s - S0] = pb.fl_lm[s]; }
    __syncthreads();
    auto slot_index = [&](int s) -> unsigned { return s - S0 < kUnitIdx ? (unsigned)L.idx[s - S0] : (unsigned)(pb.fl_pose[s] | pb.fl_place[s] << 8); };
    // one slot per lane and chunk, prefetched a chunk ahead (16 landmarks x 32 poses fit the workgroup's lanes)
    int pre_p = 0, pre_col = 0;
    bool pre_on = false;
    // The first operand W D^-1 is formed HERE from the slot's W block and its landmark's 3 x 3 block (point_dinv's arithmetic and
    // d_ba_schur_coef's expression: the same bits as the Y array the 64 x 64 units read) -- 24 doubles per slot from memory instead of 36: with a
    // batch's 256 workgroups loading at once the kernel is bound by those bytes (127 MB per 32-window launch), not by the 9 us of matrix work.
    double preW[18], preH[6];
    auto load_slot = [&](int s) {
        const unsigned ix = slot_index(s);
        pre_p = (int)(ix >> 8);
        pre_col = 6 * (int)(ix & 255u);
        const int l = s - S0 < kUnitIdx ? L.lm[s - S0] : pb.fl_lm[s];
        load_d2<18>(pb.W + 18 * (size_t)s, preW);
        load_d2<6>(pb.Hll + 6 * (size_t)l, preH);
    };
    auto put_slot = [&] {
        double preY[18];
        {
            const double d00 = preH[0] + lambda, d01 = preH[1], d02 = preH[2], d11 = preH[3] + lambda, d12 = preH[4], d22 = preH[5] + lambda;
            const double c00 = d11 * d22 - d12 * d12, c01 = d12 * d02 - d01 * d22, c02 = d01 * d12 - d11 * d02;
            const double det = d00 * c00 + d01 * c01 + d02 * c02, id = 1.0 / det;
            const double Di[9] = {c00 * id, (d02 * d12 - d01 * d22) * id, (d01 * d12 - d02 * d11) * id,
                                  c01 * id, (d00 * d22 - d02 * d02) * id, (d02 * d01 - d00 * d12) * id,
                                  c02 * id, (d01 * d02 - d00 * d12) * id, (d00 * d11 - d01 * d01) * id};
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int k = 0; k < 3; ++k) preY[3 * r + k] = preW[3 * r] * Di[k] + preW[3 * r + 1] * Di[3 + k] + preW[3 * r + 2] * Di[6 + k];
        }
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            const int col = pre_col + r;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                L.A[col >> 4][3 * pre_p + k][col & 15] = preY[3 * r + k];
                L.B[col >> 4][3 * pre_p + k][col & 15] = preW[3 * r + k];
            }
        }
    };
    auto preload = [&](int c) {
        const int s = L.off[c - c_begin] + tid;
        pre_on = s < L.off[c - c_begin + 1];
        if (pre_on) load_slot(s);
    };
    int c = c_begin;
    if (c < c_end) preload(c);
    while (c < c_end) {
        const unsigned m = __builtin_amdgcn_readlane(my_mask, c - c_begin);
        if (pre_on) put_slot();
        for (int s = L.off[c - c_begin] + kFullThreads + tid; s < L.off[c - c_begin + 1]; s += kFullThreads) { load_slot(s); put_slot(); }
        if (c + 1 < c_end) preload(c + 1); else pre_on = false;  // in flight under the product below
        __syncthreads();
#pragma unroll
        for (int q = 0; q < kFullTilesPerWave; ++q) {
            if (my_ti[q] < 0 || !((m >> my_ti[q]) & 1u) || !((m >> my_tj[q]) & 1u)) continue;  // uniform over the wavefront
#pragma unroll 4
            for (int ks = 0; ks < kUnitRows / 4; ++ks) {
                const int krow = 4 * ks + (lane >> 4);
                acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(L.A[my_ti[q]][krow][lane & 15], L.B[my_tj[q]][krow][lane & 15], acc[q], 0, 0, 0);
            }
        }
        __syncthreads();  // every wavefront has read the panels
        clear_panels();
        __syncthreads();
        ++c;
    }
    double* out = pb.S_part + (size_t)slice * pb.np_pad * pb.np_pad;
#pragma unroll
    for (int q = 0; q < kFullTilesPerWave; ++q) {
        if (my_ti[q] < 0) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) out[(size_t)(16 * my_ti[q] + (lane >> 4) + 4 * r) * pb.np_pad + 16 * my_tj[q] + (lane & 15)] = acc[q][r];
    }
}
__global__ __launch_bounds__(kFullThreads) void k_ba_schur_full(BaProblemDev pb, int chunks_per_slice, double lambda) {
    extern __shared__ double s_full_dyn[];
    d_ba_schur_full(pb, blockIdx.x, chunks_per_slice, lambda, *reinterpret_cast<FullLds*>(s_full_dyn));
}
__global__ __launch_bounds__(256) void k_ba_schur_units(BaProblemDev pb, int chunks_per_slice, double lambda) {
    __shared__ UnitsLds L;
    d_ba_schur_units(pb, blockIdx.x, blockIdx.y, chunks_per_slice, lambda, L);
}

// ---- the Schur product block by block on the f64 vector unit (default since round 3) ----
// S_part = sum over landmarks l and pairs (i >= j) of free poses that see l of (W_il D_l^-1) W_jl^T, plus the row sum_l W_il D_l^-1 b_l:
// exactly the 6x3 . 3x3 . 3x6 products g2o's BlockSolver forms (block_solver.hpp:381-432), and nothing else -- the MFMA form above
// multiplies [48 x np_pad] operands that are nine tenths structural zeros.  One workgroup per PART = kSchurGroup consecutive slices
// (<= 256 slots of <= 64 landmarks each).  Per slice: thread = slot loads its W block once (the next slice's loads are in flight while
// this slice is multiplied: the slice loop is unrolled, so every wait counts exactly the loads it needs), forms Y = W D^-1 and puts
// both into LDS; a 64-bit mask per free pose says which of the slice's landmarks it sees, a byte table which slot that is.  Then
// thread = TASK (ba_device.hpp: schur_task_*): one 6x6 block (i, j) over one range of landmark ranks -- the diagonal blocks, which
// every slot of pose i feeds, are cut into more ranges than the others, so that all tasks run loops of similar length -- or the
// coefficient row of pose i; a task walks the landmarks of mask[i] & mask[j] in ascending order and keeps its 36 sums in registers
// across the part's slices.  The order of every sum is a function of the window alone (slices and landmark ranks are cut by the
// host from the window), so a window gives the same bits alone, in a lock-step batch and from launch to launch.  Output layout = the
// MFMA form's: the lower triangle of [np_pad][np_pad] per part and the coefficient row np, so k_ba_schur_finish adds the parts as before.
// Slot stride in LDS 38 doubles: 16-byte aligned blocks that spread over all banks (at 36 doubles every read of a wavefront met
// 16-way conflicts).
// (Round 6: the 256-slot form of this product -- k_ba_schur_blocks, 80 KB of LDS and 256 registers per lane -- and the zero-padded MFMA form
// of rounds 1-2 -- k_ba_schur_sparse4 / 9 -- are gone: since round 4 the loop runs the LEAN form below, which is this product in 128-slot
// slices; the two were kept as switches with tests of their own.  Windows of more than kSchurLeanMaxFree free keyframes take the
// block-sparse MFMA kernels above.)

// ---- the same product, LEAN (pb.schur_blocks == 2; round 4, last session) ----
// k_ba_schur_blocks asks for 80 KB of LDS and 256 registers per lane: alone that is two workgroups per CU and costs nothing, but beside the
// other stages' kernels such a workgroup starts only on a CU that has half its LDS and half of every SIMD's register file free at the
// same moment -- the launch took 340-430 us in the loop against 62 us alone (the plane Hessian showed the same and lost a fifth of its
// in-loop time with its registers cut, balm_kernels.hip).  Here: slices of at most 128 slots (cut by the host, VisualProblem::setup in ba_internal.hpp), eight per
// part; threads 0..127 stage a slot each, all 256 are tasks as before; no block is held in registers across the task loop (the other
// workgroups of the CU cover a workgroup's loads: three fit); the closing sums go through LDS in range order -- the range tasks of a
// block add themselves one after the other, ((a0 + a1) + a2) + ..., the last one writes the block -- so the operand area (39 KB) is
// all the LDS the kernel needs.  Same arithmetic per product and the same order rules as above (a function of the window alone); the
// bits differ from the 256-slot form's because the slices do.
__host__ __device__ inline size_t schur_lean_lds_bytes(int nf) {
    return (size_t)kSchurLeanSlots * kSchurOps * sizeof(double) + 64 * 3 * sizeof(double) + 2 * (size_t)nf * 8 + 64 * (size_t)nf;
}
// task_base: 0, or 256 for the second workgroup of a part of a WIDE window (22 .. kSchurLeanMaxFree free keyframes: up to 512 tasks; both
// workgroups stage the part's slots, each takes its half of the tasks -- schur_ranges_wide keeps every block's range tasks in one half)
__device__ __forceinline__ void d_ba_schur_lean(const BaProblemDev& pb, const int part, const double lambda, double* __restrict__ lds, const int task_base = 0) {
    const int tid = threadIdx.x, task = tid + task_base, NF = pb.n_free, np = 6 * NF, ld = pb.np_pad;
    const int Rd = pb.schur_rd, Ro = pb.schur_ro;
    const int nD = Rd * NF, nO = Ro * (NF * (NF - 1) / 2), nT = nD + nO + NF;
    double* const ops = lds;
    double* const dbl = lds + kSchurLeanSlots * kSchurOps;                                   // [64][3]: D^-1 b_l by landmark rank
    unsigned long long* const masks = reinterpret_cast<unsigned long long*>(dbl + 64 * 3);   // [2][NF]: landmarks of the slice a pose sees
    unsigned char* const slot_of = reinterpret_cast<unsigned char*>(masks + 2 * NF);         // [64][NF]: the slot of (landmark rank, pose)
    // ---- this thread's task (the numbering of d_ba_schur_blocks) ----
    int t_i = 0, t_j = 0, t_kind = -1, t_q = 0, t_R = 1, t_close = -1;  // t_close: the block's place in the closing area (blocks of more than one range)
    unsigned long long t_sel = 0;
    if (task < nD) {
        t_i = t_j = task / Rd; t_q = task - t_i * Rd; t_R = Rd; t_sel = schur_range_mask(Rd, t_q); t_kind = 0;
        t_close = t_i;
    } else if (task < nD + nO) {
        const int n = (task - nD) / Ro;  // pair number i (i - 1) / 2 + j, i > j
        int i = 1;
        while (i * (i + 1) / 2 <= n) ++i;
        t_i = i; t_j = n - i * (i - 1) / 2; t_q = (task - nD) - n * Ro; t_R = Ro; t_sel = schur_range_mask(Ro, t_q); t_kind = 0;
        t_close = (Rd > 1 ? NF : 0) + n;
    } else if (task < nT) { t_i = t_j = task - nD - nO; t_sel = ~0ull; t_kind = 1; }
    double acc[36];
#pragma unroll
    for (int k = 0; k < 36; ++k) acc[k] = 0.0;
    const int sl0 = part * pb.schur_group, sl1 = min(sl0 + pb.schur_group, pb.n_schur_slices);
#pragma unroll 1
    for (int slice = sl0; slice < sl1; ++slice) {
        // two mask buffers in turn: this slice's is cleared while the tasks of the previous slice may still read theirs
        unsigned long long* const mask = masks + ((slice - sl0) & 1) * NF;
        if (tid < NF) mask[tid] = 0ull;
        const int s = pb.slice_off[slice] + tid;
        const bool have = tid < kSchurLeanSlots && s < pb.slice_off[slice + 1];
        int l = 0, key = 0;
        if (have) {  // the indices are asked for before the barrier, the blocks after it: nothing large is held while the workgroup waits
            l = pb.fl_lm[s];
            // a slice starts with a landmark's first slot (slices are whole landmarks)
            key = pb.fl_place[s] | pb.fl_pose[s] << 8 | (tid == 0 || pb.fl_lm[s - 1] != l ? 1 << 16 : 0);
        }
        __syncthreads();  // the masks are clear; the previous slice's tasks have read the operands and the slot table
        if (have) {
            const int place = key & 255, pose = (key >> 8) & 255;
            double h[6];
            load_d2<6>(pb.Hll + 6 * (size_t)l, h);
            const double* b = pb.bl + 3 * (size_t)l;
            const double b0 = b[0], b1 = b[1], b2 = b[2];
            const double* const Wg = pb.W + 18 * (size_t)s;
            double W01[6], W23[6], W45[6];  // the block two rows at a time: requested together, used one pair after the other
            load_d2<6>(Wg, W01); load_d2<6>(Wg + 6, W23); load_d2<6>(Wg + 12, W45);
            slot_of[place * NF + pose] = (unsigned char)tid;
            atomicOr(&mask[pose], 1ull << place);
            // (Hll + lambda I)^-1 and its product with b_l: the arithmetic of point_dinv (the back substitution forms the same inverse)
            const double d00 = h[0] + lambda, d01 = h[1], d02 = h[2], d11 = h[3] + lambda, d12 = h[4], d22 = h[5] + lambda;
            const double c00 = d11 * d22 - d12 * d12, c01 = d12 * d02 - d01 * d22, c02 = d01 * d12 - d11 * d02;
            const double det = d00 * c00 + d01 * c01 + d02 * c02, id = 1.0 / det;
            double Di[9];
            Di[0] = c00 * id; Di[1] = (d02 * d12 - d01 * d22) * id; Di[2] = (d01 * d12 - d02 * d11) * id;
            Di[3] = c01 * id; Di[4] = (d00 * d22 - d02 * d02) * id; Di[5] = (d02 * d01 - d00 * d12) * id;
            Di[6] = c02 * id; Di[7] = (d01 * d02 - d00 * d12) * id; Di[8] = (d00 * d11 - d01 * d01) * id;
            if (key >> 16) {
#pragma unroll
                for (int r = 0; r < 3; ++r) dbl[3 * place + r] = Di[3 * r] * b0 + Di[3 * r + 1] * b1 + Di[3 * r + 2] * b2;
            }
            double* const o = ops + tid * kSchurOps;
            // Y = W D^-1, two rows (three 16-byte pieces) at a time, W beside it
            auto rows = [&](const double (&w)[6], int r) {
                double y[6];
#pragma unroll
                for (int e = 0; e < 6; ++e) {
                    const int rr = e / 3, c = e % 3;
                    y[e] = w[3 * rr] * Di[c] + w[3 * rr + 1] * Di[3 + c] + w[3 * rr + 2] * Di[6 + c];
                }
                store_d2<6>(o + 3 * r, y);
                store_d2<6>(o + 18 + 3 * r, w);
            };
            rows(W01, 0); rows(W23, 2); rows(W45, 4);
        }
        __syncthreads();
        if (t_kind == 0) {
            unsigned long long m = mask[t_i] & mask[t_j] & t_sel;
            while (m) {
                const int p = __builtin_ctzll(m);
                m &= m - 1;
                // W_j whole, Y_i two rows at a time (the same three-term chains per entry as the 256-slot form: fewer values held at once)
                const double* const yp = ops + (int)slot_of[p * NF + t_i] * kSchurOps;
                double w[18];
                load_d2<18>(ops + (int)slot_of[p * NF + t_j] * kSchurOps + 18, w);
#pragma unroll
                for (int r = 0; r < 6; r += 2) {
                    double y[6];
                    load_d2<6>(yp + 3 * r, y);
#pragma unroll
                    for (int e = 0; e < 2; ++e)
#pragma unroll
                        for (int c = 0; c < 6; ++c)
                            acc[6 * (r + e) + c] = __builtin_fma(y[3 * e + 2], w[3 * c + 2], __builtin_fma(y[3 * e + 1], w[3 * c + 1], __builtin_fma(y[3 * e], w[3 * c], acc[6 * (r + e) + c])));
                }
            }
        } else if (t_kind == 1) {
            unsigned long long m = mask[t_i];
            while (m) {
                const int p = __builtin_ctzll(m);
                m &= m - 1;
                const double* Wa = ops + (int)slot_of[p * NF + t_i] * kSchurOps + 18;
                const double d0 = dbl[3 * p], d1 = dbl[3 * p + 1], d2 = dbl[3 * p + 2];
#pragma unroll
                for (int r = 0; r < 6; ++r) acc[r] = __builtin_fma(Wa[3 * r + 2], d2, __builtin_fma(Wa[3 * r + 1], d1, __builtin_fma(Wa[3 * r], d0, acc[r])));
            }
        }
    }
    // ---- the part's sums: the ranges of a block one after the other through its place in LDS, the last one keeps the sum ----
    const int Rmax = max(Rd, Ro);
    double* const cl = ops + 36 * max(t_close, 0);
    const bool ranged = t_kind == 0 && t_R > 1;
    for (int q = 0; q + 1 < Rmax; ++q) {
        __syncthreads();  // q == 0: the last slice's tasks have read the operands; later: range q - 1 is in place
        if (ranged && t_q == q && q + 1 < t_R) {
            if (q == 0) {
                store_d2<36>(cl, acc);
            } else {
#pragma unroll
                for (int hk = 0; hk < 36; hk += 12) {  // a third at a time: few values held beside the 36 sums
                    double v[12];
                    load_d2<12>(cl + hk, v);
#pragma unroll
                    for (int k = 0; k < 12; ++k) v[k] = v[k] + acc[hk + k];
                    store_d2<12>(cl + hk, v);
                }
            }
        }
    }
    __syncthreads();
    double* const out = pb.S_part + (size_t)part * ld * ld;
    if (t_kind == 0 && t_q == t_R - 1) {
        if (t_R > 1) {
#pragma unroll
            for (int hk = 0; hk < 36; hk += 12) {
                double v[12];
                load_d2<12>(cl + hk, v);
#pragma unroll
                for (int k = 0; k < 12; ++k) acc[hk + k] = v[k] + acc[hk + k];
            }
        }
        double* const o = out + (size_t)(6 * t_i) * ld + 6 * t_j;
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = 0; c < 6; ++c)
                if (t_i != t_j || c <= r) o[(size_t)r * ld + c] = acc[6 * r + c];  // the lower triangle, like the 256-slot form
    } else if (t_kind == 1) {
#pragma unroll
        for (int r = 0; r < 6; ++r) out[(size_t)np * ld + 6 * t_i + r] = acc[r];  // row np: sum_l W D^-1 b_l
    }
}
__global__ __launch_bounds__(256, 3) void k_ba_schur_lean(BaProblemDev pb, double lambda) {
    extern __shared__ double s_schur[];
    d_ba_schur_lean(pb, blockIdx.x, lambda, s_schur);
}
// windows of 22 .. kSchurLeanMaxFree free keyframes (ba_device.hpp: schur_ranges_wide): two workgroups per part, each half of the tasks
__global__ __launch_bounds__(256, 3) void k_ba_schur_lean_wide(BaProblemDev pb, double lambda) {
    extern __shared__ double s_schur[];
    d_ba_schur_lean(pb, blockIdx.x >> 1, lambda, s_schur, 256 * (blockIdx.x & 1));
}

__device__ __forceinline__ void d_ba_schur_finish(const BaProblemDev& pb, const int bx, double lambda, int n_slices, double* __restrict__ S_out,
                                                         double* __restrict__ bs_out, double* __restrict__ bp_host = nullptr) {
    const int np = 6 * pb.n_free, idx = bx * 256 + threadIdx.x;
    // Only the lower triangle leaves (the LDL^T on the host reads nothing else, ldlt_solve_small): S_out is pinned host memory, every
    // entry crosses PCIe.
    if (idx < np * np && idx / np >= idx % np) {
        const int r = idx / np, c = idx % np;
        double s = 0;
        if (r / 6 == c / 6) {  // Hpp is block diagonal in the visual problem
            const int a = min(r % 6, c % 6), b = max(r % 6, c % 6);
            const int packed = a * 6 - a * (a - 1) / 2 + (b - a);
            s = pb.Hpp[27 * (size_t)(r / 6) + packed];
            if (r == c) s += lambda;
        }
        double sub = 0;
        const double* sp = pb.S_part + (size_t)r * pb.np_pad + c;
        const size_t step = (size_t)pb.np_pad * pb.np_pad;
        int k = 0;
        for (; k + 8 <= n_slices; k += 8) {  // eight partials in flight, added in slice order
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = sp[(size_t)(k + u) * step];
#pragma unroll
            for (int u = 0; u < 8; ++u) sub += v[u];
        }
        for (; k < n_slices; ++k) sub += sp[(size_t)k * step];
        S_out[idx] = s - sub;
    }
    if (idx < np) {
        const double bp = pb.Hpp[27 * (size_t)(idx / 6) + 21 + idx % 6];
        double coef = 0;
        if (pb.sparse_schur) {  // row np of the product: sum_l W D^-1 b_l, slice after slice
            const double* sp = pb.S_part + (size_t)np * pb.np_pad + idx;
            const size_t step = (size_t)pb.np_pad * pb.np_pad;
            for (int k = 0; k < n_slices; ++k) coef += sp[(size_t)k * step];
        } else {
            coef = pb.coef[idx];
        }
        bs_out[idx] = bp - coef;
        bs_out[np + idx] = bp;
        if (bp_host) bp_host[idx] = bp;
    }
}
__global__ __launch_bounds__(256) void k_ba_schur_finish(BaProblemDev pb, double lambda, int n_slices, double* __restrict__ S_out,
                                                         double* __restrict__ bs_out) { d_ba_schur_finish(pb, blockIdx.x, lambda, n_slices, S_out, bs_out); }

// x_l = D^-1 (b_l - W^T x_p), one thread per landmark over its W blocks (landmark-major: a thread reads one contiguous run, the
// workgroup one contiguous range); x_p comes from the host's pinned memory once per workgroup, not once per use.
constexpr int kBacksubPerBlock = 256;
__device__ __forceinline__ void backsub_body(const BaProblemDev& pb, int block, const double* __restrict__ xp, double lambda, double* s_sum, double* s_x) {
    const int l = block * kBacksubPerBlock + threadIdx.x;
    const int np = 6 * pb.n_free;
    const bool staged = np <= kBacksubMaxNp;
    if (staged) {
        for (int j = threadIdx.x; j < np; j += 256) s_x[j] = xp[j];
        __syncthreads();
    }
    double sc = 0;
    if (l < pb.n_points) {
        double d0 = 0, d1 = 0, d2 = 0;
        for (int k = pb.fl_off[2 * l]; k < pb.fl_off[2 * l + 1]; ++k) {
            double W[18];
            load_d2<18>(pb.W + 18 * (size_t)k, W);
            const double* x = (staged ? s_x : xp) + 6 * pb.fl_pose[k];
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                const double xr = x[r];
                d0 += W[3 * r] * xr;
                d1 += W[3 * r + 1] * xr;
                d2 += W[3 * r + 2] * xr;
            }
        }
        const double cl[3] = {pb.bl[3 * (size_t)l] - d0, pb.bl[3 * (size_t)l + 1] - d1, pb.bl[3 * (size_t)l + 2] - d2};
        double Di[9], db_[3];
        point_dinv(pb, l, lambda, Di, db_);  // the arithmetic of the Schur product's D^-1: the same bits
        for (int r = 0; r < 3; ++r) {
            const double x = Di[3 * r] * cl[0] + Di[3 * r + 1] * cl[1] + Di[3 * r + 2] * cl[2];
            pb.points_trial[3 * (size_t)l + r] = pb.points[3 * (size_t)l + r] + x;
            sc += x * (lambda * x + pb.bl[3 * (size_t)l + r]);
        }
    }
    block_sum_256(sc, s_sum, pb.scale_part + block);  // landmark part of computeScale, per workgroup
}

// One launch: workgroups [0, nbp) back-substitute the landmarks, the rest move the poses (exp(x_p) * T)
__device__ __forceinline__ void d_ba_trial_update(const BaProblemDev& pb, const int bx, int nbp, const double* __restrict__ xp, double lambda,
                                                  ImuPose* __restrict__ iposes_host = nullptr) {
    __shared__ double s_sum[256], s_x[kBacksubMaxNp];
    if (bx < nbp) { backsub_body(pb, bx, xp, lambda, s_sum, s_x); return; }
    const int k = (bx - nbp) * 256 + threadIdx.x;
    if (k >= pb.n_poses) return;
    const int i = pb.pose_var[k];
    if (pb.inertial) {
        ImuPose T = pb.iposes[k];
        if (i >= 0) {
            double u[6];
            for (int r = 0; r < 6; ++r) u[r] = xp[6 * i + r];
            imu_pose_update(T, pb.calib, u);
        }
        pb.iposes_trial[k] = T;
        if (iposes_host) iposes_host[k] = T;  // the host's inertial cost of the trial state reads it there (a posted write; a copy launch per trial before)
        return;
    }
    if (i < 0) { pb.poses_trial[k] = pb.poses[k]; return; }
    double u[6];
    for (int r = 0; r < 6; ++r) u[r] = xp[6 * i + r];
    pb.poses_trial[k] = se3_exp_mul(u, pb.poses[k]);
}
__global__ __launch_bounds__(256) void k_ba_trial_update(BaProblemDev pb, int nbp, const double* __restrict__ xp, double lambda) { d_ba_trial_update(pb, blockIdx.x, nbp, xp, lambda); }

// [0] landmark part of the gain-ratio scale, [1] robust cost of the trial estimate
template <bool DEVICE_SCOPE = false>  // chi_part comes from other workgroups of this launch
__device__ __forceinline__ void d_ba_trial_reduce(const BaProblemDev& pb, const int bx, double* __restrict__ scale_out, double* __restrict__ chi_out) {
    __shared__ double s[256];
    if (bx == 0) block_reduce_256<false>(pb.scale_part, (pb.n_points + kBacksubPerBlock - 1) / kBacksubPerBlock, s, scale_out);
    else block_reduce_256<false, DEVICE_SCOPE>(pb.chi_part, (pb.n_edges + 255) / 256, s, chi_out);
}
__global__ __launch_bounds__(256) void k_ba_trial_reduce(BaProblemDev pb, double* __restrict__ scale_out, double* __restrict__ chi_out) { d_ba_trial_reduce(pb, blockIdx.x, scale_out, chi_out); }

template <bool DEVICE_SCOPE = false>
__device__ __forceinline__ void d_ba_errors(const BaProblemDev& pb, const int bx) {
    __shared__ double s_sum[256];
    const int e = bx * 256 + threadIdx.x;
    double rho0 = 0;
    if (e < pb.n_edges) {
        const BaEdge ed = pb.edges[e];
        double p[3], err[3], c2;
        int dim;
        edge_state(pb, true, ed, p, err, dim, c2);
        const bool stereo = ed.ur >= 0;
        double rho1;
        huber(c2, stereo ? pb.delta_stereo : pb.delta_mono, stereo ? pb.dsqr_stereo : pb.dsqr_mono, rho0, rho1);
        pb.chi2[e] = c2;
        pb.rho0[e] = rho0;
    }
    block_sum_256<DEVICE_SCOPE>(rho0, s_sum, pb.chi_part + bx);
}
__global__ __launch_bounds__(256) void k_ba_errors(BaProblemDev pb) { d_ba_errors(pb, blockIdx.x); }

// ---- the trial estimate, its cost and the closing sums in ONE launch (round 5; windows with pb.trial_fused) -------------------------------
// A workgroup per landmark GROUP of the linearisation (whole landmarks, <= 256 edges): it stages the step, moves every vertex of the
// window into LDS (exp(x_p) T: a few dozen poses, formed by every workgroup rather than fetched from a launch before), back-substitutes
// its landmarks (backsub_body's arithmetic), keeps their trial points in LDS and evaluates its edges against them -- three launches
// (k_ba_trial_update_b, k_ba_errors_reduce_b, k_balm_residual_total_b) and a round trip of the trial points through memory before.  The
// partial sums of the gain-ratio scale and of the robust cost are per GROUP (per 256 landmarks / 256 edges in the separate kernels): the
// one-window entry points run the same kernel, so a window gives the same bits alone and in a batch.  Workgroup 0 of a window writes the
// trial poses for the launches that follow; the window's last workgroup adds the partials (ba_last_of, ticket word 1).
struct TrialLds {
    double x[kBacksubMaxNp];
    double pts[3 * 256];
    double sum[256];
    alignas(16) unsigned char poses[kTrialPoseBytes];
};
template <bool INERTIAL>
__device__ __forceinline__ void trial_poses_lds(const BaProblemDev& pb, const double* s_x, unsigned char* s_poses, bool write_global, ImuPose* __restrict__ iposes_host) {
    for (int k = threadIdx.x; k < pb.n_poses; k += 256) {
        const int i = pb.pose_var[k];
        if (INERTIAL) {
            ImuPose T = pb.iposes[k];
            if (i >= 0) {
                double u[6];
                for (int r = 0; r < 6; ++r) u[r] = s_x[6 * i + r];
                imu_pose_update(T, pb.calib, u);
            }
            reinterpret_cast<ImuPose*>(s_poses)[k] = T;
            if (write_global) { pb.iposes_trial[k] = T; if (iposes_host) iposes_host[k] = T; }
        } else {
            Se3 T = pb.poses[k];
            if (i >= 0) {
                double u[6];
                for (int r = 0; r < 6; ++r) u[r] = s_x[6 * i + r];
                T = se3_exp_mul(u, T);
            }
            reinterpret_cast<Se3*>(s_poses)[k] = T;
            if (write_global) pb.poses_trial[k] = T;
        }
    }
}
template <bool INERTIAL>
__device__ __forceinline__ void d_ba_trial_group(const BaProblemDev& pb, const int g, const double* __restrict__ xp, double lambda, TrialLds& L,
                                                 ImuPose* __restrict__ iposes_host) {
    const int np = 6 * pb.n_free;
    for (int j = threadIdx.x; j < np; j += 256) L.x[j] = xp[j];
    __syncthreads();
    trial_poses_lds<INERTIAL>(pb, L.x, L.poses, g == 0, iposes_host);
    const int k0 = pb.grp_k0[g], k1 = pb.grp_k0[g + 1], l0 = pb.grp_l0[g], l1 = pb.grp_l0[g + 1];
    double sc = 0;
    const int l = l0 + (int)threadIdx.x;
    if (l < l1) {  // x_l = D^-1 (b_l - W^T x_p): backsub_body's operations in its order
        double d0 = 0, d1 = 0, d2 = 0;
        for (int k = pb.fl_off[2 * l]; k < pb.fl_off[2 * l + 1]; ++k) {
            double W[18];
            load_d2<18>(pb.W + 18 * (size_t)k, W);
            const double* x = L.x + 6 * pb.fl_pose[k];
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                const double xr = x[r];
                d0 += W[3 * r] * xr;
                d1 += W[3 * r + 1] * xr;
                d2 += W[3 * r + 2] * xr;
            }
        }
        const double cl[3] = {pb.bl[3 * (size_t)l] - d0, pb.bl[3 * (size_t)l + 1] - d1, pb.bl[3 * (size_t)l + 2] - d2};
        double Di[9], db_[3];
        point_dinv(pb, l, lambda, Di, db_);
        for (int r = 0; r < 3; ++r) {
            const double x = Di[3 * r] * cl[0] + Di[3 * r + 1] * cl[1] + Di[3 * r + 2] * cl[2];
            const double X = pb.points[3 * (size_t)l + r] + x;
            pb.points_trial[3 * (size_t)l + r] = X;
            L.pts[3 * threadIdx.x + r] = X;
            sc += x * (lambda * x + pb.bl[3 * (size_t)l + r]);
        }
    }
    __syncthreads();
    double rho0 = 0;
    const int k = k0 + (int)threadIdx.x;
    if (k < k1) {
        const int e = pb.pt_edges[k];
        const BaEdge ed = pb.edges[e];
        const double* X = L.pts + 3 * (ed.point - l0);
        double p[3], err[3], c2 = 0;
        int dim;
        if (INERTIAL) {
            dim = imu_edge_error(reinterpret_cast<const ImuPose*>(L.poses)[ed.pose], X, ed, pb.cam, p, err);
        } else {
            se3_map(reinterpret_cast<const Se3*>(L.poses)[ed.pose], X, p);
            dim = edge_error(p, ed, pb.cam, err);
        }
        for (int d = 0; d < dim; ++d) c2 += err[d] * ed.info * err[d];
        const bool stereo = ed.ur >= 0;
        double rho1;
        huber(c2, stereo ? pb.delta_stereo : pb.delta_mono, stereo ? pb.dsqr_stereo : pb.dsqr_mono, rho0, rho1);
        pb.chi2[e] = c2;
        pb.rho0[e] = rho0;
    }
    block_sum_256<true>(sc, L.sum, pb.scale_part + g);
    __syncthreads();
    block_sum_256<true>(rho0, L.sum, pb.chi_part + g);
}
// [0] landmark part of the gain-ratio scale, [1] robust cost of the trial estimate: the groups' partials in group order (one workgroup)
__device__ __forceinline__ void d_ba_trial_close(const BaProblemDev& pb, double* s, double* __restrict__ scale_out, double* __restrict__ chi_out) {
    block_reduce_256<false, true>(pb.scale_part, pb.n_groups, s, scale_out);
    block_reduce_256<false, true>(pb.chi_part, pb.n_groups, s, chi_out);
}
__global__ __launch_bounds__(256) void k_ba_trial_fused(BaProblemDev pb, const double* __restrict__ xp, double lambda, double* __restrict__ scale_out,
                                                        double* __restrict__ chi_out) {
    __shared__ TrialLds L;
    if (pb.inertial) d_ba_trial_group<true>(pb, blockIdx.x, xp, lambda, L, nullptr);
    else d_ba_trial_group<false>(pb, blockIdx.x, xp, lambda, L, nullptr);
    if (!ba_last_of(pb.ticket + 1, pb.n_groups)) return;
    d_ba_trial_close(pb, L.sum, scale_out, chi_out);
}

__device__ __forceinline__ void d_ba_depth(const BaProblemDev& pb, const int bx, uint8_t* __restrict__ depth_pos) {
    const int e = bx * 256 + threadIdx.x;
    if (e >= pb.n_edges) return;
    const BaEdge ed = pb.edges[e];
    const double* X = pb.points + 3 * (size_t)ed.point;
    if (pb.inertial) {  // ImuCamPose::isDepthPositive
        const ImuPose& T = pb.iposes[ed.pose];
        depth_pos[e] = (T.Rcw[6] * X[0] + T.Rcw[7] * X[1] + T.Rcw[8] * X[2] + T.tcw[2]) > 0.0;
        return;
    }
    double p[3];
    se3_map(pb.poses[ed.pose], X, p);
    depth_pos[e] = p[2] > 0.0;
}
__global__ __launch_bounds__(256) void k_ba_depth(BaProblemDev pb, uint8_t* __restrict__ depth_pos) { d_ba_depth(pb, blockIdx.x, depth_pos); }


// ---- lock-step batch: the same bodies, the window taken from a slot table (blockIdx.y / z = position in the active list) ----
static inline __device__ int blocks256(int n) { return (n + 255) / 256; }
// s_setprio 3: the lock-step kernels are links of a dependent chain with a host step after every phase, running beside the front end's
// long kernels; their wavefronts go first in the SIMDs' issue arbitration (41.7 against 42.1 ms per step of the whole loop).
// The window of a workgroup: the call's resident table entry of the phase's window number `pos`, with this phase's state applied to a
// private copy of its problem record (no reloads after stores): the parity bit swaps the accepted and the trial buffers.
struct BaSlotView {
    const BaBatchSlot& sl;
    BaProblemDev pb;
    double lambda;
    unsigned flags;
    const double* xp;
    bool active;  // device-side LM: the window's status is the one this launch is for (BaPhase::expect)
    __device__ __forceinline__ double* hpp_out() const { return (flags & kBaWantHpp) ? sl.hpp_out : nullptr; }
};
__device__ __forceinline__ BaSlotView ba_slot_view(const BaPhase& ph, int pos) {
    __builtin_amdgcn_s_setprio(3);
    // The phase's per-window entries are sub-dword loads from the argument block at a dynamic index: vector loads, so the window number --
    // and with it the address of everything read from the slot -- would sit in vector registers and the whole record be fetched per lane
    // (about 40 VGPRs of uniform pointers and sizes in every _b kernel: k_ba_linearize_b 140 -> 100, k_ba_trial_update_b 152 -> 104).
    // readfirstlane makes them scalar again and load_uniform reads the record through scalar loads into SGPRs.
    const int win = ba_phase_window(ph, pos);
    const unsigned flags = ba_phase_flags(ph, pos);
    union { double d; int i[2]; } lam;
    lam.d = ph.lambda[pos];
    lam.i[0] = __builtin_amdgcn_readfirstlane(lam.i[0]);
    lam.i[1] = __builtin_amdgcn_readfirstlane(lam.i[1]);
    const BaBatchSlot& sl = ph.table[win];
    const BaLmView lmv = ba_lm_view(ph, &sl, flags, lam.d);  // device-side LM: lambda, parity and the request bits from the window's state
    BaSlotView v{sl, load_uniform(&sl.pb), lmv.lambda, lmv.flags, global_ptr(load_uniform(&sl.xp)), lmv.active};
    ba_problem_pointers_are_global(v.pb);
    if (v.flags & kBaAcceptedInTrial) {
        Se3* const p = v.pb.poses; v.pb.poses = v.pb.poses_trial; v.pb.poses_trial = p;
        ImuPose* const q = v.pb.iposes; v.pb.iposes = v.pb.iposes_trial; v.pb.iposes_trial = q;
        double* const x = v.pb.points; v.pb.points = v.pb.points_trial; v.pb.points_trial = x;
    }
    if (ph.xp_area && 6 * v.pb.n_free <= kBaXpStride) v.xp = global_ptr(ph.xp_area) + (size_t)(ph.first + pos) * kBaXpStride;
    return v;
}
#define TC2LI_SLOT(axis) const BaSlotView view_ = ba_slot_view(ph, blockIdx.axis); if (!view_.active) return; const BaBatchSlot& sl = view_.sl; (void)sl; const BaProblemDev& pb = view_.pb

// workgroups [0, max_groups) of a window: the landmark role; [max_groups, ...): the pose role -- one launch (two before: the second
// waited for the first to drain although neither reads what the other writes)
// A kernel per vertex type (the windows of a batch call share it): the SE3 form does not carry the registers of the ImuCamPose Jacobians
// (196 VGPRs with both in one body: two wavefronts per SIMD for a kernel that waits on scattered loads; 142 / 148 now, three.  Holding
// the body to four with amdgpu_waves_per_eu -- 126 registers, 68 bytes of scratch per lane -- measured the same in the loop: 361 against 375 us.
// With the slot read through scalar loads (ba_slot_view) the body needs 100 registers: four wavefronts, 290-300 us in the loop; held to five
// (96 registers + 12 B of scratch) the kernel runs 254-273 us and the step does not change: 29.5 against 29.7 ms over five A/B pairs).
// fuse != 0: the closing sums (k_ba_reduce_all_b, k_ba_maxdiag_b) by the window's last workgroup instead of two more launches (round 5)
template <bool INERTIAL>
__device__ __forceinline__ void linearize_b_body(const BaPhase& ph, int max_groups, int fuse, LinearizeLds& lds) {
    TC2LI_SLOT(y);
    const int bx = blockIdx.x;
    const int nbe = blocks256(pb.n_free_edges);
    if (!fuse) {
        if (bx < max_groups) { if (bx < pb.n_groups) d_ba_linearize<INERTIAL>(pb, bx, lds); }
        else if (bx - max_groups < nbe) d_ba_linearize_pose<INERTIAL>(pb, bx - max_groups, lds);
        return;
    }
    if (bx < max_groups) { if (bx >= pb.n_groups) return; d_ba_linearize<INERTIAL, true>(pb, bx, lds); }
    else { if (bx - max_groups >= nbe) return; d_ba_linearize_pose<INERTIAL, true>(pb, bx - max_groups, lds); }
    if (!ba_last_of(pb.ticket + 0, pb.n_groups + nbe)) return;
    d_ba_linearize_close(pb, lds.big, sl.chi_out, view_.hpp_out(), (view_.flags & kBaWantMaxdiag) ? sl.maxdiag_out : nullptr);
}
__global__ __launch_bounds__(256) void k_ba_linearize_b(const BaPhase ph, int max_groups, int fuse) {
    __shared__ LinearizeLds lds;
    linearize_b_body<false>(ph, max_groups, fuse, lds);
}
__global__ __launch_bounds__(256) void k_ba_linearize_imu_b(const BaPhase ph, int max_groups, int fuse) {
    __shared__ LinearizeLds lds;
    linearize_b_body<true>(ph, max_groups, fuse, lds);
}
__global__ __launch_bounds__(256) void k_ba_reduce_all_b(const BaPhase ph) {
    TC2LI_SLOT(y);
    if ((int)blockIdx.x >= pb.n_free + 1) return;
    d_ba_reduce_all(pb, blockIdx.x, sl.chi_out, view_.hpp_out());
}
__global__ __launch_bounds__(64) void k_ba_dups_b(const BaPhase ph) {
    TC2LI_SLOT(y);
    if (pb.inertial) d_ba_dups<true>(pb, view_.hpp_out()); else d_ba_dups<false>(pb, view_.hpp_out());
}
__global__ __launch_bounds__(256) void k_ba_maxdiag_b(const BaPhase ph) {
    TC2LI_SLOT(y);
    if (!(view_.flags & kBaWantMaxdiag)) return;
    d_ba_maxdiag(pb, blockIdx.x, sl.maxdiag_out);
}
__global__ __launch_bounds__(256) void k_ba_schur_coef_b(const BaPhase ph, int full_form) {
    TC2LI_SLOT(y);
    if (pb.sparse_schur || (int)blockIdx.x >= blocks256(pb.n_free_edges)) return;
    d_ba_schur_coef(pb, blockIdx.x, view_.lambda, !(full_form && pb.np_pad <= 16 * kFullTilesMax));
}
__global__ __launch_bounds__(256) void k_ba_reduce_coef_b(const BaPhase ph) {
    TC2LI_SLOT(y);
    if (pb.sparse_schur || (int)blockIdx.x >= pb.n_free) return;
    d_ba_reduce_coef(pb, blockIdx.x);
}
// the dense windows of at most 176 columns: (slice, window) = blockIdx.(x, y)
__global__ __launch_bounds__(kFullThreads) void k_ba_schur_full_b(const BaPhase ph) {
    extern __shared__ double s_full_dyn[];
    TC2LI_SLOT(y);
    if (pb.sparse_schur || !pb.n_free || pb.np_pad > 16 * kFullTilesMax || (int)blockIdx.x >= sl.n_slices) return;
    d_ba_schur_full(pb, blockIdx.x, sl.k_per_slice, view_.lambda, *reinterpret_cast<FullLds*>(s_full_dyn));
}
// the dense windows' Schur product: (unit, slice, window) = blockIdx.(x, y, z)
__global__ __launch_bounds__(256) void k_ba_schur_units_b(const BaPhase ph) {
    __shared__ UnitsLds L;
    TC2LI_SLOT(z);
    const int ub = (pb.np_pad / 16 + 3) / 4;
    if (pb.sparse_schur || !pb.n_free || (int)blockIdx.x >= ub * (ub + 1) / 2 || (int)blockIdx.y >= sl.n_slices) return;
    if (pb.np_pad <= 16 * kFullTilesMax && !ph.pad_) return;  // such a window runs in k_ba_schur_full_b (ph.pad_: TC2LI_BA_DENSE_FULL=0, measurements)
    d_ba_schur_units(pb, blockIdx.x, blockIdx.y, sl.k_per_slice, view_.lambda, L);
}
__global__ __launch_bounds__(256, 3) void k_ba_schur_lean_b(const BaPhase ph) {
    extern __shared__ double s_schur[];
    TC2LI_SLOT(y);
    if (!pb.sparse_schur || !pb.n_free) return;
    // a wide window (more than kSchurBlocksMaxFree free keyframes): two workgroups per part, the same launch as everybody else's
    const int halves = pb.n_free > kSchurBlocksMaxFree ? 2 : 1, part = (int)blockIdx.x / halves;
    if (part >= sl.n_slices) return;
    d_ba_schur_lean(pb, part, view_.lambda, s_schur, 256 * ((int)blockIdx.x - part * halves));
}
__global__ __launch_bounds__(256) void k_ba_schur_finish_b(const BaPhase ph) {
    TC2LI_SLOT(y);
    const int np = 6 * pb.n_free;
    if (!pb.n_free || (int)blockIdx.x >= blocks256(np * np)) return;
    d_ba_schur_finish(pb, blockIdx.x, view_.lambda, sl.n_slices, sl.S_out, sl.bs_out, sl.bp_host);
}
// The reduced camera system of a window on the device: (S + Hl) x = b_s + bl by the dense LDL^T of ldlt_solve_small (ba_math.hpp), one
// workgroup per window, lane i = row i.  Every element is formed by the same operations in the same order as on the host -- a row's
// sums run over k ascending, one column after the other; the forward substitution subtracts column by column (k ascending per row); the
// backward substitution is the host's row loop, one lane at a time -- so a window gives the same step here and in the one-window path.
// The lower triangle lives packed in LDS (row i at i (i + 1) / 2).  A pivot that is zero or not finite: ok = 0 and a zero step (the
// host treats the trial as failed).
// The reduced system of a window solved on the device (TC2LI_BA_DEVICE_SOLVE=1): ldlt_solve_small's factorisation and substitutions, the same
// operations on the same operands in the same order per entry -- so the step is the host's bit for bit -- carried out by a workgroup:
//   * factorisation RIGHT-LOOKING (round 5): after column k has its pivot and L_ik = a_ik / D_k, every remaining entry (i, j), j > k, takes its
//     term (L_ik L_jk) D_k -- k = 0, 1, ... in the order the host's row-by-row sums take them -- all entries of a step side by side over the
//     256 threads (rounds 2-4: one thread per row walked its row's sums, and the closing substitution was ONE thread's chain of n (n - 1) / 2
//     LDS round trips: 80-100 us per window, slower than the host's 60);
//   * substitutions as column sweeps by wavefront 0 (x_k is final, every other row takes its term): n steps each.  The backward sweep adds a
//     row's terms with k DESCENDING; ldlt_solve_small (ba_math.hpp) does the same since round 5.
constexpr int kSolveThreads = 256;
// Round 6: the matrix lives in REGISTERS.  Rounds 2-5 kept it in LDS and found every trailing entry's place with a double-precision square root
// and two divisions per update: 128 us per launch whatever the batch -- a third of a lock-step round at 64 sequences per GPU, where the round IS
// the step.  Now thread (ty, tx) = (tid / 16, tid % 16) owns entry (16 ti + ty, 16 tj + tx) of every 16 x 16 tile (ti >= tj) of the lower
// triangle -- NT (NT + 1) / 2 doubles, NT = 5 for up to 13 free keyframes, 8 for the sparse path's limit of 21 -- and the right-hand side rides
// along (thread (ty, 0): entry 16 ti + ty).  Column k of the right-looking elimination: its owners (the 16 threads with tx = k % 16) put their
// entries into LDS, one thread per row divides by the pivot, and every thread takes its entries' terms (l_i l_j) d from
// the two short LDS vectors -- two barriers per column, no index arithmetic (the tile loops are unrolled, a tile left of the column is a
// scalar branch).  The operations per entry are the old kernel's and ldlt_solve_small's, in their order: the step is the same bit for bit.
// Then L goes to LDS once and wavefront 0 runs the backward sweep with the solution in its registers (x_k by v_readlane).
template <int NT>
__device__ __forceinline__ void d_ba_solve_tiles(const BaBatchSlot& sl, const int n, double* s_L) {
    constexpr int kTiles = NT * (NT + 1) / 2;
    constexpr int kRows = 16 * 9;   // the widest variant: 24 free keyframes
    __shared__ double s_colA[kRows], s_colL[kRows], s_dg[kRows], s_xs[kRows], s_y;
    __shared__ int s_bad;
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;   // (neighbouring lanes: neighbouring columns of a row -- the loads below run along rows)
    const double* __restrict__ S = global_ptr(load_uniform(&sl.S_out));
    const double* __restrict__ Hl = global_ptr(load_uniform(&sl.Hl));
    const double* __restrict__ bs = global_ptr(load_uniform(&sl.bs_out));
    const double* __restrict__ bl = global_ptr(load_uniform(&sl.bl_lidar));
    double a[kTiles], rhs[NT];
    if (tid == 0) s_bad = 0;
    if (tid < kRows) s_colL[tid] = 0.0;   // (never written at or beyond row n: the unpredicated products there stay zero)
    // every load at a clamped (always valid) address and without a branch around it: they are all in flight together; what lies outside
    // the window's lower triangle is zeroed afterwards
    {
        double hs[kTiles], hl[kTiles], rb[NT], rl[NT];
#pragma unroll
        for (int ti = 0; ti < NT; ++ti) {
            const int i = min(16 * ti + ty, n - 1);
#pragma unroll
            for (int tj = 0; tj <= ti; ++tj) {
                const int e = i * n + min(16 * tj + tx, n - 1), q = ti * (ti + 1) / 2 + tj;
                hs[q] = S[e];
                hl[q] = Hl ? Hl[e] : 0.0;
            }
            rb[ti] = bs[i];
            rl[ti] = Hl ? bl[i] : 0.0;
        }
#pragma unroll
        for (int ti = 0; ti < NT; ++ti) {
            const int i = 16 * ti + ty;
#pragma unroll
            for (int tj = 0; tj <= ti; ++tj) {
                const int j = 16 * tj + tx, q = ti * (ti + 1) / 2 + tj;
                const double v = Hl ? hs[q] + hl[q] : hs[q];
                a[q] = i < n && j <= i ? v : 0.0;
            }
            const double v = Hl ? rb[ti] + rl[ti] : rb[ti];
            rhs[ti] = tx == 0 && i < n ? v : 0.0;
        }
    }
    // The columns tile column by tile column: inside a segment the tile numbers are compile-time constants -- which tiles lie right of the
    // column, which register holds the column's entry of a tile row -- so a step is the products and little else (a wavefront alone on its
    // SIMD issues an instruction every four to five cycles: with the tile tests made per step the kernel spent 600 instructions on a
    // column's 45 multiply-subtracts, 92 us per solve).  What lies above the diagonal inside a diagonal tile, and beyond row n, is carried
    // along unpredicated: those registers are never read into a result (s_colL stays zero beyond n).
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
        const int k_end = min(16 * (kt + 1), n);
        for (int k = 16 * kt; k < k_end; ++k) {
            const int kr = k - 16 * kt;
            // the column's entries from the pivot down, and the right-hand side's entry k (final: it is y_k)
            if (tx == kr) {
                if (ty >= kr) s_colA[16 * kt + ty] = a[kt * (kt + 1) / 2 + kt];
#pragma unroll
                for (int ti = kt + 1; ti < NT; ++ti) s_colA[16 * ti + ty] = a[ti * (ti + 1) / 2 + kt];
            }
            if (tx == 0 && ty == kr) s_y = rhs[kt];
            __syncthreads();
            const double d = s_colA[k], y = s_y;
            if (tid == 0) { if (!(d == d) || d == 0.0 || d - d != 0.0) s_bad = 1; s_dg[k] = d; }
            if (tid > k && tid < n) s_colL[tid] = s_colA[tid] / d;
            __syncthreads();
            double li[NT], lj[NT];
#pragma unroll
            for (int t = kt; t < NT; ++t) { li[t] = s_colL[16 * t + ty]; lj[t] = s_colL[16 * t + tx]; }
            const bool pi = ty > kr, pj = tx > kr, pk = tx == kr;   // row below the pivot / column right of it / the column itself (tile column kt only)
            {   // the diagonal tile of the segment
                const int q = kt * (kt + 1) / 2 + kt;
                const double upd = a[q] - li[kt] * lj[kt] * d;
                a[q] = pi && pj ? upd : (pi && pk ? li[kt] : a[q]);
                const double r = rhs[kt] - li[kt] * y;
                rhs[kt] = pi ? r : rhs[kt];
            }
#pragma unroll
            for (int ti = kt + 1; ti < NT; ++ti) {
                {   // the tile in the column's own tile column: its entries right of the column take their term, the column's become L
                    const int q = ti * (ti + 1) / 2 + kt;
                    const double upd = a[q] - li[ti] * lj[kt] * d;
                    a[q] = pj ? upd : (pk ? li[ti] : a[q]);
                }
#pragma unroll
                for (int tj = kt + 1; tj <= ti; ++tj) { const int q = ti * (ti + 1) / 2 + tj; a[q] -= li[ti] * lj[tj] * d; }
                rhs[ti] -= li[ti] * y;
            }
        }
    }
    // L (strictly lower) and y to LDS; z = y / D; backward sweep x_i = z_i - sum_{k > i} L_ki x_k, k descending, by wavefront 0
#pragma unroll
    for (int ti = 0; ti < NT; ++ti) {
        const int i = 16 * ti + ty;
#pragma unroll
        for (int tj = 0; tj <= ti; ++tj) { const int j = 16 * tj + tx; if (i < n && j < i) s_L[i * (i + 1) / 2 + j] = a[ti * (ti + 1) / 2 + tj]; }  // (packed: 24 free keyframes are 83 KB)
        if (tx == 0 && i < n) s_xs[i] = rhs[ti];
    }
    __syncthreads();
    if (tid < 64) {
        const int lane = tid;
        double x0 = lane < n ? s_xs[lane] / s_dg[lane] : 0.0, x1 = lane + 64 < n ? s_xs[lane + 64] / s_dg[lane + 64] : 0.0,
               x2 = lane + 128 < n ? s_xs[lane + 128] / s_dg[lane + 128] : 0.0;
        for (int k = n - 1; k >= 1; --k) {
            union { double d; int w[2]; } xk;
            xk.d = k >= 128 ? x2 : k >= 64 ? x1 : x0;
            xk.w[0] = __builtin_amdgcn_readlane(xk.w[0], k & 63);
            xk.w[1] = __builtin_amdgcn_readlane(xk.w[1], k & 63);
            const double* row = s_L + k * (k + 1) / 2;
            const double l0 = row[min(lane, k - 1)], l1 = row[min(lane + 64, k - 1)], l2 = row[min(lane + 128, k - 1)];
            x0 = lane < k ? x0 - l0 * xk.d : x0;
            x1 = lane + 64 < k ? x1 - l1 * xk.d : x1;
            x2 = lane + 128 < k ? x2 - l2 * xk.d : x2;
        }
        const bool bad = s_bad != 0;
        double* x_dev = global_ptr(load_uniform(&sl.x_dev));
        double* x_host = global_ptr(load_uniform(&sl.x_host));
        if (lane < n) { const double v = bad ? 0.0 : x0; x_dev[lane] = v; if (x_host) x_host[lane] = v; }
        if (lane + 64 < n) { const double v = bad ? 0.0 : x1; x_dev[lane + 64] = v; if (x_host) x_host[lane + 64] = v; }
        if (lane + 128 < n) { const double v = bad ? 0.0 : x2; x_dev[lane + 128] = v; if (x_host) x_host[lane + 128] = v; }
        if (lane == 0) global_ptr(load_uniform(&sl.ok_host))[0] = bad ? 0 : 1;  // (device-side LM: BaLmState::solve_ok)
    }
}
// Three variants -- 5 tile rows up to 13 free keyframes, 8 up to 21, 9 up to kSolveMaxFree -- and ONE launch per phase: the launcher picks the
// variant of the widest window among those still alive, narrower windows ride along in it.  Measured in the mixed loop of bench.py (64
// distinct windows, an eighth of them with 22-24 free keyframes), frames/s at 512 sequences: one variant per launch 17.9-18.2 k; a launch per
// width class, each skipping the others' windows, 16.8-17.0 k (a launch more per round and class costs more beside the other stages than
// the narrow windows save); one kernel dispatching on the window's width 17.6 k, and 19.6 against 20.2 k on the uniform batch (256 registers
// for everybody: the workgroup waits for half of a CU's register file on all four SIMDs).
template <int NT>
__device__ __forceinline__ void solve_b_body(const BaPhase& ph) {
    extern __shared__ double s_solve[];
    const BaSlotView view_ = ba_slot_view(ph, blockIdx.x);
    if (!view_.active) return;
    const int n = 6 * view_.pb.n_free;
    if (n == 0 || !view_.sl.x_dev || n > 16 * NT) return;  // (the launcher picks NT for the widest window of the launch)
    d_ba_solve_tiles<NT>(view_.sl, n, s_solve);
}
__global__ __launch_bounds__(kSolveThreads) void k_ba_solve5_b(const BaPhase ph) { solve_b_body<5>(ph); }
__global__ __launch_bounds__(kSolveThreads) void k_ba_solve_b(const BaPhase ph) { solve_b_body<8>(ph); }
__global__ __launch_bounds__(kSolveThreads) void k_ba_solve9_b(const BaPhase ph) { solve_b_body<9>(ph); }  // 22-24 free keyframes
// ---- The reduced system of an INERTIAL window on the device (LocalInertialBA / LocalLVIBA: 6 unknowns per free keyframe pose + 9 per keyframe with
// velocity / bias vertices, 375 for the 25-keyframe bLarge window; Optimizer.cc:1635-1638 solves it with g2o's sparse LinearSolverEigen).  Rounds
// 1-4 and the first half of round 5 solved it on the host (reduced_solve.hpp: 0.55 ms of a host core per window and trial after the envelope
// form, 6.4 ms before) -- in configs[3] that was most of the sixteen CPUs of a one-GPU box.  Here a workgroup per window does the same elimination:
//   * order and envelope of reduced_solve.hpp: velocity / bias unknowns first (a band: row i starts at most kLviBand columns before its diagonal),
//     the pose rows after them; the pose block C (visual Schur complement + inertial + LiDAR part) stays in LDS for the whole solve, packed;
//   * the band is eliminated column by column (right-looking): the column's entries below the pivot -- the band rows in the ring `Aring`, the
//     pose rows in the ring `Bring` of the 32 columns ahead -- are divided by the pivot, then every entry (i, j) of the rings and of C whose row and
//     column have an entry in this column takes its term l_i a_j, side by side over the workgroup; the right-hand side rides along (forward
//     substitution).  Column k + 32 enters the rings when column k leaves (its rows' envelopes start after k: nothing was due to it yet);
//   * then the dense LDL^T of C in LDS, the division by the pivots, and the backward substitution: pose rows (a wavefront sweeping columns), the
//     pose rows' terms of the velocity / bias unknowns (a thread each, rows descending), the band's rows (a wavefront, L from LDS).
// Every entry's terms come in a fixed order (columns ascending in the elimination, rows descending in the backward sweep): a window gives the same
// bits alone (k_lvi_solve) and in a lock-step batch (k_lvi_solve_b).  A zero or non-finite pivot: ok = 0 and a zero step, as on the host.
constexpr int kLviThreads = 512;  // eight wavefronts (four measured slower: 1.27 M against 0.98 M cycles per solve -- a step is a chain of LDS round trips per wavefront, and two wavefronts per SIMD take turns on them)
constexpr int kLviTilesPerWave = 7;  // 10 x 11 / 2 = 55 lower tiles of 16 x 16 over eight wavefronts (150 pose rows)
constexpr int kLviCol = 192;        // entries of a column vector: pose rows [0, 160), band rows below the panel at 160 + j
inline size_t lvi_solve_lds_bytes(int np, int ni) {
    const size_t n = (size_t)np + ni;
    const size_t region0 = std::max((size_t)np * (np + 1) / 2 + 33 * (size_t)np, (size_t)32 * ni);
    return (region0 + 32 * 33 + 8 * (size_t)kLviCol + 2 * n) * sizeof(double) + (n + (size_t)ni + 2) * sizeof(int);
}
// a workgroup barrier that waits for this wavefront's LDS traffic only: loads and stores to memory stay in flight across it
__device__ __forceinline__ void lvi_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// 1 / x by the hardware's estimate and two Newton steps (every thread of a workgroup repeats a panel's four reciprocals: the IEEE division's
// scaling and fix-up sequence was a third of a step's chain).  Deterministic; within an ulp or two of the quotient, which is all an LDL^T's
// L = (L D) / D needs -- the pivots D themselves are stored exactly.
__device__ __forceinline__ double lvi_reciprocal(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ void lvi_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ void d_lvi_solve(const LviSolveDev& q_, const double* S, const double* bs, const double lambda,
                                            double* x_dev, double* x_host, int32_t* ok_host, double* lds) {
    constexpr int T = kLviThreads, P = 4, kWaves = T / 64;
    // (the record's pointers are device or pinned host addresses: global accesses, not flat ones -- a flat load counts on the LDS counter too, and
    // the loop's barriers wait for that counter)
    LviSolveDev q = q_;
    q.first = global_ptr(q.first); q.span_end = global_ptr(q.span_end); q.rowoff = global_ptr(q.rowoff); q.hpose = global_ptr(q.hpose); q.bi = global_ptr(q.bi); q.LB = global_ptr(q.LB); q.Lband = global_ptr(q.Lband); q.hband = global_ptr(q.hband);
    S = global_ptr(S); bs = global_ptr(bs); x_dev = global_ptr(x_dev); x_host = global_ptr(x_host); ok_host = global_ptr(ok_host);
    __shared__ int s_bad;
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_in_block(), kq = lane >> 4, lj = lane & 15;
    const int n = q.n, np = q.np, ni = q.ni;
    const int tri = np * (np + 1) / 2;
    const int region0 = max(tri + 33 * np, 32 * ni);
    double* const C = lds;                    // pose block, packed lower: (r, s) at r (r + 1) / 2 + s (the band phase keeps it in registers, see acc)
    double* const Bring = C + tri;            // pose rows x the 32 columns ahead: (r, c) at 33 r + (c & 31)
    double* const Aring = lds + region0;      // band rows: (i, c) at 33 (i & 31) + (c & 31)
    double* const colL = Aring + 32 * 33;     // [4][kLviCol] the panel's columns divided by their pivots (L): pose rows, then the band rows below the panel
    double* const colA = colL + 4 * kLviCol;  // [4][kLviCol] the same entries before the division (L D)
    double* const z = colA + 4 * kLviCol;     // right-hand side -> solution, solver order
    double* const D = z + n;
    int* const first = reinterpret_cast<int*>(D + n);
    int* const rhi = first + n;               // pose rows [0, rhi[k]) can have an entry in column k
    if (tid == 0) s_bad = 0;
    for (int i = tid; i < n; i += T) first[i] = q.first[i];
    __syncthreads();
    for (int k = tid; k < ni; k += T) {
        int h = 0;
        for (int r = 0; r < np; ++r) if (first[ni + r] <= k) h = r + 1;
        rhi[k] = h;
    }
    // The pose block as 16 x 16 tiles of the lower triangle, tile n with wavefront n % 8, IN REGISTERS through the whole elimination of the band: a
    // panel's update of a tile is one v_mfma_f64_16x16x4_f64 (K = 4 = the panel's columns) on two operands out of LDS, the block itself never
    // travels.  Tile layout as in the Schur product: acc[r] of lane -> row 16 ti + lane / 16 + 4 r, column 16 tj + lane % 16.
    const int tiles = (np + 15) / 16, n_tiles = tiles * (tiles + 1) / 2;
    int my_ti[kLviTilesPerWave], my_tj[kLviTilesPerWave];
    v4d acc[kLviTilesPerWave];
#pragma unroll
    for (int t = 0; t < kLviTilesPerWave; ++t) {
        const int tn = wave + kWaves * t;
        int ti = 0;
        while ((ti + 1) * (ti + 2) / 2 <= tn) ++ti;
        my_ti[t] = tn < n_tiles ? ti : -1;
        my_tj[t] = tn - ti * (ti + 1) / 2;
        acc[t] = v4d{0, 0, 0, 0};
        if (my_ti[t] < 0) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * my_ti[t] + kq + 4 * r, col = 16 * my_tj[t] + lj;
            if (row < np && col <= row) acc[t][r] = S[(size_t)row * np + col] + q.hpose[q.rowoff[row] + (q.span_end[row] - first[ni + row]) + col];
        }
    }
    for (int j = tid; j < n; j += T) z[j] = j < ni ? q.bi[np + j] : q.bi[j - ni] + bs[j - ni];
    for (int e = tid; e < ni * 32; e += T) q.Lband[e] = 0.0;   // (entries of rows further than the rings reach below a column: structurally zero)
    // The rings are fed from global memory: the entry of (pose row tid, column c) and of (band row c, column c - 31 + tid).  Four columns enter when a
    // panel's four leave; their entries are requested a whole step earlier and the loop's barriers wait for LDS only (lvi_lds_barrier), so neither
    // these loads nor the stores of L stand in a step's way.
    const int my_f = tid < np ? first[ni + tid] : 0, my_e = tid < np ? q.span_end[tid] : 0, my_off = tid < np ? q.rowoff[tid] : 0;
    auto fetch_pose = [&](int c) -> double { return c >= my_f && c < my_e ? q.hpose[my_off + (c - my_f)] : 0.0; };
    auto fetch_band = [&](int c) -> double { return tid < 32 && c < ni ? q.hband[32 * c + tid] : 0.0; };   // (the host left the band rows 32 wide)
    auto put_column = [&](int c, double vp, double vb) {
        if (tid < np) Bring[33 * tid + (c & 31)] = vp;
        if (tid < 32) {
            const int cc = c - 31 + tid;
            if (cc >= 0) Aring[(c & 31) * 33 + (cc & 31)] = cc == c ? vb + lambda : vb;
        }
    };
    for (int c0 = 0; c0 < min(32, ni); c0 += 8) {   // (eight columns' entries requested together: one trip to memory per batch)
        double vp[8], vb[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { vp[u] = fetch_pose(c0 + u); vb[u] = fetch_band(c0 + u); }
#pragma unroll
        for (int u = 0; u < 8; ++u) if (c0 + u < ni) put_column(c0 + u, vp[u], vb[u]);
    }
    double pre_p[P], pre_b[P];
#pragma unroll
    for (int u = 0; u < P; ++u) { pre_p[u] = fetch_pose(32 + u); pre_b[u] = fetch_band(32 + u); }
    __syncthreads();
    // FOUR columns per pair of barriers.  A step's cost is not its arithmetic but its chain of LDS round trips (the column-at-a-time build: 4 k
    // cycles per column with nothing to do, 1.3 ms per solve), so a panel of four columns is eliminated per step: every thread factorises the
    // panel's 4 x 4 pivot block for itself (reciprocals: lvi_reciprocal), solves its own row against it (y_j = a_j - sum_{m<j} y_m L_jm,
    // l_j = y_j / D_j: what four single steps would have left in the row), and after ONE barrier every trailing block takes its rank-4 update on
    // the matrix unit: the pose block's tiles in registers, the rings' tiles read from LDS and written back.  Within a phase every LDS operand is
    // requested before the first result is used, with clamped addresses instead of branches (a predicated read is a block of its own with its own
    // wait: 60 waits per step in the first panel build).
    // pv: the pivot block's lower triangle, zr: the panel rows' right-hand side, a: this thread's row in the panel's columns
#define TC2LI_LVI_PANEL(pv, zr, a, y, l)                                                                           \
        _Pragma("unroll") for (int i = 0; i < P; ++i) {                                                           \
            double yy[P];                                                                                         \
            _Pragma("unroll") for (int j = 0; j <= i; ++j) {                                                      \
                double v = i < pw ? pv[i][j] : (i == j ? 1.0 : 0.0);                                              \
                _Pragma("unroll") for (int m = 0; m < j; ++m) v -= yy[m] * Lp[j][m];                              \
                yy[j] = v;                                                                                        \
                if (j < i) Lp[i][j] = v * iD[j];                                                                  \
            }                                                                                                     \
            Dp[i] = yy[i]; iD[i] = lvi_reciprocal(yy[i]);                                                         \
            double zz = i < pw ? zr[i] : 0.0;                                                                     \
            _Pragma("unroll") for (int m = 0; m < i; ++m) zz -= Lp[i][m] * zk[m];                                 \
            zk[i] = zz;                                                                                           \
        }                                                                                                         \
        if (tid == 0) {                                                                                           \
            _Pragma("unroll") for (int i = 0; i < P; ++i)                                                         \
                if (i < pw && (!(Dp[i] == Dp[i]) || Dp[i] == 0.0 || Dp[i] - Dp[i] != 0.0)) s_bad = 1;             \
        }                                                                                                         \
        _Pragma("unroll") for (int j = 0; j < P; ++j) {                                                           \
            double v = a[j];                                                                                      \
            _Pragma("unroll") for (int m = 0; m < j; ++m) v -= y[m] * Lp[j][m];                                   \
            y[j] = v; l[j] = v * iD[j];                                                                           \
        }
    for (int k = 0; k < ni; k += P) {
        const int pw = min(P, ni - k), kp = k + pw;
        const int hi = rhi[kp - 1];
        const int nb = max(0, min(31, ni - 1 - k) - pw + 1);   // band rows (= ring columns) below the panel: kp .. kp + nb - 1
        // ---- phase 1: the panel (the wavefronts that own rows: pose rows on the first three, band rows on the last; the others wait) ----
        const int jb = tid - (T - 32);             // the last half wavefront: band row kp + jb
        const bool pose_row = tid < hi, band_row = jb >= 0 && jb < nb;
        double Lp[P][P], Dp[P], iD[P], zk[P], y[P], l[P] = {0, 0, 0, 0};
        if (wave < 3 || wave == kWaves - 1) {
        double pv[P][P], zr[P], a[P];
#pragma unroll
        for (int i = 0; i < P; ++i) {
            const int ii = min(i, pw - 1);
            zr[i] = z[k + ii];
#pragma unroll
            for (int j = 0; j <= i; ++j) pv[i][j] = Aring[((k + ii) & 31) * 33 + ((k + min(j, pw - 1)) & 31)];
        }
        if (wave == kWaves - 1) {
            const int srow = ((kp + max(jb, 0)) & 31) * 33;
#pragma unroll
            for (int j = 0; j < P; ++j) { const double v = Aring[srow + ((k + min(j, pw - 1)) & 31)]; a[j] = band_row && j < pw ? v : 0.0; }
        } else {
            const int prow = 33 * min(tid, np - 1);
#pragma unroll
            for (int j = 0; j < P; ++j) { const double v = Bring[prow + ((k + min(j, pw - 1)) & 31)]; a[j] = pose_row && j < pw ? v : 0.0; }
        }
        TC2LI_LVI_PANEL(pv, zr, a, y, l)
        if (tid < 160) {   // (rows without an entry in the panel and the tiles' padding: zero operands for the matrix unit; a = 0 gave y = l = 0)
#pragma unroll
            for (int j = 0; j < P; ++j) { colA[j * kLviCol + tid] = y[j]; colL[j * kLviCol + tid] = l[j]; }
        }
        if (tid < np) {
#pragma unroll
            for (int j = 0; j < P; ++j) if (j < pw) q.LB[(size_t)(k + j) * np + tid] = l[j];
        }
        if (jb >= 0) {
#pragma unroll
            for (int j = 0; j < P; ++j) {
                colA[j * kLviCol + 160 + jb] = y[j]; colL[j * kLviCol + 160 + jb] = l[j];
                if (j < pw && band_row) q.Lband[(size_t)(k + j) * 32 + (pw - j - 1 + jb)] = l[j];
            }
        }
        if (tid == 64) {   // (the panel's own sub-diagonal entries of L)
#pragma unroll
            for (int i = 1; i < P; ++i)
#pragma unroll
                for (int j = 0; j < i; ++j) if (i < pw) q.Lband[(size_t)(k + j) * 32 + (i - j - 1)] = Lp[i][j];
        }
        }
        lvi_lds_barrier();
        // ---- phase 2: the trailing blocks ----
        // operands of the pose block's tiles (registers)
        double ma[kLviTilesPerWave], mb[kLviTilesPerWave];
#pragma unroll
        for (int t = 0; t < kLviTilesPerWave; ++t) {
            if (my_ti[t] < 0 || 16 * my_ti[t] >= hi) continue;   // (no row of the tile has an entry in the panel; uniform)
            ma[t] = colL[kq * kLviCol + 16 * my_ti[t] + lj];
            mb[t] = colA[kq * kLviCol + 16 * my_tj[t] + lj];
        }
        // the ring of band rows: four tiles of 16 row slots x 16 column slots, on wavefronts 4 .. 7
        const int ta = wave & 3, srl = 16 * (ta >> 1) + lj, scl = 16 * (ta & 1) + lj;     // this lane's row slot (A operand) and column slot (B operand, results)
        const int jia = (srl - kp) & 31, jcb = (scl - kp) & 31;
        double aa = 0.0, ab = 0.0;
        v4d ac = v4d{0, 0, 0, 0};
        int aat[4] = {-1, -1, -1, -1};
        if (wave >= 4) {
            aa = colL[kq * kLviCol + 160 + jia]; ab = colA[kq * kLviCol + 160 + jcb];
            aa = jia < nb ? aa : 0.0; ab = jcb < nb ? ab : 0.0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int sr = 16 * (ta >> 1) + kq + 4 * r, ji = (sr - kp) & 31;
                aat[r] = jcb <= ji && ji < nb ? 33 * sr + scl : -1;
                ac[r] = Aring[max(aat[r], 0)];
            }
        }
        const double zp = z[ni + min(tid, np - 1)], zb = z[min(kp + max(jb, 0), n - 1)];
        // the products
#pragma unroll
        for (int t = 0; t < kLviTilesPerWave; ++t) {
            if (my_ti[t] < 0 || 16 * my_ti[t] >= hi) continue;   // (no row of the tile has an entry in the panel; uniform)
            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(-ma[t], mb[t], acc[t], 0, 0, 0);
        }
        // the ring of pose rows x columns ahead: tiles of 16 rows x 16 ring SLOTS, tile (ti, tc) = 2 ti + tc with wavefront % 8, one after the other
        // (all three in flight at once spilled registers: 60 scratch loads per step)
#pragma unroll 1
        for (int tb = wave; 16 * (tb >> 1) < hi; tb += kWaves) {
            const int ti = tb >> 1, slot = 16 * (tb & 1) + lj;
            const int jjl = (slot - kp) & 31;                       // the slot holds column kp + jjl
            const double av = colL[kq * kLviCol + 16 * ti + lj], bv0 = colA[kq * kLviCol + 160 + jjl];
            v4d bc;
            int bat[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * ti + kq + 4 * r;
                bat[r] = row < hi && jjl < nb ? 33 * row + slot : -1;
                bc[r] = Bring[max(bat[r], 0)];
            }
            bc = __builtin_amdgcn_mfma_f64_16x16x4f64(-av, jjl < nb ? bv0 : 0.0, bc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) if (bat[r] >= 0) Bring[bat[r]] = bc[r];
        }
        if (wave >= 4) {
            ac = __builtin_amdgcn_mfma_f64_16x16x4f64(-aa, ab, ac, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) if (aat[r] >= 0) Aring[aat[r]] = ac[r];
        }
        if (pose_row) z[ni + tid] = (((zp - l[0] * zk[0]) - l[1] * zk[1]) - l[2] * zk[2]) - l[3] * zk[3];
        if (band_row) z[kp + jb] = (((zb - l[0] * zk[0]) - l[1] * zk[1]) - l[2] * zk[2]) - l[3] * zk[3];
        if (tid == 0) {
#pragma unroll
            for (int i = 0; i < P; ++i) {
                if (i >= pw) continue;
                D[k + i] = Dp[i]; z[k + i] = zk[i];
            }
        }
#pragma unroll
        for (int u = 0; u < P; ++u) if (u < pw && k + 32 + u < ni) put_column(k + 32 + u, pre_p[u], pre_b[u]);   // into the slots of the panel's columns
#pragma unroll
        for (int u = 0; u < P; ++u) { pre_p[u] = fetch_pose(k + 32 + P + u); pre_b[u] = fetch_band(k + 32 + P + u); }
        lvi_lds_barrier();
    }
#pragma unroll
    for (int t = 0; t < kLviTilesPerWave; ++t) {   // the pose block -- now the Schur complement of the band -- into LDS for its own factorisation
        if (my_ti[t] < 0) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * my_ti[t] + kq + 4 * r, col = 16 * my_tj[t] + lj;
            if (row < np && col <= row) C[((row * (row + 1)) >> 1) + col] = acc[t][r];
        }
    }
    __syncthreads();
    for (int k = 0; k < np; k += P) {
        const int pw = min(P, np - k), kp = k + pw;
        const bool mine = tid >= kp && tid < np;   // this thread's pose row lies below the panel
        const int base = ((min(max(tid, kp), np - 1) * (min(max(tid, kp), np - 1) + 1)) >> 1) + k;
        double Lp[P][P], Dp[P], iD[P], zk[P], y[P], l[P] = {0, 0, 0, 0};
        if (wave < 3) {
        double pv[P][P], zr[P], a[P];
#pragma unroll
        for (int i = 0; i < P; ++i) {
            const int ri = k + min(i, pw - 1);
            zr[i] = z[ni + ri];
#pragma unroll
            for (int j = 0; j <= i; ++j) pv[i][j] = C[((ri * (ri + 1)) >> 1) + k + min(j, pw - 1)];
        }
#pragma unroll
        for (int j = 0; j < P; ++j) { const double v = C[base + min(j, pw - 1)]; a[j] = mine && j < pw ? v : 0.0; }
        TC2LI_LVI_PANEL(pv, zr, a, y, l)
        if (tid < 160) {   // (zeros for the rows above the panel's end and the padding)
#pragma unroll
            for (int j = 0; j < P; ++j) { colA[j * kLviCol + tid] = y[j]; colL[j * kLviCol + tid] = l[j]; }
        }
        }
        lvi_lds_barrier();
        double ma[kLviTilesPerWave], mb[kLviTilesPerWave];
        v4d cc[kLviTilesPerWave];
        int cat[kLviTilesPerWave][4];
#pragma unroll
        for (int t = 0; t < kLviTilesPerWave; ++t) {   // the tiles with rows and columns beyond the panel: out of LDS, one MFMA, back
            const bool on = my_ti[t] >= 0 && 16 * my_ti[t] + 15 >= kp && 16 * my_tj[t] + 15 >= kp;   // (uniform)
            if (!on) continue;
            ma[t] = colL[kq * kLviCol + 16 * my_ti[t] + lj];
            mb[t] = colA[kq * kLviCol + 16 * my_tj[t] + lj];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * my_ti[t] + kq + 4 * r, col = 16 * my_tj[t] + lj;
                cat[t][r] = row < np && col >= kp && col <= row ? ((row * (row + 1)) >> 1) + col : -1;
                cc[t][r] = C[max(cat[t][r], 0)];
            }
        }
        const double zp = z[ni + min(tid, np - 1)];
#pragma unroll
        for (int t = 0; t < kLviTilesPerWave; ++t) {
            if (my_ti[t] < 0 || 16 * my_ti[t] + 15 < kp || 16 * my_tj[t] + 15 < kp) continue;
            cc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(-ma[t], mb[t], cc[t], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) if (cat[t][r] >= 0) C[cat[t][r]] = cc[t][r];
        }
        if (mine) {
#pragma unroll
            for (int j = 0; j < P; ++j) if (j < pw) C[base + j] = l[j];   // L where the backward sweep reads it
            z[ni + tid] = (((zp - l[0] * zk[0]) - l[1] * zk[1]) - l[2] * zk[2]) - l[3] * zk[3];
        }
        if (tid == 0) {
#pragma unroll
            for (int i = 0; i < P; ++i) {
                if (i >= pw) continue;
                D[ni + k + i] = Dp[i]; z[ni + k + i] = zk[i];
#pragma unroll
                for (int j = 0; j < i; ++j) C[(((k + i) * (k + i + 1)) >> 1) + k + j] = Lp[i][j];
            }
        }
        lvi_lds_barrier();
    }
#undef TC2LI_LVI_PANEL
    for (int j = tid; j < n; j += T) z[j] = z[j] / D[j];
    __syncthreads();
    if (wave == 0) {
        for (int k = np - 1; k >= 1; --k) {
            const double xk = z[ni + k];
            const double* row = C + ((k * (k + 1)) >> 1);
            for (int s2 = lane; s2 < k; s2 += 64) z[ni + s2] -= row[s2] * xk;
            lvi_wave_sync();
        }
    }
    __syncthreads();
    for (int c = tid; c < ni; c += T) {
        double acc1 = z[c];
        const double* lb = q.LB + (size_t)c * np;
#pragma unroll 8
        for (int r = np - 1; r >= 0; --r) acc1 -= lb[r] * z[ni + r];
        z[c] = acc1;
    }
    double* const Lb = lds;   // the band's L, [ni][32] (the pose block is done)
    __syncthreads();
    for (int e = tid; e < ni * 32; e += T) Lb[e] = q.Lband[e];
    __syncthreads();
    if (wave == 0) {
        for (int k = ni - 1; k >= 1; --k) {
            const double xk = z[k];
            const int i = k - 1 - lane;       // row i takes L_ki x_k: entry (k, i) at [i][k - i - 1]
            if (lane < 31 && i >= 0) z[i] -= Lb[i * 32 + lane] * xk;
            lvi_wave_sync();
        }
    }
    __syncthreads();
    const bool bad = s_bad != 0;
    for (int j = tid; j < n; j += T) {
        const double v = bad ? 0.0 : (j < np ? z[ni + j] : z[j - np]);
        x_dev[j] = v;
        x_host[j] = v;
    }
    if (tid == 0) ok_host[0] = bad ? 0 : 1;
}
struct LviSolveArgs { LviSolveDev q; const double *S, *bs; double lambda; double *x_dev, *x_host; int32_t* ok_host; };
__global__ __launch_bounds__(kLviThreads) void k_lvi_solve(const LviSolveArgs a) {
    extern __shared__ double s_lvi[];
    d_lvi_solve(a.q, a.S, a.bs, a.lambda, a.x_dev, a.x_host, a.ok_host, s_lvi);
}
__global__ __launch_bounds__(kLviThreads) void k_lvi_solve_b(const BaPhase ph) {
    extern __shared__ double s_lvi[];
    TC2LI_SLOT(x);
    (void)pb;
    const LviSolveDev q = load_uniform(&sl.lvi);
    if (q.n == 0) return;
    d_lvi_solve(q, load_uniform(&sl.S_out), load_uniform(&sl.bs_out), view_.lambda, load_uniform(&sl.x_dev), load_uniform(&sl.x_host), load_uniform(&sl.ok_host), s_lvi);
}
// The fused trial launch of the lock-step batch: workgroups [0, max_groups) the window's landmark groups, workgroup max_groups its LiDAR
// plane residual at the trial poses (formed in LDS as the groups form them: the launch that wrote them to memory is this one).
template <bool INERTIAL>
__device__ __forceinline__ void trial_fused_b_body(const BaPhase& ph, int max_groups, TrialLds& L) {
    TC2LI_SLOT(y);
    if (!pb.trial_fused) return;
    const int bx = blockIdx.x;
    if (bx < max_groups) {
        if (bx >= pb.n_groups) return;
        d_ba_trial_group<INERTIAL>(pb, bx, view_.xp, view_.lambda, L, sl.iposes_host);
        if (!ba_last_of(pb.ticket + 1, pb.n_groups)) return;
        d_ba_trial_close(pb, L.sum, sl.scale_out, sl.chi_trial_out);
        return;
    }
    if (!load_uniform(&sl.has_lidar)) return;
    const int np = 6 * pb.n_free;
    for (int j = threadIdx.x; j < np; j += 256) L.x[j] = view_.xp[j];
    __syncthreads();
    trial_poses_lds<INERTIAL>(pb, L.x, L.poses, false, nullptr);
    __syncthreads();
    BalmSlotView v = balm_slot_view(ph, blockIdx.y, true);
    d_balm_residual_total(v.b, reinterpret_cast<const Se3*>(L.poses));
}
__global__ __launch_bounds__(256) void k_ba_trial_fused_b(const BaPhase ph, int max_groups) {
    __shared__ TrialLds L;
    trial_fused_b_body<false>(ph, max_groups, L);
}
__global__ __launch_bounds__(256) void k_ba_trial_fused_imu_b(const BaPhase ph, int max_groups) {
    __shared__ TrialLds L;
    trial_fused_b_body<true>(ph, max_groups, L);
}
__global__ __launch_bounds__(256) void k_ba_trial_update_b(const BaPhase ph) {
    TC2LI_SLOT(y);
    const int nbp = (pb.n_points + kBacksubPerBlock - 1) / kBacksubPerBlock;
    if (pb.trial_fused || (int)blockIdx.x >= nbp + blocks256(pb.n_poses)) return;
    d_ba_trial_update(pb, blockIdx.x, nbp, view_.xp, view_.lambda, sl.iposes_host);
}
__global__ __launch_bounds__(256) void k_ba_errors_b(const BaPhase ph) {
    TC2LI_SLOT(y);
    if (pb.trial_fused || (int)blockIdx.x >= blocks256(pb.n_edges)) return;
    d_ba_errors(pb, blockIdx.x);
}
// the same, and the last workgroup of a window does k_ba_trial_reduce_b's work for it: the gain-ratio scale's landmark part (partials of
// k_ba_trial_update_b, an earlier launch) and the trial's cost
__global__ __launch_bounds__(256) void k_ba_errors_reduce_b(const BaPhase ph) {
    TC2LI_SLOT(y);
    if (pb.trial_fused || (int)blockIdx.x >= blocks256(pb.n_edges)) return;
    d_ba_errors<true>(pb, blockIdx.x);
    if (!ba_last_of(pb.ticket + 1, blocks256(pb.n_edges))) return;
    d_ba_trial_reduce<true>(pb, 0, sl.scale_out, sl.chi_trial_out);
    __syncthreads();
    d_ba_trial_reduce<true>(pb, 1, sl.scale_out, sl.chi_trial_out);
}
__global__ __launch_bounds__(256) void k_ba_trial_reduce_b(const BaPhase ph) {
    TC2LI_SLOT(y);
    if (pb.trial_fused) return;
    d_ba_trial_reduce(pb, blockIdx.x, sl.scale_out, sl.chi_trial_out);
}
__global__ __launch_bounds__(256) void k_ba_depth_b(const BaPhase ph) {
    TC2LI_SLOT(y);
    if ((int)blockIdx.x >= blocks256(pb.n_edges)) return;
    d_ba_depth(pb, blockIdx.x, sl.depth_out);
}
#undef TC2LI_SLOT

// ---- launch wrappers ----------------------------------------------------------------------------------------------
static inline int blocks(int n) { return (n + 255) / 256; }

void ba_launch_linearize(const BaProblemDev& pb, double* chi_out, double* maxdiag_out, bool want_maxdiag, hipStream_t st) {
    TC2LI_LAUNCH(k_ba_linearize, dim3(pb.n_groups + blocks(pb.n_free_edges)), dim3(256), 0, st, pb);
    TC2LI_LAUNCH(k_ba_reduce_all, dim3(pb.n_free + 1), dim3(256), 0, st, pb, chi_out);
    if (pb.n_dups) TC2LI_LAUNCH(k_ba_dups, dim3((pb.n_free + 63) / 64), dim3(64), 0, st, pb);  // duplicate (point, free pose) edges: their blocks on top
    if (want_maxdiag) TC2LI_LAUNCH(k_ba_maxdiag, dim3(2), dim3(256), 0, st, pb, maxdiag_out);
}

int ba_schur_parts(int n_slices, int group) { return (n_slices + group - 1) / group; }
// TC2LI_BA_DENSE_FULL=0 (read per call; measurements): the dense windows' product by 64 x 64 units whatever their width
static bool dense_full_form() { const char* e = getenv("TC2LI_BA_DENSE_FULL"); return !(e && atoi(e) == 0); }


void ba_launch_schur(const BaProblemDev& pb, double lambda, double lambda_pose, int n_slices, int k_per_slice, double* S_out, double* bs_out, hipStream_t st) {
    if (!pb.n_free) return;  // a free pose may carry no visual edge when the LiDAR window brings it in; no free pose: nothing to form
    const int np = 6 * pb.n_free;
    if (pb.sparse_schur) {  // the lean block-by-block product (every window of at most kSchurLeanMaxFree free keyframes)
        if (n_slices && pb.n_free > kSchurBlocksMaxFree) TC2LI_LAUNCH(k_ba_schur_lean_wide, dim3(2 * n_slices), dim3(256), schur_lean_lds_bytes(pb.n_free), st, pb, lambda);
        else if (n_slices) TC2LI_LAUNCH(k_ba_schur_lean, dim3(n_slices), dim3(256), schur_lean_lds_bytes(pb.n_free), st, pb, lambda);
    } else {
        // the dense windows (> 24 free keyframes): edge coefficients, their per-pose sums, the block-sparse MFMA product by units
        // (k_per_slice = landmark chunks per slice on this path)
        const bool full_form = pb.np_pad <= 16 * kFullTilesMax && dense_full_form();
        if (pb.n_free_edges) TC2LI_LAUNCH(k_ba_schur_coef, dim3(blocks(pb.n_free_edges)), dim3(256), 0, st, pb, lambda, full_form ? 0 : 1);
        TC2LI_LAUNCH(k_ba_reduce_coef, dim3(pb.n_free), dim3(256), 0, st, pb);
        if (full_form) {
            (void)ensure_dynamic_lds((const void*)k_ba_schur_full, (int)sizeof(FullLds));
            TC2LI_LAUNCH(k_ba_schur_full, dim3(n_slices), dim3(kFullThreads), sizeof(FullLds), st, pb, k_per_slice, lambda);
        } else {
            const int ub = (pb.np_pad / 16 + 3) / 4;
            TC2LI_LAUNCH(k_ba_schur_units, dim3(ub * (ub + 1) / 2, n_slices), dim3(256), 0, st, pb, k_per_slice, lambda);
        }
    }
    TC2LI_LAUNCH(k_ba_schur_finish, dim3(blocks(np * np)), dim3(256), 0, st, pb, lambda_pose, n_slices, S_out, bs_out);
}

void ba_launch_trial(const BaProblemDev& pb, const double* xp, double lambda, double* scale_out, double* chi_out, hipStream_t st) {
    if (pb.trial_fused) {  // one launch: landmark groups with the closing sums by the last of them (the batch's kernel body: same bits)
        TC2LI_LAUNCH(k_ba_trial_fused, dim3(pb.n_groups), dim3(256), 0, st, pb, xp, lambda, scale_out, chi_out);
        return;
    }
    const int nbp = (pb.n_points + kBacksubPerBlock - 1) / kBacksubPerBlock;
    TC2LI_LAUNCH(k_ba_trial_update, dim3(nbp + blocks(pb.n_poses)), dim3(256), 0, st, pb, nbp, xp, lambda);
    TC2LI_LAUNCH(k_ba_errors, dim3(blocks(pb.n_edges)), dim3(256), 0, st, pb);
    TC2LI_LAUNCH(k_ba_trial_reduce, dim3(2), dim3(256), 0, st, pb, scale_out, chi_out);
}

void ba_launch_depth(const BaProblemDev& pb, uint8_t* depth_pos, hipStream_t st) {
    TC2LI_LAUNCH(k_ba_depth, dim3(blocks(pb.n_edges)), dim3(256), 0, st, pb, depth_pos);
}

void ba_batch_launch_linearize(const BaPhase& ph, int n_active, const BaBatchExtent& x, bool any_maxdiag, hipStream_t st) {
    if (!n_active) return;
    const int fuse = x.fuse_linearize && !x.any_dups;  // (a window with duplicate edges takes the separate sums: k_ba_dups_b stands between them and the maxima)
    if (x.inertial) TC2LI_LAUNCH(k_ba_linearize_imu_b, dim3(x.max_groups + blocks(x.max_free_edges), n_active), dim3(256), 0, st, ph, x.max_groups, fuse);
    else TC2LI_LAUNCH(k_ba_linearize_b, dim3(x.max_groups + blocks(x.max_free_edges), n_active), dim3(256), 0, st, ph, x.max_groups, fuse);
    if (fuse) return;  // the closing sums ran in the windows' last workgroups
    TC2LI_LAUNCH(k_ba_reduce_all_b, dim3(x.max_free + 1, n_active), dim3(256), 0, st, ph);
    if (x.any_dups) TC2LI_LAUNCH(k_ba_dups_b, dim3((x.max_free + 63) / 64, n_active), dim3(64), 0, st, ph);
    if (any_maxdiag) TC2LI_LAUNCH(k_ba_maxdiag_b, dim3(2, n_active), dim3(256), 0, st, ph);
}
void ba_batch_launch_schur(const BaPhase& ph, int n_active, const BaBatchExtent& x, hipStream_t st) {
    if (!n_active || !x.max_free) return;
    if (x.max_block_parts && x.any_block_lean)  // (a wide window -- 22 .. 24 free keyframes -- takes two workgroups per part)
        TC2LI_LAUNCH(k_ba_schur_lean_b, dim3(x.max_block_parts * (x.any_block_wide ? 2 : 1), n_active), dim3(256), schur_lean_lds_bytes(x.max_block_free), st, ph);
    if (x.any_dense) {
        if (x.max_free_edges) TC2LI_LAUNCH(k_ba_schur_coef_b, dim3(blocks(x.max_free_edges), n_active), dim3(256), 0, st, ph, dense_full_form() ? 1 : 0);
        TC2LI_LAUNCH(k_ba_reduce_coef_b, dim3(x.max_free, n_active), dim3(256), 0, st, ph);
        // windows of at most 176 columns: the full-width form; wider ones (or TC2LI_BA_DENSE_FULL=0): 64 x 64 units.  Both kernels skip the
        // windows of the other kind.
        const bool full = dense_full_form();
        if (full) {
            (void)ensure_dynamic_lds((const void*)k_ba_schur_full_b, (int)sizeof(FullLds));
            TC2LI_LAUNCH(k_ba_schur_full_b, dim3(x.max_slices, n_active), dim3(kFullThreads), sizeof(FullLds), st, ph);
        }
        if (!full || x.max_np_pad > 16 * kFullTilesMax) {
            BaPhase ph2 = ph;
            ph2.pad_ = full ? 0 : 1;
            const int ub = (x.max_np_pad / 16 + 3) / 4;
            TC2LI_LAUNCH(k_ba_schur_units_b, dim3(ub * (ub + 1) / 2, x.max_slices, n_active), dim3(256), 0, st, ph2);
        }
    }
    TC2LI_LAUNCH(k_ba_schur_finish_b, dim3(blocks(36 * x.max_free * x.max_free), n_active), dim3(256), 0, st, ph);
}
void ba_batch_launch_solve(const BaPhase& ph, int n_active, const BaBatchExtent& x, hipStream_t st) {
    if (!n_active || !x.max_free) return;
    const int n = 6 * x.max_free;   // (the callers send no window beyond kSolveMaxFree free keyframes here)
    const size_t lds = (size_t)n * (n + 1) / 2 * sizeof(double);  // L, packed, for the backward sweep: 12 free keyframes 21 KB, 24: 83 KB
    if (n <= 80) {
        TC2LI_LAUNCH(k_ba_solve5_b, dim3(n_active), dim3(kSolveThreads), lds, st, ph);
    } else if (n <= 128) {
        (void)ensure_dynamic_lds((const void*)k_ba_solve_b, 96 * 1024);
        TC2LI_LAUNCH(k_ba_solve_b, dim3(n_active), dim3(kSolveThreads), lds, st, ph);
    } else {
        (void)ensure_dynamic_lds((const void*)k_ba_solve9_b, 96 * 1024);
        TC2LI_LAUNCH(k_ba_solve9_b, dim3(n_active), dim3(kSolveThreads), lds, st, ph);
    }
}
// whether the two kernels get the LDS of their largest window (25 free keyframes: 160 KB, all a CU has); asked once, before a window is promised
// to them -- a refusal leaves every window to the host's solver instead of surfacing as a launch error later
bool lvi_device_solve_available() {
    static const bool ok = ensure_dynamic_lds((const void*)k_lvi_solve_b, (int)lvi_solve_lds_bytes(kLviMaxPoseRows, 9 * (kLviMaxPoseRows / 6))) &&
                           ensure_dynamic_lds((const void*)k_lvi_solve, (int)lvi_solve_lds_bytes(kLviMaxPoseRows, 9 * (kLviMaxPoseRows / 6)));
    return ok;
}
void lvi_batch_launch_solve(const BaPhase& ph, int n_active, int max_np, int max_ni, hipStream_t st) {
    if (!n_active || max_ni <= 0) return;
    const size_t lds = lvi_solve_lds_bytes(max_np, max_ni);
    TC2LI_LAUNCH(k_lvi_solve_b, dim3(n_active), dim3(kLviThreads), lds, st, ph);
}
void lvi_launch_solve(const LviSolveDev& q, const double* S, const double* bs, double lambda, double* x_dev, double* x_host, int32_t* ok_host, hipStream_t st) {
    if (q.n <= 0) return;
    const LviSolveArgs a{q, S, bs, lambda, x_dev, x_host, ok_host};
    TC2LI_LAUNCH(k_lvi_solve, dim3(1), dim3(kLviThreads), lvi_solve_lds_bytes(q.np, q.ni), st, a);
}
void ba_batch_launch_trial(const BaPhase& ph, int n_active, const BaBatchExtent& x, hipStream_t st) {
    if (!n_active) return;
    if (x.any_trial_fused) {  // + 1: the windows' LiDAR residual at the trial poses rides in the same launch
        if (x.inertial) TC2LI_LAUNCH(k_ba_trial_fused_imu_b, dim3(x.max_groups + 1, n_active), dim3(256), 0, st, ph, x.max_groups);
        else TC2LI_LAUNCH(k_ba_trial_fused_b, dim3(x.max_groups + 1, n_active), dim3(256), 0, st, ph, x.max_groups);
    }
    if (!x.any_trial_unfused) return;
    TC2LI_LAUNCH(k_ba_trial_update_b, dim3((x.max_points + kBacksubPerBlock - 1) / kBacksubPerBlock + blocks(x.max_poses), n_active), dim3(256), 0, st, ph);
    if (x.fuse_trial) {
        TC2LI_LAUNCH(k_ba_errors_reduce_b, dim3(blocks(x.max_edges), n_active), dim3(256), 0, st, ph);
        return;
    }
    TC2LI_LAUNCH(k_ba_errors_b, dim3(blocks(x.max_edges), n_active), dim3(256), 0, st, ph);
    TC2LI_LAUNCH(k_ba_trial_reduce_b, dim3(2, n_active), dim3(256), 0, st, ph);
}
void ba_batch_launch_depth(const BaPhase& ph, int n_active, const BaBatchExtent& x, hipStream_t st) {
    if (n_active) TC2LI_LAUNCH(k_ba_depth_b, dim3(blocks(x.max_edges), n_active), dim3(256), 0, st, ph);
}

}  // namespace tc2li
