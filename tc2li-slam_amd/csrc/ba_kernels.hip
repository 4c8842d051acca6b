// gfx950 kernels of the local bundle adjustment (visual part of Optimizer::LocalBundleAdjustment /
// OptimizerWithLidar::LocalLVBundleAdjustment, SF/src/Optimizer.cc:1118, SF/src/OptimizerWithLidar.cc:60), i.e. the
// numerical work of g2o's BlockSolver_6_3 with Schur complement (Thirdparty/g2o/g2o/core/block_solver.hpp:353-607):
//   k_ba_linearize      one thread per edge: error, Huber weight, Jacobians, the edge's blocks of J^T W J and J^T W r
//   k_ba_reduce_points  one thread per landmark: Hll, b_l from its edges (CSR, fixed order)
//   k_ba_reduce_poses   one workgroup per free pose: Hpp, b_p from its edges (LDS tree, fixed order)
//   k_ba_schur_points   (Hll + lambda I)^-1 and D^-1 b_l per landmark
//   k_ba_schur_edges    per edge W D^-1 and W (6x3) scattered into the two k-major GEMM operands; W D^-1 b_l
//   k_ba_schur_gemm     S_part = sum_k (W D^-1)[:,k] W[:,k]^T with v_mfma_f64_16x16x4_f64, split over k
//   k_ba_schur_finish   S = Hpp + lambda I - sum S_part, b_s = b_p - coefficients
//   k_ba_backsub        x_l = D^-1 (b_l - W^T x_p), trial points, the landmark part of the gain-ratio scale
//   k_ba_update_poses   trial poses exp(x_p) * T
//   k_ba_errors         robust chi2 at the trial estimate
// All arithmetic is double precision; reductions run in a fixed order, so results are reproducible run to run.
#include <hip/hip_runtime.h>
#pragma clang fp contract(off)
#include <stdint.h>

#include "ba_device.hpp"

namespace tc2li {

typedef double v4d __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void edge_state(const Se3* __restrict__ poses, const double* __restrict__ points, const BaEdge& e,
                                           const CameraD& cam, double p[3], double err[3], int& dim, double& chi2) {
    se3_map(poses[e.pose], points + 3 * (size_t)e.point, p);
    dim = edge_error(p, e, cam, err);
    chi2 = 0;
    for (int d = 0; d < dim; ++d) chi2 += err[d] * e.info * err[d];
}

__global__ __launch_bounds__(256) void k_ba_linearize(BaProblemDev pb) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= pb.n_edges) return;
    const BaEdge ed = pb.edges[e];
    double p[3], err[3], c2;
    int dim;
    edge_state(pb.poses, pb.points, ed, pb.cam, p, err, dim, c2);
    const bool stereo = ed.ur >= 0;
    double rho0, rho1;
    huber(c2, stereo ? pb.delta_stereo : pb.delta_mono, stereo ? pb.dsqr_stereo : pb.dsqr_mono, rho0, rho1);
    pb.chi2[e] = c2;
    pb.rho0[e] = rho0;
    double R[9], A[9], B[18];
    quat_to_matrix(pb.poses[ed.pose].q, R);
    point_jacobian(p, R, stereo, pb.cam, A);
    pose_jacobian(p, stereo, false, pb.cam, B);
    const double w = rho1 * ed.info;
    double wr[3];  // omega_r = -rho' * Omega * e
    for (int d = 0; d < 3; ++d) wr[d] = d < dim ? -(ed.info * err[d]) * rho1 : 0.0;
    // landmark block: A^T W A (upper 6) and A^T omega_r
    double* cl = pb.contrib_l + 9 * (size_t)e;
    {
        int h = 0;
        for (int r = 0; r < 3; ++r)
            for (int c = r; c < 3; ++c) {
                double s = 0;
                for (int d = 0; d < dim; ++d) s += A[3 * d + r] * w * A[3 * d + c];
                cl[h++] = s;
            }
        for (int r = 0; r < 3; ++r) {
            double s = 0;
            for (int d = 0; d < dim; ++d) s += A[3 * d + r] * wr[d];
            cl[6 + r] = s;
        }
    }
    if (pb.pose_var[ed.pose] >= 0) {
        double* cp = pb.contrib_p + 27 * (size_t)e;
        int h = 0;
        for (int r = 0; r < 6; ++r)
            for (int c = r; c < 6; ++c) {
                double s = 0;
                for (int d = 0; d < dim; ++d) s += B[6 * d + r] * w * B[6 * d + c];
                cp[h++] = s;
            }
        for (int r = 0; r < 6; ++r) {
            double s = 0;
            for (int d = 0; d < dim; ++d) s += B[6 * d + r] * wr[d];
            cp[21 + r] = s;
        }
        double* W = pb.W + 18 * (size_t)e;  // Hpl block: B^T W A (6 x 3)
        for (int r = 0; r < 6; ++r)
            for (int c = 0; c < 3; ++c) {
                double s = 0;
                for (int d = 0; d < dim; ++d) s += B[6 * d + r] * w * A[3 * d + c];
                W[3 * r + c] = s;
            }
    }
}

__global__ __launch_bounds__(256) void k_ba_reduce_points(BaProblemDev pb) {
    const int l = blockIdx.x * 256 + threadIdx.x;
    if (l >= pb.n_points) return;
    double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int k = pb.pt_off[l]; k < pb.pt_off[l + 1]; ++k) {
        const double* c = pb.contrib_l + 9 * (size_t)pb.pt_edges[k];
        for (int i = 0; i < 9; ++i) acc[i] += c[i];
    }
    for (int i = 0; i < 6; ++i) pb.Hll[6 * (size_t)l + i] = acc[i];
    for (int i = 0; i < 3; ++i) pb.bl[3 * (size_t)l + i] = acc[6 + i];
    pb.diag_l[l] = fmax(fabs(acc[0]), fmax(fabs(acc[3]), fabs(acc[5])));
}

// Fixed-order block sum of `width` values per item over the items [begin, end) of an index list.
template <int WIDTH>
__device__ __forceinline__ void block_sum_items(const double* __restrict__ items, const int* __restrict__ index, int begin, int end,
                                                double* s_part /*[256][WIDTH]*/, double* out) {
    double acc[WIDTH];
    for (int i = 0; i < WIDTH; ++i) acc[i] = 0;
    for (int k = begin + (int)threadIdx.x; k < end; k += 256) {
        const double* c = items + WIDTH * (size_t)(index ? index[k] : k);
        for (int i = 0; i < WIDTH; ++i) acc[i] += c[i];
    }
    for (int i = 0; i < WIDTH; ++i) s_part[threadIdx.x * WIDTH + i] = acc[i];
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if ((int)threadIdx.x < s)
            for (int i = 0; i < WIDTH; ++i) s_part[threadIdx.x * WIDTH + i] += s_part[(threadIdx.x + s) * WIDTH + i];
        __syncthreads();
    }
    if (threadIdx.x < WIDTH) out[threadIdx.x] = s_part[threadIdx.x];
    __syncthreads();
}

__global__ __launch_bounds__(256) void k_ba_reduce_poses(BaProblemDev pb) {
    __shared__ double s_part[256 * 27];
    const int i = blockIdx.x;  // free pose
    block_sum_items<27>(pb.contrib_p, pb.pv_edges, pb.pv_off[i], pb.pv_off[i + 1], s_part, pb.Hpp + 27 * (size_t)i);
    if (threadIdx.x == 0) {
        const double* h = pb.Hpp + 27 * (size_t)i;
        // diagonal entries of the packed upper triangle: 0, 6, 11, 15, 18, 20
        pb.diag_p[i] = fmax(fmax(fabs(h[0]), fabs(h[6])), fmax(fmax(fabs(h[11]), fabs(h[15])), fmax(fabs(h[18]), fabs(h[20]))));
    }
}

// out[0] = sum(in[0..n)) (or max when MAX) in a fixed order, one workgroup of 1024 threads.
template <bool MAX>
__global__ __launch_bounds__(1024) void k_reduce(const double* __restrict__ in, int n, double* __restrict__ out) {
    __shared__ double s[1024];
    double a = 0;
    for (int k = threadIdx.x; k < n; k += 1024) a = MAX ? fmax(a, in[k]) : a + in[k];
    s[threadIdx.x] = a;
    __syncthreads();
    for (int st = 512; st >= 1; st >>= 1) {
        if ((int)threadIdx.x < st) s[threadIdx.x] = MAX ? fmax(s[threadIdx.x], s[threadIdx.x + st]) : s[threadIdx.x] + s[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = s[0];
}

__global__ __launch_bounds__(256) void k_ba_schur_points(BaProblemDev pb, double lambda) {
    const int l = blockIdx.x * 256 + threadIdx.x;
    if (l >= pb.n_points) return;
    const double* h = pb.Hll + 6 * (size_t)l;
    // D = Hll + lambda I (symmetric: h = [00 01 02 11 12 22]); inverse by cofactors like Eigen's fixed 3x3 inverse
    const double d00 = h[0] + lambda, d01 = h[1], d02 = h[2], d11 = h[3] + lambda, d12 = h[4], d22 = h[5] + lambda;
    const double c00 = d11 * d22 - d12 * d12, c01 = d12 * d02 - d01 * d22, c02 = d01 * d12 - d11 * d02;
    const double det = d00 * c00 + d01 * c01 + d02 * c02, id = 1.0 / det;
    double Di[9];
    Di[0] = c00 * id; Di[1] = (d02 * d12 - d01 * d22) * id; Di[2] = (d01 * d12 - d02 * d11) * id;
    Di[3] = c01 * id; Di[4] = (d00 * d22 - d02 * d02) * id; Di[5] = (d02 * d01 - d00 * d12) * id;
    Di[6] = c02 * id; Di[7] = (d01 * d02 - d00 * d12) * id; Di[8] = (d00 * d11 - d01 * d01) * id;
    double* o = pb.Dinv + 9 * (size_t)l;
    for (int i = 0; i < 9; ++i) o[i] = Di[i];
    const double* b = pb.bl + 3 * (size_t)l;
    for (int r = 0; r < 3; ++r) pb.db[3 * (size_t)l + r] = Di[3 * r] * b[0] + Di[3 * r + 1] * b[1] + Di[3 * r + 2] * b[2];
}

__global__ __launch_bounds__(256) void k_ba_schur_edges(BaProblemDev pb) {
    const int k = blockIdx.x * 256 + threadIdx.x;  // index into the list of edges with a free pose
    if (k >= pb.n_free_edges) return;
    const int e = pb.pv_edges[k];
    const BaEdge ed = pb.edges[e];
    const int i = pb.pose_var[ed.pose], l = ed.point;
    const double* W = pb.W + 18 * (size_t)e;
    const double* Di = pb.Dinv + 9 * (size_t)l;
    const double* db = pb.db + 3 * (size_t)l;
    double* ce = pb.coef_e + 6 * (size_t)e;
    for (int r = 0; r < 6; ++r) {
        ce[r] = W[3 * r] * db[0] + W[3 * r + 1] * db[1] + W[3 * r + 2] * db[2];
        for (int c = 0; c < 3; ++c) {
            const double y = W[3 * r] * Di[c] + W[3 * r + 1] * Di[3 + c] + W[3 * r + 2] * Di[6 + c];
            const size_t at = (size_t)(3 * l + c) * pb.np_pad + 6 * i + r;
            pb.AT[at] = y;          // (W D^-1)^T, k-major
            pb.BT[at] = W[3 * r + c];  // W^T, k-major
        }
    }
}

__global__ __launch_bounds__(256) void k_ba_reduce_coef(BaProblemDev pb) {
    __shared__ double s_part[256 * 6];
    const int i = blockIdx.x;
    block_sum_items<6>(pb.coef_e, pb.pv_edges, pb.pv_off[i], pb.pv_off[i + 1], s_part, pb.coef + 6 * (size_t)i);
}

// S_part[slice] (np_pad x np_pad, row-major) = sum over the slice's k of AT[k][:]^T BT[k][:]; one wavefront per
// 16x16 tile and k-slice.  Fragment layout of v_mfma_f64_16x16x4_f64 (checked on gfx950, tools/dbg/mfma_f64_test.hip):
// A: lane -> A[i = lane % 16][k = lane / 16]; B: lane -> B[k = lane / 16][j = lane % 16];
// D: lane, r -> D[i = lane / 16 + 4 r][j = lane % 16].
__global__ __launch_bounds__(64) void k_ba_schur_gemm(const double* __restrict__ AT, const double* __restrict__ BT, int np_pad,
                                                      int k_total, int k_per_slice, double* __restrict__ S_part) {
    const int lane = threadIdx.x, tiles = np_pad / 16;
    const int ti = blockIdx.x / tiles, tj = blockIdx.x % tiles, slice = blockIdx.y;
    const int k0 = slice * k_per_slice, k1 = min(k0 + k_per_slice, k_total);
    v4d acc = {0, 0, 0, 0};
    const int i = lane % 16, kk = lane / 16;
    for (int k = k0; k < k1; k += 4) {
        const int kr = k + kk;
        double a = 0, b = 0;
        if (kr < k1) {
            a = AT[(size_t)kr * np_pad + 16 * ti + i];
            b = BT[(size_t)kr * np_pad + 16 * tj + i];
        }
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    double* out = S_part + (size_t)slice * np_pad * np_pad;
    for (int r = 0; r < 4; ++r) out[(size_t)(16 * ti + kk + 4 * r) * np_pad + 16 * tj + i] = acc[r];
}

__global__ __launch_bounds__(256) void k_ba_schur_finish(BaProblemDev pb, double lambda, int n_slices, double* __restrict__ S_out,
                                                         double* __restrict__ bs_out) {
    const int np = 6 * pb.n_free, idx = blockIdx.x * 256 + threadIdx.x;
    if (idx < np * np) {
        const int r = idx / np, c = idx % np;
        double s = 0;
        if (r / 6 == c / 6) {  // Hpp is block diagonal in the visual problem
            const int a = min(r % 6, c % 6), b = max(r % 6, c % 6);
            const int packed = a * 6 - a * (a - 1) / 2 + (b - a);
            s = pb.Hpp[27 * (size_t)(r / 6) + packed];
            if (r == c) s += lambda;
        }
        double sub = 0;
        for (int k = 0; k < n_slices; ++k) sub += pb.S_part[(size_t)k * pb.np_pad * pb.np_pad + (size_t)r * pb.np_pad + c];
        S_out[idx] = s - sub;
    }
    if (idx < np) {
        const double bp = pb.Hpp[27 * (size_t)(idx / 6) + 21 + idx % 6];
        bs_out[idx] = bp - pb.coef[idx];
        bs_out[np + idx] = bp;
    }
}

__global__ __launch_bounds__(256) void k_ba_backsub(BaProblemDev pb, const double* __restrict__ xp, double lambda) {
    const int l = blockIdx.x * 256 + threadIdx.x;
    if (l >= pb.n_points) return;
    double cl[3] = {pb.bl[3 * (size_t)l], pb.bl[3 * (size_t)l + 1], pb.bl[3 * (size_t)l + 2]};
    for (int k = pb.pt_off[l]; k < pb.pt_off[l + 1]; ++k) {
        const int e = pb.pt_edges[k];
        const int i = pb.pose_var[pb.edges[e].pose];
        if (i < 0) continue;
        const double* W = pb.W + 18 * (size_t)e;
        for (int c = 0; c < 3; ++c) {
            double s = 0;
            for (int r = 0; r < 6; ++r) s += W[3 * r + c] * xp[6 * i + r];
            cl[c] -= s;
        }
    }
    const double* Di = pb.Dinv + 9 * (size_t)l;
    double sc = 0;
    for (int r = 0; r < 3; ++r) {
        const double x = Di[3 * r] * cl[0] + Di[3 * r + 1] * cl[1] + Di[3 * r + 2] * cl[2];
        pb.points_trial[3 * (size_t)l + r] = pb.points[3 * (size_t)l + r] + x;
        sc += x * (lambda * x + pb.bl[3 * (size_t)l + r]);
    }
    pb.scale_l[l] = sc;
}

__global__ void k_ba_update_poses(BaProblemDev pb, const double* __restrict__ xp) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= pb.n_poses) return;
    const int i = pb.pose_var[k];
    if (i < 0) { pb.poses_trial[k] = pb.poses[k]; return; }
    double u[6];
    for (int r = 0; r < 6; ++r) u[r] = xp[6 * i + r];
    pb.poses_trial[k] = se3_exp_mul(u, pb.poses[k]);
}

__global__ __launch_bounds__(256) void k_ba_errors(BaProblemDev pb, const Se3* __restrict__ poses, const double* __restrict__ points) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= pb.n_edges) return;
    const BaEdge ed = pb.edges[e];
    double p[3], err[3], c2;
    int dim;
    edge_state(poses, points, ed, pb.cam, p, err, dim, c2);
    const bool stereo = ed.ur >= 0;
    double rho0, rho1;
    huber(c2, stereo ? pb.delta_stereo : pb.delta_mono, stereo ? pb.dsqr_stereo : pb.dsqr_mono, rho0, rho1);
    pb.chi2[e] = c2;
    pb.rho0[e] = rho0;
}

__global__ __launch_bounds__(256) void k_ba_depth(BaProblemDev pb, uint8_t* __restrict__ depth_pos) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= pb.n_edges) return;
    const BaEdge ed = pb.edges[e];
    double p[3];
    se3_map(pb.poses[ed.pose], pb.points + 3 * (size_t)ed.point, p);
    depth_pos[e] = p[2] > 0.0;
}

// ---- launch wrappers ----------------------------------------------------------------------------------------------
static inline int blocks(int n) { return (n + 255) / 256; }

void ba_launch_linearize(const BaProblemDev& pb, double* chi_out, double* maxdiag_out, hipStream_t st) {
    hipLaunchKernelGGL(k_ba_linearize, dim3(blocks(pb.n_edges)), dim3(256), 0, st, pb);
    hipLaunchKernelGGL(k_ba_reduce_points, dim3(blocks(pb.n_points)), dim3(256), 0, st, pb);
    if (pb.n_free) hipLaunchKernelGGL(k_ba_reduce_poses, dim3(pb.n_free), dim3(256), 0, st, pb);
    hipLaunchKernelGGL(k_reduce<false>, dim3(1), dim3(1024), 0, st, pb.rho0, pb.n_edges, chi_out);
    hipLaunchKernelGGL(k_reduce<true>, dim3(1), dim3(1024), 0, st, pb.diag_l, pb.n_points, maxdiag_out);
    hipLaunchKernelGGL(k_reduce<true>, dim3(1), dim3(1024), 0, st, pb.diag_p, pb.n_free, maxdiag_out + 1);
}

void ba_launch_schur(const BaProblemDev& pb, double lambda, int n_slices, int k_per_slice, double* S_out, double* bs_out, hipStream_t st) {
    hipLaunchKernelGGL(k_ba_schur_points, dim3(blocks(pb.n_points)), dim3(256), 0, st, pb, lambda);
    if (pb.n_free) {  // a free pose may carry no visual edge when the LiDAR window brings it in
        if (pb.n_free_edges) hipLaunchKernelGGL(k_ba_schur_edges, dim3(blocks(pb.n_free_edges)), dim3(256), 0, st, pb);
        hipLaunchKernelGGL(k_ba_reduce_coef, dim3(pb.n_free), dim3(256), 0, st, pb);
        const int tiles = pb.np_pad / 16;
        hipLaunchKernelGGL(k_ba_schur_gemm, dim3(tiles * tiles, n_slices), dim3(64), 0, st, pb.AT, pb.BT, pb.np_pad, 3 * pb.n_points,
                           k_per_slice, pb.S_part);
        const int np = 6 * pb.n_free;
        hipLaunchKernelGGL(k_ba_schur_finish, dim3(blocks(np * np)), dim3(256), 0, st, pb, lambda, n_slices, S_out, bs_out);
    }
}

void ba_launch_trial(const BaProblemDev& pb, const double* xp, double lambda, double* scale_out, double* chi_out, hipStream_t st) {
    hipLaunchKernelGGL(k_ba_backsub, dim3(blocks(pb.n_points)), dim3(256), 0, st, pb, xp, lambda);
    hipLaunchKernelGGL(k_ba_update_poses, dim3((pb.n_poses + 63) / 64), dim3(64), 0, st, pb, xp);
    hipLaunchKernelGGL(k_ba_errors, dim3(blocks(pb.n_edges)), dim3(256), 0, st, pb, pb.poses_trial, pb.points_trial);
    hipLaunchKernelGGL(k_reduce<false>, dim3(1), dim3(1024), 0, st, pb.scale_l, pb.n_points, scale_out);
    hipLaunchKernelGGL(k_reduce<false>, dim3(1), dim3(1024), 0, st, pb.rho0, pb.n_edges, chi_out);
}

void ba_launch_depth(const BaProblemDev& pb, uint8_t* depth_pos, hipStream_t st) {
    hipLaunchKernelGGL(k_ba_depth, dim3(blocks(pb.n_edges)), dim3(256), 0, st, pb, depth_pos);
}

}  // namespace tc2li
