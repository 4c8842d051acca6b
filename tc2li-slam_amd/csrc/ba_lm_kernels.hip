// The Levenberg-Marquardt decisions of a lock-step window on gfx950 (round 6; ba_device.hpp: BaLmState).
//   g2o::OptimizationAlgorithmLevenberg::solve      Thirdparty/g2o/g2o/core/optimization_algorithm_levenberg.cpp:61-169
//   computeLambdaInit / computeScale                 :171-201
//   SparseOptimizer::optimize's stop rules           Thirdparty/g2o/g2o/core/sparse_optimizer.cpp:363-441 (the reference's `_nBad >= 3`)
//   EdgeLidarSE3::computeError / linearizeOplus /
//     computeQuadraticFormLidarRes                   SF/include/G2oTypesWithLidar.h:88-236
//   LidarCovisRes::ComputeJandHSE3                   SF/src/LidarRes.cc:136-186
// Two kernels, one wavefront per window.  They do what ba_batch_lockstep's host steps did between two phases -- the same operations in the same
// order on the same operands (the shared host / device functions of ba_math.hpp and balm_math.hpp), so the window's bits do not depend on where
// its loop runs -- and leave `status`: what the window needs next.  The trip to the host that followed every phase is gone: the kernels of the
// next phase read lambda / parity / status from the state (ba_lm_view).
#include <hip/hip_runtime.h>

#include "launch.hpp"
#pragma clang fp contract(off)
#include <float.h>
#include <stdint.h>

#include "ba_device.hpp"
#include "balm_device.hpp"

namespace tc2li {

namespace {

constexpr int kLmThreads = 64;      // k_ba_lm_decide_b: one wavefront
constexpr int kLmBeginThreads = 256; // k_ba_lm_begin_b: the (6W)^2 change of variables one output entry per thread
constexpr int kLmMaxW = 7;          // windows of the batched LiDAR kernels (ba_batch_lockstep sends wider ones through the per-window path)
constexpr int kLmMaxNp = 6 * kSolveMaxFree;

struct LmWindow {
    const BaBatchSlot* sl;
    BaLmState* lm;
    int n_free, np, has_lidar;
};
__device__ __forceinline__ LmWindow lm_window(const BaPhase& ph, int pos) {
    __builtin_amdgcn_s_setprio(3);
    LmWindow w;
    w.sl = ph.table + ba_phase_window(ph, pos);
    w.lm = global_ptr(load_uniform(&w.sl->lm));
    w.n_free = load_uniform(&w.sl->pb.n_free);
    w.np = 6 * w.n_free;
    w.has_lidar = load_uniform(&w.sl->has_lidar);
    return w;
}
// Block (a, b) of D^T H D entry by entry (balm_change_block's two products, balm_math.hpp: from the left first when a <= b, from the right
// first otherwise; every entry the same six-term sum in the same order): pass 0 writes the first product of H into T, pass 1 the second of T into H
__device__ __forceinline__ void lm_change_pass(const double* in, double* out, const double* DT, int W, int pass, int tid, int nthreads) {
    const int n = 6 * W;
    for (int e = tid; e < n * n; e += nthreads) {
        const int row = e / n, col = e - row * n, a = row / 6, r = row - 6 * a, b = col / 6, c = col - 6 * b;
        const double* blk = in + (size_t)(6 * a) * n + 6 * b;
        double sum = 0;
        if ((pass == 0) == (a <= b)) { const double* D = DT + 36 * a; for (int k = 0; k < 6; ++k) sum += D[6 * r + k] * blk[(size_t)k * n + c]; }
        else { const double* D = DT + 36 * b; for (int k = 0; k < 6; ++k) sum += blk[(size_t)r * n + k] * D[6 * c + k]; }
        out[e] = sum;
    }
}
// the state as sixteen 8-byte words: one lane each
__device__ __forceinline__ void lm_store(const BaLmState& s, BaLmState* dst, BaLmState* mirror, int lane, unsigned long long* stage) {
    if (lane == 0) *reinterpret_cast<BaLmState*>(stage) = s;
    __syncthreads();
    if (lane < (int)(sizeof(BaLmState) / 8)) {
        const unsigned long long v = stage[lane];
        reinterpret_cast<unsigned long long*>(dst)[lane] = v;
        if (mirror) reinterpret_cast<unsigned long long*>(mirror)[lane] = v;
    }
}

}  // namespace

// After the linearisation of a window in kLmIterate: g2o's computeActiveErrors + the edges' constructQuadraticForm as far as the host did them
// (ba_batch_lockstep's step between phases A and B), then status = kLmTrial.
__global__ __launch_bounds__(kLmBeginThreads) void k_ba_lm_begin_b(const BaPhase ph) {
    constexpr int kLmThreads = kLmBeginThreads;
    const LmWindow w = lm_window(ph, blockIdx.x);
    if (!w.lm || load_uniform(&w.lm->status) != kLmIterate) return;
    __shared__ double s_H[36 * kLmMaxW * kLmMaxW], s_T[36 * kLmMaxW * kLmMaxW], s_J[6 * kLmMaxW], s_DT[36 * kLmMaxW], s_max[kLmBeginThreads / 64];
    __shared__ LidarPose s_twl[kLmMaxW];
    __shared__ int s_var[kLmMaxW];
    __shared__ unsigned long long s_stage[sizeof(BaLmState) / 8];
    const int lane = threadIdx.x, np = w.np;
    BaLmState st = *w.lm;
    const double* sc = global_ptr(load_uniform(&w.sl->chi_out));  // [0] robust cost, [1] / [2] largest landmark / pose diagonal (the linearisation's sums)
    double currentChi = sc[0];
    double max_pose_diag = sc[2];
    const bool want_maxdiag = st.it == 0 && !(load_uniform(&w.sl->lambda_init) > 0);
    if (w.has_lidar) {
        BalmDev b = load_uniform(&w.sl->balm);
        b.out = global_ptr(b.out); b.pose_index = global_ptr(b.pose_index);
        const int W = b.W, n = 6 * W;
        const double information = load_uniform(&w.sl->lidar_information);
        double* JH = global_ptr(load_uniform(&w.sl->lidar_JH));
        double* Hl = global_ptr(const_cast<double*>(load_uniform(&w.sl->Hl)));
        double* bl = global_ptr(const_cast<double*>(load_uniform(&w.sl->bl_lidar)));
        const int32_t* pose_var = global_ptr(load_uniform(&w.sl->pb.pose_var));
        // BalmTerm::finish_error (EdgeLidarSE3::computeError: the residual of the accepted estimate, the Hessian kept while the cost grows)
        const double r = b.n_planes ? b.out[0] : 0.0;
        st.lidar_error = r;
        st.r1 = st.r2;
        st.r2 = r;
        st.is_calc_hess = !(st.r1 - st.r2 < 0) ? 1 : 0;
        currentChi = st.lidar_error * information * st.lidar_error + currentChi;
        // BalmTerm::finish_linearization: the Hessian pass's JacT / Hessian (LiDAR-pose increments) -> the camera vertices' se3 increments
        if (st.is_calc_hess) {
            ++st.hessian_evaluations;
            if (!b.n_planes) {
                for (int k = lane; k < n; k += kLmThreads) s_J[k] = 0.0;
                for (int k = lane; k < n * n; k += kLmThreads) s_H[k] = 0.0;
            } else {
                for (int k = lane; k < n; k += kLmThreads) s_J[k] = b.out[1 + k];
                for (int k = lane; k < n * n; k += kLmThreads) s_H[k] = b.out[1 + n + k];
                const double* at = b.out + 2 + n + n * n;   // the LiDAR poses the derivatives were taken at
                for (int k = lane; k < 12 * W; k += kLmThreads) reinterpret_cast<double*>(s_twl)[k] = at[k];
                __syncthreads();
                if (lane < W) {
                    const BalmCameraFrame F = balm_camera_frame(b.Tcl);
                    balm_camera_se3_D(s_twl[lane], F, s_J + 6 * lane, s_DT + 36 * lane);
                }
                __syncthreads();
                lm_change_pass(s_H, s_T, s_DT, W, 0, lane, kLmThreads);
                __syncthreads();
                lm_change_pass(s_T, s_H, s_DT, W, 1, lane, kLmThreads);
            }
            __syncthreads();
            for (int k = lane; k < n; k += kLmThreads) JH[k] = s_J[k];
            for (int k = lane; k < n * n; k += kLmThreads) JH[n + k] = s_H[k];
        } else {
            for (int k = lane; k < n; k += kLmThreads) s_J[k] = JH[k];
            for (int k = lane; k < n * n; k += kLmThreads) s_H[k] = JH[n + k];
        }
        if (lane < W) s_var[lane] = pose_var[b.pose_index[lane]];
        // BalmTerm::add_quadratic_form into the zeroed (6K)^2 block and gradient: every entry is written by one term (0 + h: the host's sum).
        // The entries written are the same in every iteration (the window's keyframes do not move in the numbering): zeroed once per call.
        if (st.it == 0) {
            for (int k = lane; k < np * np; k += kLmThreads) Hl[k] = 0.0;
            for (int k = lane; k < np; k += kLmThreads) bl[k] = 0.0;
        }
        __syncthreads();
        // The reference reads the 6x6 blocks at ELEMENT offsets (i, i) / (i, j) of the 6W x 6W Hessian (G2oTypesWithLidar.h:168-236); kept as is.
        for (int t = lane; t < 42 * W; t += kLmThreads) {
            const int i = t / 42, q = t % 42, vi = s_var[i];
            if (vi < 0) continue;
            if (q < 6) { bl[6 * vi + q] = 0.0 - information * s_J[6 * i + q]; continue; }
            const int rr = (q - 6) / 6, c = (q - 6) % 6;
            Hl[(size_t)(6 * vi + rr) * np + 6 * vi + c] = 0.0 + s_H[(i + rr) * n + i + c] * information;
        }
        for (int t = lane; t < 36 * (W * (W - 1) / 2); t += kLmThreads) {
            int pair = t / 36, i = 0;
            const int rc = t % 36, rr = rc / 6, c = rc % 6;
            while (pair >= W - 1 - i) { pair -= W - 1 - i; ++i; }
            const int j = i + 1 + pair, vi = s_var[i], vj = s_var[j];
            if (vi < 0 || vj < 0) continue;
            const double h = s_H[(i + rr) * n + j + c] * information;
            Hl[(size_t)(6 * vi + rr) * np + 6 * vj + c] = 0.0 + h;
            Hl[(size_t)(6 * vj + c) * np + 6 * vi + rr] = 0.0 + h;
        }
        __syncthreads();
        if (want_maxdiag && w.n_free > 0) {  // computeLambdaInit over the pose blocks WITH the LiDAR term (the reduction's maximum is the visual part's)
            const double* Hpp = global_ptr(load_uniform(&w.sl->pb.Hpp));
            double m = 0;
            for (int j = lane; j < np; j += kLmThreads) {
                const int a = j % 6, dpos = a * 6 - a * (a - 1) / 2;  // diagonal of the packed upper triangle: 0, 6, 11, 15, 18, 20
                m = fmax(m, fabs(Hpp[27 * (size_t)(j / 6) + dpos] + Hl[(size_t)j * np + j]));
            }
            for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o));
            if ((lane & 63) == 0) s_max[lane >> 6] = m;
            __syncthreads();
            m = s_max[0];
            for (int q = 1; q < kLmThreads / 64; ++q) m = fmax(m, s_max[q]);
            max_pose_diag = m;
        }
    }
    st.currentChi = currentChi;
    st.tempChi = currentChi;
    st.iniChi = currentChi;
    if (st.it == 0) {
        st.initial_chi2 = currentChi;
        const double li = load_uniform(&w.sl->lambda_init);
        st.lambda = li > 0 ? li : 1e-5 * fmax(sc[1], max_pose_diag);
        st.ni = 2;
        st.n_bad = 0;
    }
    (void)want_maxdiag;
    st.rho = 0;
    st.qmax = 0;
    st.solve_ok = 1;
    st.status = kLmTrial;
    lm_store(st, w.lm, nullptr, lane, s_stage);
}

// After the trial of a window in kLmTrial: the gain ratio and what follows from it.
__global__ __launch_bounds__(kLmThreads) void k_ba_lm_decide_b(const BaPhase ph) {
    const LmWindow w = lm_window(ph, blockIdx.x);
    if (!w.lm || load_uniform(&w.lm->status) != kLmTrial) return;
    __shared__ double s_x[kLmMaxNp], s_bp[kLmMaxNp];
    __shared__ unsigned long long s_stage[sizeof(BaLmState) / 8];
    const int lane = threadIdx.x, np = w.np;
    BaLmState st = *w.lm;
    const double* sc = global_ptr(load_uniform(&w.sl->chi_out));
    {   // the step and b_p (+ the LiDAR gradient): what the host read from its pinned buffers
        const double* x = global_ptr(load_uniform(&w.sl->x_dev));
        const double* bs = global_ptr(load_uniform(&w.sl->bs_out));
        const double* bl = global_ptr(load_uniform(&w.sl->bl_lidar));
        for (int j = lane; j < np; j += kLmThreads) {
            s_x[j] = x[j];
            s_bp[j] = w.has_lidar ? bs[np + j] + bl[j] : bs[np + j];
        }
    }
    __syncthreads();
    const int32_t* stop_host = global_ptr(load_uniform(&w.sl->stop_host));
    const bool stopped = stop_host && __hip_atomic_load(stop_host, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0;
    const bool ok2 = np == 0 || st.solve_ok != 0;
    double scale = 0;
    for (int j = 0; j < np; ++j) scale += s_x[j] * (st.lambda * s_x[j] + s_bp[j]);  // pose part of computeScale(), j ascending
    if (ok2) {
        st.tempChi = sc[4];
        scale += sc[3];
        if (w.has_lidar) {
            const int n_planes = load_uniform(&w.sl->balm.n_planes);
            const double* out = global_ptr(load_uniform(&w.sl->balm.out));
            const double information = load_uniform(&w.sl->lidar_information);
            const double r = n_planes ? out[0] : 0.0;
            st.lidar_error = r;
            st.r1 = st.r2;
            st.r2 = r;
            st.is_calc_hess = !(st.r1 - st.r2 < 0) ? 1 : 0;
            st.tempChi = st.lidar_error * information * st.lidar_error + st.tempChi;
        }
    } else {
        st.tempChi = DBL_MAX;
    }
    st.rho = st.currentChi - st.tempChi;
    scale += 1e-3;
    st.rho /= scale;
    if (st.rho > 0 && isfinite(st.tempChi)) {
        st.lambda = lm_lambda_accepted(st.lambda, st.rho);
        st.ni = 2;
        st.currentChi = st.tempChi;
        st.parity ^= 1;
    } else {
        st.lambda *= st.ni;
        st.ni *= 2;
    }
    st.qmax++;
    st.trials_total++;
    st.rounds++;
    st.solve_ok = 1;
    if (st.rho < 0 && st.qmax < 10 && !stopped) {
        st.status = kLmTrial;
    } else {
        ++st.done;
        ++st.it;
        if (st.qmax == 10 || st.rho == 0) st.ok = 0;
        else {
            if ((st.iniChi - st.currentChi) * 1e3 < st.iniChi) st.n_bad++; else st.n_bad = 0;
            if (st.n_bad >= 3) st.ok = 0;
        }
        st.status = st.ok && st.it < load_uniform(&w.sl->iterations) && !stopped ? kLmIterate : kLmDone;
    }
    lm_store(st, w.lm, global_ptr(load_uniform(&w.sl->lm_host)), lane, s_stage);
}

void ba_batch_launch_lm_begin(const BaPhase& ph, int n_active, hipStream_t st) {
    if (n_active) TC2LI_LAUNCH(k_ba_lm_begin_b, dim3(n_active), dim3(kLmBeginThreads), 0, st, ph);
}
void ba_batch_launch_lm_decide(const BaPhase& ph, int n_active, hipStream_t st) {
    if (n_active) TC2LI_LAUNCH(k_ba_lm_decide_b, dim3(n_active), dim3(kLmThreads), 0, st, ph);
}

}  // namespace tc2li
