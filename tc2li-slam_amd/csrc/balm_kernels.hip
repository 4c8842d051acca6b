// LiDAR plane term of the local bundle adjustment on gfx950: the cost sum_planes N * lambda_min(cov of the plane's points
// over the window) and its first / second derivatives with respect to the window's LiDAR poses.
//   VOX_HESS::evaluate_only_residual   SF/include/bavoxel.h:276-315   -> k_balm_residual
//   VOX_HESS::acc_evaluate2            SF/include/bavoxel.h:80-196    -> k_balm_hessian
//   BALM2::divide_thread               SF/include/bavoxel.h:778-817   -> chunks of planes + k_balm_combine (fixed order)
//   LidarCovisRes::UpdatePose          SF/src/LidarRes.cc:221-235     -> window_poses (inside every kernel)
// Layout: one workgroup owns a contiguous chunk of planes; every thread owns a few entries of the upper block triangle
// of the (6W)^2 Hessian in registers (entry = (block pair, row, column)), so no atomics and a fixed summation order.
#include <hip/hip_runtime.h>

#include "launch.hpp"
#pragma clang fp contract(off)
#include <stdint.h>
#include <stdlib.h>

#include <algorithm>

#include "ba_device.hpp"
#include "balm_device.hpp"
#include "balm_residual_device.hpp"

namespace tc2li {

// 256 threads per workgroup: the per-slot algebra needs well over 128 registers (1024-thread workgroups cap a thread at
// 128 and spilled 512 B per lane to scratch: 67 MB of scratch writes per launch in the first profile).
constexpr int kHessThreads = 256;
// Hessian entries owned per thread: windows of <= 7 / <= 20 keyframes.  The small-window form is ONE wavefront per plane: the kernel
// needs ~200 VGPRs, and a 256-thread workgroup of it has to find that room on all four SIMDs of a CU at once -- on a GPU shared with
// the front end its launches waited (72 us on average in the loop against 35 us alone); a single wavefront fits wherever one SIMD has room.
constexpr int kHessThreadsSmall = 64;
constexpr int kItemsSmall = 16, kItemsLarge = 30;
static_assert(7 * 8 / 2 * 36 <= kHessThreadsSmall * kItemsSmall, "small window does not fit");
static_assert(kMaxLidarWindow * (kMaxLidarWindow + 1) / 2 * 36 <= kHessThreads * kItemsLarge, "window too large for the item ownership");

__global__ __launch_bounds__(256) void k_balm_residual_total(BalmDev b, const Se3* __restrict__ poses) { d_balm_residual_total(b, poses); }
// many planes: one thread per plane over the whole grid, then the same fixed-order sum
__global__ __launch_bounds__(256) void k_balm_residual_planes(BalmDev b, const Se3* __restrict__ poses) {
    __shared__ LidarPose s_twl[kMaxLidarWindow];
    window_poses(b, poses, s_twl);
    const int a = blockIdx.x * 256 + threadIdx.x;
    if (a < b.n_planes) b.plane_res[a] = plane_residual(b, s_twl, a);
}
__global__ __launch_bounds__(256) void k_balm_sum(BalmDev b) {
    __shared__ double s[256];
    sum_fixed_256(b.plane_res, b.n_planes, s, b.out);
}

// PB planes at a time: lane (p, i) = (tid / SL, tid % SL) does the per-slot algebra of plane p, slot i -- the long dependent chain of
// the kernel, which one plane alone runs on W of the workgroup's lanes -- then every lane adds the PB planes' terms to the Hessian
// entries it owns, in plane order (the sums are those of a plane-after-plane loop, bit for bit).  WC: slots the LDS tables hold.
// the workgroup's LDS tables as one record: the callers place it (statically, or in dynamic LDS -- k_balm_hessian_lean3_b / lean4_b)
template <int PB, int WC>
struct HessLds {
    LidarPose s_twl[kMaxLidarWindow];
    double s_A[PB][WC][18], s_MB[PB][WC][18];  // Auk (3 x 6) and umumT * Auk
    double s_w[PB][WC][3], s_E[PB][WC][9], s_k1[PB][WC], s_k2[PB][WC], s_n[PB][WC], s_cj[PB][WC][6];
    double s_uk[PB][3], s_ukuk[PB][9], s_umum[PB][9], s_vbar[PB][3], s_NN[PB], s_l0[PB], s_coe[PB];
    // the off-diagonal blocks' factors -2 / NN / NN, -2 n_j / NN / NN and -2 n_i n_j / NN / NN: formed once per plane, slot and slot pair
    // (two f64 divisions each) instead of once per Hessian entry
    double s_f0[PB], s_fn[PB][WC], s_fnn[PB][WC * (WC + 1) / 2];
    uint8_t s_pi[kMaxLidarWindow * (kMaxLidarWindow + 1) / 2], s_pj[kMaxLidarWindow * (kMaxLidarWindow + 1) / 2];
};
template <int kItemsPerThread, int NT, int PB, int SL, int WC, bool kLean = false, bool kFused = false>
__device__ __forceinline__ void d_balm_hessian(const BalmDev& b, const Se3* __restrict__ poses, const int bx, HessLds<PB, WC>& L) {
    static_assert(PB * SL <= NT && WC <= SL, "lane layout");
    auto& s_twl = L.s_twl; auto& s_A = L.s_A; auto& s_MB = L.s_MB; auto& s_w = L.s_w; auto& s_E = L.s_E; auto& s_k1 = L.s_k1; auto& s_k2 = L.s_k2;
    auto& s_n = L.s_n; auto& s_cj = L.s_cj; auto& s_uk = L.s_uk; auto& s_ukuk = L.s_ukuk; auto& s_umum = L.s_umum; auto& s_vbar = L.s_vbar;
    auto& s_NN = L.s_NN; auto& s_l0 = L.s_l0; auto& s_coe = L.s_coe; auto& s_f0 = L.s_f0; auto& s_fn = L.s_fn; auto& s_fnn = L.s_fnn;
    auto& s_pi = L.s_pi; auto& s_pj = L.s_pj;
    const int tid = threadIdx.x, W = b.W, n_items = W * (W + 1) / 2 * 36;
    window_poses(b, poses, s_twl);
    if (bx == 0 && tid < W) {  // for the change of variables on the host
        if (kFused) {
            const double* src = reinterpret_cast<const double*>(&s_twl[tid]);
            double* dst = reinterpret_cast<double*>(&b.twl[tid]);
            for (int q = 0; q < (int)(sizeof(LidarPose) / sizeof(double)); ++q) __hip_atomic_store(dst + q, src[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            b.twl[tid] = s_twl[tid];
        }
    }
    if (tid == 0) {
        int p = 0;
        for (int i = 0; i < W; ++i)
            for (int j = i; j < W; ++j) { s_pi[p] = (uint8_t)i; s_pj[p] = (uint8_t)j; ++p; }
    }
    double acc[kItemsPerThread], jac[6] = {0, 0, 0, 0, 0, 0}, res = 0;
#pragma unroll
    for (int k = 0; k < kItemsPerThread; ++k) acc[k] = 0;
    const int a0 = bx * b.planes_per_chunk, a1 = min(a0 + b.planes_per_chunk, b.n_planes);
    const int lp = tid / SL, li = tid % SL;  // this lane's plane of the batch and slot
    for (int ab = a0; ab < a1; ab += PB) {
        const int a = ab + lp;
        const bool slot_lane = lp < PB && li < W && a < a1;
        PlaneCluster mine;
        mine.n = 0;
        if (slot_lane) mine = b.clusters[(size_t)a * W + li];
        if (lp < PB && li < W) s_n[lp][li] = mine.n;  // zero for the planes past the chunk's end: their terms drop out below
        if (lp < PB && li == 0) {
            s_coe[lp] = 0;
            if (a < a1) {
                // merged cluster -> covariance -> eigen decomposition: taken from the residual pass at these poses (plane_residual; the
                // host launches it right before, see BalmTerm::enqueue_linearization), there is no second copy of that arithmetic
                const double* ei = b.eig + (size_t)kBalmEig * a;
                const double NN = ei[0];
                const double vbar[3] = {ei[1], ei[2], ei[3]}, lambda[3] = {ei[4], ei[5], ei[6]};
                const double* U = ei + 7;
                const double u0[3] = {U[0], U[3], U[6]};
                for (int k = 0; k < 3; ++k) { s_uk[lp][k] = u0[k]; s_vbar[lp][k] = vbar[k]; }
                for (int r = 0; r < 3; ++r)
                    for (int c = 0; c < 3; ++c) {
                        s_ukuk[lp][3 * r + c] = u0[r] * u0[c];
                        double m = 0;
                        for (int e = 1; e < 3; ++e) m = m + (2.0 / (lambda[0] - lambda[e])) * (U[3 * r + e] * U[3 * c + e]);
                        s_umum[lp][3 * r + c] = m;
                    }
                s_NN[lp] = NN;
                s_l0[lp] = lambda[0];
                s_coe[lp] = b.coe[a];
                s_f0[lp] = -2.0 / NN / NN;
            }
        }
        __syncthreads();
        if (slot_lane) s_fn[lp][li] = -2.0 * mine.n / s_NN[lp] / s_NN[lp];
        for (int t = tid; t < PB * (W * (W + 1) / 2); t += NT) {
            const int p = t / (W * (W + 1) / 2), pair = t - p * (W * (W + 1) / 2);
            if (ab + p < a1) s_fnn[p][pair] = -2.0 * s_n[p][s_pi[pair]] * s_n[p][s_pj[pair]] / s_NN[p] / s_NN[p];
        }
        if (slot_lane && mine.n != 0) {
            const double NN = s_NN[lp], coe = s_coe[lp];
            const LidarPose T = s_twl[li];
            const double ni = mine.n;
            double Pi[9], uk[3] = {s_uk[lp][0], s_uk[lp][1], s_uk[lp][2]};
            sym_unpack(mine.P, Pi);
            const double* vi = mine.v;
            double vihat[9], RiTuk[3], RiTukhat[9], PiRiTuk[3], w[3], tv[3];
            m3_hat(vi, vihat);
            m3_tvec(T.R, uk, RiTuk);
            m3_hat(RiTuk, RiTukhat);
            m3_vec(Pi, RiTuk, PiRiTuk);
            m3_vec(vihat, RiTuk, w);
            for (int k = 0; k < 3; ++k) tv[k] = T.p[k] - s_vbar[lp][k];
            const double ukt = uk[0] * tv[0] + uk[1] * tv[1] + uk[2] * tv[2];
            double combo1[9], h[9], combo2[3], Rv[3];
            m3_hat(PiRiTuk, h);
            for (int k = 0; k < 9; ++k) combo1[k] = h[k] + ukt * vihat[k];
            m3_vec(T.R, vi, Rv);
            for (int k = 0; k < 3; ++k) combo2[k] = Rv[k] + ni * tv[k];
            double RP[9], lhs[9], left[9], Rc[9];
            m3_mul(T.R, Pi, RP);
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) RP[3 * r + c] = RP[3 * r + c] + tv[r] * vi[c];
            m3_mul(RP, RiTukhat, lhs);
            m3_mul(T.R, combo1, Rc);
            for (int k = 0; k < 9; ++k) left[k] = lhs[k] - Rc[k];
            const double c2u = combo2[0] * uk[0] + combo2[1] * uk[1] + combo2[2] * uk[2];
            double A[18];
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) {
                    A[6 * r + c] = left[3 * r + c] / NN;
                    A[6 * r + 3 + c] = (combo2[r] * uk[c] + (r == c ? c2u : 0.0)) / NN;
                }
            double jjt[6];
            for (int c = 0; c < 6; ++c) {
                jjt[c] = A[c] * uk[0] + A[6 + c] * uk[1] + A[12 + c] * uk[2];
                s_cj[lp][li][c] = coe * jjt[c];
            }
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 6; ++c) {
                    s_A[lp][li][6 * r + c] = A[6 * r + c];
                    s_MB[lp][li][6 * r + c] = s_umum[lp][3 * r] * A[c] + s_umum[lp][3 * r + 1] * A[6 + c] + s_umum[lp][3 * r + 2] * A[12 + c];
                }
            // extra terms of the diagonal block: rotation-rotation part
            double d1[9], d2[9], RhP[9], jh[9];
            m3_mul(RiTukhat, Pi, RhP);
            for (int k = 0; k < 9; ++k) d1[k] = combo1[k] - RhP[k];
            m3_mul(d1, RiTukhat, d2);
            m3_hat(jjt, jh);
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c)
                    s_E[lp][li][3 * r + c] = (2.0 / NN) * d2[3 * r + c] + (-2.0 / NN / NN) * (w[r] * w[c]) + (-0.5) * jh[3 * r + c];
            for (int k = 0; k < 3; ++k) s_w[lp][li][k] = w[k];
            s_k1[lp][li] = 2.0 / NN * (1.0 - ni / NN);
            s_k2[lp][li] = 2.0 / NN * (ni - ni * ni / NN);
        }
        __syncthreads();
        const int np_ = min(PB, a1 - ab);
        if (tid == 0)
            for (int p = 0; p < np_; ++p) res += s_coe[p] * s_l0[p];
        if (tid < W)
            for (int p = 0; p < np_; ++p)
                if (s_n[p][tid] != 0)
                    for (int c = 0; c < 6; ++c) jac[c] += s_cj[p][tid][c];
#pragma unroll
        for (int k = 0; k < kItemsPerThread; ++k) {
            int item = k * NT + tid;
            // (lean form: the item number is made opaque once per pass, so that the sixteen items' indices and LDS addresses are formed here,
            // where they are used, instead of being carried in registers across the plane loop -- they were what the kernel's registers went to)
            if (kLean) asm volatile("" : "+v"(item));
            if (item >= n_items) continue;
            const int pair = item / 36, rc = item % 36, r = rc / 6, c = rc % 6;
            const int i = s_pi[pair], j = s_pj[pair];
            for (int p = 0; p < np_; ++p) {
                const double ni = s_n[p][i], nj = s_n[p][j];
                if (ni == 0 || nj == 0) continue;
                double val = s_A[p][i][r] * s_MB[p][j][c] + s_A[p][i][6 + r] * s_MB[p][j][6 + c] + s_A[p][i][12 + r] * s_MB[p][j][12 + c];
                if (i == j) {
                    if (r < 3 && c < 3) val += s_E[p][i][3 * r + c];
                    else if (r < 3) val += s_k1[p][i] * (s_w[p][i][r] * s_uk[p][c - 3]);
                    else if (c < 3) val += s_k1[p][i] * (s_w[p][i][c] * s_uk[p][r - 3]);
                    else val += s_k2[p][i] * s_ukuk[p][3 * (r - 3) + (c - 3)];
                } else {
                    if (r < 3 && c < 3) val += s_f0[p] * (s_w[p][i][r] * s_w[p][j][c]);
                    else if (r < 3) val += s_fn[p][j] * (s_w[p][i][r] * s_uk[p][c - 3]);
                    else if (c < 3) val += s_fn[p][i] * (s_uk[p][r - 3] * s_w[p][j][c]);
                    else val += s_fnn[p][pair] * s_ukuk[p][3 * (r - 3) + (c - 3)];
                }
                acc[k] += s_coe[p] * val;
            }
        }
        __syncthreads();
    }
    // (kFused: the window's last chunk adds the partials in this launch -- they leave at device scope, through the XCD's L2)
    double* part = b.part + (size_t)bx * (n_items + 6 * W + 1);
    auto put = [](double* p, double v) { if (kFused) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *p = v; };
#pragma unroll
    for (int k = 0; k < kItemsPerThread; ++k) {
        const int item = k * NT + tid;
        if (item < n_items) put(part + item, acc[k]);
    }
    if (tid < W)
        for (int c = 0; c < 6; ++c) put(part + n_items + 6 * tid + c, jac[c]);
    if (tid == 0) put(part + n_items + 6 * W, res);
}
// small windows: one wavefront, 8 planes x 8 slots; large ones: 256 threads, 4 planes x 32 slots
constexpr int kHessPlanesSmall = 8, kHessPlanesLarge = 4;
__global__ __launch_bounds__(kHessThreadsSmall) void k_balm_hessian_small(BalmDev b, const Se3* __restrict__ poses) {
    __shared__ HessLds<kHessPlanesSmall, 8> lds;
    d_balm_hessian<kItemsSmall, kHessThreadsSmall, kHessPlanesSmall, 8, 8>(b, poses, blockIdx.x, lds);
}
__global__ __launch_bounds__(kHessThreads) void k_balm_hessian_large(BalmDev b, const Se3* __restrict__ poses) {
    __shared__ HessLds<kHessPlanesLarge, kMaxLidarWindow> lds;
    d_balm_hessian<kItemsLarge, kHessThreads, kHessPlanesLarge, 32, kMaxLidarWindow>(b, poses, blockIdx.x, lds);
}

// chunk partials -> out (JacT, full Hessian with the lower block triangle mirrored, the Hessian pass's residual): one thread per
// output value adds the chunks in chunk order, eight loads in flight (neighbouring threads read neighbouring values of a chunk).
// (kFused: called by the last chunk of the Hessian launch itself -- the other chunks' partials and chunk 0's poses are read at device scope)
template <bool kFused = false>
__device__ __forceinline__ void d_balm_combine_value(const BalmDev& b, const int idx) {
    const int W = b.W, n = 6 * W, n_items = W * (W + 1) / 2 * 36, stride = n_items + n + 1;
    auto get = [](const double* p) { return kFused ? __hip_atomic_load(const_cast<double*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p; };
    if (idx < 12 * W) b.out[2 + n + n * n + idx] = get(reinterpret_cast<const double*>(b.twl) + idx);  // the poses the derivatives refer to
    if (idx >= stride) return;
    double s = 0;
    const double* p = b.part + idx;
    int k = 0;
    for (; k + 8 <= b.n_chunks; k += 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = get(p + (size_t)(k + u) * stride);
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; k < b.n_chunks; ++k) s += get(p + (size_t)k * stride);
    if (idx == n_items + n) { b.out[1 + n + n * n] = s; return; }  // out[0] belongs to the residual-only kernels
    if (idx >= n_items) { b.out[1 + (idx - n_items)] = s; return; }
    const int pair = idx / 36, rc = idx % 36, r = rc / 6, c = rc % 6;
    int i = 0, rem = pair;  // pair -> (i, j >= i)
    while (rem >= W - i) { rem -= W - i; ++i; }
    const int j = i + rem;
    double* H = b.out + 1 + n;
    H[(size_t)(6 * i + r) * n + 6 * j + c] = s;
    if (i != j) H[(size_t)(6 * j + c) * n + 6 * i + r] = s;
}
__device__ __forceinline__ void d_balm_combine(const BalmDev& b, const int bx) { d_balm_combine_value(b, bx * 256 + (int)threadIdx.x); }
__global__ __launch_bounds__(256) void k_balm_combine(BalmDev b) { d_balm_combine(b, blockIdx.x); }

void balm_launch_residual(const BalmDev& b, const Se3* poses, hipStream_t st) {
    if (b.n_planes <= 2048) {
        TC2LI_LAUNCH(k_balm_residual_total, dim3(1), dim3(256), 0, st, b, poses);
    } else {
        TC2LI_LAUNCH(k_balm_residual_planes, dim3((b.n_planes + 255) / 256), dim3(256), 0, st, b, poses);
        TC2LI_LAUNCH(k_balm_sum, dim3(1), dim3(256), 0, st, b);
    }
}

void balm_launch_hessian(const BalmDev& b, const Se3* poses, hipStream_t st) {
    if (b.W <= 7) TC2LI_LAUNCH(k_balm_hessian_small, dim3(b.n_chunks), dim3(kHessThreadsSmall), 0, st, b, poses);
    else TC2LI_LAUNCH(k_balm_hessian_large, dim3(b.n_chunks), dim3(kHessThreads), 0, st, b, poses);
    TC2LI_LAUNCH(k_balm_combine, dim3((std::max(balm_part_stride(b.W), 12 * b.W) + 255) / 256), dim3(256), 0, st, b);
}

// ---- lock-step batch (ba_device.hpp): the window's BalmDev and pose arrays come from its slot ----
__global__ __launch_bounds__(256) void k_balm_residual_total_b(const BaPhase ph, int trial) {
    // (a trial phase's residual of a window with pb.trial_fused ran inside k_ba_trial_fused*_b)
    if (trial && load_uniform(&(ph.table + ba_phase_window(ph, blockIdx.x))->pb.trial_fused)) return;
    const BalmSlotView v = balm_slot_view(ph, blockIdx.x, trial != 0);
    if (!v.active) return;
    d_balm_residual_total(v.b, v.poses);
}
__global__ __launch_bounds__(kHessThreadsSmall) void k_balm_hessian_b(const BaPhase ph) {
    const BalmSlotView v = balm_slot_view(ph, blockIdx.y, false);
    if (!v.active || (int)blockIdx.x >= v.b.n_chunks) return;
    __shared__ HessLds<kHessPlanesSmall, 8> lds;
    d_balm_hessian<kItemsSmall, kHessThreadsSmall, kHessPlanesSmall, 8, 8>(v.b, v.poses, blockIdx.x, lds);
}
// The same body with its tables in DYNAMIC LDS and the wavefront held to TC2LI_HESS_WAVES per SIMD.  With the 36 KB declared statically the
// compiler counts four workgroups per CU -- one wavefront per SIMD -- and lets the body take 325 registers (256 + 69 accumulation
// registers): alone that costs nothing, but beside the other stages' kernels such a wavefront starts only on a SIMD that has two thirds of
// its register file free.
#define TC2LI_HESS_LEAN_KERNEL(name, waves)                                                                                                  \
    __global__ __launch_bounds__(kHessThreadsSmall) __attribute__((amdgpu_waves_per_eu(waves, waves))) void name(const BaPhase ph, int fuse) { \
        extern __shared__ double s_hess_dyn[];                                                                                               \
        const BalmSlotView v = balm_slot_view(ph, blockIdx.y, false);                                                                        \
        if (!v.active || (int)blockIdx.x >= v.b.n_chunks) return;                                                                            \
        auto& L = *reinterpret_cast<HessLds<kHessPlanesSmall, 8>*>(s_hess_dyn);                                                              \
        /* one instantiation of the body: the partials always leave at device scope (harmless before the separate combine launch) */         \
        d_balm_hessian<kItemsSmall, kHessThreadsSmall, kHessPlanesSmall, 8, 8, true, true>(v.b, v.poses, blockIdx.x, L);                      \
        if (!fuse) return;                                                                                                                   \
        /* round 5: the window's last chunk adds the chunks' partials (k_balm_combine_b's sums, value by value in chunk order: same bits) */  \
        if (!ba_last_of(global_ptr(load_uniform(&(ph.table + ba_phase_window(ph, blockIdx.y))->pb.ticket)) + 2, v.b.n_chunks)) return;        \
        const int n_values = max(balm_part_stride_dev(v.b.W), 12 * v.b.W);                                                                   \
        for (int idx = threadIdx.x; idx < n_values; idx += kHessThreadsSmall) d_balm_combine_value<true>(v.b, idx);                          \
    }
TC2LI_HESS_LEAN_KERNEL(k_balm_hessian_lean3_b, 3)  // 168 registers, 13 values in scratch
TC2LI_HESS_LEAN_KERNEL(k_balm_hessian_lean4_b, 4)  // 128 registers, 84 values in scratch
#undef TC2LI_HESS_LEAN_KERNEL
__global__ __launch_bounds__(256) void k_balm_combine_b(const BaPhase ph) {
    const BalmSlotView v = balm_slot_view(ph, blockIdx.y, false);
    if (!v.active || (int)blockIdx.x >= (max(balm_part_stride_dev(v.b.W), 12 * v.b.W) + 255) / 256) return;
    d_balm_combine(v.b, blockIdx.x);
}
void balm_batch_launch_residual(const BaPhase& ph, int n, bool trial, hipStream_t st) {
    if (n) TC2LI_LAUNCH(k_balm_residual_total_b, dim3(n), dim3(256), 0, st, ph, trial ? 1 : 0);
}
void balm_batch_launch_hessian(const BaPhase& ph, int n, const BaBatchExtent& x, hipStream_t st) {
    if (!n) return;
    // TC2LI_BALM_HESS_LEAN = 0: the 325-register form; 3 (default) / 4: the lean forms (read per call: A/B switch of the measurements)
    const char* lean_env = getenv("TC2LI_BALM_HESS_LEAN");
    const int lean = lean_env ? atoi(lean_env) : 3;
    const size_t lds = sizeof(HessLds<kHessPlanesSmall, 8>);
    const int fuse = x.fuse_linearize && lean != 0;  // the chunks' sums by the window's last chunk (round 5; ba_last_of, ticket word 2)
    if (lean == 4) TC2LI_LAUNCH(k_balm_hessian_lean4_b, dim3(x.max_chunks, n), dim3(kHessThreadsSmall), lds, st, ph, fuse);
    else if (lean == 3) TC2LI_LAUNCH(k_balm_hessian_lean3_b, dim3(x.max_chunks, n), dim3(kHessThreadsSmall), lds, st, ph, fuse);
    else TC2LI_LAUNCH(k_balm_hessian_b, dim3(x.max_chunks, n), dim3(kHessThreadsSmall), 0, st, ph);
    if (fuse) return;
    TC2LI_LAUNCH(k_balm_combine_b, dim3((std::max(balm_part_stride(x.max_W), 12 * x.max_W) + 255) / 256, n), dim3(256), 0, st, ph);
}

}  // namespace tc2li
