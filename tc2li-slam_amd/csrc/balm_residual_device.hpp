// Device functions of the LiDAR plane term that two translation units need: the planes' residual at a set of poses (balm_kernels.hip:
// k_balm_residual_total*; ba_kernels.hip: the LiDAR role of the fused trial launch, round 5) and the lock-step batch's view of a window's
// BalmDev.  Reference: VOX_HESS::evaluate_only_residual, SF/include/bavoxel.h:276-315; LidarCovisRes::UpdatePose, SF/src/LidarRes.cc:221-235.
#pragma once
#include <hip/hip_runtime.h>

#include "ba_device.hpp"
#include "balm_device.hpp"

namespace tc2li {

// LiDAR poses of the window slots from the vertex estimates (LidarCovisRes::UpdatePose), into LDS of the calling workgroup
__device__ __forceinline__ void window_poses(const BalmDev& b, const Se3* __restrict__ poses, LidarPose* s_twl) {
    if ((int)threadIdx.x < b.W) {
        const int k = b.pose_index[threadIdx.x];
        if (b.imu_pose_bytes) {  // EdgeLidar on VertexPose (LocalLVIBA): Rcw / tcw of the ImuCamPose
            const double* rt = reinterpret_cast<const double*>(reinterpret_cast<const char*>(poses) + (size_t)k * b.imu_pose_bytes);
            s_twl[threadIdx.x] = lidar_pose_from(se3f_from_rt(rt, rt + 9), b.Tcl);
        } else {
            s_twl[threadIdx.x] = lidar_pose_from(se3f_from_vertex(poses[k]), b.Tcl);
        }
    }
    __syncthreads();
}

// merged window cluster of one plane -> covariance -> eigen decomposition -> its term of the residual
__device__ __forceinline__ double plane_residual(const BalmDev& b, const LidarPose* twl, int a) {
    double P[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, v[3] = {0, 0, 0}, n = 0;
    for (int i = 0; i < b.W; ++i) {
        const PlaneCluster s = b.clusters[(size_t)a * b.W + i];
        if (s.n == 0) continue;
        ClusterW t;
        cluster_transform(s, twl[i], t);
        for (int k = 0; k < 9; ++k) P[k] += t.P[k];
        for (int k = 0; k < 3; ++k) v[k] += t.v[k];
        n += t.n;
    }
    const double inv = 1.0 / n;
    double vb[3], C[9], lambda[3], U[9];
    for (int k = 0; k < 3; ++k) vb[k] = inv * v[k];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) C[3 * r + c] = inv * P[3 * r + c] - vb[r] * vb[c];
    eig_sym3(C, lambda, U);
    // the Hessian pass that follows at the same poses starts from this decomposition instead of repeating it in one lane
    double* eo = b.eig + (size_t)kBalmEig * a;
    eo[0] = n;
    for (int k = 0; k < 3; ++k) { eo[1 + k] = vb[k]; eo[4 + k] = lambda[k]; }
    for (int k = 0; k < 9; ++k) eo[7 + k] = U[k];
    return b.coe[a] * lambda[0];
}

// out[0] = in[0] + in[1] + ... in a fixed order (256 threads: strided partial sums, then a tree)
__device__ __forceinline__ void sum_fixed_256(const double* __restrict__ in, int n, double* s, double* __restrict__ out) {
    double a = 0;
    for (int k = threadIdx.x; k < n; k += 256) a += in[k];
    s[threadIdx.x] = a;
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {
        if ((int)threadIdx.x < st) s[threadIdx.x] += s[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = s[0];
}

// VOX_HESS::evaluate_only_residual in one launch when the planes fit one workgroup's loop: poses, per-plane terms, sum
__device__ __forceinline__ void d_balm_residual_total(const BalmDev& b, const Se3* __restrict__ poses) {
    __shared__ LidarPose s_twl[kMaxLidarWindow];
    __shared__ double s[256];
    window_poses(b, poses, s_twl);
    for (int a = threadIdx.x; a < b.n_planes; a += 256) b.plane_res[a] = plane_residual(b, s_twl, a);
    __syncthreads();
    sum_fixed_256(b.plane_res, b.n_planes, s, b.out);
}
// the poses of the window at position `pos` of the phase: the accepted estimate or the trial one (the phase's parity bit says which of
// the slot's two buffers holds the accepted estimate)
struct BalmSlotView { BalmDev b; const Se3* poses; bool active; };
__device__ __forceinline__ BalmSlotView balm_slot_view(const BaPhase& ph, int pos, bool trial) {
    __builtin_amdgcn_s_setprio(3);
    const BaBatchSlot* const sl = ph.table + ba_phase_window(ph, pos);
    const BaLmView lmv = ba_lm_view(ph, sl, ba_phase_flags(ph, pos), 0.0);  // device-side LM: the parity from the window's state
    const bool second = trial != ((lmv.flags & kBaAcceptedInTrial) != 0);
    BalmSlotView v;
    v.active = lmv.active;
    v.b = load_uniform(&sl->balm);  // (scalar loads: the record lives in SGPRs, not in every lane's registers)
    if (load_uniform(&sl->pb.inertial)) v.poses = reinterpret_cast<const Se3*>(second ? load_uniform(&sl->pb.iposes_trial) : load_uniform(&sl->pb.iposes));
    else v.poses = second ? load_uniform(&sl->pb.poses_trial) : load_uniform(&sl->pb.poses);
    // pointers out of a record are built from integers: global, not flat, accesses through them (ba_device.hpp: global_ptr)
    v.poses = global_ptr(v.poses);
    v.b.clusters = global_ptr(v.b.clusters); v.b.coe = global_ptr(v.b.coe); v.b.pose_index = global_ptr(v.b.pose_index); v.b.twl = global_ptr(v.b.twl);
    v.b.plane_res = global_ptr(v.b.plane_res); v.b.eig = global_ptr(v.b.eig); v.b.part = global_ptr(v.b.part); v.b.out = global_ptr(v.b.out);
    return v;
}

}  // namespace tc2li
