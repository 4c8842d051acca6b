// Shared between the host orchestration and the kernels of the LiDAR plane term (balm_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "balm_math.hpp"

namespace tc2li {

constexpr int kBalmEig = 16;
constexpr int kMaxLidarWindow = 20;  // LidarCovisRes::win_size_ default (SF/include/LidarRes.h); the local BA uses <= 6

// The plane list of one window resident on the device (VOX_HESS: plvec_voxels / coeffs, SF/include/bavoxel.h:48-78).
struct BalmDev {
    int32_t W, n_planes, n_chunks, planes_per_chunk;
    int32_t imu_pose_bytes, pad_;  // 0: the vertex array holds Se3; otherwise ImuPose records of this size (Rcw[9], tcw[3] first)
    const PlaneCluster* clusters;  // [n_planes][W]
    const double* coe;             // [n_planes]
    const int32_t* pose_index;     // [W] rows of the pose array
    SE3f Tcl;
    LidarPose* twl;                // [W]
    double* plane_res;             // [n_planes]
    double* eig;                   // [n_planes][kBalmEig]: NN, vbar[3], lambda[3], U[9] of the merged plane at the poses of the last residual pass
    double* part;                  // [n_chunks][balm_part_stride(W)]
    double* out;                   // [2 + 6W + (6W)^2 + 12W]: residual (residual kernels), JacT, Hessian (row-major), residual (Hessian
                                   // kernels), the LiDAR poses the derivatives were taken at
};

inline int balm_items(int W) { return W * (W + 1) / 2 * 36; }
inline int balm_part_stride(int W) { return balm_items(W) + 6 * W + 1; }
__device__ inline int balm_part_stride_dev(int W) { return W * (W + 1) / 2 * 36 + 6 * W + 1; }
inline int balm_out_size(int W) { return 2 + 6 * W + 36 * W * W + 12 * W; }

// out[0] = sum over planes of coe * lambda_min at the window poses derived from the vertex estimates `poses`
// (LidarCovisRes::UpdatePose + VOX_HESS::evaluate_only_residual)
void balm_launch_residual(const BalmDev& b, const Se3* poses, hipStream_t st);
// out = JacT, Hessian with respect to the LiDAR poses (BALM2::divide_thread / VOX_HESS::acc_evaluate2) and those poses
void balm_launch_hessian(const BalmDev& b, const Se3* poses, hipStream_t st);

}  // namespace tc2li
