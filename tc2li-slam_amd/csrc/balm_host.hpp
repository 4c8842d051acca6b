// Host side of the LiDAR plane term of LocalLVBundleAdjustment (SF/src/OptimizerWithLidar.cc:226-260): plane extraction
// from the window's surface clouds and the EdgeLidarSE3 state that the Levenberg-Marquardt loops of ba_host.cpp / ba_lockstep.cpp / lvi_host.cpp drive.
#pragma once
#include <vector>

#include "../../include/tc2li_hip.h"
#include "balm_cut_device.hpp"
#include "balm_device.hpp"
#include "common.hpp"

namespace tc2li {

// Adaptive voxelisation of the window clouds (cut_voxel + OCTO_TREE_NODE::recut / tras_opt, SF/src/bavoxel.cc:42-91,
// SF/include/bavoxel.h:492-602,723-740): points are binned into 1 m voxels in the frame of the first keyframe's LiDAR;
// a voxel whose points are planar (lambda_min / lambda_mid below 1/36, 1/25 in deeper layers) becomes one plane, a
// non-planar one is split into octants up to two times.  Returns the per-keyframe clusters [n_planes][W] and coe.
void balm_build_planes(const LidarPose* twl, int W, const float* cloud_xyz, const int32_t* cloud_offsets,
                       std::vector<PlaneCluster>& clusters, std::vector<double>& coe);

// Jacobian / Hessian with respect to the LiDAR poses (R <- R Exp(dtheta), p <- p + dp) -> with respect to the left
// se3 increment of the camera poses Tcw that g2o::VertexSE3Expmap applies (LidarCovisRes::ComputeJandHSE3,
// SF/src/LidarRes.cc:136-186).  In place; H is (6W)^2 row-major.
void balm_to_camera_se3(const LidarPose* twl, int W, const SE3f& Tcl, double* JacT, double* H);

// The same for the body-frame increment of VertexPose (ImuCamPose::Update): LidarCovisRes::ComputeJandH, SF/src/LidarRes.cc:89-128.
void balm_to_body(const LidarPose* twl, int W, const SE3f& Tbl, double* JacT, double* H);

struct BalmTerm {
    int W = 0, n_planes = 0;
    double information = 1;
    SE3f Tcl{};
    // EdgeLidar on VertexPose (LocalLVIBA, SF/src/G2oTypesWithLidar.cc:33-75): the vertex array holds ImuPose records, the edge's
    // error is sqrt(r) and the derivatives go through ComputeJandH
    bool body = false;
    SE3f Tbl{};
    std::vector<int32_t> pose_index;
    // EdgeLidarSE3 state (SF/include/G2oTypesWithLidar.h:88-236)
    double error = 0, r1 = 1000, r2 = 1000;
    bool is_calc_hess = true;
    int hessian_evaluations = 0;
    std::vector<double> JacT, Hessian;  // camera-se3 forms of the last evaluation

    BalmDev dev{};
    DevBuf<PlaneCluster> d_clusters;
    std::vector<PlaneCluster> h_clusters;  // the planes of the last build (kept alive for the asynchronous upload)
    std::vector<double> h_coe;
    DevBuf<double> d_coe, d_plane_res, d_part, d_eig;
    const void* eig_at = nullptr;  // the pose array of the last residual pass: the Hessian pass reuses its eigen decompositions
    DevBuf<int32_t> d_pose_index;
    DevBuf<LidarPose> d_twl;
    PinnedBuf<double> h_out;
    PinnedBuf<LidarPose> h_twl;
    PinnedBuf<uint8_t> h_upload;  // clusters | coe | pose_index of the last build when the uploads are gathered (CopySink)
    // Plane extraction on the device (balm_cut_kernels.hip), for the windows of a lock-step batch: build / build_body given a task
    // stage the clouds, fill the task and leave the planes pending; the batch's owner queues launch_balm_cut over all its tasks and,
    // after its synchronisation, calls finish_cut -- the number of planes from the device, or the host extraction for a window the
    // kernels declined (more than kBalmCutMaxPlanes planes, coordinates outside the key range).
    DevBuf<uint8_t> d_cut;            // the work space of the extraction
    PinnedBuf<int32_t> h_cut_result;  // [4] planes, declined, roots, planes found
    std::vector<LidarPose> twl_build;
    const tc2li_lidar_window* win_build = nullptr;
    bool cut_pending = false;
    int finish_cut(hipStream_t st);

    // argument checks + LiDAR poses of the window keyframes at `poses7`
    static int window_poses(const double* poses7, int n_poses, const tc2li_lidar_window* win, std::vector<LidarPose>& twl);
    // planes of the window at the poses `poses7` (Tcw per keyframe, rows pose_index of the array)
    int build(const double* poses7, int n_poses, const tc2li_lidar_window* win, hipStream_t st, BalmCutTask* cut = nullptr);
    // the same from keyframe records whose first 12 doubles are Rcw (row-major), tcw (tc2li_inertial_keyframe); body mode
    int build_body(const void* kfs, size_t kf_bytes, int n_kfs, const tc2li_lidar_window* win, const float* Tbl7, size_t imu_pose_bytes,
                   hipStream_t st, BalmCutTask* cut = nullptr);
    int compute_error(const Se3* d_poses, hipStream_t st);  // EdgeLidarSE3::computeError
    int linearize(const Se3* d_poses, hipStream_t st);      // EdgeLidarSE3::linearizeOplus
    // The same two steps split into "enqueue the kernels" and "use the numbers after the caller's synchronisation", so
    // that the optimiser's loop synchronises once per phase.  enqueue_linearization always evaluates the Hessian (whether
    // the edge will take it is only known once the residual is back).
    void enqueue_error(const Se3* d_poses, hipStream_t st);
    void finish_error();
    int enqueue_linearization(const Se3* d_poses, hipStream_t st);
    void finish_linearization();
    double chi2() const { return error * information * error; }
    // EdgeLidarSE3::computeQuadraticFormLidarRes: add to the dense pose-pose system (free pose numbering pose_var)
    void add_quadratic_form(const int* pose_var, int ld, double* Hpp, double* b) const;

private:
    int upload(const std::vector<LidarPose>& twl, const tc2li_lidar_window* win, hipStream_t st, BalmCutTask* cut);
    int set_planes(int n);  // everything sized by the number of planes
    static int check_window(const tc2li_lidar_window* win, int n_poses);
};

}  // namespace tc2li
