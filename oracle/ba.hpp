// TEST INFRASTRUCTURE ONLY -- CPU oracle, never linked into or called from the product path.
//
// Restatement of the optimisation back end on the visual path:
//   Optimizer::PoseOptimization                          SF/src/Optimizer.cc:816-1116
//   Optimizer::LocalBundleAdjustment / OptimizerWithLidar::LocalLVBundleAdjustment (visual part)
//                                                        SF/src/Optimizer.cc:1118-1510, SF/src/OptimizerWithLidar.cc:60-487
//   g2o::OptimizationAlgorithmLevenberg::solve           Thirdparty/g2o/g2o/core/optimization_algorithm_levenberg.cpp:61-194
//   g2o::BlockSolver buildSystem / setLambda / solve     Thirdparty/g2o/g2o/core/block_solver.hpp:353-607
//   BaseUnaryEdge / BaseBinaryEdge::constructQuadraticForm  core/base_unary_edge.hpp:43-72, core/base_binary_edge.hpp:55-137
//   RobustKernelHuber                                    core/robust_kernel_impl.cpp:65-91
//   EdgeStereoSE3ProjectXYZ[OnlyPose], EdgeSE3ProjectXYZ[OnlyPose]  types/types_six_dof_expmap.{h,cpp}, SF/src/OptimizableTypes.cpp
//   SE3Quat (exp, map, operator*)                        types/se3quat.h
// PARITY UNPINNED: the reference has no tests or vectors for these; Eigen's LDLT (linear_solver_dense.h /
// linear_solver_eigen.h) is replaced by a plain dense LDL^T, which changes round-off only.
#pragma once
#include <cstdint>
#include <vector>

#include "balm.hpp"

namespace oracle {

struct SE3Quat {
    double q[4] = {0, 0, 0, 1};  // x, y, z, w
    double t[3] = {0, 0, 0};
};

struct Camera { double fx, fy, cx, cy, bf; };

struct BAEdge {
    int point = 0, pose = 0;   // indices into the points / poses arrays (pose-only problems: pose is 0)
    double obs[3] = {0, 0, 0};  // u, v, u_right (u_right < 0: monocular observation)
    double info = 1;            // invSigma2 (information = info * I)
};

struct LMTrace {  // one entry per outer iteration: for comparing optimiser behaviour step by step
    std::vector<double> chi2, lambda;
    std::vector<int> trials;
};

// Optimizer::PoseOptimization.  Xw[i] / edges[i]: the frame's map-point correspondences.  Returns the number of
// inliers (nInitialCorrespondences - nBad); `pose` is replaced by the optimised pose rounded through float like
// Frame::SetPose(Sophus::SE3f) does; outlier[i] as mvbOutlier.
int PoseOptimization(SE3Quat& pose, const std::vector<double>& Xw, const std::vector<BAEdge>& edges, const Camera& cam,
                     std::vector<uint8_t>& outlier, LMTrace* trace = nullptr);

struct BAResult {
    std::vector<double> chi2;        // per edge, at the final estimate
    std::vector<uint8_t> depth_pos;  // per edge: isDepthPositive()
    int iterations = 0;
    LMTrace trace;
};

// The optimisation of LocalBundleAdjustment: poses (with fixed flags) in vertex-id order, points, edges; Huber
// kernels sqrt(5.991) / sqrt(7.815); optimize(iterations) with lambda_init <= 0 meaning tau * max diagonal.
// `stop` is polled like g2o's forceStopFlag.  Poses and points are updated in place (double precision).
// lidar (optional): the BALM edge of LocalLVBundleAdjustment (OptimizerWithLidar.cc:226-260) over the window poses
// `lidar_pose` (indices into `poses`, in window order), already built (AddFromKeyFrame + BuildVoxHess), information = wLBA.
BAResult LocalBundleAdjustment(std::vector<SE3Quat>& poses, const std::vector<uint8_t>& fixed, std::vector<double>& points,
                               const std::vector<BAEdge>& edges, const Camera& cam, int iterations, double lambda_init,
                               const bool* stop = nullptr, EdgeLidar* lidar = nullptr, const std::vector<int>* lidar_pose = nullptr);

// helpers exposed for unit tests
SE3Quat se3_exp(const double update[6]);
SE3Quat se3_mul(const SE3Quat& a, const SE3Quat& b);
void se3_map(const SE3Quat& T, const double X[3], double out[3]);
void quat_to_matrix_public(const double q[4], double R[9]);  // Eigen::Quaterniond::toRotationMatrix
// error (2 or 3 rows), Jacobians wrt point (A: dim x 3) and pose (B: dim x 6) of the binary projection edge
int edge_linearize(const SE3Quat& T, const double X[3], const BAEdge& e, const Camera& cam, double err[3], double A[9], double B[18]);

}  // namespace oracle
