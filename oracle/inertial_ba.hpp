// TEST INFRASTRUCTURE ONLY -- CPU oracle, never linked into or called from the product path.
//
// Restatement of the visual-inertial local bundle adjustment:
//   Optimizer::LocalInertialBA (graph, Levenberg-Marquardt settings)          SF/src/Optimizer.cc:1512-2085
//   ImuCamPose (ctor from a keyframe, Project, ProjectStereo, Update)         SF/src/G2oTypes.cc:35-83, 178-232
//   VertexPose / VertexVelocity / VertexGyroBias / VertexAccBias              SF/include/G2oTypes.h:139-255
//   EdgeMono / EdgeStereo (error, Jacobians in the body-frame parameterisation)  SF/include/G2oTypes.h:378-460, SF/src/G2oTypes.cc:358-436
//   EdgeInertial (information with eigenvalue clamp, error, Jacobians)        SF/src/G2oTypes.cc:499-601
//   EdgeGyroRW / EdgeAccRW                                                   SF/include/G2oTypes.h:645-714
//   ExpSO3, LogSO3, RightJacobianSO3, InverseRightJacobianSO3, NormalizeRotation   SF/src/G2oTypes.cc:783-865, SF/include/G2oTypes.h:76-81
//   OptimizerWithLidar::LocalLVIBA: the same graph plus EdgeLidar over the first min(N, 6) optimisable keyframes when N > 5
//                                                                            SF/src/OptimizerWithLidar.cc:489-1100 (edge: 697-725)
//   g2o: BaseMultiEdge::constructQuadraticForm / computeQuadraticForm         Thirdparty/g2o/g2o/core/base_multi_edge.hpp:60-222
//        (robustInformation = rho'[1] * information, core/base_edge.h:96-102), LM and Schur as in ba.hpp
// Eigen pieces not in tree are replaced: JacobiSVD (NormalizeRotation) by the polar factor via Newton iteration,
// SelfAdjointEigenSolver<9x9> by cyclic Jacobi rotations, the 9x9 / 3x3 inverses by Gauss-Jordan with partial pivoting.
// PARITY UNPINNED: the reference has no tests or vectors for these.
#pragma once
#include <cstdint>
#include <vector>

#include "ba.hpp"
#include "balm.hpp"
#include "imu.hpp"

namespace oracle {

struct ImuCalibD { double Rcb[9], tcb[3], Rbc[9], tbc[3]; };  // mImuCalib.mTcb / mTbc widened from float
struct InertialKeyFrame {
    double Rcw[9], tcw[3];  // camera pose (Sophus::SE3f widened)
    double Rwb[9], twb[3];  // GetImuRotation / GetImuPosition
    double v[3], bg[3], ba[3];
    uint8_t fixed = 0, has_imu = 1;
};
struct InertialLink {  // EdgeInertial + EdgeGyroRW + EdgeAccRW between keyframe kf1 (earlier) and kf2
    int kf1 = 0, kf2 = 0;
    const Preintegrated* pint = nullptr;  // pKF2->mpImuPreintegrated after SetNewBias(pKF1->GetImuBias())
    bool robust = false;                   // Huber sqrt(16.92) (the link to the fixed keyframe, or bRecInit)
    double info_scale = 1.0;               // 1e-2 for the oldest link
};
struct InertialBAResult {
    std::vector<double> chi2;        // per visual edge
    std::vector<uint8_t> depth_pos;  // per visual edge
    double err = 0, err_end = 0;     // activeRobustChi2 before / after optimize()
    int iterations = 0;
    LMTrace trace;
};

// Keyframes in vertex-id order; points, edges as in LocalBundleAdjustment (edge.pose indexes kfs).  Updates kfs (poses,
// velocities, biases) and points in place.  lidar (optional): the EdgeLidar of LocalLVIBA (body = true, already built:
// AddFromKeyFrame + BuildVoxHess, information = mWeightLocalBA) over the keyframes lidar_kf in vertex order.
InertialBAResult LocalInertialBA(std::vector<InertialKeyFrame>& kfs, const ImuCalibD& calib, std::vector<double>& points,
                                 const std::vector<BAEdge>& edges, const std::vector<InertialLink>& links, const Camera& cam,
                                 int iterations, double lambda_init, EdgeLidar* lidar = nullptr, const std::vector<int>* lidar_kf = nullptr);

// ---- Optimizer::PoseInertialOptimizationLastKeyFrame / LastFrame (SF/src/Optimizer.cc:2469-2852, 2854-3270) ---------------------------
// The per-frame optimiser of Tracking::TrackLocalMap once the IMU is initialised (Tracking.cc:2872/2877): Gauss-Newton with a dense
// LDLT (g2o OptimizationAlgorithmGaussNewton + LinearSolverDense, core/optimization_algorithm_gauss_newton.cpp:49-93,
// solvers/linear_solver_dense.h:65-113), vertices pose / velocity / gyro bias / accelerometer bias of the frame (and of the previous
// frame when last_frame), unary EdgeMonoOnlyPose / EdgeStereoOnlyPose (G2oTypes.h:400-503, G2oTypes.cc:384-463) with Huber
// sqrt(5.991) / sqrt(7.815), EdgeInertial + EdgeGyroRW + EdgeAccRW to the other state, EdgePriorPoseImu (G2oTypes.cc:727-767, Huber 5)
// on the previous frame; 4 rounds x 10 iterations with the inlier tests {12, 7.5, 5.991, 5.991} (last keyframe) or 5.991 (last frame)
// for monocular edges (x 1.5 for points closer than 10 m) and {15.6, 9.8, 7.815, 7.815} for stereo edges, chi2 compared as float;
// the recovery pass (:2738-2765, :3160-3188); the Hessian of the new prior (15 x 15: GetHessian2 blocks, or the 30 x 30 system
// marginalised over the previous frame, Optimizer::Marginalize :2087-2166) through ConstraintPoseImu's eigenvalue clamp
// (G2oTypes.h:716-740).  JacobiSVD of the (symmetric) marginalised block = its eigen decomposition (cyclic Jacobi).
struct PoseImuPrior { double Rwb[9], twb[3], vwb[3], bg[3], ba[3], H[225]; };  // ConstraintPoseImu (mpcpi)
struct PoseInertialResult {
    int n_initial = 0, n_bad = 0, n_inliers = 0;   // returns n_initial - n_bad
    std::vector<uint8_t> outlier;                   // mvbOutlier per edge
    PoseImuPrior prior;                             // pFrame->mpcpi
    bool solver_failed = false;
};
// cur: the frame (updated in place); other: the last keyframe (fixed) or, when last_frame, the previous frame (free, updated too).
// pint: the pre-integration of EdgeInertial (mpImuPreintegrated / mpImuPreintegratedFrame); pint_rw: the one whose bias-walk covariance
// gives the random-walk informations (always pFrame->mpImuPreintegrated, :2645/:3049).  edges[e].point indexes Xw; close[e] = mTrackDepth < 10.
PoseInertialResult PoseInertialOptimization(InertialKeyFrame& cur, InertialKeyFrame& other, bool last_frame, const PoseImuPrior* prior_prev,
                                            const ImuCalibD& calib, const Preintegrated& pint, const Preintegrated& pint_rw,
                                            const std::vector<double>& Xw, const std::vector<BAEdge>& edges, const std::vector<uint8_t>& close,
                                            const Camera& cam, bool bRecInit);

// ---- IMU initialisation: Optimizer::InertialOptimization, first overload (SF/src/Optimizer.cc:2169-2356) -----
// Levenberg-Marquardt (user lambda 1e3 when priorG != 0, at most `its` = 200 iterations, g2o's stop rules) over the keyframe velocities,
// one gyro and one accelerometer bias, the gravity direction (2 dof, VertexGDir) and -- monocular only -- the scale (VertexScale); the
// keyframe poses are fixed; one EdgeInertialGS per consecutive pair + EdgePriorAcc / EdgePriorGyro (bprior = 0).  kfs in temporal order
// (Rwb, twb, v used; v updated); pints[i] = kfs[i]'s pre-integration from kfs[i-1] (pints[0] unused; evaluated at the CURRENT biases
// through the bias Jacobians, as SetNewBias + GetDelta*(b) do).
struct InertialInitResult { int iterations = 0, trials = 0; double err = 0, err_end = 0; LMTrace trace; };
InertialInitResult InertialOptimization(std::vector<InertialKeyFrame>& kfs, const std::vector<const Preintegrated*>& pints, double Rwg[9], double& scale,
                                        double bg[3], double ba[3], bool mono, bool fixed_vel, float priorG, float priorA, int its = 200);
// Second overload, Optimizer::InertialOptimization(pMap, Rwg, scale) (SF/src/Optimizer.cc:2359-2466, LocalMapping::ScaleRefinement): Gauss-Newton,
// `its` = 10, gravity direction and scale only, Huber(1) on every EdgeInertialGS, the keyframes' own velocities and biases fixed.
int InertialScaleRefinement(const std::vector<InertialKeyFrame>& kfs, const std::vector<const Preintegrated*>& pints, double Rwg[9], double& scale, int its,
                            double err2[2]);
// LocalMapping::InitializeIMU's first gravity direction and keyframe velocities (SF/src/LocalMapping.cc:1241-1270), float arithmetic
void InitialGravityDirection(const std::vector<InertialKeyFrame>& kfs, const std::vector<const Preintegrated*>& pints, float vel[], float Rwg[9]);
void inertial_gs_edge(const InertialKeyFrame& k1, const InertialKeyFrame& k2, const double bg[3], const double ba[3], const double Rwg[9], double s,
                      const Preintegrated& pint, double err[9], double* J /* 9 x 15 or NULL */);

// exposed for unit tests
void ExpSO3(const double w[3], double R[9]);
void LogSO3(const double R[9], double w[3]);
// EdgeInertial: 9-vector error and the six Jacobian blocks (9 x {6,3,3,3,6,3}, row-major, concatenated: 9 x 24)
void inertial_edge(const InertialKeyFrame& k1, const InertialKeyFrame& k2, const Preintegrated& pint, double err[9], double J[9 * 24]);
// visual edge in the ImuCamPose parameterisation: error (2 or 3), d/d point (dim x 3), d/d pose increment (dim x 6)
int inertial_visual_edge(const InertialKeyFrame& kf, const ImuCalibD& calib, const double X[3], const BAEdge& e, const Camera& cam, double err[3],
                         double A[9], double B[18]);
void imu_pose_update(InertialKeyFrame& kf, int& its, const ImuCalibD& calib, const double u[6]);

}  // namespace oracle
