// TEST INFRASTRUCTURE ONLY -- CPU oracle, never linked into or called from the product path.
//
// Restatement of the LiDAR (BALM2) term of the local bundle adjustment:
//   cut_voxel                                             SF/src/bavoxel.cc:42-91
//   OCTO_TREE_NODE::{judge_eigen, cut_func, recut, tras_opt}   SF/include/bavoxel.h:492-602, 723-740
//   VOX_HESS::{push_voxel, acc_evaluate2, evaluate_only_residual}  SF/include/bavoxel.h:57-78, 80-196, 276-315
//   BALM2::divide_thread / only_residual                  SF/include/bavoxel.h:778-817, 857-862
//   PointCluster, VOXEL_LOC hash                          SF/include/tools.h:163-214, 54-79
//   LidarCovisRes::{AddFromKeyFrame, BuildVoxHess, UpdatePose(2 args), ComputeError, ComputeJandHSE3}
//                                                         SF/src/LidarRes.cc:32-75, 136-186, 221-235
//   EdgeLidarSE3::{computeError, linearizeOplus, constructQuadraticForm, computeQuadraticFormLidarRes}
//                                                         SF/include/G2oTypesWithLidar.h:118-236
//   LidarCovisRes::ComputeJandH (body-frame increment of VertexPose)  SF/src/LidarRes.cc:89-128
//   EdgeLidar::{computeError, linearizeOplus, computeQuadraticFormLidarRes} (the edge of LocalLVIBA)
//                                                         SF/src/G2oTypesWithLidar.cc:33-140
//   InverseRightJacobianSO3                               SF/src/G2oTypes.cc:823-839
// The reference quirks listed in SURVEY.md section 7 are reproduced on purpose (Hessian.block<6,6>(i,i) with element
// offsets, b -= info * A^T without the residual, Hessian reuse while the cost grows, float SE3 round trip in UpdatePose).
// Eigen::SelfAdjointEigenSolver<Matrix3d> (not in tree) is replaced by cyclic Jacobi rotations (eigenvalues ascending).
// PARITY UNPINNED: the reference has no tests or vectors for these.
#pragma once
#include <cstdint>
#include <memory>
#include <unordered_map>
#include <vector>

namespace oracle {

struct V3 { double x = 0, y = 0, z = 0; double& operator[](int i) { return (&x)[i]; } double operator[](int i) const { return (&x)[i]; } };
struct M3 { double m[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}; double& operator()(int r, int c) { return m[3 * r + c]; } double operator()(int r, int c) const { return m[3 * r + c]; } };

struct IMUST { M3 R; V3 p; };

struct PointCluster {  // tools.h:163-214
    M3 P; V3 v; int N = 0;
    void push(const V3& vec);
    M3 cov() const;
    PointCluster& operator+=(const PointCluster& o);
    void transform(const PointCluster& sigv, const IMUST& stat);
};

// Eigenvalues ascending, eigenvectors in the columns of U.
void eig3(const M3& A, double lambda[3], M3& U);

struct PlaneVoxel {            // one entry of VOX_HESS: plvec_voxels[a] (per window slot) and coeffs[a]
    std::vector<PointCluster> sig_orig;
    double coe = 0;
};

struct SE3fQ { float q[4] = {0, 0, 0, 1}; float t[3] = {0, 0, 0}; };  // Sophus::SE3f

// Sophus::SE3f(R.cast<float>(), t.cast<float>())
SE3fQ se3f_from_rt(const double R[9], const double t[3]);

class LidarCovisRes {
public:
    explicit LidarCovisRes(const SE3fQ& Tcl) : mTcl(Tcl) {}
    LidarCovisRes(const SE3fQ& Tcl, const SE3fQ& Tbl) : mTcl(Tcl), mTbl(Tbl) {}
    int win_size_ = 20;
    // pose = Tcw of the keyframe (Sophus::SE3f), cloud = its surface cloud in the LiDAR frame (x, y, z per point)
    void AddFromKeyFrame(const SE3fQ& Tcw, const std::vector<float>& cloud_xyz);
    void BuildVoxHess();
    void UpdatePose(int i, const double Rcw[9], const double tcw[3]);
    double ComputeError() const;
    void ComputeJandHSE3(std::vector<double>& JacT, std::vector<double>& Hess) const;  // 6W and (6W)^2 row-major
    void ComputeJandH(std::vector<double>& JacT, std::vector<double>& Hess) const;     // w.r.t. ImuCamPose::Update's increment
    const std::vector<PlaneVoxel>& planes() const { return mVoxHess; }
    const std::vector<IMUST>& poses() const { return mPoseBuf; }
    double divide_thread(std::vector<double>& Hess, std::vector<double>& JacT) const;

private:
    struct Node;
    struct LocHash { size_t operator()(const std::array<int64_t, 3>& s) const; };
    std::unordered_map<std::array<int64_t, 3>, std::shared_ptr<Node>, LocHash> mSurfMap;
    std::vector<PlaneVoxel> mVoxHess;
    std::vector<IMUST> mPoseBuf;
    IMUST mPose0;
    SE3fQ mTcl, mTbl;
    int mCurrPosId = 0;
    void acc_evaluate2(int head, int end, std::vector<double>& Hess, std::vector<double>& JacT, double& residual) const;
};

// The edge state machine of EdgeLidarSE3 (G2oTypesWithLidar.h:88-236) over W window vertices; body = true is EdgeLidar on
// VertexPose (G2oTypesWithLidar.cc:33-75): error = sqrt(r), derivatives from ComputeJandH.
struct EdgeLidar {
    LidarCovisRes* lio = nullptr;
    double information = 1;
    bool body = false;
    std::vector<double> JacT, Hessian;
    double error = 0, r1 = 1000, r2 = 1000;
    bool is_calc_hess = true;
    void computeError(const double* Rcw9_per_vertex, const double* tcw3_per_vertex, int W);
    void linearizeOplus(const double* Rcw9_per_vertex, const double* tcw3_per_vertex, int W);
    double chi2() const { return error * information * error; }
};

}  // namespace oracle
