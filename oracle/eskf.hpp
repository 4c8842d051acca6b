// TEST INFRASTRUCTURE ONLY -- CPU oracle, never linked into or called from the product path.
//
// Restatement of the iterated error-state Kalman filter of the camera-LiDAR-inertial branch:
//   esekf::predict                                   SF/include/IKFoM_toolkit/esekfom/esekfom.hpp:281-392
//   esekf::update_iterated_dyn_share_modified        esekfom.hpp:1621-1932
//   get_f, df_dx, df_dw                              SF/src/use-ikfom.cpp:45-91 (state / input / noise layout: SF/include/use-ikfom.hpp:43-67)
//   h_share_model (measurement model, H rows)        SF/include/lidar_front_end/LidarFrontEnd.cpp:485-602
//   MTK::S2 {boxplus, boxminus, S2_Bx, S2_Nx_yy, S2_Mx}, MTK::SO3 {boxplus, boxminus}   IKFoM_toolkit/mtk/types/S2.hpp, SOn.hpp
//   MTK::A_matrix, exp, log, tolerance               IKFoM_toolkit/mtk/src/mtkmath.hpp
// State: pos, rot, offset_R_L_I, offset_T_L_I, vel, bg, ba, grav (S2 of length 9.809, chart type 1): 24 numbers, 23 degrees of
// freedom in this order.  Rotations are kept as matrices (the reference keeps quaternions; boxplus is R <- R Exp(d), boxminus the
// logarithm of other^T R through the quaternion of that matrix).
// Reference quirks kept on purpose: `scalar(1/2)` is an integer division, so the rotation blocks of F_x1 in predict and the
// exponential inside S2_Mx are the identity (esekfom.hpp:314,344; S2.hpp:253); the neighbour search of h_share_model runs only in
// iterations that follow a converged one, otherwise the previous neighbours AND the previous selection are reused (:519-527).
// Eigen's fixed-size inverse (partial-pivot LU) is restated as LU with partial pivoting.
// PARITY UNPINNED: the reference has no tests or vectors for these.
#pragma once
#include <vector>

#include "lidar.hpp"

namespace oracle {

constexpr int kEskfN = 23;

void mtk_A_matrix(const double v[3], double A[9]);
void s2_Bx(const double g[3], double Bx[6]);                          // 3 x 2 row-major
void s2_Nx_yy(const double g[3], double Nx[6]);                       // 2 x 3
void s2_Mx(const double g[3], const double delta[2], double Mx[6]);   // 3 x 2
void eskf_boxplus(ImuState& x, const double d[kEskfN]);
void eskf_boxminus(const ImuState& x, const ImuState& other, double d[kEskfN]);

// one IMU step: x <- x oplus f(x, u) dt, P <- F P F^T + (dt G) Q (dt G)^T.  P: 23 x 23, Q: 12 x 12 (ng, na, nbg, nba), row-major
void eskf_predict(ImuState& x, double* P, const double* Q, const double acc[3], const double gyr[3], double dt);

struct EskfUpdate {
    int calls = 0;           // h_share_model invocations
    int effct_feat_num = 0;  // of the last invocation
    int searches = 0;        // invocations that ran the neighbour search
    int converged = 0;       // t
    bool finished = false;   // the covariance update ran
    double res_mean_last = 0;
};
EskfUpdate eskf_update(ImuState& x, double* P, const KdTree& tree, const PointVector& feats_down_body, double R, int maximum_iter,
                       const double* limit, bool extrinsic_est_en);

// same forward propagation as ForwardPropagate with the covariance (IMU_Processing.cpp:176-236); Q's diagonal = cov_gyr, cov_acc,
// cov_bias_gyr, cov_bias_acc
std::vector<Pose6D> ForwardPropagateCov(ImuState& st, double* P, const double cov4x3[12], const std::vector<ImuMeas>& v_imu, double pcl_beg_time,
                                        double pcl_end_time, double last_lidar_end_time, double acc_scale, const double acc_s_last[3],
                                        const double angvel_last[3]);

}  // namespace oracle
