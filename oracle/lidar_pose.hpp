// TEST INFRASTRUCTURE ONLY -- CPU oracle, never linked into or called from the product path.
//
// Restatement of the pose plumbing between the camera thread and the LiDAR front end:
//   UpdateLidarPose                         SF/include/lidar_front_end/LidarFrontEnd.cpp:786-800
//   InterpolateSE3                          SF/src/Tracking.cc:1552-1563
//   LidarFrontEndTools::transformPointCloud SF/src/LidarTypes.cc:42-65
//   Tracking::SyncWithLidar (frame chain)   SF/src/Tracking.cc:1565-1630   (transformOffset = Tlc * Tcw(frame) * cloudTwc * Tcl)
//   Tracking::BuildLidarFeat4KeyFrame       SF/src/Tracking.cc:1510-1550   (transformOffset = Tlc * Tcw(cur) * (rel * Tcw(refKF))^-1 * Tcl)
// All of it is Sophus::SE3f / Eigen float arithmetic (the vendored Sophus: Thirdparty/Sophus/sophus/so3.hpp:229-345, 583-618,
// se3.hpp:208-260, 304-310, 761-782; Eigen is not in the tree: Quaternionf(Matrix3f), slerp and toRotationMatrix are restated from
// Eigen 3.3's Geometry/Quaternion.h).  Every float operation is written out in the order the expression templates evaluate it
// coefficient by coefficient; horizontal sums are taken left to right.  PARITY UNPINNED (no vectors in the reference).
#pragma once
#include <vector>

#include "lidar.hpp"

namespace oracle {

struct SE3F { float q[4] = {0, 0, 0, 1}; float t[3] = {0, 0, 0}; };  // Sophus::SE3f: unit quaternion (x, y, z, w) + translation

SE3F se3f_inverse(const SE3F& T);
SE3F se3f_mul(const SE3F& a, const SE3F& b);
void se3f_log(const SE3F& T, float out6[6]);       // (upsilon, omega)
SE3F se3f_exp(const float a6[6]);
void se3f_rotation_matrix(const SE3F& T, float R[9]);
SE3F InterpolateSE3(const SE3F& a, const SE3F& b, float t);

// state.rot / state.pos (and pos_lid) from the last frame's pose, the velocity and the fraction of the frame period that has passed;
// offset_* of `state` are read
void UpdateLidarPose(const SE3F& Tcw_last, const SE3F& velocity, double timeFromLastFrame, const SE3F& Tcl, LidarState& state, double pos_lid[3]);

PointVector transformPointCloud(const PointVector& in, const SE3F& T);

// Tracking::SyncWithLidar: the transform applied to the scan's feature cloud when it is paired with `Tcw_frame` (= the current or the
// last frame); ratio = (t_cloud - t_last) / (t_cur - t_last)
SE3F sync_transform(const SE3F& Tcw_frame, const SE3F& Tcw_last, const SE3F& Tcw_cur, float ratio, const SE3F& Tlc, const SE3F& Tcl);
// Tracking::BuildLidarFeat4KeyFrame: rel = the scan frame's pose relative to its reference keyframe (mlRelativeFramePoses)
SE3F keyframe_transform(const SE3F& Tcw_cur, const SE3F& rel, const SE3F& Tcw_refkf, const SE3F& Tlc, const SE3F& Tcl);

}  // namespace oracle
