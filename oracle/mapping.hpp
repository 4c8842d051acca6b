// TEST INFRASTRUCTURE ONLY -- CPU oracle, never linked into or called from the product path.
//
// Restatement of the geometric core of LocalMapping::CreateNewMapPoints, pinhole cameras, no
// second camera model (mpCamera2 == NULL: the KITTI stereo / mono case):
//   ORBmatcher::SearchForTriangulation        SF/src/ORBmatcher.cc:916-1150  (matching restricted to common vocabulary nodes,
//                                             best distance <= TH_LOW with ties going to the later feature, epipole distance test
//                                             for mono-mono pairs, epipolar line test, optional rotation histogram)
//   Pinhole::epipolarConstrain                SF/src/CameraModels/Pinhole.cpp:116-138
//   LocalMapping::CreateNewMapPoints          SF/src/LocalMapping.cc:402-726 (parallax gates, triangulation or stereo un-projection,
//                                             depth / reprojection / scale-consistency gates); a keypoint of the current keyframe that
//                                             received a point from an earlier neighbour is skipped for the later ones
//   GeometricTools::Triangulate               SF/src/GeometricTools.cc:56-75
//   KeyFrame::UnprojectStereo                 SF/src/KeyFrame.cc:767-784
// Eigen::JacobiSVD<Matrix4f> (not in tree) is replaced by the eigenvector of the smallest eigenvalue of A^T A (cyclic Jacobi in
// double): the triangulated point agrees to float rounding, not bit for bit.  The vocabulary (DBoW2 FeatureVector) is an input:
// node ids ascending, per node the feature indices in insertion order.
// PARITY UNPINNED: the reference has no tests or vectors for these.
#pragma once
#include <cstdint>
#include <vector>

#include "matcher.hpp"

namespace oracle {

struct KeyFrameView {
    int n = 0;
    const KeyPoint* keys = nullptr;      // mvKeysUn
    const uint8_t* desc = nullptr;       // [n][32]
    const float* u_right = nullptr;      // mvuRight
    const float* depth = nullptr;        // mvDepth
    const uint8_t* has_point = nullptr;  // GetMapPoint(i) != NULL
    int n_nodes = 0;
    const int32_t* fv_node = nullptr;    // mFeatVec keys, ascending
    const int32_t* fv_off = nullptr;     // [n_nodes + 1]
    const int32_t* fv_idx = nullptr;     // feature indices of each node
    SE3f Tcw{};
};

// match12[idx1] = idx2 or -1; returns nmatches.  has_point1 overrides kf1.has_point when given (CreateNewMapPoints updates it).
int SearchForTriangulation(const KeyFrameView& kf1, const KeyFrameView& kf2, const CamF& cam, const std::vector<float>& scale_factors,
                           const std::vector<float>& level_sigma2, bool only_stereo, bool coarse, bool check_orientation,
                           std::vector<int>& match12, const uint8_t* has_point1 = nullptr);

struct NewMapPoint { int idx1, neighbour, idx2, stereo; float x3D[3]; };
struct MappingParams { float mb, mbf, scale_factor; bool inertial, far_points; float th_far_points; };
// the loop over the neighbours of the current keyframe (already chosen by the caller), points in creation order
// ORBmatcher::Fuse(pKF, vpMapPoints, th, bRight = false), the search part (SF/src/ORBmatcher.cc:1157-1330): per map point the keypoint
// of the keyframe it would be fused into (best descriptor distance <= TH_LOW among the keypoints inside the projection window that
// pass the level and reprojection gates) or -1.  valid[i] = pMP && !pMP->isBad() && !pMP->IsInKeyFrame(pKF).  Replace / AddObservation
// stay with the caller (they act on the object graph, in list order).
int FuseSearch(const FrameView& kf, const SE3f& Tcw, const CamF& cam, float bf, const std::vector<float>& scale_factors,
               const std::vector<float>& inv_level_sigma2, float log_scale_factor, const std::vector<MapPointView>& points,
               const std::vector<uint8_t>& valid, float th, std::vector<int>& best_idx, std::vector<int>& best_dist);

std::vector<NewMapPoint> CreateNewMapPoints(const KeyFrameView& cur, const std::vector<KeyFrameView>& neighbours, const CamF& cam,
                                            const std::vector<float>& scale_factors, const std::vector<float>& level_sigma2,
                                            const MappingParams& prm, bool coarse);

}  // namespace oracle
