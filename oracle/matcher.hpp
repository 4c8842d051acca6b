// TEST INFRASTRUCTURE ONLY -- CPU oracle, never linked into or called from the product path.
//
// Restatement of the projection-guided matching on the tracking path:
//   ORBmatcher::SearchByProjection(Frame&, const Frame& LastFrame, th, bMono)      SF/src/ORBmatcher.cc:1685-1896
//   ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th, far...)   SF/src/ORBmatcher.cc:52-222
//   ORBmatcher::ComputeThreeMaxima                                                 SF/src/ORBmatcher.cc:2021-2062
//   Frame::isInFrustum (Nleft == -1 branch), MapPoint::PredictScale                SF/src/Frame.cc:542-603, SF/src/MapPoint.cc:540-555
//   Frame::GetFeaturesInArea / PosInGrid / AssignFeaturesToGrid                    SF/src/Frame.cc:412-443,687-765 (stereo.hpp)
// Both overloads are split here into "build the per-point query" (projection, window, level range, descriptor) and
// "greedy sequential matching of the queries" (the loop bodies), which composes to the reference functions.
// PARITY UNPINNED: the reference has no tests or vectors for these.
#pragma once
#include "stereo.hpp"

namespace oracle {

struct ProjQuery {
    float u = 0, v = 0, radius = 0, u_right = -1;
    int min_level = -1, max_level = -1;
    float angle = 0;        // source keypoint angle (rotation histogram of the last-frame overload)
    int valid = 0;          // 0: this source item produces no search
    int has_observations = 1;  // pMP->Observations() > 0: a match by this query blocks later queries
    uint8_t desc[32] = {0};
};

struct FrameView {
    std::vector<KeyPoint> keys;          // mvKeysUn
    std::vector<uint8_t> desc;           // N x 32
    std::vector<float> uRight;           // mvuRight
    std::vector<uint8_t> occupied;       // mvpMapPoints[i] && Observations() > 0 before the search
    int cols = 0, rows = 0;
};

enum MatchMode { MATCH_BEST = 0, MATCH_RATIO = 1 };

// The loop bodies of both overloads: returns nmatches before the rotation filter; match_of_query[q] = keypoint or -1.
int match_queries(const FrameView& F, const std::vector<ProjQuery>& queries, MatchMode mode, float nnratio,
                  std::vector<int>& match_of_query);

// Rotation consistency (ORBmatcher.cc:1858-1881): removes matches outside the three dominant 30-bin rotation bins;
// returns how many were removed.
int rotation_filter(const FrameView& F, const std::vector<ProjQuery>& queries, std::vector<int>& match_of_query);

struct SE3f { float q[4]; float t[3]; };  // Sophus::SE3f: unit quaternion (x, y, z, w) + translation
struct CamF { float fx, fy, cx, cy; };

// Query construction of the last-frame overload (ORBmatcher.cc:1696-1739): one query per last-frame keypoint.
std::vector<ProjQuery> build_queries_last_frame(const SE3f& Tcw, const SE3f& Tlw, const CamF& cam, float mb, float mbf,
                                                const std::vector<float>& scale_factors, int cols, int rows,
                                                const std::vector<uint8_t>& has_point, const std::vector<uint8_t>& outlier,
                                                const std::vector<float>& Xw, const std::vector<KeyPoint>& last_keys,
                                                const std::vector<uint8_t>& mp_desc, float th, bool bMono);

struct MapPointView {
    float pos[3], normal[3], min_dist, max_dist;  // GetWorldPos, GetNormal, Get{Min,Max}DistanceInvariance (max = mfMaxDistance*1.2)
    float mfMaxDistance;
    uint8_t desc[32];
};
// Frame::isInFrustum(pMP, 0.5) + the window of the local-map overload (ORBmatcher.cc:62-81, Tracking.cc:3232-3282).
std::vector<ProjQuery> build_queries_local_map(const SE3f& Tcw, const CamF& cam, float mbf, const std::vector<float>& scale_factors,
                                               float log_scale_factor, int cols, int rows, const std::vector<MapPointView>& mps,
                                               float th, bool bFarPoints, float thFarPoints, float viewingCosLimit = 0.5f);

}  // namespace oracle
