// CPU ORACLE -- TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).  PARITY UNPINNED: the
// reference has no tests or fixtures for this path and cannot be built here.
//
// Restatement of Tracking::UpdateLocalKeyFrames and Tracking::UpdateLocalPoints (slam_framework/src/Tracking.cc:3296-3476) over a
// flat mirror of the keyframe graph.  The reference keys its containers by object address (std::map<KeyFrame*, int>,
// std::set<KeyFrame*>), so its iteration order is whatever the allocator produced; here a keyframe is its index and index order
// stands in for address order -- the caller lists children / observations in the order its own containers iterate.
#pragma once
#include <cstdint>
#include <vector>

namespace oracle {

struct MapGraph {
    int n_keyframes = 0, n_points = 0;
    const uint8_t* kf_bad = nullptr;                                  // KeyFrame::isBad()
    const int32_t *covis_off = nullptr, *covis = nullptr;             // mvpOrderedConnectedKeyFrames (GetBestCovisibilityKeyFrames takes the first N)
    const int32_t *child_off = nullptr, *children = nullptr;          // GetChilds()
    const int32_t *parent = nullptr, *prev_kf = nullptr;              // GetParent(), mPrevKF; -1 = none
    const int32_t *match_off = nullptr, *matches = nullptr;           // GetMapPointMatches(): point per keypoint slot, -1 = none
    const uint8_t* point_bad = nullptr;                               // MapPoint::isBad()
    const int32_t *obs_off = nullptr, *obs_kf = nullptr;              // GetObservations(): the observing keyframes
};

struct LocalMap {
    std::vector<int32_t> keyframes, points;
    int32_t reference_kf = -1;          // pKFmax, -1 when no keyframe got a vote
    std::vector<uint8_t> frame_cleared; // frame points found bad: the reference NULLs them in the frame (:3345, :3368)
};

// frame_points: mCurrentFrame.mvpMapPoints (or mLastFrame's once the IMU is initialised, :3330 / :3350), -1 = none.
// temporal_last_kf: mCurrentFrame.mpLastKeyFrame for IMU_STEREO_LIDAR (the "10 last temporal KFs" block, :3453-3469), -1 otherwise.
LocalMap UpdateLocalMap(const MapGraph& g, const int32_t* frame_points, int n_frame_points, int temporal_last_kf);

}  // namespace oracle
