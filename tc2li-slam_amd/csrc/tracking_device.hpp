// Shared between tracking_host.cpp and tracking_kernels.hip: the per-frame bookkeeping of Tracking::TrackWithMotionModel
// (SF/src/Tracking.cc:2737-2834) and Tracking::TrackLocalMap (:3119-3230) for a batch of frames, on the device.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "matcher_device.hpp"
#include "pose_opt_device.hpp"

namespace tc2li {

// One frame of a tracking batch.  Keypoint-indexed arrays (u_right, occupied, map_point_of_keypoint, edges ...) give every frame
// `capacity` slots; query-indexed arrays are packed (q_off).
struct TrackFrameDev {
    float pose7[7];        // the pose the queries are projected with (prediction / the pose TrackWithMotionModel left)
    float last_pose7[7];   // TrackWithMotionModel: LastFrame.GetPose()
    float th;              // window factor of this pass
    int32_t forward, backward;  // TrackWithMotionModel: tlc.z > b / -tlc.z > b (ORBmatcher.cc:1707-1708), computed on the host
    int32_t q_off, n_q;
    int32_t key_off, n_keys;    // the frame's keypoints in the extractor's device arrays
    int32_t slot;               // index among the frames of THIS pass (the matcher's frame index), -1: not part of the pass
    int32_t pad_;
};

struct TrackConst {
    float cam4[4], b, bf;
    float scale[kMaxLevels], inv_sigma2[kMaxLevels];
    int32_t n_levels, cols, rows, capacity;
    float log_scale, th_far, view_cos_limit;
    int32_t far_points, mono, pad_;
};

// The last frame's points of all frames, packed (structure of arrays, query-indexed).
struct LastFrameArrays {
    const uint8_t* flags;  // bit 0: mvpMapPoints[i] != NULL, bit 1: mvbOutlier[i]
    const float* Xw;       // [3] each
    const float* angle;    // mvKeysUn[i].angle
    const int32_t* octave;
    const uint8_t* desc;   // [32] each
};

// tc2li_map_point, 68 bytes
struct LocalPointDev {
    float pos[3], normal[3];
    float min_distance, max_distance, max_distance_raw;
    uint8_t desc[32];
};
static_assert(sizeof(LocalPointDev) == 68, "layout of tc2li_map_point");

void launch_track_queries_last(const TrackFrameDev* frames, int n_frames, const TrackConst& C, const LastFrameArrays& A, int total_q, MatchQuery* queries,
                               int32_t* query_frame, int32_t* match, hipStream_t st);
// amb_count[0] = number of queries whose predicted scale level the host's logf has to decide (k_track_queries_local); their indices,
// ratios mfMaxDistance / dist and window factors follow in amb_ids / amb_ratio / amb_r (room for total_q each)
void launch_track_queries_local(const TrackFrameDev* frames, int n_frames, const TrackConst& C, const LocalPointDev* points, int total_q, MatchQuery* queries,
                                int32_t* query_frame, int32_t* match, int32_t* amb_count, int32_t* amb_ids, float* amb_ratio, float* amb_r, hipStream_t st);
void launch_track_patch_levels(const int32_t* ids, const int32_t* levels, const float* r, int n, const TrackConst& C, MatchQuery* queries, hipStream_t st);
void launch_track_occupied(const uint8_t* held, size_t n, uint8_t* occ, hipStream_t st);
// per frame of the pass: rotation histogram filter (check_orientation) and the number of matches
void launch_track_count(const TrackFrameDev* frames, const int32_t* pass_frames, int n_pass, const MatchQuery* queries, const float* key_angles,
                        int check_orientation, int32_t* match, int32_t* n_matches, hipStream_t st);
// TrackWithMotionModel: mvpMapPoints of the current frame + the edges of Optimizer::PoseOptimization
void launch_track_edges_last(const TrackFrameDev* frames, int n_frames, const TrackConst& C, const MatchKey* keys, const float* u_right, const int32_t* match,
                             const int32_t* n_matches, const float* last_Xw, int32_t* mp_of_key, PoseProblem* probs, BaEdge* edges, double* Xw,
                             int32_t* edge_kp, double* poses, hipStream_t st);
void launch_track_finish_last(const TrackFrameDev* frames, int n_frames, int capacity, const int32_t* n_matches, const PoseProblem* probs, const uint8_t* outlier,
                              const int32_t* edge_kp, const int32_t* inliers, int32_t* mp_of_key, double* poses, int32_t* n_inliers, hipStream_t st);
// TrackLocalMap
void launch_track_edges_local(const TrackFrameDev* frames, int n_frames, const TrackConst& C, const MatchKey* keys, const float* u_right, const int32_t* match,
                              const uint8_t* held, const float* held_Xw, const LocalPointDev* points, int32_t* local_of_key, PoseProblem* probs, BaEdge* edges,
                              double* Xw, int32_t* edge_kp, double* poses, hipStream_t st);
void launch_track_finish_local(const TrackFrameDev* frames, int n_frames, int capacity, const PoseProblem* probs, const uint8_t* outlier, const int32_t* edge_kp,
                               const uint8_t* held, const int32_t* local_of_key, uint8_t* outlier_of_key, int32_t* n_inliers, hipStream_t st);

}  // namespace tc2li
