// Shared between the host entry and the kernels of the local-map bookkeeping (Tracking::UpdateLocalKeyFrames / UpdateLocalPoints).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tc2li {

// device mirror of tc2li_map_graph plus the per-update scratch
struct LocalMapDev {
    int32_t n_keyframes, n_points;
    const uint8_t* kf_bad;
    const int32_t *covis_off, *covis, *child_off, *children, *parent, *prev_kf, *match_off, *matches;
    const uint8_t* point_bad;
    const int32_t *obs_off, *obs_kf;
    // scratch
    int32_t* votes;        // [n_keyframes]
    uint8_t* marked;       // [n_keyframes] mnTrackReferenceForFrame == current frame
    int32_t* kf_list;      // [n_keyframes] mvpLocalKeyFrames
    int32_t* rev_base;     // [n_keyframes + 1] start of every local keyframe's matches in the reversed, concatenated list
    int32_t* header;       // [4]: n_local_keyframes, total entries, reference keyframe, n_local_points
    int32_t* first_pos;    // [n_points] first position of a point in the concatenated list
    int32_t* block_counts; // kept entries per 256-entry block
    int32_t* points;       // [n_points] mvpLocalMapPoints
};

void launch_local_map_votes(const LocalMapDev& m, const int32_t* frame_points, int n_frame_points, uint8_t* cleared, hipStream_t st);
void launch_local_map_keyframes(const LocalMapDev& m, int temporal_last_kf, hipStream_t st);
void launch_local_map_points(const LocalMapDev& m, int n_local, int total, hipStream_t st);

}  // namespace tc2li
