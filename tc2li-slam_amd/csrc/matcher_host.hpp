// Internal interface of the projection matcher used by the batched tracking entry (tracking_host.cpp).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/tc2li_hip.h"

namespace tc2li {

struct BatchSearchFrame {
    int key_off, n_keys;               // range of the frame's keypoints in the extractor's device feature arrays
    const tc2li_keypoint* keys_host;   // the same keypoints on the host (angles for the rotation histogram)
    const float* u_right_host;         // mvuRight
    int q_off, n_q;                    // range of the frame's queries
    const uint8_t* occupied_host = nullptr;  // mvpMapPoints[i] && Observations() > 0 before the search (NULL: none)
};

// ORBmatcher::SearchByProjection loop bodies for many frames in one launch; match_of_query is indexed like `queries`.
int search_batch_device(tc2li_orb* o, const BatchSearchFrame* frames, int n_frames, const tc2li_proj_query* queries, int mode,
                        float nn_ratio, bool check_orientation, int32_t* match_of_query, int32_t* n_matches, hipStream_t st);

}  // namespace tc2li
