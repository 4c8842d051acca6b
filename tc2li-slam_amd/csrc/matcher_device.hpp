// Shared between the host orchestration and the projection-matching kernel.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "orb_device.hpp"  // MatchKey

namespace tc2li {

constexpr int kMaxMatchKeys = 3072;  // keypoints of one frame that fit the LDS working set

struct MatchQuery {  // tc2li_proj_query, 64 bytes
    float u, v, radius, u_right;
    int32_t min_level, max_level;
    float angle;
    int16_t valid, has_observations;
    uint8_t desc[32];
};
static_assert(sizeof(MatchQuery) == 64, "layout");

struct MatchFrameDev {
    const MatchKey* keys;
    const uint8_t* desc;
    const float* u_right;
    const uint8_t* occupied;
    const MatchQuery* queries;
    int32_t n_keys, n_queries, query_off, pad_;
    float min_x, max_x, min_y, max_y;
};

void launch_match_by_projection(const MatchFrameDev* frames, int nframes, int mode, float nn_ratio, int32_t* match_of_query,
                                int32_t* prev_claim, int32_t* rounds_out, hipStream_t st);

}  // namespace tc2li
