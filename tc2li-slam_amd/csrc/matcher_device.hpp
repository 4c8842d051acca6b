// Shared between the host orchestration and the projection-matching kernel.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "global_ptr.hpp"
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

#if defined(__HIPCC__)
__device__ __forceinline__ MatchFrameDev global_record(MatchFrameDev f) {  // global_ptr.hpp
    TC2LI_GLOBAL_FIELD(f, keys); TC2LI_GLOBAL_FIELD(f, desc); TC2LI_GLOBAL_FIELD(f, u_right); TC2LI_GLOBAL_FIELD(f, occupied); TC2LI_GLOBAL_FIELD(f, queries);
    return f;
}
#endif

// the three-launch form (grid, candidate lists, rounds over the lists); the tables live in the caller's workspace.  When
// pool_top[1] comes back non-zero the candidate pool was too small and launch_match_by_projection has to be used instead.
struct MatchLists {
    int32_t* cell_start;   // [nframes][kCells + 1]
    uint16_t* items;       // [total keys], frame f at key_base[f]
    const int32_t* key_base;  // [nframes]
    int32_t* cand_off;     // [total queries]
    int32_t* cand_cnt;     // [total queries], bit 30 = has_observations
    uint32_t* pool;
    int32_t* pool_top;     // [0] = entries used, [1] = overflow flag
    int32_t pool_cap, pad_;
};
// only the feature grids (cell_start, items) of the frames: used by the fuse search as well
void launch_match_grid(const MatchFrameDev* frames, int nframes, const MatchLists& L, hipStream_t st);
void launch_match_lists(const MatchFrameDev* frames, int nframes, const int32_t* query_frame, int total_q, const MatchLists& L, int mode,
                        float nn_ratio, int32_t* match_of_query, int32_t* prev_claim, int32_t* rounds_out, hipStream_t st);
void launch_match_by_projection(const MatchFrameDev* frames, int nframes, int mode, float nn_ratio, int32_t* match_of_query,
                                int32_t* prev_claim, int32_t* rounds_out, hipStream_t st);

}  // namespace tc2li
