// Shared between the host entry and the map-point refresh kernel.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tc2li {

constexpr int kMaxObservations = 112;  // observations of one point the LDS table holds (112^2 x 2 B + 112 x 32 B = 28 KB)

struct MapPointRefresh {
    const int32_t* obs_off;        // [n_points + 1]
    const uint8_t* descriptors;    // [total][32]
    const float* centres;          // [total][3]
    const float* positions;        // [n_points][3]
    const float* ref_centres;      // [n_points][3]
    const float* level_scale;      // [n_points]
    float last_scale;
    int32_t pad_;
    int32_t* best_obs;
    float *normals, *min_dist, *max_dist;
};
void launch_map_points_refresh(const MapPointRefresh& a, int n_points, hipStream_t st);

}  // namespace tc2li
