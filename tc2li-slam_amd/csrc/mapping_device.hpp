// Shared between the host entry points and the kernels of the CreateNewMapPoints core (mapping_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tc2li {

struct KfDev {  // one keyframe resident on the device
    int32_t n, n_nodes;
    const float* keys;          // tc2li_keypoint records (6 x 4 bytes)
    const uint8_t* desc;
    const float *u_right, *depth;
    const uint8_t* has_point;
    const int32_t *fv_node, *fv_off, *fv_idx;
    float q[4], t[3];           // Tcw
    float Rcw[9], Ow[3];        // rotation matrix and camera centre (host-computed in the reference's float arithmetic)
    float F12[9], ep[2];        // neighbour only: fundamental matrix and epipole with respect to the current keyframe
    int32_t skip;               // neighbour only: baseline < mb
};
struct MappingDev {
    KfDev cur;
    const KfDev* neigh;
    int32_t n_neigh, n_levels;
    float fx, fy, cx, cy, mb, mbf, ratio_factor, th_far;
    int32_t inertial, far_points, only_stereo, coarse;
    const float *scale_factors, *level_sigma2;
    int32_t* match;             // [n_neigh][cur.n]
    uint8_t* ok;                // [n_neigh][cur.n]: 1 = a point passes every gate, bit 1 = stereo point
    float* x3D;                 // [n_neigh][cur.n][3]
};
struct FuseDev {  // ORBmatcher::Fuse, search part
    const float* keys;            // tc2li_keypoint records
    const uint8_t* desc;
    const float* u_right;
    const int32_t* cell_start;    // 64 x 48 grid of the keyframe (k_match_grid): [kCells + 1]
    const uint16_t* items;
    int32_t n_keys, n_points, n_levels, pad_;
    float q[4], t[3], Ow[3];
    float fx, fy, cx, cy, bf, th, log_scale_factor;
    float min_x, max_x, min_y, max_y;
    const float *scale_factors, *inv_level_sigma2;
    const uint8_t* points;        // tc2li_map_point records (68 bytes)
    const uint8_t* valid;
    int32_t *best_idx, *best_dist;
};
void launch_fuse_search(const FuseDev& f, hipStream_t st);
void launch_tri_search(const MappingDev& m, int max_entries, hipStream_t st);
void launch_tri_points(const MappingDev& m, hipStream_t st);

}  // namespace tc2li
