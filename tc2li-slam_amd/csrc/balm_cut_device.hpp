// Shared between the host orchestration and the kernels of the plane extraction of a LiDAR window (balm_cut_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "balm_math.hpp"

#include "global_ptr.hpp"

namespace tc2li {

constexpr int kBalmCutMaxW = 7;           // keyframes of a window the batched LiDAR kernels take (ba_lockstep.cpp, ba_engine.cpp); 8 lanes work on a cell
constexpr int kBalmCutMaxPoints = 65535;  // points of a window: the sort's 16-bit counters
constexpr int kBalmCutMaxPlanes = 2048;   // planes of a window the batched LiDAR kernels take

// One window's extraction: inputs, work space (carved from one allocation by BalmTerm), outputs.  `n` = points of the window.
struct BalmCutTask {
    int32_t W, n_points, table_bits, pad_;
    int32_t cloud_off[kBalmCutMaxW + 1];
    struct Rel { double R[9], p[3]; } rel[kBalmCutMaxW];  // keyframe i's LiDAR frame -> keyframe 0's
    const float* cloud;              // [n][3] the keyframes' surface points, each in its own LiDAR frame
    // work space
    unsigned long long* table_key;   // [2^table_bits] root voxel keys, 0 = empty
    int32_t* table_first;            // [2^table_bits] smallest point index of the key
    int32_t* table_id;               // [2^table_bits] the root's number (order of first appearance)
    double* world;                   // [n][3] the points in the common frame
    unsigned char* oct;              // [n] octant in the root << 3 | octant in that octant
    int32_t* point_slot;             // [n] table slot of the point's root
    unsigned int *sort_key_a, *sort_key_b;  // [n]
    int32_t *sort_val_a, *sort_val_b;       // [n]
    int32_t* order;                  // [3][n] the point list in the order of layer 0 / 1 / 2
    int32_t* cell_begin;             // [3][n + 1] first place of every cell in that order, and the end
    unsigned int* cell_key;          // [3][n] root << 3 layer | octants
    unsigned char* cell_flag;        // [3][n] 0: <= 15 points; 1: not planar; 2: a plane; 3: planar but seen from one keyframe
    int32_t* n_cells;                // [3]
    int32_t* plane_cell;             // [kBalmCutMaxPlanes] layer << 28 | cell, in the host walk's order
    int32_t* state;                  // [4] device: planes, "this window goes to the host", roots
    int32_t* result_host;            // [4] pinned: the same once the walk has run
    // outputs: BalmDev::clusters / coe
    PlaneCluster* clusters;          // [n_planes][W]
    double* coe;                     // [n_planes]
};

#if defined(__HIPCC__)
// a kernel's private copy of the record with its pointers marked global (global_ptr.hpp); only the fields a kernel uses are loaded
__device__ __forceinline__ BalmCutTask global_record(BalmCutTask t) {
    TC2LI_GLOBAL_FIELD(t, cloud); TC2LI_GLOBAL_FIELD(t, table_key); TC2LI_GLOBAL_FIELD(t, table_first); TC2LI_GLOBAL_FIELD(t, table_id);
    TC2LI_GLOBAL_FIELD(t, world); TC2LI_GLOBAL_FIELD(t, oct); TC2LI_GLOBAL_FIELD(t, point_slot); TC2LI_GLOBAL_FIELD(t, sort_key_a);
    TC2LI_GLOBAL_FIELD(t, sort_key_b); TC2LI_GLOBAL_FIELD(t, sort_val_a); TC2LI_GLOBAL_FIELD(t, sort_val_b); TC2LI_GLOBAL_FIELD(t, order);
    TC2LI_GLOBAL_FIELD(t, cell_begin); TC2LI_GLOBAL_FIELD(t, cell_key); TC2LI_GLOBAL_FIELD(t, cell_flag); TC2LI_GLOBAL_FIELD(t, n_cells);
    TC2LI_GLOBAL_FIELD(t, plane_cell); TC2LI_GLOBAL_FIELD(t, state); TC2LI_GLOBAL_FIELD(t, result_host); TC2LI_GLOBAL_FIELD(t, clusters);
    TC2LI_GLOBAL_FIELD(t, coe);
    return t;
}
#endif

// queues the extraction of `n_tasks` windows (tasks in device memory; max_points / max_table over them)
void launch_balm_cut(const BalmCutTask* tasks, int n_tasks, int max_points, int max_table, hipStream_t st);

}  // namespace tc2li
