// Structures shared by the host orchestration and the gfx950 kernels of the LiDAR front end.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "global_ptr.hpp"

namespace tc2li {

struct VelodynePoint {  // velodyne_ros::Point (SF/include/lidar_front_end/preprocess.h:62-70), 32 bytes
    float x, y, z, pad0;
    float intensity, time;
    uint16_t ring, pad1;
    float pad2;
};
struct PointXYZINormal {  // pcl::PointXYZINormal, 48 bytes
    float x, y, z, pad0;
    float normal_x, normal_y, normal_z, pad1;
    float intensity, curvature, pad2, pad3;
};
static_assert(sizeof(VelodynePoint) == 32 && sizeof(PointXYZINormal) == 48, "PCL layouts");

constexpr int kSegBlock = 1024;  // points per block of the segmented (per-scan) passes

// A block of a segmented pass: which scan it belongs to and where it starts inside that scan's slot.
struct SegBlock { int32_t scan, start; };

// Per-scan slots: every per-point array reserves `cap` entries starting at `base` for a scan; how many are in use
// at each stage lives in device counters so that stages chain without host synchronisation.
// raw_base: where the scan's raw points start in the caller's array (packed scans are read in place).
struct ScanSlot { int32_t base, cap, first_block, n_blocks, raw_base, pad_[3]; };

struct VoxelParams {  // pcl::VoxelGrid::applyFilter bookkeeping for one scan
    int32_t min_b[3], mul[3];
    int32_t passthrough;  // index space overflow: the filter returns its input unchanged
    int32_t table_base, table_mask;
};

struct LidarStateDev { double rot[9], pos[3], off_r[9], off_t[3]; };

struct PreprocessParams { int32_t point_filter_num; float time_unit_scale; double blind_sq; };

// Dense uniform grid over the map's bounding box (replaces the ikd-Tree as the spatial index; same 5 nearest
// neighbours).  Cell (ix, iy, iz) relative to (x0, y0, z0) has linear index (iz * ny + iy) * nx + ix.
// Layout since round 4 (in-place insertion, map_kernels.hip "k_map_ins_*"): the cell-sorted copy is cut into SEGMENTS of kMapSegCells
// consecutive cells of one row (iy, iz) -- segment (row, sx) covers the cells ix = 16 sx .. 16 sx + 15 -- whose entries are contiguous
// and ordered by cell; segments follow each other with SLACK between them (a build leaves every segment room for as many entries again
// + seg_slack), so points are added by rewriting the few segments they fall into, whatever the shape of the map (a street's map has
// rows of thousands of entries).  bucket_start has kMapSegStride = 17 entries per segment: [seg * 17 + j] = first entry of the
// segment's cell j, [seg * 17 + 16] = end of the segment's entries; the next segment's first entry ([(seg + 1) * 17]; n_slots behind
// the last segment) is where the segment's room ends.  A run of cells along x is one range per segment it touches.  An entry whose
// index (w) is negative is a TOMBSTONE (a deleted point's place, or unused room): every reader skips it.
constexpr int kMapSegCells = 16, kMapSegStride = kMapSegCells + 1;
struct MapGrid {
    const PointXYZINormal* points;  // the map points in insertion order (what Nearest_Search returns copies of)
    const float4* pts;              // [n_slots] row-wise cell-sorted xyz + original index (as int bits in w; < 0: tombstone)
    const int32_t* bucket_start;    // [ny * nz * nsx * kMapSegStride + 1]
    int32_t x0, y0, z0, nx, ny, nz;
    int32_t n_points;
    float inv_cell, cell;
    int32_t nsx, n_slots;           // segments per row = ceil(nx / kMapSegCells); entries of pts
};

void launch_pre_count(const VelodynePoint* raw, const int* raw_count, const ScanSlot* slots, const SegBlock* blocks,
                      int nblocks, PreprocessParams prm, int* block_counts, hipStream_t st);
void launch_seg_scan(const ScanSlot* slots, int nscans, const int* block_counts, int* block_offsets, int* totals, hipStream_t st);
// the three launches above as one pass per scan (a workgroup per scan: batches of many scans); bbox_enc [6 per scan] = the bounding box of
// the kept finite points as k_voxel_bbox leaves it
void launch_pre_stream(const VelodynePoint* raw, const int* raw_count, const ScanSlot* slots, int nscans, PreprocessParams prm, PointXYZINormal* out,
                       int* out_count, int* bbox_enc, float* time_out /* NULL, or where the kept points' time stamps go too */,
                       int* vkey_out /* NULL, or the packed voxel coordinates at `leaf` */, float leaf, int* vk_ok /* [nscans] */, hipStream_t st);
void launch_pre_scatter(const VelodynePoint* raw, const int* raw_count, const ScanSlot* slots, const SegBlock* blocks,
                        int nblocks, PreprocessParams prm, const int* block_offsets, PointXYZINormal* out, hipStream_t st);

void launch_voxel_bbox(const PointXYZINormal* pts, const int* count, const ScanSlot* slots, const SegBlock* blocks, int nblocks,
                       int* bbox_enc, hipStream_t st);
void launch_voxel_params(const int* bbox_enc, const int* count, const ScanSlot* slots, int nscans, float leaf, VoxelParams* vp,
                         hipStream_t st);
void launch_voxel_insert(const PointXYZINormal* pts, const int* count, const ScanSlot* slots, const SegBlock* blocks, int nblocks,
                         float leaf, const VoxelParams* vp, int* table_keys, int* table_counts, int* pt_slot, int* n_vox, int* vox_keys, hipStream_t st);
void launch_fill_int(int* p, size_t n, int v, hipStream_t st);
void launch_voxel_sort(const ScanSlot* slots, int nscans, const VoxelParams* vp, const int* count, const int* table_keys, const int* table_counts,
                       int* table_rank, int* vox_keys, int* vox_member_off, int* vox_fill /* cleared per voxel */, int* vox_count /* points per voxel */,
                       int* n_vox, int* status, hipStream_t st);
void launch_voxel_fill(const int* count, const ScanSlot* slots, const SegBlock* blocks, int nblocks, const VoxelParams* vp, const int* pt_slot,
                       const int* table_rank, const int* vox_member_off, int* vox_fill /* runs per voxel */,
                       int* members /* one word per run: first index | (length - 1) << 24 */, hipStream_t st);
// the sorted form of the voxel filter (after launch_voxel_bbox / launch_voxel_params): one workgroup per scan sorts (voxel, point), then the centroids
void launch_voxel_sort_points(const PointXYZINormal* pts, const int* count, const ScanSlot* slots, int nscans, float leaf, const VoxelParams* vp, int* key_a,
                              int* idx_a, int* key_b, int* idx_b, int* vox_start, int* vox_info, int* n_vox, const int* vk_ok /* NULL, or per scan: key_b holds k_pre_stream's packed voxel coordinates */, hipStream_t st);
void launch_voxel_sums(const PointXYZINormal* pts, const ScanSlot* slots, const SegBlock* blocks, int nblocks, const ScanSlot* vslots, const SegBlock* vblocks,
                       int nvblocks, const VoxelParams* vp, const int* idx_a, const int* idx_b, const int* vox_start, const int* vox_info, const int* n_vox,
                       void* recs, PointXYZINormal* out, int* out_count, hipStream_t st);
void launch_voxel_centroid(const PointXYZINormal* pts, const int* count, const ScanSlot* slots, const SegBlock* blocks, int nblocks,
                           float leaf, const VoxelParams* vp, const int* pt_slot, const int* table_rank, const int* n_vox,
                           const int* vox_member_off, const int* vox_fill, const int* vox_count, const int* members, void* recs /* 32 B per point */,
                           PointXYZINormal* out, int* out_count, hipStream_t st);

// ---- map maintenance (map_incremental / Add_Points with down-sampling / Delete_Point_Boxes), batched over maps ----
struct MapIncRec { unsigned long long key; int idx; int pad; };
constexpr int kMapIncMax = 8192;
constexpr int kMapIncOut = 16;  // per task: [0] n_add [1] n_groups [2] n_noneed [3] overflow [4] kept K [5] appended [6..11] bbox of the added points [12] kept among the first K [13] grid inconsistency found by the compaction [14] listed deletions (lean tasks) [15] more than kMapDelMax of them
// One (scan, map) pair of a batched map_incremental / compaction (map_kernels.hip).  has_inc = 0: compaction only (box deletion).
struct MapIncTask {
    // scan side: slots of the tc2li_lidar workspace
    const PointXYZINormal* body;   // [n] down-sampled points of the scan
    const int* nearest_idx;        // [n][5]
    const int* nfound;             // [n]
    PointXYZINormal* world;        // [n]
    uint8_t* cls;                  // [n]
    MapIncRec* recs;               // [kMapIncMax]
    int* group_start;              // [kMapIncMax + 1]
    int* noneed;                   // [n]
    PointXYZINormal* appended;     // [kMapIncMax]
    uint8_t* has_append;           // [kMapIncMax]
    int* out;                      // [kMapIncOut]
    // map side
    MapGrid grid;                  // the map before the update (grid.points = the points in insertion order)
    uint8_t* deleted;              // [n_map], all zero between calls
    int* keep_counts;              // [keep_blocks]
    PointXYZINormal* dst;          // the map's point buffer itself (compaction in place): [0, K) kept, then appended + no-need points
    int* holes;                    // [n_map] deleted places below K, in index order | [n_map] kept points at or above K, in index order
    int* batch_overflow;           // one word per batch: set when any task's insertion list overflowed -- then no map is touched
    int* remap;                    // [n_map] new index of every old point, -1: deleted (the grid rebuild walks the OLD cell order with it)
    LidarStateDev st;
    double fs;                     // filter_size_map_min
    float ds;                      // ikdtree downsample_size
    int n, n_map, keep_blocks, ekf_inited, has_inc;
    const float* boxes;            // [n_boxes][6] (min, max) of KD_TREE::Delete_Point_Boxes; has_inc = 0 tasks only
    int n_boxes;
    int fix_grid;                  // the grid is maintained in place: deleted points' entries become tombstones (k_mapinc_apply), moved points' entries get the new index (k_map_fill)
    int lean;                      // round 5: the deletions are LISTED (k_mapinc_apply appends to `holes`, count in out[14]) and the compaction works from
                                   // that list (k_map_compact_list: O(deletions)) instead of passing over the map's flags three times (O(map));
                                   // tasks with fix_grid and without boxes
};
constexpr int kMapDelMax = 8192;   // listed deletions per map and step (more: out[15], and the host repeats the map's compaction through the flag passes)
// Counting sort of one map's points into its dense grid.
struct MapGridTask {
    MapGrid g;         // geometry + points of the map (g.pts / g.bucket_start = what the build writes)
    int* counts;       // [n_cells], zero outside a build
    int* start;        // [n_cells + 1] exclusive prefix of the counts (work array of the build)
    int* row_start;    // = g.bucket_start, writable: [segments * kMapSegStride + 1]
    float4* sorted;    // = g.pts, writable: [g.n_slots], all tombstones when the build starts
    int* tile_sums;    // [ceil(n_cells / 4096)]
    int n_cells, seg_slack;
    // Points that were in the map before the last compaction are visited in the order of the OLD grid (cell-coherent: neighbouring lanes
    // hit the same or neighbouring counters, one atomic per run of equal cells); what was added since, in insertion order.
    const float4* old_sorted;  // [n_old] the old grid's entries (xyz + old index; tombstones skipped), NULL: no old grid (first build)
    const int* remap;          // old index -> new index, -1: deleted
    int n_old, n_kept;         // n_old = the old grid's n_slots; g.points[0 .. n_kept) are the kept old points, [n_kept .. g.n_points) the added ones
};
// In-place insertion of the points [first, first + count) of a map into its grid (k_map_ins_sort + k_map_ins_rows).
constexpr int kMapInsMax = 8192;   // added points per map and step the in-place path takes (more: the grid is rebuilt)
constexpr int kMapRowMax = 512;    // entries of one segment (old + new) the merge holds in LDS (more: rebuilt): 16 cells x 32
struct MapInsTask {
    MapGrid g;
    float4* pts;                   // = g.pts, writable
    int* row_start;                // = g.bucket_start, writable
    unsigned long long* keys;      // [kMapInsMax] ((segment * 16 + cell in the segment) << 32 | point index), sorted
    int* row_list;                 // [kMapInsMax + 1] first key of every segment that receives points
    int* out;                      // [4]: [0] rows that receive points [1] 1: the grid could not take the points (outside the box, a row without room, ...) [2] tombstones the rewritten rows dropped
    int first, count;
};
#if defined(__HIPCC__)
// a kernel's private copy of a record with its pointers marked global (global_ptr.hpp); only the fields a kernel uses are loaded
__device__ __forceinline__ MapGrid global_record(MapGrid g) {
    TC2LI_GLOBAL_FIELD(g, points); TC2LI_GLOBAL_FIELD(g, pts); TC2LI_GLOBAL_FIELD(g, bucket_start);
    return g;
}
__device__ __forceinline__ MapIncTask global_record(MapIncTask t) {
    TC2LI_GLOBAL_FIELD(t, body); TC2LI_GLOBAL_FIELD(t, nearest_idx); TC2LI_GLOBAL_FIELD(t, nfound); TC2LI_GLOBAL_FIELD(t, world); TC2LI_GLOBAL_FIELD(t, cls);
    TC2LI_GLOBAL_FIELD(t, recs); TC2LI_GLOBAL_FIELD(t, group_start); TC2LI_GLOBAL_FIELD(t, noneed); TC2LI_GLOBAL_FIELD(t, appended);
    TC2LI_GLOBAL_FIELD(t, has_append); TC2LI_GLOBAL_FIELD(t, out); t.grid = global_record(t.grid); TC2LI_GLOBAL_FIELD(t, deleted);
    TC2LI_GLOBAL_FIELD(t, keep_counts); TC2LI_GLOBAL_FIELD(t, dst); TC2LI_GLOBAL_FIELD(t, holes); TC2LI_GLOBAL_FIELD(t, batch_overflow);
    TC2LI_GLOBAL_FIELD(t, remap); TC2LI_GLOBAL_FIELD(t, boxes);
    return t;
}
__device__ __forceinline__ MapGridTask global_record(MapGridTask t) {
    t.g = global_record(t.g); TC2LI_GLOBAL_FIELD(t, counts); TC2LI_GLOBAL_FIELD(t, start); TC2LI_GLOBAL_FIELD(t, row_start); TC2LI_GLOBAL_FIELD(t, sorted);
    TC2LI_GLOBAL_FIELD(t, tile_sums); TC2LI_GLOBAL_FIELD(t, old_sorted); TC2LI_GLOBAL_FIELD(t, remap);
    return t;
}
__device__ __forceinline__ MapInsTask global_record(MapInsTask t) {
    t.g = global_record(t.g); TC2LI_GLOBAL_FIELD(t, pts); TC2LI_GLOBAL_FIELD(t, row_start); TC2LI_GLOBAL_FIELD(t, keys); TC2LI_GLOBAL_FIELD(t, row_list);
    TC2LI_GLOBAL_FIELD(t, out);
    return t;
}
#endif
void launch_map_insert(const MapInsTask* tasks, int n_tasks, hipStream_t st);
void launch_mapinc_lists(const MapIncTask* tasks, int n_tasks, int max_points, hipStream_t st);  // classify, group, apply
void launch_map_mark_boxes(const MapIncTask* tasks, int n_tasks, int max_map_points, hipStream_t st);
void launch_map_compact(const MapIncTask* tasks, int n_tasks, int max_map_points, bool any_lean, bool any_flagged, hipStream_t st);
void launch_map_grid_build(const MapGridTask* tasks, int n_tasks, int max_work /* max over tasks of n_old + added */, int max_cells,
                           int max_row_entries /* max over tasks of segments * kMapSegStride + 1 */, hipStream_t st);

// LidarFrontEndTools::transformPointCloud (SF/src/LidarTypes.cc:42-65) for one cloud: out[i] = (R in[i] + t, intensity kept, the rest as a
// default-constructed point)
struct TransformTask { const PointXYZINormal* in; PointXYZINormal* out; float R[9], t[3]; int32_t n, pad_; };
void launch_transform_points(const TransformTask* tasks, int n_tasks, int max_points, hipStream_t st);
void fill_transform_task(TransformTask& t, const float T7[7]);  // lidar_pose_host.cpp

struct Pose6DDev { double offset_time, acc[3], gyr[3], vel[3], pos[3], rot[9]; };
constexpr int kMaxImuPoses = 64;
// points[i] = in[perm[i]] compensated into the scan-end frame (ImuProcess::UndistortPcl)
void launch_undistort(const PointXYZINormal* in, const int* perm, int n, const Pose6DDev* poses, int n_poses, const LidarStateDev* end,
                      PointXYZINormal* out, hipStream_t st);

// the same for every scan of a batch, the time sort included (k_time_sort: std::sort's permutation replayed on the device):
// key [total] floats, ints5 [5 total] / ints3 [3 total] ints, flag [total] bytes of work space in the scans' slots; fallback [n_scans] = 1
// where the recursion reached std::sort's depth limit (the host sorts such a scan); depth_override >= 0 replaces that limit (tests)
void launch_time_sort(const PointXYZINormal* pts, const int* count, const ScanSlot* slots, int n_scans, float* key, int* ints5, int* ints3, uint8_t* flag,
                      size_t total, int* perm, int* fallback, int* ranges /* [total] */, int* n_ranges /* [n_scans] */, int depth_override, bool keys_ready /* key[] holds the time stamps */,
                      hipStream_t st);
void launch_undistort_batch(const PointXYZINormal* in, const int* perm, const int* count, const ScanSlot* slots, const SegBlock* blocks, int nblocks,
                            const Pose6DDev* poses /* [n_scans][kMaxImuPoses] */, const int* n_poses, const LidarStateDev* ends, PointXYZINormal* out,
                            hipStream_t st);

void launch_knn_plane(const MapGrid* grids, const PointXYZINormal* body, const int* count,
                      const ScanSlot* slots, const SegBlock* blocks, int nblocks, const LidarStateDev* states,
                      PointXYZINormal* world, uint8_t* selected, PointXYZINormal* normvec, int* nearest_idx, float* nearest_d,
                      int* nfound, int* hard_count, int2* hard_list, hipStream_t st, hipEvent_t after_first = nullptr);
// iterated ESKF (row b7): re-evaluation of the kept neighbours at a new state; normal equations of the measurement rows ->
// out[158] = H^T H (12 x 12), H^T h (12), sum |pd2|, number of rows (partial: [(n + 255) / 256][158])
constexpr int kEskfOutSize = 144 + 12 + 2;
void launch_eskf_refit(const MapGrid& grid, const PointXYZINormal* body, int n, const LidarStateDev* state, const int* nearest_idx,
                       PointXYZINormal* world, uint8_t* selected, PointXYZINormal* normvec, hipStream_t st);
void launch_eskf_normal(const PointXYZINormal* body, int n, const LidarStateDev* state, const uint8_t* selected, const PointXYZINormal* normvec,
                        int extrinsic_est_en, double* partial, double* out, hipStream_t st);
// the batch forms: `blocks` lists the (scan, 1024-point block) pairs taking part, `scans` the scans whose sums are wanted;
// partial [total / 256][kEskfOutSize], out [max_scans][kEskfOutSize] (pinned), a scan's result at its slot number
void launch_eskf_refit_batch(const MapGrid* grids, const PointXYZINormal* body, const int* count, const ScanSlot* slots, const SegBlock* blocks, int nblocks,
                             const LidarStateDev* states, const int* nearest_idx, PointXYZINormal* world, uint8_t* selected, PointXYZINormal* normvec,
                             hipStream_t st);
void launch_eskf_normal_batch(const PointXYZINormal* body, const int* count, const ScanSlot* slots, const SegBlock* blocks, int nblocks,
                              const LidarStateDev* states, const uint8_t* selected, const PointXYZINormal* normvec, int extrinsic_est_en, double* partial,
                              const int* scans, int n_scans, double* out, hipStream_t st);
void launch_sel_count(const uint8_t* selected, const int* count, const ScanSlot* slots, const SegBlock* blocks, int nblocks,
                      int* block_counts, hipStream_t st);
void launch_sel_scatter(const uint8_t* selected, const int* count, const ScanSlot* slots, const SegBlock* blocks, int nblocks,
                        const int* block_offsets, const PointXYZINormal* body, const PointXYZINormal* normvec,
                        PointXYZINormal* cloud_ori, PointXYZINormal* corr_norm, hipStream_t st);

}  // namespace tc2li
