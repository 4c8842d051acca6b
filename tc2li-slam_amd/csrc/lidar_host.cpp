// Host orchestration of the gfx950 LiDAR front end behind tc2li_lidar_* (include/tc2li_hip.h).  Mirrors the
// camera-LiDAR branch of SF/include/lidar_front_end: Preprocess::process -> pcl::VoxelGrid -> feature_extraction
// against the incremental map (LidarFrontEnd.cpp:233-261, 886-962, 999-1073).  Every stage works on a batch of
// scans laid out in fixed per-scan slots; the per-scan point counts produced by one stage are consumed by the next
// one from device memory.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>

#include "common.hpp"
#include "eskf_math.hpp"
#include "lidar_device.hpp"

using namespace tc2li;

static_assert(sizeof(tc2li_velodyne_point) == sizeof(VelodynePoint), "ABI layout");
static_assert(sizeof(tc2li_point) == sizeof(PointXYZINormal), "ABI layout");
static_assert(sizeof(tc2li_lidar_state) == sizeof(LidarStateDev), "ABI layout");

struct tc2li_lidar_map {
    DevBuf<PointXYZINormal> d_points;
    DevBuf<int> d_holes;  // [2 n] work lists of the in-place compaction
    DevBuf<uint8_t> d_deleted;  // all zero between calls (k_map_keep_scatter clears what the marking kernels set)
    DevBuf<int> d_keep_counts, d_out;  // d_out [kMapIncOut]: a compaction without a scan (box deletion)
    DevBuf<float> d_boxes;
    PinnedBuf<int> h_out;
    DevBuf<float4> d_sorted, d_sorted_alt;  // the grid's cell-sorted copy; a rebuild after a compaction reads the old one while it writes the other
    DevBuf<int> d_remap;                    // old index -> new index of the last compaction (-1: deleted)
    bool have_remap = false;                // d_remap / d_sorted describe the map before the compaction that was just committed
    int remap_n_old = 0, remap_n_kept = 0;
    DevBuf<int> d_bucket_counts, d_bucket_start, d_tile_sums;  // d_bucket_start: the plain prefix over the cells (work array of a build)
    DevBuf<int> d_row_start;          // the grid's row-wise starts (MapGrid::bucket_start)
    DevBuf<MapIncTask> d_inc_task;    // batches of one map: Build / Add_Points / Delete_Point_Boxes
    DevBuf<MapGridTask> d_grid_task;
    // in-place insertion (k_map_ins_*): work arrays; grid_valid: `grid` describes the points as they are numbered now
    DevBuf<unsigned long long> d_ins_keys;
    DevBuf<int> d_ins_rows;
    bool grid_valid = false;
    bool counts_dirty = false;        // a build was queued and has not been seen to complete: the cell counters may not be zero (grid_prepare clears them)
    // how the next build finds the points that were there before: 0 from the point list (all cells' counters contended), 1 in the old
    // grid's order through d_remap (after a compaction), 2 in the old grid's order as it stands (its entries already carry the new indices)
    int rebuild_mode = 0;
    int tombstones = 0;               // entries of the grid that mark deleted points (an upper bound: a rewritten row drops its own)
    int n_grid_builds = 0, n_grid_updates = 0;  // tc2li_lidar_map_stats
    int n = 0, n_cells = 0;
    // bumped whenever the points are renumbered or replaced (Build / Add_Points / compaction): tc2li_lidar_map_incremental replays neighbour
    // INDICES found by an earlier feature extraction and refuses when the map has changed in between
    uint64_t generation = 0;
    float cell = 1.0f;
    float lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};  // bounding box of the points
    MapGrid grid{};
    // Thread contract (include/tc2li_hip.h, "LiDAR map handle"): every entry point that reads or changes the map holds this lock from
    // its first access until its device work on the map has completed, so calls from the tracking thread (UpdateMap -> map_incremental,
    // Tracking.cc:1602-1603) and the LiDAR thread (feature_extraction) on one handle serialise inside the library.
    mutable std::mutex mu;
};

namespace {
// Locks the distinct maps of a batch in address order (two batches that share maps cannot deadlock).
struct MapLocks {
    std::vector<std::mutex*> mus;
    MapLocks(tc2li_lidar_map* const* maps, int n) {
        for (int i = 0; i < n; ++i) if (maps[i]) mus.push_back(&maps[i]->mu);
        std::sort(mus.begin(), mus.end());
        mus.erase(std::unique(mus.begin(), mus.end()), mus.end());
        for (auto* m : mus) m->lock();
    }
    ~MapLocks() { for (auto it = mus.rbegin(); it != mus.rend(); ++it) (*it)->unlock(); }
    MapLocks(const MapLocks&) = delete;
    MapLocks& operator=(const MapLocks&) = delete;
};
}  // namespace

struct tc2li_lidar {
    int max_scans = 0, cap = 0;  // scans per call, points per scan slot
    size_t total = 0;
    int table_size = 0;          // hash-table entries per scan (power of two)
    DevBuf<VelodynePoint> d_raw;
    DevBuf<PointXYZINormal> d_pre, d_down, d_world, d_normvec, d_cloud_ori, d_corr;
    DevBuf<uint8_t> d_selected;
    DevBuf<int> d_nearest_idx, d_nfound;
    DevBuf<float> d_nearest_d;
    DevBuf<int> d_vk_ok;           // per scan: k_pre_stream's packed voxel coordinates are usable (every kept point fits)
    float pre_vkey_leaf = 0.f;     // > 0: d_members holds them for this leaf (between run_preprocess and the run_voxel that follows)
    DevBuf<int> d_raw_count, d_pre_count, d_down_count, d_sel_count, d_block_counts, d_block_offsets;
    DevBuf<ScanSlot> d_slots;
    DevBuf<SegBlock> d_blocks;
    DevBuf<int> d_bbox, d_table_keys, d_table_counts, d_table_rank, d_vox_keys, d_member_off, d_members, d_pt_slot, d_n_vox,
        d_status;
    DevBuf<int> d_vox_fill, d_vox_count;  // per voxel of a scan: runs (k_voxel_fill) and points (k_voxel_sort)
    DevBuf<VoxelParams> d_vp;
    DevBuf<LidarStateDev> d_states;
    DevBuf<MapGrid> d_grids;
    DevBuf<int> d_perm, d_hard_count;
    // map_incremental, one slot per scan of the batch (allocated on first use)
    DevBuf<int> d_group_start, d_noneed, d_mapinc_out, d_batch_overflow;
    DevBuf<uint8_t> d_cls, d_has_append;
    DevBuf<MapIncRec> d_inc_recs;
    DevBuf<PointXYZINormal> d_appended;
    DevBuf<MapIncTask> d_inc_tasks;
    DevBuf<MapGridTask> d_grid_tasks;
    bool pre_bbox_valid = false;      // d_bbox holds the boxes of d_pre (k_pre_stream) -- until something else writes either
    DevBuf<MapInsTask> d_ins_tasks;   // in-place grid insertion of a batch (k_map_ins_*): tasks, [2] result words per task
    DevBuf<int> d_ins_out;
    PinnedBuf<int> h_ins_out;
    PinnedBuf<int> h_mapinc_out;
    std::vector<int> last_down;  // per slot: down-sampled points of the last feature extraction
    std::vector<std::pair<const tc2li_lidar_map*, uint64_t>> last_map;  // per slot: the map searched and its generation then
    std::vector<int> last_sel;   // per slot: selected features (laserCloudOri) of the last tc2li_lidar_frontend_batch
    DevBuf<TransformTask> d_xform_tasks;
    DevBuf<PointXYZINormal> d_xform_out;
    DevBuf<float4> d_recs;  // 2 per point: the voxel filter's records in summation order
    DevBuf<int2> d_hard_list;
    DevBuf<Pose6DDev> d_imu_poses;
    // tc2li_lidar_inertial_frontend_batch: IMU poses per scan, the time sort's flags, block / scan lists of an iteration, pinned staging
    DevBuf<int> d_n_poses, d_sort_fallback /* [max_scans] flags | [max_scans] range counts */, d_sort_ranges, d_scan_list;
    DevBuf<SegBlock> d_blocks_a, d_blocks_b, d_blocks_c;
    PinnedBuf<uint8_t> h_batch;
    DevBuf<double> d_eskf_partial;  // [(cap + 255) / 256][kEskfOutSize]
    PinnedBuf<double> h_eskf_out;   // [kEskfOutSize], written by k_eskf_reduce
    PinnedBuf<int> h_counts;  // [4 * max_scans + 1]: pre, down, sel counts and the status word
    PinnedBuf<int> h_sort_flags;  // [max_scans]: scans whose time sort reached std::sort's depth limit
    int prepared_scans = 0;       // tc2li_lidar_inertial_prepare_batch: d_pre / d_perm / h_counts hold that many scans, ready for the front end
    std::vector<ScanSlot> slots;
    std::vector<SegBlock> blocks;
    // The same tables cut for the DOWN-SAMPLED clouds of a batch (compact_segments): the block list above covers the raw scans' sizes
    // (128 blocks per KITTI scan), the 4 000 points a scan keeps after the voxel filter fill 4 of them -- a pass over the down-sampled
    // clouds launched on that list started 97 % of its workgroups (10^6 per k_knn_plane launch of 512 scans) only to let them leave.
    DevBuf<ScanSlot> d_slots_down;
    DevBuf<SegBlock> d_blocks_down;
    PinnedBuf<int> h_down;  // [max_scans]
    std::vector<ScanSlot> slots_down;
    std::vector<SegBlock> blocks_down;
    int nb_down = -1;       // blocks of the compact list, -1: none (the passes use d_slots / d_blocks)
    int n_scans = 0;
    // stage boundaries of the last tc2li_lidar_frontend_batch call: start, after preprocess, before / after the centroid
    // kernel, after the kNN + plane kernel, end
    hipEvent_t ev[7] = {};  // [6]: between the two kernels of the neighbour search
    bool timed = false;
    void record(int k, hipStream_t st) { if (ev[k] || hipEventCreate(&ev[k]) == hipSuccess) (void)hipEventRecord(ev[k], st); }
    ~tc2li_lidar() { for (auto& e : ev) if (e) (void)hipEventDestroy(e); }
};

namespace {

int next_pow2(int v) { int p = 1; while (p < v) p <<= 1; return p; }

// Builds the slot/block tables for `n_scans` scans with the given upper bounds of points per scan.
int setup_segments(tc2li_lidar* L, int n_scans, const int* upper, hipStream_t st, const int32_t* raw_offsets = nullptr) {
    L->slots.resize(n_scans);
    L->blocks.clear();
    L->pre_bbox_valid = false;  // new scans in the slots: d_bbox describes nothing yet
    L->nb_down = -1;
    for (int s = 0; s < n_scans; ++s) {
        if (upper[s] > L->cap) { set_error("scan %d has %d points, slot capacity is %d", s, upper[s], L->cap); return TC2LI_ERR_CAPACITY; }
        ScanSlot& sl = L->slots[s];
        sl.base = s * L->cap; sl.cap = L->cap; sl.first_block = (int)L->blocks.size();
        sl.raw_base = raw_offsets ? raw_offsets[s] : sl.base;
        sl.pad_[0] = sl.pad_[1] = sl.pad_[2] = 0;
        sl.n_blocks = (upper[s] + kSegBlock - 1) / kSegBlock;
        for (int b = 0; b < sl.n_blocks; ++b) L->blocks.push_back(SegBlock{s, b * kSegBlock});
    }
    L->n_scans = n_scans;
    TC2LI_HIP_CHECK(hipMemcpyAsync(L->d_slots.p, L->slots.data(), n_scans * sizeof(ScanSlot), hipMemcpyHostToDevice, st));
    if (!L->blocks.empty())
        TC2LI_HIP_CHECK(hipMemcpyAsync(L->d_blocks.p, L->blocks.data(), L->blocks.size() * sizeof(SegBlock), hipMemcpyHostToDevice, st));
    return TC2LI_OK;
}

// The slot / block tables for the down-sampled clouds of the scans in the slots: their sizes (`d_counts`, one per scan, final on the
// stream) come to the host -- one wait in the middle of the call -- and the passes over those clouds are launched with the blocks they need.
int compact_segments(tc2li_lidar* L, const int* d_counts, hipStream_t st) {
    const int S = L->n_scans;
    L->nb_down = -1;
    TC2LI_HIP_CHECK(hipMemcpyAsync(L->h_down.p, d_counts, S * sizeof(int), hipMemcpyDeviceToHost, st));
    TC2LI_HIP_CHECK(stream_wait_blocking(st));
    std::vector<ScanSlot>& slots = L->slots_down;
    std::vector<SegBlock>& blocks = L->blocks_down;
    slots.assign(L->slots.begin(), L->slots.begin() + S);
    blocks.clear();
    for (int s = 0; s < S; ++s) {
        const int n = L->h_down.p[s];
        if (n < 0 || n > L->cap) { set_error("scan %d: %d down-sampled points in a slot of %d", s, n, L->cap); return TC2LI_ERR_CAPACITY; }
        slots[s].first_block = (int)blocks.size();
        slots[s].n_blocks = (n + kSegBlock - 1) / kSegBlock;
        for (int b = 0; b < slots[s].n_blocks; ++b) blocks.push_back(SegBlock{s, b * kSegBlock});
    }
    TC2LI_HIP_CHECK(hipMemcpyAsync(L->d_slots_down.p, slots.data(), S * sizeof(ScanSlot), hipMemcpyHostToDevice, st));
    if (!blocks.empty()) TC2LI_HIP_CHECK(hipMemcpyAsync(L->d_blocks_down.p, blocks.data(), blocks.size() * sizeof(SegBlock), hipMemcpyHostToDevice, st));
    L->nb_down = (int)blocks.size();
    return TC2LI_OK;
}

// b1
int run_preprocess(tc2li_lidar* L, const VelodynePoint* d_raw, int point_filter_num, double blind, float time_unit_scale, hipStream_t st,
                   float* time_out = nullptr, bool* times_written = nullptr, float voxel_leaf = 0.f /* > 0: the voxel filter follows at this leaf */) {
    if (times_written) *times_written = false;
    L->pre_vkey_leaf = 0.f;
    PreprocessParams prm{point_filter_num, time_unit_scale, blind * blind};
    const int nb = (int)L->blocks.size();
    // a batch of scans: one pass over every raw scan (k_pre_stream, a workgroup per scan), which also leaves the voxel filter its bounding
    // boxes; a few scans: the three-launch form, whose passes are spread over all points.  TC2LI_PRE_STREAM=0 / 1 forces one (tests run both).
    const char* env = getenv("TC2LI_PRE_STREAM");
    L->pre_bbox_valid = false;
    if (env ? atoi(env) != 0 : L->n_scans >= 64) {
        // (the voxel filter's sorted form follows at voxel_leaf: the pass leaves it the points' voxel coordinates in the sort's second key buffer)
        const char* vk_env = getenv("TC2LI_VOXEL_PRE_KEYS");  // =0: the filter reads the points' positions itself (A/B, tests)
        const bool vkeys = voxel_leaf > 0.f && !(vk_env && atoi(vk_env) == 0);
        if (vkeys) TC2LI_HIP_CHECK(L->d_vk_ok.ensure(L->max_scans));
        launch_pre_stream(d_raw, L->d_raw_count.p, L->d_slots.p, L->n_scans, prm, L->d_pre.p, L->d_pre_count.p, L->d_bbox.p, time_out,
                          vkeys ? L->d_members.p : nullptr, voxel_leaf, vkeys ? L->d_vk_ok.p : nullptr, st);
        if (vkeys) L->pre_vkey_leaf = voxel_leaf;
        TC2LI_HIP_CHECK(hipGetLastError());
        L->pre_bbox_valid = true;
        if (times_written) *times_written = time_out != nullptr;
        return TC2LI_OK;
    }
    launch_pre_count(d_raw, L->d_raw_count.p, L->d_slots.p, L->d_blocks.p, nb, prm, L->d_block_counts.p, st);
    launch_seg_scan(L->d_slots.p, L->n_scans, L->d_block_counts.p, L->d_block_offsets.p, L->d_pre_count.p, st);
    launch_pre_scatter(d_raw, L->d_raw_count.p, L->d_slots.p, L->d_blocks.p, nb, prm, L->d_block_offsets.p, L->d_pre.p, st);
    TC2LI_HIP_CHECK(hipGetLastError());
    return TC2LI_OK;
}

// b3: in = (pts, count) per slot -> d_down / d_down_count
int run_voxel(tc2li_lidar* L, const PointXYZINormal* d_in, const int* d_in_count, float leaf, hipStream_t st, bool compact = false) {
    const int S = L->n_scans, nb = (int)L->blocks.size();
    L->nb_down = -1;
    std::vector<VoxelParams> vp(S);
    for (int s = 0; s < S; ++s) { memset(&vp[s], 0, sizeof(VoxelParams)); vp[s].table_base = s * L->table_size; vp[s].table_mask = L->table_size - 1; }
    TC2LI_HIP_CHECK(hipMemcpyAsync(L->d_vp.p, vp.data(), S * sizeof(VoxelParams), hipMemcpyHostToDevice, st));
    // the bounding boxes: left by the one-pass preprocess when the filter's input is its output (k_pre_stream), else computed here
    const bool have_bbox = L->pre_bbox_valid && d_in == L->d_pre.p;
    const bool have_vkeys = have_bbox && L->pre_vkey_leaf == leaf && leaf > 0.f;  // k_pre_stream packed the points' voxel coordinates for this leaf
    L->pre_bbox_valid = false;
    L->pre_vkey_leaf = 0.f;
    if (!have_bbox) {
        std::vector<int> bbox_init(6 * S);
        for (int s = 0; s < S; ++s) for (int a = 0; a < 3; ++a) { bbox_init[6 * s + a] = 0x7fffffff; bbox_init[6 * s + 3 + a] = (int)0x80000000; }
        TC2LI_HIP_CHECK(hipMemcpyAsync(L->d_bbox.p, bbox_init.data(), bbox_init.size() * sizeof(int), hipMemcpyHostToDevice, st));
    }
    TC2LI_HIP_CHECK(hipMemsetAsync(L->d_n_vox.p, 0, (size_t)S * sizeof(int), st));
    TC2LI_HIP_CHECK(hipMemsetAsync(L->d_down_count.p, 0, (size_t)S * sizeof(int), st));  // an empty scan has no block that would write its count
    if (!have_bbox) launch_voxel_bbox(d_in, d_in_count, L->d_slots.p, L->d_blocks.p, nb, L->d_bbox.p, st);
    launch_voxel_params(L->d_bbox.p, d_in_count, L->d_slots.p, S, leaf, L->d_vp.p, st);
    // a batch of scans: one workgroup per scan sorts its (voxel, point) pairs (lidar_kernels.hip, "sorted form"); a few scans: the hash
    // form, whose passes are spread over all points.  TC2LI_VOXEL_SORTED=0 / 1 forces one (the tests run both).
    const char* env = getenv("TC2LI_VOXEL_SORTED");
    const bool sorted = env ? atoi(env) != 0 : S >= 32;
    if (sorted) {
        L->record(2, st);
        launch_voxel_sort_points(d_in, d_in_count, L->d_slots.p, S, leaf, L->d_vp.p, L->d_pt_slot.p, L->d_vox_keys.p, L->d_members.p, L->d_member_off.p,
                                 L->d_vox_fill.p, L->d_vox_count.p, L->d_n_vox.p, have_vkeys ? L->d_vk_ok.p : nullptr, st);
        TC2LI_HIP_CHECK(hipGetLastError());
        // a batch: the passes over the voxels (and, in run_features, over the down-sampled points) get block tables cut for n_vox
        if (compact) { const int rc = compact_segments(L, L->d_n_vox.p, st); if (rc != TC2LI_OK) return rc; }
        const bool c = L->nb_down >= 0;
        launch_voxel_sums(d_in, L->d_slots.p, L->d_blocks.p, nb, c ? L->d_slots_down.p : L->d_slots.p, c ? L->d_blocks_down.p : L->d_blocks.p, c ? L->nb_down : nb,
                          L->d_vp.p, L->d_vox_keys.p, L->d_member_off.p, L->d_vox_fill.p, L->d_vox_count.p, L->d_n_vox.p, L->d_recs.p, L->d_down.p,
                          L->d_down_count.p, st);
        TC2LI_HIP_CHECK(hipGetLastError());
        return TC2LI_OK;
    }
    launch_fill_int(L->d_table_keys.p, (size_t)S * L->table_size, -1, st);
    TC2LI_HIP_CHECK(hipMemsetAsync(L->d_table_counts.p, 0, (size_t)S * L->table_size * sizeof(int), st));
    launch_voxel_insert(d_in, d_in_count, L->d_slots.p, L->d_blocks.p, nb, leaf, L->d_vp.p, L->d_table_keys.p, L->d_table_counts.p, L->d_pt_slot.p, L->d_n_vox.p, L->d_vox_keys.p, st);
    launch_voxel_sort(L->d_slots.p, S, L->d_vp.p, d_in_count, L->d_table_keys.p, L->d_table_counts.p, L->d_table_rank.p, L->d_vox_keys.p,
                      L->d_member_off.p, L->d_vox_fill.p, L->d_vox_count.p, L->d_n_vox.p, L->d_status.p, st);
    launch_voxel_fill(d_in_count, L->d_slots.p, L->d_blocks.p, nb, L->d_vp.p, L->d_pt_slot.p, L->d_table_rank.p, L->d_member_off.p, L->d_vox_fill.p,
                      L->d_members.p, st);
    L->record(2, st);
    launch_voxel_centroid(d_in, d_in_count, L->d_slots.p, L->d_blocks.p, nb, leaf, L->d_vp.p, L->d_pt_slot.p, L->d_table_rank.p, L->d_n_vox.p,
                          L->d_member_off.p, L->d_vox_fill.p, L->d_vox_count.p, L->d_members.p, L->d_recs.p, L->d_down.p, L->d_down_count.p, st);
    TC2LI_HIP_CHECK(hipGetLastError());
    return TC2LI_OK;
}

// b4-b6
int run_features(tc2li_lidar* L, const PointXYZINormal* d_body, const int* d_body_count, tc2li_lidar_map* const* maps,
                 const tc2li_lidar_state* states, hipStream_t st) {
    // (the tables cut for the down-sampled clouds when the voxel filter of this call left them: compact_segments)
    const bool c = L->nb_down >= 0 && d_body == L->d_down.p;
    const int S = L->n_scans, nb = c ? L->nb_down : (int)L->blocks.size();
    const ScanSlot* const d_slots = c ? L->d_slots_down.p : L->d_slots.p;
    const SegBlock* const d_blocks = c ? L->d_blocks_down.p : L->d_blocks.p;
    std::vector<MapGrid> grids(S);
    for (int s = 0; s < S; ++s) grids[s] = maps[s]->grid;
    TC2LI_HIP_CHECK(hipMemcpyAsync(L->d_grids.p, grids.data(), S * sizeof(MapGrid), hipMemcpyHostToDevice, st));
    TC2LI_HIP_CHECK(hipMemcpyAsync(L->d_states.p, states, S * sizeof(LidarStateDev), hipMemcpyHostToDevice, st));
    launch_knn_plane(L->d_grids.p, d_body, d_body_count, d_slots, d_blocks, nb, L->d_states.p, L->d_world.p, L->d_selected.p,
                     L->d_normvec.p, L->d_nearest_idx.p, L->d_nearest_d.p, L->d_nfound.p, L->d_hard_count.p, L->d_hard_list.p, st,
                     (L->ev[6] || hipEventCreate(&L->ev[6]) == hipSuccess) ? L->ev[6] : nullptr);
    L->record(4, st);
    launch_sel_count(L->d_selected.p, d_body_count, d_slots, d_blocks, nb, L->d_block_counts.p, st);
    launch_seg_scan(d_slots, S, L->d_block_counts.p, L->d_block_offsets.p, L->d_sel_count.p, st);
    launch_sel_scatter(L->d_selected.p, d_body_count, d_slots, d_blocks, nb, L->d_block_offsets.p, d_body, L->d_normvec.p,
                       L->d_cloud_ori.p, L->d_corr.p, st);
    TC2LI_HIP_CHECK(hipGetLastError());
    return TC2LI_OK;
}

inline float dec_enc(int i) { const int v = i >= 0 ? i : i ^ 0x7fffffff; float f; memcpy(&f, &v, 4); return f; }

// Geometry of the dense grid over the map's bounding box: 1 m cells unless that would need more than 4M of them.
// A margin of cells around the box (8 along x and y, 2 along z, when that stays within the 4M): the points a moving sensor adds beyond
// the present box find cells waiting, so the grid is updated in place for some scans before the box has to grow (a rebuild).
MapGrid grid_geometry(const tc2li_lidar_map* m, float* cell_out) {
    MapGrid g{};
    float cell = 1.0f;
    static const bool kNoMargin = getenv("TC2LI_MAP_NO_MARGIN") != nullptr;  // tests: the box as tight as rounds 1-3 had it
    for (;;) {
        const float inv = 1.0f / cell;
        long long cells = 1, cells_m = 1;
        int o[3], d[3];
        const int margin[3] = {8, 8, 2};
        for (int a = 0; a < 3; ++a) {
            o[a] = m->n ? (int)std::floor(m->lo[a] * inv) : 0;
            d[a] = m->n ? (int)std::floor(m->hi[a] * inv) - o[a] + 1 : 1;
            cells *= d[a];
            cells_m *= d[a] + 2 * margin[a];
        }
        if (cells <= (4ll << 20)) {
            if (m->n && cells_m <= (4ll << 20) && !kNoMargin)
                for (int a = 0; a < 3; ++a) { o[a] -= margin[a]; d[a] += 2 * margin[a]; }
            g.x0 = o[0]; g.y0 = o[1]; g.z0 = o[2]; g.nx = d[0]; g.ny = d[1]; g.nz = d[2];
            g.inv_cell = inv; g.cell = cell;
            break;
        }
        cell *= 1.5f;
    }
    g.nsx = (g.nx + kMapSegCells - 1) / kMapSegCells;
    *cell_out = cell;
    return g;
}

// Sizes the map's grid arrays for its current points and fills the build task (no launch).
int grid_prepare(tc2li_lidar_map* m, MapGridTask* t, hipStream_t st) {
    float cell;
    MapGrid g = grid_geometry(m, &cell);
    m->cell = cell;
    const int nc = g.nx * g.ny * g.nz;
    const long long nseg = (long long)g.ny * g.nz * g.nsx;
    if (nc > m->n_cells) {
        TC2LI_HIP_CHECK(m->d_bucket_counts.alloc(nc + nc / 2));
        TC2LI_HIP_CHECK(hipMemsetAsync(m->d_bucket_counts.p, 0, ((size_t)nc + nc / 2) * sizeof(int), st));  // zero outside a build (map_kernels.hip)
        TC2LI_HIP_CHECK(m->d_bucket_start.alloc((size_t)nc + nc / 2 + 1));
        TC2LI_HIP_CHECK(m->d_tile_sums.alloc((size_t)(nc + nc / 2) / 4096 + 2));
        m->n_cells = nc + nc / 2;
        m->counts_dirty = false;
    }
    if (m->counts_dirty) {  // a build that did not complete (a HIP error between its count and its scatter) leaves counters raised
        TC2LI_HIP_CHECK(hipMemsetAsync(m->d_bucket_counts.p, 0, (size_t)m->n_cells * sizeof(int), st));
        m->counts_dirty = false;
    }
    // a segment gets room for as many entries again as it has, + seg_slack
    const int seg_slack = (int)std::max(4ll, std::min(32ll, 4ll * m->n / std::max(nseg, 1ll)));
    const long long n_slots = 2ll * m->n + (long long)seg_slack * nseg;
    if (n_slots > 0x7fffffffll - 1024 || nseg * kMapSegStride + 1 > 0x7fffffffll || nseg >= (1ll << 27)) {
        set_error("LiDAR map: %d points in %lld segments do not fit the grid's 31-bit places", m->n, nseg);
        return TC2LI_ERR_CAPACITY;
    }
    TC2LI_HIP_CHECK(m->d_row_start.ensure((size_t)nseg * kMapSegStride + 1));
    // the points that were in the map before are taken from the old grid's order (cell-coherent, see MapGridTask): the new grid goes to the other buffer
    const bool from_old = m->rebuild_mode != 0 && m->grid_valid && m->d_sorted.p && m->grid.n_slots > 0;
    DevBuf<float4>& target = from_old ? m->d_sorted_alt : m->d_sorted;
    TC2LI_HIP_CHECK(target.ensure((size_t)std::max(n_slots, 1ll)));
    TC2LI_HIP_CHECK(hipMemsetAsync(target.p, 0xff, (size_t)n_slots * sizeof(float4), st));  // all tombstones (index -1): the build fills in the entries
    const int old_slots = m->grid.n_slots;
    g.points = m->d_points.p; g.pts = target.p; g.bucket_start = m->d_row_start.p; g.n_points = m->n; g.n_slots = (int)n_slots;
    t->g = g; t->counts = m->d_bucket_counts.p; t->start = m->d_bucket_start.p; t->row_start = m->d_row_start.p; t->sorted = target.p;
    t->tile_sums = m->d_tile_sums.p; t->n_cells = nc; t->seg_slack = seg_slack;
    t->old_sorted = from_old ? m->d_sorted.p : nullptr;
    t->remap = from_old && m->rebuild_mode == 1 && m->have_remap ? m->d_remap.p : nullptr;
    t->n_old = from_old ? old_slots : 0; t->n_kept = from_old ? m->remap_n_kept : 0;
    if (from_old && m->rebuild_mode == 1 && !m->have_remap) { set_error("LiDAR map: compaction without its index map"); return TC2LI_ERR_INVALID; }
    if (from_old) { std::swap(m->d_sorted.p, m->d_sorted_alt.p); std::swap(m->d_sorted.n, m->d_sorted_alt.n); }  // d_sorted = the new grid from here on
    m->have_remap = false;
    m->rebuild_mode = 0;
    return TC2LI_OK;
}

// (Re)builds the grids of a set of maps with one launch per phase; tasks go up through d_tasks; waits for the stream (the maps' counters
// are known to be zero again when it returns).
int rebuild_grids(tc2li_lidar_map* const* maps, int n_maps, DevBuf<MapGridTask>& d_tasks, hipStream_t st) {
    if (n_maps <= 0) return TC2LI_OK;
    std::vector<MapGridTask> tasks(n_maps);
    int max_points = 0, max_cells = 0, max_row_entries = 0;
    for (int i = 0; i < n_maps; ++i) {
        maps[i]->grid_valid = maps[i]->grid_valid && maps[i]->rebuild_mode != 0;  // mode 0 does not read the old grid
        const int rc = grid_prepare(maps[i], &tasks[i], st);
        if (rc != TC2LI_OK) { maps[i]->grid_valid = false; return rc; }
        max_points = std::max(max_points, tasks[i].n_old + (maps[i]->n - tasks[i].n_kept));
        max_cells = std::max(max_cells, tasks[i].n_cells);
        max_row_entries = std::max(max_row_entries, tasks[i].g.ny * tasks[i].g.nz * tasks[i].g.nsx * kMapSegStride + 1);
    }
    for (int i = 0; i < n_maps; ++i) { maps[i]->counts_dirty = true; maps[i]->grid_valid = false; }
    TC2LI_HIP_CHECK(d_tasks.ensure(n_maps));
    TC2LI_HIP_CHECK(hipMemcpyAsync(d_tasks.p, tasks.data(), n_maps * sizeof(MapGridTask), hipMemcpyHostToDevice, st));
    launch_map_grid_build(d_tasks.p, n_maps, max_points, max_cells, max_row_entries, st);
    TC2LI_HIP_CHECK(hipGetLastError());
    TC2LI_HIP_CHECK(stream_wait_blocking(st));
    for (int i = 0; i < n_maps; ++i) {
        tc2li_lidar_map* m = maps[i];
        m->grid = tasks[i].g; m->grid_valid = true; m->counts_dirty = false; m->tombstones = 0; ++m->n_grid_builds;
    }
    return TC2LI_OK;
}
int rebuild_grid(tc2li_lidar_map* m, hipStream_t st) { return rebuild_grids(&m, 1, m->d_grid_task, st); }

// The flags must be zero outside a call: a grown buffer starts zeroed.
int ensure_deleted(tc2li_lidar_map* m, int n, hipStream_t st) {
    if ((size_t)n <= m->d_deleted.n) return TC2LI_OK;
    TC2LI_HIP_CHECK(m->d_deleted.alloc((size_t)n + n / 2 + 1024));
    TC2LI_HIP_CHECK(hipMemsetAsync(m->d_deleted.p, 0, m->d_deleted.n, st));
    return TC2LI_OK;
}

// Room for `need` points, the present ones kept (the compaction works in place and appends behind the kept points).
int ensure_point_capacity(tc2li_lidar_map* m, size_t need, hipStream_t st) {
    if (need <= m->d_points.n) return TC2LI_OK;
    DevBuf<PointXYZINormal> bigger;
    TC2LI_HIP_CHECK(bigger.alloc(need + need / 2 + 1024));
    if (m->n) TC2LI_HIP_CHECK(copy_sync(bigger.p, m->d_points.p, (size_t)m->n * sizeof(PointXYZINormal), hipMemcpyDeviceToDevice, st));
    std::swap(bigger.p, m->d_points.p);
    std::swap(bigger.n, m->d_points.n);
    m->grid.points = m->d_points.p;
    return TC2LI_OK;
}

// After the compaction kernels of a batch have finished (results of task i at out + kMapIncOut * i on the host): the maps take over
// their new point lists.
void commit_compaction(tc2li_lidar_map* m, const int* out, bool has_inc) {
    const int kept = out[4], appended = has_inc ? out[5] : 0, noneed = has_inc ? out[2] : 0;
    m->have_remap = m->d_remap.n >= (size_t)m->n && m->n > 0;  // the compaction wrote d_remap for the m->n points the map had
    m->remap_n_old = m->n; m->remap_n_kept = kept;
    m->rebuild_mode = m->have_remap ? 1 : 0;  // the callers that maintained the grid in place change it to 2 (or update the grid themselves)
    m->tombstones += m->n - kept;
    const int added = appended + noneed;
    if (added > 0)
        for (int a = 0; a < 3; ++a) {
            const float lo = dec_enc(out[6 + a]), hi = dec_enc(out[9 + a]);
            if (kept == 0) { m->lo[a] = lo; m->hi[a] = hi; }  // otherwise the old box stays a (possibly loose) superset
            m->lo[a] = std::min(m->lo[a], lo);
            m->hi[a] = std::max(m->hi[a], hi);
        }
    m->n = kept + added;
    ++m->generation;
}

}  // namespace

extern "C" {

int tc2li_lidar_create(int max_points_per_scan, int max_scans, tc2li_lidar** out) {
    if (!out || max_points_per_scan <= 0 || max_scans <= 0) { set_error("tc2li_lidar_create: invalid argument"); return TC2LI_ERR_INVALID; }
    // the voxel filter's run records keep a point index in 24 bits (k_voxel_fill)
    if (max_points_per_scan > (1 << 24) - 1024) { set_error("tc2li_lidar_create: at most %d points per scan", (1 << 24) - 1024); return TC2LI_ERR_INVALID; }
    if (!device_ready()) return TC2LI_ERR_NO_DEVICE;
    std::unique_ptr<tc2li_lidar> L(new tc2li_lidar());
    hipStream_t ps = private_stream();
    L->max_scans = max_scans;
    L->cap = (max_points_per_scan + kSegBlock - 1) / kSegBlock * kSegBlock;
    L->total = (size_t)L->cap * max_scans;
    L->table_size = next_pow2(2 * L->cap);
    const size_t T = L->total, S = max_scans, NB = T / kSegBlock;
    TC2LI_HIP_CHECK(L->d_raw.alloc(T));
    TC2LI_HIP_CHECK(L->d_pre.alloc(T)); TC2LI_HIP_CHECK(L->d_down.alloc(T)); TC2LI_HIP_CHECK(L->d_world.alloc(T));
    TC2LI_HIP_CHECK(L->d_normvec.alloc(T)); TC2LI_HIP_CHECK(L->d_cloud_ori.alloc(T)); TC2LI_HIP_CHECK(L->d_corr.alloc(T));
    TC2LI_HIP_CHECK(L->d_selected.alloc(T)); TC2LI_HIP_CHECK(L->d_nearest_idx.alloc(T * 5)); TC2LI_HIP_CHECK(L->d_nearest_d.alloc(T * 5));
    TC2LI_HIP_CHECK(L->d_nfound.alloc(T));
    TC2LI_HIP_CHECK(L->d_raw_count.alloc(S)); TC2LI_HIP_CHECK(L->d_pre_count.alloc(S)); TC2LI_HIP_CHECK(L->d_down_count.alloc(S));
    TC2LI_HIP_CHECK(L->d_sel_count.alloc(S)); TC2LI_HIP_CHECK(L->d_block_counts.alloc(NB)); TC2LI_HIP_CHECK(L->d_block_offsets.alloc(NB));
    TC2LI_HIP_CHECK(L->d_slots.alloc(S)); TC2LI_HIP_CHECK(L->d_blocks.alloc(NB));
    TC2LI_HIP_CHECK(L->d_slots_down.alloc(S)); TC2LI_HIP_CHECK(L->d_blocks_down.alloc(NB)); TC2LI_HIP_CHECK(L->h_down.alloc(S));
    TC2LI_HIP_CHECK(L->d_bbox.alloc(6 * S)); TC2LI_HIP_CHECK(L->d_vp.alloc(S));
    TC2LI_HIP_CHECK(L->d_table_keys.alloc(S * L->table_size)); TC2LI_HIP_CHECK(L->d_table_counts.alloc(S * L->table_size));
    TC2LI_HIP_CHECK(L->d_table_rank.alloc(S * L->table_size));
    TC2LI_HIP_CHECK(L->d_vox_keys.alloc(T)); TC2LI_HIP_CHECK(L->d_member_off.alloc(T)); TC2LI_HIP_CHECK(L->d_vox_fill.alloc(T)); TC2LI_HIP_CHECK(L->d_vox_count.alloc(T));
    TC2LI_HIP_CHECK(L->d_members.alloc(T)); TC2LI_HIP_CHECK(L->d_pt_slot.alloc(T)); TC2LI_HIP_CHECK(L->d_n_vox.alloc(S)); TC2LI_HIP_CHECK(L->d_status.alloc(1));
    TC2LI_HIP_CHECK(L->d_states.alloc(S)); TC2LI_HIP_CHECK(L->d_grids.alloc(S));
    TC2LI_HIP_CHECK(L->d_hard_count.alloc(1)); TC2LI_HIP_CHECK(L->d_hard_list.alloc(T)); TC2LI_HIP_CHECK(L->d_recs.alloc(2 * T));
    TC2LI_HIP_CHECK(L->h_counts.alloc(4 * S + 1));
    TC2LI_HIP_CHECK(memset_sync(L->d_status.p, 0, sizeof(int), ps));
    *out = L.release();
    return TC2LI_OK;
}

void tc2li_lidar_destroy(tc2li_lidar* L) { delete L; }

int tc2li_lidar_last_timings(tc2li_lidar* L, float ms[8]) {
    if (!L || !ms) { set_error("tc2li_lidar_last_timings: invalid argument"); return TC2LI_ERR_INVALID; }
    for (int k = 0; k < 8; ++k) ms[k] = 0;
    if (!L->timed) return TC2LI_OK;
    for (int k = 0; k < 5; ++k) if (hipEventElapsedTime(&ms[k], L->ev[k], L->ev[k + 1]) != hipSuccess) ms[k] = 0;
    if (hipEventElapsedTime(&ms[5], L->ev[0], L->ev[5]) != hipSuccess) ms[5] = 0;
    // the neighbour search split into its two kernels: k_knn_plane, k_knn_hard
    if (!L->ev[6] || hipEventElapsedTime(&ms[6], L->ev[3], L->ev[6]) != hipSuccess) ms[6] = 0;
    if (!L->ev[6] || hipEventElapsedTime(&ms[7], L->ev[6], L->ev[4]) != hipSuccess) ms[7] = 0;
    return TC2LI_OK;
}

int tc2li_lidar_preprocess(tc2li_lidar* L, const tc2li_velodyne_point* raw, int n, int point_filter_num, double blind,
                           float time_unit_scale, tc2li_point* out, int capacity) {
    if (!L || n < 0 || (n > 0 && !raw) || point_filter_num < 1 || !out) { set_error("tc2li_lidar_preprocess: invalid argument"); return TC2LI_ERR_INVALID; }
    if (n == 0) return 0;  // preprocess.cpp:97 `if (plsize == 0) return;`
    hipStream_t ps = private_stream();
    int rc = setup_segments(L, 1, &n, ps);
    if (rc != TC2LI_OK) return rc;
    TC2LI_HIP_CHECK(copy_sync(L->d_raw.p, raw, (size_t)n * sizeof(VelodynePoint), hipMemcpyHostToDevice, ps));
    TC2LI_HIP_CHECK(copy_sync(L->d_raw_count.p, &n, sizeof(int), hipMemcpyHostToDevice, ps));
    rc = run_preprocess(L, L->d_raw.p, point_filter_num, blind, time_unit_scale, ps);
    if (rc != TC2LI_OK) return rc;
    int m = 0;
    TC2LI_HIP_CHECK(copy_sync(&m, L->d_pre_count.p, sizeof(int), hipMemcpyDeviceToHost, ps));
    if (m > capacity) { set_error("output capacity %d < %d", capacity, m); return TC2LI_ERR_CAPACITY; }
    if (m) TC2LI_HIP_CHECK(copy_sync(out, L->d_pre.p, (size_t)m * sizeof(PointXYZINormal), hipMemcpyDeviceToHost, ps));
    return m;
}

int tc2li_lidar_voxel_filter(tc2li_lidar* L, const tc2li_point* in, int n, float leaf, tc2li_point* out, int capacity) {
    if (!L || n < 0 || (n > 0 && !in) || !(leaf > 0) || !out) { set_error("tc2li_lidar_voxel_filter: invalid argument"); return TC2LI_ERR_INVALID; }
    if (n == 0) return 0;
    hipStream_t ps = private_stream();
    int rc = setup_segments(L, 1, &n, ps);
    if (rc != TC2LI_OK) return rc;
    TC2LI_HIP_CHECK(copy_sync(L->d_pre.p, in, (size_t)n * sizeof(PointXYZINormal), hipMemcpyHostToDevice, ps));
    TC2LI_HIP_CHECK(copy_sync(L->d_pre_count.p, &n, sizeof(int), hipMemcpyHostToDevice, ps));
    rc = run_voxel(L, L->d_pre.p, L->d_pre_count.p, leaf, ps);
    if (rc != TC2LI_OK) return rc;
    int m = 0, status = 0;
    TC2LI_HIP_CHECK(copy_sync(&m, L->d_down_count.p, sizeof(int), hipMemcpyDeviceToHost, ps));
    TC2LI_HIP_CHECK(copy_sync(&status, L->d_status.p, sizeof(int), hipMemcpyDeviceToHost, ps));
    if (status) { TC2LI_HIP_CHECK(memset_sync(L->d_status.p, 0, sizeof(int), ps)); set_error("more than 32768 occupied voxels in one scan"); return TC2LI_ERR_CAPACITY; }
    if (m > capacity) { set_error("output capacity %d < %d", capacity, m); return TC2LI_ERR_CAPACITY; }
    if (m) TC2LI_HIP_CHECK(copy_sync(out, L->d_down.p, (size_t)m * sizeof(PointXYZINormal), hipMemcpyDeviceToHost, ps));
    return m;
}

int tc2li_lidar_undistort(tc2li_lidar* L, tc2li_point* points, int n, const tc2li_imu_pose6d* imu_poses, int n_poses,
                          const tc2li_lidar_state* end_state) {
    if (!L || n < 0 || (n > 0 && !points) || n_poses < 0 || (n_poses > 0 && !imu_poses) || !end_state) {
        set_error("tc2li_lidar_undistort: invalid argument");
        return TC2LI_ERR_INVALID;
    }
    if (n_poses > kMaxImuPoses) { set_error("more than %d IMU poses in one scan", kMaxImuPoses); return TC2LI_ERR_CAPACITY; }
    if (n > L->cap) { set_error("scan has %d points, slot capacity is %d", n, L->cap); return TC2LI_ERR_CAPACITY; }
    if (n == 0) return 0;
    // sort(pcl_out.points.begin(), pcl_out.points.end(), time_list): std::sort is not stable, and the order it leaves equal
    // time stamps in is part of the reference's result (the voxel filter sums in point order).  Its moves depend only on the
    // comparison outcomes, so sorting (time, index) records with the same comparator yields the same permutation.
    struct Rec { float t; int idx; };
    std::vector<Rec> rec(n);
    const PointXYZINormal* P = reinterpret_cast<const PointXYZINormal*>(points);
    for (int i = 0; i < n; ++i) rec[i] = Rec{P[i].curvature, i};
    std::sort(rec.begin(), rec.end(), [](const Rec& x, const Rec& y) { return x.t < y.t; });
    std::vector<int> perm(n);
    for (int i = 0; i < n; ++i) perm[i] = rec[i].idx;
    static_assert(sizeof(tc2li_imu_pose6d) == sizeof(Pose6DDev), "ABI layout");
    hipStream_t ps = private_stream();
    TC2LI_HIP_CHECK(L->d_perm.ensure(n));
    TC2LI_HIP_CHECK(L->d_imu_poses.ensure(std::max(n_poses, 1)));
    TC2LI_HIP_CHECK(copy_sync(L->d_pre.p, points, (size_t)n * sizeof(PointXYZINormal), hipMemcpyHostToDevice, ps));
    TC2LI_HIP_CHECK(copy_sync(L->d_perm.p, perm.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, ps));
    if (n_poses) TC2LI_HIP_CHECK(copy_sync(L->d_imu_poses.p, imu_poses, (size_t)n_poses * sizeof(Pose6DDev), hipMemcpyHostToDevice, ps));
    TC2LI_HIP_CHECK(copy_sync(L->d_states.p, end_state, sizeof(LidarStateDev), hipMemcpyHostToDevice, ps));
    launch_undistort(L->d_pre.p, L->d_perm.p, n, L->d_imu_poses.p, n_poses, L->d_states.p, L->d_down.p, ps);
    TC2LI_HIP_CHECK(hipGetLastError());
    TC2LI_HIP_CHECK(copy_sync(points, L->d_down.p, (size_t)n * sizeof(PointXYZINormal), hipMemcpyDeviceToHost, ps));
    return n;
}

// Forward propagation at the IMU samples of one scan (host; a dozen samples): the state, and with P / Q the covariance as well.
static int imu_propagate_impl(tc2li_imu_state* st, eskf::Cov* P, const eskf::Mat<eskf::kW, eskf::kW>* Q, const tc2li_imu_meas* v, int n_imu,
                              double pcl_beg_time, double pcl_end_time, double last_lidar_end_time, double acc_scale, double acc_s_last[3],
                              double angvel_last[3], tc2li_imu_pose6d* poses, int capacity) {
    if (!st || n_imu < 1 || !v || !acc_s_last || !angvel_last || !poses || capacity < 1) { set_error("tc2li_lidar_imu_propagate: invalid argument"); return TC2LI_ERR_INVALID; }
    int np = 0;
    auto save = [&](double t) {
        tc2li_imu_pose6d& p = poses[np++];
        p.offset_time = t;
        memcpy(p.acc, acc_s_last, 24); memcpy(p.gyr, angvel_last, 24); memcpy(p.vel, st->vel, 24); memcpy(p.pos, st->pos, 24); memcpy(p.rot, st->rot, 72);
    };
    auto rotv = [](const double* R, const double* x, double* o) { for (int r = 0; r < 3; ++r) o[r] = R[3 * r] * x[0] + R[3 * r + 1] * x[1] + R[3 * r + 2] * x[2]; };
    auto predict = [&](double dt, const double* acc, const double* gyr) {  // x <- x [+] f(x, u) dt  (use-ikfom.hpp get_f)
        if (P) { eskf::predict(*st, *P, *Q, acc, gyr, dt); return; }
        double am[3], Ra[3], th[3];
        for (int k = 0; k < 3; ++k) { th[k] = (gyr[k] - st->bg[k]) * dt; am[k] = acc[k] - st->ba[k]; }
        rotv(st->rot, am, Ra);
        for (int k = 0; k < 3; ++k) st->pos[k] += st->vel[k] * dt;
        const double n = std::sqrt(th[0] * th[0] + th[1] * th[1] + th[2] * th[2]);
        if (n > 0.0000001) {
            const double a[3] = {th[0] / n, th[1] / n, th[2] / n};
            const double K[9] = {0, -a[2], a[1], a[2], 0, -a[0], -a[1], a[0], 0};
            const double s = std::sin(n), c1 = 1.0 - std::cos(n);
            double E[9], Rn[9];
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) {
                    const double kk = (c1 * K[3 * r]) * K[c] + (c1 * K[3 * r + 1]) * K[3 + c] + (c1 * K[3 * r + 2]) * K[6 + c];
                    E[3 * r + c] = ((r == c ? 1.0 : 0.0) + s * K[3 * r + c]) + kk;
                }
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) Rn[3 * r + c] = st->rot[3 * r] * E[c] + st->rot[3 * r + 1] * E[3 + c] + st->rot[3 * r + 2] * E[6 + c];
            memcpy(st->rot, Rn, sizeof(Rn));
        }
        for (int k = 0; k < 3; ++k) st->vel[k] += (Ra[k] + st->grav[k]) * dt;
    };
    save(0.0);
    double acc_avr[3] = {0, 0, 0}, w_avr[3] = {0, 0, 0};
    for (int i = 0; i + 1 < n_imu; ++i) {
        const tc2li_imu_meas& head = v[i];
        const tc2li_imu_meas& tail = v[i + 1];
        if (tail.t < last_lidar_end_time) continue;
        for (int k = 0; k < 3; ++k) { w_avr[k] = 0.5 * (head.gyr[k] + tail.gyr[k]); acc_avr[k] = 0.5 * (head.acc[k] + tail.acc[k]) * acc_scale; }
        predict(head.t < last_lidar_end_time ? tail.t - last_lidar_end_time : tail.t - head.t, acc_avr, w_avr);
        double am[3];
        for (int k = 0; k < 3; ++k) { angvel_last[k] = w_avr[k] - st->bg[k]; am[k] = acc_avr[k] - st->ba[k]; }
        rotv(st->rot, am, acc_s_last);
        for (int k = 0; k < 3; ++k) acc_s_last[k] += st->grav[k];
        if (np >= capacity) { set_error("pose capacity %d too small", capacity); return TC2LI_ERR_CAPACITY; }
        save(tail.t - pcl_beg_time);
    }
    const double imu_end = v[n_imu - 1].t;
    predict((pcl_end_time > imu_end ? 1.0 : -1.0) * (pcl_end_time - imu_end), acc_avr, w_avr);
    return np;
}

int tc2li_lidar_imu_propagate(tc2li_imu_state* st, const tc2li_imu_meas* v, int n_imu, double pcl_beg_time, double pcl_end_time,
                              double last_lidar_end_time, double acc_scale, double acc_s_last[3], double angvel_last[3],
                              tc2li_imu_pose6d* poses, int capacity) {
    return imu_propagate_impl(st, nullptr, nullptr, v, n_imu, pcl_beg_time, pcl_end_time, last_lidar_end_time, acc_scale, acc_s_last, angvel_last,
                              poses, capacity);
}

int tc2li_lidar_imu_propagate_cov(tc2li_imu_state* st, double* P529, const double cov12[12], const tc2li_imu_meas* v, int n_imu,
                                  double pcl_beg_time, double pcl_end_time, double last_lidar_end_time, double acc_scale,
                                  double acc_s_last[3], double angvel_last[3], tc2li_imu_pose6d* poses, int capacity) {
    if (!P529 || !cov12) { set_error("tc2li_lidar_imu_propagate_cov: invalid argument"); return TC2LI_ERR_INVALID; }
    eskf::Cov P;
    memcpy(P.a, P529, sizeof(P.a));
    eskf::Mat<eskf::kW, eskf::kW> Q = eskf::Mat<eskf::kW, eskf::kW>::zero();
    for (int k = 0; k < eskf::kW; ++k) Q(k, k) = cov12[k];  // cov_gyr, cov_acc, cov_bias_gyr, cov_bias_acc (IMU_Processing.cpp:215-218)
    const int r = imu_propagate_impl(st, &P, &Q, v, n_imu, pcl_beg_time, pcl_end_time, last_lidar_end_time, acc_scale, acc_s_last, angvel_last, poses,
                                     capacity);
    if (r >= 0) memcpy(P529, P.a, sizeof(P.a));
    return r;
}

int tc2li_eskf_predict(tc2li_imu_state* st, double* P529, const double* Q144, const double acc[3], const double gyr[3], double dt) {
    if (!st || !P529 || !Q144 || !acc || !gyr) { set_error("tc2li_eskf_predict: invalid argument"); return TC2LI_ERR_INVALID; }
    eskf::Cov P;
    eskf::Mat<eskf::kW, eskf::kW> Q;
    memcpy(P.a, P529, sizeof(P.a));
    memcpy(Q.a, Q144, sizeof(Q.a));
    eskf::predict(*st, P, Q, acc, gyr, dt);
    memcpy(P529, P.a, sizeof(P.a));
    return TC2LI_OK;
}

}  // extern "C"

namespace {
// The host side of one iterated update (esekf::update_iterated_dyn_share_modified, esekfom.hpp:1621-1932): everything between two
// evaluations of h_share_model.  One object per scan; the one-scan entry point and the batch drive it with the same calls, so a scan
// gives the same state alone and in a batch.
struct EskfRun {
    tc2li_imu_state* x = nullptr;
    double* P529 = nullptr;
    tc2li_imu_state x_propagated;
    eskf::Cov P_propagated, P, K_x;
    double K_h[eskf::kN] = {0}, dx_new[eskf::kN] = {0};
    bool converge = true;
    int t = 0, calls = 0, searches = 0, effct = 0, finished = 0, rc = 0;
    double res_mean = 0;
    bool done = false;

    void begin(tc2li_imu_state* x_, double* P_) {
        x = x_; P529 = P_;
        x_propagated = *x;
        memcpy(P_propagated.a, P529, sizeof(P_propagated.a));
        P = P_propagated; K_x = eskf::Cov::zero();
    }
    tc2li_lidar_state lidar_state() const {
        tc2li_lidar_state ls;
        memcpy(ls.rot, x->rot, 72); memcpy(ls.pos, x->pos, 24); memcpy(ls.offset_R_L_I, x->offset_R_L_I, 72); memcpy(ls.offset_T_L_I, x->offset_T_L_I, 24);
        return ls;
    }
    void write_stats(tc2li_eskf_stats* stats) const {
        if (!stats) return;
        stats->calls = calls; stats->effct_feat_num = effct; stats->searches = searches; stats->converged = t; stats->finished = finished; stats->res_mean_last = res_mean;
    }
    // After the evaluation of iteration i (i = -1 .. maximum_iter - 1): o = H^T H (144) | H^T h (12) | sum |pd2| | rows.  rows_of(sel, nv)
    // fetches the scan's selection flags and normal vectors (only when there are fewer rows than states).  Sets `done` when the loop
    // of the reference has ended (covariance update, or the last iteration); rc < 0 on a singular matrix.
    template <class Fetch>
    void step(int i, int maximum_iter, const double* o, int n, const PointXYZINormal* body, double R, const double* limit23, int extrinsic_est_en, Fetch&& rows_of) {
        using namespace eskf;
        const int so3_idx[2] = {3, 6};
        ++calls;
        const int M = (int)o[157];
        effct = M;
        if (M >= 1) {
        res_mean = o[156] / M;
        // ---- the iteration (esekfom.hpp:1653-1800) ----
        double dx[kN];
        boxminus(*x, x_propagated, dx);
        for (int k = 0; k < kN; ++k) dx_new[k] = dx[k];
        P = P_propagated;
        for (int s : so3_idx) {
            const M3 At = transpose(A_matrix(dx + s));
            double v[3];
            for (int r = 0; r < 3; ++r) v[r] = At(r, 0) * dx_new[s] + At(r, 1) * dx_new[s + 1] + At(r, 2) * dx_new[s + 2];
            for (int r = 0; r < 3; ++r) dx_new[s + r] = v[r];
            rows_apply<3>(P, s, At);
            cols_apply<3>(P, s, At);
        }
        {
            const Mat<2, 2> T = mul(s2_Nx_yy(x->grav), s2_Mx(x_propagated.grav, dx + 21));
            const double v0 = T(0, 0) * dx_new[21] + T(0, 1) * dx_new[22], v1 = T(1, 0) * dx_new[21] + T(1, 1) * dx_new[22];
            dx_new[21] = v0; dx_new[22] = v1;
            rows_apply<2>(P, 21, T);
            cols_apply<2>(P, 21, T);
        }
        K_x = Cov::zero();
        if (kN > M) {
            // fewer rows than states: K = P Hc^T (Hc P Hc^T / R + I)^-1 / R on the explicit rows (at most 22 of them)
            std::vector<uint8_t> sel(n);
            std::vector<PointXYZINormal> nv(n);
            if (!rows_of(sel.data(), nv.data())) { rc = TC2LI_ERR_HIP; done = true; return; }
            std::vector<double> H((size_t)M * 12, 0.0), h(M), PHt((size_t)kN * M), S((size_t)M * M), Si((size_t)M * M), K((size_t)kN * M);
            const M3 Rw = m3_from(x->rot), Ro = m3_from(x->offset_R_L_I);
            int k = 0;
            for (int p = 0; p < n && k < M; ++p) {
                if (!sel[p]) continue;
                const double pbe[3] = {body[p].x, body[p].y, body[p].z}, nr[3] = {nv[p].x, nv[p].y, nv[p].z};
                double pt[3], C[3], D[3];
                for (int r = 0; r < 3; ++r) pt[r] = (Ro(r, 0) * pbe[0] + Ro(r, 1) * pbe[1] + Ro(r, 2) * pbe[2]) + x->offset_T_L_I[r];
                for (int r = 0; r < 3; ++r) C[r] = Rw(0, r) * nr[0] + Rw(1, r) * nr[1] + Rw(2, r) * nr[2];
                for (int r = 0; r < 3; ++r) D[r] = Ro(0, r) * C[0] + Ro(1, r) * C[1] + Ro(2, r) * C[2];
                double* row = &H[(size_t)k * 12];
                row[0] = nr[0]; row[1] = nr[1]; row[2] = nr[2];
                row[3] = pt[1] * C[2] - pt[2] * C[1]; row[4] = pt[2] * C[0] - pt[0] * C[2]; row[5] = pt[0] * C[1] - pt[1] * C[0];
                if (extrinsic_est_en) {
                    row[6] = pbe[1] * D[2] - pbe[2] * D[1]; row[7] = pbe[2] * D[0] - pbe[0] * D[2]; row[8] = pbe[0] * D[1] - pbe[1] * D[0];
                    row[9] = C[0]; row[10] = C[1]; row[11] = C[2];
                }
                h[k] = -(double)nv[p].intensity;
                ++k;
            }
            for (int r = 0; r < kN; ++r) for (int c = 0; c < M; ++c) { double s_ = 0; for (int q = 0; q < 12; ++q) s_ += P(r, q) * H[(size_t)c * 12 + q]; PHt[(size_t)r * M + c] = s_; }
            for (int r = 0; r < M; ++r) for (int c = 0; c < M; ++c) { double s_ = 0; for (int q = 0; q < 12; ++q) s_ += H[(size_t)r * 12 + q] * PHt[(size_t)q * M + c]; S[(size_t)r * M + c] = s_ / R + (r == c ? 1.0 : 0.0); }
            if (!lu_inverse(S.data(), M, Si.data())) { set_error("tc2li_lidar_eskf_update: singular innovation matrix"); rc = TC2LI_ERR_INVALID; done = true; return; }
            for (int r = 0; r < kN; ++r) for (int c = 0; c < M; ++c) { double s_ = 0; for (int q = 0; q < M; ++q) s_ += PHt[(size_t)r * M + q] * Si[(size_t)q * M + c]; K[(size_t)r * M + c] = s_ / R; }
            for (int r = 0; r < kN; ++r) { double s_ = 0; for (int q = 0; q < M; ++q) s_ += K[(size_t)r * M + q] * h[q]; K_h[r] = s_; }
            for (int r = 0; r < kN; ++r) for (int c = 0; c < 12; ++c) { double s_ = 0; for (int q = 0; q < M; ++q) s_ += K[(size_t)r * M + q] * H[(size_t)q * 12 + c]; K_x(r, c) = s_; }
        } else {
            Cov PR, P_temp, P_inv;
            for (int k = 0; k < kN * kN; ++k) PR.a[k] = P.a[k] / R;
            if (!lu_inverse(PR.a, kN, P_temp.a)) { set_error("tc2li_lidar_eskf_update: singular covariance"); rc = TC2LI_ERR_INVALID; done = true; return; }
            for (int r = 0; r < 12; ++r) for (int c = 0; c < 12; ++c) P_temp(r, c) += o[12 * r + c];
            if (!lu_inverse(P_temp.a, kN, P_inv.a)) { set_error("tc2li_lidar_eskf_update: singular information matrix"); rc = TC2LI_ERR_INVALID; done = true; return; }
            for (int r = 0; r < kN; ++r) { double s_ = 0; for (int k = 0; k < 12; ++k) s_ += P_inv(r, k) * o[144 + k]; K_h[r] = s_; }
            for (int r = 0; r < kN; ++r) for (int c = 0; c < 12; ++c) { double s_ = 0; for (int k = 0; k < 12; ++k) s_ += P_inv(r, k) * o[12 * k + c]; K_x(r, c) = s_; }
        }
        double dx_[kN];
        for (int r = 0; r < kN; ++r) {
            double s_ = K_h[r];
            for (int c = 0; c < kN; ++c) s_ += (K_x(r, c) - (r == c ? 1.0 : 0.0)) * dx_new[c];
            dx_[r] = s_;
        }
        boxplus(*x, dx_);
        converge = true;
        for (int k = 0; k < kN; ++k) if (std::fabs(dx_[k]) > limit23[k]) { converge = false; break; }
        if (converge) t++;
        if (!t && i == maximum_iter - 2) converge = true;
        if (t > 1 || i == maximum_iter - 1) {
            Cov Lm = P;
            for (int s : so3_idx) {
                const M3 At = transpose(A_matrix(dx_ + s));
                for (int c = 0; c < kN; ++c)
                    for (int r = 0; r < 3; ++r) Lm(s + r, c) = At(r, 0) * P(s, c) + At(r, 1) * P(s + 1, c) + At(r, 2) * P(s + 2, c);
                for (int c = 0; c < 12; ++c) {
                    const double k3[3] = {K_x(s, c), K_x(s + 1, c), K_x(s + 2, c)};
                    for (int r = 0; r < 3; ++r) K_x(s + r, c) = At(r, 0) * k3[0] + At(r, 1) * k3[1] + At(r, 2) * k3[2];
                }
                cols_apply<3>(Lm, s, At);
                cols_apply<3>(P, s, At);
            }
            {
                const Mat<2, 2> T = mul(s2_Nx_yy(x->grav), s2_Mx(x_propagated.grav, dx_ + 21));
                for (int c = 0; c < kN; ++c)
                    for (int r = 0; r < 2; ++r) Lm(21 + r, c) = T(r, 0) * P(21, c) + T(r, 1) * P(22, c);
                for (int c = 0; c < 12; ++c) {
                    const double k2[2] = {K_x(21, c), K_x(22, c)};
                    for (int r = 0; r < 2; ++r) K_x(21 + r, c) = T(r, 0) * k2[0] + T(r, 1) * k2[1];
                }
                cols_apply<2>(Lm, 21, T);
                cols_apply<2>(P, 21, T);
            }
            for (int r = 0; r < kN; ++r)
                for (int c = 0; c < kN; ++c) {
                    double s_ = 0;
                    for (int k = 0; k < 12; ++k) s_ += K_x(r, k) * P(k, c);
                    P529[r * kN + c] = Lm(r, c) - s_;
                }
            finished = 1;
            done = true;
            return;
        }
        }  // M >= 1 (otherwise ekfom_data.valid = false: the iteration is skipped)
        if (i == maximum_iter - 1) {  // the loop ends without the covariance update
            memcpy(P529, P.a, sizeof(P.a));
            done = true;
        }
    }
};
}  // namespace

extern "C" {

// esekf::update_iterated_dyn_share_modified with h_share_model as the measurement model (esekfom.hpp:1621-1932,
// LidarFrontEnd.cpp:485-602): neighbour search / plane fit / selection and the normal equations of the measurement rows on the
// device, the 23 x 23 algebra of the iteration on the host (EskfRun).
int tc2li_lidar_eskf_update(tc2li_lidar* L, tc2li_lidar_map* map, const tc2li_point* feats_down_body, int n, tc2li_imu_state* x, double* P529,
                            double R, int maximum_iter, const double* limit23, int extrinsic_est_en, tc2li_eskf_stats* stats) {
    if (!L || !map || n < 0 || (n > 0 && !feats_down_body) || !x || !P529 || !(R > 0) || maximum_iter < 0 || !limit23) {
        set_error("tc2li_lidar_eskf_update: invalid argument");
        return TC2LI_ERR_INVALID;
    }
    if (stats) memset(stats, 0, sizeof(*stats));
    if (!device_ready()) return TC2LI_ERR_NO_DEVICE;
    if (n == 0) return 0;
    hipStream_t ps = private_stream();
    std::lock_guard<std::mutex> lock(map->mu);
    int rc = setup_segments(L, 1, &n, ps);
    if (rc != TC2LI_OK) return rc;
    TC2LI_HIP_CHECK(copy_sync(L->d_down.p, feats_down_body, (size_t)n * sizeof(PointXYZINormal), hipMemcpyHostToDevice, ps));
    TC2LI_HIP_CHECK(copy_sync(L->d_down_count.p, &n, sizeof(int), hipMemcpyHostToDevice, ps));
    TC2LI_HIP_CHECK(L->d_eskf_partial.ensure((size_t)((n + 255) / 256) * kEskfOutSize));
    TC2LI_HIP_CHECK(L->h_eskf_out.ensure(kEskfOutSize));
    L->last_down.assign(1, n);
    L->last_map.assign(1, {map, map->generation});
    const PointXYZINormal* body = reinterpret_cast<const PointXYZINormal*>(feats_down_body);
    EskfRun run;
    run.begin(x, P529);
    for (int i = -1; i < maximum_iter && !run.done; i++) {
        // ---- h_share_model at the current state ----
        const tc2li_lidar_state ls = run.lidar_state();
        if (run.converge) {
            rc = run_features(L, L->d_down.p, L->d_down_count.p, &map, &ls, ps);
            if (rc != TC2LI_OK) return rc;
            ++run.searches;
        } else {
            TC2LI_HIP_CHECK(hipMemcpyAsync(L->d_states.p, &ls, sizeof(LidarStateDev), hipMemcpyHostToDevice, ps));
            launch_eskf_refit(map->grid, L->d_down.p, n, L->d_states.p, L->d_nearest_idx.p, L->d_world.p, L->d_selected.p, L->d_normvec.p, ps);
        }
        launch_eskf_normal(L->d_down.p, n, L->d_states.p, L->d_selected.p, L->d_normvec.p, extrinsic_est_en, L->d_eskf_partial.p, L->h_eskf_out.p, ps);
        TC2LI_HIP_CHECK(hipGetLastError());
        TC2LI_HIP_CHECK(hipStreamSynchronize(ps));
        run.step(i, maximum_iter, L->h_eskf_out.p, n, body, R, limit23, extrinsic_est_en, [&](uint8_t* sel, PointXYZINormal* nv) {
            return copy_sync(sel, L->d_selected.p, n, hipMemcpyDeviceToHost, ps) == hipSuccess &&
                   copy_sync(nv, L->d_normvec.p, (size_t)n * sizeof(PointXYZINormal), hipMemcpyDeviceToHost, ps) == hipSuccess;
        });
        if (run.rc < 0) return run.rc;
    }
    if (!run.done) memcpy(P529, run.P.a, sizeof(run.P.a));  // maximum_iter = 0: no evaluation
    run.write_stats(stats);
    return run.effct;
}

}  // extern "C"
namespace {
WorkerPool& lidar_pool() { return named_pool(kPoolLidar); }
}  // namespace
extern "C" {

int tc2li_device_time_sort(tc2li_lidar* L, const tc2li_point* points, int n, int depth_limit, int32_t* perm) {
    if (!L || n < 0 || (n > 0 && (!points || !perm))) { set_error("tc2li_device_time_sort: invalid argument"); return TC2LI_ERR_INVALID; }
    if (n == 0) return 0;
    hipStream_t ps = private_stream();
    int rc = setup_segments(L, 1, &n, ps);
    if (rc != TC2LI_OK) return rc;
    const size_t T = L->total;
    TC2LI_HIP_CHECK(L->d_perm.ensure(T)); TC2LI_HIP_CHECK(L->d_sort_fallback.ensure(2 * (size_t)L->max_scans)); TC2LI_HIP_CHECK(L->d_sort_ranges.ensure(T));
    TC2LI_HIP_CHECK(copy_sync(L->d_pre.p, points, (size_t)n * sizeof(PointXYZINormal), hipMemcpyHostToDevice, ps));
    TC2LI_HIP_CHECK(copy_sync(L->d_pre_count.p, &n, sizeof(int), hipMemcpyHostToDevice, ps));
    launch_time_sort(L->d_pre.p, L->d_pre_count.p, L->d_slots.p, 1, L->d_nearest_d.p, L->d_nearest_idx.p, reinterpret_cast<int*>(L->d_nearest_d.p + T),
                     L->d_selected.p, T, L->d_perm.p, L->d_sort_fallback.p, L->d_sort_ranges.p, L->d_sort_fallback.p + L->max_scans, depth_limit, false, ps);
    TC2LI_HIP_CHECK(hipGetLastError());
    int fb = 0;
    TC2LI_HIP_CHECK(copy_sync(&fb, L->d_sort_fallback.p, sizeof(int), hipMemcpyDeviceToHost, ps));
    TC2LI_HIP_CHECK(copy_sync(perm, L->d_perm.p, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, ps));
    return fb;
}

// LidarInertialProcess for a batch of sequences (LidarFrontEnd.cpp:615-785: sync_packages -> ImuProcess::Process (forward propagation,
// UndistortPcl) -> VoxelGrid -> update_iterated_dyn_share_modified with h_share_model): one launch per phase for all scans, the scans'
// iterated updates in lock step (an iteration's neighbour search for the scans whose last step converged, the re-evaluation of the kept
// neighbours for the others, the normal equations of all of them; then the 23 x 23 algebra of every scan on the host pool).  Same
// kernels bodies and host steps as the one-scan entry points: a scan gives the same state alone and in a batch.
// Preprocess::process of every raw scan and the permutation UndistortPcl's std::sort(time_list) leaves: the part of LidarInertialProcess
// that depends on the scan alone (the reference runs Preprocess::process in the scan callback, LidarFrontEnd.cpp:253, ahead of the thread
// that consumes lidar_buffer).  On return d_pre / d_pre_count / d_perm hold the scans and h_counts[0..S) their sizes.
// (two halves: everything queued; the wait and the scans the host has to sort -- the one-call form propagates the IMU states in between)
static int inertial_prepare_launch(tc2li_lidar* L, int S, const tc2li_velodyne_point* dev_raw, const int32_t* raw_offsets, int point_filter_num, double blind,
                                   float time_unit_scale, hipStream_t st) {
    const size_t T = L->total;
    L->prepared_scans = 0;
    std::vector<int> upper(S);
    for (int s = 0; s < S; ++s) upper[s] = raw_offsets[s + 1] - raw_offsets[s];
    int rc = setup_segments(L, S, upper.data(), st, raw_offsets);
    if (rc != TC2LI_OK) return rc;
    TC2LI_HIP_CHECK(hipMemcpyAsync(L->d_raw_count.p, upper.data(), S * sizeof(int), hipMemcpyHostToDevice, st));
    bool keys_ready = false;  // the time sort's key array filled by the preprocess pass itself (the batch form)
    rc = run_preprocess(L, (const VelodynePoint*)dev_raw, point_filter_num, blind, time_unit_scale, st, L->d_nearest_d.p, &keys_ready);
    if (rc != TC2LI_OK) return rc;
    TC2LI_HIP_CHECK(L->d_perm.ensure(T)); TC2LI_HIP_CHECK(L->d_sort_fallback.ensure(2 * (size_t)L->max_scans)); TC2LI_HIP_CHECK(L->d_sort_ranges.ensure(T));
    TC2LI_HIP_CHECK(L->h_sort_flags.ensure(L->max_scans));
    // the time sort: std::sort's permutation, replayed on the device
    const char* depth_env = getenv("TC2LI_TEST_SORT_DEPTH");  // tests: a small depth limit sends scans through the host fallback
    launch_time_sort(L->d_pre.p, L->d_pre_count.p, L->d_slots.p, S, L->d_nearest_d.p, L->d_nearest_idx.p, reinterpret_cast<int*>(L->d_nearest_d.p + T),
                     L->d_selected.p, T, L->d_perm.p, L->d_sort_fallback.p, L->d_sort_ranges.p, L->d_sort_fallback.p + L->max_scans, depth_env ? atoi(depth_env) : -1, keys_ready, st);
    int* hc = L->h_counts.p;
    int* h_fallback = L->h_sort_flags.p;
    TC2LI_HIP_CHECK(hipGetLastError());
    TC2LI_HIP_CHECK(hipMemcpyAsync(hc, L->d_pre_count.p, S * sizeof(int), hipMemcpyDeviceToHost, st));
    TC2LI_HIP_CHECK(hipMemcpyAsync(h_fallback, L->d_sort_fallback.p, S * sizeof(int), hipMemcpyDeviceToHost, st));
    return TC2LI_OK;
}
static int inertial_prepare_finish(tc2li_lidar* L, int S, hipStream_t st) {
    const int* hc = L->h_counts.p;
    const int* h_fallback = L->h_sort_flags.p;
    TC2LI_HIP_CHECK(stream_wait_blocking(st));
    for (int s = 0; s < S; ++s) {
        if (!h_fallback[s]) continue;
        // the recursion reached std::sort's depth limit (heap sort there): this scan is sorted on the host, like tc2li_lidar_undistort
        const int n = hc[s];
        std::vector<PointXYZINormal> pts(n);
        TC2LI_HIP_CHECK(copy_sync(pts.data(), L->d_pre.p + (size_t)s * L->cap, (size_t)n * sizeof(PointXYZINormal), hipMemcpyDeviceToHost, st));
        struct Rec { float t; int idx; };
        std::vector<Rec> rec(n);
        for (int i = 0; i < n; ++i) rec[i] = Rec{pts[i].curvature, i};
        std::sort(rec.begin(), rec.end(), [](const Rec& a, const Rec& b) { return a.t < b.t; });
        std::vector<int> perm(n);
        for (int i = 0; i < n; ++i) perm[i] = rec[i].idx;
        TC2LI_HIP_CHECK(copy_sync(L->d_perm.p + (size_t)s * L->cap, perm.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, st));
    }
    return TC2LI_OK;
}

int tc2li_lidar_inertial_prepare_batch(tc2li_lidar* L, int n_scans, const tc2li_velodyne_point* dev_raw, const int32_t* raw_offsets, int point_filter_num,
                                       double blind, float time_unit_scale, void* stream_) {
    if (!L || n_scans < 0 || n_scans > (L ? L->max_scans : 0) || !dev_raw || !raw_offsets || point_filter_num < 1) {
        set_error("tc2li_lidar_inertial_prepare_batch: invalid argument");
        return TC2LI_ERR_INVALID;
    }
    if (n_scans == 0) { L->prepared_scans = 0; return 0; }
    if (!device_ready()) return TC2LI_ERR_NO_DEVICE;
    int rc = inertial_prepare_launch(L, n_scans, dev_raw, raw_offsets, point_filter_num, blind, time_unit_scale, (hipStream_t)stream_);
    if (rc == TC2LI_OK) rc = inertial_prepare_finish(L, n_scans, (hipStream_t)stream_);
    if (rc != TC2LI_OK) return rc;
    L->prepared_scans = n_scans;
    return n_scans;
}

int tc2li_lidar_inertial_frontend_batch(tc2li_lidar* L, int n_scans, const tc2li_velodyne_point* dev_raw, const int32_t* raw_offsets,
                                        int point_filter_num, double blind, float time_unit_scale, float leaf, tc2li_lidar_map* const* maps,
                                        tc2li_lidar_inertial_scan* scans, const double cov12[12], double R, int maximum_iter, const double* limit23,
                                        int extrinsic_est_en, void* stream_) {
    using namespace eskf;
    if (!L || n_scans < 0 || n_scans > (L ? L->max_scans : 0) || (dev_raw && !raw_offsets) || !maps || !scans || !cov12 || (dev_raw && point_filter_num < 1) ||
        !(leaf > 0) || !(R > 0) || maximum_iter < 0 || !limit23) {
        set_error("tc2li_lidar_inertial_frontend_batch: invalid argument");
        return TC2LI_ERR_INVALID;
    }
    if (n_scans == 0) return 0;
    if (!device_ready()) return TC2LI_ERR_NO_DEVICE;
    for (int s = 0; s < n_scans; ++s) {
        if (!maps[s] || !scans[s].P || scans[s].n_imu < 1 || !scans[s].imu) { set_error("tc2li_lidar_inertial_frontend_batch: scan %d: NULL map / P / IMU samples", s); return TC2LI_ERR_INVALID; }
        if (scans[s].n_imu + 1 > kMaxImuPoses) { set_error("more than %d IMU poses in one scan", kMaxImuPoses); return TC2LI_ERR_CAPACITY; }
        memset(&scans[s].stats, 0, sizeof(scans[s].stats));
        scans[s].n_preprocessed = scans[s].n_downsampled = 0;
    }
    hipStream_t st = (hipStream_t)stream_;
    MapLocks locks(maps, n_scans);
    const int S = n_scans;
    const size_t T = L->total;
    // ---- Preprocess::process of every raw scan and the order of UndistortPcl's time sort: prepared ahead, or now ----
    int rc = TC2LI_OK;
    if (dev_raw) {
        rc = inertial_prepare_launch(L, S, dev_raw, raw_offsets, point_filter_num, blind, time_unit_scale, st);  // (the host propagates meanwhile)
        if (rc != TC2LI_OK) return rc;
    } else if (L->prepared_scans != S) {
        set_error("tc2li_lidar_inertial_frontend_batch: dev_raw is NULL and the handle holds %d prepared scans, not %d", L->prepared_scans, S);
        return TC2LI_ERR_INVALID;
    }
    L->prepared_scans = 0;  // consumed
    // ---- work space of the batch (allocated on first use) ----
    const size_t rows = T / 256;
    TC2LI_HIP_CHECK(L->d_imu_poses.ensure((size_t)L->max_scans * kMaxImuPoses));
    TC2LI_HIP_CHECK(L->d_n_poses.ensure(L->max_scans)); TC2LI_HIP_CHECK(L->d_scan_list.ensure(L->max_scans));
    TC2LI_HIP_CHECK(L->d_blocks_a.ensure(T / kSegBlock)); TC2LI_HIP_CHECK(L->d_blocks_b.ensure(T / kSegBlock)); TC2LI_HIP_CHECK(L->d_blocks_c.ensure(T / kSegBlock));
    TC2LI_HIP_CHECK(L->d_eskf_partial.ensure(rows * kEskfOutSize)); TC2LI_HIP_CHECK(L->h_eskf_out.ensure((size_t)L->max_scans * kEskfOutSize));
    TC2LI_HIP_CHECK(L->h_batch.ensure((size_t)L->max_scans * (kMaxImuPoses * sizeof(Pose6DDev) + sizeof(LidarStateDev) + 4 * sizeof(int)) + 3 * (T / kSegBlock) * sizeof(SegBlock)));
    // pinned staging: [poses S x 64][states S][n_poses S][fallback S][scan list S][block lists 3 x NB]
    uint8_t* hb = L->h_batch.p;
    Pose6DDev* h_poses = (Pose6DDev*)hb; hb += (size_t)L->max_scans * kMaxImuPoses * sizeof(Pose6DDev);
    LidarStateDev* h_states = (LidarStateDev*)hb; hb += (size_t)L->max_scans * sizeof(LidarStateDev);
    int* h_n_poses = (int*)hb; hb += (size_t)L->max_scans * sizeof(int);
    hb += (size_t)L->max_scans * sizeof(int);
    int* h_scan_list = (int*)hb; hb += (size_t)L->max_scans * 2 * sizeof(int);
    SegBlock* h_blocks_a = (SegBlock*)hb; SegBlock* h_blocks_b = h_blocks_a + T / kSegBlock; SegBlock* h_blocks_c = h_blocks_b + T / kSegBlock;
    // ---- forward propagation of every sequence on the host (IMU_Processing.cpp:176-233; esekf::predict with the covariance) ----
    Mat<kW, kW> Q = Mat<kW, kW>::zero();
    for (int k = 0; k < kW; ++k) Q(k, k) = cov12[k];
    std::vector<int> prc(S, 0);
    lidar_pool().parallel_for(S, [&](int s) {
        tc2li_lidar_inertial_scan& sc = scans[s];
        Cov P;
        memcpy(P.a, sc.P, sizeof(P.a));
        static_assert(sizeof(tc2li_imu_pose6d) == sizeof(Pose6DDev), "ABI layout");
        const int np = imu_propagate_impl(&sc.state, &P, &Q, sc.imu, sc.n_imu, sc.pcl_beg_time, sc.pcl_end_time, sc.last_lidar_end_time, sc.acc_scale,
                                          sc.acc_s_last, sc.angvel_last, reinterpret_cast<tc2li_imu_pose6d*>(h_poses + (size_t)s * kMaxImuPoses), kMaxImuPoses);
        prc[s] = np;
        if (np < 0) return;
        memcpy(sc.P, P.a, sizeof(P.a));
        h_n_poses[s] = np;
        memcpy(h_states[s].rot, sc.state.rot, 72); memcpy(h_states[s].pos, sc.state.pos, 24);
        memcpy(h_states[s].off_r, sc.state.offset_R_L_I, 72); memcpy(h_states[s].off_t, sc.state.offset_T_L_I, 24);
    });
    for (int s = 0; s < S; ++s) if (prc[s] < 0) return prc[s];
    TC2LI_HIP_CHECK(hipMemcpyAsync(L->d_imu_poses.p, h_poses, (size_t)S * kMaxImuPoses * sizeof(Pose6DDev), hipMemcpyHostToDevice, st));
    TC2LI_HIP_CHECK(hipMemcpyAsync(L->d_n_poses.p, h_n_poses, S * sizeof(int), hipMemcpyHostToDevice, st));
    TC2LI_HIP_CHECK(hipMemcpyAsync(L->d_states.p, h_states, S * sizeof(LidarStateDev), hipMemcpyHostToDevice, st));
    // ---- UndistortPcl: the compensation, in the order of the time sort ----
    if (dev_raw) {
        rc = inertial_prepare_finish(L, S, st);
        if (rc != TC2LI_OK) return rc;
    }
    int* hc = L->h_counts.p;
    for (int s = 0; s < S; ++s) scans[s].n_preprocessed = hc[s];
    // the compensated scans land in d_cloud_ori (free until a selection is compacted), the voxel filter reads them there
    launch_undistort_batch(L->d_pre.p, L->d_perm.p, L->d_pre_count.p, L->d_slots.p, L->d_blocks.p, (int)L->blocks.size(), L->d_imu_poses.p, L->d_n_poses.p,
                           L->d_states.p, L->d_cloud_ori.p, st);
    rc = run_voxel(L, L->d_cloud_ori.p, L->d_pre_count.p, leaf, st, S >= 32);
    if (rc != TC2LI_OK) return rc;
    TC2LI_HIP_CHECK(hipMemcpyAsync(hc + S, L->d_down_count.p, S * sizeof(int), hipMemcpyDeviceToHost, st));
    TC2LI_HIP_CHECK(hipMemcpyAsync(hc + 3 * S, L->d_status.p, sizeof(int), hipMemcpyDeviceToHost, st));
    TC2LI_HIP_CHECK(stream_wait_blocking(st));
    if (hc[3 * S]) { TC2LI_HIP_CHECK(memset_sync(L->d_status.p, 0, sizeof(int), st)); set_error("more than 32768 occupied voxels in one scan"); return TC2LI_ERR_CAPACITY; }
    L->last_down.assign(hc + S, hc + 2 * S);
    L->last_map.resize(S);
    for (int s = 0; s < S; ++s) L->last_map[s] = {maps[s], maps[s]->generation};
    L->last_sel.assign(S, 0);
    L->timed = false;
    // ---- the iterated update, all scans in lock step ----
    std::vector<MapGrid> grids(S);
    for (int s = 0; s < S; ++s) grids[s] = maps[s]->grid;
    TC2LI_HIP_CHECK(hipMemcpyAsync(L->d_grids.p, grids.data(), S * sizeof(MapGrid), hipMemcpyHostToDevice, st));
    std::vector<EskfRun> run(S);
    std::vector<int> nd(S);
    for (int s = 0; s < S; ++s) {
        nd[s] = hc[S + s];
        scans[s].n_downsampled = nd[s];
        run[s].begin(&scans[s].state, scans[s].P);
        if (nd[s] == 0) run[s].done = true;  // tc2li_lidar_eskf_update returns at once for an empty scan
    }
    std::vector<int> act, search, refit;
    for (int i = -1; i < maximum_iter; ++i) {
        act.clear(); search.clear(); refit.clear();
        for (int s = 0; s < S; ++s) if (!run[s].done) { act.push_back(s); (run[s].converge ? search : refit).push_back(s); }
        if (act.empty()) break;
        int na = 0, nb = 0, nc = 0;
        for (int s : act) {
            const tc2li_lidar_state ls = run[s].lidar_state();
            memcpy(&h_states[s], &ls, sizeof(LidarStateDev));
            const int blocks_of = (nd[s] + kSegBlock - 1) / kSegBlock;
            for (int b = 0; b < blocks_of; ++b) {
                const SegBlock sb{s, b * kSegBlock};
                h_blocks_c[nc++] = sb;
                if (run[s].converge) h_blocks_a[na++] = sb; else h_blocks_b[nb++] = sb;
            }
        }
        for (size_t k = 0; k < act.size(); ++k) h_scan_list[k] = act[k];
        TC2LI_HIP_CHECK(hipMemcpyAsync(L->d_states.p, h_states, S * sizeof(LidarStateDev), hipMemcpyHostToDevice, st));
        TC2LI_HIP_CHECK(hipMemcpyAsync(L->d_scan_list.p, h_scan_list, act.size() * sizeof(int), hipMemcpyHostToDevice, st));
        if (na) TC2LI_HIP_CHECK(hipMemcpyAsync(L->d_blocks_a.p, h_blocks_a, na * sizeof(SegBlock), hipMemcpyHostToDevice, st));
        if (nb) TC2LI_HIP_CHECK(hipMemcpyAsync(L->d_blocks_b.p, h_blocks_b, nb * sizeof(SegBlock), hipMemcpyHostToDevice, st));
        TC2LI_HIP_CHECK(hipMemcpyAsync(L->d_blocks_c.p, h_blocks_c, nc * sizeof(SegBlock), hipMemcpyHostToDevice, st));
        if (na)
            launch_knn_plane(L->d_grids.p, L->d_down.p, L->d_down_count.p, L->d_slots.p, L->d_blocks_a.p, na, L->d_states.p, L->d_world.p, L->d_selected.p,
                             L->d_normvec.p, L->d_nearest_idx.p, L->d_nearest_d.p, L->d_nfound.p, L->d_hard_count.p, L->d_hard_list.p, st);
        launch_eskf_refit_batch(L->d_grids.p, L->d_down.p, L->d_down_count.p, L->d_slots.p, L->d_blocks_b.p, nb, L->d_states.p, L->d_nearest_idx.p,
                                L->d_world.p, L->d_selected.p, L->d_normvec.p, st);
        launch_eskf_normal_batch(L->d_down.p, L->d_down_count.p, L->d_slots.p, L->d_blocks_c.p, nc, L->d_states.p, L->d_selected.p, L->d_normvec.p,
                                 extrinsic_est_en, L->d_eskf_partial.p, L->d_scan_list.p, (int)act.size(), L->h_eskf_out.p, st);
        TC2LI_HIP_CHECK(hipGetLastError());
        TC2LI_HIP_CHECK(stream_wait_blocking(st));
        for (int s : search) ++run[s].searches;
        lidar_pool().parallel_for((int)act.size(), [&](int k) {
            const int s = act[k];
            EskfRun& r = run[s];
            const size_t base = (size_t)s * L->cap;
            std::vector<PointXYZINormal> body;  // only fetched when there are fewer rows than states
            const double* o = L->h_eskf_out.p + (size_t)s * kEskfOutSize;
            if ((int)o[157] >= 1 && (int)o[157] < kN) {
                body.resize(nd[s]);
                if (copy_sync(body.data(), L->d_down.p + base, (size_t)nd[s] * sizeof(PointXYZINormal), hipMemcpyDeviceToHost, private_stream()) != hipSuccess) { r.rc = TC2LI_ERR_HIP; r.done = true; return; }
            }
            r.step(i, maximum_iter, o, nd[s], body.data(), R, limit23, extrinsic_est_en, [&](uint8_t* sel, PointXYZINormal* nv) {
                hipStream_t ps = private_stream();
                return copy_sync(sel, L->d_selected.p + base, nd[s], hipMemcpyDeviceToHost, ps) == hipSuccess &&
                       copy_sync(nv, L->d_normvec.p + base, (size_t)nd[s] * sizeof(PointXYZINormal), hipMemcpyDeviceToHost, ps) == hipSuccess;
            });
        });
        for (int s : act) if (run[s].rc < 0) return run[s].rc;
    }
    for (int s = 0; s < S; ++s) {
        if (!run[s].done && nd[s] > 0) memcpy(scans[s].P, run[s].P.a, sizeof(run[s].P.a));  // maximum_iter = 0
        run[s].write_stats(&scans[s].stats);
    }
    return n_scans;
}

int tc2li_lidar_map_create(tc2li_lidar_map** out) {
    if (!out) return TC2LI_ERR_INVALID;
    if (!device_ready()) return TC2LI_ERR_NO_DEVICE;
    tc2li_lidar_map* m = new tc2li_lidar_map();
    hipStream_t ps = private_stream();
    int rc = rebuild_grid(m, ps);
    if (rc != TC2LI_OK) { delete m; return rc; }
    *out = m;
    return TC2LI_OK;
}
void tc2li_lidar_map_destroy(tc2li_lidar_map* m) { delete m; }
int tc2li_lidar_map_size(const tc2li_lidar_map* m) {
    if (!m) return TC2LI_ERR_INVALID;
    std::lock_guard<std::mutex> lock(m->mu);
    return m->n;
}

static int map_append(tc2li_lidar_map* m, const tc2li_point* pts, int n, bool reset) {
    if (!m || n < 0 || (n > 0 && !pts)) { set_error("tc2li_lidar_map: invalid argument"); return TC2LI_ERR_INVALID; }
    hipStream_t ps = private_stream();
    std::lock_guard<std::mutex> lock(m->mu);
    const int old = reset ? 0 : m->n;
    if ((size_t)old + n > m->d_points.n) {
        DevBuf<PointXYZINormal> bigger;
        TC2LI_HIP_CHECK(bigger.alloc(((size_t)old + n) * 3 / 2 + 1024));
        if (old) TC2LI_HIP_CHECK(copy_sync(bigger.p, m->d_points.p, (size_t)old * sizeof(PointXYZINormal), hipMemcpyDeviceToDevice, ps));
        std::swap(bigger.p, m->d_points.p);
        std::swap(bigger.n, m->d_points.n);
    }
    if (n) TC2LI_HIP_CHECK(copy_sync(m->d_points.p + old, pts, (size_t)n * sizeof(PointXYZINormal), hipMemcpyHostToDevice, ps));
    for (int i = 0; i < n; ++i) {
        const float c[3] = {pts[i].x, pts[i].y, pts[i].z};
        if (!std::isfinite(c[0]) || !std::isfinite(c[1]) || !std::isfinite(c[2])) { set_error("non-finite map point"); return TC2LI_ERR_INVALID; }
        for (int a = 0; a < 3; ++a) {
            if (old + i == 0) { m->lo[a] = m->hi[a] = c[a]; }
            m->lo[a] = std::min(m->lo[a], c[a]);
            m->hi[a] = std::max(m->hi[a], c[a]);
        }
    }
    m->n = old + n;
    ++m->generation;
    m->rebuild_mode = 0; m->have_remap = false;  // Build / Add_Points from the caller's list: the grid is made from the points
    int rc = rebuild_grid(m, ps);
    if (rc != TC2LI_OK) return rc;
    return m->n;
}
int tc2li_lidar_map_stats(const tc2li_lidar_map* m, int32_t* out, int capacity) {
    if (!m || !out || capacity < 6) { set_error("tc2li_lidar_map_stats: invalid argument"); return TC2LI_ERR_INVALID; }
    std::lock_guard<std::mutex> lock(m->mu);
    out[0] = m->n; out[1] = m->grid.n_slots; out[2] = m->n_grid_builds; out[3] = m->n_grid_updates; out[4] = m->tombstones;
    out[5] = m->grid.nx * m->grid.ny * m->grid.nz;
    return 6;
}
int tc2li_lidar_map_grid_download(const tc2li_lidar_map* m, int32_t* cells, int32_t* indices, int capacity) {
    if (!m || capacity < 0 || (capacity > 0 && (!cells || !indices))) { set_error("tc2li_lidar_map_grid_download: invalid argument"); return TC2LI_ERR_INVALID; }
    std::lock_guard<std::mutex> lock(m->mu);
    if (!m->grid_valid) { set_error("tc2li_lidar_map_grid_download: the map has no valid grid"); return TC2LI_ERR_INVALID; }
    const MapGrid& g = m->grid;
    const int rows = g.ny * g.nz, nseg = rows * g.nsx, stride = kMapSegStride;
    std::vector<float4> ent((size_t)std::max(g.n_slots, 1));
    std::vector<int> rs((size_t)nseg * stride + 1);
    std::vector<PointXYZINormal> pts((size_t)std::max(m->n, 1));
    hipStream_t ps = private_stream();
    TC2LI_HIP_CHECK(copy_sync(ent.data(), g.pts, (size_t)g.n_slots * sizeof(float4), hipMemcpyDeviceToHost, ps));
    TC2LI_HIP_CHECK(copy_sync(rs.data(), g.bucket_start, rs.size() * sizeof(int), hipMemcpyDeviceToHost, ps));
    if (m->n) TC2LI_HIP_CHECK(copy_sync(pts.data(), m->d_points.p, (size_t)m->n * sizeof(PointXYZINormal), hipMemcpyDeviceToHost, ps));
    auto idx_of = [](const float4& e) { int i; memcpy(&i, &e.w, 4); return i; };
    int n_live = 0;
    if (rs[(size_t)nseg * stride] != g.n_slots) { set_error("grid: the last start is %d, not n_slots = %d", rs[(size_t)nseg * stride], g.n_slots); return TC2LI_ERR_INVALID; }
    for (int q = 0; q < nseg; ++q) {
        const int* cs = rs.data() + (size_t)q * stride;
        const int limit = cs[stride], r = q / g.nsx, sx = q - r * g.nsx;
        if (cs[0] > cs[kMapSegCells] || cs[kMapSegCells] > limit) { set_error("grid: segment %d spans [%d, %d) beyond its room up to %d", q, cs[0], cs[kMapSegCells], limit); return TC2LI_ERR_INVALID; }
        for (int j = 0; j < kMapSegCells; ++j) {
            const int ix = sx * kMapSegCells + j;
            if (cs[j] > cs[j + 1]) { set_error("grid: segment %d cell %d starts at %d behind the next cell's %d", q, j, cs[j], cs[j + 1]); return TC2LI_ERR_INVALID; }
            if (ix >= g.nx && cs[j] != cs[j + 1]) { set_error("grid: segment %d holds entries in cell %d beyond the row's %d cells", q, ix, g.nx); return TC2LI_ERR_INVALID; }
            for (int k = cs[j]; k < cs[j + 1]; ++k) {
                const int i = idx_of(ent[k]);
                if (i < 0) continue;
                if (i >= m->n) { set_error("grid: entry %d names point %d of %d", k, i, m->n); return TC2LI_ERR_INVALID; }
                const PointXYZINormal& p = pts[i];
                const int cx = (int)std::floor(p.x * g.inv_cell) - g.x0, cy = (int)std::floor(p.y * g.inv_cell) - g.y0, cz = (int)std::floor(p.z * g.inv_cell) - g.z0;
                if (p.x != ent[k].x || p.y != ent[k].y || p.z != ent[k].z || (cz * g.ny + cy) * g.nx + cx != r * g.nx + ix) {
                    set_error("grid: entry %d (point %d) stands in cell %d, its coordinates say %d", k, i, r * g.nx + ix, (cz * g.ny + cy) * g.nx + cx);
                    return TC2LI_ERR_INVALID;
                }
                if (n_live < capacity) { cells[n_live] = r * g.nx + ix; indices[n_live] = i; }
                ++n_live;
            }
        }
        for (int k = cs[kMapSegCells]; k < limit; ++k)
            if (idx_of(ent[k]) >= 0) { set_error("grid: a live entry at %d in the unused room of segment %d", k, q); return TC2LI_ERR_INVALID; }
    }
    return n_live;
}
int tc2li_lidar_map_build(tc2li_lidar_map* m, const tc2li_point* pts, int n) { return map_append(m, pts, n, true); }
int tc2li_lidar_map_add(tc2li_lidar_map* m, const tc2li_point* pts, int n) { return map_append(m, pts, n, false); }

namespace {
// map_incremental for a batch of (scan slot, map) pairs: one launch per phase for all of them, one synchronisation between the
// compaction and the grid rebuild (the host sizes the new grids) and one at the end.
int map_incremental_impl(tc2li_lidar* L, int n_tasks, const int32_t* scans, tc2li_lidar_map* const* maps, const tc2li_lidar_state* states,
                         int ekf_inited, double fs, int32_t* n_to_add, int32_t* n_no_need, int32_t* map_sizes, hipStream_t st) {
    for (int i = 0; i < n_tasks; ++i) {
        if (!maps[i] || scans[i] < 0 || scans[i] >= (int)L->last_down.size()) {
            set_error("tc2li_lidar_map_incremental: invalid argument (the scan slots must come from the last feature extraction)");
            return TC2LI_ERR_INVALID;
        }
        if (n_to_add) n_to_add[i] = 0;
        if (n_no_need) n_no_need[i] = 0;
    }
    {
        std::vector<const tc2li_lidar_map*> seen(maps, maps + n_tasks);
        std::sort(seen.begin(), seen.end());
        if (std::adjacent_find(seen.begin(), seen.end()) != seen.end()) { set_error("tc2li_lidar_map_incremental_batch: a map appears twice in one batch"); return TC2LI_ERR_INVALID; }
        std::vector<int32_t> slots_seen(scans, scans + n_tasks);
        std::sort(slots_seen.begin(), slots_seen.end());
        if (std::adjacent_find(slots_seen.begin(), slots_seen.end()) != slots_seen.end()) { set_error("tc2li_lidar_map_incremental_batch: a scan slot appears twice in one batch"); return TC2LI_ERR_INVALID; }
    }
    MapLocks locks(maps, n_tasks);
    // the neighbours of a scan slot are indices into the map its feature extraction searched: that map, unchanged since (read under the
    // maps' locks: another thread's Build / Add_Points bumps the generation under them)
    for (int i = 0; i < n_tasks; ++i) {
        const auto& lm = L->last_map[scans[i]];
        if (lm.first != maps[i] || lm.second != maps[i]->generation) {
            set_error("tc2li_lidar_map_incremental: map %d is not the map scan slot %d was matched against, or it has changed since (Build / Add_Points / "
                      "map_incremental / Delete_Point_Boxes renumber the points the neighbour indices refer to)", i, scans[i]);
            return TC2LI_ERR_INVALID;
        }
    }
    const size_t S = L->max_scans, T = L->total;
    if (L->d_inc_tasks.n < S) {
        TC2LI_HIP_CHECK(L->d_cls.alloc(T)); TC2LI_HIP_CHECK(L->d_noneed.alloc(T));
        TC2LI_HIP_CHECK(L->d_inc_recs.alloc(S * kMapIncMax)); TC2LI_HIP_CHECK(L->d_group_start.alloc(S * (kMapIncMax + 1)));
        TC2LI_HIP_CHECK(L->d_appended.alloc(S * kMapIncMax)); TC2LI_HIP_CHECK(L->d_has_append.alloc(S * kMapIncMax));
        TC2LI_HIP_CHECK(L->d_mapinc_out.alloc(S * kMapIncOut)); TC2LI_HIP_CHECK(L->h_mapinc_out.alloc(S * kMapIncOut));
        TC2LI_HIP_CHECK(L->d_grid_tasks.alloc(S)); TC2LI_HIP_CHECK(L->d_inc_tasks.alloc(S)); TC2LI_HIP_CHECK(L->d_batch_overflow.alloc(1));
        TC2LI_HIP_CHECK(L->d_ins_tasks.alloc(S)); TC2LI_HIP_CHECK(L->d_ins_out.alloc(4 * S)); TC2LI_HIP_CHECK(L->h_ins_out.alloc(4 * S));
    }
    // TC2LI_MAP_ALWAYS_REBUILD=1 (read per call; tests and A/B measurements): every changed grid is rebuilt, as in rounds 1-3
    const char* always_env = getenv("TC2LI_MAP_ALWAYS_REBUILD");
    const bool always_rebuild = always_env && atoi(always_env) != 0;
    std::vector<MapIncTask> tasks;
    std::vector<int> which;  // task -> index in the caller's arrays
    tasks.reserve(n_tasks);
    int max_points = 0, max_map = 0;
    bool any_lean = false, any_flagged = false;
    const char* lean_env = getenv("TC2LI_MAP_COMPACT_LIST");
    const bool no_lean = lean_env && atoi(lean_env) == 0;
    const float ds = (float)fs;  // ikdtree.set_downsample_param(filter_size_map_min): float downsample_size
    for (int i = 0; i < n_tasks; ++i) {
        tc2li_lidar_map* m = maps[i];
        const int scan = scans[i], n = L->last_down[scan];
        if (n == 0) continue;  // nothing to insert: the map stays as it is
        const size_t base = (size_t)scan * L->cap;
        int rc = ensure_deleted(m, std::max(m->n, 1), st);
        if (rc != TC2LI_OK) return rc;
        const int kb = (m->n + 1023) / 1024;
        TC2LI_HIP_CHECK(m->d_keep_counts.ensure(std::max(kb, 1)));
        rc = ensure_point_capacity(m, (size_t)m->n + std::min(n, kMapIncMax) + n + 1, st);
        if (rc != TC2LI_OK) return rc;
        TC2LI_HIP_CHECK(m->d_remap.ensure(std::max(m->n, 1))); TC2LI_HIP_CHECK(m->d_holes.ensure(2 * (size_t)std::max(m->n, 1)));
        MapIncTask t{};
        t.body = L->d_down.p + base; t.nearest_idx = L->d_nearest_idx.p + base * 5; t.nfound = L->d_nfound.p + base;
        t.world = L->d_world.p + base; t.cls = L->d_cls.p + base; t.noneed = L->d_noneed.p + base;
        t.recs = L->d_inc_recs.p + (size_t)scan * kMapIncMax; t.group_start = L->d_group_start.p + (size_t)scan * (kMapIncMax + 1);
        t.appended = L->d_appended.p + (size_t)scan * kMapIncMax; t.has_append = L->d_has_append.p + (size_t)scan * kMapIncMax;
        t.out = L->d_mapinc_out.p + (size_t)tasks.size() * kMapIncOut;
        t.grid = m->grid; t.grid.points = m->d_points.p; t.deleted = m->d_deleted.p; t.keep_counts = m->d_keep_counts.p; t.dst = m->d_points.p;
        t.remap = m->d_remap.p; t.holes = m->d_holes.p; t.batch_overflow = L->d_batch_overflow.p;
        memcpy(&t.st, &states[i], sizeof(LidarStateDev));
        t.fs = fs; t.ds = ds; t.n = n; t.n_map = m->n; t.keep_blocks = kb; t.ekf_inited = ekf_inited; t.has_inc = 1;
        t.fix_grid = m->grid_valid && m->grid.n_slots > 0 && !always_rebuild;
        // round 5: a map whose grid is maintained in place compacts from the LIST of its deletions (k_map_compact_list) instead of three passes
        // over all its points; TC2LI_MAP_COMPACT_LIST=0 (read per call): the flag passes for every map
        t.lean = t.fix_grid && !no_lean ? 1 : 0;
        tasks.push_back(t);
        which.push_back(i);
        max_points = std::max(max_points, n);
        if (!t.lean) max_map = std::max(max_map, m->n);
        any_lean |= t.lean != 0;
        any_flagged |= t.lean == 0;
    }
    const int nt = (int)tasks.size();
    if (nt) {
        TC2LI_HIP_CHECK(hipMemcpyAsync(L->d_inc_tasks.p, tasks.data(), nt * sizeof(MapIncTask), hipMemcpyHostToDevice, st));
        TC2LI_HIP_CHECK(hipMemsetAsync(L->d_batch_overflow.p, 0, sizeof(int), st));
        launch_mapinc_lists(L->d_inc_tasks.p, nt, max_points, st);
        launch_map_compact(L->d_inc_tasks.p, nt, max_map, any_lean, any_flagged, st);
        TC2LI_HIP_CHECK(hipGetLastError());
        TC2LI_HIP_CHECK(hipMemcpyAsync(L->h_mapinc_out.p, L->d_mapinc_out.p, (size_t)nt * kMapIncOut * sizeof(int), hipMemcpyDeviceToHost, st));
        TC2LI_HIP_CHECK(stream_wait_blocking(st));
        {   // a lean map with more deletions than its list holds (out[15]) has not been compacted: once more, through the flag passes
            std::vector<int> again;
            for (int k = 0; k < nt; ++k)
                if (tasks[k].lean && L->h_mapinc_out.p[k * kMapIncOut + 15] && !L->h_mapinc_out.p[k * kMapIncOut + 3]) again.push_back(k);
            if (!again.empty()) {
                std::vector<MapIncTask> redo;
                int redo_max = 0;
                for (int k : again) { tasks[k].lean = 0; redo.push_back(tasks[k]); redo_max = std::max(redo_max, tasks[k].n_map); }
                // (the batch's own tasks are done with -- the stream has been waited for -- so the repeated ones take their place; their `out`
                // pointers are those of the first pass)
                TC2LI_HIP_CHECK(hipMemcpyAsync(L->d_inc_tasks.p, redo.data(), redo.size() * sizeof(MapIncTask), hipMemcpyHostToDevice, st));
                launch_map_compact(L->d_inc_tasks.p, (int)redo.size(), redo_max, false, true, st);
                TC2LI_HIP_CHECK(hipGetLastError());
                TC2LI_HIP_CHECK(hipMemcpyAsync(L->h_mapinc_out.p, L->d_mapinc_out.p, (size_t)nt * kMapIncOut * sizeof(int), hipMemcpyDeviceToHost, st));
                TC2LI_HIP_CHECK(stream_wait_blocking(st));
            }
        }
        for (int k = 0; k < nt; ++k)
            if (L->h_mapinc_out.p[k * kMapIncOut + 3]) {
                // the compaction kernels saw the batch word and touched no map; the deletion marks of the lists are taken back
                for (int q = 0; q < nt; ++q) {
                    tc2li_lidar_map* m = maps[which[q]];
                    if (m->d_deleted.p) TC2LI_HIP_CHECK(hipMemsetAsync(m->d_deleted.p, 0, m->d_deleted.n, st));
                }
                TC2LI_HIP_CHECK(stream_wait_blocking(st));
                set_error("more than %d points in the down-sampled insertion list of one scan", kMapIncMax);
                return TC2LI_ERR_CAPACITY;
            }
        // Every map takes over its new point list; its grid is then brought up to date IN PLACE where it was maintained through the
        // compaction (tombstones for the deleted points, new indices for the moved ones: fix_grid) -- the added points are merged into the
        // rows they fall into (k_map_ins_*) -- and rebuilt otherwise: no grid yet, more added points than the insertion takes, too many
        // tombstones, or a row / the box without room (the insertion says so: out[1]).
        std::vector<tc2li_lidar_map*> rebuild, inserted;
        std::vector<MapInsTask> ins;
        for (int k = 0; k < nt; ++k) {
            const int* o = L->h_mapinc_out.p + k * kMapIncOut;
            tc2li_lidar_map* m = maps[which[k]];
            const bool fixed = tasks[k].fix_grid != 0;
            const bool grid_bad = o[13] != 0;  // the renumbering met an entry that does not fit its map: the grid is not trusted, rebuilt from the points
            if (grid_bad) fprintf(stderr, "tc2li: compaction of map %d found an inconsistent grid (code %d): rebuilt from the point list\n", k, o[13]);
            commit_compaction(m, o, true);
            if (n_to_add) n_to_add[which[k]] = o[0];
            if (n_no_need) n_no_need[which[k]] = o[2];
            const int kept = o[4], added = o[5] + o[2];
            if (fixed) { m->have_remap = false; m->rebuild_mode = 2; }  // the old grid's entries carry the new numbering already
            if (grid_bad) { m->have_remap = false; m->rebuild_mode = 0; rebuild.push_back(m); continue; }
            if (!fixed || added > kMapInsMax) { rebuild.push_back(m); continue; }
            m->grid.n_points = m->n;
            if (added == 0) {
                if (m->tombstones > std::max(1024, m->n / 8)) { m->remap_n_kept = m->n; rebuild.push_back(m); }  // mode 2: the old grid holds every point
                else { m->rebuild_mode = 0; ++m->n_grid_updates; }
                continue;
            }
            TC2LI_HIP_CHECK(m->d_ins_keys.ensure(kMapInsMax)); TC2LI_HIP_CHECK(m->d_ins_rows.ensure(kMapInsMax + 1));
            MapInsTask t{};
            t.g = m->grid; t.g.points = m->d_points.p;
            t.pts = const_cast<float4*>(m->grid.pts); t.row_start = const_cast<int*>(m->grid.bucket_start);
            t.keys = m->d_ins_keys.p; t.row_list = m->d_ins_rows.p; t.out = L->d_ins_out.p + 4 * ins.size();
            t.first = kept; t.count = added;
            ins.push_back(t);
            inserted.push_back(m);
        }
        if (!ins.empty()) {
            const int ni = (int)ins.size();
            TC2LI_HIP_CHECK(hipMemcpyAsync(L->d_ins_tasks.p, ins.data(), ni * sizeof(MapInsTask), hipMemcpyHostToDevice, st));
            launch_map_insert(L->d_ins_tasks.p, ni, st);
            TC2LI_HIP_CHECK(hipGetLastError());
            TC2LI_HIP_CHECK(hipMemcpyAsync(L->h_ins_out.p, L->d_ins_out.p, 4 * (size_t)ni * sizeof(int), hipMemcpyDeviceToHost, st));
            TC2LI_HIP_CHECK(stream_wait_blocking(st));
            for (int k = 0; k < ni; ++k) {
                tc2li_lidar_map* m = inserted[k];
                const bool ins_bad = L->h_ins_out.p[4 * k + 3] != 0;  // an index outside its bounds (the kernel skipped it): rebuilt from the points
                if (ins_bad) fprintf(stderr, "tc2li: in-place insertion of map %d found an inconsistent grid (code %d): rebuilt from the point list\n", k, L->h_ins_out.p[4 * k + 3]);
                if (L->h_ins_out.p[4 * k + 1] || ins_bad) { m->have_remap = false; m->rebuild_mode = 0; rebuild.push_back(m); }  // some rows may hold the new points already: from the point list
                else {
                    // the rewritten segments dropped their tombstones; what is left elsewhere is scanned by every search that passes: beyond an
                    // eighth of the map the grid is rebuilt from its own order (mode 2: it holds every point under its new index)
                    ++m->n_grid_updates;
                    m->tombstones = std::max(0, m->tombstones - L->h_ins_out.p[4 * k + 2]);
                    if (m->tombstones > std::max(1024, m->n / 8)) { m->rebuild_mode = 2; m->remap_n_kept = m->n; rebuild.push_back(m); }
                    else m->rebuild_mode = 0;
                }
            }
        }
        if (!rebuild.empty()) {
            const int rc = rebuild_grids(rebuild.data(), (int)rebuild.size(), L->d_grid_tasks, st);
            if (rc != TC2LI_OK) return rc;
        }
    }
    if (map_sizes) for (int i = 0; i < n_tasks; ++i) map_sizes[i] = maps[i]->n;
    return n_tasks;
}
}  // namespace

int tc2li_lidar_map_incremental(tc2li_lidar* L, int scan, tc2li_lidar_map* m, const tc2li_lidar_state* state, int ekf_inited,
                                double filter_size_map_min, int32_t* n_to_add, int32_t* n_no_need, void* stream_) {
    if (!L || !m || !state || !(filter_size_map_min > 0)) { set_error("tc2li_lidar_map_incremental: invalid argument"); return TC2LI_ERR_INVALID; }
    const int32_t sc = scan;
    int32_t size = 0;
    const int rc = map_incremental_impl(L, 1, &sc, &m, state, ekf_inited, filter_size_map_min, n_to_add, n_no_need, &size, (hipStream_t)stream_);
    return rc < 0 ? rc : size;
}

int tc2li_lidar_map_incremental_batch(tc2li_lidar* L, int n, const int32_t* scans, tc2li_lidar_map* const* maps, const tc2li_lidar_state* states,
                                      int ekf_inited, double filter_size_map_min, int32_t* n_to_add, int32_t* n_no_need, int32_t* map_sizes,
                                      void* stream_) {
    if (!L || n < 0 || (n > 0 && (!scans || !maps || !states)) || n > (L ? L->max_scans : 0) || !(filter_size_map_min > 0)) {
        set_error("tc2li_lidar_map_incremental_batch: invalid argument");
        return TC2LI_ERR_INVALID;
    }
    if (n == 0) return 0;
    return map_incremental_impl(L, n, scans, maps, states, ekf_inited, filter_size_map_min, n_to_add, n_no_need, map_sizes, (hipStream_t)stream_);
}

namespace {
// Work space of a batch of box deletions (they have no tc2li_lidar handle to borrow one from): one per calling thread.
struct DeleteBoxesWs {
    DevBuf<MapIncTask> d_tasks;
    DevBuf<MapGridTask> d_grid_tasks;
    DevBuf<float> d_boxes;
    DevBuf<int> d_out, d_overflow;  // d_overflow: the batch word of MapIncTask (always 0 here: a deletion has no insertion list)
    PinnedBuf<int> h_out;
};
DeleteBoxesWs* delete_boxes_ws() { thread_local DeleteBoxesWs ws; return &ws; }
}  // namespace

int tc2li_lidar_map_delete_boxes_batch(int n_maps, tc2li_lidar_map* const* maps, const float* boxes6, const int32_t* box_offsets, int32_t* n_removed,
                                       void* stream_) {
    if (n_maps < 0 || (n_maps > 0 && (!maps || !box_offsets)) || (n_maps > 0 && box_offsets[0] != 0)) {
        set_error("tc2li_lidar_map_delete_boxes_batch: invalid argument");
        return TC2LI_ERR_INVALID;
    }
    if (n_maps == 0) return 0;
    for (int i = 0; i < n_maps; ++i)
        if (!maps[i] || box_offsets[i + 1] < box_offsets[i]) { set_error("tc2li_lidar_map_delete_boxes_batch: invalid argument"); return TC2LI_ERR_INVALID; }
    const int total_boxes = box_offsets[n_maps];
    if (total_boxes > 0 && !boxes6) { set_error("tc2li_lidar_map_delete_boxes_batch: invalid argument"); return TC2LI_ERR_INVALID; }
    {
        std::vector<const tc2li_lidar_map*> seen(maps, maps + n_maps);
        std::sort(seen.begin(), seen.end());
        if (std::adjacent_find(seen.begin(), seen.end()) != seen.end()) { set_error("tc2li_lidar_map_delete_boxes_batch: a map appears twice in one batch"); return TC2LI_ERR_INVALID; }
    }
    hipStream_t st = (hipStream_t)stream_;
    MapLocks locks(maps, n_maps);
    if (n_removed) for (int i = 0; i < n_maps; ++i) n_removed[i] = 0;
    DeleteBoxesWs& ws = *delete_boxes_ws();
    std::vector<MapIncTask> tasks;
    std::vector<int> which;
    int max_map = 0;
    TC2LI_HIP_CHECK(ws.d_boxes.ensure(6 * (size_t)std::max(total_boxes, 1)));
    TC2LI_HIP_CHECK(ws.d_overflow.ensure(1));
    TC2LI_HIP_CHECK(hipMemsetAsync(ws.d_overflow.p, 0, sizeof(int), st));
    for (int i = 0; i < n_maps; ++i) {
        tc2li_lidar_map* m = maps[i];
        const int nb = box_offsets[i + 1] - box_offsets[i];
        if (nb == 0 || m->n == 0) continue;
        const int rc = ensure_deleted(m, m->n, st);
        if (rc != TC2LI_OK) return rc;
        const int kb = (m->n + 1023) / 1024;
        TC2LI_HIP_CHECK(m->d_keep_counts.ensure(kb));
        TC2LI_HIP_CHECK(m->d_remap.ensure(std::max(m->n, 1))); TC2LI_HIP_CHECK(m->d_holes.ensure(2 * (size_t)std::max(m->n, 1)));
        MapIncTask t{};
        t.grid = m->grid; t.grid.points = m->d_points.p;
        t.deleted = m->d_deleted.p; t.keep_counts = m->d_keep_counts.p; t.dst = m->d_points.p; t.remap = m->d_remap.p; t.holes = m->d_holes.p;
        t.batch_overflow = ws.d_overflow.p;
        t.n_map = m->n; t.keep_blocks = kb; t.has_inc = 0;
        t.boxes = ws.d_boxes.p + 6 * (size_t)box_offsets[i]; t.n_boxes = nb;
        tasks.push_back(t);
        which.push_back(i);
        max_map = std::max(max_map, m->n);
    }
    const int nt = (int)tasks.size();
    if (nt == 0) return 0;
    TC2LI_HIP_CHECK(ws.d_tasks.ensure(nt)); TC2LI_HIP_CHECK(ws.d_out.ensure((size_t)nt * kMapIncOut)); TC2LI_HIP_CHECK(ws.h_out.ensure((size_t)nt * kMapIncOut));
    for (int k = 0; k < nt; ++k) tasks[k].out = ws.d_out.p + (size_t)k * kMapIncOut;
    TC2LI_HIP_CHECK(hipMemcpyAsync(ws.d_boxes.p, boxes6, 6 * (size_t)total_boxes * sizeof(float), hipMemcpyHostToDevice, st));
    TC2LI_HIP_CHECK(hipMemcpyAsync(ws.d_tasks.p, tasks.data(), nt * sizeof(MapIncTask), hipMemcpyHostToDevice, st));
    launch_map_mark_boxes(ws.d_tasks.p, nt, max_map, st);
    launch_map_compact(ws.d_tasks.p, nt, max_map, false, true, st);
    TC2LI_HIP_CHECK(hipGetLastError());
    TC2LI_HIP_CHECK(hipMemcpyAsync(ws.h_out.p, ws.d_out.p, (size_t)nt * kMapIncOut * sizeof(int), hipMemcpyDeviceToHost, st));
    TC2LI_HIP_CHECK(stream_wait_blocking(st));  // the uploads above read the caller's and this function's host memory: done here too
    std::vector<tc2li_lidar_map*> changed;
    int removed_total = 0;
    for (int k = 0; k < nt; ++k) {
        tc2li_lidar_map* m = maps[which[k]];
        const int before = m->n;
        if (ws.h_out.p[k * kMapIncOut + 4] == before) continue;  // nothing inside the boxes: points, flags and grid stay as they are
        commit_compaction(m, ws.h_out.p + k * kMapIncOut, false);
        changed.push_back(m);
        if (n_removed) n_removed[which[k]] = before - m->n;
        removed_total += before - m->n;
    }
    if (!changed.empty()) {
        const int rc = rebuild_grids(changed.data(), (int)changed.size(), ws.d_grid_tasks, st);
        if (rc != TC2LI_OK) return rc;
    }
    return removed_total;
}

int tc2li_lidar_map_delete_boxes(tc2li_lidar_map* m, const float* boxes6, int n_boxes, void* stream_) {
    if (!m || n_boxes < 0 || (n_boxes > 0 && !boxes6)) { set_error("tc2li_lidar_map_delete_boxes: invalid argument"); return TC2LI_ERR_INVALID; }
    const int32_t offsets[2] = {0, n_boxes};
    return tc2li_lidar_map_delete_boxes_batch(1, &m, boxes6, offsets, nullptr, stream_);
}

// Tracking::SyncWithLidar / BuildLidarFeat4KeyFrame: the scans' feature clouds (mCurrFeatPoints = laserCloudOri of the front end, still on
// the device) moved by one rigid transform each (LidarFrontEndTools::transformPointCloud), all scans in one launch
int tc2li_lidar_transform_features_batch(tc2li_lidar* L, int n, const int32_t* scans, const float* T7, tc2li_point* out, int capacity, int32_t* n_points,
                                         void* stream_) {
    if (!L || n < 0 || (n > 0 && (!scans || !T7 || !out)) || capacity < 0) { set_error("tc2li_lidar_transform_features_batch: invalid argument"); return TC2LI_ERR_INVALID; }
    if (n == 0) return 0;
    if (!device_ready()) return TC2LI_ERR_NO_DEVICE;
    hipStream_t st = (hipStream_t)stream_;
    std::vector<TransformTask> tasks(n);
    TC2LI_HIP_CHECK(L->d_xform_tasks.ensure(n));
    TC2LI_HIP_CHECK(L->d_xform_out.ensure((size_t)n * std::max(capacity, 1)));
    int max_pts = 0;
    for (int i = 0; i < n; ++i) {
        if (scans[i] < 0 || scans[i] >= (int)L->last_sel.size()) { set_error("scan slot %d is not part of the last tc2li_lidar_frontend_batch", scans[i]); return TC2LI_ERR_INVALID; }
        const int m = L->last_sel[scans[i]];
        if (m > capacity) { set_error("output capacity %d < %d", capacity, m); return TC2LI_ERR_CAPACITY; }
        TransformTask& t = tasks[i];
        t.in = L->d_cloud_ori.p + (size_t)scans[i] * L->cap; t.out = L->d_xform_out.p + (size_t)i * capacity; t.n = m; t.pad_ = 0;
        fill_transform_task(t, T7 + 7 * (size_t)i);
        max_pts = std::max(max_pts, m);
        if (n_points) n_points[i] = m;
    }
    TC2LI_HIP_CHECK(hipMemcpyAsync(L->d_xform_tasks.p, tasks.data(), n * sizeof(TransformTask), hipMemcpyHostToDevice, st));
    launch_transform_points(L->d_xform_tasks.p, n, max_pts, st);
    TC2LI_HIP_CHECK(hipGetLastError());
    for (int i = 0; i < n; ++i)
        if (tasks[i].n)
            TC2LI_HIP_CHECK(hipMemcpyAsync(out + (size_t)i * capacity, tasks[i].out, (size_t)tasks[i].n * sizeof(PointXYZINormal), hipMemcpyDeviceToHost, st));
    TC2LI_HIP_CHECK(stream_wait_blocking(st));
    return n;
}

int tc2li_lidar_map_download(const tc2li_lidar_map* m, tc2li_point* out, int capacity) {
    if (!m || (capacity > 0 && !out)) { set_error("tc2li_lidar_map_download: invalid argument"); return TC2LI_ERR_INVALID; }
    hipStream_t ps = private_stream();
    std::lock_guard<std::mutex> lock(m->mu);
    const int k = std::min(m->n, capacity);
    if (k > 0) TC2LI_HIP_CHECK(copy_sync(out, m->d_points.p, (size_t)k * sizeof(PointXYZINormal), hipMemcpyDeviceToHost, ps));
    return m->n;
}

// lasermap_fov_segment (LidarFrontEnd.cpp:183-231): host logic; returns the boxes to pass to tc2li_lidar_map_delete_boxes
int tc2li_lidar_fov_segment(tc2li_local_map_box* lm, const double pos_lid[3], double cube_len, double det_range, float boxes6[18]) {
    if (!lm || !pos_lid || !boxes6) { set_error("tc2li_lidar_fov_segment: invalid argument"); return TC2LI_ERR_INVALID; }
    const float MOV_THRESHOLD = 1.5f;
    if (!lm->initialized) {
        for (int i = 0; i < 3; i++) { lm->vertex_min[i] = (float)(pos_lid[i] - cube_len / 2.0); lm->vertex_max[i] = (float)(pos_lid[i] + cube_len / 2.0); }
        lm->initialized = 1;
        return 0;
    }
    float edge[3][2];
    bool need_move = false;
    const double thr = (double)MOV_THRESHOLD * det_range;
    for (int i = 0; i < 3; i++) {
        edge[i][0] = (float)std::fabs(pos_lid[i] - lm->vertex_min[i]);
        edge[i][1] = (float)std::fabs(pos_lid[i] - lm->vertex_max[i]);
        if (edge[i][0] <= thr || edge[i][1] <= thr) need_move = true;
    }
    if (!need_move) return 0;
    float nmin[3], nmax[3];
    memcpy(nmin, lm->vertex_min, 12); memcpy(nmax, lm->vertex_max, 12);
    const float mov_dist = (float)std::max((cube_len - 2.0 * MOV_THRESHOLD * det_range) * 0.5 * 0.9, double((float)det_range * (MOV_THRESHOLD - 1)));
    int nb = 0;
    for (int i = 0; i < 3; i++) {
        float* b = boxes6 + 6 * nb;
        memcpy(b, lm->vertex_min, 12); memcpy(b + 3, lm->vertex_max, 12);
        if (edge[i][0] <= thr) {
            nmax[i] -= mov_dist; nmin[i] -= mov_dist;
            b[i] = lm->vertex_max[i] - mov_dist;
            ++nb;
        } else if (edge[i][1] <= thr) {
            nmax[i] += mov_dist; nmin[i] += mov_dist;
            b[3 + i] = lm->vertex_min[i] + mov_dist;
            ++nb;
        }
    }
    memcpy(lm->vertex_min, nmin, 12); memcpy(lm->vertex_max, nmax, 12);
    return nb;
}

// the same for the sensors of n sequences in one call (a batch driver: one lasermap_fov_segment per sequence and scan)
int tc2li_lidar_fov_segment_batch(tc2li_local_map_box* local_maps, const double* pos_lid3, int n, double cube_len, double det_range, float* boxes6,
                                  int32_t* n_boxes) {
    if (n < 0 || (n > 0 && (!local_maps || !pos_lid3 || !boxes6 || !n_boxes))) { set_error("tc2li_lidar_fov_segment_batch: invalid argument"); return TC2LI_ERR_INVALID; }
    int total = 0;
    for (int i = 0; i < n; ++i) {
        const int k = tc2li_lidar_fov_segment(local_maps + i, pos_lid3 + 3 * (size_t)i, cube_len, det_range, boxes6 + 18 * (size_t)i);
        if (k < 0) return k;
        n_boxes[i] = k;
        total += k;
    }
    return total;
}

int tc2li_lidar_feature_extraction(tc2li_lidar* L, tc2li_lidar_map* map, const tc2li_point* feats_down_body, int n,
                                   const tc2li_lidar_state* state, tc2li_point* feats_down_world, uint8_t* point_selected,
                                   tc2li_point* normvec, tc2li_point* nearest_points, float* nearest_sqdist, int32_t* n_nearest,
                                   tc2li_point* laser_cloud_ori, tc2li_point* corr_normvect, int capacity) {
    if (!L || !map || n < 0 || (n > 0 && !feats_down_body) || !state) { set_error("tc2li_lidar_feature_extraction: invalid argument"); return TC2LI_ERR_INVALID; }
    if (n == 0) return 0;
    hipStream_t ps = private_stream();
    std::lock_guard<std::mutex> lock(map->mu);
    int rc = setup_segments(L, 1, &n, ps);
    if (rc != TC2LI_OK) return rc;
    TC2LI_HIP_CHECK(copy_sync(L->d_down.p, feats_down_body, (size_t)n * sizeof(PointXYZINormal), hipMemcpyHostToDevice, ps));
    TC2LI_HIP_CHECK(copy_sync(L->d_down_count.p, &n, sizeof(int), hipMemcpyHostToDevice, ps));
    rc = run_features(L, L->d_down.p, L->d_down_count.p, &map, state, ps);
    if (rc != TC2LI_OK) return rc;
    L->last_down.assign(1, n);
    L->last_map.assign(1, {map, map->generation});
    int m = 0;
    TC2LI_HIP_CHECK(copy_sync(&m, L->d_sel_count.p, sizeof(int), hipMemcpyDeviceToHost, ps));
    if (feats_down_world) TC2LI_HIP_CHECK(copy_sync(feats_down_world, L->d_world.p, (size_t)n * sizeof(PointXYZINormal), hipMemcpyDeviceToHost, ps));
    if (point_selected) TC2LI_HIP_CHECK(copy_sync(point_selected, L->d_selected.p, (size_t)n, hipMemcpyDeviceToHost, ps));
    if (normvec) TC2LI_HIP_CHECK(copy_sync(normvec, L->d_normvec.p, (size_t)n * sizeof(PointXYZINormal), hipMemcpyDeviceToHost, ps));
    if (n_nearest) TC2LI_HIP_CHECK(copy_sync(n_nearest, L->d_nfound.p, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, ps));
    if (nearest_sqdist) TC2LI_HIP_CHECK(copy_sync(nearest_sqdist, L->d_nearest_d.p, (size_t)n * 5 * sizeof(float), hipMemcpyDeviceToHost, ps));
    if (nearest_points) {
        std::vector<int> idx((size_t)n * 5);
        std::vector<PointXYZINormal> mp(map->n);
        TC2LI_HIP_CHECK(copy_sync(idx.data(), L->d_nearest_idx.p, idx.size() * sizeof(int), hipMemcpyDeviceToHost, ps));
        if (map->n) TC2LI_HIP_CHECK(copy_sync(mp.data(), map->d_points.p, (size_t)map->n * sizeof(PointXYZINormal), hipMemcpyDeviceToHost, ps));
        for (size_t k = 0; k < idx.size(); ++k)
            if (idx[k] >= 0) memcpy(&nearest_points[k], &mp[idx[k]], sizeof(PointXYZINormal));
            else memset(&nearest_points[k], 0, sizeof(PointXYZINormal));
    }
    if (m > capacity && (laser_cloud_ori || corr_normvect)) { set_error("output capacity %d < %d", capacity, m); return TC2LI_ERR_CAPACITY; }
    if (laser_cloud_ori && m) TC2LI_HIP_CHECK(copy_sync(laser_cloud_ori, L->d_cloud_ori.p, (size_t)m * sizeof(PointXYZINormal), hipMemcpyDeviceToHost, ps));
    if (corr_normvect && m) TC2LI_HIP_CHECK(copy_sync(corr_normvect, L->d_corr.p, (size_t)m * sizeof(PointXYZINormal), hipMemcpyDeviceToHost, ps));
    return m;
}

int tc2li_lidar_frontend_batch(tc2li_lidar* L, int n_scans, const tc2li_velodyne_point* dev_raw, const int32_t* raw_offsets,
                               int point_filter_num, double blind, float time_unit_scale, float leaf,
                               tc2li_lidar_map* const* maps, const tc2li_lidar_state* states, int32_t* n_preprocessed,
                               int32_t* n_downsampled, int32_t* n_selected, tc2li_point* laser_cloud_ori,
                               tc2li_point* corr_normvect, int capacity, void* stream_) {
    if (!L || n_scans < 0 || n_scans > (L ? L->max_scans : 0) || !dev_raw || !raw_offsets || !maps || !states || point_filter_num < 1 ||
        !(leaf > 0)) {
        set_error("tc2li_lidar_frontend_batch: invalid argument");
        return TC2LI_ERR_INVALID;
    }
    if (n_scans == 0) return 0;
    hipStream_t st = (hipStream_t)stream_;
    for (int s = 0; s < n_scans; ++s) if (!maps[s]) { set_error("tc2li_lidar_frontend_batch: map %d is NULL", s); return TC2LI_ERR_INVALID; }
    MapLocks locks(maps, n_scans);
    std::vector<int> upper(n_scans);
    for (int s = 0; s < n_scans; ++s) upper[s] = raw_offsets[s + 1] - raw_offsets[s];
    // raw scans are packed back to back by the caller and read in place (ScanSlot::raw_base)
    int rc = setup_segments(L, n_scans, upper.data(), st, raw_offsets);
    if (rc != TC2LI_OK) return rc;
    L->record(0, st);
    TC2LI_HIP_CHECK(hipMemcpyAsync(L->d_raw_count.p, upper.data(), n_scans * sizeof(int), hipMemcpyHostToDevice, st));
    rc = run_preprocess(L, (const VelodynePoint*)dev_raw, point_filter_num, blind, time_unit_scale, st, nullptr, nullptr, n_scans >= 32 ? leaf : 0.f);
    if (rc != TC2LI_OK) return rc;
    L->record(1, st);
    rc = run_voxel(L, L->d_pre.p, L->d_pre_count.p, leaf, st, /* compact tables for the down-sampled clouds of a batch */ n_scans >= 32);
    if (rc != TC2LI_OK) return rc;
    L->record(3, st);
    rc = run_features(L, L->d_down.p, L->d_down_count.p, maps, states, st);
    if (rc != TC2LI_OK) return rc;
    L->record(5, st);
    L->timed = true;
    int* hc = L->h_counts.p;
    TC2LI_HIP_CHECK(hipMemcpyAsync(hc, L->d_pre_count.p, n_scans * sizeof(int), hipMemcpyDeviceToHost, st));
    TC2LI_HIP_CHECK(hipMemcpyAsync(hc + n_scans, L->d_down_count.p, n_scans * sizeof(int), hipMemcpyDeviceToHost, st));
    TC2LI_HIP_CHECK(hipMemcpyAsync(hc + 2 * n_scans, L->d_sel_count.p, n_scans * sizeof(int), hipMemcpyDeviceToHost, st));
    TC2LI_HIP_CHECK(hipMemcpyAsync(hc + 3 * n_scans, L->d_status.p, sizeof(int), hipMemcpyDeviceToHost, st));
    TC2LI_HIP_CHECK(stream_wait_blocking(st));
    if (hc[3 * n_scans]) { TC2LI_HIP_CHECK(memset_sync(L->d_status.p, 0, sizeof(int), st)); set_error("more than 32768 occupied voxels in one scan"); return TC2LI_ERR_CAPACITY; }
    L->last_down.assign(hc + n_scans, hc + 2 * n_scans);
    L->last_map.resize(n_scans);
    for (int s = 0; s < n_scans; ++s) L->last_map[s] = {maps[s], maps[s]->generation};
    L->last_sel.assign(hc + 2 * n_scans, hc + 3 * n_scans);
    for (int s = 0; s < n_scans; ++s) {
        if (n_preprocessed) n_preprocessed[s] = hc[s];
        if (n_downsampled) n_downsampled[s] = hc[n_scans + s];
        if (n_selected) n_selected[s] = hc[2 * n_scans + s];
        const int m = hc[2 * n_scans + s];
        if ((laser_cloud_ori || corr_normvect) && m > capacity) { set_error("output capacity %d < %d", capacity, m); return TC2LI_ERR_CAPACITY; }
        if (laser_cloud_ori && m)
            TC2LI_HIP_CHECK(hipMemcpyAsync(laser_cloud_ori + (size_t)s * capacity, L->d_cloud_ori.p + (size_t)s * L->cap,
                                           (size_t)m * sizeof(PointXYZINormal), hipMemcpyDeviceToHost, st));
        if (corr_normvect && m)
            TC2LI_HIP_CHECK(hipMemcpyAsync(corr_normvect + (size_t)s * capacity, L->d_corr.p + (size_t)s * L->cap,
                                           (size_t)m * sizeof(PointXYZINormal), hipMemcpyDeviceToHost, st));
    }
    TC2LI_HIP_CHECK(stream_wait_blocking(st));
    return n_scans;
}

}  // extern "C"
