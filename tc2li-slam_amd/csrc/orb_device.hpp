// Structures shared between the host orchestration and the gfx950 ORB kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tc2li {

constexpr int kMaxLevels = 12;
constexpr int kEdgeThreshold = 19;            // SF/src/ORBextractor.cc:47
constexpr int kMinBorder = kEdgeThreshold - 3;  // minBorderX/Y, :763-764
constexpr int kFastTilePitch = 80;            // >= widest cell window (wCell + 6 < 76)
constexpr int kFastTileH = 76;
constexpr int kMaxCellsPerLevel = 2048;

// One pyramid level of a batch: image i starts at img + i*img_stride, rows are pitch bytes apart.
struct LevelDesc {
    const uint8_t* img;
    size_t img_stride;
    int pitch, w, h, pad_;
};
struct LevelTable { LevelDesc lv[kMaxLevels]; };

// One FAST cell window of a level (identical for every image of the batch): SF/src/ORBextractor.cc:776-797.
struct FastCell {
    int16_t level, x0, y0, w, h, pad_;  // window origin/size in level pixels
    int32_t slab_off, slab_cap;         // slot of this cell's candidates in the per-image slab
};

// A keypoint handed to the orientation/descriptor kernel: level pixel coordinates.
struct DevKeypoint {
    uint32_t packed;     // y << 20 | x << 8 | score
    uint32_t img_level;  // image << 8 | level
};

// One (image, level) keypoint distribution for k_quadtree (ORBextractor::DistributeOctTree, SF/src/ORBextractor.cc:529-753).
struct QuadJob {
    int64_t cand_off;     // first candidate of the level's region in the dense candidate array
    int64_t scratch_off;  // byte offset of the job's work space
    int64_t out_off;      // first slot of the job's picks
    int32_t count_idx;    // image * nlevels + level: index of the candidate count, and of the pick count
    int32_t out_cap, max_keys, max_nodes;
    int32_t min_x, max_x, min_y, max_y, n_target, pad_;
};
size_t quadtree_scratch_bytes(int max_keys, int max_nodes);
size_t quadtree_class_work_ints(int n_jobs);  // launch_quadtree's class_work for a batch of that many jobs (per-class job lists + counters)
void launch_quadtree(const QuadJob* jobs, int first_job, int n_jobs, const uint32_t* dense, const int32_t* level_counts, uint8_t* scratch, uint32_t* picked,
                     int32_t* picked_count, int32_t* status, int threads, int nlevels /* jobs per image, stored (image, level) */, hipStream_t st, int32_t* class_work = nullptr);
void launch_quadtree_gather(const QuadJob* jobs, const uint32_t* picked, const int32_t* picked_count, const int32_t* level_counts, int first_image, int n_images,
                            int nlevels, int kp_stride, DevKeypoint* kps, DevKeypoint* kps_host, int32_t* n_kp, int32_t* n_kp_host, int32_t* level_counts_host,
                            int32_t* status, hipStream_t st);

void launch_resize(const LevelDesc& src, const LevelDesc& dst, const int* xofs, const short* ialpha, const int* yofs,
                   const short* ibeta, int nimg, hipStream_t st);
// levels l0 .. n_levels - 1 of every image in one launch (one workgroup per image walks them; batches: the chain of dependent launches is the cost)
void launch_resize_tail(const LevelTable& lv, const int* const* xofs, const short* const* ialpha, const int* const* yofs, const short* const* ibeta, int l0, int n_levels,
                        int nimg, hipStream_t st);
void launch_fast(const LevelTable& levels, const FastCell* cells, int ncells, int ini_th, int min_th, uint32_t* slab,
                 size_t slab_img_stride, int* cell_counts, int nimg, const int* small_ids, int n_small, const int* large_ids, int n_large,
                 hipStream_t st);
void launch_compact(const FastCell* cells, const int* level_cell_begin, const int* cell_counts, int ncells,
                    const uint32_t* slab, size_t slab_img_stride, uint32_t* dense, const int* level_dense_off,
                    int* level_counts, int nlevels, int nimg, hipStream_t st);
void launch_blur_all(const LevelTable& src, const LevelTable& dst, int nlevels, int nimg, bool rounded_taps, hipStream_t st);
struct MatchKey;
struct ScaleTable;
// Keypoint slots: image i owns [i * kp_stride, i * kp_stride + n_kp[i]) of kps and of every output array; images first_image .. + n_images.
void launch_orient_describe(const LevelTable& raw, const LevelTable& blurred, const ScaleTable& sc, const DevKeypoint* kps, const int32_t* n_kp,
                            int first_image, int n_images, int kp_stride, float* angles, float* angles_dev, uint8_t* desc, MatchKey* mkeys, uint8_t* desc_dev,
                            hipStream_t st);
hipError_t upload_umax(const int* umax16);

}  // namespace tc2li

namespace tc2li {

// Keypoint as the matchers see it: level-0 pixel coordinates and octave (cv::KeyPoint pt/octave).
struct MatchKey {
    float x, y;
    int32_t octave;
};

struct ScaleTable { float scale[kMaxLevels], inv_scale[kMaxLevels]; };

// One stereo frame for k_stereo_match: key/descriptor ranges and pyramid image indices.
struct StereoFrame {
    int32_t left_off, n_left, right_off, n_right;  // ranges in the key/descriptor arrays
    int32_t left_img, right_img;                   // image index inside the left / right level table
    int32_t out_off, pad_;
};

void launch_stereo_match(const LevelTable& left, const LevelTable& right, const ScaleTable& sc, const StereoFrame* frames,
                         int nframes, int max_left, const MatchKey* keys, const uint8_t* desc, float mbf, float max_d,
                         float* u_right, float* depth, int* best_sad, int rows, int entry_cap, int32_t* row_start, uint16_t* entries, hipStream_t st);
// candidates per frame the row lists of the stereo matcher can take: every right keypoint appears in the rows of its band
int stereo_row_entry_cap(const ScaleTable& sc, int n_levels, int max_right);

}  // namespace tc2li
