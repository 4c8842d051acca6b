/*
 * tc2li_hip.h -- C ABI of the MI355X (gfx950) implementation of TC2LI-SLAM's per-frame front end and
 * local bundle adjustment.  Plain pointers and sizes only; every entry point names the reference call
 * site it replaces (paths relative to the reference tree, SF/ = slam_framework/).
 *
 * Conventions
 *   - return value: >= 0 success (often a count), < 0 a tc2li_status error.  tc2li_last_error() gives text.
 *   - "host" pointers are ordinary process memory, "dev" pointers are HIP device memory on the current device.
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).
 *   - there is no CPU fallback: every compute entry point fails with TC2LI_ERR_NO_DEVICE without a GPU.
 */
#ifndef TC2LI_HIP_H
#define TC2LI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum tc2li_status {
    TC2LI_OK = 0,
    TC2LI_ERR_INVALID = -2,    /* bad argument */
    TC2LI_ERR_NO_DEVICE = -3,  /* no HIP device / HIP runtime failure at init */
    TC2LI_ERR_HIP = -4,        /* a HIP call failed; see tc2li_last_error() */
    TC2LI_ERR_CAPACITY = -5,   /* caller-provided buffer too small */
    TC2LI_ERR_COMM = -6,       /* the all-reduce of a sharded window failed (or librccl could not be loaded) */
    TC2LI_ERR_EMPTY = -1       /* empty image: the reference returns -1 (SF/src/ORBextractor.cc:1063-1064) */
} tc2li_status;

const char* tc2li_last_error(void);
/* ABI version, bumped on any signature change. */
int tc2li_abi_version(void);
/* Number of visible HIP devices (0 without a GPU; never fails). */
int tc2li_device_count(void);
/* Hardware queues the HIP runtime maps this process's streams onto (the runtime's GPU_MAX_HW_QUEUES, 4 by default).  A process that runs
 * ONE sequence -- the reference's own configuration: tracking, LiDAR and local-mapping threads with a stream each, every kernel tiny --
 * should ask for 8 (streams that share a queue wait for each other: 479 against 761 frames/s) -- 16 to 24 when it also
 * runs streams of its own, e.g. the uploads of the next frame's images and scan (557 against 810-827 frames/s host-fed); batched callers keep
 * the default.  Returns TC2LI_ERR_INVALID for n < 1 or n > 32.  No reference counterpart.
 * It only takes effect before the process's first HIP call: once this library has called into HIP (any entry that needs the device,
 * tc2li_device_count included) it returns TC2LI_ERR_INVALID instead of silently doing nothing.  It sets an environment variable (setenv):
 * call it before the process starts other threads. */
int tc2li_set_hardware_queues(int n);
/* CPUs this process may really keep busy (a launcher passes min(affinity, cgroup quota) / ranks on the node): the library sizes its worker
 * pools from it -- TC2LI_HOST_THREADS_PER_CPU (default 8: measured on a 16-CPU grant) threads per CPU in all, of
 * which the extractor pool takes a quarter, the tracking pool, the LiDAR pool and every lock-step BA group an eighth after the caller's own
 * stage threads; caps 32 / 16 / 16 / 16, the sizes the pools were tuned at on a one-GPU box.  Default: the environment variable
 * TC2LI_HOST_THREAD_BUDGET, else the cores the process may run on (sched_getaffinity).  Returns
 * TC2LI_ERR_INVALID for threads < 1 or when a pool exists already (call it first, or after tc2li_shutdown).  No reference counterpart (the
 * reference's four threads are fixed, SF/src/System.cc:184-224). */
int tc2li_set_host_thread_budget(int threads);
/* counts[0] = the budget in force, [1] extractor pool, [2] tracking pool, [3] LiDAR pool, [4] threads per lock-step BA group,
 * [5] largest number of lock-step groups; capacity >= 6.  Returns 6. */
int tc2li_host_threads(int32_t* counts, int capacity);
/* Orderly end (or pause) of the library's own threads: joins every worker pool -- each worker's thread-local work spaces (device and pinned
 * buffers, streams) are released by its exit -- releases the process-wide work spaces (lock-step BA contexts, mapping work spaces) and
 * synchronises the device, all while the HIP runtime is alive.  The caller guarantees that no other thread is inside the library; its own
 * threads that called the library should have ended (their thread-local work spaces go with them).  Handles stay valid and every entry
 * keeps working afterwards (pools and work spaces are made again on demand).  Call it before main() returns: a process that leaves with
 * library threads alive runs their teardown concurrently with the HIP runtime's (the reference's System::Shutdown, SF/src/System.cc:325-377,
 * joins its threads for the same reason). */
int tc2li_shutdown(void);

/* ------------------------------------------------------------------------------------------------
 * ORB extractor -- replaces TC2LI_SLAM::ORBextractor (SF/include/ORBextractor.h:46-121,
 * SF/src/ORBextractor.cc:383-443 ctor, :1060-1141 operator()).  One handle per extractor object; the
 * reference runs the left and right extractor concurrently from two threads (SF/src/Frame.cc:139-142),
 * so handles are independent and re-entrant per handle.
 * ---------------------------------------------------------------------------------------------- */

/* Subset of cv::KeyPoint the reference reads (pt, size, angle, response, octave). */
typedef struct tc2li_keypoint {
    float x, y;      /* level-0 pixel coordinates (SF/src/ORBextractor.cc:1122-1124) */
    float size;      /* 31 * scale[octave], truncated (:851,:861) */
    float angle;     /* degrees, cv::fastAtan2 of the intensity centroid (:50-77) */
    float response;  /* FAST score */
    int32_t octave;
} tc2li_keypoint;

/* ORBextractor ctor arguments (SF/src/ORBextractor.cc:383-384; values from config/.../KITTI00-02.yaml:60-73). */
typedef struct tc2li_orb_params {
    int32_t nfeatures;
    float scale_factor;
    int32_t nlevels;
    int32_t ini_th_fast;
    int32_t min_th_fast;
} tc2li_orb_params;

typedef struct tc2li_orb tc2li_orb;

/* Creates an extractor able to process up to `max_images` images of at most max_width x max_height per call. */
int tc2li_orb_create(const tc2li_orb_params* params, int max_width, int max_height, int max_images, tc2li_orb** out);
void tc2li_orb_destroy(tc2li_orb* orb);

/* ORBextractor::operator()(image, mask, keypoints, descriptors, vLappingArea)  (SF/src/ORBextractor.cc:1060).
 * One host image in; keypoints and N x 32 descriptor bytes out in the reference's order (level-major, mono
 * indices ascending from the front, lapping-area keys descending from the back).  *n_keypoints receives N.
 * Returns monoIndex like the reference, TC2LI_ERR_EMPTY (-1) for an empty image. */
int tc2li_orb_extract(tc2li_orb* orb, const uint8_t* image, int width, int height, int stride,
                      const int32_t lapping_area[2], tc2li_keypoint* keypoints, uint8_t* descriptors, int capacity,
                      int32_t* n_keypoints);

/* Batched form for images already resident in device memory: image i starts at dev_images + i*image_pitch_bytes,
 * rows are `stride` bytes apart.  Results go to host arrays laid out [n_images][capacity].  mono_index may be NULL.
 * The device images must stay valid until the handle's pyramids are no longer needed (level 0 is read in place). */
int tc2li_orb_extract_batch(tc2li_orb* orb, const uint8_t* dev_images, int n_images, int width, int height, int stride,
                            size_t image_pitch_bytes, const int32_t lapping_area[2], tc2li_keypoint* keypoints,
                            uint8_t* descriptors, int capacity, int32_t* n_keypoints, int32_t* mono_index, void* stream);

/* Accessors the reference reads off the extractor: GetLevels/GetScaleFactors/... (SF/include/ORBextractor.h:70-90)
 * and mvImagePyramid (:92; used by Frame::ComputeStereoMatches, SF/src/Frame.cc:848,938,953). */
int tc2li_orb_levels(const tc2li_orb* orb);
int tc2li_orb_scale_factors(const tc2li_orb* orb, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2);
int tc2li_orb_features_per_level(const tc2li_orb* orb, int32_t* per_level);
int tc2li_orb_level_size(const tc2li_orb* orb, int level, int* width, int* height);
/* Copies pyramid level `level` of image `image_index` of the last call to host, tightly packed width*height. */
int tc2li_orb_download_level(tc2li_orb* orb, int image_index, int level, uint8_t* dst);
/* Same for the 7x7 Gaussian-blurred level the descriptors were sampled from (SF/src/ORBextractor.cc:1105-1106). */
int tc2li_orb_download_blurred(tc2li_orb* orb, int image_index, int level, uint8_t* dst);
/* FAST candidates of the last call before the quadtree (diagnostic; x, y in level pixels, response): returns count. */
int tc2li_orb_download_candidates(tc2li_orb* orb, int image_index, int level, float* xyr, int capacity);

/* Times of the last batch call in milliseconds.  Device stages are measured with HIP events on the stream each
 * kernel is launched on: [0] pyramid (nlevels-1 resize launches), [1] FAST cells kernel, [2] candidate compaction
 * kernel, [3] blur (nlevels launches), [4] orientation+descriptor kernel, [5] keypoint distribution (quadtree + gather kernels);
 * host wall clock: [6] call entry until everything is queued, [7] whole call.
 * With profiling enabled every kernel is issued on the caller's stream (no overlap of blur with FAST), so that
 * [0]..[4] are clean per-stage durations. */
int tc2li_orb_set_profiling(tc2li_orb* orb, int enabled);
int tc2li_orb_last_timings(const tc2li_orb* orb, float ms[8]);
/* A batch call is queued in chunks of images (the blur of one chunk runs on a second stream beside the keypoint distribution of the
 * chunk before): every device stage is launched once per chunk, and [0]..[5] above are the sums over the chunks.  Returns the chunk
 * count of the last call. */
int tc2li_orb_last_chunks(const tc2li_orb* orb);

/* ------------------------------------------------------------------------------------------------
 * Stereo matching -- replaces Frame::ComputeStereoMatches (SF/src/Frame.cc:841-1011; called from the stereo
 * Frame constructor, :160).  bf = mbf, b = mb (= mbf / fx, :197).  Outputs are mvuRight / mvDepth: -1 where a left
 * keypoint has no accepted match.  best_sad (may be NULL) receives the SAD of the sub-pixel stage before the
 * final 1.5*1.4*median cut (-1 = none).
 * ---------------------------------------------------------------------------------------------- */

/* Single frame, reference-style: the two extractor handles hold the pyramids of the left / right image of their
 * last tc2li_orb_extract call (mvImagePyramid), keypoints and descriptors come back from the caller. */
int tc2li_stereo_match(tc2li_orb* left, tc2li_orb* right, const tc2li_keypoint* keys_left, const uint8_t* desc_left,
                       int n_left, const tc2li_keypoint* keys_right, const uint8_t* desc_right, int n_right, float bf, float b,
                       float* u_right, float* depth, int32_t* best_sad);

/* Batched: frame f = images 2f (left) and 2f+1 (right) of the handle's last tc2li_orb_extract_batch call (lapping
 * area {0,0}); features are taken from device memory where that call left them.  Outputs are [n_frames][capacity]. */
int tc2li_stereo_match_batch(tc2li_orb* orb, int n_frames, float bf, float b, float* u_right, float* depth,
                             int32_t* best_sad, int capacity, void* stream);

/* ------------------------------------------------------------------------------------------------
 * LiDAR front end, camera-LiDAR branch -- replaces the internals of
 *   Preprocess::process / velodyne_handler      SF/include/lidar_front_end/preprocess.cpp:63-167 (non-feature branch)
 *   pcl::VoxelGrid<PointXYZINormal>::filter      call sites LidarFrontEnd.cpp:712-714, 913-915
 *   ikdtree.Build / Add_Points / Nearest_Search  SF/include/ikd-Tree/ikd_Tree.cpp:409-461 (a hash grid here, same 5-NN)
 *   feature_extraction / EstiPlane               LidarFrontEnd.cpp:964-1073, with pointBodyToWorld :130-139
 * Point layouts are PCL's: velodyne_ros::Point (32 B, preprocess.h:62-70) and pcl::PointXYZINormal (48 B).
 * ---------------------------------------------------------------------------------------------- */
typedef struct tc2li_velodyne_point {
    float x, y, z, pad0;
    float intensity, time;
    uint16_t ring, pad1;
    float pad2;
} tc2li_velodyne_point;

typedef struct tc2li_point { /* pcl::PointXYZINormal */
    float x, y, z, pad0;
    float normal_x, normal_y, normal_z, pad1;
    float intensity, curvature, pad2, pad3;
} tc2li_point;

/* The parts of state_ikfom that pointBodyToWorld reads (LidarFrontEnd.cpp:130-139); matrices row-major. */
typedef struct tc2li_lidar_state {
    double rot[9], pos[3], offset_R_L_I[9], offset_T_L_I[3];
} tc2li_lidar_state;

typedef struct tc2li_lidar tc2li_lidar;         /* stage workspace for up to max_scans scans per call */
typedef struct tc2li_lidar_map tc2li_lidar_map; /* the incremental map (the reference's global `ikdtree`) */

int tc2li_lidar_create(int max_points_per_scan, int max_scans, tc2li_lidar** out);
void tc2li_lidar_destroy(tc2li_lidar* lidar);

/* Preprocess::process for a Velodyne cloud with feature_enabled = false: keeps point i when i % point_filter_num == 0
 * and |p|^2 > blind^2; curvature = time * time_unit_scale (ms).  Returns the number of points written. */
int tc2li_lidar_preprocess(tc2li_lidar* lidar, const tc2li_velodyne_point* raw, int n, int point_filter_num, double blind,
                           float time_unit_scale, tc2li_point* out, int capacity);

/* downSizeFilterSurf.setInputCloud(in); downSizeFilterSurf.filter(out) with leaf size `leaf` on all three axes:
 * one centroid (of every field) per occupied voxel, in ascending voxel-index order. */
int tc2li_lidar_voxel_filter(tc2li_lidar* lidar, const tc2li_point* in, int n, float leaf, tc2li_point* out, int capacity);

/* LiDAR map handle -- thread contract.  In the reference the global `ikdtree` is touched from two threads: the LiDAR thread
 * (feature_extraction / h_share_model, LidarFrontEnd.cpp:942,749) and the tracking thread (UpdateMap -> map_incremental,
 * Tracking.cc:1602-1603, under `finishMutex`).  A tc2li_lidar_map may be used from any number of host threads: every entry
 * point that reads or changes a map (build / add / size / download / feature_extraction / frontend_batch / eskf_update /
 * map_incremental / delete_boxes) holds the handle's internal lock from its first access until its device work on the map
 * has completed (each of them synchronises its stream before it returns), so calls on one handle serialise inside the
 * library; a batch call locks the distinct maps of its batch in address order.  What the lock cannot give is the reference's
 * sequencing: tc2li_lidar_map_incremental replays the neighbours found by the handle's LAST feature extraction against the
 * same map, so the caller must not let another thread change that map in between (the reference's finishMutex does this).
 * A tc2li_lidar workspace itself is NOT shareable between threads (one per calling thread, like the ORB extractor handle).
 * tc2li_lidar_map_destroy must not race with any other call on the handle.
 * Entry points without a `stream` argument run on a private non-blocking stream of the calling thread (never the NULL stream). */
int tc2li_lidar_map_create(tc2li_lidar_map** out);
void tc2li_lidar_map_destroy(tc2li_lidar_map* map);
/* ikdtree.Build(points) (LidarFrontEnd.cpp:918-931): replaces the map content.  Returns the map size. */
int tc2li_lidar_map_build(tc2li_lidar_map* map, const tc2li_point* world_points, int n);
/* ikdtree.Add_Points(points, false): appends without down-sampling.  Returns the map size. */
int tc2li_lidar_map_add(tc2li_lidar_map* map, const tc2li_point* world_points, int n);
int tc2li_lidar_map_size(const tc2li_lidar_map* map);
/* Measurement / test hooks of the map's spatial index (no reference counterpart: ikd-Tree keeps its own counters, ikd_Tree.h:117-123).
 * tc2li_lidar_map_stats: out[0] points, [1] places of the grid (entries + the rows' room), [2] grid builds so far, [3] in-place grid updates
 * so far (map_incremental calls that merged their points into the existing rows, KD_TREE::Add_Points' own way, ikd_Tree.cpp:478-584),
 * [4] tombstones (upper bound), [5] cells; capacity >= 6, returns 6.
 * tc2li_lidar_map_grid_download: walks the grid on the host, checks it (every live entry stands in the cell its coordinates name and names
 * an existing point, rows stay inside their room, unused room holds no live entry: TC2LI_ERR_INVALID with the finding otherwise) and
 * returns the number of live entries, the first `capacity` of them as (cell, point index) in grid order. */
int tc2li_lidar_map_stats(const tc2li_lidar_map* map, int32_t* out, int capacity);
int tc2li_lidar_map_grid_download(const tc2li_lidar_map* map, int32_t* cells, int32_t* indices, int capacity);

/* feature_extraction() (LidarFrontEnd.cpp:999-1073) for one down-sampled scan.  Per input point i (arrays of n, any
 * may be NULL): feats_down_world[i], point_selected[i], normvec[i] (plane normal, intensity = pd2), the up-to-5
 * Nearest_Points[i] ([n][5], ascending distance) with their squared distances and count.  laser_cloud_ori /
 * corr_normvect receive the compacted selection; the return value is effct_feat_num. */
int tc2li_lidar_feature_extraction(tc2li_lidar* lidar, tc2li_lidar_map* map, const tc2li_point* feats_down_body, int n,
                                   const tc2li_lidar_state* state, tc2li_point* feats_down_world, uint8_t* point_selected,
                                   tc2li_point* normvec, tc2li_point* nearest_points, float* nearest_sqdist, int32_t* n_nearest,
                                   tc2li_point* laser_cloud_ori, tc2li_point* corr_normvect, int capacity);

/* Whole front end for a batch of raw scans resident in device memory (scan s = dev_raw[raw_offsets[s] .. raw_offsets[s+1])):
 * preprocess -> voxel filter -> feature extraction against maps[s] with states[s]; stages chain on the device.
 * Per-scan counts come back in the three int arrays; the compacted selections in [n_scans][capacity] host arrays
 * (either may be NULL). */
int tc2li_lidar_frontend_batch(tc2li_lidar* lidar, int n_scans, const tc2li_velodyne_point* dev_raw, const int32_t* raw_offsets,
                               int point_filter_num, double blind, float time_unit_scale, float leaf,
                               tc2li_lidar_map* const* maps, const tc2li_lidar_state* states, int32_t* n_preprocessed,
                               int32_t* n_downsampled, int32_t* n_selected, tc2li_point* laser_cloud_ori,
                               tc2li_point* corr_normvect, int capacity, void* stream);
/* ---- persistent map maintenance on the device (the map never returns to the host) ----
 * map_incremental (SF/include/lidar_front_end/LidarFrontEnd.cpp:387-435) for scan slot `scan` of the handle's last
 * feature extraction (tc2li_lidar_feature_extraction: slot 0; tc2li_lidar_frontend_batch: the scan's index) against the
 * SAME map, unchanged since: world coordinates at `state` (UpdateLidarPose may have moved it), the insertion rule
 * (one point per filter_size_map_min voxel, the one nearest the voxel centre), then
 * ikdtree.Add_Points(PointToAdd, true) / Add_Points(PointNoNeedDownsample, false) (ikd_Tree.cpp:478-584).
 * ekf_inited = flg_EKF_inited.  Returns the new map size; the list sizes go to n_to_add / n_no_need (may be NULL). */
int tc2li_lidar_map_incremental(tc2li_lidar* lidar, int scan, tc2li_lidar_map* map, const tc2li_lidar_state* state, int ekf_inited,
                                double filter_size_map_min, int32_t* n_to_add, int32_t* n_no_need, void* stream);
/* The same for n (scan slot, map) pairs of the handle's last tc2li_lidar_frontend_batch in one call -- what the tracking threads of n
 * sequences do at SyncWithLidar (Tracking.cc:1602-1603), with one kernel launch per phase for all maps instead of a dozen per map.
 * scans[i] = scan slot, maps[i] = its map (every map at most once), states[i] = the state map_incremental reads.  The list sizes and
 * the new map sizes go to the three int arrays (any may be NULL).  Returns n.  On TC2LI_ERR_CAPACITY no map has been changed. */
int tc2li_lidar_map_incremental_batch(tc2li_lidar* lidar, int n, const int32_t* scans, tc2li_lidar_map* const* maps,
                                      const tc2li_lidar_state* states, int ekf_inited, double filter_size_map_min, int32_t* n_to_add,
                                      int32_t* n_no_need, int32_t* map_sizes, void* stream);
/* ---- pose plumbing between the camera thread and the LiDAR front end ----
 * Poses are Sophus::SE3f as qx qy qz qw tx ty tz; the arithmetic is float, in Sophus' / Eigen's order.
 * tc2li_lidar_update_pose = UpdateLidarPose (SF/include/lidar_front_end/LidarFrontEnd.cpp:786-800): Twc = Tcw_last^-1 * exp(t * log(velocity^-1)),
 * the LiDAR pose in the front end's world frame (Rw2_w1 * Twc * Tcl) into state->rot / state->pos, pos_lid = pos + rot * offset_T_L_I (may be NULL).
 * tc2li_se3_interpolate = InterpolateSE3 (SF/src/Tracking.cc:1552-1563): quaternion slerp + linear translation.
 * tc2li_lidar_sync_transform = the transform Tracking::SyncWithLidar applies to a scan's feature cloud (:1600-1626):
 *   Tlc * Tcw_frame * InterpolateSE3(Tcw_last^-1, Tcw_cur^-1, ratio) * Tcl, Tcw_frame = the frame the scan pairs with (current or last).
 * tc2li_lidar_keyframe_transform = the one of Tracking::BuildLidarFeat4KeyFrame (:1537-1547): Tlc * Tcw_cur * (rel * Tcw_refkf)^-1 * Tcl.
 * tc2li_transform_point_cloud = LidarFrontEndTools::transformPointCloud (SF/src/LidarTypes.cc:42-65) on host arrays; returns n.
 * tc2li_lidar_transform_features_batch: the same for the selected feature clouds (laserCloudOri = mCurrFeatPoints) of scan slots `scans` of the
 * handle's last tc2li_lidar_frontend_batch, read on the device where the front end left them, one launch for all scans:
 * out [n][capacity] (host), n_points[i] = points written for scans[i] (may be NULL). */
int tc2li_lidar_update_pose(const float Tcw_last7[7], const float velocity7[7], double time_from_last_frame, const float Tcl7[7],
                            tc2li_lidar_state* state, double pos_lid[3]);
int tc2li_se3_interpolate(const float a7[7], const float b7[7], float t, float out7[7]);
int tc2li_lidar_sync_transform(const float Tcw_frame7[7], const float Tcw_last7[7], const float Tcw_cur7[7], float ratio, const float Tlc7[7],
                               const float Tcl7[7], float out7[7]);
int tc2li_lidar_keyframe_transform(const float Tcw_cur7[7], const float rel7[7], const float Tcw_refkf7[7], const float Tlc7[7],
                                   const float Tcl7[7], float out7[7]);
int tc2li_transform_point_cloud(const tc2li_point* in, int n, const float T7[7], tc2li_point* out, void* stream);
int tc2li_lidar_transform_features_batch(tc2li_lidar* lidar, int n, const int32_t* scans, const float* T7, tc2li_point* out, int capacity,
                                         int32_t* n_points, void* stream);
/* ikdtree.Delete_Point_Boxes (ikd_Tree.cpp:643): removes the points inside the boxes [min, max) given as
 * min x y z, max x y z per box; returns how many were removed. */
int tc2li_lidar_map_delete_boxes(tc2li_lidar_map* map, const float* boxes6, int n_boxes, void* stream);
/* The same for n_maps different maps in one go (one launch per phase; the per-sequence lasermap_fov_segment calls of a batch of
 * sequences): map i gets the boxes box_offsets[i] .. box_offsets[i+1] of boxes6 (box_offsets[0] = 0).  n_removed [n_maps] (may be
 * NULL) receives the per-map counts; returns their sum. */
int tc2li_lidar_map_delete_boxes_batch(int n_maps, tc2li_lidar_map* const* maps, const float* boxes6, const int32_t* box_offsets,
                                       int32_t* n_removed, void* stream);
/* Copies the map points to the host (diagnostics / tests); returns the map size. */
int tc2li_lidar_map_download(const tc2li_lidar_map* map, tc2li_point* out, int capacity);
/* lasermap_fov_segment (LidarFrontEnd.cpp:183-231), host logic: keeps the local-map cube around the sensor and returns
 * the number of boxes (<= 3, written to boxes6) whose points must be deleted. */
typedef struct tc2li_local_map_box { float vertex_min[3], vertex_max[3]; int32_t initialized; } tc2li_local_map_box;
int tc2li_lidar_fov_segment(tc2li_local_map_box* local_map, const double pos_lid[3], double cube_len, double det_range, float boxes6[18]);
/* The same for n sensors in one call (the per-sequence calls of a batch driver): local_maps [n], pos_lid3 [n][3]; boxes6 [n][18] and n_boxes [n]
 * receive every sequence's boxes and their number; returns the total. */
int tc2li_lidar_fov_segment_batch(tc2li_local_map_box* local_maps, const double* pos_lid3, int n, double cube_len, double det_range, float* boxes6,
                                  int32_t* n_boxes);

/* ---- camera-LiDAR-inertial branch: motion compensation of the scan (ImuProcess::UndistortPcl,
 * SF/include/lidar_front_end/IMU_Processing.cpp:160-277) ---- */
typedef struct tc2li_imu_pose6d {   /* Pose6D saved at every IMU sample during the forward propagation */
    double offset_time;             /* seconds since the scan start */
    double acc[3], gyr[3];          /* world-frame acceleration / bias-free angular velocity of the interval ending here */
    double vel[3], pos[3], rot[9];  /* IMU state at the sample */
} tc2li_imu_pose6d;
typedef struct tc2li_imu_state {    /* the parts of state_ikfom the propagation reads / writes (rotations row-major) */
    double pos[3], rot[9], vel[3], bg[3], ba[3], grav[3], offset_R_L_I[9], offset_T_L_I[3];
} tc2li_imu_state;
typedef struct tc2li_imu_meas { double t, acc[3], gyr[3]; } tc2li_imu_meas;   /* sensor_msgs::Imu fields used */

/* Forward propagation of UndistortPcl (:176-233), state part of esekf::predict (covariance: see the ESKF entry):
 * v_imu = last scan's tail sample followed by this scan's samples; acc_scale = G_m_s2 / mean_acc.norm();
 * acc_s_last / angvel_last from the previous call.  Writes the poses (capacity >= n_imu) and the scan-end state into
 * *state; returns the number of poses.  Host-only. */
int tc2li_lidar_imu_propagate(tc2li_imu_state* state, const tc2li_imu_meas* v_imu, int n_imu, double pcl_beg_time,
                              double pcl_end_time, double last_lidar_end_time, double acc_scale, double acc_s_last[3],
                              double angvel_last[3], tc2li_imu_pose6d* poses, int capacity);

/* The same forward propagation with the covariance (esekf::predict, SF/include/IKFoM_toolkit/esekfom/esekfom.hpp:281-392 with
 * get_f / df_dx / df_dw of SF/src/use-ikfom.cpp:45-91): P is the 23 x 23 row-major error-state covariance in the order pos, rot,
 * offset_R_L_I, offset_T_L_I, vel, bg, ba, grav (2); cov12 = cov_gyr, cov_acc, cov_bias_gyr, cov_bias_acc, the diagonal of Q
 * (IMU_Processing.cpp:215-218).  Host-only: 23 x 23 products per IMU sample. */
int tc2li_lidar_imu_propagate_cov(tc2li_imu_state* state, double* P, const double cov12[12], const tc2li_imu_meas* v_imu, int n_imu,
                                  double pcl_beg_time, double pcl_end_time, double last_lidar_end_time, double acc_scale,
                                  double acc_s_last[3], double angvel_last[3], tc2li_imu_pose6d* poses, int capacity);
/* One esekf::predict step with a full 12 x 12 process noise Q (ng, na, nbg, nba). */
int tc2li_eskf_predict(tc2li_imu_state* state, double* P, const double* Q, const double acc[3], const double gyr[3], double dt);

/* esekf::update_iterated_dyn_share_modified (esekfom.hpp:1621-1932) with h_share_model (LidarFrontEnd.cpp:485-602) as the
 * measurement model, as called at LidarFrontEnd.cpp:749: R = LASER_POINT_COV, maximum_iter = NUM_MAX_ITERATIONS, limit23 = epsi.
 * Every iteration evaluates the point-to-plane residuals of feats_down_body against `map` at the current state (the neighbour
 * search only when the previous iteration converged, as the reference does), reduces the 12 active columns of H to H^T H and
 * H^T h on the device, and solves the 23-dof update on the host.  State and P (23 x 23) are updated in place.  Afterwards the
 * handle holds Nearest_Points / the selection of the last evaluation for tc2li_lidar_map_incremental.  Returns effct_feat_num
 * of the last evaluation. */
typedef struct tc2li_eskf_stats {
    int32_t calls;            /* h_share_model evaluations */
    int32_t effct_feat_num;   /* selected points of the last one */
    int32_t searches;         /* evaluations that ran the neighbour search */
    int32_t converged;        /* iterations whose step stayed below limit23 */
    int32_t finished;         /* the covariance update ran (:1823-1928) */
    int32_t pad_;
    double res_mean_last;
} tc2li_eskf_stats;
int tc2li_lidar_eskf_update(tc2li_lidar* lidar, tc2li_lidar_map* map, const tc2li_point* feats_down_body, int n, tc2li_imu_state* state,
                            double* P, double R, int maximum_iter, const double* limit23, int extrinsic_est_en, tc2li_eskf_stats* stats);

/* The point part (:170-172, 236-276): sorts the scan by time offset (curvature, ms) in the order std::sort(time_list)
 * produces and moves every point into the scan-end frame; in place on the host array.  end_state = imu_state after the
 * last predict (rot, pos, offset_R_L_I, offset_T_L_I are read). */
int tc2li_lidar_undistort(tc2li_lidar* lidar, tc2li_point* points, int n, const tc2li_imu_pose6d* imu_poses, int n_poses,
                          const tc2li_lidar_state* end_state);

/* ---- the LiDAR thread of the camera-LiDAR-inertial configuration for a batch of sequences ----
 * LidarInertialProcess (SF/include/lidar_front_end/LidarFrontEnd.cpp:615-785) for n_scans scans of n_scans sequences in one call:
 * Preprocess::process of the raw scans (resident in device memory like tc2li_lidar_frontend_batch's), ImuProcess::Process = forward
 * propagation with the covariance on the host (tc2li_lidar_imu_propagate_cov) + UndistortPcl on the device -- the time sort included: the
 * permutation std::sort leaves is replayed on the device --, downSizeFilterSurf.filter, and kf.update_iterated_dyn_share_modified with
 * h_share_model (:749) for all scans in lock step: per iteration one launch per phase over the scans still iterating, then every
 * scan's 23-dof algebra on host threads.  Every scan's result is the one of the one-scan entry points called in that order
 * (tc2li_lidar_preprocess, tc2li_lidar_imu_propagate_cov, tc2li_lidar_undistort, tc2li_lidar_voxel_filter, tc2li_lidar_eskf_update).
 * Afterwards the handle holds every scan's down-sampled points and Nearest_Points for tc2li_lidar_map_incremental_batch (scan slot s). */
typedef struct tc2li_lidar_inertial_scan {
    const tc2li_imu_meas* imu;          /* v_imu: the last scan's tail sample followed by this scan's samples */
    int32_t n_imu, pad_;
    double pcl_beg_time, pcl_end_time, last_lidar_end_time, acc_scale;
    double acc_s_last[3], angvel_last[3];   /* in / out, as tc2li_lidar_imu_propagate */
    tc2li_imu_state state;              /* in: the filter state at the last scan end; out: after the iterated update */
    double* P;                          /* [23 * 23] in / out */
    tc2li_eskf_stats stats;             /* out */
    int32_t n_preprocessed, n_downsampled;  /* out */
} tc2li_lidar_inertial_scan;
int tc2li_lidar_inertial_frontend_batch(tc2li_lidar* lidar, int n_scans, const tc2li_velodyne_point* dev_raw, const int32_t* raw_offsets,
                                        int point_filter_num, double blind, float time_unit_scale, float leaf, tc2li_lidar_map* const* maps,
                                        tc2li_lidar_inertial_scan* scans, const double cov12[12], double R, int maximum_iter,
                                        const double* limit23, int extrinsic_est_en, void* stream);
/* The part of LidarInertialProcess that depends on the scans alone, as a call of its own: Preprocess::process of every raw scan (the reference
 * runs it in the scan callback, LidarFrontEnd.cpp:253, ahead of the thread that consumes lidar_buffer) and the order UndistortPcl's
 * std::sort(time_list) will leave the points in.  The handle then holds n_scans prepared scans; the next
 * tc2li_lidar_inertial_frontend_batch on it with dev_raw = NULL (raw_offsets, point_filter_num, blind, time_unit_scale are not read then) and
 * the same n_scans consumes them and gives exactly the results of the one-call form.  Two handles let a caller prepare the scans of step k + 1
 * (own stream, own thread) while step k's iterated update runs.  Returns n_scans. */
int tc2li_lidar_inertial_prepare_batch(tc2li_lidar* lidar, int n_scans, const tc2li_velodyne_point* dev_raw, const int32_t* raw_offsets,
                                       int point_filter_num, double blind, float time_unit_scale, void* stream);
/* The time sort of UndistortPcl alone (tests / diagnostics): perm[i] = index of the point std::sort(points, time_list) leaves at place i,
 * computed by the device kernel of the batch entry (one scan; depth_limit < 0: std::sort's own 2 floor(log2 n)).  Returns 1 when the
 * recursion reached the depth limit (perm is then unspecified: the batch entry sorts such a scan on the host), else 0. */
int tc2li_device_time_sort(tc2li_lidar* lidar, const tc2li_point* points, int n, int depth_limit, int32_t* perm);

/* Device time of the stages of the last tc2li_lidar_frontend_batch call, from HIP events on its stream: ms[0]
 * preprocess, [1] voxel hashing/sorting, [2] voxel centroids, [3] 5-NN + plane fit, [4] selection, [5] total, [6] / [7] the two
 * kernels of stage [3] (k_knn_plane, k_knn_hard). */
int tc2li_lidar_last_timings(tc2li_lidar* lidar, float ms[8]);

/* ------------------------------------------------------------------------------------------------
 * Projection-guided matching of the tracking thread -- replaces the two overloads of ORBmatcher::SearchByProjection
 * that Tracking uses: (Frame&, const Frame& LastFrame, th, bMono) SF/src/ORBmatcher.cc:1685 (TrackWithMotionModel,
 * Tracking.cc:2771,2780) and (Frame&, const vector<MapPoint*>&, th, bFarPoints, thFarPoints) :52 (SearchLocalPoints,
 * Tracking.cc:3282).  Map-point pointers stay on the host: each source point becomes one query.
 * ---------------------------------------------------------------------------------------------- */
typedef struct tc2li_proj_query {
    float u, v;                 /* projection in the current frame */
    float radius;               /* window half size handed to Frame::GetFeaturesInArea */
    float u_right;              /* predicted right coordinate (stereo consistency check) */
    int32_t min_level, max_level; /* GetFeaturesInArea level arguments */
    float angle;                /* keypoint angle in the source frame (rotation histogram) */
    int16_t valid;              /* 0: the source point produces no search */
    int16_t has_observations;   /* pMP->Observations() > 0: a match blocks the keypoint for later points */
    uint8_t descriptor[32];     /* pMP->GetDescriptor() */
} tc2li_proj_query;

typedef struct tc2li_frame_view {  /* the parts of the current Frame the matcher reads */
    const tc2li_keypoint* keys;    /* mvKeysUn */
    const uint8_t* descriptors;    /* mDescriptors, n x 32 */
    const float* u_right;          /* mvuRight */
    const uint8_t* occupied;       /* mvpMapPoints[i] && Observations() > 0 before the call (may be NULL) */
    int32_t n;
    float min_x, max_x, min_y, max_y; /* mnMinX .. mnMaxY */
} tc2li_frame_view;

typedef struct tc2li_map_point {  /* what Frame::isInFrustum / PredictScale read off a MapPoint */
    float pos[3], normal[3];
    float min_distance, max_distance; /* Get{Min,Max}DistanceInvariance() */
    float max_distance_raw;           /* mfMaxDistance */
    uint8_t descriptor[32];
} tc2li_map_point;

/* The matching loops.  mode 0: best Hamming distance <= TH_HIGH (last-frame overload); mode 1: best / second best with
 * nn_ratio when both lie on one level (local-map overload).  check_orientation applies the 30-bin rotation histogram.
 * match_of_query[q] = matched keypoint or -1; query_of_keypoint[i] (may be NULL) = the query now held by keypoint i.
 * Returns nmatches like the reference. */
int tc2li_search_by_projection(const tc2li_frame_view* frame, const tc2li_proj_query* queries, int n_queries, int mode,
                               float nn_ratio, int check_orientation, int32_t* match_of_query, int32_t* query_of_keypoint);

/* Query construction of the last-frame overload (ORBmatcher.cc:1696-1739).  Poses are Sophus::SE3f as 7 floats
 * (qx, qy, qz, qw, tx, ty, tz); cam4 = fx, fy, cx, cy; b = mb, bf = mbf.  One query per last-frame keypoint i
 * (has_point[i] = LastFrame.mvpMapPoints[i] != NULL, outlier[i] = mvbOutlier[i], Xw = world positions).  Returns the
 * number of valid queries. */
int tc2li_project_last_frame(const float pose_cur7[7], const float pose_last7[7], const float cam4[4], float b, float bf,
                             const float* scale_factors, int n_levels, int cols, int rows, int n, const uint8_t* has_point,
                             const uint8_t* outlier, const float* Xw, const tc2li_keypoint* last_keys, const uint8_t* mp_descriptors,
                             float th, int mono, tc2li_proj_query* queries);

/* Frame::isInFrustum(pMP, viewing_cos_limit) (SF/src/Frame.cc:542-603) + MapPoint::PredictScale + the window of the
 * local-map overload (ORBmatcher.cc:62-81) for n local map points. */
int tc2li_project_local_map(const float pose7[7], const float cam4[4], float bf, const float* scale_factors, int n_levels,
                            float log_scale_factor, int cols, int rows, int n, const tc2li_map_point* points, float th,
                            int far_points, float th_far_points, float viewing_cos_limit, tc2li_proj_query* queries);

/* ------------------------------------------------------------------------------------------------
 * Optimisation back end.  Poses are Tcw as 7 doubles (qx, qy, qz, qw, tx, ty, tz) -- g2o::SE3Quat of
 * VertexSE3Expmap (Thirdparty/g2o/g2o/types/types_six_dof_expmap.h:60-77); map points 3 doubles.
 * ---------------------------------------------------------------------------------------------- */
typedef struct tc2li_camera { double fx, fy, cx, cy, bf; } tc2li_camera;

/* One projection edge: EdgeStereoSE3ProjectXYZ[OnlyPose] when u_right >= 0, else the monocular
 * EdgeSE3ProjectXYZ[OnlyPose]; information = inv_sigma2 * I (mvInvLevelSigma2[octave]). */
typedef struct tc2li_ba_edge {
    int32_t point, pose; /* indices into the point / pose arrays of the call */
    double u, v, u_right, inv_sigma2;
} tc2li_ba_edge;

/* Optimizer::PoseOptimization(Frame*) (SF/src/Optimizer.cc:816; callers Tracking.cc:2628,2796,2858): motion-only
 * optimisation of one frame pose against its n map-point correspondences (Xw[3*edges[i].point], edges[i].pose
 * ignored), 4 rounds x optimize(10) with Huber sqrt(5.991)/sqrt(7.815) and chi2 gates 5.991/7.815.  pose7 is updated
 * (rounded through float like Frame::SetPose), outlier[i] = mvbOutlier; returns nInitialCorrespondences - nBad. */
int tc2li_pose_optimization(double pose7[7], const double* Xw, const tc2li_ba_edge* edges, int n, const tc2li_camera* cam,
                            uint8_t* outlier);
/* Many frames in one launch (one workgroup per frame): frame f owns edges/Xw/outlier [edge_offsets[f], edge_offsets[f+1]),
 * its edges' `point` index is relative to that range. */
int tc2li_pose_optimization_batch(int n_frames, double* poses7, const int32_t* edge_offsets, const double* Xw,
                                  const tc2li_ba_edge* edges, const tc2li_camera* cam, uint8_t* outlier, int32_t* n_inliers,
                                  void* stream);

/* Statistics of one bundle adjustment (all optional). */
typedef struct tc2li_ba_stats {
    int32_t iterations, trials, n_free_poses, pad_;
    double initial_chi2, final_chi2, final_lambda;
} tc2li_ba_stats;

/* The optimisation of Optimizer::LocalBundleAdjustment / OptimizerWithLidar::LocalLVBundleAdjustment (visual edges;
 * SF/src/Optimizer.cc:1118, SF/src/OptimizerWithLidar.cc:60; callers LocalMapping.cc:170,173): the host shim gathers
 * local / fixed keyframes and map points exactly as the reference does (OptimizerWithLidar.cc:63-130) and passes them
 * flattened -- poses in vertex-id order with their fixed flags, points, one edge per observation.  Runs
 * optimizer.optimize(iterations) with Huber sqrt(5.991) / sqrt(7.815); lambda_init <= 0 selects tau * max diagonal,
 * the inertial-map branch passes 100 (OptimizerWithLidar.cc:141-142).  stop_flag is *pbStopFlag, polled between
 * Levenberg trials like g2o's forceStopFlag.  Outputs: poses and points updated in place (double; the shim casts to
 * float, :468-484), per-edge chi2 as the optimiser left it and isDepthPositive() for the outlier rules (:402-449).
 * Returns the number of iterations performed.  Limits of the graph: every point has an edge and at most 256 edges in all.  Several edges
 * between a point and the same optimisable keyframe are added into the same Hessian blocks, as g2o does
 * (Thirdparty/g2o/g2o/core/base_binary_edge.hpp:55-137; since round 5 -- rounds 2-4 returned TC2LI_ERR_INVALID for such a pair). */
int tc2li_local_bundle_adjustment(double* poses7, const uint8_t* fixed, int n_poses, double* points3, int n_points,
                                  const tc2li_ba_edge* edges, int n_edges, const tc2li_camera* cam, int iterations,
                                  double lambda_init, const volatile uint8_t* stop_flag, double* edge_chi2,
                                  uint8_t* edge_depth_positive, tc2li_ba_stats* stats, void* stream);

/* ------------------------------------------------------------------------------------------------
 * IMU pre-integration (camera-LiDAR-inertial configuration) -- replaces IMU::Preintegrated (SF/src/ImuTypes.cc:152-316),
 * the sample interpolation of Tracking::PreintegrateIMU (SF/src/Tracking.cc:1710-1822) and Tracking::PredictStateIMU
 * (:1825-1875).  Host-only (about ten float samples per frame); the result feeds the inertial edges of the local BA.
 * Matrices are row-major floats.
 * ---------------------------------------------------------------------------------------------- */
typedef struct tc2li_imu_sample { double t; float a[3], w[3]; } tc2li_imu_sample;              /* IMU::Point */
typedef struct tc2li_imu_bias { float bax, bay, baz, bwx, bwy, bwz; } tc2li_imu_bias;          /* IMU::Bias */
typedef struct tc2li_preintegrated {                                                           /* IMU::Preintegrated */
    float dT;
    int32_t n_measurements;
    float dR[9], dV[3], dP[3], JRg[9], JVg[9], JVa[9], JPg[9], JPa[9], avgA[3], avgW[3];
    float C[225];                  /* 15 x 15 covariance: rotation, velocity, position, gyro walk, acc walk */
    float noise[6], noise_walk[6]; /* diagonals of Nga / NgaWalk (IMU::Calib::Set, ImuTypes.cc:403-416) */
    tc2li_imu_bias bias;           /* the bias the integration was made with */
} tc2li_preintegrated;

/* Preintegrated(bias, calib): ng, na, ngw, naw as passed to IMU::Calib::Set. */
int tc2li_imu_preintegrated_init(tc2li_preintegrated* p, const tc2li_imu_bias* bias, float ng, float na, float ngw, float naw);
/* Preintegrated::IntegrateNewMeasurement */
int tc2li_imu_integrate(tc2li_preintegrated* p, const float acc[3], const float ang_vel[3], float dt);
/* The loop of Tracking::PreintegrateIMU over mvImuFromLastFrame (samples between the two frame stamps, one before and
 * one after included); returns the number of integration steps. */
int tc2li_imu_preintegrate(tc2li_preintegrated* p, const tc2li_imu_sample* samples, int n_samples, double t_prev, double t_cur);
/* The same for one frame of each of n_frames sequences (a batch of tracking threads): pre[f] is initialised at bias[f] with the calibration's
 * noise values and integrates samples[sample_offsets[f] .. sample_offsets[f + 1]) between t_prev[f] and t_cur[f].  Returns n_frames. */
int tc2li_imu_preintegrate_frames(int n_frames, tc2li_preintegrated* pre, const tc2li_imu_bias* bias, float ng, float na, float ngw, float naw,
                                  const tc2li_imu_sample* samples, const int32_t* sample_offsets, const double* t_prev, const double* t_cur);
/* GetDeltaRotation / GetDeltaVelocity / GetDeltaPosition at another bias (outputs may be NULL) */
int tc2li_imu_delta(const tc2li_preintegrated* p, const tc2li_imu_bias* bias, float dR[9], float dV[3], float dP[3]);
/* Tracking::PredictStateIMU: (Rwb1, twb1, Vwb1) of the last keyframe / frame -> the current frame's IMU state */
int tc2li_imu_predict_state(const tc2li_preintegrated* p, const tc2li_imu_bias* bias, const float Rwb1[9], const float twb1[3],
                            const float Vwb1[3], float Rwb2[9], float twb2[3], float Vwb2[3]);

/* IMU initialisation (host code by design: tens of keyframes, one 9-d edge per consecutive pair).
 * tc2li_imu_init_gravity = the first estimate of LocalMapping::InitializeIMU (SF/src/LocalMapping.cc:1241-1270): keyframes in temporal
 * order (Rwb [n][9] row-major, twb [n][3], float), pre[i] = keyframe i's pre-integration from keyframe i - 1 (pre[0] ignored, NULL
 * entries skipped) -> the finite-difference velocities vel3 [n][3] and Rwg (gravity direction of the IMU world).  Returns the links used.
 * tc2li_inertial_optimization = Optimizer::InertialOptimization(pMap, Rwg, scale, bg, ba, bMono, covInertial, bFixedVel, bGauss, priorG,
 * priorA) (SF/src/Optimizer.cc:2169-2356): Levenberg-Marquardt (lambda 1e3 when prior_g != 0, `iterations` = 200 in the reference,
 * g2o's stop rules) over the keyframe velocities (in / out, double), one gyro and one accelerometer bias (in: the first keyframe's; out:
 * the estimate), the gravity direction Rwg (in / out) and, when mono, the scale (in / out); fixed_vel freezes velocities and biases.
 * The pre-integrations are evaluated at the estimated biases through their bias Jacobians (SetNewBias + GetDelta*(b)).  What the
 * reference does with the result on its objects (SetVelocity / SetNewBias / Reintegrate, ApplyScaledRotation) stays with the caller.
 * Returns the iterations run. */
typedef struct tc2li_inertial_init_stats { int32_t iterations, trials; double initial_chi2, final_chi2, final_lambda; } tc2li_inertial_init_stats;
int tc2li_imu_init_gravity(int n_kfs, const float* Rwb9, const float* twb3, const tc2li_preintegrated* const* pre, float* vel3, float Rwg9[9]);
/* Optimizer::InertialOptimization(pMap, Rwg, scale), the second overload (SF/src/Optimizer.cc:2359-2466; LocalMapping::ScaleRefinement):
 * Gauss-Newton, `iterations` = 10 in the reference, gravity direction and scale only; the keyframes' velocities and biases ([n][3] each,
 * edge i takes keyframe i - 1's biases) are fixed, every edge carries Huber(1).  chi2 (may be NULL) = activeRobustChi2 before / after.
 * Returns the iterations run. */
int tc2li_inertial_scale_refinement(int n_kfs, const double* Rwb9, const double* twb3, const double* vel3, const double* bg3, const double* ba3,
                                    const tc2li_preintegrated* const* pre, double Rwg9[9], double* scale, int iterations, double chi2[2]);
int tc2li_inertial_optimization(int n_kfs, const double* Rwb9, const double* twb3, double* vel3, const tc2li_preintegrated* const* pre,
                                double Rwg9[9], double* scale, double bg[3], double ba[3], int mono, int fixed_vel, float prior_g, float prior_a,
                                int iterations, tc2li_inertial_init_stats* stats);


/* ------------------------------------------------------------------------------------------------
 * Tracking::TrackWithMotionModel (SF/src/Tracking.cc:2737-2834), data path only, for a batch of independent frames
 * whose features are device-resident: SearchByProjection(cur, last, th) with ORBmatcher(0.9, true), the 2*th retry
 * when fewer than 20 matches, Optimizer::PoseOptimization, outlier bookkeeping.  Frame f is images 2f / 2f+1 of the
 * handle's last tc2li_orb_extract_batch call; keypoints ([2*n_frames][capacity]) and u_right ([n_frames][capacity])
 * are the host arrays that call and tc2li_stereo_match_batch returned.
 * ---------------------------------------------------------------------------------------------- */
typedef struct tc2li_last_frame {      /* what the matcher reads of mLastFrame */
    int32_t n;
    int32_t pad_;
    const uint8_t* has_point;          /* mvpMapPoints[i] != NULL */
    const uint8_t* outlier;            /* mvbOutlier[i] */
    const float* Xw;                   /* pMP->GetWorldPos(), 3 per keypoint */
    const tc2li_keypoint* keys;        /* mvKeysUn (octave, angle are read) */
    const uint8_t* descriptors;        /* pMP->GetDescriptor(), 32 B per keypoint */
    float pose7[7];                    /* LastFrame.GetPose() */
    float pad2_;
} tc2li_last_frame;

/* pose_pred7 [n_frames][7] = mVelocity * mLastFrame.GetPose() (Sophus::SE3f).  Outputs: poses7 [n_frames][7] (double:
 * the optimised pose rounded through float as Frame::SetPose does, or the prediction when tracking failed),
 * map_point_of_keypoint [n_frames][capacity] = index of the last-frame point now held by keypoint i (mvpMapPoints[i])
 * or -1, outliers already discarded; n_matches[f] = matches after the search stage; n_inliers[f] = the value
 * PoseOptimization returned, -1 when fewer than 20 matches were found (Tracking.cc:2785-2793). */
int tc2li_track_motion_model_batch(tc2li_orb* orb, int n_frames, const tc2li_keypoint* keypoints, const float* u_right,
                                   int capacity, const tc2li_last_frame* last, const float* pose_pred7,
                                   const tc2li_camera* cam, float b, float th, double* poses7,
                                   int32_t* map_point_of_keypoint, int32_t* n_matches, int32_t* n_inliers, void* stream);

/* The data path of Tracking::TrackLocalMap (SF/src/Tracking.cc:3119-3230) after TrackWithMotionModel, for the same batch of
 * frames: SearchLocalPoints (:3232-3294: Frame::isInFrustum(pMP, 0.5) + ORBmatcher(0.8).SearchByProjection(F, mvpLocalMapPoints,
 * th, mbFarPoints, mThFarPoints)), Optimizer::PoseOptimization over every map point the frame holds, mnMatchesInliers.
 * poses7 [n_frames][7] (float) = the frames' current poses; held [n_frames][capacity]: 0 = keypoint i holds no map point, 1 = holds
 * one with Observations() > 0 (it blocks the search), 2 = holds one without observations; held_Xw [n_frames][capacity][3] its
 * world position.  local_points + local_offsets [n_frames + 1]: per frame the local map points still to be matched (not bad,
 * mnLastFrameSeen != this frame).  th as chosen at :3262-3281.  Out: poses7_out (double), local_of_keypoint [n_frames][capacity] =
 * index (within the frame's list) of the local point now held by keypoint i or -1, outlier [n_frames][capacity] = mvbOutlier
 * of every held point, n_matches[f] = SearchByProjection's result, n_inliers[f] = mnMatchesInliers.  The caller applies the
 * sensor-specific clean-up (:3198-3199) and the thresholds of :3205-3229. */
int tc2li_track_local_map_batch(tc2li_orb* orb, int n_frames, const tc2li_keypoint* keypoints, const float* u_right, int capacity,
                                const float* poses7, const uint8_t* held, const float* held_Xw, const tc2li_map_point* local_points,
                                const int32_t* local_offsets, const tc2li_camera* cam, float th, int far_points, float th_far_points,
                                double* poses7_out, int32_t* local_of_keypoint, uint8_t* outlier, int32_t* n_matches,
                                int32_t* n_inliers, void* stream);
/* Tracking::SearchLocalPoints alone (SF/src/Tracking.cc:3232-3294), same arguments: with the IMU initialised TrackLocalMap does not call
 * PoseOptimization but PoseInertialOptimizationLastFrame / LastKeyFrame on the frame's map points (Tracking.cc:2857-2878;
 * tc2li_pose_inertial_optimization_batch), and TrackWithMotionModel is PredictStateIMU alone (:2746-2752).  local_of_keypoint and
 * n_matches as above. */
int tc2li_search_local_points_batch(tc2li_orb* orb, int n_frames, const tc2li_keypoint* keypoints, const float* u_right, int capacity,
                                    const float* poses7, const uint8_t* held, const float* held_Xw, const tc2li_map_point* local_points,
                                    const int32_t* local_offsets, const tc2li_camera* cam, float th, int far_points, float th_far_points,
                                    int32_t* local_of_keypoint, int32_t* n_matches, void* stream);

/* The LiDAR co-visibility window of LocalLVBundleAdjustment (SF/src/OptimizerWithLidar.cc:226-260): the first
 * min(6, .) local keyframes with a non-empty surface cloud, in list order.  Replaces LidarCovisRes::AddFromKeyFrame /
 * BuildVoxHess (SF/src/LidarRes.cc:32-80) and the EdgeLidarSE3 they feed (SF/include/G2oTypesWithLidar.h:88-236). */
typedef struct tc2li_lidar_window {
    int32_t n_keyframes;          /* win_size_, 1 .. 20 */
    int32_t pad_;
    const int32_t* pose_index;    /* [n_keyframes] rows of poses7, the order of eBalm->setVertex(i, .) */
    const float* cloud_xyz;       /* GetSurfacePcl() of the keyframes back to back, x y z per point, LiDAR frame */
    const int32_t* cloud_offsets; /* [n_keyframes + 1], in points; every keyframe must have points */
    float Tcl[7];                 /* mLidarParam->mTcl as qx qy qz qw tx ty tz */
    float pad2_;
    double weight;                /* mLidarParam->mWeightLocalBA (the edge's information) */
} tc2li_lidar_window;

typedef struct tc2li_lidar_ba_stats {
    int32_t n_planes, hessian_evaluations;
    double residual, chi2;        /* the edge's last error and chi2 */
} tc2li_lidar_ba_stats;

/* OptimizerWithLidar::LocalLVBundleAdjustment with the LiDAR edge (SF/src/OptimizerWithLidar.cc:60-487): the arguments of
 * tc2li_local_bundle_adjustment plus the window.  The planes are extracted at the input poses; the optimiser then
 * minimises the visual cost + weight * (sum over planes of N * lambda_min)^2 as the reference's edge does, including
 * its bookkeeping (the Hessian is kept while the LiDAR cost grows, 6x6 blocks read at element offsets).  lidar == NULL
 * is the visual-only optimisation.  Returns the number of iterations performed. */
int tc2li_local_lv_bundle_adjustment(double* poses7, const uint8_t* fixed, int n_poses, double* points3, int n_points,
                                     const tc2li_ba_edge* edges, int n_edges, const tc2li_camera* cam, int iterations,
                                     double lambda_init, const volatile uint8_t* stop_flag, double* edge_chi2,
                                     uint8_t* edge_depth_positive, tc2li_ba_stats* stats, const tc2li_lidar_window* lidar,
                                     tc2li_lidar_ba_stats* lidar_stats, void* stream);

/* Many independent windows (multi-sequence operation, BASELINE configs[4]): every problem is what one
 * tc2li_local_lv_bundle_adjustment call takes; up to max_concurrency of them are in flight at a time, each on its own
 * HIP stream with its own device workspace, so that the small kernels of different windows overlap on the GPU.
 * results[i] receives the return value of problem i.  Returns the number of problems that succeeded. */
typedef struct tc2li_ba_problem {
    double* poses7; const uint8_t* fixed; double* points3; const tc2li_ba_edge* edges;
    int32_t n_poses, n_points, n_edges, iterations;
    double lambda_init;
    const volatile uint8_t* stop_flag;
    double* edge_chi2; uint8_t* edge_depth_positive;
    tc2li_ba_stats* stats;
    const tc2li_lidar_window* lidar; tc2li_lidar_ba_stats* lidar_stats;
} tc2li_ba_problem;
int tc2li_local_bundle_adjustment_batch(const tc2li_ba_problem* problems, int n_problems, const tc2li_camera* cam,
                                        int max_concurrency, int32_t* results);
/* The same for callers that run several local-mapping workers of their own (one LocalMapping thread per sequence in the reference,
 * SF/src/LocalMapping.cc:66-160; a multi-sequence system has a pool of them): the windows of this call form ONE lock-step group on the
 * context `group` (0 .. 7: its stream, device work spaces and host pool).  Calls on different groups run side by side and return
 * independently -- no worker waits for the slowest group of a common call --, calls on the same group serialise.  Every window's result is
 * the one tc2li_local_bundle_adjustment_batch / tc2li_local_lv_bundle_adjustment give for it. */
int tc2li_local_bundle_adjustment_batch_group(const tc2li_ba_problem* problems, int n_problems, const tc2li_camera* cam, int group,
                                              int32_t* results);

/* The same windows through a running ENGINE instead of a call per batch: the local-mapping threads of many sequences (one
 * LocalMapping::Run loop per sequence, SF/src/LocalMapping.cc:66-160, each reaching Optimizer::LocalBundleAdjustment /
 * OptimizerWithLidar::LocalLVBundleAdjustment at its own time) submit their windows as they come and collect them one ticket at a time.
 * The engine keeps up to max_windows windows in flight in ONE lock-step Levenberg-Marquardt queue on a stream of its own: a window joins
 * the queue at the next round after its setup, leaves it at the round its optimisation ends, and its slot is handed to the next waiting
 * window -- no window waits for the slowest one of a batch, and the host work of setting a window up and of writing its results back
 * runs beside the rounds of the others.  Every window's result is bit for bit the one tc2li_local_bundle_adjustment_batch /
 * tc2li_local_lv_bundle_adjustment give for it (a window's arithmetic never depends on its neighbours in the queue).
 *   submit: the windows of `problems` (arrays that stay valid and untouched until the ticket has been waited for) -> ticket > 0, or an
 *           error code < 0; results[i] receives window i's return value.  May be called from any thread, also while tickets are open.
 *   poll:   1 when every window of the ticket has finished (wait will not block), 0 while one is still in the queue.
 *   wait:   blocks until every window of the ticket has finished -> number of windows that succeeded; a ticket is collected once.
 *   destroy: finishes the windows already submitted, then stops the engine's thread. */
typedef struct tc2li_ba_engine tc2li_ba_engine;
int tc2li_ba_engine_create(const tc2li_camera* cam, int max_windows, tc2li_ba_engine** out);
void tc2li_ba_engine_destroy(tc2li_ba_engine* engine);
int64_t tc2li_ba_engine_submit(tc2li_ba_engine* engine, const tc2li_ba_problem* problems, int n_problems, int32_t* results);
int tc2li_ba_engine_poll(tc2li_ba_engine* engine, int64_t ticket);
int tc2li_ba_engine_wait(tc2li_ba_engine* engine, int64_t ticket);

/* One window split over the GPUs of a node (BASELINE configs[4]): the landmarks -- and with them the stereo / mono
 * edges, W, Hll and the back-substitution -- are partitioned over the ranks (landmark l belongs to rank l % world); every rank
 * keeps all keyframe poses.  What the ranks exchange are the shared-pose blocks only: per LM trial ONE sum of
 * [S | b_schur | b_p] (the rank's part of the reduced camera system: its Hpp, minus its landmarks' W Hll^-1 W^T), and one sum
 * of [scale, chi2] after the trial update; once per call the max / sum that g2o's initial lambda needs, and at the end one
 * sum that hands every rank all points, per-edge chi2 and depth flags.  The LM control flow, the LDL^T of the reduced system
 * and the LiDAR edge are replicated (they are deterministic, so every rank takes the same decisions).
 * The reference has no counterpart: its g2o solver is single-threaded (SF/Thirdparty/g2o/config.h:4); this entry is the
 * "RCCL all-reduce of the shared-pose Hessian" BASELINE.json names.  EVERY rank passes the SAME arguments (the whole window)
 * and every rank receives the whole result.  The sums run in rank order inside the collective, so the result agrees with the
 * single-GPU entry to rounding, not bit for bit.
 *
 * allreduce(ctx, device_buf, count, op, stream): in-place all-reduce of `count` doubles in device memory, enqueued on (or
 * ordered after the work already on) `stream`; returns 0 on success.  tc2li_rccl_allreduce below is such a function over an
 * RCCL communicator; a host may pass its own (torch.distributed, MPI). */
enum { TC2LI_REDUCE_SUM = 0, TC2LI_REDUCE_MAX = 1 };
typedef int (*tc2li_allreduce_fn)(void* ctx, double* device_buf, size_t count, int op, void* stream);
typedef struct tc2li_ba_shard {
    int32_t rank, world;
    tc2li_allreduce_fn allreduce;
    void* ctx;
} tc2li_ba_shard;
int tc2li_local_lv_bundle_adjustment_sharded(double* poses7, const uint8_t* fixed, int n_poses, double* points3, int n_points,
                                             const tc2li_ba_edge* edges, int n_edges, const tc2li_camera* cam, int iterations,
                                             double lambda_init, const volatile uint8_t* stop_flag, double* edge_chi2,
                                             uint8_t* edge_depth_positive, tc2li_ba_stats* stats, const tc2li_lidar_window* lidar,
                                             tc2li_lidar_ba_stats* lidar_stats, const tc2li_ba_shard* shard, void* stream);
/* The rank's share of a window: landmark_owned[l] = 1 where l % world == rank, edge_owned[e] likewise for the edge's landmark.
 * Host logic only (no device needed).  Returns the number of owned edges. */
int tc2li_ba_shard_select(const tc2li_ba_edge* edges, int n_edges, int n_points, int rank, int world, uint8_t* landmark_owned,
                          uint8_t* edge_owned);

/* RCCL glue for the sharded window (one process per GPU; librccl is loaded on first use, the library has no link-time
 * dependency on it).  unique_id: 128 bytes made by rank 0 with tc2li_rccl_unique_id and handed to the other ranks by the
 * launcher (bench.py broadcasts it over torch.distributed).  tc2li_rccl_allreduce has the tc2li_allreduce_fn signature with
 * ctx = the communicator. */
int tc2li_rccl_unique_id(void* unique_id_128);
int tc2li_rccl_comm_create(const void* unique_id_128, int rank, int world, void** comm);
int tc2li_rccl_comm_destroy(void* comm);
int tc2li_rccl_allreduce(void* comm, double* device_buf, size_t count, int op, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Visual-inertial local bundle adjustment -- the optimisation of Optimizer::LocalInertialBA (SF/src/Optimizer.cc:1512-2085;
 * caller LocalMapping.cc:158,161): vertices VertexPose (ImuCamPose, body-frame update), VertexVelocity, VertexGyroBias,
 * VertexAccBias per keyframe and marginalised points; edges EdgeMono / EdgeStereo (Huber), EdgeInertial (+ Huber
 * sqrt(16.92) where `robust`), EdgeGyroRW, EdgeAccRW (SF/include/G2oTypes.h, SF/src/G2oTypes.cc).  The host shim gathers
 * the temporal window, the fixed keyframes and the points exactly as :1520-1640 does and passes them flattened, keyframes in
 * vertex-id order.  Projection edges run on the GPU, the few inertial edges on the host (row c6); the reduced system
 * (6 + 9 unknowns per optimisable keyframe; g2o's sparse LinearSolverEigen, :1635-1638) is solved on the GPU inside the envelope of
 * the velocity / bias band (windows of at most 25 optimisable keyframes whose inertial edges join keyframes at most two places apart
 * in the numbering; other windows, and every window with TC2LI_LVI_DEVICE_SOLVE=0, on the host by the same elimination).
 * ---------------------------------------------------------------------------------------------- */
typedef struct tc2li_inertial_keyframe {  /* ImuCamPose(KeyFrame*) + velocity + biases, widened from the map's floats */
    double Rcw[9], tcw[3];                /* GetRotation(), GetTranslation() */
    double Rwb[9], twb[3];                /* GetImuRotation(), GetImuPosition() */
    double velocity[3], gyro_bias[3], acc_bias[3];
} tc2li_inertial_keyframe;
typedef struct tc2li_imu_calib { double Rcb[9], tcb[3], Rbc[9], tbc[3]; } tc2li_imu_calib;  /* mImuCalib.mTcb / mTbc */
typedef struct tc2li_inertial_link {      /* EdgeInertial + EdgeGyroRW + EdgeAccRW between keyframe kf1 (= kf2->mPrevKF) and kf2 */
    int32_t kf1, kf2;
    int32_t robust;                       /* i == N-1 || bRecInit (Optimizer.cc:1757-1766) */
    int32_t pad_;
    double info_scale;                    /* 1e-2 for the link to the keyframe before the window, else 1 */
    const tc2li_preintegrated* preintegrated; /* kf2->mpImuPreintegrated after SetNewBias(kf1->GetImuBias()) */
} tc2li_inertial_link;

/* fixed[k]: the keyframe's pose, velocity and biases are constants; has_imu[k] = pKFi->bImu.  iterations / lambda_init:
 * 10 and 1e0, or 4 and 1e-2 for bLarge (:1516-1523, 1637-1650).  stats->initial_chi2 / final_chi2 are activeRobustChi2()
 * before and after optimize() (err / err_end of :1968-1971).  Keyframes and points are updated in place; edge_chi2 /
 * edge_depth_positive as in tc2li_local_bundle_adjustment.  Returns the number of iterations performed. */
int tc2li_local_inertial_bundle_adjustment(tc2li_inertial_keyframe* keyframes, const uint8_t* fixed, const uint8_t* has_imu,
                                           int n_keyframes, const tc2li_imu_calib* calib, double* points3, int n_points,
                                           const tc2li_ba_edge* edges, int n_edges, const tc2li_inertial_link* links,
                                           int n_links, const tc2li_camera* cam, int iterations, double lambda_init,
                                           const volatile uint8_t* stop_flag, double* edge_chi2,
                                           uint8_t* edge_depth_positive, tc2li_ba_stats* stats, void* stream);

/* OptimizerWithLidar::LocalLVIBA (SF/src/OptimizerWithLidar.cc:489-1100): the visual-inertial local BA plus the LiDAR edge
 * on the body-frame pose vertices (EdgeLidar, SF/src/G2oTypesWithLidar.cc:33-140; LidarCovisRes::ComputeJandH,
 * SF/src/LidarRes.cc:89-128).  lidar->pose_index are rows of `keyframes` -- the reference takes the first min(N, 6) optimisable
 * keyframes when N > 5 (:704-724); Tbl = mLidarParam->mTbl as qx qy qz qw tx ty tz.  The edge's error is sqrt(sum over planes of
 * N * lambda_min), its information lidar->weight.  lidar == NULL is tc2li_local_inertial_bundle_adjustment. */
int tc2li_local_lvi_bundle_adjustment(tc2li_inertial_keyframe* keyframes, const uint8_t* fixed, const uint8_t* has_imu,
                                      int n_keyframes, const tc2li_imu_calib* calib, double* points3, int n_points,
                                      const tc2li_ba_edge* edges, int n_edges, const tc2li_inertial_link* links, int n_links,
                                      const tc2li_camera* cam, int iterations, double lambda_init,
                                      const volatile uint8_t* stop_flag, double* edge_chi2, uint8_t* edge_depth_positive,
                                      tc2li_ba_stats* stats, const tc2li_lidar_window* lidar, const float* Tbl,
                                      tc2li_lidar_ba_stats* lidar_stats, void* stream);

/* Many independent LocalLVIBA windows (the local-mapping threads of many sequences in the camera-LiDAR-inertial configuration): every
 * problem is what one tc2li_local_lvi_bundle_adjustment call takes (same IMU calibration and camera for all).  With max_concurrency > 1
 * the windows advance through the Levenberg-Marquardt phases in lock step like tc2li_local_bundle_adjustment_batch's -- one launch per
 * kernel and one synchronisation per phase for all windows (Schur product, solve of the reduced system and trial estimate are one
 * queue with one synchronisation per Levenberg trial), the inertial edges on host threads between the phases -- and every window's
 * result is the one of the one-window call.  (A window's reduced system is solved on the device or by the host's envelope LDL^T according
 * to the window's own shape -- at most 25 optimisable keyframes with IMU state, inertial edges at most two keyframes apart --; the two
 * solvers agree to 1e-9 relative, not bit for bit.  A lock-step group runs one of them: windows that differ from their group's majority
 * go through the one-window call inside the batch call.)  results[i] = iterations of window i or its error code;
 * returns the number of windows that succeeded. */
typedef struct tc2li_lvi_problem {
    tc2li_inertial_keyframe* keyframes; const uint8_t* fixed; const uint8_t* has_imu;
    double* points3; const tc2li_ba_edge* edges; const tc2li_inertial_link* links;
    int32_t n_keyframes, n_points, n_edges, n_links, iterations, pad_;
    double lambda_init;
    const volatile uint8_t* stop_flag;
    double* edge_chi2; uint8_t* edge_depth_positive;
    tc2li_ba_stats* stats;
    const tc2li_lidar_window* lidar; const float* Tbl; tc2li_lidar_ba_stats* lidar_stats;
} tc2li_lvi_problem;
int tc2li_local_lvi_bundle_adjustment_batch(const tc2li_lvi_problem* problems, int n_problems, const tc2li_imu_calib* calib,
                                            const tc2li_camera* cam, int max_concurrency, int32_t* results);
/* The same as ONE lock-step group on the context `group` (0 .. 7), for callers with several local-mapping workers: as
 * tc2li_local_bundle_adjustment_batch_group.  Every window's result is the one tc2li_local_lvi_bundle_adjustment gives for it. */
int tc2li_local_lvi_bundle_adjustment_batch_group(const tc2li_lvi_problem* problems, int n_problems, const tc2li_imu_calib* calib,
                                                  const tc2li_camera* cam, int group, int32_t* results);

/* ---- local mapping: new map points ----
 * What ORBmatcher::SearchForTriangulation (SF/src/ORBmatcher.cc:916) and the pair loop of LocalMapping::CreateNewMapPoints
 * (SF/src/LocalMapping.cc:402-726) read of a keyframe (pinhole camera, no second camera model). */
typedef struct tc2li_keyframe_view {
    int32_t n, n_nodes;            /* keypoints; entries of mFeatVec */
    const tc2li_keypoint* keys;    /* mvKeysUn */
    const uint8_t* descriptors;    /* mDescriptors, [n][32] */
    const float* u_right;          /* mvuRight */
    const float* depth;            /* mvDepth */
    const uint8_t* has_point;      /* GetMapPoint(i) != NULL */
    const int32_t* fv_node;        /* mFeatVec: node ids, ascending */
    const int32_t* fv_offset;      /* [n_nodes + 1] */
    const int32_t* fv_index;       /* the feature indices of every node, in insertion order */
    float pose7[7];                /* GetPose(): qx qy qz qw tx ty tz of Tcw */
    float pad_;
} tc2li_keyframe_view;

/* ORBmatcher::SearchForTriangulation(pKF1, pKF2, vMatchedPairs, bOnlyStereo, bCoarse): match12[i] = matched keypoint of kf2 or
 * -1 for every keypoint of kf1 (vMatchedPairs = the pairs with match12[i] >= 0, i ascending).  level_sigma2 = mvLevelSigma2,
 * scale_factors = mvScaleFactors.  Returns nmatches. */
int tc2li_search_for_triangulation(const tc2li_keyframe_view* kf1, const tc2li_keyframe_view* kf2, const tc2li_camera* cam,
                                   const float* scale_factors, const float* level_sigma2, int n_levels, int only_stereo,
                                   int coarse, int check_orientation, int32_t* match12, void* stream);

typedef struct tc2li_new_map_point {
    int32_t idx1, neighbour, idx2; /* keypoint of the current keyframe, index into `neighbours`, keypoint there */
    int32_t stereo;                /* the point came from UnprojectStereo (bPointStereo) */
    float x3D[3];
    float pad_;
} tc2li_new_map_point;

/* The geometric loop of LocalMapping::CreateNewMapPoints over the neighbours the caller chose (vpNeighKFs, in order): search,
 * parallax gates, triangulation or stereo un-projection, depth / reprojection / scale gates; a keypoint that received a point
 * from an earlier neighbour is skipped for the later ones.  mb = pKF->mb, cam->bf = mbf, scale_factor = mfScaleFactor, inertial =
 * mbInertial, far_points / th_far_points = mbFarPoints / mThFarPoints, coarse = bCoarse.  The points come in creation order;
 * the caller creates the MapPoint objects, adds the observations and refreshes them (tc2li_map_points_refresh).  Triangulated
 * coordinates agree with the reference to float rounding (Eigen's JacobiSVD is not reproduced bit for bit).  Returns the count. */
int tc2li_create_new_map_points(const tc2li_keyframe_view* current, const tc2li_keyframe_view* neighbours, int n_neighbours,
                                const tc2li_camera* cam, float mb, const float* scale_factors, const float* level_sigma2,
                                int n_levels, float scale_factor, int inertial, int far_points, float th_far_points, int coarse,
                                tc2li_new_map_point* points, int capacity, void* stream);

/* ORBmatcher::Fuse(pKF, vpMapPoints, th, bRight = false), the search (SF/src/ORBmatcher.cc:1157-1330; called from
 * LocalMapping::SearchInNeighbors :728-837): for every map point the keypoint of the keyframe it is fused with -- projection and
 * gates (depth, image, distance range, viewing direction), MapPoint::PredictScale, the keypoints inside th * scale on the
 * keyframe's feature grid, level and reprojection gates (7.8 / 5.99 on mvInvLevelSigma2), least descriptor distance <= TH_LOW --
 * or -1.  valid[i] = pMP && !pMP->isBad() && !pMP->IsInKeyFrame(pKF).  A point's result depends on no other point; the caller
 * walks the results in list order and does Replace / AddObservation on its objects (re-checking isBad / IsInKeyFrame there).
 * best_dist may be NULL.  Returns how many points found a keypoint. */
int tc2li_fuse_search(const tc2li_frame_view* keyframe, const float pose7[7], const float cam4[4], float bf, const float* scale_factors,
                      const float* inv_level_sigma2, int n_levels, float log_scale_factor, const tc2li_map_point* points,
                      const uint8_t* valid, int n_points, float th, int32_t* best_idx, int32_t* best_dist, void* stream);

/* Per-map-point refresh of local mapping after a local BA / after creating or fusing points (LocalMapping.cc, Optimizer.cc:1506,
 * OptimizerWithLidar.cc:484 `pMP->UpdateNormalAndDepth()`; `ComputeDistinctiveDescriptors` in CreateNewMapPoints / Fuse):
 * MapPoint::ComputeDistinctiveDescriptors (SF/src/MapPoint.cc:338-412) and MapPoint::UpdateNormalAndDepth (:444-503) for a flat
 * list of points.  Observations of point p are [obs_offsets[p], obs_offsets[p + 1]) in the iteration order of mObservations
 * (left, then right index of every keyframe that is not bad): obs_descriptors [total][32], obs_centres [total][3] = the
 * observing camera's centre.  positions / ref_centres (mpRefKF->GetCameraCenter()) [n][3], ref_level_scale[p] =
 * mvScaleFactors[octave of the reference observation], last_level_scale = mvScaleFactors[nLevels - 1].
 * Out: best_obs[p] = the observation whose descriptor becomes mDescriptor (-1: no observations, outputs untouched),
 * normals [n][3] = mNormalVector, min_distance / max_distance = mfMinDistance / mfMaxDistance.  At most 112 observations
 * per point (TC2LI_ERR_CAPACITY beyond).  Returns n_points. */
int tc2li_map_points_refresh(int n_points, const int32_t* obs_offsets, const uint8_t* obs_descriptors, const float* obs_centres,
                             const float* positions, const float* ref_centres, const float* ref_level_scale, float last_level_scale,
                             int32_t* best_obs, float* normals, float* min_distance, float* max_distance, void* stream);

/* Local-map bookkeeping that feeds SearchLocalPoints: Tracking::UpdateLocalKeyFrames + Tracking::UpdateLocalPoints
 * (SF/src/Tracking.cc:3326-3476, :3296-3323; caller Tracking::UpdateLocalMap :3286) on a device-resident mirror of the graph
 * pieces they read.  Keyframes and map points are indices into the mirror.  The reference keys its containers by object
 * address (std::map<KeyFrame*, int> keyframeCounter, std::set<KeyFrame*> children, std::map<KeyFrame*, ...> observations), so
 * its iteration order is the allocator's; here index order stands in for address order and the caller lists children /
 * observations in the order its containers iterate.
 *   covis        mvpOrderedConnectedKeyFrames per keyframe (GetBestCovisibilityKeyFrames(10) takes the first 10)
 *   children     GetChilds();  parent / prev_kf: GetParent() / mPrevKF, -1 = none
 *   matches      GetMapPointMatches(): the map point of every keypoint slot, -1 = none
 *   obs_kf       the keyframes of GetObservations() per map point */
typedef struct tc2li_map_graph {
    int32_t n_keyframes, n_points;
    const uint8_t* kf_bad;                               /* [n_keyframes] KeyFrame::isBad() */
    const int32_t *covis_offsets, *covis;                /* CSR over keyframes */
    const int32_t *child_offsets, *children;
    const int32_t *parent, *prev_kf;                     /* [n_keyframes] */
    const int32_t *match_offsets, *matches;
    const uint8_t* point_bad;                            /* [n_points] MapPoint::isBad() */
    const int32_t *obs_offsets, *obs_kf;                 /* CSR over map points */
} tc2li_map_graph;
typedef struct tc2li_local_map tc2li_local_map;
int tc2li_local_map_create(tc2li_local_map** out);
void tc2li_local_map_destroy(tc2li_local_map* map);
/* Uploads (replaces) the mirror; call when keyframes / points / observations changed.  The arrays are copied. */
int tc2li_local_map_set_graph(tc2li_local_map* map, const tc2li_map_graph* graph, void* stream);
/* One UpdateLocalMap.  frame_points = mCurrentFrame.mvpMapPoints (mLastFrame's once the IMU is initialised, :3350), -1 = none;
 * temporal_last_kf = mCurrentFrame.mpLastKeyFrame for IMU_STEREO_LIDAR (the temporal block :3453-3469), -1 otherwise.
 * Out: mvpLocalKeyFrames in the reference's order (voted keyframes by index, then the neighbour / child / parent extensions with
 * the reference's early exits, then up to 20 temporal keyframes), reference_kf = pKFmax (-1: none), mvpLocalMapPoints in the
 * reference's order (local keyframes walked backwards, slots forwards, first occurrence kept, bad points skipped), and
 * frame_point_cleared[i] = 1 where the reference sets the frame's point to NULL because it is bad.  The local point list also
 * stays on the device (tc2li_local_map_device_points).  Returns the number of local points; TC2LI_ERR_CAPACITY when a list does not fit. */
int tc2li_local_map_update(tc2li_local_map* map, const int32_t* frame_points, int n_frame_points, int temporal_last_kf,
                           int32_t* local_keyframes, int keyframe_capacity, int32_t* n_local_keyframes, int32_t* reference_kf,
                           int32_t* local_points, int point_capacity, int32_t* n_local_points, uint8_t* frame_point_cleared,
                           void* stream);
const int32_t* tc2li_local_map_device_points(const tc2li_local_map* map);

/* The LiDAR term alone at the poses poses7 (Tcw of the window keyframes are rows lidar->pose_index): planes from the
 * window, then *residual = LidarCovisRes::ComputeError() and JacT [6W] / Hessian [(6W)^2, row-major] =
 * LidarCovisRes::ComputeJandHSE3 (SF/src/LidarRes.cc:136-186, with respect to the camera se3 increments).  JacT and
 * Hessian may be NULL.  Returns the number of planes. */
int tc2li_lidar_window_evaluate(const double* poses7, int n_poses, const tc2li_lidar_window* lidar, double* residual,
                                double* JacT, double* Hessian, void* stream);

/* Host-only: the envelope LDL^T of the inertial windows' reduced system (csrc/reduced_solve.hpp -- what tc2li_local_lvi_bundle_adjustment uses
 * with TC2LI_LVI_DEVICE_SOLVE=0 and for windows the device solve does not take), exposed so that it can be checked without a GPU.
 * Hi [n][n]: the inertial + LiDAR part in the caller's numbering (np pose unknowns first, lower triangle read); S [np][np]: the visual Schur
 * complement with its damping (lower triangle read); lambda is added to the diagonal of the other n - np unknowns.  Solves for x [n] from rhs [n];
 * returns 1, or 0 when a pivot is zero or not finite. */
int tc2li_host_reduced_solve(const double* Hi, const double* S, int n, int np, double lambda, const double* rhs, double* x);
/* The same system through the device solve (k_lvi_solve: what the inertial windows use by default), for tests of the kernel alone: returns 1,
 * 0 for a failed pivot, TC2LI_ERR_INVALID when the kernel does not take the system (more than 150 pose unknowns, no velocity / bias
 * unknowns, a velocity / bias row wider than 28). */
int tc2li_device_reduced_solve(const double* Hi, const double* S, int n, int np, double lambda, const double* rhs, double* x, void* stream);

/* Host-only stage of the LiDAR term, exposed so that it can be checked without a GPU: the planes of the window
 * (cut_voxel + recut + tras_opt, SF/src/bavoxel.cc:42-91, SF/include/bavoxel.h:492-602,723-740).  clusters receives, per
 * plane and window keyframe, 10 doubles: P00 P01 P02 P11 P12 P22 (sum x x^T), v (sum x), N in the keyframe's LiDAR
 * frame; coe the plane weights.  Returns the number of planes (which may exceed `capacity`; only that many are written). */
int tc2li_host_lidar_planes(const double* poses7, int n_poses, const tc2li_lidar_window* lidar, double* clusters, double* coe,
                            int capacity);

/* The same planes from the kernels that extract them for the windows of the batched local BA entry points (round 4: cut_voxel / recut as
 * three stable sorts of the window's points and a plane test per cell, balm_cut_kernels.hip): same layout, and the same bits as
 * tc2li_host_lidar_planes.  info (may be NULL) receives [planes, declined, root voxels, planes found].  A window outside the kernels'
 * range (more than 7 keyframes, 65535 points or 2048 planes, coordinates beyond +-1e6 voxels) is TC2LI_ERR_INVALID here; the BA entry
 * points take the host extraction for such a window. */
int tc2li_device_lidar_planes(const double* poses7, int n_poses, const tc2li_lidar_window* lidar, double* clusters, double* coe,
                              int capacity, int32_t* info);

/* Host-only stage of the extractor, exposed so that it can be checked without a GPU: keypoint distribution of
 * ORBextractor::DistributeOctTree (SF/src/ORBextractor.cc:529-753).  Candidates are (x, y, response) triples with
 * integer-valued x, y in the border-free level frame, in cv::FAST emission order; writes the retained triples in
 * the reference's output order and returns their number. */
int tc2li_host_distribute_quadtree(const float* xyr, int n, int min_x, int max_x, int min_y, int max_y, int n_target,
                                   float* out_xyr, int capacity);
/* The same distribution as one job of the device kernels the extractor runs: `threads` = 0 is the extractor's choice (k_quadtree_sorted:
 * the keys sorted once by their path through the tree, everything in LDS; a job too large for LDS falls to k_quadtree; `threads` = -1 - c starts with LDS class c), `threads` =
 * 256 / 512 / 1024 forces k_quadtree (work arrays in global memory) with that many lanes.  Same arguments, same result.  The extractor itself calls the kernel on the device-resident
 * candidates of a whole batch; this entry exists to check the kernel on arbitrary candidate sets. */
int tc2li_device_distribute_quadtree(const float* xyr, int n, int min_x, int max_x, int min_y, int max_y, int n_target, float* out_xyr,
                                     int capacity, int threads);

/* ------------------------------------------------------------------------------------------------
 * Optimizer::PoseInertialOptimizationLastKeyFrame (SF/src/Optimizer.cc:2469-2852) and PoseInertialOptimizationLastFrame
 * (:2854-3270) -- the per-frame optimiser of Tracking::TrackLocalMap once the IMU is initialised (Tracking.cc:2872 / 2877) -- for a
 * batch of independent frames: pose, velocity and biases of the frame (and, in the last-frame form, of the previous frame) against the
 * map points the frame holds (EdgeMonoOnlyPose / EdgeStereoOnlyPose, Huber sqrt(5.991) / sqrt(7.815)), EdgeInertial + EdgeGyroRW +
 * EdgeAccRW to the other state and, in the last-frame form, EdgePriorPoseImu (Huber 5) on the previous frame's mpcpi; Gauss-Newton with
 * a dense LDL^T, 4 rounds x 10 iterations, the inlier tests {12, 7.5, 5.991, 5.991} (keyframe form) or 5.991 (last-frame form) for
 * monocular edges (x 1.5 for points with mTrackDepth < 10) and {15.6, 9.8, 7.815, 7.815} for stereo edges, the recovery pass when fewer
 * than 30 inliers remain and !bRecInit; then the frame's new prior: state + Hessian (the last-frame form marginalises the previous
 * frame out of the 30 x 30 system, Optimizer::Marginalize :2087-2166; eigenvalues below 1e-12 cleared as ConstraintPoseImu does).
 * The whole optimisation of every frame runs inside one kernel launch (one workgroup per frame).
 * ---------------------------------------------------------------------------------------------- */
typedef struct tc2li_pose_imu_prior {      /* ConstraintPoseImu (SF/include/G2oTypes.h:716-740): pFrame->mpcpi */
    double Rwb[9], twb[3], vwb[3], bg[3], ba[3];
    double H[225];                         /* 15 x 15: rotation, translation, velocity, gyro bias, accelerometer bias */
} tc2li_pose_imu_prior;
typedef struct tc2li_pose_inertial_problem {
    tc2li_inertial_keyframe frame;         /* in / out: VertexPose(pFrame), VertexVelocity, VertexGyroBias, VertexAccBias */
    tc2li_inertial_keyframe other;         /* mpLastKeyFrame (constant) or, with last_frame, mpPrevFrame (in / out) */
    const tc2li_pose_imu_prior* prior;     /* pFp->mpcpi; last_frame only */
    const tc2li_preintegrated* preintegrated;     /* EdgeInertial: mpImuPreintegrated (keyframe form) / mpImuPreintegratedFrame */
    const tc2li_preintegrated* preintegrated_rw;  /* the bias-walk covariance of InfoG / InfoA: always pFrame->mpImuPreintegrated (:2645, :3049) */
    const double* Xw;                      /* [n_edges][3]: pMP->GetWorldPos() widened */
    const tc2li_ba_edge* edges;            /* u, v, u_right (< 0: monocular), inv_sigma2; point / pose are not read */
    const uint8_t* close_point;            /* [n_edges]: mTrackDepth < 10 */
    uint8_t* outlier;                      /* out [n_edges]: mvbOutlier */
    tc2li_pose_imu_prior* prior_out;       /* out (may be NULL): the frame's new mpcpi */
    int32_t n_edges, last_frame, rec_init;
    int32_t n_initial, n_bad, n_inliers, solver_failed;  /* out */
    int32_t pad_;
} tc2li_pose_inertial_problem;
/* results[f] (may be NULL) = nInitialCorrespondences - nBad, the value the reference returns.  Returns n_frames. */
int tc2li_pose_inertial_optimization_batch(tc2li_pose_inertial_problem* problems, int n_frames, const tc2li_imu_calib* calib,
                                           const tc2li_camera* cam, int32_t* results, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Measurement (bench.py; not on the hot path).
 * tc2li_profile_enable(1): every kernel launch of the library is bracketed by two HIP events on the stream it is launched on.
 * tc2li_profile_report: call when the streams are idle; writes "name<TAB>launches<TAB>total_ms<NL>" per kernel (sorted by total
 * time) into text, forgets the recorded launches and returns the bytes the whole report needs.
 * tc2li_diag_peaks: what this GPU reaches on back-to-back v_mfma_f64_16x16x4_f64 (TFLOP/s), on f64 vector FMAs (TFLOP/s) and on a
 * 1 GiB float4 copy (GB/s, read + write) -- the peaks the roofline of bench.py is priced against next to the datasheet's 8 TB/s.
 * Any pointer may be NULL.
 * ---------------------------------------------------------------------------------------------- */
int tc2li_profile_enable(int on);
int tc2li_profile_report(char* text, int capacity);
/* The environment switches the bundle-adjustment entry points would run under if called now, as one line of JSON (what a benchmark logs beside
 * its numbers): {"device_lm": 1, "device_solve": 0, "fuse_linearize": 0, "fuse_trial": 0, "lvi_device_solve": 1, "lockstep": 1, "groups": 3}.
 * Returns the number of bytes the text needs (including the terminator); writes at most `capacity`.  No reference counterpart. */
int tc2li_ba_options(char* text, int capacity);
int tc2li_diag_peaks(double* mfma_f64_tflops, double* fma_f64_tflops, double* hbm_copy_gbps);
/* The shader clock (GHz) the chip holds inside the two arithmetic loops of tc2li_diag_peaks, from s_memtime against the constant 100 MHz
 * counter: the data sheet's 78.6 TFLOP/s of f64 matrix / vector arithmetic assume 2.4 GHz. */
int tc2li_diag_clocks(double* mfma_loop_ghz, double* fma_loop_ghz);

#ifdef __cplusplus
}
#endif
#endif /* TC2LI_HIP_H */
