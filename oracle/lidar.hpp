// TEST INFRASTRUCTURE ONLY -- CPU oracle, never linked into or called from the product path.
//
// Restatement of the LiDAR front end on the camera-LiDAR path:
//   Preprocess::velodyne_handler, non-feature branch      SF/include/lidar_front_end/preprocess.cpp:145-166
//   pcl::VoxelGrid<PointXYZINormal>::filter call sites     LidarFrontEnd.cpp:712-714, 913-915 (PCL 1.12, NOT in tree)
//   pointBodyToWorld                                       LidarFrontEnd.cpp:130-139
//   KD_TREE::Build / BuildTree / Nearest_Search / Search   SF/include/ikd-Tree/ikd_Tree.cpp:409-423,690-744,426-461,1074-1256
//   EstiPlane (Eigen colPivHouseholderQr, NOT in tree)     LidarFrontEnd.cpp:964-997
//   feature_extraction                                     LidarFrontEnd.cpp:999-1073
// PARITY UNPINNED: no reference tests/vectors exist for these; PCL and Eigen are un-vendored dependencies
// (CMakeLists.txt:72-73) whose published algorithms are restated here with sequential float summation order.
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>

namespace oracle {

struct VelodynePoint {  // velodyne_ros::Point, preprocess.h:62-70 (32 bytes)
    float x, y, z, pad0;
    float intensity, time;
    uint16_t ring, pad1;
    float pad2;
};
static_assert(sizeof(VelodynePoint) == 32, "layout");

struct PointXYZINormal {  // pcl::PointXYZINormal (48 bytes)
    float x, y, z, pad0;
    float normal_x, normal_y, normal_z, pad1;
    float intensity, curvature, pad2, pad3;
};
static_assert(sizeof(PointXYZINormal) == 48, "layout");

using PointVector = std::vector<PointXYZINormal>;

// preprocess.cpp:63-86 (time unit scale) and :145-166
PointVector preprocess_velodyne(const VelodynePoint* raw, int n, int point_filter_num, double blind, float time_unit_scale);

// pcl::VoxelGrid::applyFilter with downsample_all_data = true, min_points_per_voxel = 0
PointVector voxel_grid_filter(const PointVector& in, float leaf);

struct LidarState {  // the parts of state_ikfom that pointBodyToWorld reads (row-major 3x3)
    double rot[9], pos[3], offset_R_L_I[9], offset_T_L_I[3];
};
PointXYZINormal pointBodyToWorld(const PointXYZINormal& pi, const LidarState& s);

// ImuProcess::UndistortPcl, the point part (SF/include/lidar_front_end/IMU_Processing.cpp:170-172, 236-276): points sorted
// by curvature (time offset in ms) with std::sort(time_list), then every point compensated into the scan-end frame from
// the IMU poses saved during the forward propagation (Pose6D: offset_time, acc, gyr, vel, pos, rot).  `end` is imu_state
// after the last predict.  Includes the reference's handling of the first point (re-compensated once per earlier segment).
struct Pose6D { double offset_time, acc[3], gyr[3], vel[3], pos[3], rot[9]; };
void UndistortPcl(PointVector& pcl, const std::vector<Pose6D>& IMUpose, const LidarState& end);
// The forward propagation that produces IMUpose (IMU_Processing.cpp:176-233), state part of esekf::predict only
// (x <- x boxplus f(x, u) dt: pos += vel dt, rot = rot Exp((w - bg) dt), vel += (rot (a - ba) + grav) dt; use-ikfom.hpp get_f,
// esekfom.hpp:281-...).  imu: (t, acc, gyr) samples v_imu (the previous tail sample first); returns the end state in `st`.
struct ImuState { double pos[3], rot[9], vel[3], bg[3], ba[3], grav[3], offset_R_L_I[9], offset_T_L_I[3]; };
struct ImuMeas { double t, acc[3], gyr[3]; };
std::vector<Pose6D> ForwardPropagate(ImuState& st, const std::vector<ImuMeas>& v_imu, double pcl_beg_time, double pcl_end_time,
                                     double last_lidar_end_time, double acc_scale, const double acc_s_last[3], const double angvel_last[3]);

struct BoxPointType { float vertex_min[3], vertex_max[3]; };

// k-d tree with ikd-Tree's build rule, search procedure and lazy deletion (point_deleted flags; no re-balancing, so the shape
// differs from a tree the reference has rebuilt -- the search is exact, its result does not depend on the shape).
class KdTree {
public:
    void Build(PointVector pts);                                   // ikd_Tree.cpp:409-423, 690-744
    void Add_Point(const PointXYZINormal& p);                      // Add_by_point without downsampling, :1263-1312
    void Nearest_Search(const PointXYZINormal& q, int k, PointVector& near, std::vector<float>& sqdist) const;  // :426-461
    size_t size() const { return nodes.size(); }                   // nodes, deleted ones included
    const PointXYZINormal& point(size_t i) const { return nodes[i].p; }
    // ---- the incremental operations of the map thread (what keeps the CPU baseline's map a tree instead of a list) ----
    void Search_by_range(const BoxPointType& box, PointVector& storage) const;             // ikd_Tree.cpp:1259-1306, box = [min, max)
    int Delete_by_range(const BoxPointType& box);                                          // :779-860 (flags only)
    int Add_Points(const PointVector& to_add, bool downsample_on, float downsample_size);  // :478-584
    int Delete_Point_Boxes(const std::vector<BoxPointType>& boxes);                        // :643-668
    size_t valid_size() const { return nodes.size() - n_deleted; }
    PointVector valid_points() const;                                                      // insertion order

private:
    struct Node { PointXYZINormal p; int left = -1, right = -1, axis = 0; float lo[3], hi[3]; bool deleted = false; };
    size_t n_deleted = 0;
    void range_search(int n, const BoxPointType& box, PointVector* storage, int* n_del);
    std::vector<Node> nodes;
    int root = -1;
    int build(PointVector& s, int l, int r);
    void update(int n);
    struct HeapItem { PointXYZINormal p; float dist; };
    struct Heap;
    void search(int n, int k, const PointXYZINormal& q, Heap& h) const;
    float box_dist(int n, const PointXYZINormal& q) const;
};

// EstiPlane<float>: solve A n = -1 (5x3) by column-pivoted Householder QR, normalise, test |n.p + d| <= threshold
bool EstiPlane(float pca_result[4], const PointVector& pts, float threshold);

struct FeatureExtraction {
    PointVector feats_down_world;            // every down-sampled point in the world frame
    std::vector<PointVector> Nearest_Points;  // per point: neighbours in ascending distance
    std::vector<uint8_t> point_selected_surf;
    PointVector normvec;                      // per point: plane normal, intensity = pd2 (valid where selected)
    PointVector laserCloudOri, corr_normvect; // compacted selection (body-frame points / normals)
    int effct_feat_num = 0;
};
FeatureExtraction feature_extraction(const PointVector& feats_down_body, const LidarState& st, const KdTree& tree);

// ---- persistent map maintenance ---------------------------------------------------------
//   map_incremental            SF/include/lidar_front_end/LidarFrontEnd.cpp:387-435
//   lasermap_fov_segment       LidarFrontEnd.cpp:183-231
//   KD_TREE::Add_Points        SF/include/ikd-Tree/ikd_Tree.cpp:478-584 (down-sampling branch :493-530)
//   KD_TREE::Delete_Point_Boxes / Delete_by_range / Search_by_range   ikd_Tree.cpp:643, 776-860, 1259-1306 (box = [min, max))
// The map is kept as the multiset of its points: what a query returns does not depend on the shape of the tree (exact ties of
// the distance to a voxel centre between stored points, which the tree's traversal order would decide, are not modelled).
struct MapPoints {
    PointVector pts;
    int Add_Points(const PointVector& to_add, bool downsample_on, float downsample_size);   // returns tmp_counter
    int Delete_Point_Boxes(const std::vector<BoxPointType>& boxes);                        // returns the number of deleted points
};
struct MapIncrement { PointVector PointToAdd, PointNoNeedDownsample; };
// feats_down_body + the neighbours found by feature_extraction -> the two insertion lists (world frame at state st)
MapIncrement map_incremental_lists(const PointVector& feats_down_body, const LidarState& st, const std::vector<PointVector>& Nearest_Points,
                                   bool flg_EKF_inited, double filter_size_map_min);
struct LocalMapBox { BoxPointType box; bool initialized = false; };
// lasermap_fov_segment: moves the local-map cube when the sensor comes near its border; returns the boxes to delete
std::vector<BoxPointType> lasermap_fov_segment(LocalMapBox& lm, const double pos_lid[3], double cube_len, double det_range, float mov_threshold = 1.5f);

}  // namespace oracle
