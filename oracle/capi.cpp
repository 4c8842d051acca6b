// TEST INFRASTRUCTURE ONLY -- C entry points of the CPU oracle for ctypes (tests/, bench.py cpu_baseline leg,
// __graft_entry__.smoke()).  Nothing under tc2li-slam_amd/ may link or load this library.
#include <cstring>
#include <thread>

#include "orb.hpp"
#include "stereo.hpp"
#include "lidar.hpp"
#include "lidar_pose.hpp"
#include "eskf.hpp"
#include "localmap.hpp"
#include "mappoint.hpp"
#include "mapping.hpp"
#include "ba.hpp"
#include "imu.hpp"
#include "inertial_ba.hpp"
#include "matcher.hpp"

using namespace oracle;

extern "C" {

void* oracle_orb_create(int nfeatures, float scale, int nlevels, int ini_th, int min_th) {
    return new ORBextractor(nfeatures, scale, nlevels, ini_th, min_th);
}
void oracle_orb_destroy(void* h) { delete (ORBextractor*)h; }

// keypoints out as 6 floats each: x, y, size, angle, response, octave.  Returns N; *mono receives monoIndex.
int oracle_orb_extract(void* h, const uint8_t* img, int w, int hgt, int stride, int lap0, int lap1, float* kps,
                       uint8_t* desc, int cap, int* mono) {
    ORBextractor* e = (ORBextractor*)h;
    Img view = Img::view(img, w, hgt, stride);
    std::vector<KeyPoint> k;
    std::vector<uint8_t> d;
    const int lap[2] = {lap0, lap1};
    int m = e->extract(view, k, d, lap);
    if (mono) *mono = m;
    if (m < 0) return 0;
    if ((int)k.size() > cap) return -(int)k.size();
    for (size_t i = 0; i < k.size(); ++i) {
        float* o = kps + 6 * i;
        o[0] = k[i].x; o[1] = k[i].y; o[2] = k[i].size; o[3] = k[i].angle; o[4] = k[i].response; o[5] = (float)k[i].octave;
    }
    if (!d.empty()) std::memcpy(desc, d.data(), d.size());
    return (int)k.size();
}

// Left and right image on two threads, as Frame::Frame does (SF/src/Frame.cc:139-142).
int oracle_orb_extract_pair(void* hl, void* hr, const uint8_t* il, const uint8_t* ir, int w, int hgt, int stride,
                            float* kl, uint8_t* dl, float* kr, uint8_t* dr, int cap, int* nl, int* nr) {
    int ml = 0, mr = 0;
    std::thread tl([&] { *nl = oracle_orb_extract(hl, il, w, hgt, stride, 0, 0, kl, dl, cap, &ml); });
    std::thread tr([&] { *nr = oracle_orb_extract(hr, ir, w, hgt, stride, 0, 0, kr, dr, cap, &mr); });
    tl.join();
    tr.join();
    return 0;
}

int oracle_orb_level_size(void* h, int level, int* w, int* hgt) {
    ORBextractor* e = (ORBextractor*)h;
    if (level < 0 || level >= e->nlevels) return -1;
    *w = e->mvImagePyramid[level].w;
    *hgt = e->mvImagePyramid[level].h;
    return 0;
}

void oracle_orb_level_copy(void* h, int level, uint8_t* dst) {
    ORBextractor* e = (ORBextractor*)h;
    const Img& im = e->mvImagePyramid[level];
    for (int y = 0; y < im.h; ++y) std::memcpy(dst + (size_t)y * im.w, im.row(y), im.w);
}

void oracle_orb_blurred_copy(void* h, int level, uint8_t* dst) {
    ORBextractor* e = (ORBextractor*)h;
    Img work = e->mvImagePyramid[level].clone();
    Img out(work.w, work.h);
    gaussianBlur7(work, out);
    std::memcpy(dst, out.store.data(), (size_t)out.w * out.h);
}

int oracle_orb_candidates(void* h, int level, float* xyr, int cap) {
    ORBextractor* e = (ORBextractor*)h;
    const auto& c = e->mvCandidates[level];
    if (!xyr) return (int)c.size();
    if ((int)c.size() > cap) return -1;
    for (size_t i = 0; i < c.size(); ++i) {
        xyr[3 * i] = c[i].x + (EDGE_THRESHOLD - 3);
        xyr[3 * i + 1] = c[i].y + (EDGE_THRESHOLD - 3);
        xyr[3 * i + 2] = c[i].response;
    }
    return (int)c.size();
}

void oracle_orb_tables(void* h, float* scale, int* per_level, int* umax16) {
    ORBextractor* e = (ORBextractor*)h;
    for (int i = 0; i < e->nlevels; ++i) { scale[i] = e->mvScaleFactor[i]; per_level[i] = e->mnFeaturesPerLevel[i]; }
    for (int i = 0; i < 16; ++i) umax16[i] = e->umax[i];
}

// DistributeOctTree on raw candidates (x, y, response triples in the border-free frame); returns kept (x,y,response).
int oracle_orb_distribute(void* h, const float* xyr, int n, int minX, int maxX, int minY, int maxY, int N, float* out, int cap) {
    ORBextractor* e = (ORBextractor*)h;
    std::vector<KeyPoint> in(n);
    for (int i = 0; i < n; ++i) { in[i].x = xyr[3 * i]; in[i].y = xyr[3 * i + 1]; in[i].response = xyr[3 * i + 2]; }
    auto r = e->DistributeOctTree(in, minX, maxX, minY, maxY, N, 0);
    if ((int)r.size() > cap) return -1;
    for (size_t i = 0; i < r.size(); ++i) { out[3 * i] = r[i].x; out[3 * i + 1] = r[i].y; out[3 * i + 2] = r[i].response; }
    return (int)r.size();
}

static std::vector<KeyPoint> kps_from(const float* k, int n) {
    std::vector<KeyPoint> v(n);
    for (int i = 0; i < n; ++i) {
        v[i].x = k[6 * i]; v[i].y = k[6 * i + 1]; v[i].size = k[6 * i + 2]; v[i].angle = k[6 * i + 3];
        v[i].response = k[6 * i + 4]; v[i].octave = (int)k[6 * i + 5];
    }
    return v;
}

// Frame::ComputeStereoMatches on the pyramids the two oracle extractors hold from their last extract() call.
void oracle_stereo_match(void* hl, void* hr, const float* kl, const uint8_t* dl, int nl, const float* kr, const uint8_t* dr,
                         int nr, float mbf, float mb, float* u_right, float* depth, int* best_sad) {
    auto keysL = kps_from(kl, nl), keysR = kps_from(kr, nr);
    std::vector<uint8_t> descL(dl, dl + (size_t)nl * 32), descR(dr, dr + (size_t)nr * 32);
    StereoResult r = ComputeStereoMatches(*(ORBextractor*)hl, *(ORBextractor*)hr, keysL, descL, keysR, descR, mbf, mb);
    for (int i = 0; i < nl; ++i) { u_right[i] = r.uRight[i]; depth[i] = r.depth[i]; if (best_sad) best_sad[i] = r.bestDist[i]; }
}

int oracle_descriptor_distance(const uint8_t* a, const uint8_t* b) { return DescriptorDistance(a, b); }

// Feature grid: returns the flattened GetFeaturesInArea result for one query.
int oracle_features_in_area(const float* k, int n, int cols, int rows, float x, float y, float r, int minLevel, int maxLevel,
                            int* out, int cap) {
    auto keys = kps_from(k, n);
    FeatureGrid g;
    g.init(cols, rows);
    g.assign(keys);
    auto v = g.GetFeaturesInArea(keys, x, y, r, minLevel, maxLevel);
    if ((int)v.size() > cap) return -1;
    for (size_t i = 0; i < v.size(); ++i) out[i] = (int)v[i];
    return (int)v.size();
}

// ---- LiDAR -----------------------------------------------------------------------------------------------------
int oracle_lidar_preprocess(const VelodynePoint* raw, int n, int point_filter_num, double blind, float time_unit_scale,
                            PointXYZINormal* out, int cap) {
    PointVector v = preprocess_velodyne(raw, n, point_filter_num, blind, time_unit_scale);
    if ((int)v.size() > cap) return -1;
    if (!v.empty()) std::memcpy(out, v.data(), v.size() * sizeof(PointXYZINormal));
    return (int)v.size();
}

int oracle_lidar_voxel_grid(const PointXYZINormal* in, int n, float leaf, PointXYZINormal* out, int cap) {
    PointVector v = voxel_grid_filter(PointVector(in, in + n), leaf);
    if ((int)v.size() > cap) return -1;
    if (!v.empty()) std::memcpy(out, v.data(), v.size() * sizeof(PointXYZINormal));
    return (int)v.size();
}

void* oracle_kdtree_build(const PointXYZINormal* pts, int n) {
    KdTree* t = new KdTree();
    t->Build(PointVector(pts, pts + n));
    return t;
}
void oracle_kdtree_add(void* h, const PointXYZINormal* pts, int n) {
    for (int i = 0; i < n; ++i) ((KdTree*)h)->Add_Point(pts[i]);
}
void oracle_kdtree_destroy(void* h) { delete (KdTree*)h; }
// the incremental operations on the tree, and the same on the plain point list (MapPoints) that pins them in the tests
int oracle_kdtree_add_points(void* h, const PointXYZINormal* pts, int n, int downsample_on, float downsample_size) {
    return ((KdTree*)h)->Add_Points(PointVector(pts, pts + n), downsample_on != 0, downsample_size);
}
int oracle_kdtree_delete_boxes(void* h, const float* boxes6, int n_boxes) {
    std::vector<BoxPointType> boxes(n_boxes);
    for (int b = 0; b < n_boxes; ++b) { std::memcpy(boxes[b].vertex_min, boxes6 + 6 * b, 12); std::memcpy(boxes[b].vertex_max, boxes6 + 6 * b + 3, 12); }
    return ((KdTree*)h)->Delete_Point_Boxes(boxes);
}
int oracle_kdtree_valid_points(void* h, PointXYZINormal* out, int capacity) {
    const PointVector v = ((KdTree*)h)->valid_points();
    std::memcpy(out, v.data(), std::min((size_t)capacity, v.size()) * sizeof(PointXYZINormal));
    return (int)v.size();
}
int oracle_mappoints_add(const PointXYZINormal* map_pts, int n_map, const PointXYZINormal* pts, int n, int downsample_on, float downsample_size,
                         PointXYZINormal* out, int capacity) {
    MapPoints mp;
    mp.pts.assign(map_pts, map_pts + n_map);
    mp.Add_Points(PointVector(pts, pts + n), downsample_on != 0, downsample_size);
    std::memcpy(out, mp.pts.data(), std::min((size_t)capacity, mp.pts.size()) * sizeof(PointXYZINormal));
    return (int)mp.pts.size();
}
int oracle_kdtree_size(void* h) { return (int)((KdTree*)h)->size(); }

// k nearest of each query: near [nq][k] points, sqdist [nq][k], found [nq]
void oracle_kdtree_knn(void* h, const PointXYZINormal* q, int nq, int k, PointXYZINormal* near, float* sqdist, int* found) {
    const KdTree* t = (KdTree*)h;
    PointVector pn;
    std::vector<float> d;
    for (int i = 0; i < nq; ++i) {
        t->Nearest_Search(q[i], k, pn, d);
        found[i] = (int)pn.size();
        for (size_t j = 0; j < pn.size(); ++j) { near[(size_t)i * k + j] = pn[j]; sqdist[(size_t)i * k + j] = d[j]; }
    }
}

int oracle_esti_plane(const PointXYZINormal* five, float threshold, float* pabcd) {
    return EstiPlane(pabcd, PointVector(five, five + 5), threshold) ? 1 : 0;
}

// feature_extraction(): state = rot[9], pos[3], offset_R[9], offset_T[3] (row-major doubles).  Outputs per input point:
// world point, selected flag, normal+pd2 (normvec), 5 neighbours; plus the compacted laserCloudOri / corr_normvect.
int oracle_lidar_feature_extraction(void* tree, const PointXYZINormal* body, int n, const double* state24,
                                    PointXYZINormal* world, uint8_t* selected, PointXYZINormal* normvec,
                                    PointXYZINormal* nearest5, int* nfound, PointXYZINormal* cloud_ori,
                                    PointXYZINormal* corr_norm) {
    LidarState st;
    std::memcpy(st.rot, state24, 9 * sizeof(double));
    std::memcpy(st.pos, state24 + 9, 3 * sizeof(double));
    std::memcpy(st.offset_R_L_I, state24 + 12, 9 * sizeof(double));
    std::memcpy(st.offset_T_L_I, state24 + 21, 3 * sizeof(double));
    FeatureExtraction fe = feature_extraction(PointVector(body, body + n), st, *(KdTree*)tree);
    for (int i = 0; i < n; ++i) {
        if (world) world[i] = fe.feats_down_world[i];
        if (selected) selected[i] = fe.point_selected_surf[i];
        if (normvec) normvec[i] = fe.normvec[i];
        if (nfound) nfound[i] = (int)fe.Nearest_Points[i].size();
        if (nearest5)
            for (size_t j = 0; j < fe.Nearest_Points[i].size() && j < 5; ++j) nearest5[(size_t)i * 5 + j] = fe.Nearest_Points[i][j];
    }
    for (int i = 0; i < fe.effct_feat_num; ++i) {
        if (cloud_ori) cloud_ori[i] = fe.laserCloudOri[i];
        if (corr_norm) corr_norm[i] = fe.corr_normvect[i];
    }
    return fe.effct_feat_num;
}

// One frame of the front end with the reference's threading: left/right ORB on two threads (SF/src/Frame.cc:139-142)
// followed by stereo matching on the tracking thread, while the LiDAR thread (src/examples/camera_lidar.cc:84) runs
// preprocess -> voxel filter -> feature_extraction.  Returns the number of stereo matches; *n_sel gets effct_feat_num.
int oracle_frontend_frame(void* hl, void* hr, const uint8_t* il, const uint8_t* ir, int w, int hgt, float mbf, float mb,
                          const VelodynePoint* raw, int n_raw, void* tree, const double* state24, int* n_sel) {
    int sel = 0;
    std::thread lidar([&] {
        PointVector pre = preprocess_velodyne(raw, n_raw, 2, 2.0, 1e-3f);
        PointVector down = voxel_grid_filter(pre, 0.5f);
        LidarState st;
        std::memcpy(st.rot, state24, 9 * sizeof(double));
        std::memcpy(st.pos, state24 + 9, 3 * sizeof(double));
        std::memcpy(st.offset_R_L_I, state24 + 12, 9 * sizeof(double));
        std::memcpy(st.offset_T_L_I, state24 + 21, 3 * sizeof(double));
        sel = feature_extraction(down, st, *(KdTree*)tree).effct_feat_num;
    });
    ORBextractor* el = (ORBextractor*)hl;
    ORBextractor* er = (ORBextractor*)hr;
    std::vector<KeyPoint> kl, kr;
    std::vector<uint8_t> dl, dr;
    const int lap[2] = {0, 0};
    std::thread tl([&] { el->extract(Img::view(il, w, hgt, w), kl, dl, lap); });
    std::thread tr([&] { er->extract(Img::view(ir, w, hgt, w), kr, dr, lap); });
    tl.join();
    tr.join();
    StereoResult r = ComputeStereoMatches(*el, *er, kl, dl, kr, dr, mbf, mb);
    int matches = 0;
    for (float d : r.depth) matches += d > 0;
    lidar.join();
    if (n_sel) *n_sel = sel;
    return matches;
}

// ---- optimisation back end ---------------------------------------------------------------------------------------
// poses as 7 doubles (qx, qy, qz, qw, tx, ty, tz); edges as 6 doubles (point, pose, u, v, uR, invSigma2); cam = fx fy cx cy bf
static std::vector<BAEdge> edges_from(const double* e, int n) {
    std::vector<BAEdge> v(n);
    for (int i = 0; i < n; ++i) {
        v[i].point = (int)e[6 * i]; v[i].pose = (int)e[6 * i + 1];
        v[i].obs[0] = e[6 * i + 2]; v[i].obs[1] = e[6 * i + 3]; v[i].obs[2] = e[6 * i + 4]; v[i].info = e[6 * i + 5];
    }
    return v;
}
static SE3Quat pose_from(const double* p) { SE3Quat T; std::memcpy(T.q, p, 4 * sizeof(double)); std::memcpy(T.t, p + 4, 3 * sizeof(double)); return T; }
static void pose_to(const SE3Quat& T, double* p) { std::memcpy(p, T.q, 4 * sizeof(double)); std::memcpy(p + 4, T.t, 3 * sizeof(double)); }

int oracle_pose_optimization(double* pose7, const double* Xw, const double* edges6, int n, const double* cam5, uint8_t* outlier,
                             double* trace_chi2, double* trace_lambda, int* trace_trials, int trace_cap, int* trace_n) {
    SE3Quat T = pose_from(pose7);
    Camera cam{cam5[0], cam5[1], cam5[2], cam5[3], cam5[4]};
    std::vector<uint8_t> out;
    LMTrace tr;
    int inl = PoseOptimization(T, std::vector<double>(Xw, Xw + 3 * (size_t)n), edges_from(edges6, n), cam, out, &tr);
    pose_to(T, pose7);
    for (int i = 0; i < n && i < (int)out.size(); ++i) outlier[i] = out[i];
    const int m = std::min((int)tr.chi2.size(), trace_cap);
    for (int i = 0; i < m; ++i) { if (trace_chi2) trace_chi2[i] = tr.chi2[i]; if (trace_lambda) trace_lambda[i] = tr.lambda[i]; if (trace_trials) trace_trials[i] = tr.trials[i]; }
    if (trace_n) *trace_n = (int)tr.chi2.size();
    return inl;
}

int oracle_local_ba(double* poses7, const uint8_t* fixed, int n_poses, double* points3, int n_points, const double* edges6, int n_edges,
                    const double* cam5, int iterations, double lambda_init, double* chi2_out, uint8_t* depth_pos,
                    double* trace_chi2, double* trace_lambda, int* trace_trials, int trace_cap) {
    std::vector<SE3Quat> poses(n_poses);
    for (int i = 0; i < n_poses; ++i) poses[i] = pose_from(poses7 + 7 * i);
    std::vector<double> pts(points3, points3 + 3 * (size_t)n_points);
    Camera cam{cam5[0], cam5[1], cam5[2], cam5[3], cam5[4]};
    BAResult r = LocalBundleAdjustment(poses, std::vector<uint8_t>(fixed, fixed + n_poses), pts, edges_from(edges6, n_edges), cam,
                                       iterations, lambda_init, nullptr);
    for (int i = 0; i < n_poses; ++i) pose_to(poses[i], poses7 + 7 * i);
    std::memcpy(points3, pts.data(), pts.size() * sizeof(double));
    for (int e = 0; e < n_edges; ++e) { if (chi2_out) chi2_out[e] = r.chi2[e]; if (depth_pos) depth_pos[e] = r.depth_pos[e]; }
    const int m = std::min((int)r.trace.chi2.size(), trace_cap);
    for (int i = 0; i < m; ++i) { if (trace_chi2) trace_chi2[i] = r.trace.chi2[i]; if (trace_lambda) trace_lambda[i] = r.trace.lambda[i]; if (trace_trials) trace_trials[i] = r.trace.trials[i]; }
    return r.iterations;
}

// Local BA with the BALM edge: window = the first n_win entries of win_pose (pose indices, window order), clouds packed
// back to back (cloud_off[n_win + 1], xyz floats in the LiDAR frame), Tcl as 7 floats, information = wLBA.
int oracle_local_ba_lidar(double* poses7, const uint8_t* fixed, int n_poses, double* points3, int n_points, const double* edges6,
                          int n_edges, const double* cam5, int iterations, double lambda_init, const int* win_pose, int n_win,
                          const float* clouds, const int* cloud_off, const float* Tcl7, double wLBA, double* chi2_out,
                          uint8_t* depth_pos, double* trace_chi2, double* trace_lambda, int* trace_trials, int trace_cap,
                          int* n_planes, double* lidar_out /* [2 + 6W + 36W^2]: residual, chi2, JacT, Hessian at the end */) {
    std::vector<SE3Quat> poses(n_poses);
    for (int i = 0; i < n_poses; ++i) poses[i] = pose_from(poses7 + 7 * i);
    std::vector<double> pts(points3, points3 + 3 * (size_t)n_points);
    Camera cam{cam5[0], cam5[1], cam5[2], cam5[3], cam5[4]};
    SE3fQ Tcl;
    std::memcpy(Tcl.q, Tcl7, 16); std::memcpy(Tcl.t, Tcl7 + 4, 12);
    LidarCovisRes lio(Tcl);
    lio.win_size_ = n_win;
    std::vector<int> lp(win_pose, win_pose + n_win);
    for (int i = 0; i < n_win; ++i) {
        SE3fQ Tcw;  // KeyFrame::GetPose() is a Sophus::SE3f
        for (int k = 0; k < 4; ++k) Tcw.q[k] = (float)poses[lp[i]].q[k];
        for (int k = 0; k < 3; ++k) Tcw.t[k] = (float)poses[lp[i]].t[k];
        lio.AddFromKeyFrame(Tcw, std::vector<float>(clouds + 3 * (size_t)cloud_off[i], clouds + 3 * (size_t)cloud_off[i + 1]));
    }
    lio.BuildVoxHess();
    if (n_planes) *n_planes = (int)lio.planes().size();
    EdgeLidar edge;
    edge.lio = &lio;
    edge.information = wLBA;
    BAResult r = LocalBundleAdjustment(poses, std::vector<uint8_t>(fixed, fixed + n_poses), pts, edges_from(edges6, n_edges), cam,
                                       iterations, lambda_init, nullptr, &edge, &lp);
    for (int i = 0; i < n_poses; ++i) pose_to(poses[i], poses7 + 7 * i);
    std::memcpy(points3, pts.data(), pts.size() * sizeof(double));
    for (int e = 0; e < n_edges; ++e) { if (chi2_out) chi2_out[e] = r.chi2[e]; if (depth_pos) depth_pos[e] = r.depth_pos[e]; }
    const int m = std::min((int)r.trace.chi2.size(), trace_cap);
    for (int i = 0; i < m; ++i) { if (trace_chi2) trace_chi2[i] = r.trace.chi2[i]; if (trace_lambda) trace_lambda[i] = r.trace.lambda[i]; if (trace_trials) trace_trials[i] = r.trace.trials[i]; }
    if (lidar_out) {
        lidar_out[0] = edge.error; lidar_out[1] = edge.chi2();
        for (size_t k = 0; k < edge.JacT.size(); ++k) lidar_out[2 + k] = edge.JacT[k];
        for (size_t k = 0; k < edge.Hessian.size(); ++k) lidar_out[2 + edge.JacT.size() + k] = edge.Hessian[k];
    }
    return r.iterations;
}

// The LiDAR edge alone: planes from the window at poses7, then ComputeError and ComputeJandHSE3 at the same poses
// (the quantities tc2li_lidar_window_evaluate returns).
int oracle_lidar_window_evaluate(const double* poses7, int n_poses, const int* win_pose, int n_win, const float* clouds,
                                 const int* cloud_off, const float* Tcl7, double* residual, double* JacT, double* Hess) {
    (void)n_poses;
    SE3fQ Tcl;
    std::memcpy(Tcl.q, Tcl7, 16); std::memcpy(Tcl.t, Tcl7 + 4, 12);
    LidarCovisRes lio(Tcl);
    lio.win_size_ = n_win;
    std::vector<double> R(9 * n_win), t(3 * n_win);
    for (int i = 0; i < n_win; ++i) {
        const SE3Quat T = pose_from(poses7 + 7 * win_pose[i]);
        SE3fQ Tcw;
        for (int k = 0; k < 4; ++k) Tcw.q[k] = (float)T.q[k];
        for (int k = 0; k < 3; ++k) Tcw.t[k] = (float)T.t[k];
        lio.AddFromKeyFrame(Tcw, std::vector<float>(clouds + 3 * (size_t)cloud_off[i], clouds + 3 * (size_t)cloud_off[i + 1]));
        double Rt[9];
        oracle::quat_to_matrix_public(T.q, Rt);
        std::memcpy(&R[9 * i], Rt, sizeof(Rt));
        std::memcpy(&t[3 * i], T.t, 3 * sizeof(double));
    }
    lio.BuildVoxHess();
    EdgeLidar edge;
    edge.lio = &lio;
    edge.computeError(R.data(), t.data(), n_win);
    if (residual) *residual = edge.error;
    if (JacT && Hess) {
        edge.is_calc_hess = true;
        edge.linearizeOplus(R.data(), t.data(), n_win);
        std::memcpy(JacT, edge.JacT.data(), edge.JacT.size() * sizeof(double));
        std::memcpy(Hess, edge.Hessian.data(), edge.Hessian.size() * sizeof(double));
    }
    return (int)lio.planes().size();
}

// The planes of the window (VOX_HESS after BuildVoxHess): per plane and keyframe P (9), v (3), N -> 13 doubles; coe per plane.
int oracle_lidar_planes(const double* poses7, const int* win_pose, int n_win, const float* clouds, const int* cloud_off,
                        const float* Tcl7, double* clusters13, double* coe, int capacity) {
    SE3fQ Tcl;
    std::memcpy(Tcl.q, Tcl7, 16); std::memcpy(Tcl.t, Tcl7 + 4, 12);
    LidarCovisRes lio(Tcl);
    lio.win_size_ = n_win;
    for (int i = 0; i < n_win; ++i) {
        const SE3Quat T = pose_from(poses7 + 7 * win_pose[i]);
        SE3fQ Tcw;
        for (int k = 0; k < 4; ++k) Tcw.q[k] = (float)T.q[k];
        for (int k = 0; k < 3; ++k) Tcw.t[k] = (float)T.t[k];
        lio.AddFromKeyFrame(Tcw, std::vector<float>(clouds + 3 * (size_t)cloud_off[i], clouds + 3 * (size_t)cloud_off[i + 1]));
    }
    lio.BuildVoxHess();
    const int n = (int)lio.planes().size();
    for (int a = 0; a < std::min(n, capacity); ++a) {
        const PlaneVoxel& pv = lio.planes()[a];
        coe[a] = pv.coe;
        for (int i = 0; i < n_win; ++i) {
            double* o = clusters13 + 13 * ((size_t)a * n_win + i);
            std::memcpy(o, pv.sig_orig[i].P.m, 9 * sizeof(double));
            o[9] = pv.sig_orig[i].v.x; o[10] = pv.sig_orig[i].v.y; o[11] = pv.sig_orig[i].v.z; o[12] = pv.sig_orig[i].N;
        }
    }
    return n;
}

// BALM residual and raw (LiDAR-pose) Jacobian / Hessian of BALM2::divide_thread for given window poses Twl (R 9 + p 3 doubles each)
// on planes built from `n_win` clouds at those poses: used to check the analytic derivatives against finite differences.
int oracle_balm_evaluate(const double* Twl12, int n_win, const float* clouds, const int* cloud_off, double* residual, double* JacT,
                         double* Hess, const double* eval_Twl12, int n_eval, double* eval_residual) {
    // poses are given directly as Twl; build with Tcl = identity and Tcw = Twl^-1 (float round trip as in the reference)
    SE3fQ Tcl;
    LidarCovisRes lio(Tcl);
    lio.win_size_ = n_win;
    std::vector<double> Rcw(9 * n_win), tcw(3 * n_win);
    for (int i = 0; i < n_win; ++i) {
        const double* R = Twl12 + 12 * i;
        const double* p = R + 9;
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Rcw[9 * i + 3 * r + c] = R[3 * c + r];
        for (int r = 0; r < 3; ++r) tcw[3 * i + r] = -(Rcw[9 * i + 3 * r] * p[0] + Rcw[9 * i + 3 * r + 1] * p[1] + Rcw[9 * i + 3 * r + 2] * p[2]);
        SE3Quat q;
        // Tcw as SE3f
        SE3fQ Tcw;
        float Rf[9];
        for (int k = 0; k < 9; ++k) Rf[k] = (float)Rcw[9 * i + k];
        // quaternion via the double helper of ba.cpp is not visible here: use UpdatePose after AddFromKeyFrame instead
        (void)q; (void)Rf;
        double tr = Rcw[9 * i] + Rcw[9 * i + 4] + Rcw[9 * i + 8];
        double qw = std::sqrt(std::max(0.0, 1 + tr)) / 2;
        Tcw.q[3] = (float)qw;
        Tcw.q[0] = (float)((Rcw[9 * i + 7] - Rcw[9 * i + 5]) / (4 * qw));
        Tcw.q[1] = (float)((Rcw[9 * i + 2] - Rcw[9 * i + 6]) / (4 * qw));
        Tcw.q[2] = (float)((Rcw[9 * i + 3] - Rcw[9 * i + 1]) / (4 * qw));
        for (int k = 0; k < 3; ++k) Tcw.t[k] = (float)tcw[3 * i + k];
        lio.AddFromKeyFrame(Tcw, std::vector<float>(clouds + 3 * (size_t)cloud_off[i], clouds + 3 * (size_t)cloud_off[i + 1]));
    }
    lio.BuildVoxHess();
    for (int i = 0; i < n_win; ++i) lio.UpdatePose(i, &Rcw[9 * i], &tcw[3 * i]);
    if (residual) *residual = lio.ComputeError();
    if (JacT && Hess) {
        std::vector<double> H, J;
        lio.divide_thread(H, J);
        std::memcpy(JacT, J.data(), J.size() * sizeof(double));
        std::memcpy(Hess, H.data(), H.size() * sizeof(double));
    }
    for (int e = 0; e < n_eval; ++e) {  // residual of the same planes at other window poses
        for (int i = 0; i < n_win; ++i) {
            const double* R = eval_Twl12 + 12 * ((size_t)e * n_win + i);
            const double* p = R + 9;
            double Rc[9], tc[3];
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Rc[3 * r + c] = R[3 * c + r];
            for (int r = 0; r < 3; ++r) tc[r] = -(Rc[3 * r] * p[0] + Rc[3 * r + 1] * p[1] + Rc[3 * r + 2] * p[2]);
            lio.UpdatePose(i, Rc, tc);
        }
        eval_residual[e] = lio.ComputeError();
    }
    return (int)lio.planes().size();
}

// error + analytic Jacobians of one binary projection edge (dim returned)
int oracle_edge_linearize(const double* pose7, const double* X, const double* edge6, const double* cam5, double* err, double* A, double* B) {
    Camera cam{cam5[0], cam5[1], cam5[2], cam5[3], cam5[4]};
    return edge_linearize(pose_from(pose7), X, edges_from(edge6, 1)[0], cam, err, A, B);
}
void oracle_se3_exp_mul(const double* update6, const double* pose7, double* out7) { pose_to(se3_mul(se3_exp(update6), pose_from(pose7)), out7); }

// ---- projection matching -------------------------------------------------------------------------------------------
struct QueryPOD {  // same 64-byte layout as tc2li_proj_query
    float u, v, radius, u_right;
    int32_t min_level, max_level;
    float angle;
    int16_t valid, has_observations;
    uint8_t desc[32];
};
static_assert(sizeof(QueryPOD) == 64, "layout");
static std::vector<ProjQuery> queries_from(const QueryPOD* q, int n) {
    std::vector<ProjQuery> v(n);
    for (int i = 0; i < n; ++i) {
        v[i].u = q[i].u; v[i].v = q[i].v; v[i].radius = q[i].radius; v[i].u_right = q[i].u_right; v[i].min_level = q[i].min_level;
        v[i].max_level = q[i].max_level; v[i].angle = q[i].angle; v[i].valid = q[i].valid; v[i].has_observations = q[i].has_observations;
        std::memcpy(v[i].desc, q[i].desc, 32);
    }
    return v;
}
static void queries_to(const std::vector<ProjQuery>& v, QueryPOD* q) {
    for (size_t i = 0; i < v.size(); ++i) {
        std::memset(&q[i], 0, sizeof(QueryPOD));
        q[i].u = v[i].u; q[i].v = v[i].v; q[i].radius = v[i].radius; q[i].u_right = v[i].u_right; q[i].min_level = v[i].min_level;
        q[i].max_level = v[i].max_level; q[i].angle = v[i].angle; q[i].valid = (int16_t)v[i].valid; q[i].has_observations = (int16_t)v[i].has_observations;
        std::memcpy(q[i].desc, v[i].desc, 32);
    }
}

// frame: keys as 6 floats each, desc, uRight, occupied; returns nmatches after the optional rotation filter
int oracle_search_by_projection(const float* keys6, const uint8_t* desc, const float* uright, const uint8_t* occupied, int n, int cols,
                                int rows, const QueryPOD* queries, int m, int mode, float nnratio, int check_orientation, int* match_of_query) {
    FrameView F;
    F.keys = kps_from(keys6, n);
    F.desc.assign(desc, desc + (size_t)n * 32);
    F.uRight.assign(uright, uright + n);
    if (occupied) F.occupied.assign(occupied, occupied + n); else F.occupied.assign(n, 0);
    F.cols = cols; F.rows = rows;
    std::vector<ProjQuery> qs = queries_from(queries, m);
    std::vector<int> match;
    int nm = match_queries(F, qs, mode == 0 ? MATCH_BEST : MATCH_RATIO, nnratio, match);
    if (check_orientation) nm -= rotation_filter(F, qs, match);
    for (int q = 0; q < m; ++q) match_of_query[q] = match[q];
    return nm;
}

void oracle_project_last_frame(const float* pose_cur7, const float* pose_last7, const float* cam4, float mb, float mbf, const float* scales,
                               int nlevels, int cols, int rows, int n, const uint8_t* has_point, const uint8_t* outlier, const float* Xw,
                               const float* last_keys6, const uint8_t* mp_desc, float th, int mono, QueryPOD* out) {
    SE3f Tcw, Tlw;
    std::memcpy(Tcw.q, pose_cur7, 16); std::memcpy(Tcw.t, pose_cur7 + 4, 12);
    std::memcpy(Tlw.q, pose_last7, 16); std::memcpy(Tlw.t, pose_last7 + 4, 12);
    CamF cam{cam4[0], cam4[1], cam4[2], cam4[3]};
    auto qs = build_queries_last_frame(Tcw, Tlw, cam, mb, mbf, std::vector<float>(scales, scales + nlevels), cols, rows,
                                       std::vector<uint8_t>(has_point, has_point + n), std::vector<uint8_t>(outlier, outlier + n),
                                       std::vector<float>(Xw, Xw + 3 * (size_t)n), kps_from(last_keys6, n),
                                       std::vector<uint8_t>(mp_desc, mp_desc + (size_t)n * 32), th, mono != 0);
    queries_to(qs, out);
}

// Tracking::TrackWithMotionModel (SF/src/Tracking.cc:2737-2834), data path of one frame: SearchByProjection(th) with
// ORBmatcher(0.9, true), the 2*th retry below 20 matches, PoseOptimization, outliers discarded.  Returns the value of
// PoseOptimization, -1 when the search found fewer than 20 matches.
static int track_core(const FrameView& F, const float* scales, const float* inv_sigma2, int nlevels, const float* pose_pred7,
                      const float* pose_last7, const double* cam5, float mb, float th, int n_last, const uint8_t* has_point,
                      const uint8_t* outlier_last, const float* Xw, const std::vector<KeyPoint>& lk, const uint8_t* mp_desc,
                      double* pose_out7, int* map_point_of_keypoint, int* n_matches) {
    const int n = (int)F.keys.size(), cols = F.cols, rows = F.rows;
    SE3f Tcw, Tlw;
    std::memcpy(Tcw.q, pose_pred7, 16); std::memcpy(Tcw.t, pose_pred7 + 4, 12);
    std::memcpy(Tlw.q, pose_last7, 16); std::memcpy(Tlw.t, pose_last7 + 4, 12);
    CamF camf{(float)cam5[0], (float)cam5[1], (float)cam5[2], (float)cam5[3]};
    const std::vector<float> sc(scales, scales + nlevels);
    const std::vector<uint8_t> hp(has_point, has_point + n_last), ol(outlier_last, outlier_last + n_last);
    const std::vector<float> X(Xw, Xw + 3 * (size_t)n_last);
    const std::vector<uint8_t> md(mp_desc, mp_desc + (size_t)n_last * 32);
    std::vector<int> match;
    int nm = 0;
    for (int attempt = 0; attempt < 2; ++attempt) {
        auto qs = build_queries_last_frame(Tcw, Tlw, camf, mb, (float)cam5[4], sc, cols, rows, hp, ol, X, lk, md, attempt == 0 ? th : 2 * th, false);
        nm = match_queries(F, qs, MATCH_BEST, 0.9f, match);
        nm -= rotation_filter(F, qs, match);
        if (nm >= 20) break;
    }
    *n_matches = nm;
    for (int i = 0; i < n; ++i) map_point_of_keypoint[i] = -1;
    for (int q = 0; q < n_last; ++q) if (match[q] >= 0) map_point_of_keypoint[match[q]] = q;
    for (int c = 0; c < 7; ++c) pose_out7[c] = (double)pose_pred7[c];
    if (nm < 20) return -1;
    std::vector<double> Xd;
    std::vector<BAEdge> edges;
    std::vector<int> kp_of_edge;
    for (int i = 0; i < n; ++i) {
        const int q = map_point_of_keypoint[i];
        if (q < 0) continue;
        BAEdge e;
        e.point = (int)edges.size(); e.pose = 0;
        e.obs[0] = F.keys[i].x; e.obs[1] = F.keys[i].y; e.obs[2] = F.uRight[i];
        e.info = inv_sigma2[F.keys[i].octave];
        edges.push_back(e);
        for (int c = 0; c < 3; ++c) Xd.push_back((double)Xw[3 * (size_t)q + c]);
        kp_of_edge.push_back(i);
    }
    SE3Quat T = pose_from(pose_out7);
    Camera cam{cam5[0], cam5[1], cam5[2], cam5[3], cam5[4]};
    std::vector<uint8_t> out;
    const int inl = PoseOptimization(T, Xd, edges, cam, out);
    pose_to(T, pose_out7);
    for (size_t e = 0; e < edges.size() && e < out.size(); ++e) if (out[e]) map_point_of_keypoint[kp_of_edge[e]] = -1;
    return inl;
}

int oracle_track_motion_model(const float* keys6, const uint8_t* desc, const float* uright, int n, int cols, int rows, const float* scales,
                              const float* inv_sigma2, int nlevels, const float* pose_pred7, const float* pose_last7, const double* cam5,
                              float mb, float th, int n_last, const uint8_t* has_point, const uint8_t* outlier_last, const float* Xw,
                              const float* last_keys6, const uint8_t* mp_desc, double* pose_out7, int* map_point_of_keypoint,
                              int* n_matches) {
    FrameView F;
    F.keys = kps_from(keys6, n);
    F.desc.assign(desc, desc + (size_t)n * 32);
    F.uRight.assign(uright, uright + n);
    F.occupied.assign(n, 0);
    F.cols = cols; F.rows = rows;
    return track_core(F, scales, inv_sigma2, nlevels, pose_pred7, pose_last7, cam5, mb, th, n_last, has_point, outlier_last, Xw,
                      kps_from(last_keys6, n_last), mp_desc, pose_out7, map_point_of_keypoint, n_matches);
}

// One frame of the tracking loop with the reference's threading: oracle_frontend_frame (left / right ORB on two threads,
// stereo matching, the LiDAR front end on its own thread) followed on the tracking thread by TrackWithMotionModel's data
// path against the given last frame.  Returns PoseOptimization's inlier count (-1: lost).
int oracle_loop_frame(void* hl, void* hr, const uint8_t* il, const uint8_t* ir, int w, int hgt, float mbf, float mb,
                      const VelodynePoint* raw, int n_raw, void* tree, const double* state24, const float* pose_pred7,
                      const float* pose_last7, const double* cam5, float th, int n_last, const uint8_t* has_point,
                      const uint8_t* outlier_last, const float* Xw, const float* last_keys6, const uint8_t* mp_desc, double* pose_out7,
                      int* n_sel, int* n_matches) {
    int sel = 0;
    std::thread lidar([&] {
        PointVector pre = preprocess_velodyne(raw, n_raw, 2, 2.0, 1e-3f);
        PointVector down = voxel_grid_filter(pre, 0.5f);
        LidarState st;
        std::memcpy(st.rot, state24, 9 * sizeof(double));
        std::memcpy(st.pos, state24 + 9, 3 * sizeof(double));
        std::memcpy(st.offset_R_L_I, state24 + 12, 9 * sizeof(double));
        std::memcpy(st.offset_T_L_I, state24 + 21, 3 * sizeof(double));
        sel = feature_extraction(down, st, *(KdTree*)tree).effct_feat_num;
    });
    ORBextractor* el = (ORBextractor*)hl;
    ORBextractor* er = (ORBextractor*)hr;
    std::vector<KeyPoint> kl, kr;
    std::vector<uint8_t> dl, dr;
    const int lap[2] = {0, 0};
    std::thread tl([&] { el->extract(Img::view(il, w, hgt, w), kl, dl, lap); });
    std::thread tr([&] { er->extract(Img::view(ir, w, hgt, w), kr, dr, lap); });
    tl.join();
    tr.join();
    StereoResult r = ComputeStereoMatches(*el, *er, kl, dl, kr, dr, mbf, mb);
    FrameView F;
    F.keys = kl; F.desc = dl; F.uRight = r.uRight; F.occupied.assign(kl.size(), 0); F.cols = w; F.rows = hgt;
    std::vector<int> mp(kl.size() + 1);
    const int inl = track_core(F, el->mvScaleFactor.data(), el->mvInvLevelSigma2.data(), (int)el->mvScaleFactor.size(), pose_pred7,
                               pose_last7, cam5, mb, th, n_last, has_point, outlier_last, Xw, kps_from(last_keys6, n_last), mp_desc,
                               pose_out7, mp.data(), n_matches);
    lidar.join();
    if (n_sel) *n_sel = sel;
    return inl;
}

struct MapPointPOD { float pos[3], normal[3], min_distance, max_distance, max_distance_raw; uint8_t desc[32]; };  // tc2li_map_point
void oracle_project_local_map(const float* pose7, const float* cam4, float mbf, const float* scales, int nlevels, float log_scale, int cols,
                              int rows, int n, const MapPointPOD* pts, float th, int far_points, float th_far, float cos_limit, QueryPOD* out) {
    SE3f Tcw;
    std::memcpy(Tcw.q, pose7, 16); std::memcpy(Tcw.t, pose7 + 4, 12);
    CamF cam{cam4[0], cam4[1], cam4[2], cam4[3]};
    std::vector<MapPointView> mps(n);
    for (int i = 0; i < n; ++i) {
        std::memcpy(mps[i].pos, pts[i].pos, 12); std::memcpy(mps[i].normal, pts[i].normal, 12);
        mps[i].min_dist = pts[i].min_distance; mps[i].max_dist = pts[i].max_distance; mps[i].mfMaxDistance = pts[i].max_distance_raw;
        std::memcpy(mps[i].desc, pts[i].desc, 32);
    }
    auto qs = build_queries_local_map(Tcw, cam, mbf, std::vector<float>(scales, scales + nlevels), log_scale, cols, rows, mps, th,
                                      far_points != 0, th_far, cos_limit);
    queries_to(qs, out);
}

// Tracking::TrackLocalMap's data path for one frame (SearchLocalPoints -> PoseOptimization -> mnMatchesInliers); returns mnMatchesInliers
}  // extern "C"
static int track_local_core(FrameView& F, const float* scales, const float* inv_sigma2, int nlevels, float log_scale, const float* pose7,
                            const double* cam5, const uint8_t* held, const float* held_Xw, const MapPointPOD* pts, int m, float th, int far_points,
                            float th_far, double* pose_out7, int* local_of_keypoint, uint8_t* outlier, int* n_matches, bool optimise = true) {
    const int n = (int)F.keys.size(), cols = F.cols, rows = F.rows;
    F.occupied.resize(n);
    for (int i = 0; i < n; ++i) F.occupied[i] = held[i] == 1;
    SE3f Tcw;
    std::memcpy(Tcw.q, pose7, 16); std::memcpy(Tcw.t, pose7 + 4, 12);
    CamF camf{(float)cam5[0], (float)cam5[1], (float)cam5[2], (float)cam5[3]};
    std::vector<MapPointView> mps(m);
    for (int i = 0; i < m; ++i) {
        std::memcpy(mps[i].pos, pts[i].pos, 12); std::memcpy(mps[i].normal, pts[i].normal, 12);
        mps[i].min_dist = pts[i].min_distance; mps[i].max_dist = pts[i].max_distance; mps[i].mfMaxDistance = pts[i].max_distance_raw;
        std::memcpy(mps[i].desc, pts[i].desc, 32);
    }
    std::vector<int> match(m, -1);
    int nm = 0;
    if (m > 0) {
        auto qs = build_queries_local_map(Tcw, camf, (float)cam5[4], std::vector<float>(scales, scales + nlevels), log_scale, cols, rows, mps, th,
                                          far_points != 0, th_far, 0.5f);
        nm = match_queries(F, qs, MATCH_RATIO, 0.8f, match);
    }
    *n_matches = nm;
    for (int i = 0; i < n; ++i) { local_of_keypoint[i] = -1; outlier[i] = 0; }
    for (int q = 0; q < m; ++q) if (match[q] >= 0) local_of_keypoint[match[q]] = q;
    for (int c = 0; c < 7; ++c) pose_out7[c] = (double)pose7[c];
    // SearchLocalPoints only (Tracking.cc:3232-3294): with the IMU initialised TrackLocalMap hands the correspondences to
    // PoseInertialOptimizationLastFrame / LastKeyFrame instead of PoseOptimization (Tracking.cc:2857-2878)
    if (!optimise) return nm;
    std::vector<double> Xd;
    std::vector<BAEdge> edges;
    std::vector<int> kp_of_edge;
    for (int i = 0; i < n; ++i) {
        if (!(held[i] != 0 || local_of_keypoint[i] >= 0)) continue;
        BAEdge e;
        e.point = (int)edges.size(); e.pose = 0;
        e.obs[0] = F.keys[i].x; e.obs[1] = F.keys[i].y; e.obs[2] = F.uRight[i];
        e.info = inv_sigma2[F.keys[i].octave];
        edges.push_back(e);
        const float* X = local_of_keypoint[i] >= 0 ? pts[local_of_keypoint[i]].pos : held_Xw + 3 * (size_t)i;
        for (int c = 0; c < 3; ++c) Xd.push_back((double)X[c]);
        kp_of_edge.push_back(i);
    }
    for (int c = 0; c < 7; ++c) pose_out7[c] = (double)pose7[c];
    SE3Quat T = pose_from(pose_out7);
    Camera cam{cam5[0], cam5[1], cam5[2], cam5[3], cam5[4]};
    std::vector<uint8_t> out;
    PoseOptimization(T, Xd, edges, cam, out);
    pose_to(T, pose_out7);
    int good = 0;
    for (size_t e = 0; e < edges.size() && e < out.size(); ++e) {
        const int i = kp_of_edge[e];
        outlier[i] = out[e];
        if (!out[e] && (local_of_keypoint[i] >= 0 || held[i] == 1)) ++good;
    }
    return good;
}
extern "C" {
int oracle_track_local_map(const float* keys6, const uint8_t* desc, const float* uright, int n, int cols, int rows, const float* scales,
                           const float* inv_sigma2, int nlevels, float log_scale, const float* pose7, const double* cam5, const uint8_t* held,
                           const float* held_Xw, const MapPointPOD* pts, int m, float th, int far_points, float th_far, double* pose_out7,
                           int* local_of_keypoint, uint8_t* outlier, int* n_matches) {
    FrameView F;
    F.keys = kps_from(keys6, n);
    F.desc.assign(desc, desc + (size_t)n * 32);
    F.uRight.assign(uright, uright + n);
    F.cols = cols; F.rows = rows;
    return track_local_core(F, scales, inv_sigma2, nlevels, log_scale, pose7, cam5, held, held_Xw, pts, m, th, far_points, th_far, pose_out7,
                            local_of_keypoint, outlier, n_matches);
}

// ---- one sequence of the per-frame loop, as the CPU baseline of bench.py runs it -----------------------------------------------------
// State that persists from frame to frame in the reference: the two ORB extractors (Tracking.cc mpORBextractorLeft/Right), the ikd-Tree map
// and the local-map cube (LidarFrontEnd.cpp globals `ikdtree`, `LocalMap_Points`).
struct OracleSequence {
    ORBextractor el, er;
    KdTree map;
    LocalMapBox box;
    OracleSequence(int nfeatures, float scale, int nlevels, int ini, int mn) : el(nfeatures, scale, nlevels, ini, mn), er(nfeatures, scale, nlevels, ini, mn) {}
};
void* oracle_sequence_create(int nfeatures, float scale, int nlevels, int ini_th, int min_th, const PointXYZINormal* map_pts, int n_map) {
    OracleSequence* s = new OracleSequence(nfeatures, scale, nlevels, ini_th, min_th);
    s->map.Build(PointVector(map_pts, map_pts + n_map));
    return s;
}
void oracle_sequence_destroy(void* h) { delete (OracleSequence*)h; }
int oracle_sequence_map_size(void* h) { return (int)((OracleSequence*)h)->map.valid_size(); }

// One frame with the reference's threads: the LiDAR thread runs Preprocess::process, lasermap_fov_segment (+ the box
// deletions), the voxel filter and feature_extraction (LidarFrontEnd.cpp:886-962); the tracking thread builds the Frame (left / right ORB
// on two threads, ComputeStereoMatches), then TrackWithMotionModel and TrackLocalMap (Tracking.cc:2038,2218), then, at SyncWithLidar,
// UpdateMap -> map_incremental on the shared tree (Tracking.cc:1602-1603).  `local` = the frame's local map points as UpdateLocalMap
// left them; held / held_Xw = the map points the frame's keypoints hold after TrackWithMotionModel (precomputed by the caller: they are
// object state in the reference).  out4: inliers of the motion-model step, mnMatchesInliers, selected LiDAR features, map size.
int oracle_sequence_frame(void* h, const uint8_t* il, const uint8_t* ir, int w, int hgt, float mbf, float mb, const VelodynePoint* raw, int n_raw,
                          const double* state24, const float* pose_pred7, const float* pose_last7, const double* cam5, float th, int n_last,
                          const uint8_t* has_point, const uint8_t* outlier_last, const float* Xw, const float* last_keys6, const uint8_t* mp_desc,
                          int n_held, const uint8_t* held, const float* held_Xw, const MapPointPOD* local, int n_local, float th_local,
                          double cube_len, double det_range, double* pose_out7, int* out4, int imu_mode) {
    OracleSequence* S = (OracleSequence*)h;
    LidarState st;
    std::memcpy(st.rot, state24, 9 * sizeof(double)); std::memcpy(st.pos, state24 + 9, 3 * sizeof(double));
    std::memcpy(st.offset_R_L_I, state24 + 12, 9 * sizeof(double)); std::memcpy(st.offset_T_L_I, state24 + 21, 3 * sizeof(double));
    PointVector down;
    FeatureExtraction fe;
    std::thread lidar([&] {
        PointVector pre = preprocess_velodyne(raw, n_raw, 2, 2.0, 1e-3f);
        double pos_lid[3];  // pos_lid = state.pos + state.rot * offset_T_L_I (LidarFrontEnd.cpp:706 / :905)
        for (int r = 0; r < 3; ++r) pos_lid[r] = st.pos[r] + st.rot[3 * r] * st.offset_T_L_I[0] + st.rot[3 * r + 1] * st.offset_T_L_I[1] + st.rot[3 * r + 2] * st.offset_T_L_I[2];
        const std::vector<BoxPointType> rm = lasermap_fov_segment(S->box, pos_lid, cube_len, det_range);
        if (!rm.empty()) S->map.Delete_Point_Boxes(rm);
        down = voxel_grid_filter(pre, 0.5f);
        fe = feature_extraction(down, st, S->map);
    });
    std::vector<KeyPoint> kl, kr;
    std::vector<uint8_t> dl, dr;
    const int lap[2] = {0, 0};
    std::thread tl([&] { S->el.extract(Img::view(il, w, hgt, w), kl, dl, lap); });
    std::thread tr([&] { S->er.extract(Img::view(ir, w, hgt, w), kr, dr, lap); });
    tl.join();
    tr.join();
    StereoResult r = ComputeStereoMatches(S->el, S->er, kl, dl, kr, dr, mbf, mb);
    FrameView F;
    F.keys = kl; F.desc = dl; F.uRight = r.uRight; F.occupied.assign(kl.size(), 0); F.cols = w; F.rows = hgt;
    const int n = (int)kl.size();
    std::vector<int> mp(n + 1), lk(n + 1);
    int nm = 0, nm2 = 0;
    double pose_mm[7];
    // imu_mode: the IMU is initialised -- TrackWithMotionModel is PredictStateIMU() and nothing else (Tracking.cc:2746-2752: no search against the last
    // frame, no PoseOptimization), and TrackLocalMap's optimiser is the pose-inertial one, which the caller runs on the correspondences
    if (imu_mode) { out4[0] = 0; for (int c = 0; c < 7; ++c) pose_mm[c] = (double)pose_pred7[c]; }
    else
    out4[0] = track_core(F, S->el.mvScaleFactor.data(), S->el.mvInvLevelSigma2.data(), (int)S->el.mvScaleFactor.size(), pose_pred7, pose_last7, cam5, mb,
                         th, n_last, has_point, outlier_last, Xw, kps_from(last_keys6, n_last), mp_desc, pose_mm, mp.data(), &nm);
    // TrackLocalMap starts from the pose TrackWithMotionModel left (float on the Frame)
    float pose_f[7];
    for (int c = 0; c < 7; ++c) pose_f[c] = (float)pose_mm[c];
    std::vector<uint8_t> held_n(n, 0), outl(n + 1);
    std::vector<float> heldX(3 * (size_t)n, 0.f);
    const int nh = std::min(n, n_held);
    std::memcpy(held_n.data(), held, nh);
    std::memcpy(heldX.data(), held_Xw, 3 * (size_t)nh * sizeof(float));
    const float log_scale = std::log(S->el.mvScaleFactor[1]);
    out4[1] = track_local_core(F, S->el.mvScaleFactor.data(), S->el.mvInvLevelSigma2.data(), (int)S->el.mvScaleFactor.size(), log_scale, pose_f, cam5,
                               held_n.data(), heldX.data(), local, n_local, th_local, 0, 0.f, pose_out7, lk.data(), outl.data(), &nm2, imu_mode == 0);
    lidar.join();
    out4[2] = fe.effct_feat_num;
    // UpdateMap (LidarFrontEnd.cpp:1075-1079) -> map_incremental (:387-435)
    const MapIncrement inc = map_incremental_lists(down, st, fe.Nearest_Points, true, 0.5);
    S->map.Add_Points(inc.PointToAdd, true, 0.5f);
    S->map.Add_Points(inc.PointNoNeedDownsample, false, 0.5f);
    out4[3] = (int)S->map.valid_size();
    return out4[1];
}

// ORBmatcher::Fuse, search part: best keypoint (or -1) and best distance per map point; returns how many would be fused
int oracle_fuse_search(const float* keys6, const uint8_t* desc, const float* uright, int n, int cols, int rows, const float* pose7, const float* cam4,
                       float bf, const float* scales, const float* inv_sigma2, int nlevels, float log_scale, const MapPointPOD* pts,
                       const uint8_t* valid, int m, float th, int32_t* best_idx, int32_t* best_dist) {
    FrameView F;
    F.keys = kps_from(keys6, n);
    F.desc.assign(desc, desc + (size_t)n * 32);
    F.uRight.assign(uright, uright + n);
    F.occupied.assign(n, 0);
    F.cols = cols; F.rows = rows;
    SE3f Tcw;
    std::memcpy(Tcw.q, pose7, 16); std::memcpy(Tcw.t, pose7 + 4, 12);
    std::vector<MapPointView> mps(m);
    for (int i = 0; i < m; ++i) {
        std::memcpy(mps[i].pos, pts[i].pos, 12); std::memcpy(mps[i].normal, pts[i].normal, 12);
        mps[i].min_dist = pts[i].min_distance; mps[i].max_dist = pts[i].max_distance; mps[i].mfMaxDistance = pts[i].max_distance_raw;
        std::memcpy(mps[i].desc, pts[i].desc, 32);
    }
    std::vector<int> bi, bd;
    const int nf = FuseSearch(F, Tcw, CamF{cam4[0], cam4[1], cam4[2], cam4[3]}, bf, std::vector<float>(scales, scales + nlevels),
                              std::vector<float>(inv_sigma2, inv_sigma2 + nlevels), log_scale, mps, std::vector<uint8_t>(valid, valid + m), th, bi, bd);
    for (int i = 0; i < m; ++i) { best_idx[i] = bi[i]; best_dist[i] = bd[i]; }
    return nf;
}

// single-function probes for unit tests
float oracle_fast_atan2(float y, float x) { return fastAtan2(y, x); }
int oracle_fast9_16(const uint8_t* img, int stride, int w, int h, int th, int nms, float* xyr, int cap) {
    std::vector<KeyPoint> k;
    FAST9_16(img, stride, w, h, th, nms != 0, k);
    if ((int)k.size() > cap) return -1;
    for (size_t i = 0; i < k.size(); ++i) { xyr[3 * i] = k[i].x; xyr[3 * i + 1] = k[i].y; xyr[3 * i + 2] = k[i].response; }
    return (int)k.size();
}
void oracle_resize_linear(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh) {
    Img s = Img::view(src, sw, sh, sw);
    Img d(dw, dh);
    resizeLinearU8(s, d);
    std::memcpy(dst, d.store.data(), (size_t)dw * dh);
}
void oracle_set_gauss_variant(int v) { gauss_variant() = v == 1 ? 1 : 0; }
int oracle_get_gauss_variant() { return gauss_variant(); }
void oracle_gaussian_blur7(const uint8_t* src, int w, int h, uint8_t* dst) {
    Img s = Img::view(src, w, h, w);
    Img d(w, h);
    gaussianBlur7(s, d);
    std::memcpy(dst, d.store.data(), (size_t)w * h);
}

// ---- persistent map maintenance ----------------------------------------------------------------------------------------------
// One map_incremental step on a map given as a point list: feature extraction of the down-sampled scan at state_extract
// (a k-d tree built from the map), the two insertion lists at state_update, Add_Points twice.  Returns the new map size;
// lists: n_to_add / n_no_need.
int oracle_map_incremental(const PointXYZINormal* map_pts, int n_map, const PointXYZINormal* feats_down_body, int n_down,
                           const double* state_extract24, const double* state_update24, int ekf_inited, double filter_size_map_min,
                           PointXYZINormal* map_out, int capacity, int* n_to_add, int* n_no_need) {
    auto state_from = [](const double* s24) {
        LidarState st;
        std::memcpy(st.rot, s24, 9 * sizeof(double)); std::memcpy(st.pos, s24 + 9, 3 * sizeof(double));
        std::memcpy(st.offset_R_L_I, s24 + 12, 9 * sizeof(double)); std::memcpy(st.offset_T_L_I, s24 + 21, 3 * sizeof(double));
        return st;
    };
    KdTree tree;
    tree.Build(PointVector(map_pts, map_pts + n_map));
    const PointVector down(feats_down_body, feats_down_body + n_down);
    const FeatureExtraction fe = feature_extraction(down, state_from(state_extract24), tree);
    const MapIncrement inc = map_incremental_lists(down, state_from(state_update24), fe.Nearest_Points, ekf_inited != 0, filter_size_map_min);
    MapPoints mp;
    mp.pts.assign(map_pts, map_pts + n_map);
    mp.Add_Points(inc.PointToAdd, true, (float)filter_size_map_min);
    mp.Add_Points(inc.PointNoNeedDownsample, false, (float)filter_size_map_min);
    if (n_to_add) *n_to_add = (int)inc.PointToAdd.size();
    if (n_no_need) *n_no_need = (int)inc.PointNoNeedDownsample.size();
    const int k = std::min((int)mp.pts.size(), capacity);
    std::memcpy(map_out, mp.pts.data(), (size_t)k * sizeof(PointXYZINormal));
    return (int)mp.pts.size();
}
int oracle_map_delete_boxes(const PointXYZINormal* map_pts, int n_map, const float* boxes6, int n_boxes, PointXYZINormal* map_out) {
    MapPoints mp;
    mp.pts.assign(map_pts, map_pts + n_map);
    std::vector<BoxPointType> boxes(n_boxes);
    for (int b = 0; b < n_boxes; ++b) { std::memcpy(boxes[b].vertex_min, boxes6 + 6 * b, 12); std::memcpy(boxes[b].vertex_max, boxes6 + 6 * b + 3, 12); }
    mp.Delete_Point_Boxes(boxes);
    std::memcpy(map_out, mp.pts.data(), mp.pts.size() * sizeof(PointXYZINormal));
    return (int)mp.pts.size();
}
// lm7: vertex_min 3, vertex_max 3, initialized (as floats, in/out); returns the number of boxes written to boxes6
int oracle_fov_segment(float* lm7, const double* pos, double cube_len, double det_range, float* boxes6) {
    LocalMapBox lm;
    std::memcpy(lm.box.vertex_min, lm7, 12); std::memcpy(lm.box.vertex_max, lm7 + 3, 12);
    lm.initialized = lm7[6] != 0;
    const std::vector<BoxPointType> b = lasermap_fov_segment(lm, pos, cube_len, det_range);
    std::memcpy(lm7, lm.box.vertex_min, 12); std::memcpy(lm7 + 3, lm.box.vertex_max, 12);
    lm7[6] = lm.initialized ? 1.f : 0.f;
    for (size_t i = 0; i < b.size(); ++i) { std::memcpy(boxes6 + 6 * i, b[i].vertex_min, 12); std::memcpy(boxes6 + 6 * i + 3, b[i].vertex_max, 12); }
    return (int)b.size();
}

// ---- pose plumbing between the camera thread and the LiDAR front end (row b4) ------------------------------------------------------
static SE3F se3_from7(const float* p) { SE3F T; std::memcpy(T.q, p, 16); std::memcpy(T.t, p + 4, 12); return T; }
static void se3_to7(const SE3F& T, float* p) { std::memcpy(p, T.q, 16); std::memcpy(p + 4, T.t, 12); }
void oracle_se3f_ops(const float* a7, const float* b7, float t, float* inv7, float* mul7, float* log6, float* exp7_of_log, float* interp7) {
    const SE3F A = se3_from7(a7), B = se3_from7(b7);
    se3_to7(se3f_inverse(A), inv7);
    se3_to7(se3f_mul(A, B), mul7);
    se3f_log(A, log6);
    se3_to7(se3f_exp(log6), exp7_of_log);
    se3_to7(InterpolateSE3(A, B, t), interp7);
}
void oracle_update_lidar_pose(const float* Tcw_last7, const float* velocity7, double time_ratio, const float* Tcl7, double* state24, double* pos_lid3) {
    LidarState st;
    std::memcpy(st.rot, state24, 72); std::memcpy(st.pos, state24 + 9, 24); std::memcpy(st.offset_R_L_I, state24 + 12, 72); std::memcpy(st.offset_T_L_I, state24 + 21, 24);
    UpdateLidarPose(se3_from7(Tcw_last7), se3_from7(velocity7), time_ratio, se3_from7(Tcl7), st, pos_lid3);
    std::memcpy(state24, st.rot, 72); std::memcpy(state24 + 9, st.pos, 24);
}
void oracle_transform_point_cloud(const PointXYZINormal* in, int n, const float* T7, PointXYZINormal* out) {
    const PointVector o = transformPointCloud(PointVector(in, in + n), se3_from7(T7));
    std::memcpy(out, o.data(), (size_t)n * sizeof(PointXYZINormal));
}
void oracle_sync_transform(const float* Tcw_frame7, const float* Tcw_last7, const float* Tcw_cur7, float ratio, const float* Tlc7, const float* Tcl7, float* out7) {
    se3_to7(sync_transform(se3_from7(Tcw_frame7), se3_from7(Tcw_last7), se3_from7(Tcw_cur7), ratio, se3_from7(Tlc7), se3_from7(Tcl7)), out7);
}
void oracle_keyframe_transform(const float* Tcw_cur7, const float* rel7, const float* Tcw_refkf7, const float* Tlc7, const float* Tcl7, float* out7) {
    se3_to7(keyframe_transform(se3_from7(Tcw_cur7), se3_from7(rel7), se3_from7(Tcw_refkf7), se3_from7(Tlc7), se3_from7(Tcl7)), out7);
}

// ---- LiDAR motion compensation ---------------------------------------------------------------------------------------------
// poses: 22 doubles each (offset_time, acc, gyr, vel, pos, rot); state24 like the other LiDAR entries
void oracle_undistort(PointXYZINormal* pts, int n, const double* poses22, int n_poses, const double* state24) {
    PointVector v(pts, pts + n);
    std::vector<Pose6D> P(n_poses);
    for (int i = 0; i < n_poses; ++i) std::memcpy(&P[i], poses22 + 22 * i, sizeof(Pose6D));
    LidarState st;
    std::memcpy(st.rot, state24, 9 * sizeof(double));
    std::memcpy(st.pos, state24 + 9, 3 * sizeof(double));
    std::memcpy(st.offset_R_L_I, state24 + 12, 9 * sizeof(double));
    std::memcpy(st.offset_T_L_I, state24 + 21, 3 * sizeof(double));
    UndistortPcl(v, P, st);
    std::memcpy(pts, v.data(), (size_t)n * sizeof(PointXYZINormal));
}
// state42: pos 3, rot 9, vel 3, bg 3, ba 3, grav 3, offset_R 9, offset_T 3 (in/out); imu7: t, acc, gyr; last6: acc_s_last, angvel_last
int oracle_imu_propagate(double* state36, const double* imu7, int n_imu, double beg, double end, double last_end, double acc_scale,
                         const double* last6, double* poses22, int capacity) {
    static_assert(sizeof(ImuState) == 36 * sizeof(double), "layout");
    ImuState st;
    std::memcpy(&st, state36, sizeof(st));
    std::vector<ImuMeas> v(n_imu);
    for (int i = 0; i < n_imu; ++i) std::memcpy(&v[i], imu7 + 7 * i, sizeof(ImuMeas));
    std::vector<Pose6D> P = ForwardPropagate(st, v, beg, end, last_end, acc_scale, last6, last6 + 3);
    std::memcpy(state36, &st, sizeof(st));
    for (int i = 0; i < (int)P.size() && i < capacity; ++i) std::memcpy(poses22 + 22 * i, &P[i], sizeof(Pose6D));
    return (int)P.size();
}

// ---- iterated ESKF of the LiDAR-inertial front end (row b7) ------------------------------------------------------------------
void oracle_eskf_predict(double* state36, double* P529, const double* Q144, const double* acc, const double* gyr, double dt) {
    ImuState st;
    std::memcpy(&st, state36, sizeof(st));
    eskf_predict(st, P529, Q144, acc, gyr, dt);
    std::memcpy(state36, &st, sizeof(st));
}
void oracle_eskf_boxplus(double* state36, const double* d23) {
    ImuState st;
    std::memcpy(&st, state36, sizeof(st));
    eskf_boxplus(st, d23);
    std::memcpy(state36, &st, sizeof(st));
}
void oracle_eskf_boxminus(const double* a36, const double* b36, double* d23) {
    ImuState a, b;
    std::memcpy(&a, a36, sizeof(a)); std::memcpy(&b, b36, sizeof(b));
    eskf_boxminus(a, b, d23);
}
void oracle_s2(const double* g, const double* delta2, double* Bx6, double* Nx6, double* Mx6) {
    s2_Bx(g, Bx6); s2_Nx_yy(g, Nx6); s2_Mx(g, delta2, Mx6);
}
// out6: calls, effct_feat_num, searches, converged, finished, res_mean_last
void oracle_eskf_update(double* state36, double* P529, void* tree, const PointXYZINormal* body, int n, double R, int max_iter, const double* limit23,
                        int extrinsic_est_en, double* out6) {
    ImuState st;
    std::memcpy(&st, state36, sizeof(st));
    PointVector b(body, body + n);
    const EskfUpdate u = eskf_update(st, P529, *static_cast<KdTree*>(tree), b, R, max_iter, limit23, extrinsic_est_en != 0);
    std::memcpy(state36, &st, sizeof(st));
    out6[0] = u.calls; out6[1] = u.effct_feat_num; out6[2] = u.searches; out6[3] = u.converged; out6[4] = u.finished; out6[5] = u.res_mean_last;
}
int oracle_imu_propagate_cov(double* state36, double* P529, const double* cov12, const double* imu7, int n_imu, double beg, double end, double last_end,
                             double acc_scale, const double* last6, double* poses22, int capacity) {
    ImuState st;
    std::memcpy(&st, state36, sizeof(st));
    std::vector<ImuMeas> v(n_imu);
    for (int i = 0; i < n_imu; ++i) std::memcpy(&v[i], imu7 + 7 * i, sizeof(ImuMeas));
    std::vector<Pose6D> P = ForwardPropagateCov(st, P529, cov12, v, beg, end, last_end, acc_scale, last6, last6 + 3);
    std::memcpy(state36, &st, sizeof(st));
    for (int i = 0; i < (int)P.size() && i < capacity; ++i) std::memcpy(poses22 + 22 * i, &P[i], sizeof(Pose6D));
    return (int)P.size();
}

// ---- map-point refresh (section 8f item 3) ---------------------------------------------------------------------------------
void oracle_map_points_refresh(int n_points, const int32_t* obs_off, const uint8_t* desc, const float* centres, const float* positions,
                               const float* ref_centres, const float* level_scale, float last_scale, int32_t* best_obs, float* normals,
                               float* min_dist, float* max_dist) {
    for (int p = 0; p < n_points; ++p) {
        const int b = obs_off[p], n = obs_off[p + 1] - b;
        if (n <= 0) { best_obs[p] = -1; continue; }  // the reference returns early and leaves the point as it is
        best_obs[p] = ComputeDistinctiveDescriptor(desc + 32 * (size_t)b, n);
        UpdateNormalAndDepth(centres + 3 * (size_t)b, n, positions + 3 * p, ref_centres + 3 * p, level_scale[p], last_scale, normals + 3 * p,
                             min_dist + p, max_dist + p);
    }
}

// ---- CreateNewMapPoints core (section 8f item 1) ---------------------------------------------------------------------------
struct KeyFrameViewPOD {  // the same layout as tc2li_keyframe_view
    int32_t n, n_nodes;
    const float* keys;  // tc2li_keypoint: x, y, size, angle, response (floats), octave (int32)
    const uint8_t* desc;
    const float *u_right, *depth;
    const uint8_t* has_point;
    const int32_t *fv_node, *fv_off, *fv_idx;
    float pose7[7];
    float pad_;
};
static KeyFrameView view_from(const KeyFrameViewPOD& p, std::vector<KeyPoint>& keys) {
    keys.resize(p.n);
    for (int i = 0; i < p.n; ++i) {
        const float* k = p.keys + 6 * (size_t)i;
        int32_t oct;
        std::memcpy(&oct, k + 5, 4);
        keys[i] = KeyPoint{k[0], k[1], k[2], k[3], k[4], oct};
    }
    KeyFrameView v;
    v.n = p.n; v.keys = keys.data(); v.desc = p.desc; v.u_right = p.u_right; v.depth = p.depth; v.has_point = p.has_point;
    v.n_nodes = p.n_nodes; v.fv_node = p.fv_node; v.fv_off = p.fv_off; v.fv_idx = p.fv_idx;
    std::memcpy(v.Tcw.q, p.pose7, 16); std::memcpy(v.Tcw.t, p.pose7 + 4, 12);
    return v;
}
int oracle_search_for_triangulation(const KeyFrameViewPOD* kf1, const KeyFrameViewPOD* kf2, const float* cam4, const float* scale_factors,
                                    const float* level_sigma2, int n_levels, int only_stereo, int coarse, int check_orientation, int32_t* match12) {
    std::vector<KeyPoint> k1, k2;
    const KeyFrameView a = view_from(*kf1, k1), b = view_from(*kf2, k2);
    std::vector<int> m;
    const int n = SearchForTriangulation(a, b, CamF{cam4[0], cam4[1], cam4[2], cam4[3]}, std::vector<float>(scale_factors, scale_factors + n_levels),
                                         std::vector<float>(level_sigma2, level_sigma2 + n_levels), only_stereo != 0, coarse != 0, check_orientation != 0, m);
    for (int i = 0; i < a.n; ++i) match12[i] = m[i];
    return n;
}
// out: per point idx1, neighbour, idx2, stereo (int32 x 4) and x3D (float x 3)
int oracle_create_new_map_points(const KeyFrameViewPOD* cur, const KeyFrameViewPOD* neigh, int n_neigh, const float* cam4, float mb, float mbf,
                                 const float* scale_factors, const float* level_sigma2, int n_levels, float scale_factor, int inertial,
                                 int far_points, float th_far, int coarse, int32_t* out_idx4, float* out_x3, int capacity) {
    std::vector<KeyPoint> kc;
    std::vector<std::vector<KeyPoint>> kn(n_neigh);
    const KeyFrameView c = view_from(*cur, kc);
    std::vector<KeyFrameView> nb(n_neigh);
    for (int j = 0; j < n_neigh; ++j) nb[j] = view_from(neigh[j], kn[j]);
    MappingParams prm{mb, mbf, scale_factor, inertial != 0, far_points != 0, th_far};
    const std::vector<NewMapPoint> pts = CreateNewMapPoints(c, nb, CamF{cam4[0], cam4[1], cam4[2], cam4[3]},
                                                            std::vector<float>(scale_factors, scale_factors + n_levels),
                                                            std::vector<float>(level_sigma2, level_sigma2 + n_levels), prm, coarse != 0);
    for (size_t k = 0; k < pts.size() && (int)k < capacity; ++k) {
        out_idx4[4 * k] = pts[k].idx1; out_idx4[4 * k + 1] = pts[k].neighbour; out_idx4[4 * k + 2] = pts[k].idx2; out_idx4[4 * k + 3] = pts[k].stereo;
        std::memcpy(out_x3 + 3 * k, pts[k].x3D, 12);
    }
    return (int)pts.size();
}

// ---- IMU pre-integration -------------------------------------------------------------------------------------------------
// out: dT, dR 9, dV 3, dP 3, JRg 9, JVg 9, JVa 9, JPg 9, JPa 9, avgA 3, avgW 3, C 225 (= 292 floats); samples as (t, a, w) with t double
struct ImuSamplePOD { double t; float a[3], w[3]; };
static int imu_preintegrate_impl(const ImuSamplePOD* samples, int n, double t_prev, double t_cur, const float* bias6, float ng, float na,
                                 float ngw, float naw, float* out292, bool float_eval) {
    ImuBias b{bias6[0], bias6[1], bias6[2], bias6[3], bias6[4], bias6[5]};
    Preintegrated p(b, ng, na, ngw, naw);
    std::vector<ImuSample> v(n);
    for (int i = 0; i < n; ++i) { v[i].t = samples[i].t; std::memcpy(v[i].a, samples[i].a, 12); std::memcpy(v[i].w, samples[i].w, 12); }
    const int steps = PreintegrateIMU(v, t_prev, t_cur, p, float_eval);
    float* o = out292;
    *o++ = p.dT;
    auto put = [&](const float* src, int k) { std::memcpy(o, src, k * sizeof(float)); o += k; };
    put(p.dR, 9); put(p.dV, 3); put(p.dP, 3); put(p.JRg, 9); put(p.JVg, 9); put(p.JVa, 9); put(p.JPg, 9); put(p.JPa, 9); put(p.avgA, 3);
    put(p.avgW, 3); put(p.C, 225);
    return steps;
}
int oracle_imu_preintegrate(const ImuSamplePOD* samples, int n, double t_prev, double t_cur, const float* bias6, float ng, float na,
                            float ngw, float naw, float* out292) {
    return imu_preintegrate_impl(samples, n, t_prev, t_cur, bias6, ng, na, ngw, naw, out292, false);
}
// the float evaluation (IntegrateNewMeasurementFloat)
int oracle_imu_preintegrate_f32(const ImuSamplePOD* samples, int n, double t_prev, double t_cur, const float* bias6, float ng, float na,
                                float ngw, float naw, float* out292) {
    return imu_preintegrate_impl(samples, n, t_prev, t_cur, bias6, ng, na, ngw, naw, out292, true);
}
void oracle_normalize_rotation_f32(const float* R, float* out) { NormalizeRotationFloat(R, out); }
// state prediction from the pre-integration of the given samples at another bias: out = Rwb2 9, twb2 3, Vwb2 3, dR 9, dV 3, dP 3
int oracle_imu_predict(const ImuSamplePOD* samples, int n, double t_prev, double t_cur, const float* bias6, const float* bias_eval6,
                       float ng, float na, float ngw, float naw, const float* Rwb1, const float* twb1, const float* Vwb1, float* out30) {
    ImuBias b{bias6[0], bias6[1], bias6[2], bias6[3], bias6[4], bias6[5]};
    ImuBias be{bias_eval6[0], bias_eval6[1], bias_eval6[2], bias_eval6[3], bias_eval6[4], bias_eval6[5]};
    Preintegrated p(b, ng, na, ngw, naw);
    std::vector<ImuSample> v(n);
    for (int i = 0; i < n; ++i) { v[i].t = samples[i].t; std::memcpy(v[i].a, samples[i].a, 12); std::memcpy(v[i].w, samples[i].w, 12); }
    const int steps = PreintegrateIMU(v, t_prev, t_cur, p);
    PredictStateIMU(p, be, Rwb1, twb1, Vwb1, out30, out30 + 9, out30 + 12);
    p.GetDeltaRotation(be, out30 + 15); p.GetDeltaVelocity(be, out30 + 24); p.GetDeltaPosition(be, out30 + 27);
    return steps;
}
void oracle_normalize_rotation(const float* R, float* out) { NormalizeRotation(R, out); }

// ---- visual-inertial local BA -----------------------------------------------------------------------------------------------
// kf33: Rcw 9, tcw 3, Rwb 9, twb 3, v 3, bg 3, ba 3 per keyframe (in/out); calib24: Rcb 9, tcb 3, Rbc 9, tbc 3;
// links: link4 = kf1, kf2, robust, info_scale (doubles); pre = per link the 292 floats of oracle_imu_preintegrate + 6 bias floats
static InertialKeyFrame kf_from(const double* s, uint8_t fixed, uint8_t has_imu) {
    InertialKeyFrame k;
    std::memcpy(k.Rcw, s, 72); std::memcpy(k.tcw, s + 9, 24); std::memcpy(k.Rwb, s + 12, 72); std::memcpy(k.twb, s + 21, 24);
    std::memcpy(k.v, s + 24, 24); std::memcpy(k.bg, s + 27, 24); std::memcpy(k.ba, s + 30, 24);
    k.fixed = fixed; k.has_imu = has_imu;
    return k;
}
static void kf_to(const InertialKeyFrame& k, double* s) {
    std::memcpy(s, k.Rcw, 72); std::memcpy(s + 9, k.tcw, 24); std::memcpy(s + 12, k.Rwb, 72); std::memcpy(s + 21, k.twb, 24);
    std::memcpy(s + 24, k.v, 24); std::memcpy(s + 27, k.bg, 24); std::memcpy(s + 30, k.ba, 24);
}
static Preintegrated preint_from(const float* f298) {
    ImuBias b{f298[292], f298[293], f298[294], f298[295], f298[296], f298[297]};
    Preintegrated p(b, 0, 0, 0, 0);
    const float* o = f298;
    p.dT = *o++;
    auto get = [&](float* dst, int k) { std::memcpy(dst, o, k * sizeof(float)); o += k; };
    get(p.dR, 9); get(p.dV, 3); get(p.dP, 3); get(p.JRg, 9); get(p.JVg, 9); get(p.JVa, 9); get(p.JPg, 9); get(p.JPa, 9); get(p.avgA, 3);
    get(p.avgW, 3); get(p.C, 225);
    return p;
}
static ImuCalibD calib_from(const double* c) {
    ImuCalibD k;
    std::memcpy(k.Rcb, c, 72); std::memcpy(k.tcb, c + 9, 24); std::memcpy(k.Rbc, c + 12, 72); std::memcpy(k.tbc, c + 21, 24);
    return k;
}
int oracle_local_inertial_ba(double* kf33, const uint8_t* fixed, const uint8_t* has_imu, int n_kf, const double* calib24, double* points3,
                             int n_points, const double* edges6, int n_edges, const double* link4, const float* pre298, int n_links,
                             const double* cam5, int iterations, double lambda_init, double* chi2_out, uint8_t* depth_pos, double* err2,
                             double* trace_chi2, double* trace_lambda, int* trace_trials, int trace_cap) {
    std::vector<InertialKeyFrame> kfs(n_kf);
    for (int k = 0; k < n_kf; ++k) kfs[k] = kf_from(kf33 + 33 * k, fixed[k], has_imu[k]);
    std::vector<Preintegrated> pre;
    pre.reserve(n_links);
    std::vector<InertialLink> links(n_links);
    for (int l = 0; l < n_links; ++l) pre.push_back(preint_from(pre298 + 298 * (size_t)l));
    for (int l = 0; l < n_links; ++l) {
        links[l].kf1 = (int)link4[4 * l]; links[l].kf2 = (int)link4[4 * l + 1]; links[l].robust = link4[4 * l + 2] != 0; links[l].info_scale = link4[4 * l + 3];
        links[l].pint = &pre[l];
    }
    std::vector<double> pts(points3, points3 + 3 * (size_t)n_points);
    Camera cam{cam5[0], cam5[1], cam5[2], cam5[3], cam5[4]};
    InertialBAResult r = LocalInertialBA(kfs, calib_from(calib24), pts, edges_from(edges6, n_edges), links, cam, iterations, lambda_init);
    for (int k = 0; k < n_kf; ++k) kf_to(kfs[k], kf33 + 33 * k);
    std::memcpy(points3, pts.data(), pts.size() * sizeof(double));
    for (int e = 0; e < n_edges; ++e) { if (chi2_out) chi2_out[e] = r.chi2[e]; if (depth_pos) depth_pos[e] = r.depth_pos[e]; }
    if (err2) { err2[0] = r.err; err2[1] = r.err_end; }
    const int m = std::min((int)r.trace.chi2.size(), trace_cap);
    for (int i = 0; i < m; ++i) { if (trace_chi2) trace_chi2[i] = r.trace.chi2[i]; if (trace_lambda) trace_lambda[i] = r.trace.lambda[i]; if (trace_trials) trace_trials[i] = r.trace.trials[i]; }
    return r.iterations;
}
// Optimizer::PoseInertialOptimizationLastKeyFrame (last_frame = 0) / LastFrame (1).  cur33 / other33: one kf33 record each (in/out);
// prior246: ConstraintPoseImu of the previous frame = Rwb 9, twb 3, vwb 3, bg 3, ba 3, H 225 (NULL for the keyframe form);
// edges6[e] = (index into Xw, unused, u, v, uR, invSigma2); prior_out246: the frame's new mpcpi; stats3: n_initial, n_bad, n_inliers.
// Returns nInitialCorrespondences - nBad.
int oracle_pose_inertial(double* cur33, double* other33, int last_frame, const double* prior246, const double* calib24, const float* pre298,
                         const float* pre_rw298, const double* Xw, const double* edges6, int n_edges, const uint8_t* close, const double* cam5,
                         int rec_init, uint8_t* outlier, double* prior_out246, int* stats3) {
    InertialKeyFrame cur = kf_from(cur33, 0, 1), other = kf_from(other33, last_frame ? 0 : 1, 1);
    PoseImuPrior prior;
    if (prior246) {
        std::memcpy(prior.Rwb, prior246, 72); std::memcpy(prior.twb, prior246 + 9, 24); std::memcpy(prior.vwb, prior246 + 12, 24);
        std::memcpy(prior.bg, prior246 + 15, 24); std::memcpy(prior.ba, prior246 + 18, 24); std::memcpy(prior.H, prior246 + 21, 225 * sizeof(double));
    }
    const Preintegrated pint = preint_from(pre298), pint_rw = preint_from(pre_rw298);
    int max_pt = -1;
    const std::vector<BAEdge> edges = edges_from(edges6, n_edges);
    for (const BAEdge& e : edges) max_pt = std::max(max_pt, e.point);
    const std::vector<double> X(Xw, Xw + 3 * (size_t)(max_pt + 1));
    Camera cam{cam5[0], cam5[1], cam5[2], cam5[3], cam5[4]};
    const PoseInertialResult r = PoseInertialOptimization(cur, other, last_frame != 0, prior246 ? &prior : nullptr, calib_from(calib24), pint, pint_rw, X, edges,
                                                          std::vector<uint8_t>(close, close + n_edges), cam, rec_init != 0);
    kf_to(cur, cur33); kf_to(other, other33);
    for (int e = 0; e < n_edges; ++e) outlier[e] = r.outlier[e];
    if (prior_out246) {
        std::memcpy(prior_out246, r.prior.Rwb, 72); std::memcpy(prior_out246 + 9, r.prior.twb, 24); std::memcpy(prior_out246 + 12, r.prior.vwb, 24);
        std::memcpy(prior_out246 + 15, r.prior.bg, 24); std::memcpy(prior_out246 + 18, r.prior.ba, 24); std::memcpy(prior_out246 + 21, r.prior.H, 225 * sizeof(double));
    }
    if (stats3) { stats3[0] = r.n_initial; stats3[1] = r.n_bad; stats3[2] = r.n_inliers; }
    return r.n_initial - r.n_bad;
}
// Optimizer::InertialOptimization (IMU initialisation): kf33 rows in temporal order (Rwb, twb, v read; v written), pre298[i] = row i's
// pre-integration from row i - 1 (row 0 unused); state17 = Rwg 9 | scale | bg 3 | ba 3 | (pad).  Returns the iterations.
int oracle_inertial_optimization(double* kf33, int n_kf, const float* pre298, double* state17, int mono, int fixed_vel, float priorG, float priorA, int its,
                                 double* err2, int* trials, double* trace_chi2, double* trace_lambda, int* trace_trials, int trace_cap) {
    std::vector<InertialKeyFrame> kfs(n_kf);
    for (int k = 0; k < n_kf; ++k) kfs[k] = kf_from(kf33 + 33 * k, 0, 1);
    std::vector<Preintegrated> pre;
    pre.reserve(n_kf);
    for (int k = 0; k < n_kf; ++k) pre.push_back(preint_from(pre298 + 298 * (size_t)k));
    std::vector<const Preintegrated*> pp(n_kf, nullptr);
    for (int k = 1; k < n_kf; ++k) pp[k] = &pre[k];
    double Rwg[9], scale = state17[9], bg[3], ba[3];
    std::memcpy(Rwg, state17, 72); std::memcpy(bg, state17 + 10, 24); std::memcpy(ba, state17 + 13, 24);
    const InertialInitResult r = InertialOptimization(kfs, pp, Rwg, scale, bg, ba, mono != 0, fixed_vel != 0, priorG, priorA, its);
    for (int k = 0; k < n_kf; ++k) kf_to(kfs[k], kf33 + 33 * k);
    std::memcpy(state17, Rwg, 72); state17[9] = scale; std::memcpy(state17 + 10, bg, 24); std::memcpy(state17 + 13, ba, 24);
    if (err2) { err2[0] = r.err; err2[1] = r.err_end; }
    if (trials) *trials = r.trials;
    const int m = std::min((int)r.trace.chi2.size(), trace_cap);
    for (int i = 0; i < m; ++i) { if (trace_chi2) trace_chi2[i] = r.trace.chi2[i]; if (trace_lambda) trace_lambda[i] = r.trace.lambda[i]; if (trace_trials) trace_trials[i] = r.trace.trials[i]; }
    return r.iterations;
}
int oracle_inertial_scale_refinement(const double* kf33, int n_kf, const float* pre298, double* Rwg9, double* scale, int its, double* err2) {
    std::vector<InertialKeyFrame> kfs(n_kf);
    for (int k = 0; k < n_kf; ++k) kfs[k] = kf_from(kf33 + 33 * k, 0, 1);
    std::vector<Preintegrated> pre;
    pre.reserve(n_kf);
    for (int k = 0; k < n_kf; ++k) pre.push_back(preint_from(pre298 + 298 * (size_t)k));
    std::vector<const Preintegrated*> pp(n_kf, nullptr);
    for (int k = 1; k < n_kf; ++k) pp[k] = &pre[k];
    return InertialScaleRefinement(kfs, pp, Rwg9, *scale, its, err2);
}
void oracle_initial_gravity_direction(const double* kf33, int n_kf, const float* pre298, float* vel, float* Rwg9) {
    std::vector<InertialKeyFrame> kfs(n_kf);
    for (int k = 0; k < n_kf; ++k) kfs[k] = kf_from(kf33 + 33 * k, 0, 1);
    std::vector<Preintegrated> pre;
    pre.reserve(n_kf);
    for (int k = 0; k < n_kf; ++k) pre.push_back(preint_from(pre298 + 298 * (size_t)k));
    std::vector<const Preintegrated*> pp(n_kf, nullptr);
    for (int k = 1; k < n_kf; ++k) pp[k] = &pre[k];
    InitialGravityDirection(kfs, pp, vel, Rwg9);
}
void oracle_inertial_gs_edge(const double* kf33_1, const double* kf33_2, const double* bg, const double* ba, const double* Rwg, double s, const float* pre298,
                             double* err9, double* J135) {
    const Preintegrated p = preint_from(pre298);
    inertial_gs_edge(kf_from(kf33_1, 0, 1), kf_from(kf33_2, 0, 1), bg, ba, Rwg, s, p, err9, J135);
}
// OptimizerWithLidar::LocalLVIBA: the same with EdgeLidar over the keyframes win_kf (rows of kf33)
int oracle_local_lviba(double* kf33, const uint8_t* fixed, const uint8_t* has_imu, int n_kf, const double* calib24, double* points3,
                       int n_points, const double* edges6, int n_edges, const double* link4, const float* pre298, int n_links,
                       const double* cam5, int iterations, double lambda_init, const int* win_kf, int n_win, const float* clouds,
                       const int* cloud_off, const float* Tcl7, const float* Tbl7, double weight, double* chi2_out, uint8_t* depth_pos,
                       double* err2, double* trace_chi2, double* trace_lambda, int* trace_trials, int trace_cap, int* n_planes,
                       double* lidar_out /* [2 + 6W + 36W^2]: error, chi2, JacT, Hessian at the end */) {
    std::vector<InertialKeyFrame> kfs(n_kf);
    for (int k = 0; k < n_kf; ++k) kfs[k] = kf_from(kf33 + 33 * k, fixed[k], has_imu[k]);
    std::vector<Preintegrated> pre;
    pre.reserve(n_links);
    std::vector<InertialLink> links(n_links);
    for (int l = 0; l < n_links; ++l) pre.push_back(preint_from(pre298 + 298 * (size_t)l));
    for (int l = 0; l < n_links; ++l) {
        links[l].kf1 = (int)link4[4 * l]; links[l].kf2 = (int)link4[4 * l + 1]; links[l].robust = link4[4 * l + 2] != 0; links[l].info_scale = link4[4 * l + 3];
        links[l].pint = &pre[l];
    }
    std::vector<double> pts(points3, points3 + 3 * (size_t)n_points);
    Camera cam{cam5[0], cam5[1], cam5[2], cam5[3], cam5[4]};
    SE3fQ Tcl, Tbl;
    std::memcpy(Tcl.q, Tcl7, 16); std::memcpy(Tcl.t, Tcl7 + 4, 12);
    std::memcpy(Tbl.q, Tbl7, 16); std::memcpy(Tbl.t, Tbl7 + 4, 12);
    LidarCovisRes lio(Tcl, Tbl);
    lio.win_size_ = n_win;
    std::vector<int> lk(win_kf, win_kf + n_win);
    for (int i = 0; i < n_win; ++i)
        lio.AddFromKeyFrame(se3f_from_rt(kfs[lk[i]].Rcw, kfs[lk[i]].tcw), std::vector<float>(clouds + 3 * (size_t)cloud_off[i], clouds + 3 * (size_t)cloud_off[i + 1]));
    lio.BuildVoxHess();
    if (n_planes) *n_planes = (int)lio.planes().size();
    EdgeLidar edge;
    edge.lio = &lio;
    edge.information = weight;
    edge.body = true;
    InertialBAResult r = LocalInertialBA(kfs, calib_from(calib24), pts, edges_from(edges6, n_edges), links, cam, iterations, lambda_init, &edge, &lk);
    for (int k = 0; k < n_kf; ++k) kf_to(kfs[k], kf33 + 33 * k);
    std::memcpy(points3, pts.data(), pts.size() * sizeof(double));
    for (int e = 0; e < n_edges; ++e) { if (chi2_out) chi2_out[e] = r.chi2[e]; if (depth_pos) depth_pos[e] = r.depth_pos[e]; }
    if (err2) { err2[0] = r.err; err2[1] = r.err_end; }
    const int m = std::min((int)r.trace.chi2.size(), trace_cap);
    for (int i = 0; i < m; ++i) { if (trace_chi2) trace_chi2[i] = r.trace.chi2[i]; if (trace_lambda) trace_lambda[i] = r.trace.lambda[i]; if (trace_trials) trace_trials[i] = r.trace.trials[i]; }
    if (lidar_out) {
        lidar_out[0] = edge.error; lidar_out[1] = edge.chi2();
        for (size_t k = 0; k < edge.JacT.size(); ++k) lidar_out[2 + k] = edge.JacT[k];
        for (size_t k = 0; k < edge.Hessian.size(); ++k) lidar_out[2 + edge.JacT.size() + k] = edge.Hessian[k];
    }
    return r.iterations;
}
// EdgeLidar (body variant) alone: planes from the window at the keyframe states kf33_build, then the edge's error (sqrt r) and
// ComputeJandH at kf33_eval
int oracle_lidar_window_evaluate_body(const double* kf33_build, const double* kf33_eval, const int* win_kf, int n_win, const float* clouds,
                                      const int* cloud_off, const float* Tcl7, const float* Tbl7, double* error, double* JacT, double* Hess) {
    SE3fQ Tcl, Tbl;
    std::memcpy(Tcl.q, Tcl7, 16); std::memcpy(Tcl.t, Tcl7 + 4, 12);
    std::memcpy(Tbl.q, Tbl7, 16); std::memcpy(Tbl.t, Tbl7 + 4, 12);
    LidarCovisRes lio(Tcl, Tbl);
    lio.win_size_ = n_win;
    std::vector<double> R(9 * n_win), t(3 * n_win);
    for (int i = 0; i < n_win; ++i) {
        const double* b = kf33_build + 33 * win_kf[i];
        lio.AddFromKeyFrame(se3f_from_rt(b, b + 9), std::vector<float>(clouds + 3 * (size_t)cloud_off[i], clouds + 3 * (size_t)cloud_off[i + 1]));
        std::memcpy(&R[9 * i], kf33_eval + 33 * win_kf[i], 72);
        std::memcpy(&t[3 * i], kf33_eval + 33 * win_kf[i] + 9, 24);
    }
    lio.BuildVoxHess();
    EdgeLidar edge;
    edge.lio = &lio;
    edge.body = true;
    edge.computeError(R.data(), t.data(), n_win);
    if (error) *error = edge.error;
    if (JacT && Hess) {
        edge.is_calc_hess = true;
        edge.linearizeOplus(R.data(), t.data(), n_win);
        std::memcpy(JacT, edge.JacT.data(), edge.JacT.size() * sizeof(double));
        std::memcpy(Hess, edge.Hessian.data(), edge.Hessian.size() * sizeof(double));
    }
    return (int)lio.planes().size();
}
// one inertial edge: error (9) and Jacobians (9 x 24) at the two keyframe states
void oracle_inertial_edge(const double* kf33_1, const double* kf33_2, const float* pre298, double* err9, double* J216) {
    const InertialKeyFrame k1 = kf_from(kf33_1, 0, 1), k2 = kf_from(kf33_2, 0, 1);
    const Preintegrated p = preint_from(pre298);
    inertial_edge(k1, k2, p, err9, J216);
}
// one visual edge in the ImuCamPose parameterisation: returns dim
int oracle_inertial_visual_edge(const double* kf33, const double* calib24, const double* X, const double* edge6, const double* cam5, double* err3,
                                double* A9, double* B18) {
    Camera cam{cam5[0], cam5[1], cam5[2], cam5[3], cam5[4]};
    return inertial_visual_edge(kf_from(kf33, 0, 1), calib_from(calib24), X, edges_from(edge6, 1)[0], cam, err3, A9, B18);
}
// ImuCamPose::Update applied to a keyframe state (its counter in/out)
void oracle_imu_pose_update(double* kf33, int* its, const double* calib24, const double* u6) {
    InertialKeyFrame k = kf_from(kf33, 0, 1);
    imu_pose_update(k, *its, calib_from(calib24), u6);
    kf_to(k, kf33);
}

}  // extern "C"

// ---- UpdateLocalKeyFrames / UpdateLocalPoints (section 8f item 3) ----------------------------------------------------------
extern "C" int oracle_update_local_map(int n_keyframes, int n_points, const uint8_t* kf_bad, const int32_t* covis_off, const int32_t* covis,
                                       const int32_t* child_off, const int32_t* children, const int32_t* parent, const int32_t* prev_kf,
                                       const int32_t* match_off, const int32_t* matches, const uint8_t* point_bad, const int32_t* obs_off,
                                       const int32_t* obs_kf, const int32_t* frame_points, int n_frame_points, int temporal_last_kf,
                                       int32_t* local_kfs, int32_t* n_local_kfs, int32_t* reference_kf, int32_t* local_points,
                                       int32_t* n_local_points, uint8_t* frame_cleared) {
    oracle::MapGraph g;
    g.n_keyframes = n_keyframes; g.n_points = n_points; g.kf_bad = kf_bad; g.covis_off = covis_off; g.covis = covis; g.child_off = child_off;
    g.children = children; g.parent = parent; g.prev_kf = prev_kf; g.match_off = match_off; g.matches = matches; g.point_bad = point_bad;
    g.obs_off = obs_off; g.obs_kf = obs_kf;
    const oracle::LocalMap m = oracle::UpdateLocalMap(g, frame_points, n_frame_points, temporal_last_kf);
    for (size_t i = 0; i < m.keyframes.size(); ++i) local_kfs[i] = m.keyframes[i];
    for (size_t i = 0; i < m.points.size(); ++i) local_points[i] = m.points[i];
    for (int i = 0; i < n_frame_points; ++i) frame_cleared[i] = m.frame_cleared[i];
    *n_local_kfs = (int)m.keyframes.size(); *n_local_points = (int)m.points.size(); *reference_kf = m.reference_kf;
    return 0;
}
