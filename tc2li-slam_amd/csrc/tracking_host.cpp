// Host side of tc2li_track_motion_model_batch / tc2li_track_local_map_batch (include/tc2li_hip.h): the data path of
// Tracking::TrackWithMotionModel (SF/src/Tracking.cc:2737-2834) and Tracking::TrackLocalMap (:3119-3230) for a batch of independent
// frames.  Everything per keypoint / per map point runs on the device (tracking_kernels.hip: query construction, rotation filter, edge
// lists, outlier bookkeeping; matcher_kernels.hip: the search; pose_opt_kernel.hip: Optimizer::PoseOptimization).  The host packs the
// callers' arrays into one pinned staging block per call (one upload), decides the few per-frame branches the reference takes between the
// stages (fewer than 20 matches: search again with a wider window, Tracking.cc:2774-2783) and receives the per-frame results.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "common.hpp"
#include "matcher_device.hpp"
#include "orb_handle.hpp"
#include "pose_opt_device.hpp"
#include "tracking_device.hpp"

using namespace tc2li;

namespace {

constexpr int kCellsPlus1 = 64 * 48 + 1;  // the matcher's feature grid (matcher_kernels.hip)

struct TrackWs {
    PinnedBuf<uint8_t> h_stage;
    DevBuf<uint8_t> d_stage;
    PinnedBuf<TrackFrameDev> h_frames;
    DevBuf<TrackFrameDev> d_frames;
    PinnedBuf<MatchFrameDev> h_mframes;
    DevBuf<MatchFrameDev> d_mframes;
    PinnedBuf<int32_t> h_pass, h_key_base, h_small, h_nmatch, h_amb;
    DevBuf<int32_t> d_pass, d_key_base;
    DevBuf<MatchQuery> d_queries;
    DevBuf<float> d_amb_ratio, d_amb_r;
    PinnedBuf<float> h_amb_ratio;
    DevBuf<int32_t> d_amb_ids, d_amb_level;
    PinnedBuf<int32_t> h_amb_level;
    DevBuf<int32_t> d_query_frame, d_match, d_prev, d_rounds, d_nmatch, d_amb;
    DevBuf<int32_t> d_cell_start, d_cand_off, d_cand_cnt, d_pool_top;
    DevBuf<uint16_t> d_items;
    DevBuf<uint32_t> d_pool;
    DevBuf<PoseProblem> d_probs;
    DevBuf<BaEdge> d_edges;
    DevBuf<double> d_Xw, d_poses, d_chi2;
    DevBuf<uint8_t> d_outlier, d_occ, d_outlier_key;
    DevBuf<int32_t> d_edge_kp, d_inliers, d_ninl, d_of_key;
};
// one work space per host thread: TrackWithMotionModel and TrackLocalMap of different batches run side by side
TrackWs& tws() { static thread_local TrackWs w; return w; }

inline size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }

inline void quat_rotate_f(const float q[4], const float v[3], float out[3]) {  // Eigen::Quaternionf::_transformVector
    float uv[3] = {q[1] * v[2] - q[2] * v[1], q[2] * v[0] - q[0] * v[2], q[0] * v[1] - q[1] * v[0]};
    uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
    out[0] = v[0] + q[3] * uv[0] + (q[1] * uv[2] - q[2] * uv[1]);
    out[1] = v[1] + q[3] * uv[1] + (q[2] * uv[0] - q[0] * uv[2]);
    out[2] = v[2] + q[3] * uv[2] + (q[0] * uv[1] - q[1] * uv[0]);
}

void fill_const(const tc2li_orb* o, const tc2li_camera* cam, float b, int capacity, TrackConst& C) {
    memset(&C, 0, sizeof(C));
    C.cam4[0] = (float)cam->fx; C.cam4[1] = (float)cam->fy; C.cam4[2] = (float)cam->cx; C.cam4[3] = (float)cam->cy;
    C.b = b; C.bf = (float)cam->bf;
    C.n_levels = o->prm.nlevels;
    for (int l = 0; l < C.n_levels; ++l) { C.scale[l] = o->scale[l]; C.inv_sigma2[l] = o->inv_sigma2[l]; }
    C.cols = o->cur_w; C.rows = o->cur_h; C.capacity = capacity;
    C.log_scale = std::log(o->prm.scale_factor);  // mfLogScaleFactor = log(mfScaleFactor) (SF/src/Frame.cc:96)
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------------------------------
namespace {

struct Pass {
    TrackWs& w;
    tc2li_orb* o;
    int n_frames, total_q, capacity;
    const float* d_ur;       // [n_frames][capacity]
    const uint8_t* d_occ;    // [n_frames][capacity] or NULL
    int mode;
    float nn_ratio;
    bool check_orientation;
    hipStream_t st;
    int pool_cap = 0;

    int prepare() {
        const char* per_query = getenv("TC2LI_MATCH_POOL_PER_QUERY");  // tests shrink the pool to reach the overflow path
        pool_cap = std::max(1, per_query ? atoi(per_query) : 32) * std::max(total_q, 1);
        TC2LI_HIP_CHECK(w.d_mframes.ensure(n_frames)); TC2LI_HIP_CHECK(w.h_mframes.ensure(n_frames));
        TC2LI_HIP_CHECK(w.d_pass.ensure(n_frames)); TC2LI_HIP_CHECK(w.h_pass.ensure(n_frames));
        TC2LI_HIP_CHECK(w.d_key_base.ensure(n_frames)); TC2LI_HIP_CHECK(w.h_key_base.ensure(n_frames));
        TC2LI_HIP_CHECK(w.d_prev.ensure(std::max(total_q, 1))); TC2LI_HIP_CHECK(w.d_rounds.ensure(n_frames));
        TC2LI_HIP_CHECK(w.d_nmatch.ensure(n_frames)); TC2LI_HIP_CHECK(w.h_nmatch.ensure(n_frames));
        TC2LI_HIP_CHECK(w.d_cell_start.ensure((size_t)n_frames * kCellsPlus1));
        TC2LI_HIP_CHECK(w.d_cand_off.ensure(std::max(total_q, 1))); TC2LI_HIP_CHECK(w.d_cand_cnt.ensure(std::max(total_q, 1)));
        TC2LI_HIP_CHECK(w.d_pool_top.ensure(2)); TC2LI_HIP_CHECK(w.h_small.ensure(8));
        TC2LI_HIP_CHECK(w.d_items.ensure((size_t)n_frames * std::max(capacity, 1))); TC2LI_HIP_CHECK(w.d_pool.ensure(pool_cap));
        return TC2LI_OK;
    }
    // frames of the pass are in w.h_pass[0 .. n_pass); their TrackFrameDev (slot set) are on the device already
    int queue(int n_pass, bool lists) {
        for (int k = 0; k < n_pass; ++k) {
            const int f = w.h_pass.p[k];
            const TrackFrameDev& F = w.h_frames.p[f];
            w.h_mframes.p[k] = MatchFrameDev{o->d_mkeys.p + F.key_off, o->d_desc.p + (size_t)F.key_off * 32, d_ur + (size_t)f * capacity,
                                             d_occ ? d_occ + (size_t)f * capacity : nullptr, w.d_queries.p + F.q_off, F.n_keys, F.n_q, F.q_off, 0,
                                             0.0f, (float)o->cur_w, 0.0f, (float)o->cur_h};
            w.h_key_base.p[k] = f * capacity;
        }
        TC2LI_HIP_CHECK(hipMemcpyAsync(w.d_mframes.p, w.h_mframes.p, n_pass * sizeof(MatchFrameDev), hipMemcpyHostToDevice, st));
        TC2LI_HIP_CHECK(hipMemcpyAsync(w.d_key_base.p, w.h_key_base.p, n_pass * sizeof(int32_t), hipMemcpyHostToDevice, st));
        TC2LI_HIP_CHECK(hipMemcpyAsync(w.d_pass.p, w.h_pass.p, n_pass * sizeof(int32_t), hipMemcpyHostToDevice, st));
        if (lists) {
            MatchLists L{w.d_cell_start.p, w.d_items.p, w.d_key_base.p, w.d_cand_off.p, w.d_cand_cnt.p, w.d_pool.p, w.d_pool_top.p, pool_cap, 0};
            launch_match_lists(w.d_mframes.p, n_pass, w.d_query_frame.p, total_q, L, mode, nn_ratio, w.d_match.p, w.d_prev.p, w.d_rounds.p, st);
            TC2LI_HIP_CHECK(hipMemcpyAsync(w.h_small.p, w.d_pool_top.p, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        } else {  // the one-kernel form: same result, no candidate pool
            launch_match_by_projection(w.d_mframes.p, n_pass, mode, nn_ratio, w.d_match.p, w.d_prev.p, w.d_rounds.p, st);
            w.h_small.p[1] = 0;
        }
        launch_track_count(w.d_frames.p, w.d_pass.p, n_pass, w.d_queries.p, o->d_angles.p, check_orientation ? 1 : 0, w.d_match.p, w.d_nmatch.p, st);
        TC2LI_HIP_CHECK(hipGetLastError());
        TC2LI_HIP_CHECK(hipMemcpyAsync(w.h_nmatch.p, w.d_nmatch.p, n_frames * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        return TC2LI_OK;
    }
};

// Resets the matches of the pass's frames (the query kernels do that when they rebuild the queries; the overflow path needs it alone).
int reset_matches(TrackWs& w, int n_pass, hipStream_t st) {
    for (int k = 0; k < n_pass; ++k) {
        const TrackFrameDev& F = w.h_frames.p[w.h_pass.p[k]];
        if (F.n_q) TC2LI_HIP_CHECK(hipMemsetAsync(w.d_match.p + F.q_off, 0xff, (size_t)F.n_q * sizeof(int32_t), st));
    }
    return TC2LI_OK;
}

}  // namespace

extern "C" int tc2li_track_motion_model_batch(tc2li_orb* o, int n_frames, const tc2li_keypoint* keypoints, const float* u_right,
                                              int capacity, const tc2li_last_frame* last, const float* pose_pred7,
                                              const tc2li_camera* cam, float b, float th, double* poses7,
                                              int32_t* map_point_of_keypoint, int32_t* n_matches, int32_t* n_inliers, void* stream_) {
    if (!o || n_frames < 0 || capacity < 0 || !keypoints || !u_right || !last || !pose_pred7 || !cam || !poses7 || !map_point_of_keypoint ||
        !n_matches || !n_inliers) {
        set_error("tc2li_track_motion_model_batch: invalid argument");
        return TC2LI_ERR_INVALID;
    }
    if (n_frames == 0) return 0;
    if (2 * n_frames > o->last_nimg || !o->last_plain_order) {
        set_error("tc2li_track_motion_model_batch: needs the features of a preceding tc2li_orb_extract_batch call with lapping area "
                  "{0,0} and 2*n_frames images");
        return TC2LI_ERR_INVALID;
    }
    if (!device_ready()) return TC2LI_ERR_NO_DEVICE;
    hipStream_t st = (hipStream_t)stream_;
    static const bool kTiming = getenv("TC2LI_TRACK_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tm[6] = {0, 0, 0, 0, 0, 0}, t0 = now();
    TrackWs& w = tws();
    TrackConst C;
    fill_const(o, cam, b, capacity, C);
    const int L = C.n_levels;

    TC2LI_HIP_CHECK(w.h_frames.ensure(n_frames)); TC2LI_HIP_CHECK(w.d_frames.ensure(n_frames));
    int total_q = 0;
    for (int f = 0; f < n_frames; ++f) {
        TrackFrameDev& F = w.h_frames.p[f];
        memset(&F, 0, sizeof(F));
        F.key_off = o->last_kp_off[2 * f];
        F.n_keys = o->last_kp_cnt[2 * f];
        if (F.n_keys > capacity) { set_error("capacity %d < %d keypoints", capacity, F.n_keys); return TC2LI_ERR_CAPACITY; }
        if (F.n_keys > kMaxMatchKeys) { set_error("frame has %d keypoints, the matcher supports %d", F.n_keys, kMaxMatchKeys); return TC2LI_ERR_CAPACITY; }
        if (last[f].n < 0 || (last[f].n > 0 && (!last[f].has_point || !last[f].outlier || !last[f].Xw || !last[f].keys || !last[f].descriptors))) {
            set_error("tc2li_track_motion_model_batch: last frame %d has null arrays", f);
            return TC2LI_ERR_INVALID;
        }
        F.q_off = total_q; F.n_q = last[f].n;
        total_q += last[f].n;
        memcpy(F.pose7, pose_pred7 + 7 * (size_t)f, 7 * sizeof(float));
        memcpy(F.last_pose7, last[f].pose7, 7 * sizeof(float));
        F.th = th; F.slot = f;
        // forward / backward (ORBmatcher.cc:1703-1708): tlc = Tlw * twc
        float twc[3], tlc[3];
        const float qi[4] = {-F.pose7[0], -F.pose7[1], -F.pose7[2], F.pose7[3]};
        const float nt[3] = {F.pose7[4] * -1.f, F.pose7[5] * -1.f, F.pose7[6] * -1.f};
        quat_rotate_f(qi, nt, twc);
        quat_rotate_f(F.last_pose7, twc, tlc);
        for (int c = 0; c < 3; ++c) tlc[c] += F.last_pose7[4 + c];
        F.forward = tlc[2] > b; F.backward = -tlc[2] > b;
    }
    // ---- one staging block: the last frames' points (structure of arrays) and mvuRight of the current frames ----
    const size_t nq = (size_t)std::max(total_q, 1), nk = (size_t)n_frames * std::max(capacity, 1);
    const size_t o_flags = 0, o_Xw = up256(o_flags + nq), o_ang = up256(o_Xw + 12 * nq), o_oct = up256(o_ang + 4 * nq), o_desc = up256(o_oct + 4 * nq),
                 o_ur = up256(o_desc + 32 * nq), stage_bytes = up256(o_ur + 4 * nk);
    TC2LI_HIP_CHECK(w.h_stage.ensure(stage_bytes)); TC2LI_HIP_CHECK(w.d_stage.ensure(stage_bytes));
    std::vector<int> bad(n_frames, 0);
    tracking_pool().parallel_for(n_frames, [&](int f) {
        const TrackFrameDev& F = w.h_frames.p[f];
        const tc2li_last_frame& lf = last[f];
        uint8_t* hs = w.h_stage.p;
        uint8_t* fl = hs + o_flags + F.q_off;
        float* ang = reinterpret_cast<float*>(hs + o_ang) + F.q_off;
        int32_t* oct = reinterpret_cast<int32_t*>(hs + o_oct) + F.q_off;
        for (int i = 0; i < lf.n; ++i) {
            fl[i] = (uint8_t)((lf.has_point[i] ? 1 : 0) | (lf.outlier[i] ? 2 : 0));
            ang[i] = lf.keys[i].angle;
            oct[i] = lf.keys[i].octave;
            if (fl[i] == 1 && (oct[i] < 0 || oct[i] >= L)) bad[f] = 1;
        }
        if (lf.n) {
            memcpy(hs + o_Xw + 12 * (size_t)F.q_off, lf.Xw, 12 * (size_t)lf.n);
            memcpy(hs + o_desc + 32 * (size_t)F.q_off, lf.descriptors, 32 * (size_t)lf.n);
        }
        memcpy(hs + o_ur + 4 * (size_t)f * capacity, u_right + (size_t)f * capacity, 4 * (size_t)F.n_keys);
    });
    for (int f = 0; f < n_frames; ++f) if (bad[f]) { set_error("octave out of range"); return TC2LI_ERR_INVALID; }
    tm[0] = now() - t0; t0 = now();
    TC2LI_HIP_CHECK(hipMemcpyAsync(w.d_stage.p, w.h_stage.p, stage_bytes, hipMemcpyHostToDevice, st));
    TC2LI_HIP_CHECK(hipMemcpyAsync(w.d_frames.p, w.h_frames.p, n_frames * sizeof(TrackFrameDev), hipMemcpyHostToDevice, st));
    const uint8_t* ds = w.d_stage.p;
    const LastFrameArrays A{ds + o_flags, reinterpret_cast<const float*>(ds + o_Xw), reinterpret_cast<const float*>(ds + o_ang),
                            reinterpret_cast<const int32_t*>(ds + o_oct), ds + o_desc};
    const float* d_ur = reinterpret_cast<const float*>(ds + o_ur);
    TC2LI_HIP_CHECK(w.d_queries.ensure(nq)); TC2LI_HIP_CHECK(w.d_query_frame.ensure(nq)); TC2LI_HIP_CHECK(w.d_match.ensure(nq));
    Pass pass{w, o, n_frames, total_q, capacity, d_ur, nullptr, 0, 0.9f, true, st};
    int rc = pass.prepare();
    if (rc != TC2LI_OK) return rc;
    for (int f = 0; f < n_frames; ++f) w.h_pass.p[f] = f;
    if (total_q > 0) {
        launch_track_queries_last(w.d_frames.p, n_frames, C, A, total_q, w.d_queries.p, w.d_query_frame.p, w.d_match.p, st);
        rc = pass.queue(n_frames, true);
        if (rc != TC2LI_OK) return rc;
        TC2LI_HIP_CHECK(stream_wait_blocking(st));
        if (w.h_small.p[1]) {  // candidate pool exhausted (very dense windows)
            rc = reset_matches(w, n_frames, st);
            if (rc == TC2LI_OK) rc = pass.queue(n_frames, false);
            if (rc != TC2LI_OK) return rc;
            TC2LI_HIP_CHECK(stream_wait_blocking(st));
        }
        memcpy(n_matches, w.h_nmatch.p, n_frames * sizeof(int32_t));
    } else {
        for (int f = 0; f < n_frames; ++f) n_matches[f] = 0;
        TC2LI_HIP_CHECK(hipMemsetAsync(w.d_nmatch.p, 0, n_frames * sizeof(int32_t), st));
    }
    tm[1] = now() - t0; t0 = now();
    // fewer than 20 matches: wider window (Tracking.cc:2774-2783), those frames only
    int n_retry = 0;
    for (int f = 0; f < n_frames; ++f) {
        TrackFrameDev& F = w.h_frames.p[f];
        if (n_matches[f] < 20 && F.n_q > 0) { F.slot = n_retry; F.th = 2 * th; w.h_pass.p[n_retry++] = f; }
        else F.slot = -1;
    }
    if (n_retry > 0) {
        TC2LI_HIP_CHECK(hipMemcpyAsync(w.d_frames.p, w.h_frames.p, n_frames * sizeof(TrackFrameDev), hipMemcpyHostToDevice, st));
        launch_track_queries_last(w.d_frames.p, n_frames, C, A, total_q, w.d_queries.p, w.d_query_frame.p, w.d_match.p, st);
        rc = pass.queue(n_retry, true);
        if (rc != TC2LI_OK) return rc;
        TC2LI_HIP_CHECK(stream_wait_blocking(st));
        if (w.h_small.p[1]) {
            rc = reset_matches(w, n_retry, st);
            if (rc == TC2LI_OK) rc = pass.queue(n_retry, false);
            if (rc != TC2LI_OK) return rc;
            TC2LI_HIP_CHECK(stream_wait_blocking(st));
        }
        memcpy(n_matches, w.h_nmatch.p, n_frames * sizeof(int32_t));  // d_nmatch holds every frame: the pass rewrote its own
    }
    tm[2] = now() - t0; t0 = now();
    // ---- Optimizer::PoseOptimization: one edge per keypoint that now holds a map point, in keypoint order ----
    const size_t ne = (size_t)n_frames * std::max(capacity, 1);
    TC2LI_HIP_CHECK(w.d_of_key.ensure(ne)); TC2LI_HIP_CHECK(w.d_probs.ensure(n_frames)); TC2LI_HIP_CHECK(w.d_edges.ensure(ne)); TC2LI_HIP_CHECK(w.d_Xw.ensure(3 * ne));
    TC2LI_HIP_CHECK(w.d_edge_kp.ensure(ne)); TC2LI_HIP_CHECK(w.d_poses.ensure(7 * (size_t)n_frames)); TC2LI_HIP_CHECK(w.d_outlier.ensure(ne));
    TC2LI_HIP_CHECK(w.d_chi2.ensure(ne)); TC2LI_HIP_CHECK(w.d_inliers.ensure(n_frames)); TC2LI_HIP_CHECK(w.d_ninl.ensure(n_frames));
    launch_track_edges_last(w.d_frames.p, n_frames, C, o->d_mkeys.p, d_ur, w.d_match.p, w.d_nmatch.p, A.Xw, w.d_of_key.p, w.d_probs.p, w.d_edges.p, w.d_Xw.p,
                            w.d_edge_kp.p, w.d_poses.p, st);
    CameraD cd;
    memcpy(&cd, cam, sizeof(cd));
    launch_pose_optimization(w.d_probs.p, n_frames, w.d_Xw.p, w.d_edges.p, cd, w.d_poses.p, w.d_outlier.p, w.d_chi2.p, w.d_inliers.p, capacity, st);
    launch_track_finish_last(w.d_frames.p, n_frames, capacity, w.d_nmatch.p, w.d_probs.p, w.d_outlier.p, w.d_edge_kp.p, w.d_inliers.p, w.d_of_key.p, w.d_poses.p,
                             w.d_ninl.p, st);
    TC2LI_HIP_CHECK(hipGetLastError());
    TC2LI_HIP_CHECK(hipMemcpyAsync(map_point_of_keypoint, w.d_of_key.p, ne * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    TC2LI_HIP_CHECK(hipMemcpyAsync(poses7, w.d_poses.p, 7 * (size_t)n_frames * sizeof(double), hipMemcpyDeviceToHost, st));
    TC2LI_HIP_CHECK(hipMemcpyAsync(n_inliers, w.d_ninl.p, n_frames * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    TC2LI_HIP_CHECK(stream_wait_blocking(st));
    tm[3] = now() - t0;
    if (kTiming) fprintf(stderr, "track timing ms: stage %.3f search %.3f retry %.3f edges+pose-opt+results %.3f\n", tm[0], tm[1], tm[2], tm[3]);
    return n_frames;
}

// The data path of Tracking::TrackLocalMap (SF/src/Tracking.cc:3119-3230) for a batch of independent frames: SearchLocalPoints
// (:3232-3294: isInFrustum + SearchByProjection(F, mvpLocalMapPoints, th, far points) with ORBmatcher(0.8)) on the device-resident
// features, then Optimizer::PoseOptimization over every map point the frame holds, and mnMatchesInliers.
// optimise = false: SearchLocalPoints alone (tc2li_search_local_points_batch) -- with the IMU initialised TrackLocalMap hands the correspondences to
// PoseInertialOptimizationLastFrame / LastKeyFrame (Tracking.cc:2857-2878; tc2li_pose_inertial_optimization_batch) instead of PoseOptimization
static int track_local_map_impl(tc2li_orb* o, int n_frames, const tc2li_keypoint* keypoints, const float* u_right, int capacity,
                                const float* poses7, const uint8_t* held, const float* held_Xw, const tc2li_map_point* local_points,
                                const int32_t* local_offsets, const tc2li_camera* cam, float th, int far_points, float th_far_points,
                                double* poses7_out, int32_t* local_of_keypoint, uint8_t* outlier, int32_t* n_matches,
                                int32_t* n_inliers, void* stream_, const bool optimise) {
    if (!o || n_frames < 0 || capacity < 0 || !keypoints || !u_right || !poses7 || !held || !held_Xw || !local_offsets || !cam || !local_of_keypoint || !n_matches ||
        (optimise && (!poses7_out || !outlier || !n_inliers))) {
        set_error("tc2li_track_local_map_batch: invalid argument");
        return TC2LI_ERR_INVALID;
    }
    if (n_frames == 0) return 0;
    if (2 * n_frames > o->last_nimg || !o->last_plain_order) {
        set_error("tc2li_track_local_map_batch: needs the features of a preceding tc2li_orb_extract_batch call with lapping area {0,0} and "
                  "2*n_frames images");
        return TC2LI_ERR_INVALID;
    }
    if (local_offsets[0] != 0) { set_error("tc2li_track_local_map_batch: local_offsets[0] must be 0"); return TC2LI_ERR_INVALID; }
    const int total_q = local_offsets[n_frames];
    if (total_q < 0 || (total_q > 0 && !local_points)) { set_error("tc2li_track_local_map_batch: invalid local points"); return TC2LI_ERR_INVALID; }
    if (!device_ready()) return TC2LI_ERR_NO_DEVICE;
    static_assert(sizeof(tc2li_map_point) == sizeof(LocalPointDev), "ABI layout");
    hipStream_t st = (hipStream_t)stream_;
    static const bool kTiming = getenv("TC2LI_TRACK_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tm[5] = {0, 0, 0, 0, 0}, t0 = now();
    TrackWs& w = tws();
    TrackConst C;
    fill_const(o, cam, 0.f, capacity, C);
    C.far_points = far_points; C.th_far = th_far_points; C.view_cos_limit = 0.5f;

    TC2LI_HIP_CHECK(w.h_frames.ensure(n_frames)); TC2LI_HIP_CHECK(w.d_frames.ensure(n_frames));
    for (int f = 0; f < n_frames; ++f) {
        TrackFrameDev& F = w.h_frames.p[f];
        memset(&F, 0, sizeof(F));
        F.key_off = o->last_kp_off[2 * f];
        F.n_keys = o->last_kp_cnt[2 * f];
        if (F.n_keys > capacity) { set_error("capacity %d < %d keypoints", capacity, F.n_keys); return TC2LI_ERR_CAPACITY; }
        if (F.n_keys > kMaxMatchKeys) { set_error("frame has %d keypoints, the matcher supports %d", F.n_keys, kMaxMatchKeys); return TC2LI_ERR_CAPACITY; }
        if (local_offsets[f + 1] < local_offsets[f]) { set_error("tc2li_track_local_map_batch: local_offsets must not decrease"); return TC2LI_ERR_INVALID; }
        F.q_off = local_offsets[f]; F.n_q = local_offsets[f + 1] - local_offsets[f];
        memcpy(F.pose7, poses7 + 7 * (size_t)f, 7 * sizeof(float));
        F.th = th; F.slot = f;
    }
    // ---- one staging block: the local points, mvuRight, the held flags and positions ----
    const size_t nq = (size_t)std::max(total_q, 1), nk = (size_t)n_frames * std::max(capacity, 1);
    const size_t o_pts = 0, o_ur = up256(o_pts + sizeof(LocalPointDev) * nq), o_held = up256(o_ur + 4 * nk), o_hx = up256(o_held + nk),
                 stage_bytes = up256(o_hx + 12 * nk);
    TC2LI_HIP_CHECK(w.h_stage.ensure(stage_bytes)); TC2LI_HIP_CHECK(w.d_stage.ensure(stage_bytes));
    {
        // the callers' arrays are contiguous: copied in 1 MiB pieces by the pool
        struct Piece { uint8_t* dst; const uint8_t* src; size_t n; };
        std::vector<Piece> pieces;
        auto add = [&](size_t off, const void* src, size_t bytes) {
            for (size_t at = 0; at < bytes; at += (size_t)1 << 20)
                pieces.push_back(Piece{w.h_stage.p + off + at, static_cast<const uint8_t*>(src) + at, std::min((size_t)1 << 20, bytes - at)});
        };
        if (total_q) add(o_pts, local_points, sizeof(LocalPointDev) * (size_t)total_q);
        add(o_ur, u_right, 4 * nk); add(o_held, held, nk); add(o_hx, held_Xw, 12 * nk);
        tracking_pool().parallel_for((int)pieces.size(), [&](int k) { memcpy(pieces[k].dst, pieces[k].src, pieces[k].n); });
    }
    tm[0] = now() - t0; t0 = now();
    TC2LI_HIP_CHECK(hipMemcpyAsync(w.d_stage.p, w.h_stage.p, stage_bytes, hipMemcpyHostToDevice, st));
    TC2LI_HIP_CHECK(hipMemcpyAsync(w.d_frames.p, w.h_frames.p, n_frames * sizeof(TrackFrameDev), hipMemcpyHostToDevice, st));
    const uint8_t* ds = w.d_stage.p;
    const LocalPointDev* d_pts = reinterpret_cast<const LocalPointDev*>(ds + o_pts);
    const float* d_ur = reinterpret_cast<const float*>(ds + o_ur);
    const uint8_t* d_held = ds + o_held;
    const float* d_hx = reinterpret_cast<const float*>(ds + o_hx);
    TC2LI_HIP_CHECK(w.d_queries.ensure(nq)); TC2LI_HIP_CHECK(w.d_query_frame.ensure(nq)); TC2LI_HIP_CHECK(w.d_match.ensure(nq));
    TC2LI_HIP_CHECK(w.d_occ.ensure(nk)); TC2LI_HIP_CHECK(w.d_amb.ensure(1)); TC2LI_HIP_CHECK(w.h_amb.ensure(1));
    TC2LI_HIP_CHECK(w.d_amb_ids.ensure(nq)); TC2LI_HIP_CHECK(w.d_amb_ratio.ensure(nq)); TC2LI_HIP_CHECK(w.d_amb_r.ensure(nq)); TC2LI_HIP_CHECK(w.d_amb_level.ensure(nq));
    TC2LI_HIP_CHECK(w.h_amb_ratio.ensure(nq)); TC2LI_HIP_CHECK(w.h_amb_level.ensure(nq));
    Pass pass{w, o, n_frames, total_q, capacity, d_ur, w.d_occ.p, 1, 0.8f, false, st};
    int rc = pass.prepare();
    if (rc != TC2LI_OK) return rc;
    for (int f = 0; f < n_frames; ++f) w.h_pass.p[f] = f;
    launch_track_occupied(d_held, nk, w.d_occ.p, st);
    if (total_q > 0) {
        TC2LI_HIP_CHECK(hipMemsetAsync(w.d_amb.p, 0, sizeof(int32_t), st));
        launch_track_queries_local(w.d_frames.p, n_frames, C, d_pts, total_q, w.d_queries.p, w.d_query_frame.p, w.d_match.p, w.d_amb.p, w.d_amb_ids.p, w.d_amb_ratio.p,
                                   w.d_amb_r.p, st);
        TC2LI_HIP_CHECK(hipGetLastError());
        // MapPoint::PredictScale on a level boundary: the host's logf decides (see k_track_queries_local).  One count comes back; the
        // listed ratios get their level here and a patch kernel writes level and window before the search starts.
        TC2LI_HIP_CHECK(hipMemcpyAsync(w.h_amb.p, w.d_amb.p, sizeof(int32_t), hipMemcpyDeviceToHost, st));
        TC2LI_HIP_CHECK(stream_wait_blocking(st));
        const int n_amb = w.h_amb.p[0];
        if (n_amb > 0) {
            TC2LI_HIP_CHECK(hipMemcpyAsync(w.h_amb_ratio.p, w.d_amb_ratio.p, n_amb * sizeof(float), hipMemcpyDeviceToHost, st));
            TC2LI_HIP_CHECK(stream_wait_blocking(st));
            const float ls = C.log_scale;
            const int nl = C.n_levels;
            const int chunk = 16384, n_chunks = (n_amb + chunk - 1) / chunk;
            tracking_pool().parallel_for(n_chunks, [&](int c) {
                const int k1 = std::min(n_amb, (c + 1) * chunk);
                for (int k = c * chunk; k < k1; ++k) {
                    int level = (int)ceilf(logf(w.h_amb_ratio.p[k]) / ls);  // as tc2li_project_local_map / MapPoint::PredictScale
                    if (level < 0) level = 0; else if (level >= nl) level = nl - 1;
                    w.h_amb_level.p[k] = level;
                }
            });
            TC2LI_HIP_CHECK(hipMemcpyAsync(w.d_amb_level.p, w.h_amb_level.p, n_amb * sizeof(int32_t), hipMemcpyHostToDevice, st));
            launch_track_patch_levels(w.d_amb_ids.p, w.d_amb_level.p, w.d_amb_r.p, n_amb, C, w.d_queries.p, st);
        }
        tm[1] = now() - t0; t0 = now();
        rc = pass.queue(n_frames, true);
        if (rc != TC2LI_OK) return rc;
        TC2LI_HIP_CHECK(stream_wait_blocking(st));
        if (w.h_small.p[1]) {
            rc = reset_matches(w, n_frames, st);
            if (rc == TC2LI_OK) rc = pass.queue(n_frames, false);
            if (rc != TC2LI_OK) return rc;
            TC2LI_HIP_CHECK(stream_wait_blocking(st));
        }
        memcpy(n_matches, w.h_nmatch.p, n_frames * sizeof(int32_t));
    } else {
        for (int f = 0; f < n_frames; ++f) n_matches[f] = 0;
    }
    tm[2] = now() - t0; t0 = now();
    // ---- Optimizer::PoseOptimization over every map point the frame now holds, in keypoint order ----
    const size_t ne = nk;
    TC2LI_HIP_CHECK(w.d_of_key.ensure(ne)); TC2LI_HIP_CHECK(w.d_probs.ensure(n_frames)); TC2LI_HIP_CHECK(w.d_edges.ensure(ne)); TC2LI_HIP_CHECK(w.d_Xw.ensure(3 * ne));
    TC2LI_HIP_CHECK(w.d_edge_kp.ensure(ne)); TC2LI_HIP_CHECK(w.d_poses.ensure(7 * (size_t)n_frames)); TC2LI_HIP_CHECK(w.d_outlier.ensure(ne));
    TC2LI_HIP_CHECK(w.d_chi2.ensure(ne)); TC2LI_HIP_CHECK(w.d_inliers.ensure(n_frames)); TC2LI_HIP_CHECK(w.d_ninl.ensure(n_frames));
    TC2LI_HIP_CHECK(w.d_outlier_key.ensure(ne));
    launch_track_edges_local(w.d_frames.p, n_frames, C, o->d_mkeys.p, d_ur, w.d_match.p, d_held, d_hx, d_pts, w.d_of_key.p, w.d_probs.p, w.d_edges.p, w.d_Xw.p,
                             w.d_edge_kp.p, w.d_poses.p, st);
    if (!optimise) {  // the keypoints' new map points are what the caller wants; the optimisation is the pose-inertial one, elsewhere
        TC2LI_HIP_CHECK(hipGetLastError());
        TC2LI_HIP_CHECK(hipMemcpyAsync(local_of_keypoint, w.d_of_key.p, ne * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        TC2LI_HIP_CHECK(stream_wait_blocking(st));
        tm[3] = now() - t0;
        if (kTiming) fprintf(stderr, "search-local-points timing ms: stage %.3f queries %.3f search %.3f results %.3f\n", tm[0], tm[1], tm[2], tm[3]);
        return n_frames;
    }
    CameraD cd;
    memcpy(&cd, cam, sizeof(cd));
    launch_pose_optimization(w.d_probs.p, n_frames, w.d_Xw.p, w.d_edges.p, cd, w.d_poses.p, w.d_outlier.p, w.d_chi2.p, w.d_inliers.p, capacity, st);
    launch_track_finish_local(w.d_frames.p, n_frames, capacity, w.d_probs.p, w.d_outlier.p, w.d_edge_kp.p, d_held, w.d_of_key.p, w.d_outlier_key.p, w.d_ninl.p, st);
    TC2LI_HIP_CHECK(hipGetLastError());
    TC2LI_HIP_CHECK(hipMemcpyAsync(local_of_keypoint, w.d_of_key.p, ne * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    TC2LI_HIP_CHECK(hipMemcpyAsync(outlier, w.d_outlier_key.p, ne, hipMemcpyDeviceToHost, st));
    TC2LI_HIP_CHECK(hipMemcpyAsync(poses7_out, w.d_poses.p, 7 * (size_t)n_frames * sizeof(double), hipMemcpyDeviceToHost, st));
    TC2LI_HIP_CHECK(hipMemcpyAsync(n_inliers, w.d_ninl.p, n_frames * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    TC2LI_HIP_CHECK(stream_wait_blocking(st));
    tm[3] = now() - t0;
    if (kTiming) fprintf(stderr, "track-local-map timing ms: stage %.3f queries %.3f search %.3f edges+pose-opt+results %.3f\n", tm[0], tm[1], tm[2], tm[3]);
    return n_frames;
}

extern "C" int tc2li_track_local_map_batch(tc2li_orb* o, int n_frames, const tc2li_keypoint* keypoints, const float* u_right, int capacity,
                                           const float* poses7, const uint8_t* held, const float* held_Xw, const tc2li_map_point* local_points,
                                           const int32_t* local_offsets, const tc2li_camera* cam, float th, int far_points, float th_far_points,
                                           double* poses7_out, int32_t* local_of_keypoint, uint8_t* outlier, int32_t* n_matches,
                                           int32_t* n_inliers, void* stream_) {
    return track_local_map_impl(o, n_frames, keypoints, u_right, capacity, poses7, held, held_Xw, local_points, local_offsets, cam, th, far_points,
                                th_far_points, poses7_out, local_of_keypoint, outlier, n_matches, n_inliers, stream_, true);
}

// Tracking::SearchLocalPoints (SF/src/Tracking.cc:3232-3294) for a batch of frames: the first half of tc2li_track_local_map_batch, for the
// camera-LiDAR-inertial configuration's TrackLocalMap (the optimiser there is PoseInertialOptimization*)
extern "C" int tc2li_search_local_points_batch(tc2li_orb* o, int n_frames, const tc2li_keypoint* keypoints, const float* u_right, int capacity,
                                               const float* poses7, const uint8_t* held, const float* held_Xw, const tc2li_map_point* local_points,
                                               const int32_t* local_offsets, const tc2li_camera* cam, float th, int far_points, float th_far_points,
                                               int32_t* local_of_keypoint, int32_t* n_matches, void* stream_) {
    return track_local_map_impl(o, n_frames, keypoints, u_right, capacity, poses7, held, held_Xw, local_points, local_offsets, cam, th, far_points,
                                th_far_points, nullptr, local_of_keypoint, nullptr, n_matches, nullptr, stream_, false);
}
