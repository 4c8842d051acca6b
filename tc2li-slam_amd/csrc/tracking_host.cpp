// Host side of tc2li_track_motion_model_batch (include/tc2li_hip.h): the data path of Tracking::TrackWithMotionModel
// (SF/src/Tracking.cc:2737-2834) for a batch of independent frames.  The stages are the library's own entry points:
// query construction (tc2li_project_last_frame), the fixed-point greedy matcher on the device-resident features,
// the pose-only optimisation kernel; this file only sequences them and does the per-frame bookkeeping the reference
// does between them (edge construction of Optimizer::PoseOptimization :858-990, outlier discarding :2798-2822).
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "common.hpp"
#include "matcher_host.hpp"
#include "orb_handle.hpp"

using namespace tc2li;

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
    const int L = o->prm.nlevels;
    const float cam4[4] = {(float)cam->fx, (float)cam->fy, (float)cam->cx, (float)cam->cy};
    const float bf = (float)cam->bf;

    std::vector<BatchSearchFrame> frames(n_frames);
    int total_q = 0;
    for (int f = 0; f < n_frames; ++f) {
        BatchSearchFrame& fr = frames[f];
        fr.key_off = o->last_kp_off[2 * f];
        fr.n_keys = o->last_kp_cnt[2 * f];
        if (fr.n_keys > capacity) { set_error("capacity %d < %d keypoints", capacity, fr.n_keys); return TC2LI_ERR_CAPACITY; }
        fr.keys_host = keypoints + (size_t)(2 * f) * capacity;
        fr.u_right_host = u_right + (size_t)f * capacity;
        if (last[f].n < 0 || (last[f].n > 0 && (!last[f].has_point || !last[f].outlier || !last[f].Xw || !last[f].keys || !last[f].descriptors))) {
            set_error("tc2li_track_motion_model_batch: last frame %d has null arrays", f);
            return TC2LI_ERR_INVALID;
        }
        fr.q_off = total_q;
        fr.n_q = last[f].n;
        total_q += last[f].n;
    }
    std::vector<tc2li_proj_query> queries(std::max(total_q, 1));
    std::vector<int32_t> match(std::max(total_q, 1), -1);
    std::vector<int> rc(n_frames, 0);
    auto build_queries = [&](int f, float radius) {
        const tc2li_last_frame& lf = last[f];
        rc[f] = tc2li_project_last_frame(pose_pred7 + 7 * f, lf.pose7, cam4, b, bf, o->scale.data(), L, o->cur_w, o->cur_h, lf.n, lf.has_point,
                                         lf.outlier, lf.Xw, lf.keys, lf.descriptors, radius, 0, queries.data() + frames[f].q_off);
    };
    tracking_pool().parallel_for(n_frames, [&](int f) { build_queries(f, th); });
    for (int f = 0; f < n_frames; ++f) if (rc[f] < 0) return rc[f];
    tm[0] = now() - t0; t0 = now();
    int r = search_batch_device(o, frames.data(), n_frames, queries.data(), 0, 0.9f, true, match.data(), n_matches, st);
    if (r < 0) return r;
    tm[1] = now() - t0; t0 = now();
    // fewer than 20 matches: wider window (Tracking.cc:2774-2783)
    std::vector<int> retry;
    for (int f = 0; f < n_frames; ++f) if (n_matches[f] < 20) retry.push_back(f);
    if (!retry.empty()) {
        std::vector<BatchSearchFrame> again(retry.size());
        std::vector<int32_t> nm(retry.size());
        tracking_pool().parallel_for((int)retry.size(), [&](int k) { build_queries(retry[k], 2 * th); });
        for (size_t k = 0; k < retry.size(); ++k) { if (rc[retry[k]] < 0) return rc[retry[k]]; again[k] = frames[retry[k]]; }
        r = search_batch_device(o, again.data(), (int)again.size(), queries.data(), 0, 0.9f, true, match.data(), nm.data(), st);
        if (r < 0) return r;
        for (size_t k = 0; k < retry.size(); ++k) n_matches[retry[k]] = nm[k];
    }
    // ---- Optimizer::PoseOptimization: one edge per keypoint that now holds a map point, in keypoint order ----
    std::vector<int32_t> edge_off(n_frames + 1, 0);
    for (int f = 0; f < n_frames; ++f) edge_off[f + 1] = edge_off[f] + (n_matches[f] >= 20 ? n_matches[f] : 0);
    const int total_e = edge_off[n_frames];
    std::vector<double> Xw(3 * (size_t)std::max(total_e, 1));
    std::vector<tc2li_ba_edge> edges(std::max(total_e, 1));
    std::vector<int32_t> edge_kp(std::max(total_e, 1));
    std::vector<uint8_t> outlier(std::max(total_e, 1), 0);
    tracking_pool().parallel_for(n_frames, [&](int f) {
        const BatchSearchFrame& fr = frames[f];
        int32_t* mp = map_point_of_keypoint + (size_t)f * capacity;
        for (int i = 0; i < capacity; ++i) mp[i] = -1;
        const int32_t* m = match.data() + fr.q_off;
        for (int q = 0; q < fr.n_q; ++q) if (m[q] >= 0) mp[m[q]] = q;
        for (int c = 0; c < 7; ++c) poses7[7 * f + c] = (double)pose_pred7[7 * f + c];
        if (n_matches[f] < 20) return;
        int e = edge_off[f];
        for (int i = 0; i < fr.n_keys; ++i) {
            const int q = mp[i];
            if (q < 0) continue;
            const tc2li_keypoint& kp = fr.keys_host[i];
            tc2li_ba_edge& ed = edges[e];
            ed.point = e - edge_off[f]; ed.pose = 0;
            ed.u = kp.x; ed.v = kp.y; ed.u_right = fr.u_right_host[i];
            ed.inv_sigma2 = o->inv_sigma2[kp.octave];
            for (int c = 0; c < 3; ++c) Xw[3 * (size_t)e + c] = (double)last[f].Xw[3 * (size_t)q + c];
            edge_kp[e] = i;
            ++e;
        }
    });
    tm[2] = now() - t0; t0 = now();
    std::vector<int32_t> inl(n_frames, 0);
    r = tc2li_pose_optimization_batch(n_frames, poses7, edge_off.data(), Xw.data(), edges.data(), cam, outlier.data(), inl.data(), stream_);
    if (r < 0) return r;
    tm[3] = now() - t0; t0 = now();
    for (int f = 0; f < n_frames; ++f) {
        if (n_matches[f] < 20) { n_inliers[f] = -1; for (int c = 0; c < 7; ++c) poses7[7 * f + c] = (double)pose_pred7[7 * f + c]; continue; }
        n_inliers[f] = inl[f];
        int32_t* mp = map_point_of_keypoint + (size_t)f * capacity;
        for (int e = edge_off[f]; e < edge_off[f + 1]; ++e) if (outlier[e]) mp[edge_kp[e]] = -1;  // Tracking.cc:2804-2818
    }
    if (kTiming) fprintf(stderr, "track timing ms: queries %.3f search %.3f edges %.3f pose-opt %.3f finish %.3f\n", tm[0], tm[1], tm[2], tm[3], now() - t0);
    return n_frames;
}

// The data path of Tracking::TrackLocalMap (SF/src/Tracking.cc:3119-3230) for a batch of independent frames: SearchLocalPoints
// (:3232-3294: isInFrustum + SearchByProjection(F, mvpLocalMapPoints, th, far points) with ORBmatcher(0.8)) on the device-resident
// features, then Optimizer::PoseOptimization over every map point the frame holds, and mnMatchesInliers.
extern "C" int tc2li_track_local_map_batch(tc2li_orb* o, int n_frames, const tc2li_keypoint* keypoints, const float* u_right, int capacity,
                                           const float* poses7, const uint8_t* held, const float* held_Xw, const tc2li_map_point* local_points,
                                           const int32_t* local_offsets, const tc2li_camera* cam, float th, int far_points, float th_far_points,
                                           double* poses7_out, int32_t* local_of_keypoint, uint8_t* outlier, int32_t* n_matches,
                                           int32_t* n_inliers, void* stream_) {
    if (!o || n_frames < 0 || capacity < 0 || !keypoints || !u_right || !poses7 || !held || !held_Xw || !local_offsets || !cam || !poses7_out ||
        !local_of_keypoint || !outlier || !n_matches || !n_inliers) {
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
    hipStream_t st = (hipStream_t)stream_;
    const int L = o->prm.nlevels;
    const float cam4[4] = {(float)cam->fx, (float)cam->fy, (float)cam->cx, (float)cam->cy};
    const float bf = (float)cam->bf;
    const float log_scale = std::log(o->prm.scale_factor);  // mfLogScaleFactor = log(mfScaleFactor) (SF/src/Frame.cc:96)
    static const bool kTiming = getenv("TC2LI_TRACK_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tm[5] = {0, 0, 0, 0, 0}, t0 = now();
    std::vector<BatchSearchFrame> frames(n_frames);
    std::vector<std::vector<uint8_t>> occ(n_frames);
    for (int f = 0; f < n_frames; ++f) {
        BatchSearchFrame& fr = frames[f];
        fr.key_off = o->last_kp_off[2 * f];
        fr.n_keys = o->last_kp_cnt[2 * f];
        if (fr.n_keys > capacity) { set_error("capacity %d < %d keypoints", capacity, fr.n_keys); return TC2LI_ERR_CAPACITY; }
        if (local_offsets[f + 1] < local_offsets[f]) { set_error("tc2li_track_local_map_batch: local_offsets must not decrease"); return TC2LI_ERR_INVALID; }
        fr.keys_host = keypoints + (size_t)(2 * f) * capacity;
        fr.u_right_host = u_right + (size_t)f * capacity;
        fr.q_off = local_offsets[f];
        fr.n_q = local_offsets[f + 1] - local_offsets[f];
        occ[f].resize(std::max(fr.n_keys, 1));
        const uint8_t* h = held + (size_t)f * capacity;
        for (int i = 0; i < fr.n_keys; ++i) occ[f][i] = h[i] == 1;  // held with Observations() > 0 (ORBmatcher.cc:100-102)
        fr.occupied_host = occ[f].data();
    }
    std::vector<tc2li_proj_query> queries(std::max(total_q, 1));
    std::vector<int32_t> match(std::max(total_q, 1), -1);
    std::vector<int> rc(n_frames, 0);
    tracking_pool().parallel_for(n_frames, [&](int f) {
        if (frames[f].n_q > 0)
            rc[f] = tc2li_project_local_map(poses7 + 7 * f, cam4, bf, o->scale.data(), L, log_scale, o->cur_w, o->cur_h, frames[f].n_q,
                                            local_points + frames[f].q_off, th, far_points, th_far_points, 0.5f, queries.data() + frames[f].q_off);
    });
    for (int f = 0; f < n_frames; ++f) if (rc[f] < 0) return rc[f];
    tm[0] = now() - t0; t0 = now();
    int r = search_batch_device(o, frames.data(), n_frames, queries.data(), 1, 0.8f, false, match.data(), n_matches, st);
    if (r < 0) return r;
    tm[1] = now() - t0; t0 = now();
    // ---- Optimizer::PoseOptimization over every map point the frame now holds, in keypoint order ----
    std::vector<int32_t> edge_off(n_frames + 1, 0);
    tracking_pool().parallel_for(n_frames, [&](int f) {
        const BatchSearchFrame& fr = frames[f];
        int32_t* lk = local_of_keypoint + (size_t)f * capacity;
        for (int i = 0; i < capacity; ++i) lk[i] = -1;
        const int32_t* m = match.data() + fr.q_off;
        for (int q = 0; q < fr.n_q; ++q) if (m[q] >= 0) lk[m[q]] = q;  // F.mvpMapPoints[bestIdx] = pMP
    });
    for (int f = 0; f < n_frames; ++f) {
        const uint8_t* h = held + (size_t)f * capacity;
        const int32_t* lk = local_of_keypoint + (size_t)f * capacity;
        int ne = 0;
        for (int i = 0; i < frames[f].n_keys; ++i) ne += (h[i] != 0 || lk[i] >= 0) ? 1 : 0;
        edge_off[f + 1] = edge_off[f] + ne;
    }
    const int total_e = edge_off[n_frames];
    std::vector<double> Xw(3 * (size_t)std::max(total_e, 1));
    std::vector<tc2li_ba_edge> edges(std::max(total_e, 1));
    std::vector<int32_t> edge_kp(std::max(total_e, 1));
    std::vector<uint8_t> out(std::max(total_e, 1), 0);
    tracking_pool().parallel_for(n_frames, [&](int f) {
        const BatchSearchFrame& fr = frames[f];
        const uint8_t* h = held + (size_t)f * capacity;
        const float* hx = held_Xw + 3 * (size_t)f * capacity;
        const int32_t* lk = local_of_keypoint + (size_t)f * capacity;
        for (int c = 0; c < 7; ++c) poses7_out[7 * f + c] = (double)poses7[7 * f + c];
        int e = edge_off[f];
        for (int i = 0; i < fr.n_keys; ++i) {
            if (!(h[i] != 0 || lk[i] >= 0)) continue;
            const tc2li_keypoint& kp = fr.keys_host[i];
            tc2li_ba_edge& ed = edges[e];
            ed.point = e - edge_off[f]; ed.pose = 0;
            ed.u = kp.x; ed.v = kp.y; ed.u_right = fr.u_right_host[i];
            ed.inv_sigma2 = o->inv_sigma2[kp.octave];
            const float* X = lk[i] >= 0 ? local_points[fr.q_off + lk[i]].pos : hx + 3 * (size_t)i;
            for (int c = 0; c < 3; ++c) Xw[3 * (size_t)e + c] = (double)X[c];
            edge_kp[e] = i;
            ++e;
        }
    });
    std::vector<int32_t> inl(n_frames, 0);
    tm[2] = now() - t0; t0 = now();
    r = tc2li_pose_optimization_batch(n_frames, poses7_out, edge_off.data(), Xw.data(), edges.data(), cam, out.data(), inl.data(), stream_);
    if (r < 0) return r;
    tm[3] = now() - t0; t0 = now();
    if (kTiming) fprintf(stderr, "track-local-map timing ms: queries %.3f search %.3f edges %.3f pose-opt %.3f\n", tm[0], tm[1], tm[2], tm[3]);
    for (int f = 0; f < n_frames; ++f) {
        uint8_t* ol = outlier + (size_t)f * capacity;
        memset(ol, 0, capacity);
        const uint8_t* h = held + (size_t)f * capacity;
        const int32_t* lk = local_of_keypoint + (size_t)f * capacity;
        int good = 0;
        for (int e = edge_off[f]; e < edge_off[f + 1]; ++e) {
            const int i = edge_kp[e];
            ol[i] = out[e];
            // mnMatchesInliers: not an outlier and Observations() > 0 (held == 2: a point without observations; local points have them)
            if (!out[e] && (lk[i] >= 0 || h[i] == 1)) ++good;
        }
        n_inliers[f] = good;
    }
    return n_frames;
}
