// Host side of tc2li_search_by_projection* (include/tc2li_hip.h): replaces the two tracking overloads of
// ORBmatcher::SearchByProjection (SF/src/ORBmatcher.cc:52-222, 1685-1896).  Query construction (projection of the
// source points, search window, level range) is a few hundred float operations per frame and is done here on the
// host with the reference's arithmetic; the window search + Hamming + greedy assignment runs in k_match_by_projection;
// the rotation-histogram filter (:1858-1881, ComputeThreeMaxima :2021-2062) is applied to the result on the host.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "common.hpp"
#include "matcher_device.hpp"
#include "orb_handle.hpp"

using namespace tc2li;

static_assert(sizeof(tc2li_proj_query) == sizeof(MatchQuery), "ABI layout");

namespace {

struct MatcherWorkspace {
    DevBuf<MatchFrameDev> d_frames;
    DevBuf<MatchKey> d_keys;
    DevBuf<uint8_t> d_desc, d_occ;
    DevBuf<float> d_ur;
    DevBuf<MatchQuery> d_queries;
    DevBuf<int32_t> d_match, d_prev, d_rounds;
    PinnedBuf<float> h_ur;
    PinnedBuf<MatchFrameDev> h_frames;
    PinnedBuf<int32_t> h_match;
    // candidate-list form of the batched search
    DevBuf<int32_t> d_cell_start, d_key_base, d_cand_off, d_cand_cnt, d_pool_top, d_query_frame;
    DevBuf<uint16_t> d_items;
    DevBuf<uint32_t> d_pool;
    PinnedBuf<int32_t> h_query_frame, h_key_base, h_pool_top;
    std::mutex mu;
};
// one workspace per host thread: two threads that search different batches side by side do not wait for each other
MatcherWorkspace& mws() { static thread_local MatcherWorkspace w; return w; }

inline void quat_rotate_f(const float q[4], const float v[3], float out[3]) {  // Eigen::Quaternionf::_transformVector
    float uv[3] = {q[1] * v[2] - q[2] * v[1], q[2] * v[0] - q[0] * v[2], q[0] * v[1] - q[1] * v[0]};
    uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
    out[0] = v[0] + q[3] * uv[0] + (q[1] * uv[2] - q[2] * uv[1]);
    out[1] = v[1] + q[3] * uv[1] + (q[2] * uv[0] - q[0] * uv[2]);
    out[2] = v[2] + q[3] * uv[2] + (q[0] * uv[1] - q[1] * uv[0]);
}
inline void camera_centre(const float pose7[7], float out[3]) {  // Tcw.inverse().translation()
    const float qi[4] = {-pose7[0], -pose7[1], -pose7[2], pose7[3]};
    const float nt[3] = {pose7[4] * -1.f, pose7[5] * -1.f, pose7[6] * -1.f};
    quat_rotate_f(qi, nt, out);
}

// ORBmatcher::ComputeThreeMaxima (SF/src/ORBmatcher.cc:2021-2062) on bin populations
void three_maxima(const int* count, int L, int& ind1, int& ind2, int& ind3) {
    int max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < L; i++) {
        const int s = count[i];
        if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
        else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
        else if (s > max3) { max3 = s; ind3 = i; }
    }
    if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
    else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
}

// Rotation consistency of the last-frame overload (SF/src/ORBmatcher.cc:1783-1799, 1858-1881): matches outside the three
// dominant 30-bin rotation bins are removed; returns how many.
int rotation_filter(const tc2li_proj_query* queries, int M, const tc2li_keypoint* keys, int32_t* match_of_query) {
    const int L = 30;
    const float factor = 1.0f / L;
    std::vector<int> bin_of(M, -1);
    int count[30] = {0};
    for (int q = 0; q < M; ++q) {
        if (match_of_query[q] < 0) continue;
        float rot = queries[q].angle - keys[match_of_query[q]].angle;
        if (rot < 0.0) rot += 360.0f;
        int bin = (int)roundf(rot * factor);
        if (bin == L) bin = 0;
        bin_of[q] = bin;
        count[bin]++;
    }
    int ind1 = -1, ind2 = -1, ind3 = -1, removed = 0;
    three_maxima(count, L, ind1, ind2, ind3);
    for (int q = 0; q < M; ++q)
        if (bin_of[q] >= 0 && bin_of[q] != ind1 && bin_of[q] != ind2 && bin_of[q] != ind3) { match_of_query[q] = -1; removed++; }
    return removed;
}

}  // namespace

extern "C" {

int tc2li_search_by_projection(const tc2li_frame_view* frame, const tc2li_proj_query* queries, int n_queries, int mode,
                               float nn_ratio, int check_orientation, int32_t* match_of_query, int32_t* query_of_keypoint) {
    if (!frame || n_queries < 0 || (n_queries > 0 && (!queries || !match_of_query)) || frame->n < 0 || (mode != 0 && mode != 1) ||
        !(frame->max_x > frame->min_x) || !(frame->max_y > frame->min_y)) {
        set_error("tc2li_search_by_projection: invalid argument");
        return TC2LI_ERR_INVALID;
    }
    const int N = frame->n, M = n_queries;
    if (N > kMaxMatchKeys) { set_error("frame has %d keypoints, the matcher supports %d", N, kMaxMatchKeys); return TC2LI_ERR_CAPACITY; }
    if (query_of_keypoint) for (int i = 0; i < N; ++i) query_of_keypoint[i] = -1;
    if (M == 0) return 0;
    if (N == 0) { for (int q = 0; q < M; ++q) match_of_query[q] = -1; return 0; }
    if (!frame->keys || !frame->descriptors || !frame->u_right) { set_error("tc2li_search_by_projection: null frame arrays"); return TC2LI_ERR_INVALID; }
    if (!device_ready()) return TC2LI_ERR_NO_DEVICE;
    MatcherWorkspace& w = mws();
    std::lock_guard<std::mutex> lk(w.mu);
    hipStream_t ps = private_stream();
    std::vector<MatchKey> keys(N);
    for (int i = 0; i < N; ++i) keys[i] = MatchKey{frame->keys[i].x, frame->keys[i].y, frame->keys[i].octave};
    TC2LI_HIP_CHECK(w.d_frames.ensure(1)); TC2LI_HIP_CHECK(w.d_keys.ensure(N)); TC2LI_HIP_CHECK(w.d_desc.ensure((size_t)N * 32));
    TC2LI_HIP_CHECK(w.d_occ.ensure(N)); TC2LI_HIP_CHECK(w.d_ur.ensure(N)); TC2LI_HIP_CHECK(w.d_queries.ensure(M));
    TC2LI_HIP_CHECK(w.d_match.ensure(M)); TC2LI_HIP_CHECK(w.d_prev.ensure(M)); TC2LI_HIP_CHECK(w.d_rounds.ensure(1));
    TC2LI_HIP_CHECK(copy_sync(w.d_keys.p, keys.data(), N * sizeof(MatchKey), hipMemcpyHostToDevice, ps));
    TC2LI_HIP_CHECK(copy_sync(w.d_desc.p, frame->descriptors, (size_t)N * 32, hipMemcpyHostToDevice, ps));
    TC2LI_HIP_CHECK(copy_sync(w.d_ur.p, frame->u_right, N * sizeof(float), hipMemcpyHostToDevice, ps));
    if (frame->occupied) TC2LI_HIP_CHECK(copy_sync(w.d_occ.p, frame->occupied, N, hipMemcpyHostToDevice, ps));
    else TC2LI_HIP_CHECK(memset_sync(w.d_occ.p, 0, N, ps));
    TC2LI_HIP_CHECK(copy_sync(w.d_queries.p, queries, (size_t)M * sizeof(MatchQuery), hipMemcpyHostToDevice, ps));
    MatchFrameDev fd{w.d_keys.p, w.d_desc.p, w.d_ur.p, w.d_occ.p, w.d_queries.p, N, M, 0, 0, frame->min_x, frame->max_x, frame->min_y, frame->max_y};
    TC2LI_HIP_CHECK(copy_sync(w.d_frames.p, &fd, sizeof(fd), hipMemcpyHostToDevice, ps));
    launch_match_by_projection(w.d_frames.p, 1, mode, nn_ratio, w.d_match.p, w.d_prev.p, w.d_rounds.p, ps);
    TC2LI_HIP_CHECK(hipGetLastError());
    TC2LI_HIP_CHECK(copy_sync(match_of_query, w.d_match.p, M * sizeof(int32_t), hipMemcpyDeviceToHost, ps));
    int nmatches = 0;
    for (int q = 0; q < M; ++q) nmatches += match_of_query[q] >= 0;
    if (check_orientation) nmatches -= rotation_filter(queries, M, frame->keys, match_of_query);
    if (query_of_keypoint)
        for (int q = 0; q < M; ++q) if (match_of_query[q] >= 0) query_of_keypoint[match_of_query[q]] = q;
    return nmatches;
}

int tc2li_project_last_frame(const float pose_cur7[7], const float pose_last7[7], const float cam4[4], float b, float bf,
                             const float* scale_factors, int n_levels, int cols, int rows, int n, const uint8_t* has_point,
                             const uint8_t* outlier, const float* Xw, const tc2li_keypoint* last_keys, const uint8_t* mp_descriptors,
                             float th, int mono, tc2li_proj_query* queries) {
    if (!pose_cur7 || !pose_last7 || !cam4 || !scale_factors || n < 0 || (n > 0 && (!has_point || !outlier || !Xw || !last_keys || !mp_descriptors || !queries))) {
        set_error("tc2li_project_last_frame: invalid argument");
        return TC2LI_ERR_INVALID;
    }
    float twc[3], tlc[3];
    camera_centre(pose_cur7, twc);
    quat_rotate_f(pose_last7, twc, tlc);
    for (int c = 0; c < 3; ++c) tlc[c] += pose_last7[4 + c];
    const bool forward = tlc[2] > b && !mono, backward = -tlc[2] > b && !mono;
    int valid = 0;
    for (int i = 0; i < n; ++i) {
        tc2li_proj_query& Q = queries[i];
        memset(&Q, 0, sizeof(Q));
        Q.u_right = -1; Q.min_level = -1; Q.max_level = -1; Q.has_observations = 1;
        if (!has_point[i] || outlier[i]) continue;
        const int oct = last_keys[i].octave;
        if (oct < 0 || oct >= n_levels) { set_error("octave out of range"); return TC2LI_ERR_INVALID; }
        float pc[3];
        quat_rotate_f(pose_cur7, Xw + 3 * (size_t)i, pc);
        for (int c = 0; c < 3; ++c) pc[c] += pose_cur7[4 + c];
        const float invzc = (float)(1.0 / pc[2]);
        if (invzc < 0) continue;
        const float u = cam4[0] * pc[0] / pc[2] + cam4[2], v = cam4[1] * pc[1] / pc[2] + cam4[3];
        if (u < 0.f || u > (float)cols) continue;
        if (v < 0.f || v > (float)rows) continue;
        Q.radius = th * scale_factors[oct];
        if (forward) { Q.min_level = oct; Q.max_level = -1; }
        else if (backward) { Q.min_level = 0; Q.max_level = oct; }
        else { Q.min_level = oct - 1; Q.max_level = oct + 1; }
        Q.u = u; Q.v = v;
        Q.u_right = u - bf * invzc;
        Q.angle = last_keys[i].angle;
        Q.valid = 1;
        memcpy(Q.descriptor, mp_descriptors + (size_t)i * 32, 32);
        ++valid;
    }
    return valid;
}

int tc2li_project_local_map(const float pose7[7], const float cam4[4], float bf, const float* scale_factors, int n_levels,
                            float log_scale_factor, int cols, int rows, int n, const tc2li_map_point* points, float th,
                            int far_points, float th_far_points, float viewing_cos_limit, tc2li_proj_query* queries) {
    if (!pose7 || !cam4 || !scale_factors || n < 0 || (n > 0 && (!points || !queries)) || n_levels < 1) {
        set_error("tc2li_project_local_map: invalid argument");
        return TC2LI_ERR_INVALID;
    }
    float Ow[3], R[9];
    camera_centre(pose7, Ow);
    {  // Eigen::Quaternionf::toRotationMatrix (mRcw of Frame::UpdatePoseMatrices)
        const float* q = pose7;
        const float tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
        const float twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
        const float txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
        const float tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
        R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
        R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
        R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
    }
    const bool factor = th != 1.0;
    int valid = 0;
    for (int k = 0; k < n; ++k) {
        const tc2li_map_point& M = points[k];
        tc2li_proj_query& Q = queries[k];
        memset(&Q, 0, sizeof(Q));
        Q.u_right = -1; Q.min_level = -1; Q.max_level = -1; Q.has_observations = 1;
        float pc[3];
        for (int r = 0; r < 3; ++r) pc[r] = ((R[3 * r] * M.pos[0] + R[3 * r + 1] * M.pos[1]) + R[3 * r + 2] * M.pos[2]) + pose7[4 + r];
        const float pc_dist = sqrtf((pc[0] * pc[0] + pc[1] * pc[1]) + pc[2] * pc[2]);
        const float invz = 1.0f / pc[2];
        if (pc[2] < 0.0f) continue;
        const float u = cam4[0] * pc[0] / pc[2] + cam4[2], v = cam4[1] * pc[1] / pc[2] + cam4[3];
        if (u < 0.f || u > (float)cols) continue;
        if (v < 0.f || v > (float)rows) continue;
        const float po[3] = {M.pos[0] - Ow[0], M.pos[1] - Ow[1], M.pos[2] - Ow[2]};
        const float dist = sqrtf((po[0] * po[0] + po[1] * po[1]) + po[2] * po[2]);
        if (dist < M.min_distance || dist > M.max_distance) continue;
        const float view_cos = ((po[0] * M.normal[0] + po[1] * M.normal[1]) + po[2] * M.normal[2]) / dist;
        if (view_cos < viewing_cos_limit) continue;
        const float ratio = M.max_distance_raw / dist;  // MapPoint::PredictScale
        int level = (int)ceilf(logf(ratio) / log_scale_factor);
        if (level < 0) level = 0; else if (level >= n_levels) level = n_levels - 1;
        if (far_points && pc_dist > th_far_points) continue;
        float r = view_cos > 0.998f ? 2.5f : 4.0f;
        if (factor) r *= th;
        Q.u = u; Q.v = v;
        Q.u_right = u - bf * invz;
        Q.radius = r * scale_factors[level];
        Q.min_level = level - 1; Q.max_level = level;
        Q.valid = 1;
        memcpy(Q.descriptor, M.descriptor, 32);
        ++valid;
    }
    return valid;
}

}  // extern "C"
