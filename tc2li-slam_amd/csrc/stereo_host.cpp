// Host side of the stereo matcher behind tc2li_stereo_match*: replaces Frame::ComputeStereoMatches
// (SF/src/Frame.cc:841-1011).  The per-keypoint search and refinement run in k_stereo_match; the closing
// median cut over the accepted matches (:997-1010) is a few thousand integers per frame and stays on the host.
#include <algorithm>
#include <cstring>

#include "common.hpp"
#include "orb_handle.hpp"

using namespace tc2li;

namespace {

struct StereoWorkspace {
    DevBuf<StereoFrame> d_frames;
    DevBuf<MatchKey> d_keys;
    DevBuf<uint8_t> d_desc;
    PinnedBuf<float> h_u, h_d;
    PinnedBuf<int> h_sad;
    DevBuf<int32_t> d_row_start;  // the matcher's row lists (k_stereo_rows)
    DevBuf<uint16_t> d_entries;
};

// One workspace per left handle, kept in a side table so that the handle struct stays ORB-only.
std::mutex g_ws_mu;
std::vector<std::pair<const tc2li_orb*, std::unique_ptr<StereoWorkspace>>> g_ws;

StereoWorkspace* workspace_for(const tc2li_orb* o) {
    std::lock_guard<std::mutex> lk(g_ws_mu);
    for (auto& e : g_ws) if (e.first == o) return e.second.get();
    g_ws.emplace_back(o, std::unique_ptr<StereoWorkspace>(new StereoWorkspace()));
    return g_ws.back().second.get();
}

// SF/src/Frame.cc:997-1010: threshold 1.5*1.4*median of the accepted SADs; matches at or above it are dropped.
void median_cut(const int* sad, int n, float* u_right, float* depth, std::vector<int>& tmp) {
    tmp.clear();
    for (int i = 0; i < n; ++i) if (sad[i] >= 0) tmp.push_back(sad[i]);
    if (tmp.empty()) return;  // the reference reads vDistIdx[0] of an empty vector here (undefined)
    const size_t mid = tmp.size() / 2;
    std::nth_element(tmp.begin(), tmp.begin() + mid, tmp.end());
    const float median = (float)tmp[mid];
    const float thDist = 1.5f * 1.4f * median;
    for (int i = 0; i < n; ++i)
        if (sad[i] >= 0 && !((float)sad[i] < thDist)) { u_right[i] = -1; depth[i] = -1; }
}

}  // namespace

namespace tc2li {
void stereo_release_workspace(const tc2li_orb* o) {
    std::lock_guard<std::mutex> lk(g_ws_mu);
    for (size_t i = 0; i < g_ws.size(); ++i)
        if (g_ws[i].first == o) { g_ws.erase(g_ws.begin() + i); return; }
}
}  // namespace tc2li

extern "C" {

int tc2li_stereo_match_batch(tc2li_orb* o, int n_frames, float bf, float b, float* u_right, float* depth,
                             int32_t* best_sad, int capacity, void* stream_) {
    if (!o || n_frames < 0 || !u_right || !depth || capacity < 0 || !(b > 0)) {
        set_error("tc2li_stereo_match_batch: invalid argument");
        return TC2LI_ERR_INVALID;
    }
    if (n_frames == 0) return 0;
    if (2 * n_frames > o->last_nimg || !o->last_plain_order) {
        set_error("tc2li_stereo_match_batch: needs the features of a preceding tc2li_orb_extract_batch call with "
                  "lapping area {0,0} and 2*n_frames images");
        return TC2LI_ERR_INVALID;
    }
    hipStream_t st = (hipStream_t)stream_;
    StereoWorkspace* ws = workspace_for(o);
    std::vector<StereoFrame> frames(n_frames);
    int max_left = 0, out_total = 0;
    for (int f = 0; f < n_frames; ++f) {
        StereoFrame& fr = frames[f];
        fr.left_img = 2 * f; fr.right_img = 2 * f + 1;
        fr.left_off = o->last_kp_off[2 * f]; fr.n_left = o->last_kp_cnt[2 * f];
        fr.right_off = o->last_kp_off[2 * f + 1]; fr.n_right = o->last_kp_cnt[2 * f + 1];
        fr.out_off = out_total; fr.pad_ = 0;
        out_total += fr.n_left;
        max_left = std::max(max_left, fr.n_left);
        if (fr.n_left > capacity) { set_error("stereo output capacity %d < %d keypoints", capacity, fr.n_left); return TC2LI_ERR_CAPACITY; }
        if (fr.n_right > 4096) { set_error("more than 4096 right keypoints"); return TC2LI_ERR_INVALID; }
    }
    TC2LI_HIP_CHECK(ws->d_frames.ensure(n_frames));
    TC2LI_HIP_CHECK(ws->h_u.ensure(std::max(out_total, 1)));
    TC2LI_HIP_CHECK(ws->h_d.ensure(std::max(out_total, 1)));
    TC2LI_HIP_CHECK(ws->h_sad.ensure(std::max(out_total, 1)));
    TC2LI_HIP_CHECK(hipMemcpyAsync(ws->d_frames.p, frames.data(), n_frames * sizeof(StereoFrame), hipMemcpyHostToDevice, st));
    const float mb = b, max_d = bf / mb;  // minZ = mb, maxD = mbf / minZ (SF/src/Frame.cc:868-871)
    int max_right = 0;
    for (int f = 0; f < n_frames; ++f) max_right = std::max(max_right, frames[f].n_right);
    const int rows = o->cur_h, entry_cap = stereo_row_entry_cap(o->scale_tab, o->prm.nlevels, std::max(max_right, 1));
    TC2LI_HIP_CHECK(ws->d_row_start.ensure((size_t)n_frames * (rows + 1)));
    TC2LI_HIP_CHECK(ws->d_entries.ensure((size_t)n_frames * entry_cap));
    launch_stereo_match(o->raw_tab, o->raw_tab, o->scale_tab, ws->d_frames.p, n_frames, max_left, o->d_mkeys.p, o->d_desc.p,
                        bf, max_d, ws->h_u.p, ws->h_d.p, ws->h_sad.p, rows, entry_cap, ws->d_row_start.p, ws->d_entries.p, st);
    TC2LI_HIP_CHECK(hipGetLastError());
    TC2LI_HIP_CHECK(stream_wait_blocking(st));
    tracking_pool().parallel_for(n_frames, [&](int f) {
        const StereoFrame& fr = frames[f];
        float* u = u_right + (size_t)f * capacity;
        float* d = depth + (size_t)f * capacity;
        memcpy(u, ws->h_u.p + fr.out_off, fr.n_left * sizeof(float));
        memcpy(d, ws->h_d.p + fr.out_off, fr.n_left * sizeof(float));
        if (best_sad) memcpy(best_sad + (size_t)f * capacity, ws->h_sad.p + fr.out_off, fr.n_left * sizeof(int));
        std::vector<int> tmp;
        median_cut(ws->h_sad.p + fr.out_off, fr.n_left, u, d, tmp);
    });
    return n_frames;
}

int tc2li_stereo_match(tc2li_orb* left, tc2li_orb* right, const tc2li_keypoint* keys_left, const uint8_t* desc_left,
                       int n_left, const tc2li_keypoint* keys_right, const uint8_t* desc_right, int n_right, float bf, float b,
                       float* u_right, float* depth, int32_t* best_sad) {
    if (!left || !right || n_left < 0 || n_right < 0 || !u_right || !depth || !(b > 0) ||
        (n_left > 0 && (!keys_left || !desc_left)) || (n_right > 0 && (!keys_right || !desc_right))) {
        set_error("tc2li_stereo_match: invalid argument");
        return TC2LI_ERR_INVALID;
    }
    if (left->last_nimg < 1 || right->last_nimg < 1 || left->cur_w != right->cur_w || left->cur_h != right->cur_h) {
        set_error("tc2li_stereo_match: both extractors must have processed an image of the same size");
        return TC2LI_ERR_INVALID;
    }
    if (n_right > 4096) { set_error("more than 4096 right keypoints"); return TC2LI_ERR_INVALID; }
    if (n_left == 0) return 0;
    const int L = left->prm.nlevels;
    StereoWorkspace* ws = workspace_for(left);
    std::vector<MatchKey> keys((size_t)n_left + n_right);
    for (int i = 0; i < n_left; ++i) {
        if (keys_left[i].octave < 0 || keys_left[i].octave >= L) { set_error("bad octave"); return TC2LI_ERR_INVALID; }
        keys[i] = MatchKey{keys_left[i].x, keys_left[i].y, keys_left[i].octave};
    }
    for (int i = 0; i < n_right; ++i) {
        if (keys_right[i].octave < 0 || keys_right[i].octave >= L) { set_error("bad octave"); return TC2LI_ERR_INVALID; }
        keys[n_left + i] = MatchKey{keys_right[i].x, keys_right[i].y, keys_right[i].octave};
    }
    const size_t n = keys.size();
    TC2LI_HIP_CHECK(ws->d_keys.ensure(n));
    TC2LI_HIP_CHECK(ws->d_desc.ensure(n * 32));
    TC2LI_HIP_CHECK(ws->d_frames.ensure(1));
    TC2LI_HIP_CHECK(ws->h_u.ensure(n_left));
    TC2LI_HIP_CHECK(ws->h_d.ensure(n_left));
    TC2LI_HIP_CHECK(ws->h_sad.ensure(n_left));
    StereoFrame fr{0, n_left, n_left, n_right, 0, 0, 0, 0};
    hipStream_t ps = private_stream();
    TC2LI_HIP_CHECK(copy_sync(ws->d_keys.p, keys.data(), n * sizeof(MatchKey), hipMemcpyHostToDevice, ps));
    TC2LI_HIP_CHECK(copy_sync(ws->d_desc.p, desc_left, (size_t)n_left * 32, hipMemcpyHostToDevice, ps));
    if (n_right) TC2LI_HIP_CHECK(copy_sync(ws->d_desc.p + (size_t)n_left * 32, desc_right, (size_t)n_right * 32, hipMemcpyHostToDevice, ps));
    TC2LI_HIP_CHECK(copy_sync(ws->d_frames.p, &fr, sizeof(fr), hipMemcpyHostToDevice, ps));
    const float max_d = bf / b;
    const int rows = left->cur_h, entry_cap = stereo_row_entry_cap(left->scale_tab, L, std::max(n_right, 1));
    TC2LI_HIP_CHECK(ws->d_row_start.ensure((size_t)rows + 1));
    TC2LI_HIP_CHECK(ws->d_entries.ensure((size_t)entry_cap));
    launch_stereo_match(left->raw_tab, right->raw_tab, left->scale_tab, ws->d_frames.p, 1, n_left, ws->d_keys.p, ws->d_desc.p, bf,
                        max_d, ws->h_u.p, ws->h_d.p, ws->h_sad.p, rows, entry_cap, ws->d_row_start.p, ws->d_entries.p, ps);
    TC2LI_HIP_CHECK(hipGetLastError());
    TC2LI_HIP_CHECK(hipStreamSynchronize(ps));
    memcpy(u_right, ws->h_u.p, n_left * sizeof(float));
    memcpy(depth, ws->h_d.p, n_left * sizeof(float));
    if (best_sad) memcpy(best_sad, ws->h_sad.p, n_left * sizeof(int));
    std::vector<int> tmp;
    median_cut(ws->h_sad.p, n_left, u_right, depth, tmp);
    return n_left;
}

}  // extern "C"
