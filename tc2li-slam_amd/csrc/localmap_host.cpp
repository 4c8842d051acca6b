// tc2li_local_map_* (include/tc2li_hip.h): a device-resident mirror of the keyframe-graph pieces Tracking::UpdateLocalKeyFrames /
// UpdateLocalPoints read (SF/src/Tracking.cc:3296-3476), uploaded when the map changes, and the per-frame update on it.
#include <cstring>
#include <mutex>
#include <vector>

#include "common.hpp"
#include "localmap_device.hpp"

using namespace tc2li;

struct tc2li_local_map {
    std::mutex mu;
    int n_keyframes = 0, n_points = 0, total_matches = 0;
    bool has_graph = false;
    DevBuf<uint8_t> d_kf_bad, d_point_bad, d_marked, d_cleared;
    DevBuf<int32_t> d_covis_off, d_covis, d_child_off, d_children, d_parent, d_prev, d_match_off, d_matches, d_obs_off, d_obs_kf;
    DevBuf<int32_t> d_votes, d_kf_list, d_rev_base, d_header, d_first, d_block_counts, d_points, d_frame_points;
    PinnedBuf<int32_t> h_header;
    LocalMapDev dev{};
};

namespace {

// offsets [n + 1] start at 0, never decrease; the values they index lie in [lo, hi)
bool csr_ok(const int32_t* off, const int32_t* val, int n, int lo, int hi, const char* what) {
    if (!off) { set_error("map graph: %s offsets are null", what); return false; }
    if (off[0] != 0) { set_error("map graph: %s offsets must start at 0", what); return false; }
    for (int i = 0; i < n; ++i) if (off[i + 1] < off[i]) { set_error("map graph: %s offsets decrease at %d", what, i); return false; }
    if (off[n] > 0 && !val) { set_error("map graph: %s values are null", what); return false; }
    for (int k = 0; k < off[n]; ++k) if (val[k] < lo || val[k] >= hi) { set_error("map graph: %s entry %d = %d out of range", what, k, val[k]); return false; }
    return true;
}

template <typename T>
hipError_t put(DevBuf<T>& d, const T* src, size_t n, hipStream_t st) {
    hipError_t e = d.ensure(n > 0 ? n : 1);
    if (e != hipSuccess || n == 0) return e;
    return hipMemcpyAsync(d.p, src, n * sizeof(T), hipMemcpyHostToDevice, st);
}

}  // namespace

extern "C" {

int tc2li_local_map_create(tc2li_local_map** out) {
    if (!out) { set_error("tc2li_local_map_create: invalid argument"); return TC2LI_ERR_INVALID; }
    *out = nullptr;
    if (!device_ready()) return TC2LI_ERR_NO_DEVICE;
    *out = new tc2li_local_map();
    return TC2LI_OK;
}

void tc2li_local_map_destroy(tc2li_local_map* m) { delete m; }

int tc2li_local_map_set_graph(tc2li_local_map* m, const tc2li_map_graph* g, void* stream_) {
    if (!m || !g || g->n_keyframes < 0 || g->n_points < 0) { set_error("tc2li_local_map_set_graph: invalid argument"); return TC2LI_ERR_INVALID; }
    const int K = g->n_keyframes, P = g->n_points;
    if (K > 0 && (!g->kf_bad || !g->parent || !g->prev_kf)) { set_error("map graph: null keyframe arrays"); return TC2LI_ERR_INVALID; }
    if (P > 0 && !g->point_bad) { set_error("map graph: null point arrays"); return TC2LI_ERR_INVALID; }
    if (!csr_ok(g->covis_offsets, g->covis, K, 0, K, "covisibility") || !csr_ok(g->child_offsets, g->children, K, 0, K, "children") ||
        !csr_ok(g->match_offsets, g->matches, K, -1, P, "matches") || !csr_ok(g->obs_offsets, g->obs_kf, P, 0, K, "observations"))
        return TC2LI_ERR_INVALID;
    for (int k = 0; k < K; ++k)
        if (g->parent[k] < -1 || g->parent[k] >= K || g->prev_kf[k] < -1 || g->prev_kf[k] >= K) { set_error("map graph: parent / previous keyframe of %d out of range", k); return TC2LI_ERR_INVALID; }
    if (!device_ready()) return TC2LI_ERR_NO_DEVICE;
    hipStream_t st = (hipStream_t)stream_;
    std::lock_guard<std::mutex> lk(m->mu);
    TC2LI_HIP_CHECK(put(m->d_kf_bad, g->kf_bad, K, st)); TC2LI_HIP_CHECK(put(m->d_point_bad, g->point_bad, P, st));
    TC2LI_HIP_CHECK(put(m->d_covis_off, g->covis_offsets, (size_t)K + 1, st)); TC2LI_HIP_CHECK(put(m->d_covis, g->covis, g->covis_offsets[K], st));
    TC2LI_HIP_CHECK(put(m->d_child_off, g->child_offsets, (size_t)K + 1, st)); TC2LI_HIP_CHECK(put(m->d_children, g->children, g->child_offsets[K], st));
    TC2LI_HIP_CHECK(put(m->d_parent, g->parent, K, st)); TC2LI_HIP_CHECK(put(m->d_prev, g->prev_kf, K, st));
    TC2LI_HIP_CHECK(put(m->d_match_off, g->match_offsets, (size_t)K + 1, st)); TC2LI_HIP_CHECK(put(m->d_matches, g->matches, g->match_offsets[K], st));
    TC2LI_HIP_CHECK(put(m->d_obs_off, g->obs_offsets, (size_t)P + 1, st)); TC2LI_HIP_CHECK(put(m->d_obs_kf, g->obs_kf, g->obs_offsets[P], st));
    m->total_matches = g->match_offsets[K];
    const size_t Kc = std::max(K, 1), Pc = std::max(P, 1);
    TC2LI_HIP_CHECK(m->d_votes.ensure(Kc)); TC2LI_HIP_CHECK(m->d_marked.ensure(Kc)); TC2LI_HIP_CHECK(m->d_kf_list.ensure(Kc));
    TC2LI_HIP_CHECK(m->d_rev_base.ensure(Kc + 1)); TC2LI_HIP_CHECK(m->d_header.ensure(4)); TC2LI_HIP_CHECK(m->d_first.ensure(Pc));
    TC2LI_HIP_CHECK(m->d_block_counts.ensure((size_t)m->total_matches / 256 + 2)); TC2LI_HIP_CHECK(m->d_points.ensure(Pc));
    TC2LI_HIP_CHECK(m->h_header.ensure(4));
    TC2LI_HIP_CHECK(hipStreamSynchronize(st));  // the caller's arrays may go away
    m->n_keyframes = K; m->n_points = P; m->has_graph = true;
    LocalMapDev& d = m->dev;
    d.n_keyframes = K; d.n_points = P; d.kf_bad = m->d_kf_bad.p; d.covis_off = m->d_covis_off.p; d.covis = m->d_covis.p;
    d.child_off = m->d_child_off.p; d.children = m->d_children.p; d.parent = m->d_parent.p; d.prev_kf = m->d_prev.p;
    d.match_off = m->d_match_off.p; d.matches = m->d_matches.p; d.point_bad = m->d_point_bad.p; d.obs_off = m->d_obs_off.p; d.obs_kf = m->d_obs_kf.p;
    d.votes = m->d_votes.p; d.marked = m->d_marked.p; d.kf_list = m->d_kf_list.p; d.rev_base = m->d_rev_base.p; d.header = m->d_header.p;
    d.first_pos = m->d_first.p; d.block_counts = m->d_block_counts.p; d.points = m->d_points.p;
    return TC2LI_OK;
}

int tc2li_local_map_update(tc2li_local_map* m, const int32_t* frame_points, int n_frame_points, int temporal_last_kf,
                           int32_t* local_keyframes, int keyframe_capacity, int32_t* n_local_keyframes, int32_t* reference_kf,
                           int32_t* local_points, int point_capacity, int32_t* n_local_points, uint8_t* frame_point_cleared, void* stream_) {
    if (!m || n_frame_points < 0 || (n_frame_points > 0 && (!frame_points || !frame_point_cleared)) || !local_keyframes || !n_local_keyframes ||
        !reference_kf || !local_points || !n_local_points || keyframe_capacity < 0 || point_capacity < 0) {
        set_error("tc2li_local_map_update: invalid argument");
        return TC2LI_ERR_INVALID;
    }
    std::lock_guard<std::mutex> lk(m->mu);
    if (!m->has_graph) { set_error("tc2li_local_map_update: no graph set"); return TC2LI_ERR_INVALID; }
    if (temporal_last_kf < -1 || temporal_last_kf >= m->n_keyframes) { set_error("tc2li_local_map_update: temporal_last_kf out of range"); return TC2LI_ERR_INVALID; }
    for (int i = 0; i < n_frame_points; ++i)
        if (frame_points[i] < -1 || frame_points[i] >= m->n_points) { set_error("frame point %d = %d out of range", i, frame_points[i]); return TC2LI_ERR_INVALID; }
    *n_local_keyframes = 0; *n_local_points = 0; *reference_kf = -1;
    if (!device_ready()) return TC2LI_ERR_NO_DEVICE;
    hipStream_t st = (hipStream_t)stream_;
    const LocalMapDev& d = m->dev;
    const size_t K = m->n_keyframes, P = m->n_points;
    TC2LI_HIP_CHECK(m->d_frame_points.ensure(std::max(n_frame_points, 1))); TC2LI_HIP_CHECK(m->d_cleared.ensure(std::max(n_frame_points, 1)));
    if (K) { TC2LI_HIP_CHECK(hipMemsetAsync(d.votes, 0, K * sizeof(int32_t), st)); TC2LI_HIP_CHECK(hipMemsetAsync(d.marked, 0, K, st)); }
    if (P) TC2LI_HIP_CHECK(hipMemsetAsync(d.first_pos, 0x7f, P * sizeof(int32_t), st));  // 0x7f7f7f7f: beyond any position
    if (n_frame_points) TC2LI_HIP_CHECK(hipMemcpyAsync(m->d_frame_points.p, frame_points, n_frame_points * sizeof(int32_t), hipMemcpyHostToDevice, st));
    launch_local_map_votes(d, m->d_frame_points.p, n_frame_points, m->d_cleared.p, st);
    launch_local_map_keyframes(d, temporal_last_kf, st);
    TC2LI_HIP_CHECK(hipGetLastError());
    TC2LI_HIP_CHECK(hipMemcpyAsync(m->h_header.p, d.header, 3 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    TC2LI_HIP_CHECK(hipStreamSynchronize(st));  // the grid of the point kernels is the length of the concatenated match lists
    const int n_local = m->h_header.p[0], total = m->h_header.p[1];
    *reference_kf = m->h_header.p[2];
    if (n_local > keyframe_capacity) { set_error("%d local keyframes, capacity %d", n_local, keyframe_capacity); return TC2LI_ERR_CAPACITY; }
    int n_pts = 0;
    if (total > 0) {
        launch_local_map_points(d, n_local, total, st);
        TC2LI_HIP_CHECK(hipGetLastError());
        TC2LI_HIP_CHECK(hipMemcpyAsync(m->h_header.p + 3, d.header + 3, sizeof(int32_t), hipMemcpyDeviceToHost, st));
        TC2LI_HIP_CHECK(hipStreamSynchronize(st));
        n_pts = m->h_header.p[3];
        if (n_pts > point_capacity) { set_error("%d local points, capacity %d", n_pts, point_capacity); return TC2LI_ERR_CAPACITY; }
    }
    if (n_local) TC2LI_HIP_CHECK(hipMemcpyAsync(local_keyframes, d.kf_list, n_local * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    if (n_pts) TC2LI_HIP_CHECK(hipMemcpyAsync(local_points, d.points, n_pts * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    if (n_frame_points) TC2LI_HIP_CHECK(hipMemcpyAsync(frame_point_cleared, m->d_cleared.p, n_frame_points, hipMemcpyDeviceToHost, st));
    TC2LI_HIP_CHECK(hipStreamSynchronize(st));
    *n_local_keyframes = n_local;
    *n_local_points = n_pts;
    return n_pts;
}

const int32_t* tc2li_local_map_device_points(const tc2li_local_map* m) { return m ? m->dev.points : nullptr; }

}  // extern "C"
