// Host side of tc2li_search_for_triangulation / tc2li_create_new_map_points (include/tc2li_hip.h): uploads of the keyframe
// views, the per-pair constants of ORBmatcher::SearchForTriangulation (epipole, fundamental matrix: float arithmetic in the
// order of Sophus / Eigen, SF/src/ORBmatcher.cc:919-944, SF/src/CameraModels/Pinhole.cpp:118-121), the rotation histogram, and
// the "first neighbour that yields a point keeps the keypoint" rule of LocalMapping::CreateNewMapPoints.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <mutex>
#include <vector>

#include "common.hpp"
#include "mapping_device.hpp"
#include "matcher_device.hpp"

using namespace tc2li;

static_assert(sizeof(tc2li_keypoint) == 24, "ABI layout");

namespace {

struct Q7 { float q[4], t[3]; };
inline void q_mul(const float a[4], const float b[4], float o[4]) {
    o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
    o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    o[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
    o[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
}
inline void q_rot(const float q[4], const float v[3], float out[3]) {
    float uv[3] = {q[1] * v[2] - q[2] * v[1], q[2] * v[0] - q[0] * v[2], q[0] * v[1] - q[1] * v[0]};
    uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
    out[0] = v[0] + q[3] * uv[0] + (q[1] * uv[2] - q[2] * uv[1]);
    out[1] = v[1] + q[3] * uv[1] + (q[2] * uv[0] - q[0] * uv[2]);
    out[2] = v[2] + q[3] * uv[2] + (q[0] * uv[1] - q[1] * uv[0]);
}
inline void q_mat(const float q[4], float R[9]) {
    const float tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
    const float twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
    const float txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
    const float tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
inline Q7 inv7(const Q7& T) {
    Q7 o;
    o.q[0] = -T.q[0]; o.q[1] = -T.q[1]; o.q[2] = -T.q[2]; o.q[3] = T.q[3];
    const float nt[3] = {T.t[0] * -1.f, T.t[1] * -1.f, T.t[2] * -1.f};
    q_rot(o.q, nt, o.t);
    return o;
}
inline Q7 mul7(const Q7& a, const Q7& b) {
    Q7 o;
    q_mul(a.q, b.q, o.q);
    float r[3];
    q_rot(a.q, b.t, r);
    for (int c = 0; c < 3; ++c) o.t[c] = r[c] + a.t[c];
    return o;
}
inline void m3_mul(const float* a, const float* b, float* o) {
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) o[3 * r + c] = (a[3 * r] * b[c] + a[3 * r + 1] * b[3 + c]) + a[3 * r + 2] * b[6 + c];
}
inline void m3_inv(const float* m, float* o) {
    const float c00 = m[4] * m[8] - m[5] * m[7], c10 = m[5] * m[6] - m[3] * m[8], c20 = m[3] * m[7] - m[4] * m[6];
    const float det = (m[0] * c00 + m[1] * c10) + m[2] * c20;
    const float id = 1.0f / det;
    o[0] = c00 * id; o[1] = (m[2] * m[7] - m[1] * m[8]) * id; o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
    o[3] = c10 * id; o[4] = (m[0] * m[8] - m[2] * m[6]) * id; o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
    o[6] = c20 * id; o[7] = (m[1] * m[6] - m[0] * m[7]) * id; o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
}

struct KfBuffers {
    DevBuf<float> keys, u_right, depth;
    DevBuf<uint8_t> desc, has_point;
    DevBuf<int32_t> fv_node, fv_off, fv_idx;
};
struct FuseWorkspace {
    DevBuf<float> keys, u_right, scale, inv_sigma;
    DevBuf<uint8_t> desc, points, valid;
    DevBuf<MatchKey> mkeys;
    DevBuf<MatchFrameDev> frame;
    DevBuf<int32_t> cell_start, key_base, best_idx, best_dist;
    DevBuf<uint16_t> items;
    std::mutex mu;
};
FuseWorkspace& fws() { return shutdown_owned<FuseWorkspace>(); }

struct Workspace {
    std::vector<std::unique_ptr<KfBuffers>> kf;  // [0] = current keyframe, [1 + j] = neighbour j
    DevBuf<KfDev> d_neigh;
    DevBuf<float> d_scale, d_sigma, d_x3D;
    DevBuf<int32_t> d_match;
    DevBuf<uint8_t> d_ok;
    std::mutex mu;
};
Workspace& ws() { return shutdown_owned<Workspace>(); }

int check_view(const tc2li_keyframe_view* v, const char* what) {
    if (!v || v->n < 0 || v->n_nodes < 0 || (v->n > 0 && (!v->keys || !v->descriptors || !v->u_right || !v->depth || !v->has_point)) ||
        (v->n_nodes > 0 && (!v->fv_node || !v->fv_offset || !v->fv_index))) {
        set_error("%s: invalid keyframe view", what);
        return TC2LI_ERR_INVALID;
    }
    if (v->n_nodes > 0) {
        if (v->fv_offset[0] != 0) { set_error("%s: fv_offset[0] must be 0", what); return TC2LI_ERR_INVALID; }
        for (int a = 0; a < v->n_nodes; ++a) {
            if (v->fv_offset[a + 1] < v->fv_offset[a] || (a > 0 && v->fv_node[a] <= v->fv_node[a - 1])) { set_error("%s: feature vector not ascending", what); return TC2LI_ERR_INVALID; }
        }
        for (int k = 0; k < v->fv_offset[v->n_nodes]; ++k)
            if (v->fv_index[k] < 0 || v->fv_index[k] >= v->n) { set_error("%s: feature index out of range", what); return TC2LI_ERR_INVALID; }
    }
    return 0;
}

int upload(const tc2li_keyframe_view& v, KfBuffers& b, KfDev& d, hipStream_t st) {
    const size_t n = std::max(v.n, 1), nn = std::max(v.n_nodes, 1), ne = std::max(v.n_nodes > 0 ? v.fv_offset[v.n_nodes] : 0, 1);
    TC2LI_HIP_CHECK(b.keys.ensure(6 * n)); TC2LI_HIP_CHECK(b.u_right.ensure(n)); TC2LI_HIP_CHECK(b.depth.ensure(n)); TC2LI_HIP_CHECK(b.desc.ensure(32 * n));
    TC2LI_HIP_CHECK(b.has_point.ensure(n)); TC2LI_HIP_CHECK(b.fv_node.ensure(nn)); TC2LI_HIP_CHECK(b.fv_off.ensure(nn + 1)); TC2LI_HIP_CHECK(b.fv_idx.ensure(ne));
    if (v.n > 0) {
        TC2LI_HIP_CHECK(hipMemcpyAsync(b.keys.p, v.keys, (size_t)v.n * sizeof(tc2li_keypoint), hipMemcpyHostToDevice, st));
        TC2LI_HIP_CHECK(hipMemcpyAsync(b.u_right.p, v.u_right, v.n * sizeof(float), hipMemcpyHostToDevice, st));
        TC2LI_HIP_CHECK(hipMemcpyAsync(b.depth.p, v.depth, v.n * sizeof(float), hipMemcpyHostToDevice, st));
        TC2LI_HIP_CHECK(hipMemcpyAsync(b.desc.p, v.descriptors, 32 * (size_t)v.n, hipMemcpyHostToDevice, st));
        TC2LI_HIP_CHECK(hipMemcpyAsync(b.has_point.p, v.has_point, v.n, hipMemcpyHostToDevice, st));
    }
    const int32_t zero = 0;
    if (v.n_nodes > 0) {
        TC2LI_HIP_CHECK(hipMemcpyAsync(b.fv_node.p, v.fv_node, v.n_nodes * sizeof(int32_t), hipMemcpyHostToDevice, st));
        TC2LI_HIP_CHECK(hipMemcpyAsync(b.fv_off.p, v.fv_offset, (v.n_nodes + 1) * sizeof(int32_t), hipMemcpyHostToDevice, st));
        if (v.fv_offset[v.n_nodes] > 0) TC2LI_HIP_CHECK(hipMemcpyAsync(b.fv_idx.p, v.fv_index, v.fv_offset[v.n_nodes] * sizeof(int32_t), hipMemcpyHostToDevice, st));
    } else {
        TC2LI_HIP_CHECK(hipMemcpyAsync(b.fv_off.p, &zero, sizeof(int32_t), hipMemcpyHostToDevice, st));
    }
    d = KfDev{};
    d.n = v.n; d.n_nodes = v.n_nodes;
    d.keys = b.keys.p; d.desc = b.desc.p; d.u_right = b.u_right.p; d.depth = b.depth.p; d.has_point = b.has_point.p;
    d.fv_node = b.fv_node.p; d.fv_off = b.fv_off.p; d.fv_idx = b.fv_idx.p;
    memcpy(d.q, v.pose7, 16); memcpy(d.t, v.pose7 + 4, 12);
    q_mat(d.q, d.Rcw);
    Q7 T; memcpy(T.q, d.q, 16); memcpy(T.t, d.t, 12);
    const Q7 Tw = inv7(T);
    memcpy(d.Ow, Tw.t, 12);
    return 0;
}

// epipole of the current keyframe in the neighbour and F12 (the same pinhole camera on both sides)
void pair_constants(const KfDev& k1, KfDev& k2, const tc2li_camera* cam) {
    Q7 T1, T2;
    memcpy(T1.q, k1.q, 16); memcpy(T1.t, k1.t, 12); memcpy(T2.q, k2.q, 16); memcpy(T2.t, k2.t, 12);
    float C2[3];
    q_rot(T2.q, k1.Ow, C2);
    for (int c = 0; c < 3; ++c) C2[c] += T2.t[c];
    const float fx = (float)cam->fx, fy = (float)cam->fy, cx = (float)cam->cx, cy = (float)cam->cy;
    k2.ep[0] = fx * C2[0] / C2[2] + cx; k2.ep[1] = fy * C2[1] / C2[2] + cy;
    const Q7 T12 = mul7(T1, inv7(T2));
    float R12[9];
    q_mat(T12.q, R12);
    const float K[9] = {fx, 0.f, cx, 0.f, fy, cy, 0.f, 0.f, 1.f};
    const float Kt[9] = {K[0], K[3], K[6], K[1], K[4], K[7], K[2], K[5], K[8]};
    const float tx[9] = {0.f, -T12.t[2], T12.t[1], T12.t[2], 0.f, -T12.t[0], -T12.t[1], T12.t[0], 0.f};
    float KtInv[9], KInv[9], a[9], b[9];
    m3_inv(Kt, KtInv);
    m3_inv(K, KInv);
    m3_mul(KtInv, tx, a);
    m3_mul(a, R12, b);
    m3_mul(b, KInv, k2.F12);
}

// uploads everything, runs the search (and the point kernel) for every neighbour; results stay in the workspace
int run(Workspace& w, const tc2li_keyframe_view* cur, const tc2li_keyframe_view* neigh, int n_neigh, const tc2li_camera* cam, float mb, float mbf,
        const float* scale_factors, const float* level_sigma2, int n_levels, float scale_factor, int inertial, int far_points, float th_far,
        int only_stereo, int coarse, bool points, std::vector<KfDev>& hn, MappingDev& m, hipStream_t st) {
    while ((int)w.kf.size() < n_neigh + 1) w.kf.emplace_back(new KfBuffers());
    hn.assign(n_neigh, KfDev{});
    m = MappingDev{};
    int rc = upload(*cur, *w.kf[0], m.cur, st);
    if (rc < 0) return rc;
    for (int j = 0; j < n_neigh; ++j) {
        rc = upload(neigh[j], *w.kf[1 + j], hn[j], st);
        if (rc < 0) return rc;
        pair_constants(m.cur, hn[j], cam);
        const float vb[3] = {hn[j].Ow[0] - m.cur.Ow[0], hn[j].Ow[1] - m.cur.Ow[1], hn[j].Ow[2] - m.cur.Ow[2]};
        const float baseline = std::sqrt(vb[0] * vb[0] + vb[1] * vb[1] + vb[2] * vb[2]);
        hn[j].skip = (points && baseline < mb) ? 1 : 0;  // LocalMapping.cc:456-460 (stereo / RGB-D branch)
    }
    const size_t slots = (size_t)std::max(n_neigh, 1) * std::max(cur->n, 1);
    TC2LI_HIP_CHECK(w.d_neigh.ensure(std::max(n_neigh, 1))); TC2LI_HIP_CHECK(w.d_scale.ensure(n_levels)); TC2LI_HIP_CHECK(w.d_sigma.ensure(n_levels));
    TC2LI_HIP_CHECK(w.d_match.ensure(slots)); TC2LI_HIP_CHECK(w.d_ok.ensure(slots)); TC2LI_HIP_CHECK(w.d_x3D.ensure(3 * slots));
    if (n_neigh) TC2LI_HIP_CHECK(hipMemcpyAsync(w.d_neigh.p, hn.data(), n_neigh * sizeof(KfDev), hipMemcpyHostToDevice, st));
    TC2LI_HIP_CHECK(hipMemcpyAsync(w.d_scale.p, scale_factors, n_levels * sizeof(float), hipMemcpyHostToDevice, st));
    TC2LI_HIP_CHECK(hipMemcpyAsync(w.d_sigma.p, level_sigma2, n_levels * sizeof(float), hipMemcpyHostToDevice, st));
    TC2LI_HIP_CHECK(hipMemsetAsync(w.d_match.p, 0xff, slots * sizeof(int32_t), st));
    TC2LI_HIP_CHECK(hipMemsetAsync(w.d_ok.p, 0, slots, st));
    m.neigh = w.d_neigh.p; m.n_neigh = n_neigh; m.n_levels = n_levels;
    m.fx = (float)cam->fx; m.fy = (float)cam->fy; m.cx = (float)cam->cx; m.cy = (float)cam->cy; m.mb = mb; m.mbf = mbf;
    m.ratio_factor = 1.5f * scale_factor; m.th_far = th_far;
    m.inertial = inertial; m.far_points = far_points; m.only_stereo = only_stereo; m.coarse = coarse;
    m.scale_factors = w.d_scale.p; m.level_sigma2 = w.d_sigma.p; m.match = w.d_match.p; m.ok = w.d_ok.p; m.x3D = w.d_x3D.p;
    launch_tri_search(m, cur->n_nodes > 0 ? cur->fv_offset[cur->n_nodes] : 0, st);
    if (points) launch_tri_points(m, st);
    TC2LI_HIP_CHECK(hipGetLastError());
    return 0;
}

bool octaves_ok(const tc2li_keyframe_view* v, int n_levels) {
    for (int i = 0; i < v->n; ++i) if (v->keys[i].octave < 0 || v->keys[i].octave >= n_levels) return false;
    return true;
}

}  // namespace

extern "C" int tc2li_search_for_triangulation(const tc2li_keyframe_view* kf1, const tc2li_keyframe_view* kf2, const tc2li_camera* cam,
                                              const float* scale_factors, const float* level_sigma2, int n_levels, int only_stereo, int coarse,
                                              int check_orientation, int32_t* match12, void* stream_) {
    if (!cam || !scale_factors || !level_sigma2 || n_levels < 1 || !match12) { set_error("tc2li_search_for_triangulation: invalid argument"); return TC2LI_ERR_INVALID; }
    int rc = check_view(kf1, "tc2li_search_for_triangulation (kf1)");
    if (rc < 0) return rc;
    rc = check_view(kf2, "tc2li_search_for_triangulation (kf2)");
    if (rc < 0) return rc;
    if (!octaves_ok(kf1, n_levels) || !octaves_ok(kf2, n_levels)) { set_error("tc2li_search_for_triangulation: keypoint octave out of range"); return TC2LI_ERR_INVALID; }
    for (int i = 0; i < kf1->n; ++i) match12[i] = -1;
    if (kf1->n == 0 || kf2->n == 0) return 0;
    if (!device_ready()) return TC2LI_ERR_NO_DEVICE;
    hipStream_t st = (hipStream_t)stream_;
    Workspace& w = ws();
    std::lock_guard<std::mutex> lk(w.mu);
    std::vector<KfDev> hn;
    MappingDev m;
    rc = run(w, kf1, kf2, 1, cam, 0.f, 0.f, scale_factors, level_sigma2, n_levels, 1.f, 0, 0, 0.f, only_stereo, coarse, false, hn, m, st);
    if (rc < 0) return rc;
    TC2LI_HIP_CHECK(hipMemcpyAsync(match12, w.d_match.p, kf1->n * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    TC2LI_HIP_CHECK(hipStreamSynchronize(st));
    int nmatches = 0;
    for (int i = 0; i < kf1->n; ++i) nmatches += match12[i] >= 0;
    if (check_orientation) {  // rotation histogram over the matches in feature-vector order (ORBmatcher.cc:1096-1131)
        std::vector<int> hist[30];
        const float factor = 1.0f / 30;
        for (int e = 0; e < (kf1->n_nodes > 0 ? kf1->fv_offset[kf1->n_nodes] : 0); ++e) {
            const int idx1 = kf1->fv_index[e];
            if (match12[idx1] < 0) continue;
            float rot = kf1->keys[idx1].angle - kf2->keys[match12[idx1]].angle;
            if (rot < 0.0) rot += 360.0f;
            int bin = (int)std::round(rot * factor);
            if (bin == 30) bin = 0;
            if (std::find(hist[bin].begin(), hist[bin].end(), idx1) == hist[bin].end()) hist[bin].push_back(idx1);
        }
        int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
        for (int i = 0; i < 30; i++) {
            const int s = (int)hist[i].size();
            if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
            else if (s > max3) { max3 = s; ind3 = i; }
        }
        if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
        else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
        for (int i = 0; i < 30; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int idx : hist[i]) { match12[idx] = -1; nmatches--; }
        }
    }
    return nmatches;
}

extern "C" int tc2li_create_new_map_points(const tc2li_keyframe_view* cur, const tc2li_keyframe_view* neighbours, int n_neighbours,
                                           const tc2li_camera* cam, float mb, const float* scale_factors, const float* level_sigma2, int n_levels,
                                           float scale_factor, int inertial, int far_points, float th_far_points, int coarse,
                                           tc2li_new_map_point* points, int capacity, void* stream_) {
    if (!cam || !scale_factors || !level_sigma2 || n_levels < 1 || n_neighbours < 0 || (n_neighbours > 0 && !neighbours) || capacity < 0 ||
        (capacity > 0 && !points)) {
        set_error("tc2li_create_new_map_points: invalid argument");
        return TC2LI_ERR_INVALID;
    }
    int rc = check_view(cur, "tc2li_create_new_map_points (current keyframe)");
    if (rc < 0) return rc;
    if (!octaves_ok(cur, n_levels)) { set_error("tc2li_create_new_map_points: keypoint octave out of range"); return TC2LI_ERR_INVALID; }
    for (int j = 0; j < n_neighbours; ++j) {
        rc = check_view(neighbours + j, "tc2li_create_new_map_points (neighbour)");
        if (rc < 0) return rc;
        if (!octaves_ok(neighbours + j, n_levels)) { set_error("tc2li_create_new_map_points: keypoint octave out of range"); return TC2LI_ERR_INVALID; }
    }
    if (cur->n == 0 || n_neighbours == 0) return 0;
    if (!device_ready()) return TC2LI_ERR_NO_DEVICE;
    hipStream_t st = (hipStream_t)stream_;
    Workspace& w = ws();
    std::lock_guard<std::mutex> lk(w.mu);
    std::vector<KfDev> hn;
    MappingDev m;
    rc = run(w, cur, neighbours, n_neighbours, cam, mb, (float)cam->bf, scale_factors, level_sigma2, n_levels, scale_factor, inertial, far_points,
             th_far_points, 0, coarse, true, hn, m, st);
    if (rc < 0) return rc;
    const size_t slots = (size_t)n_neighbours * cur->n;
    std::vector<int32_t> match(slots);
    std::vector<uint8_t> ok(slots);
    std::vector<float> x3D(3 * slots);
    TC2LI_HIP_CHECK(hipMemcpyAsync(match.data(), w.d_match.p, slots * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    TC2LI_HIP_CHECK(hipMemcpyAsync(ok.data(), w.d_ok.p, slots, hipMemcpyDeviceToHost, st));
    TC2LI_HIP_CHECK(hipMemcpyAsync(x3D.data(), w.d_x3D.p, 3 * slots * sizeof(float), hipMemcpyDeviceToHost, st));
    TC2LI_HIP_CHECK(hipStreamSynchronize(st));
    // neighbour-major, keypoint-ascending: the order in which the reference creates the points; a keypoint that got its point from
    // an earlier neighbour is not looked at again (mpCurrentKeyFrame->AddMapPoint before the next SearchForTriangulation)
    std::vector<uint8_t> taken(cur->n, 0);
    int count = 0;
    for (int j = 0; j < n_neighbours; ++j)
        for (int i = 0; i < cur->n; ++i) {
            const size_t s = (size_t)j * cur->n + i;
            if (!ok[s] || taken[i]) continue;
            taken[i] = 1;
            if (count < capacity) {
                tc2li_new_map_point& p = points[count];
                p.idx1 = i; p.neighbour = j; p.idx2 = match[s]; p.stereo = (ok[s] & 2) ? 1 : 0;
                memcpy(p.x3D, &x3D[3 * s], 12);
                p.pad_ = 0;
            }
            ++count;
        }
    if (count > capacity) { set_error("tc2li_create_new_map_points: %d points, capacity %d", count, capacity); return TC2LI_ERR_CAPACITY; }
    return count;
}

static_assert(sizeof(tc2li_map_point) == 68, "ABI layout");

extern "C" int tc2li_fuse_search(const tc2li_frame_view* kf, const float pose7[7], const float cam4[4], float bf, const float* scale_factors,
                                 const float* inv_level_sigma2, int n_levels, float log_scale_factor, const tc2li_map_point* points,
                                 const uint8_t* valid, int n_points, float th, int32_t* best_idx, int32_t* best_dist, void* stream_) {
    if (!kf || !pose7 || !cam4 || !scale_factors || !inv_level_sigma2 || n_levels < 1 || n_points < 0 || (n_points > 0 && (!points || !valid || !best_idx)) ||
        kf->n < 0 || !(kf->max_x > kf->min_x) || !(kf->max_y > kf->min_y) || !(log_scale_factor > 0)) {
        set_error("tc2li_fuse_search: invalid argument");
        return TC2LI_ERR_INVALID;
    }
    if (n_points == 0) return 0;
    for (int i = 0; i < n_points; ++i) { best_idx[i] = -1; if (best_dist) best_dist[i] = 256; }
    if (kf->n == 0) return 0;
    if (!kf->keys || !kf->descriptors || !kf->u_right) { set_error("tc2li_fuse_search: null keyframe arrays"); return TC2LI_ERR_INVALID; }
    if (kf->n > kMaxMatchKeys) { set_error("keyframe has %d keypoints, the feature grid supports %d", kf->n, kMaxMatchKeys); return TC2LI_ERR_CAPACITY; }
    for (int i = 0; i < kf->n; ++i) if (kf->keys[i].octave < 0 || kf->keys[i].octave >= n_levels) { set_error("tc2li_fuse_search: keypoint octave out of range"); return TC2LI_ERR_INVALID; }
    if (!device_ready()) return TC2LI_ERR_NO_DEVICE;
    hipStream_t st = (hipStream_t)stream_;
    FuseWorkspace& w = fws();
    std::lock_guard<std::mutex> lk(w.mu);
    const int N = kf->n;
    std::vector<MatchKey> mk(N);
    for (int i = 0; i < N; ++i) mk[i] = MatchKey{kf->keys[i].x, kf->keys[i].y, kf->keys[i].octave};
    TC2LI_HIP_CHECK(w.keys.ensure(6 * (size_t)N)); TC2LI_HIP_CHECK(w.u_right.ensure(N)); TC2LI_HIP_CHECK(w.desc.ensure(32 * (size_t)N)); TC2LI_HIP_CHECK(w.mkeys.ensure(N));
    TC2LI_HIP_CHECK(w.scale.ensure(n_levels)); TC2LI_HIP_CHECK(w.inv_sigma.ensure(n_levels)); TC2LI_HIP_CHECK(w.points.ensure(68 * (size_t)n_points));
    TC2LI_HIP_CHECK(w.valid.ensure(n_points)); TC2LI_HIP_CHECK(w.frame.ensure(1)); TC2LI_HIP_CHECK(w.cell_start.ensure(64 * 48 + 1)); TC2LI_HIP_CHECK(w.key_base.ensure(1));
    TC2LI_HIP_CHECK(w.items.ensure(N)); TC2LI_HIP_CHECK(w.best_idx.ensure(n_points)); TC2LI_HIP_CHECK(w.best_dist.ensure(n_points));
    TC2LI_HIP_CHECK(hipMemcpyAsync(w.keys.p, kf->keys, (size_t)N * sizeof(tc2li_keypoint), hipMemcpyHostToDevice, st));
    TC2LI_HIP_CHECK(hipMemcpyAsync(w.mkeys.p, mk.data(), N * sizeof(MatchKey), hipMemcpyHostToDevice, st));
    TC2LI_HIP_CHECK(hipMemcpyAsync(w.u_right.p, kf->u_right, N * sizeof(float), hipMemcpyHostToDevice, st));
    TC2LI_HIP_CHECK(hipMemcpyAsync(w.desc.p, kf->descriptors, 32 * (size_t)N, hipMemcpyHostToDevice, st));
    TC2LI_HIP_CHECK(hipMemcpyAsync(w.scale.p, scale_factors, n_levels * sizeof(float), hipMemcpyHostToDevice, st));
    TC2LI_HIP_CHECK(hipMemcpyAsync(w.inv_sigma.p, inv_level_sigma2, n_levels * sizeof(float), hipMemcpyHostToDevice, st));
    TC2LI_HIP_CHECK(hipMemcpyAsync(w.points.p, points, 68 * (size_t)n_points, hipMemcpyHostToDevice, st));
    TC2LI_HIP_CHECK(hipMemcpyAsync(w.valid.p, valid, n_points, hipMemcpyHostToDevice, st));
    TC2LI_HIP_CHECK(hipMemsetAsync(w.key_base.p, 0, sizeof(int32_t), st));
    const MatchFrameDev fd{w.mkeys.p, w.desc.p, w.u_right.p, nullptr, nullptr, N, 0, 0, 0, kf->min_x, kf->max_x, kf->min_y, kf->max_y};
    TC2LI_HIP_CHECK(hipMemcpyAsync(w.frame.p, &fd, sizeof(fd), hipMemcpyHostToDevice, st));
    MatchLists L{};
    L.cell_start = w.cell_start.p; L.items = w.items.p; L.key_base = w.key_base.p;
    launch_match_grid(w.frame.p, 1, L, st);
    FuseDev f{};
    f.keys = w.keys.p; f.desc = w.desc.p; f.u_right = w.u_right.p; f.cell_start = w.cell_start.p; f.items = w.items.p;
    f.n_keys = N; f.n_points = n_points; f.n_levels = n_levels;
    memcpy(f.q, pose7, 16); memcpy(f.t, pose7 + 4, 12);
    Q7 T; memcpy(T.q, pose7, 16); memcpy(T.t, pose7 + 4, 12);
    const Q7 Tw = inv7(T);
    memcpy(f.Ow, Tw.t, 12);
    f.fx = cam4[0]; f.fy = cam4[1]; f.cx = cam4[2]; f.cy = cam4[3]; f.bf = bf; f.th = th; f.log_scale_factor = log_scale_factor;
    f.min_x = kf->min_x; f.max_x = kf->max_x; f.min_y = kf->min_y; f.max_y = kf->max_y;
    f.scale_factors = w.scale.p; f.inv_level_sigma2 = w.inv_sigma.p; f.points = w.points.p; f.valid = w.valid.p;
    f.best_idx = w.best_idx.p; f.best_dist = w.best_dist.p;
    launch_fuse_search(f, st);
    TC2LI_HIP_CHECK(hipGetLastError());
    TC2LI_HIP_CHECK(hipMemcpyAsync(best_idx, w.best_idx.p, n_points * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    if (best_dist) TC2LI_HIP_CHECK(hipMemcpyAsync(best_dist, w.best_dist.p, n_points * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    TC2LI_HIP_CHECK(hipStreamSynchronize(st));
    int nfused = 0;
    for (int i = 0; i < n_points; ++i) nfused += best_idx[i] >= 0;
    return nfused;
}
