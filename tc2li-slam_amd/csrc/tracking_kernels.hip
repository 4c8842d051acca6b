// The per-frame bookkeeping of the batched tracking entries on the device (tracking_host.cpp sequences it):
//   * query construction of the two ORBmatcher::SearchByProjection overloads the tracking thread uses -- the last-frame overload
//     (SF/src/ORBmatcher.cc:1696-1739) and the local-map overload with Frame::isInFrustum + MapPoint::PredictScale
//     (SF/src/Frame.cc:542-603, ORBmatcher.cc:62-81) -- one lane per source point, float arithmetic in the reference's order;
//   * the rotation-consistency filter of the last-frame overload (:1783-1799, 1858-1881, ComputeThreeMaxima :2021-2062);
//   * mvpMapPoints of the current frame and the edge list of Optimizer::PoseOptimization (SF/src/Optimizer.cc:858-990) in keypoint
//     order, and what the callers do with its result (Tracking.cc:2798-2822, 3192-3227).
#include <hip/hip_runtime.h>

#include "launch.hpp"
#pragma clang fp contract(off)
#include <stdint.h>

#include "tracking_device.hpp"

namespace tc2li {

namespace {

__device__ __forceinline__ void quat_rotate_f(const float* q, const float* v, float* out) {  // Eigen::Quaternionf::_transformVector
    float uv0 = q[1] * v[2] - q[2] * v[1], uv1 = q[2] * v[0] - q[0] * v[2], uv2 = q[0] * v[1] - q[1] * v[0];
    uv0 += uv0; uv1 += uv1; uv2 += uv2;
    out[0] = v[0] + q[3] * uv0 + (q[1] * uv2 - q[2] * uv1);
    out[1] = v[1] + q[3] * uv1 + (q[2] * uv0 - q[0] * uv2);
    out[2] = v[2] + q[3] * uv2 + (q[0] * uv1 - q[1] * uv0);
}

// the frame query g belongs to: frames are ordered by q_off, empty frames share their neighbour's offset
__device__ __forceinline__ int frame_of_query(const TrackFrameDev* __restrict__ frames, int n_frames, int g) {
    int lo = 0, hi = n_frames;  // first frame with q_off > g
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (frames[mid].q_off > g) hi = mid; else lo = mid + 1;
    }
    return lo - 1;
}

__device__ __forceinline__ void store_query(MatchQuery* dst, const MatchQuery& q) {
    const uint4* s = reinterpret_cast<const uint4*>(&q);
    uint4* d = reinterpret_cast<uint4*>(dst);
    d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; d[3] = s[3];
}

__device__ __forceinline__ int block_excl_scan_256(int v, int* s_wave, int& total) {
    const int lane = threadIdx.x & 63, wave = wave_in_block();
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { const int w = s_wave[k]; base += k < wave ? w : 0; tot += w; }
    total = tot;
    __syncthreads();
    return base + incl - v;
}

}  // namespace

// ---- ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono): one query per last-frame keypoint -------------------
__global__ __launch_bounds__(256) void k_track_queries_last(const TrackFrameDev* __restrict__ frames, int n_frames, TrackConst C, LastFrameArrays A, int total_q,
                                                            MatchQuery* __restrict__ queries, int32_t* __restrict__ query_frame, int32_t* __restrict__ match) {
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= total_q) return;
    const int f = frame_of_query(frames, n_frames, g);
    if (f < 0 || g >= frames[f].q_off + frames[f].n_q) return;
    const TrackFrameDev& F = frames[f];
    if (F.slot < 0) { query_frame[g] = -1; return; }  // not part of this pass: the query and its match stay as they are
    query_frame[g] = F.slot;
    match[g] = -1;
    MatchQuery Q;
    Q.u = 0; Q.v = 0; Q.radius = 0; Q.u_right = -1; Q.min_level = -1; Q.max_level = -1; Q.angle = 0; Q.valid = 0; Q.has_observations = 1;
#pragma unroll
    for (int k = 0; k < 32; ++k) Q.desc[k] = 0;
    const uint8_t fl = A.flags[g];
    bool ok = (fl & 1) && !(fl & 2);
    float u = 0, v = 0, invzc = 0;
    const int oct = A.octave[g];
    if (ok) {
        const float X[3] = {A.Xw[3 * (size_t)g], A.Xw[3 * (size_t)g + 1], A.Xw[3 * (size_t)g + 2]};
        float pc[3];
        quat_rotate_f(F.pose7, X, pc);
        pc[0] += F.pose7[4]; pc[1] += F.pose7[5]; pc[2] += F.pose7[6];
        invzc = (float)(1.0 / (double)pc[2]);
        if (invzc < 0) ok = false;
        u = C.cam4[0] * pc[0] / pc[2] + C.cam4[2];
        v = C.cam4[1] * pc[1] / pc[2] + C.cam4[3];
        if (u < 0.f || u > (float)C.cols) ok = false;
        if (v < 0.f || v > (float)C.rows) ok = false;
    }
    if (ok) {
        Q.radius = F.th * C.scale[oct];
        if (F.forward) { Q.min_level = oct; Q.max_level = -1; }
        else if (F.backward) { Q.min_level = 0; Q.max_level = oct; }
        else { Q.min_level = oct - 1; Q.max_level = oct + 1; }
        Q.u = u; Q.v = v;
        Q.u_right = u - C.bf * invzc;
        Q.angle = A.angle[g];
        Q.valid = 1;
        const uint4* d = reinterpret_cast<const uint4*>(A.desc + 32 * (size_t)g);
        const uint4 d0 = d[0], d1 = d[1];
        *reinterpret_cast<uint4*>(Q.desc) = d0;
        *reinterpret_cast<uint4*>(Q.desc + 16) = d1;
    }
    store_query(queries + g, Q);
}

// ---- SearchLocalPoints: Frame::isInFrustum + PredictScale + the window of the local-map overload, one query per local point -----
__global__ __launch_bounds__(256) void k_track_queries_local(const TrackFrameDev* __restrict__ frames, int n_frames, TrackConst C,
                                                             const LocalPointDev* __restrict__ points, int total_q, MatchQuery* __restrict__ queries,
                                                             int32_t* __restrict__ query_frame, int32_t* __restrict__ match, int32_t* __restrict__ amb_count,
                                                             int32_t* __restrict__ amb_ids, float* __restrict__ amb_ratio, float* __restrict__ amb_r) {
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= total_q) return;
    const int f = frame_of_query(frames, n_frames, g);
    if (f < 0 || g >= frames[f].q_off + frames[f].n_q) return;
    const TrackFrameDev& F = frames[f];
    if (F.slot < 0) { query_frame[g] = -1; return; }
    query_frame[g] = F.slot;
    match[g] = -1;
    MatchQuery Q;
    Q.u = 0; Q.v = 0; Q.radius = 0; Q.u_right = -1; Q.min_level = -1; Q.max_level = -1; Q.angle = 0; Q.valid = 0; Q.has_observations = 1;
#pragma unroll
    for (int k = 0; k < 32; ++k) Q.desc[k] = 0;
    const float* q = F.pose7;
    float Ow[3], R[9];
    {  // Tcw.inverse().translation()
        const float qi[4] = {-q[0], -q[1], -q[2], q[3]};
        const float nt[3] = {q[4] * -1.f, q[5] * -1.f, q[6] * -1.f};
        quat_rotate_f(qi, nt, Ow);
    }
    {  // Eigen::Quaternionf::toRotationMatrix (mRcw of Frame::UpdatePoseMatrices)
        const float tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
        const float twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
        const float txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
        const float tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
        R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
        R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
        R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
    }
    const float* M = reinterpret_cast<const float*>(points + g);  // pos[3], normal[3], min, max, max_raw, then the descriptor
    const float pos[3] = {M[0], M[1], M[2]}, normal[3] = {M[3], M[4], M[5]};
    const float min_distance = M[6], max_distance = M[7], max_distance_raw = M[8];
    float pc[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) pc[r] = ((R[3 * r] * pos[0] + R[3 * r + 1] * pos[1]) + R[3 * r + 2] * pos[2]) + q[4 + r];
    const float pc_dist = sqrtf((pc[0] * pc[0] + pc[1] * pc[1]) + pc[2] * pc[2]);
    const float invz = 1.0f / pc[2];
    bool ok = !(pc[2] < 0.0f);
    const float u = C.cam4[0] * pc[0] / pc[2] + C.cam4[2], v = C.cam4[1] * pc[1] / pc[2] + C.cam4[3];
    if (u < 0.f || u > (float)C.cols) ok = false;
    if (v < 0.f || v > (float)C.rows) ok = false;
    const float po[3] = {pos[0] - Ow[0], pos[1] - Ow[1], pos[2] - Ow[2]};
    const float dist = sqrtf((po[0] * po[0] + po[1] * po[1]) + po[2] * po[2]);
    if (dist < min_distance || dist > max_distance) ok = false;
    const float view_cos = ((po[0] * normal[0] + po[1] * normal[1]) + po[2] * normal[2]) / dist;
    if (view_cos < C.view_cos_limit) ok = false;
    if (C.far_points && pc_dist > C.th_far) ok = false;
    if (ok) {
        // MapPoint::PredictScale: (int)ceil(logf(mfMaxDistance / dist) / mfLogScaleFactor) with the HOST libm's logf, whose last bit no
        // device function reproduces.  The quotient is evaluated in double here: the host's float result lies within 1.4 units in the
        // last place of it (logf: 0.82 ulp, the division: 0.5), so unless the double value is within 4 float ulps of an integer the
        // ceiling is decided.  The others (a point seen from exactly the distance it was created at: ratio = 1.2^k) are listed with their
        // ratio and window factor; the host takes logf of those and a patch kernel writes level and radius (tracking_host.cpp).
        const float ratio = max_distance_raw / dist;
        const double quot = log((double)ratio) / (double)C.log_scale;
        float r = view_cos > 0.998f ? 2.5f : 4.0f;
        if (F.th != 1.0f) r *= F.th;
        int level = 0;
        const double nearest = rint(quot);
        if (!(fabs(quot) < 1e6) || fabs(quot - nearest) <= 4.8e-7 * fmax(1.0, fabs(quot))) {
            const int at = atomicAdd(amb_count, 1);
            amb_ids[at] = g; amb_ratio[at] = ratio; amb_r[at] = r;
        } else {
            level = (int)ceil(quot);
            if (level < 0) level = 0; else if (level >= C.n_levels) level = C.n_levels - 1;
        }
        Q.u = u; Q.v = v;
        Q.u_right = u - C.bf * invz;
        Q.radius = r * C.scale[level];
        Q.min_level = level - 1; Q.max_level = level;
        Q.valid = 1;
        const uint32_t* d = reinterpret_cast<const uint32_t*>(M + 9);  // 68-byte records: 4-byte aligned only
#pragma unroll
        for (int k = 0; k < 8; ++k) reinterpret_cast<uint32_t*>(Q.desc)[k] = d[k];
    }
    store_query(queries + g, Q);
}

// level (from the host's logf) and window of the listed queries
__global__ __launch_bounds__(256) void k_track_patch_levels(const int32_t* __restrict__ ids, const int32_t* __restrict__ levels, const float* __restrict__ r, int n,
                                                            TrackConst C, MatchQuery* __restrict__ queries) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    MatchQuery& Q = queries[ids[k]];
    const int level = levels[k];
    Q.radius = r[k] * C.scale[level];
    Q.min_level = level - 1; Q.max_level = level;
}

// occupied[i] = held[i] == 1: the keypoint holds a map point with observations before the search (ORBmatcher.cc:100-102)
__global__ __launch_bounds__(256) void k_track_occupied(const uint8_t* __restrict__ held, size_t n, uint8_t* __restrict__ occ) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) occ[i] = held[i] == 1;
}

// ---- after the search: rotation consistency (last-frame overload) and nmatches; one workgroup per frame of the pass -------------
__global__ __launch_bounds__(256) void k_track_count(const TrackFrameDev* __restrict__ frames, const int32_t* __restrict__ pass_frames,
                                                     const MatchQuery* __restrict__ queries, const float* __restrict__ key_angles, int check_orientation,
                                                     int32_t* __restrict__ match, int32_t* __restrict__ n_matches) {
    const int f = pass_frames ? pass_frames[blockIdx.x] : (int)blockIdx.x;
    const TrackFrameDev& F = frames[f];
    __shared__ int s_count[30];
    __shared__ int s_ind[3];
    __shared__ int s_nm;
    const int tid = threadIdx.x;
    if (tid < 30) s_count[tid] = 0;
    if (tid == 0) s_nm = 0;
    __syncthreads();
    const float factor = 1.0f / 30;
    int nm = 0;
    for (int q = tid; q < F.n_q; q += 256) {
        int m = match[F.q_off + q];
        if (F.n_keys == 0) { m = -1; match[F.q_off + q] = -1; }
        if (m < 0) continue;
        ++nm;
        if (check_orientation) {
            float rot = queries[F.q_off + q].angle - key_angles[F.key_off + m];
            if (rot < 0.0) rot += 360.0f;
            int bin = (int)roundf(rot * factor);
            if (bin == 30) bin = 0;
            atomicAdd(&s_count[bin], 1);
        }
    }
    atomicAdd(&s_nm, nm);
    __syncthreads();
    if (!check_orientation) { if (tid == 0) n_matches[f] = s_nm; return; }
    if (tid == 0) {  // ORBmatcher::ComputeThreeMaxima
        int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
        for (int i = 0; i < 30; i++) {
            const int s = s_count[i];
            if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
            else if (s > max3) { max3 = s; ind3 = i; }
        }
        if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
        else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
        s_ind[0] = ind1; s_ind[1] = ind2; s_ind[2] = ind3;
    }
    __syncthreads();
    int removed = 0;
    for (int q = tid; q < F.n_q; q += 256) {
        const int m = match[F.q_off + q];
        if (m < 0) continue;
        float rot = queries[F.q_off + q].angle - key_angles[F.key_off + m];
        if (rot < 0.0) rot += 360.0f;
        int bin = (int)roundf(rot * factor);
        if (bin == 30) bin = 0;
        if (bin != s_ind[0] && bin != s_ind[1] && bin != s_ind[2]) { match[F.q_off + q] = -1; ++removed; }
    }
    atomicSub(&s_nm, removed);
    __syncthreads();
    if (tid == 0) n_matches[f] = s_nm;
}

// ---- TrackWithMotionModel: mvpMapPoints[i] = the last-frame point keypoint i now holds; with at least 20 matches the edge list of
// Optimizer::PoseOptimization in keypoint order (the frame's slots: capacity * f ...), else no edges ---------------------------------
__global__ __launch_bounds__(256) void k_track_edges_last(const TrackFrameDev* __restrict__ frames, TrackConst C, const MatchKey* __restrict__ keys,
                                                          const float* __restrict__ u_right, const int32_t* __restrict__ match,
                                                          const int32_t* __restrict__ n_matches, const float* __restrict__ last_Xw, int32_t* __restrict__ mp_of_key,
                                                          PoseProblem* __restrict__ probs, BaEdge* __restrict__ edges, double* __restrict__ Xw,
                                                          int32_t* __restrict__ edge_kp, double* __restrict__ poses) {
    const int f = blockIdx.x, tid = threadIdx.x;
    const TrackFrameDev& F = frames[f];
    const size_t base = (size_t)f * C.capacity;
    int32_t* mp = mp_of_key + base;
    __shared__ int s_wave[4];
    for (int i = tid; i < C.capacity; i += 256) mp[i] = -1;
    if (tid < 7) poses[7 * f + tid] = (double)F.pose7[tid];
    __syncthreads();
    for (int q = tid; q < F.n_q; q += 256) {
        const int m = match[F.q_off + q];
        if (m >= 0) atomicMax(&mp[m], q);  // the host loop writes in query order: the largest query index stays
    }
    __syncthreads();
    if (n_matches[f] < 20) { if (tid == 0) probs[f] = PoseProblem{(int32_t)base, 0}; return; }
    int carry = 0;
    for (int i0 = 0; i0 < F.n_keys; i0 += 256) {
        const int i = i0 + tid;
        const int q = i < F.n_keys ? mp[i] : -1;
        int tot;
        const int e = carry + block_excl_scan_256(q >= 0 ? 1 : 0, s_wave, tot);
        if (q >= 0) {
            const MatchKey k = keys[F.key_off + i];
            BaEdge ed;
            ed.point = e; ed.pose = 0;
            ed.u = (double)k.x; ed.v = (double)k.y; ed.ur = (double)u_right[base + i];
            ed.info = (double)C.inv_sigma2[k.octave];
            edges[base + e] = ed;
            const float* X = last_Xw + 3 * (size_t)(F.q_off + q);
            Xw[3 * (base + e)] = (double)X[0]; Xw[3 * (base + e) + 1] = (double)X[1]; Xw[3 * (base + e) + 2] = (double)X[2];
            edge_kp[base + e] = i;
        }
        carry += tot;
    }
    if (tid == 0) probs[f] = PoseProblem{(int32_t)base, carry};
}

// Tracking.cc:2798-2822: outliers lose their map point; a frame with fewer than 20 matches keeps the predicted pose (n_inliers = -1)
__global__ __launch_bounds__(256) void k_track_finish_last(const TrackFrameDev* __restrict__ frames, int capacity, const int32_t* __restrict__ n_matches,
                                                           const PoseProblem* __restrict__ probs, const uint8_t* __restrict__ outlier,
                                                           const int32_t* __restrict__ edge_kp, const int32_t* __restrict__ inliers, int32_t* __restrict__ mp_of_key,
                                                           double* __restrict__ poses, int32_t* __restrict__ n_inliers) {
    const int f = blockIdx.x, tid = threadIdx.x;
    const size_t base = (size_t)f * capacity;
    if (n_matches[f] < 20) {
        if (tid < 7) poses[7 * f + tid] = (double)frames[f].pose7[tid];
        if (tid == 0) n_inliers[f] = -1;
        return;
    }
    if (tid == 0) n_inliers[f] = inliers[f];
    const int n = probs[f].n;
    for (int e = tid; e < n; e += 256)
        if (outlier[base + e]) mp_of_key[base + edge_kp[base + e]] = -1;
}

// ---- TrackLocalMap: F.mvpMapPoints[bestIdx] = pMP for the matched local points, then one edge per keypoint that holds a point
// (held before, or matched now), in keypoint order --------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_track_edges_local(const TrackFrameDev* __restrict__ frames, TrackConst C, const MatchKey* __restrict__ keys,
                                                           const float* __restrict__ u_right, const int32_t* __restrict__ match, const uint8_t* __restrict__ held,
                                                           const float* __restrict__ held_Xw, const LocalPointDev* __restrict__ points,
                                                           int32_t* __restrict__ local_of_key, PoseProblem* __restrict__ probs, BaEdge* __restrict__ edges,
                                                           double* __restrict__ Xw, int32_t* __restrict__ edge_kp, double* __restrict__ poses) {
    const int f = blockIdx.x, tid = threadIdx.x;
    const TrackFrameDev& F = frames[f];
    const size_t base = (size_t)f * C.capacity;
    int32_t* lk = local_of_key + base;
    __shared__ int s_wave[4];
    for (int i = tid; i < C.capacity; i += 256) lk[i] = -1;
    if (tid < 7) poses[7 * f + tid] = (double)F.pose7[tid];
    __syncthreads();
    for (int q = tid; q < F.n_q; q += 256) {
        const int m = match[F.q_off + q];
        if (m >= 0) atomicMax(&lk[m], q);
    }
    __syncthreads();
    int carry = 0;
    for (int i0 = 0; i0 < F.n_keys; i0 += 256) {
        const int i = i0 + tid;
        const bool in = i < F.n_keys;
        const int q = in ? lk[i] : -1;
        const bool has = in && (held[base + i] != 0 || q >= 0);
        int tot;
        const int e = carry + block_excl_scan_256(has ? 1 : 0, s_wave, tot);
        if (has) {
            const MatchKey k = keys[F.key_off + i];
            BaEdge ed;
            ed.point = e; ed.pose = 0;
            ed.u = (double)k.x; ed.v = (double)k.y; ed.ur = (double)u_right[base + i];
            ed.info = (double)C.inv_sigma2[k.octave];
            edges[base + e] = ed;
            const float* X = q >= 0 ? reinterpret_cast<const float*>(points + F.q_off + q) : held_Xw + 3 * (base + i);
            Xw[3 * (base + e)] = (double)X[0]; Xw[3 * (base + e) + 1] = (double)X[1]; Xw[3 * (base + e) + 2] = (double)X[2];
            edge_kp[base + e] = i;
        }
        carry += tot;
    }
    if (tid == 0) probs[f] = PoseProblem{(int32_t)base, carry};
}

// mvbOutlier per keypoint and mnMatchesInliers (Tracking.cc:3192-3227): not an outlier and Observations() > 0 (held == 2 is a point
// without observations; local points have them)
__global__ __launch_bounds__(256) void k_track_finish_local(const TrackFrameDev* __restrict__ frames, int capacity, const PoseProblem* __restrict__ probs,
                                                            const uint8_t* __restrict__ outlier, const int32_t* __restrict__ edge_kp, const uint8_t* __restrict__ held,
                                                            const int32_t* __restrict__ local_of_key, uint8_t* __restrict__ outlier_of_key,
                                                            int32_t* __restrict__ n_inliers) {
    const int f = blockIdx.x, tid = threadIdx.x;
    const size_t base = (size_t)f * capacity;
    __shared__ int s_good;
    if (tid == 0) s_good = 0;
    for (int i = tid; i < capacity; i += 256) outlier_of_key[base + i] = 0;
    __syncthreads();
    const int n = probs[f].n;
    int good = 0;
    for (int e = tid; e < n; e += 256) {
        const int i = edge_kp[base + e];
        const uint8_t o = outlier[base + e];
        outlier_of_key[base + i] = o;
        if (!o && (local_of_key[base + i] >= 0 || held[base + i] == 1)) ++good;
    }
    atomicAdd(&s_good, good);
    __syncthreads();
    if (tid == 0) n_inliers[f] = s_good;
}

// ---- launchers ------------------------------------------------------------------------------------------------------------------
void launch_track_queries_last(const TrackFrameDev* frames, int n_frames, const TrackConst& C, const LastFrameArrays& A, int total_q, MatchQuery* queries,
                               int32_t* query_frame, int32_t* match, hipStream_t st) {
    if (total_q > 0) TC2LI_LAUNCH(k_track_queries_last, dim3((total_q + 255) / 256), dim3(256), 0, st, frames, n_frames, C, A, total_q, queries, query_frame, match);
}
void launch_track_queries_local(const TrackFrameDev* frames, int n_frames, const TrackConst& C, const LocalPointDev* points, int total_q, MatchQuery* queries,
                                int32_t* query_frame, int32_t* match, int32_t* amb_count, int32_t* amb_ids, float* amb_ratio, float* amb_r, hipStream_t st) {
    if (total_q > 0)
        TC2LI_LAUNCH(k_track_queries_local, dim3((total_q + 255) / 256), dim3(256), 0, st, frames, n_frames, C, points, total_q, queries, query_frame, match, amb_count,
                     amb_ids, amb_ratio, amb_r);
}
void launch_track_patch_levels(const int32_t* ids, const int32_t* levels, const float* r, int n, const TrackConst& C, MatchQuery* queries, hipStream_t st) {
    if (n > 0) TC2LI_LAUNCH(k_track_patch_levels, dim3((n + 255) / 256), dim3(256), 0, st, ids, levels, r, n, C, queries);
}
void launch_track_occupied(const uint8_t* held, size_t n, uint8_t* occ, hipStream_t st) {
    if (n > 0) TC2LI_LAUNCH(k_track_occupied, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, held, n, occ);
}
void launch_track_count(const TrackFrameDev* frames, const int32_t* pass_frames, int n_pass, const MatchQuery* queries, const float* key_angles,
                        int check_orientation, int32_t* match, int32_t* n_matches, hipStream_t st) {
    if (n_pass > 0) TC2LI_LAUNCH(k_track_count, dim3(n_pass), dim3(256), 0, st, frames, pass_frames, queries, key_angles, check_orientation, match, n_matches);
}
void launch_track_edges_last(const TrackFrameDev* frames, int n_frames, const TrackConst& C, const MatchKey* keys, const float* u_right, const int32_t* match,
                             const int32_t* n_matches, const float* last_Xw, int32_t* mp_of_key, PoseProblem* probs, BaEdge* edges, double* Xw,
                             int32_t* edge_kp, double* poses, hipStream_t st) {
    if (n_frames > 0)
        TC2LI_LAUNCH(k_track_edges_last, dim3(n_frames), dim3(256), 0, st, frames, C, keys, u_right, match, n_matches, last_Xw, mp_of_key, probs, edges, Xw, edge_kp,
                     poses);
}
void launch_track_finish_last(const TrackFrameDev* frames, int n_frames, int capacity, const int32_t* n_matches, const PoseProblem* probs, const uint8_t* outlier,
                              const int32_t* edge_kp, const int32_t* inliers, int32_t* mp_of_key, double* poses, int32_t* n_inliers, hipStream_t st) {
    if (n_frames > 0)
        TC2LI_LAUNCH(k_track_finish_last, dim3(n_frames), dim3(256), 0, st, frames, capacity, n_matches, probs, outlier, edge_kp, inliers, mp_of_key, poses, n_inliers);
}
void launch_track_edges_local(const TrackFrameDev* frames, int n_frames, const TrackConst& C, const MatchKey* keys, const float* u_right, const int32_t* match,
                              const uint8_t* held, const float* held_Xw, const LocalPointDev* points, int32_t* local_of_key, PoseProblem* probs, BaEdge* edges,
                              double* Xw, int32_t* edge_kp, double* poses, hipStream_t st) {
    if (n_frames > 0)
        TC2LI_LAUNCH(k_track_edges_local, dim3(n_frames), dim3(256), 0, st, frames, C, keys, u_right, match, held, held_Xw, points, local_of_key, probs, edges, Xw,
                     edge_kp, poses);
}
void launch_track_finish_local(const TrackFrameDev* frames, int n_frames, int capacity, const PoseProblem* probs, const uint8_t* outlier, const int32_t* edge_kp,
                               const uint8_t* held, const int32_t* local_of_key, uint8_t* outlier_of_key, int32_t* n_inliers, hipStream_t st) {
    if (n_frames > 0)
        TC2LI_LAUNCH(k_track_finish_local, dim3(n_frames), dim3(256), 0, st, frames, capacity, probs, outlier, edge_kp, held, local_of_key, outlier_of_key, n_inliers);
}

}  // namespace tc2li
