// Projection-guided matching of the tracking thread on gfx950 -- the loop bodies shared by
// ORBmatcher::SearchByProjection(Frame&, const Frame&, th, bMono) (SF/src/ORBmatcher.cc:1685-1896) and
// ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th, ...) (:52-222), with Frame::GetFeaturesInArea
// (SF/src/Frame.cc:687-753) on the 64x48 feature grid (AssignFeaturesToGrid / PosInGrid, :412-443,755-765).
//
// The reference loop is sequential: a keypoint matched by an earlier map point is skipped by later ones.  One
// workgroup per frame reproduces exactly that result by fixed-point iteration: in every round all queries are
// evaluated in parallel against "keypoints claimed in the previous round by a query with a smaller index"; query q's
// answer depends only on the answers of queries < q, so after round k the first k queries are final and the
// iteration stops at the first round that changes nothing (unique fixed point = the sequential result).
#include <hip/hip_runtime.h>

#include "launch.hpp"
#pragma clang fp contract(off)
#include <stdint.h>

#include "matcher_device.hpp"

namespace tc2li {

constexpr int kGridCols = 64, kGridRows = 48, kCells = kGridCols * kGridRows;
constexpr int kThreads = 512;

__global__ __launch_bounds__(kThreads) void k_match_by_projection(const MatchFrameDev* __restrict__ frames, int mode, float nn_ratio,
                                                                 int32_t* __restrict__ match_of_query, int32_t* __restrict__ prev_claim,
                                                                 int32_t* __restrict__ rounds_out) {
    __shared__ int s_cell_start[kCells + 1];
    __shared__ uint16_t s_items[kMaxMatchKeys];
    __shared__ float s_x[kMaxMatchKeys], s_y[kMaxMatchKeys], s_ur[kMaxMatchKeys];
    __shared__ uint8_t s_oct[kMaxMatchKeys], s_occ[kMaxMatchKeys];
    __shared__ int s_claim[2][kMaxMatchKeys];
    __shared__ int s_tmp[64];
    const MatchFrameDev fr = global_record(frames[blockIdx.x]);
    const int tid = threadIdx.x, N = fr.n_keys, M = fr.n_queries;
    const float invW = (float)kGridCols / (fr.max_x - fr.min_x), invH = (float)kGridRows / (fr.max_y - fr.min_y);
    int32_t* out = match_of_query + fr.query_off;
    int32_t* prev = prev_claim + fr.query_off;

    // ---- feature grid ----
    for (int c = tid; c <= kCells; c += kThreads) s_cell_start[c] = 0;
    __syncthreads();
    int* cell_of = s_claim[1];  // scratch until the matching rounds start
    for (int i = tid; i < N; i += kThreads) {
        const MatchKey k = fr.keys[i];
        s_x[i] = k.x; s_y[i] = k.y; s_oct[i] = (uint8_t)k.octave;
        s_ur[i] = fr.u_right[i];
        s_occ[i] = fr.occupied ? fr.occupied[i] : 0;
        const int px = (int)roundf((k.x - fr.min_x) * invW), py = (int)roundf((k.y - fr.min_y) * invH);
        const bool in = !(px < 0 || px >= kGridCols || py < 0 || py >= kGridRows);
        const int c = in ? px * kGridRows + py : -1;
        cell_of[i] = c;
        if (in) atomicAdd(&s_cell_start[c + 1], 1);
    }
    __syncthreads();
    // exclusive scan over the cell counts (one wavefront, 48 cells per lane)
    if (tid < 64) {
        constexpr int per = kCells / 64;
        int sum = 0;
        for (int k = 0; k < per; ++k) sum += s_cell_start[1 + tid * per + k];
        int incl = sum;
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (tid >= o) incl += t; }
        int run = incl - sum;
        for (int k = 0; k < per; ++k) { const int c = s_cell_start[1 + tid * per + k]; s_cell_start[1 + tid * per + k] = run + c; run += c; }
    }
    __syncthreads();
    // fill (cursor = claim[0] scratch), then restore ascending keypoint order inside each cell
    int* cursor = s_claim[0];
    for (int c = tid; c < kCells; c += kThreads) cursor[c] = s_cell_start[c];
    __syncthreads();
    for (int i = tid; i < N; i += kThreads) {
        const int c = cell_of[i];
        if (c >= 0) s_items[atomicAdd(&cursor[c], 1)] = (uint16_t)i;
    }
    __syncthreads();
    for (int c = tid; c < kCells; c += kThreads) {
        const int b = s_cell_start[c], e = s_cell_start[c + 1];
        for (int a = b + 1; a < e; ++a) {
            const uint16_t key = s_items[a];
            int k = a - 1;
            while (k >= b && s_items[k] > key) { s_items[k + 1] = s_items[k]; --k; }
            s_items[k + 1] = key;
        }
    }
    __syncthreads();
    for (int i = tid; i < N; i += kThreads) { s_claim[0][i] = 0x7fffffff; s_claim[1][i] = 0x7fffffff; }
    for (int q = tid; q < M; q += kThreads) prev[q] = -2;
    __syncthreads();

    const uint32_t* D = reinterpret_cast<const uint32_t*>(fr.desc);
    int cur = 0, round = 0;
    for (;;) {
        int* claim_prev = s_claim[cur];
        int* claim_new = s_claim[cur ^ 1];
        int changed = 0;
        for (int q = tid; q < M; q += kThreads) {
            const MatchQuery Q = fr.queries[q];
            int claim = -1;
            if (Q.valid) {
                // Frame::GetFeaturesInArea(u, v, r, minLevel, maxLevel)
                const float r = Q.radius;
                const int minCX = max(0, (int)floorf((Q.u - fr.min_x - r) * invW));
                const int maxCX = min(kGridCols - 1, (int)ceilf((Q.u - fr.min_x + r) * invW));
                const int minCY = max(0, (int)floorf((Q.v - fr.min_y - r) * invH));
                const int maxCY = min(kGridRows - 1, (int)ceilf((Q.v - fr.min_y + r) * invH));
                const bool any = !(minCX >= kGridCols || maxCX < 0 || minCY >= kGridRows || maxCY < 0);
                const bool check_levels = (Q.min_level > 0) || (Q.max_level >= 0);
                const uint32_t* qd = reinterpret_cast<const uint32_t*>(Q.desc);
                int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
                if (any)
                    for (int ix = minCX; ix <= maxCX; ++ix)
                        for (int iy = minCY; iy <= maxCY; ++iy) {
                            const int c = ix * kGridRows + iy;
                            for (int k = s_cell_start[c]; k < s_cell_start[c + 1]; ++k) {
                                const int idx = s_items[k];
                                const int oct = s_oct[idx];
                                if (check_levels) {
                                    if (oct < Q.min_level) continue;
                                    if (Q.max_level >= 0 && oct > Q.max_level) continue;
                                }
                                if (!(fabsf(s_x[idx] - Q.u) < r && fabsf(s_y[idx] - Q.v) < r)) continue;
                                if (s_occ[idx] || claim_prev[idx] < q) continue;  // already matched by an earlier point
                                if (s_ur[idx] > 0) {
                                    if (fabsf(Q.u_right - s_ur[idx]) > r) continue;
                                }
                                const uint32_t* kd = D + (size_t)idx * 8;
                                int dist = 0;
#pragma unroll
                                for (int w = 0; w < 8; ++w) dist += __popc(qd[w] ^ kd[w]);
                                if (mode == 0) {
                                    if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
                                } else if (dist < bestDist) {
                                    bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = oct; bestIdx = idx;
                                } else if (dist < bestDist2) {
                                    bestLevel2 = oct; bestDist2 = dist;
                                }
                            }
                        }
                if (bestDist <= 100) {  // TH_HIGH
                    bool ok = true;
                    if (mode != 0) {
                        if (bestLevel == bestLevel2 && (float)bestDist > nn_ratio * (float)bestDist2) ok = false;
                    }
                    if (ok) claim = bestIdx;
                }
            }
            if (claim >= 0 && Q.has_observations) atomicMin(&claim_new[claim], q);
            if (prev[q] != claim) { changed = 1; prev[q] = claim; }
            out[q] = claim;
        }
        const int any_changed = __syncthreads_or(changed);
        ++round;
        if (!any_changed || round > M + 1) break;
        for (int i = tid; i < N; i += kThreads) claim_prev[i] = 0x7fffffff;  // becomes the next round's "new"
        cur ^= 1;
        __syncthreads();
    }
    if (tid == 0) rounds_out[blockIdx.x] = round;
    (void)s_tmp;
}

// ---- the same result in three launches -----------------------------------------------------------------------------------
// k_match_by_projection recomputes every Hamming distance in every round with one workgroup per frame.  Everything except
// "claimed by an earlier query" is independent of the rounds, so:
//   k_match_grid        one workgroup per frame: the 64x48 feature grid (cell starts, keypoint indices ascending per cell) in HBM
//   k_match_candidates  one thread per query over the whole batch: the candidates that pass the static tests of the loop
//                       body, in scan order, with their descriptor distance (idx | dist << 12 | octave << 21), into a pool
//   k_match_resolve     one workgroup per frame: the fixed-point rounds walk the short candidate lists only

__global__ __launch_bounds__(kThreads) void k_match_grid(const MatchFrameDev* __restrict__ frames, MatchLists L) {
    __shared__ int s_cell_start[kCells + 1];
    __shared__ int s_cursor[kCells];
    __shared__ int s_cell_of[kMaxMatchKeys];
    __shared__ uint16_t s_items[kMaxMatchKeys];
    const MatchFrameDev fr = global_record(frames[blockIdx.x]);
    const int tid = threadIdx.x, N = fr.n_keys;
    const float invW = (float)kGridCols / (fr.max_x - fr.min_x), invH = (float)kGridRows / (fr.max_y - fr.min_y);
    for (int c = tid; c <= kCells; c += kThreads) s_cell_start[c] = 0;
    __syncthreads();
    for (int i = tid; i < N; i += kThreads) {
        const MatchKey k = fr.keys[i];
        const int px = (int)roundf((k.x - fr.min_x) * invW), py = (int)roundf((k.y - fr.min_y) * invH);
        const bool in = !(px < 0 || px >= kGridCols || py < 0 || py >= kGridRows);
        const int c = in ? px * kGridRows + py : -1;
        s_cell_of[i] = c;
        if (in) atomicAdd(&s_cell_start[c + 1], 1);
    }
    __syncthreads();
    if (tid < 64) {
        constexpr int per = kCells / 64;
        int sum = 0;
        for (int k = 0; k < per; ++k) sum += s_cell_start[1 + tid * per + k];
        int incl = sum;
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (tid >= o) incl += t; }
        int run = incl - sum;
        for (int k = 0; k < per; ++k) { const int c = s_cell_start[1 + tid * per + k]; s_cell_start[1 + tid * per + k] = run + c; run += c; }
    }
    __syncthreads();
    for (int c = tid; c < kCells; c += kThreads) s_cursor[c] = s_cell_start[c];
    __syncthreads();
    for (int i = tid; i < N; i += kThreads) {
        const int c = s_cell_of[i];
        if (c >= 0) s_items[atomicAdd(&s_cursor[c], 1)] = (uint16_t)i;
    }
    __syncthreads();
    for (int c = tid; c < kCells; c += kThreads) {  // ascending keypoint order inside each cell (the order the reference's grid has)
        const int b = s_cell_start[c], e = s_cell_start[c + 1];
        for (int a = b + 1; a < e; ++a) {
            const uint16_t key = s_items[a];
            int k = a - 1;
            while (k >= b && s_items[k] > key) { s_items[k + 1] = s_items[k]; --k; }
            s_items[k + 1] = key;
        }
    }
    __syncthreads();
    int32_t* cs = L.cell_start + (size_t)blockIdx.x * (kCells + 1);
    uint16_t* it = L.items + L.key_base[blockIdx.x];
    for (int c = tid; c <= kCells; c += kThreads) cs[c] = s_cell_start[c];
    const int n_in = s_cell_start[kCells];
    for (int i = tid; i < n_in; i += kThreads) it[i] = s_items[i];
}

// the static part of the loop body for one candidate: level window, search window, occupancy, stereo coordinate
__device__ __forceinline__ bool candidate_ok(const MatchFrameDev& fr, const MatchQuery& Q, bool check_levels, int idx, int& oct) {
    const MatchKey k = fr.keys[idx];
    oct = k.octave;
    if (check_levels) {
        if (oct < Q.min_level) return false;
        if (Q.max_level >= 0 && oct > Q.max_level) return false;
    }
    const float r = Q.radius;
    if (!(fabsf(k.x - Q.u) < r && fabsf(k.y - Q.v) < r)) return false;
    if (fr.occupied && fr.occupied[idx]) return false;
    const float ur = fr.u_right[idx];
    if (ur > 0) {
        if (fabsf(Q.u_right - ur) > r) return false;
    }
    return true;
}

__global__ __launch_bounds__(256) void k_match_candidates(const MatchFrameDev* __restrict__ frames, int nframes, const int32_t* __restrict__ query_frame,
                                                          int total_q, MatchLists L) {
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= total_q) return;
    const int f = query_frame[g];
    if (f < 0) { L.cand_cnt[g] = 0; L.cand_off[g] = 0; return; }
    const MatchFrameDev fr = global_record(frames[f]);
    const MatchQuery Q = fr.queries[g - fr.query_off];
    const int flags = Q.has_observations ? (1 << 30) : 0;
    if (!Q.valid || fr.n_keys == 0) { L.cand_cnt[g] = flags; L.cand_off[g] = 0; return; }
    const float invW = (float)kGridCols / (fr.max_x - fr.min_x), invH = (float)kGridRows / (fr.max_y - fr.min_y);
    const float r = Q.radius;
    const int minCX = max(0, (int)floorf((Q.u - fr.min_x - r) * invW));
    const int maxCX = min(kGridCols - 1, (int)ceilf((Q.u - fr.min_x + r) * invW));
    const int minCY = max(0, (int)floorf((Q.v - fr.min_y - r) * invH));
    const int maxCY = min(kGridRows - 1, (int)ceilf((Q.v - fr.min_y + r) * invH));
    const bool any = !(minCX >= kGridCols || maxCX < 0 || minCY >= kGridRows || maxCY < 0);
    const bool check_levels = (Q.min_level > 0) || (Q.max_level >= 0);
    const int32_t* cs = L.cell_start + (size_t)f * (kCells + 1);
    const uint16_t* items = L.items + L.key_base[f];
    int cnt = 0;
    if (any)
        for (int ix = minCX; ix <= maxCX; ++ix)
            for (int iy = minCY; iy <= maxCY; ++iy) {
                const int c = ix * kGridRows + iy;
                for (int k = cs[c]; k < cs[c + 1]; ++k) {
                    int oct;
                    cnt += candidate_ok(fr, Q, check_levels, items[k], oct) ? 1 : 0;
                }
            }
    int off = 0;
    if (cnt) {
        off = atomicAdd(L.pool_top, cnt);
        if (off + cnt > L.pool_cap) { L.pool_top[1] = 1; cnt = 0; off = 0; }
    }
    L.cand_off[g] = off;
    L.cand_cnt[g] = cnt | flags;
    if (!cnt) return;
    const uint32_t* D = reinterpret_cast<const uint32_t*>(fr.desc);
    const uint32_t* qd = reinterpret_cast<const uint32_t*>(Q.desc);
    uint32_t* out = L.pool + off;
    int w_ = 0;
    for (int ix = minCX; ix <= maxCX; ++ix)
        for (int iy = minCY; iy <= maxCY; ++iy) {
            const int c = ix * kGridRows + iy;
            for (int k = cs[c]; k < cs[c + 1]; ++k) {
                const int idx = items[k];
                int oct;
                if (!candidate_ok(fr, Q, check_levels, idx, oct)) continue;
                const uint32_t* kd = D + (size_t)idx * 8;
                int dist = 0;
#pragma unroll
                for (int w = 0; w < 8; ++w) dist += __popc(qd[w] ^ kd[w]);
                out[w_++] = (uint32_t)idx | ((uint32_t)dist << 12) | ((uint32_t)oct << 21);
            }
        }
}

__global__ __launch_bounds__(kThreads) void k_match_resolve(const MatchFrameDev* __restrict__ frames, int mode, float nn_ratio, MatchLists L,
                                                           int32_t* __restrict__ match_of_query, int32_t* __restrict__ prev_claim,
                                                           int32_t* __restrict__ rounds_out) {
    __shared__ int s_claim[2][kMaxMatchKeys];
    const MatchFrameDev fr = global_record(frames[blockIdx.x]);
    const int tid = threadIdx.x, N = fr.n_keys, M = fr.n_queries;
    int32_t* out = match_of_query + fr.query_off;
    int32_t* prev = prev_claim + fr.query_off;
    const int32_t* coff = L.cand_off + fr.query_off;
    const int32_t* ccnt = L.cand_cnt + fr.query_off;
    for (int i = tid; i < N; i += kThreads) { s_claim[0][i] = 0x7fffffff; s_claim[1][i] = 0x7fffffff; }
    for (int q = tid; q < M; q += kThreads) prev[q] = -2;
    __syncthreads();
    int cur = 0, round = 0;
    for (;;) {
        int* claim_prev = s_claim[cur];
        int* claim_new = s_claim[cur ^ 1];
        int changed = 0;
        for (int q = tid; q < M; q += kThreads) {
            const int cc = ccnt[q], cnt = cc & 0x3fffffff;
            const uint32_t* e = L.pool + coff[q];
            int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
            for (int k = 0; k < cnt; ++k) {
                const uint32_t v = e[k];
                const int idx = v & 0xfff, dist = (v >> 12) & 0x1ff, oct = (v >> 21) & 0xf;
                if (claim_prev[idx] < q) continue;  // already matched by an earlier point
                if (mode == 0) {
                    if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
                } else if (dist < bestDist) {
                    bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = oct; bestIdx = idx;
                } else if (dist < bestDist2) {
                    bestLevel2 = oct; bestDist2 = dist;
                }
            }
            int claim = -1;
            if (bestDist <= 100) {  // TH_HIGH
                bool ok = true;
                if (mode != 0) {
                    if (bestLevel == bestLevel2 && (float)bestDist > nn_ratio * (float)bestDist2) ok = false;
                }
                if (ok) claim = bestIdx;
            }
            if (claim >= 0 && (cc & (1 << 30))) atomicMin(&claim_new[claim], q);
            if (prev[q] != claim) { changed = 1; prev[q] = claim; }
            out[q] = claim;
        }
        const int any_changed = __syncthreads_or(changed);
        ++round;
        if (!any_changed || round > M + 1) break;
        for (int i = tid; i < N; i += kThreads) claim_prev[i] = 0x7fffffff;  // becomes the next round's "new"
        cur ^= 1;
        __syncthreads();
    }
    if (tid == 0) rounds_out[blockIdx.x] = round;
}

void launch_match_grid(const MatchFrameDev* frames, int nframes, const MatchLists& L, hipStream_t st) {
    if (nframes > 0) TC2LI_LAUNCH(k_match_grid, dim3(nframes), dim3(kThreads), 0, st, frames, L);
}

void launch_match_lists(const MatchFrameDev* frames, int nframes, const int32_t* query_frame, int total_q, const MatchLists& L, int mode,
                        float nn_ratio, int32_t* match_of_query, int32_t* prev_claim, int32_t* rounds_out, hipStream_t st) {
    if (nframes <= 0) return;
    (void)hipMemsetAsync(L.pool_top, 0, 2 * sizeof(int32_t), st);
    TC2LI_LAUNCH(k_match_grid, dim3(nframes), dim3(kThreads), 0, st, frames, L);
    TC2LI_LAUNCH(k_match_candidates, dim3((total_q + 255) / 256), dim3(256), 0, st, frames, nframes, query_frame, total_q, L);
    TC2LI_LAUNCH(k_match_resolve, dim3(nframes), dim3(kThreads), 0, st, frames, mode, nn_ratio, L, match_of_query, prev_claim, rounds_out);
}

void launch_match_by_projection(const MatchFrameDev* frames, int nframes, int mode, float nn_ratio, int32_t* match_of_query,
                                int32_t* prev_claim, int32_t* rounds_out, hipStream_t st) {
    if (nframes > 0)
        TC2LI_LAUNCH(k_match_by_projection, dim3(nframes), dim3(kThreads), 0, st, frames, mode, nn_ratio, match_of_query, prev_claim,
                           rounds_out);
}

}  // namespace tc2li
