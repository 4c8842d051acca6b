// Local-map bookkeeping on flat arrays -- Tracking::UpdateLocalKeyFrames and Tracking::UpdateLocalPoints
// (SF/src/Tracking.cc:3296-3476).  Integer work only; every result is order-exact:
//   k_lm_votes      one thread per frame point: keyframeCounter[kf]++ over the point's observations (integer atomics)
//   k_lm_keyframes  one workgroup: the voted keyframes in index order (ordered compaction), the first keyframe with the most
//                   votes, then -- one lane, it is a chain of at most 80 dependent steps -- the neighbour / child / parent and
//                   temporal extensions, and the offsets of every local keyframe's matches in the reversed concatenation
//   k_lm_first      one thread per match slot of the local keyframes: the first position at which a point occurs (atomicMin)
//   k_lm_count / k_lm_scatter   ordered compaction of the first occurrences = mvpLocalMapPoints in the reference's order
#include <hip/hip_runtime.h>

#include "launch.hpp"
#include <stdint.h>

#include "localmap_device.hpp"

namespace tc2li {

__global__ __launch_bounds__(256) void k_lm_votes(LocalMapDev m, const int32_t* __restrict__ frame_points, int n, uint8_t* __restrict__ cleared) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int p = frame_points[i];
    uint8_t c = 0;
    if (p >= 0) {
        if (!m.point_bad[p]) {
            for (int k = m.obs_off[p]; k < m.obs_off[p + 1]; ++k) atomicAdd(m.votes + m.obs_kf[k], 1);
        } else {
            c = 1;
        }
    }
    cleared[i] = c;
}

// inclusive scan of one int per thread over a 1024-thread workgroup; returns the exclusive prefix, *total = the block's sum
__device__ __forceinline__ int block_scan_1024(int v, int* s_wave, int* total) {
    const int lane = threadIdx.x & 63, wave = wave_in_block();
    int incl = v;
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += s_wave[w];
    int tot = 0;
    for (int w = 0; w < 16; ++w) tot += s_wave[w];
    __syncthreads();
    *total = tot;
    return base + incl - v;
}

__global__ __launch_bounds__(1024) void k_lm_keyframes(LocalMapDev m, int temporal_last_kf) {
    __shared__ int s_wave[16];
    __shared__ unsigned long long s_best;
    if (threadIdx.x == 0) s_best = 0ull;
    __syncthreads();
    // ---- the voted, not-bad keyframes in index order; the first one with the most votes ----
    int n_list = 0;
    for (int chunk = 0; chunk < m.n_keyframes; chunk += 1024) {
        const int kf = chunk + (int)threadIdx.x;
        const int v = kf < m.n_keyframes ? m.votes[kf] : 0;
        const int keep = v > 0 && !m.kf_bad[kf];
        int tot;
        const int at = block_scan_1024(keep, s_wave, &tot);
        if (keep) {
            m.kf_list[n_list + at] = kf;
            m.marked[kf] = 1;
            // more votes win; among equal votes the lower index (the reference's `>` keeps the first it meets)
            atomicMax(&s_best, ((unsigned long long)(unsigned)v << 32) | (unsigned)(0x7fffffff - kf));
        }
        n_list += tot;
    }
    __threadfence_block();
    __syncthreads();
    if (threadIdx.x != 0) return;
    // ---- extensions: a chain of dependent steps over the voted keyframes only (the loop's end iterator is taken before it appends) ----
    const int n_voted = n_list;
    for (int j = 0; j < n_voted; ++j) {
        if (n_list > 80) break;
        const int kf = m.kf_list[j];
        const int ce = min(m.covis_off[kf + 1], m.covis_off[kf] + 10);  // GetBestCovisibilityKeyFrames(10)
        for (int k = m.covis_off[kf]; k < ce; ++k) {
            const int n = m.covis[k];
            if (!m.kf_bad[n] && !m.marked[n]) { m.kf_list[n_list++] = n; m.marked[n] = 1; break; }
        }
        for (int k = m.child_off[kf]; k < m.child_off[kf + 1]; ++k) {
            const int c = m.children[k];
            if (!m.kf_bad[c] && !m.marked[c]) { m.kf_list[n_list++] = c; m.marked[c] = 1; break; }
        }
        const int par = m.parent[kf];
        if (par >= 0 && !m.marked[par]) { m.kf_list[n_list++] = par; m.marked[par] = 1; break; }  // the reference's break leaves the keyframe loop
    }
    if (temporal_last_kf >= 0 && n_list < 80) {
        int t = temporal_last_kf;
        for (int i = 0; i < 20 && t >= 0; ++i)
            if (!m.marked[t]) { m.kf_list[n_list++] = t; m.marked[t] = 1; t = m.prev_kf[t]; }
    }
    // ---- UpdateLocalPoints walks the local keyframes backwards: offsets of their match lists in that concatenation ----
    int total = 0;
    for (int jr = 0; jr < n_list; ++jr) {
        const int kf = m.kf_list[n_list - 1 - jr];
        m.rev_base[jr] = total;
        total += m.match_off[kf + 1] - m.match_off[kf];
    }
    m.rev_base[n_list] = total;
    m.header[0] = n_list;
    m.header[1] = total;
    m.header[2] = s_best ? 0x7fffffff - (int)(unsigned)(s_best & 0xffffffffull) : -1;
}

// entry `pos` of the reversed concatenation -> the map point there, or -1 (empty slot / bad point)
__device__ __forceinline__ int entry_point(const LocalMapDev& m, int n_local, int pos) {
    int lo = 0, hi = n_local;  // rev_base[lo] <= pos < rev_base[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (m.rev_base[mid] <= pos) lo = mid; else hi = mid;
    }
    const int kf = m.kf_list[n_local - 1 - lo];
    const int p = m.matches[m.match_off[kf] + (pos - m.rev_base[lo])];
    return p >= 0 && !m.point_bad[p] ? p : -1;
}

__global__ __launch_bounds__(256) void k_lm_first(LocalMapDev m, int n_local, int total) {
    const int pos = blockIdx.x * 256 + threadIdx.x;
    if (pos >= total) return;
    const int p = entry_point(m, n_local, pos);
    if (p >= 0) atomicMin(m.first_pos + p, pos);
}

__global__ __launch_bounds__(256) void k_lm_count(LocalMapDev m, int n_local, int total) {
    __shared__ int s_cnt[4];
    const int pos = blockIdx.x * 256 + threadIdx.x;
    int keep = 0;
    if (pos < total) {
        const int p = entry_point(m, n_local, pos);
        keep = p >= 0 && m.first_pos[p] == pos;
    }
    const unsigned long long b = __ballot(keep);
    if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = __popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) m.block_counts[blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}

__global__ __launch_bounds__(256) void k_lm_scatter(LocalMapDev m, int n_local, int total) {
    __shared__ int s_part[256];
    __shared__ int s_cnt[4];
    // this workgroup's start = the kept entries of all earlier workgroups
    int a = 0;
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += 256) a += m.block_counts[b];
    s_part[threadIdx.x] = a;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if ((int)threadIdx.x < s) s_part[threadIdx.x] += s_part[threadIdx.x + s];
        __syncthreads();
    }
    const int base = s_part[0];
    const int pos = blockIdx.x * 256 + threadIdx.x;
    int keep = 0, p = -1;
    if (pos < total) {
        p = entry_point(m, n_local, pos);
        keep = p >= 0 && m.first_pos[p] == pos;
    }
    const unsigned long long bal = __ballot(keep);
    const int lane = threadIdx.x & 63, wave = wave_in_block();
    if (lane == 0) s_cnt[wave] = __popcll(bal);
    __syncthreads();
    int off = base;
    for (int w = 0; w < wave; ++w) off += s_cnt[w];
    if (keep) m.points[off + __popcll(bal & ((1ull << lane) - 1ull))] = p;
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) m.header[3] = base + s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}

void launch_local_map_votes(const LocalMapDev& m, const int32_t* frame_points, int n, uint8_t* cleared, hipStream_t st) {
    if (n > 0) TC2LI_LAUNCH(k_lm_votes, dim3((n + 255) / 256), dim3(256), 0, st, m, frame_points, n, cleared);
}
void launch_local_map_keyframes(const LocalMapDev& m, int temporal_last_kf, hipStream_t st) {
    TC2LI_LAUNCH(k_lm_keyframes, dim3(1), dim3(1024), 0, st, m, temporal_last_kf);
}
void launch_local_map_points(const LocalMapDev& m, int n_local, int total, hipStream_t st) {
    const int blocks = (total + 255) / 256;
    TC2LI_LAUNCH(k_lm_first, dim3(blocks), dim3(256), 0, st, m, n_local, total);
    TC2LI_LAUNCH(k_lm_count, dim3(blocks), dim3(256), 0, st, m, n_local, total);
    TC2LI_LAUNCH(k_lm_scatter, dim3(blocks), dim3(256), 0, st, m, n_local, total);
}

}  // namespace tc2li
