// libstdc++'s std::sort replayed level by level by a whole workgroup (the derivation is in lidar_kernels.hip, "the time sort of
// ImuProcess::UndistortPcl"): used for the LiDAR time stamps (float keys) and for the closing sort of the keypoint distribution
// (quadtree_kernels.hip: 32-bit keys).  Keys of type K compared with `<` only; idx is the payload that travels with a key.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tc2li {

// One pass over the recursion levels of the ranges of key[0 .. n) that are longer than `stop`: NT threads, arrays of index type T (global
// ints or range-local shorts in LDS).  depth: levels left for the ranges of the first level; dep[first of a range] receives the levels
// left for that range when `dep` is given.  Returns false when a range longer than `stop` remains at depth 0.
template <int NT, typename T, typename K>
__device__ __forceinline__ bool sort_levels(K* __restrict__ key, int* __restrict__ idx, T* __restrict__ sf, T* __restrict__ sl, T* __restrict__ cl,
                                            T* __restrict__ cr, T* __restrict__ lp, T* __restrict__ rp, T* __restrict__ cut, uint8_t* __restrict__ flag,
                                            int* __restrict__ dep, int n, int depth, int stop, int (*s_tot)[NT / 64], int (*s_base)[NT / 64 + 1], int* s_any) {
    constexpr int NW = NT / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_in_block();
    // contiguous ranges of the wavefronts for the prefix counts: C elements each, a multiple of 64
    const int C = ((n + NW - 1) / NW + 63) / 64 * 64;
    bool any = n > stop;
    __syncthreads();
    while (any) {
        if (depth == 0) return false;
        --depth;
        // ---- pivots: __move_median_to_first(first, first + 1, mid, last - 1) ----
        for (int x = tid; x < n; x += NT) {
            if ((int)sf[x] != x) continue;
            const int last = sl[x];
            if (last - x <= stop) continue;
            const int a = x + 1, b = x + (last - x) / 2, c = last - 1;
            const K ka = key[a], kb = key[b], kc = key[c];
            int m;
            if (ka < kb) m = kb < kc ? b : (ka < kc ? c : a);
            else m = ka < kc ? a : (kb < kc ? c : b);
            const K kf = key[x], km = key[m];
            const int jf = idx[x], jm = idx[m];
            key[x] = km; idx[x] = jm; key[m] = kf; idx[m] = jf;
        }
        __syncthreads();
        // ---- candidates of the partitions and their prefix counts along the array ----
        {
            int run_l = 0, run_r = 0;
            const int x0 = wave * C, x1 = min(x0 + C, n);
            for (int xb = x0; xb < x1; xb += 64) {
                const int x = xb + lane;
                bool fl = false, fr = false;
                if (x < x1) {
                    const int f = sf[x];
                    if ((int)sl[x] - f > stop && x != f) {
                        const K p = key[f], k = key[x];
                        fl = !(k < p);
                        fr = !(p < k);
                    }
                }
                const unsigned long long bl = __ballot(fl), br = __ballot(fr), le = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);
                if (x < x1) {
                    cl[x] = (T)(run_l + __popcll(bl & le));
                    cr[x] = (T)(run_r + __popcll(br & le));
                    flag[x] = (uint8_t)((fl ? 1 : 0) | (fr ? 2 : 0));
                }
                run_l += __popcll(bl);
                run_r += __popcll(br);
            }
            if (lane == 0) { s_tot[0][wave] = run_l; s_tot[1][wave] = run_r; }
        }
        __syncthreads();
        if (tid < 2) {
            int acc = 0;
            for (int w = 0; w < NW; ++w) { s_base[tid][w] = acc; acc += s_tot[tid][w]; }
            s_base[tid][NW] = acc;
        }
        if (tid == 0) *s_any = 0;
        __syncthreads();
#define TS_GL(x) ((int)cl[x] + s_base[0][(x) / C])
#define TS_GR(x) ((int)cr[x] + s_base[1][(x) / C])
        // ---- candidates by rank: lp[f + 1 + k] = k-th from the left, rp[f + 1 + k] = k-th from the right ----
        for (int x = tid; x < n; x += NT) {
            const int fg = flag[x];
            if (!fg) continue;
            const int f = sf[x], l = sl[x];
            if (fg & 1) lp[f + 1 + (TS_GL(x) - TS_GL(f) - 1)] = (T)x;
            if (fg & 2) rp[f + 1 + (TS_GR(l - 1) - TS_GR(x))] = (T)x;
        }
        __syncthreads();
        // ---- the swaps and the cut ----
        for (int x = tid; x < n; x += NT) {
            const int f = sf[x], l = sl[x];
            if (l - f <= stop || x == f) continue;
            const int k = x - (f + 1), nl = TS_GL(l - 1) - TS_GL(f), nr = TS_GR(l - 1) - TS_GR(f);
            const int L = k < nl ? (int)lp[x] : 0x7fffffff, R = k < nr ? (int)rp[x] : -1;
            if (L < R) {  // swap number k
                const K k1 = key[L], k2 = key[R];
                const int j1 = idx[L], j2 = idx[R];
                key[L] = k2; idx[L] = j2; key[R] = k1; idx[R] = j1;
            } else {
                bool prev = k == 0;
                int rprev = 0x7fffffff;
                if (!prev) {
                    const int Lp_ = k - 1 < nl ? (int)lp[x - 1] : 0x7fffffff, Rp_ = k - 1 < nr ? (int)rp[x - 1] : -1;
                    prev = Lp_ < Rp_;
                    rprev = Rp_;
                }
                if (prev) cut[f] = (T)min(L, k >= 1 ? rprev : 0x7fffffff);  // the first k without a swap: __unguarded_partition returns here
            }
        }
        __syncthreads();
#undef TS_GL
#undef TS_GR
        // ---- the two ranges of every partition ----
        bool mine = false;
        for (int x = tid; x < n; x += NT) {
            const int f = sf[x], l = sl[x];
            if (l - f <= stop) continue;
            const int c = cut[f];
            int nf = f, nl_ = l;
            if (x < c) nl_ = c; else nf = c;
            sf[x] = (T)nf; sl[x] = (T)nl_;
            if (dep && x == nf) dep[x] = depth;
            mine |= nl_ - nf > stop;
        }
        if (mine) *s_any = 1;
        __syncthreads();
        any = *s_any != 0;
        __syncthreads();
    }
    return true;
}

// __final_insertion_sort: a stable sort inside every range (of at most 16 elements); out[first of the range + rank] = idx
template <int NT, typename T, typename K>
__device__ __forceinline__ void sort_final(const K* __restrict__ key, const int* __restrict__ idx, const T* __restrict__ sf, const T* __restrict__ sl, int n,
                                           int* __restrict__ out) {
    for (int x = threadIdx.x; x < n; x += NT) {
        const int f = sf[x], l = sl[x];
        const K k = key[x];
        int rank = 0;
        for (int y = f; y < l; ++y) {
            const K ky = key[y];
            rank += (ky < k || (ky == k && y < x)) ? 1 : 0;
        }
        out[f + rank] = idx[x];
    }
}


}  // namespace tc2li
