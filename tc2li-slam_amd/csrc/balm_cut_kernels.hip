// gfx950 kernels of the plane extraction of the LiDAR window: cut_voxel (SF/src/bavoxel.cc:42-91), OCTO_TREE_NODE::judge_eigen / recut
// (SF/include/bavoxel.h:492-602) and VOX_HESS::push_voxel (:57-78) for a BATCH of windows, with clusters bit-identical to
// balm_build_planes (balm_host.cpp: the test oracle of these kernels, and the path of a window outside their range).
//
// What the host walk does lazily the kernels do eagerly, because nothing in it depends on the data but the plane tests:
//   * a point's root voxel (1 m), its octant in the root and its octant in that octant follow from its coordinates alone, so the three
//     orders the walk meets the points in -- by root; by (root, octant); by (root, octant, octant) -- are three stable sorts of the point
//     list (stable: inside a cell the points keep the (keyframe, scan) order, the order the host's sums run in);
//   * the statistics of EVERY cell of every layer with more than 15 points are formed (per keyframe a sequential sum over the cell's
//     contiguous run of that keyframe: the host's additions in the host's order) and the plane test of every such cell evaluated;
//   * a thread per root then walks its three layers as the host's stack does -- a planar cell ends the descent, a cell of layer 2 too --
//     and the planes leave in the host's order: roots by first appearance, octants ascending, depth first.
// Root voxels are numbered by first appearance as on the host: a hash table keeps the smallest point index of every root key and the
// roots are sorted by it.  One workgroup per window does the sorts (4-bit LSD radix; a thread ranks its contiguous block of the list in
// order); the per-cell work runs over all windows' cells at once.
#include <hip/hip_runtime.h>

#include "launch.hpp"
#pragma clang fp contract(off)
#include <limits.h>
#include <stdint.h>

#include <algorithm>

#include "balm_cut_device.hpp"

namespace tc2li {

namespace {
constexpr int kCutThreads = 1024;
constexpr int kMinPoints = 15;  // SF/include/bavoxel.h:35-41 (min_ps)

__device__ __forceinline__ unsigned int cut_hash(unsigned long long k, int bits) {
    return (unsigned int)((k * 0x9E3779B97F4A7C15ull) >> (64 - bits));
}
// [lo, hi) of the places of cell [b, e) of `order` whose point index lies in [j0, j1): the cell's points stand in ascending index order
__device__ __forceinline__ void cut_slot_run(const int* __restrict__ order, int b, int e, int j0, int j1, int& lo, int& hi) {
    int x = b, y = e;
    while (x < y) { const int m = (x + y) >> 1; if (order[m] < j0) x = m + 1; else y = m; }
    lo = x;
    y = e;
    while (x < y) { const int m = (x + y) >> 1; if (order[m] < j1) x = m + 1; else y = m; }
    hi = x;
}
}  // namespace

// ---- pass 0: empty tables ----
__global__ __launch_bounds__(256) void k_balm_cut_init(const BalmCutTask* __restrict__ tasks) {
    const BalmCutTask T = global_record(tasks[blockIdx.y]);
    const int cap = 1 << T.table_bits;
    for (int h = blockIdx.x * 256 + threadIdx.x; h < cap; h += gridDim.x * 256) { T.table_key[h] = 0ull; T.table_first[h] = INT_MAX; }
    if (blockIdx.x == 0 && threadIdx.x < 4) T.state[threadIdx.x] = 0;
}

// ---- pass 1: every point into the common frame; its root key into the window's hash table (smallest point index per key) ----
__global__ __launch_bounds__(256) void k_balm_cut_points(const BalmCutTask* __restrict__ tasks) {
    const BalmCutTask T = global_record(tasks[blockIdx.y]);
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= T.n_points) return;
    int slot = 0;
#pragma unroll
    for (int i = 1; i < kBalmCutMaxW; ++i) slot += (i < T.W && j >= T.cloud_off[i]) ? 1 : 0;
    const double local[3] = {(double)T.cloud[3 * (size_t)j], (double)T.cloud[3 * (size_t)j + 1], (double)T.cloud[3 * (size_t)j + 2]};
    double Rx[3], w[3];
    m3_vec(tasks[blockIdx.y].rel[slot].R, local, Rx);  // (indexed per lane: from the record in memory, not from the private copy)
    long long kk[3];
    bool in_range = true;
    int o1 = 0, o2 = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        w[k] = Rx[k] + tasks[blockIdx.y].rel[slot].p[k];
        float loc = (float)(w[k] / 1.0);  // voxel_size 1 (bavoxel.cc:50-55)
        if (loc < 0) loc -= 1.0f;
        const bool ok = loc > -1.0e6f && loc < 1.0e6f;  // (false for a NaN too)
        in_range = in_range && ok;
        kk[k] = ok ? (long long)loc : 0;
        // the root's centre and the two octants (the cells' centres are float: voxel_center / quater_length, bavoxel.h:560-580)
        const float c0 = (float)((0.5 + (double)kk[k]) * 1.0);
        const int b1 = w[k] > (double)c0;
        const float c1 = c0 + (float)(2 * b1 - 1) * 0.25f;
        const int b2 = w[k] > (double)c1;
        o1 = 2 * o1 + b1;
        o2 = 2 * o2 + b2;
    }
    T.world[3 * (size_t)j] = w[0]; T.world[3 * (size_t)j + 1] = w[1]; T.world[3 * (size_t)j + 2] = w[2];
    T.oct[j] = (unsigned char)(o1 << 3 | o2);
    if (!in_range) { T.state[1] = 1; T.point_slot[j] = 0; return; }  // a coordinate the 21-bit key fields do not hold: the host takes the window
    const unsigned long long key = ((((unsigned long long)(kk[0] + (1 << 20)) & 0x1fffff) << 42) | (((unsigned long long)(kk[1] + (1 << 20)) & 0x1fffff) << 21) |
                                    ((unsigned long long)(kk[2] + (1 << 20)) & 0x1fffff)) + 1ull;  // + 1: no key is 0 (= empty)
    const unsigned int mask = (1u << T.table_bits) - 1u;
    unsigned int h = cut_hash(key, T.table_bits);
    for (unsigned int probe = 0; probe < mask; ++probe) {  // (the table has twice as many slots as the window has points)
        const unsigned long long old = atomicCAS(&T.table_key[h], 0ull, key);
        if (old == 0ull || old == key) break;
        h = (h + 1) & mask;
    }
    atomicMin(&T.table_first[h], j);
    T.point_slot[j] = (int)h;
}

// ---- pass 2 (one workgroup per window): roots by first appearance; the three stable orders; the cells of the three layers ----
// LSD radix sort of `n` (key, value) pairs by the low `bits` of the key, 4 bits per pass, stable: thread t owns the contiguous block
// [t per, (t + 1) per) of the list and ranks its elements in order.  The sorted list is in (key_a, val_a) on return.
__device__ void cut_radix_sort(unsigned int*& key_a, int*& val_a, unsigned int*& key_b, int*& val_b, int n, int bits, unsigned short* s_cnt /* [16][kCutThreads] */,
                               int* s_wave) {
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_in_block(), per = (n + kCutThreads - 1) / kCutThreads;
    const int lo = min(tid * per, n), hi = min(lo + per, n);
    for (int sh = 0; sh < bits; sh += 4) {
        int cnt[16];
#pragma unroll
        for (int d = 0; d < 16; ++d) cnt[d] = 0;
        for (int k = lo; k < hi; ++k) {
            const int d = (key_a[k] >> sh) & 15;
#pragma unroll
            for (int q = 0; q < 16; ++q) cnt[q] += q == d;
        }
#pragma unroll
        for (int d = 0; d < 16; ++d) s_cnt[d * kCutThreads + tid] = (unsigned short)cnt[d];
        __syncthreads();
        // exclusive scan of the 16 x 1024 counters in (digit, thread) order: thread t takes counters [16 t, 16 t + 16)
        int part[16], sum = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) { part[q] = s_cnt[16 * tid + q]; sum += part[q]; }
        int incl = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(incl, o, 64); if (lane >= o) incl += u; }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        int run = incl - sum;
        for (int w = 0; w < wave; ++w) run += s_wave[w];
#pragma unroll
        for (int q = 0; q < 16; ++q) { s_cnt[16 * tid + q] = (unsigned short)run; run += part[q]; }  // n <= 65535
        __syncthreads();
        int at[16];
#pragma unroll
        for (int d = 0; d < 16; ++d) at[d] = s_cnt[d * kCutThreads + tid];
        for (int k = lo; k < hi; ++k) {
            const unsigned int kv = key_a[k];
            const int d = (kv >> sh) & 15;
            int pos = 0;
#pragma unroll
            for (int q = 0; q < 16; ++q) { pos = q == d ? at[q] : pos; at[q] += q == d; }
            if (pos < n) { key_b[pos] = kv; val_b[pos] = val_a[k]; }
        }
        __syncthreads();
        { unsigned int* t = key_a; key_a = key_b; key_b = t; }
        { int* t = val_a; val_a = val_b; val_b = t; }
    }
}

__global__ __launch_bounds__(kCutThreads) void k_balm_cut_sort(const BalmCutTask* __restrict__ tasks) {
    const BalmCutTask T = global_record(tasks[blockIdx.x]);
    __shared__ unsigned short s_cnt[16 * kCutThreads];
    __shared__ int s_wave[kCutThreads / 64];
    __shared__ int s_n;
    if (T.state[1]) return;  // (set by an earlier launch: the same for every thread)
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_in_block(), n = T.n_points, cap = 1 << T.table_bits;
    // ---- the roots: the occupied table slots (in any order), sorted by the smallest point index = first appearance ----
    if (tid == 0) s_n = 0;
    __syncthreads();
    unsigned int* ka = T.sort_key_a; int* va = T.sort_val_a; unsigned int* kb = T.sort_key_b; int* vb = T.sort_val_b;
    for (int h0 = 0; h0 < cap; h0 += kCutThreads) {
        const int h = h0 + tid;
        const bool occ = h < cap && T.table_key[h] != 0ull;
        const unsigned long long bal = __ballot(occ);
        int base = 0;
        if (lane == 0 && bal) base = atomicAdd(&s_n, __popcll(bal));
        base = __shfl(base, 0, 64);
        if (occ) {
            const int at = base + __popcll(bal & ((1ull << lane) - 1ull));
            if (at < n) { ka[at] = (unsigned int)T.table_first[h]; va[at] = h; }  // (a key has a point: at most n keys)
        }
    }
    __syncthreads();
    const int n_roots = min(s_n, n);
    int idx_bits = 1;
    while ((1 << idx_bits) < n) ++idx_bits;
    cut_radix_sort(ka, va, kb, vb, n_roots, idx_bits, s_cnt, s_wave);
    for (int r = tid; r < n_roots; r += kCutThreads) T.table_id[va[r]] = r;
    __syncthreads();
    int root_bits = 1;
    while ((1 << root_bits) < n_roots) ++root_bits;
    // ---- the three orders: layer 0 by root; layer 1 by (root, octant); layer 2 by (root, octant, octant) ----
    for (int layer = 0; layer < 3; ++layer) {
        unsigned int* k0 = T.sort_key_a; int* v0 = T.sort_val_a; unsigned int* k1 = T.sort_key_b; int* v1 = T.sort_val_b;
        const int shift = 3 * layer;  // bits of the octants in the key
        for (int j = tid; j < n; j += kCutThreads) {
            const unsigned int root = (unsigned int)T.table_id[T.point_slot[j]];
            const unsigned int oc = (unsigned int)T.oct[j] >> (6 - shift);  // layer 0: none, 1: the first octant, 2: both
            k0[j] = root << shift | oc;
            v0[j] = j;
        }
        __syncthreads();
        cut_radix_sort(k0, v0, k1, v1, n, root_bits + shift, s_cnt, s_wave);
        // the order and the cells (runs of equal keys) of this layer
        int* order = T.order + (size_t)layer * n;
        int* cell_begin = T.cell_begin + (size_t)layer * (n + 1);
        unsigned int* cell_key = T.cell_key + (size_t)layer * n;
        if (tid == 0) s_n = 0;
        __syncthreads();
        for (int a0 = 0; a0 < n; a0 += kCutThreads) {
            const int a = a0 + tid;
            bool head = false;
            if (a < n) { order[a] = v0[a]; head = a == 0 || k0[a] != k0[a - 1]; }
            const unsigned long long bal = __ballot(head);
            if (lane == 0) s_wave[wave] = __popcll(bal);
            __syncthreads();
            int before = 0, tot = 0;
            for (int w = 0; w < kCutThreads / 64; ++w) { before += w < wave ? s_wave[w] : 0; tot += s_wave[w]; }
            if (head) { const int c = s_n + before + __popcll(bal & ((1ull << lane) - 1ull)); cell_begin[c] = a; cell_key[c] = k0[a]; }
            __syncthreads();
            if (tid == 0) s_n += tot;
            __syncthreads();
        }
        if (tid == 0) { cell_begin[s_n] = n; T.n_cells[layer] = s_n; }
        __syncthreads();
    }
    if (tid == 0) T.state[2] = n_roots;
}

// ---- pass 3: the plane test of every cell of more than 15 points, all three layers in one launch (blockIdx.z) ----
// Eight lanes per cell, lane i the statistics of keyframe i in the common frame (a sequential sum over the keyframe's run inside the
// cell); the first lane adds them in keyframe order, forms the covariance and its eigenvalues.
__global__ __launch_bounds__(256) void k_balm_cut_judge(const BalmCutTask* __restrict__ tasks) {
    const BalmCutTask T = global_record(tasks[blockIdx.y]);
    if (T.state[1]) return;
    const int layer = blockIdx.z, n = T.n_points, W = T.W, n_cells = T.n_cells[layer];
    const int slot = threadIdx.x & 7, first = (threadIdx.x & 63) & ~7;
    const int* cell_begin = T.cell_begin + (size_t)layer * (n + 1);
    const int* order = T.order + (size_t)layer * n;
    unsigned char* flag = T.cell_flag + (size_t)layer * n;
    for (int c = (blockIdx.x * 256 + threadIdx.x) >> 3; c < n_cells; c += gridDim.x * 32) {  // (the eight lanes of a cell stay together)
        const int b = cell_begin[c], e = cell_begin[c + 1];
        if (e - b <= kMinPoints) { if (slot == 0) flag[c] = 0; continue; }
        double S[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (slot < W) {
            int lo, hi;
            cut_slot_run(order, b, e, tasks[blockIdx.y].cloud_off[slot], tasks[blockIdx.y].cloud_off[slot + 1], lo, hi);  // (indexed per lane: from the record in memory)
            for (int a = lo; a < hi; ++a) {
                const double* x = T.world + 3 * (size_t)order[a];
                const double x0 = x[0], x1 = x[1], x2 = x[2];
                S[0] += x0 * x0; S[1] += x0 * x1; S[2] += x0 * x2; S[3] += x1 * x1; S[4] += x1 * x2; S[5] += x2 * x2;
                S[6] += x0; S[7] += x1; S[8] += x2;
            }
            S[9] = (double)(hi - lo);
        }
        double A[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        int cnt = 0, seen = 0;
        for (int i = 0; i < W; ++i) {
#pragma unroll
            for (int k = 0; k < 9; ++k) A[k] += __shfl(S[k], first + i, 64);
            const int ni = (int)__shfl(S[9], first + i, 64);
            cnt += ni;
            seen += ni != 0;
        }
        if (slot != 0) continue;
        const double inv = 1.0 / cnt;
        const double cm[3] = {inv * A[6], inv * A[7], inv * A[8]};
        double Ps[9], C[9], lambda[3], U[9];
        sym_unpack(A, Ps);
        for (int a = 0; a < 3; ++a) for (int q = 0; q < 3; ++q) C[3 * a + q] = inv * Ps[3 * a + q] - cm[a] * cm[q];
        eig_sym3(C, lambda, U);
        const float ratio = layer == 0 ? 1.0f / 36 : 1.0f / 25;  // eigen_value_array
        const bool plane = lambda[0] / lambda[1] < (double)ratio;
        flag[c] = plane ? (seen >= 2 ? 2 : 3) : 1;
    }
}

// ---- pass 4 (one workgroup per window): the walk.  thread = root: its planes counted, a prefix over the roots, the planes listed ----
__device__ __forceinline__ int cut_find_cell(const unsigned int* __restrict__ keys, int n, unsigned int key) {  // index of `key`; -1: no such cell
    int lo = 0, hi = n;
    while (lo < hi) { const int m = (lo + hi) >> 1; if (keys[m] < key) lo = m + 1; else hi = m; }
    return lo < n && keys[lo] == key ? lo : -1;
}
// the planes below root r in the host's order: emit(layer, cell) for each; returns their number
template <typename Emit>
__device__ __forceinline__ int cut_walk_root(const BalmCutTask& T, int r, Emit&& emit) {
    const int n = T.n_points;
    const unsigned char *f0 = T.cell_flag, *f1 = T.cell_flag + n, *f2 = T.cell_flag + 2 * (size_t)n;
    const unsigned int *k1 = T.cell_key + n, *k2 = T.cell_key + 2 * (size_t)n;
    const int n1 = T.n_cells[1], n2 = T.n_cells[2];
    // the cells of layer 0 are the roots in the order of their numbers (every root has a point)
    const int fl0 = f0[r];
    if (fl0 == 2) { emit(0, r); return 1; }
    if (fl0 != 1) return 0;
    int count = 0;
    for (unsigned int o1 = 0; o1 < 8; ++o1) {
        const int c1 = cut_find_cell(k1, n1, (unsigned int)r << 3 | o1);
        if (c1 < 0) continue;
        const int fl1 = f1[c1];
        if (fl1 == 2) { emit(1, c1); ++count; continue; }
        if (fl1 != 1) continue;
        for (unsigned int o2 = 0; o2 < 8; ++o2) {
            const int c2 = cut_find_cell(k2, n2, ((unsigned int)r << 3 | o1) << 3 | o2);
            if (c2 >= 0 && f2[c2] == 2) { emit(2, c2); ++count; }  // layer_limit 2: a cell of layer 2 that is no plane is dropped
        }
    }
    return count;
}
__global__ __launch_bounds__(kCutThreads) void k_balm_cut_walk(const BalmCutTask* __restrict__ tasks) {
    const BalmCutTask T = global_record(tasks[blockIdx.x]);
    __shared__ int s_wave[kCutThreads / 64];
    __shared__ int s_base;
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_in_block();
    if (T.state[1]) { if (tid == 0) { T.result_host[0] = 0; T.result_host[1] = 1; } return; }
    const int n_roots = T.state[2];
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int r0 = 0; r0 < n_roots; r0 += kCutThreads) {
        const int r = r0 + tid;
        const int cnt = r < n_roots ? cut_walk_root(T, r, [](int, int) {}) : 0;
        int incl = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(incl, o, 64); if (lane >= o) incl += u; }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        int at = s_base + incl - cnt, tot = 0;
        for (int w = 0; w < kCutThreads / 64; ++w) { at += w < wave ? s_wave[w] : 0; tot += s_wave[w]; }
        if (cnt) cut_walk_root(T, r, [&](int layer, int cell) { if (at < kBalmCutMaxPlanes) T.plane_cell[at] = layer << 28 | cell; ++at; });
        __syncthreads();
        if (tid == 0) s_base += tot;
        __syncthreads();
    }
    if (tid == 0) {
        const int over = s_base > kBalmCutMaxPlanes;  // more planes than the batched LiDAR kernels take: the host path's window
        T.state[0] = over ? 0 : s_base; T.state[3] = over;  // (state[1] stays as the launches before left it: every thread of this one read it)
        T.result_host[0] = over ? 0 : s_base; T.result_host[1] = over; T.result_host[2] = n_roots; T.result_host[3] = s_base;
    }
}

// ---- pass 5: the planes' clusters: per (plane, keyframe) the points of the cell's run of that keyframe in the keyframe's OWN frame ----
__global__ __launch_bounds__(256) void k_balm_cut_clusters(const BalmCutTask* __restrict__ tasks) {
    const BalmCutTask T = global_record(tasks[blockIdx.y]);
    if (T.state[1]) return;
    const int t = blockIdx.x * 256 + threadIdx.x, W = T.W, n = T.n_points;
    const int p = t / W, slot = t - p * W;
    if (p >= T.state[0]) return;  // (0 for a window with more planes than kBalmCutMaxPlanes)
    const int pc = T.plane_cell[p], layer = pc >> 28, c = pc & 0x0fffffff;
    const int* cell_begin = T.cell_begin + (size_t)layer * (n + 1);
    const int* order = T.order + (size_t)layer * n;
    const int b = cell_begin[c], e = cell_begin[c + 1];
    int lo, hi;
    cut_slot_run(order, b, e, tasks[blockIdx.y].cloud_off[slot], tasks[blockIdx.y].cloud_off[slot + 1], lo, hi);  // (indexed per lane: from the record in memory)
    PlaneCluster pcl;
#pragma unroll
    for (int k = 0; k < 6; ++k) pcl.P[k] = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) pcl.v[k] = 0;
    for (int a = lo; a < hi; ++a) {
        const float* xf = T.cloud + 3 * (size_t)order[a];
        const double x0 = (double)xf[0], x1 = (double)xf[1], x2 = (double)xf[2];
        pcl.P[0] += x0 * x0; pcl.P[1] += x0 * x1; pcl.P[2] += x0 * x2; pcl.P[3] += x1 * x1; pcl.P[4] += x1 * x2; pcl.P[5] += x2 * x2;
        pcl.v[0] += x0; pcl.v[1] += x1; pcl.v[2] += x2;
    }
    pcl.n = (double)(hi - lo);
    T.clusters[(size_t)p * W + slot] = pcl;
    if (slot == 0) T.coe[p] = (double)(e - b);  // coe = the keyframes' point counts added up (bavoxel.h:66-71): the cell's points
}

void launch_balm_cut(const BalmCutTask* tasks, int n_tasks, int max_points, int max_table, hipStream_t st) {
    if (n_tasks <= 0 || max_points <= 0) return;
    TC2LI_LAUNCH(k_balm_cut_init, dim3(std::min((max_table + 255) / 256, 64), n_tasks), dim3(256), 0, st, tasks);
    TC2LI_LAUNCH(k_balm_cut_points, dim3((max_points + 255) / 256, n_tasks), dim3(256), 0, st, tasks);
    TC2LI_LAUNCH(k_balm_cut_sort, dim3(n_tasks), dim3(kCutThreads), 0, st, tasks);
    TC2LI_LAUNCH(k_balm_cut_judge, dim3(std::min((max_points * 8 + 255) / 256, 32), n_tasks, 3), dim3(256), 0, st, tasks);
    TC2LI_LAUNCH(k_balm_cut_walk, dim3(n_tasks), dim3(kCutThreads), 0, st, tasks);
    TC2LI_LAUNCH(k_balm_cut_clusters, dim3((kBalmCutMaxPlanes * kBalmCutMaxW + 255) / 256, n_tasks), dim3(256), 0, st, tasks);
}

}  // namespace tc2li
