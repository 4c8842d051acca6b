// gfx950 kernels of the persistent LiDAR map: map_incremental
// (SF/include/lidar_front_end/LidarFrontEnd.cpp:387-435), KD_TREE::Add_Points with down-sampling (ikd_Tree.cpp:478-584),
// Delete_Point_Boxes (:643) and the (re)build of the dense grid that stands in for the ikd-Tree.
// Every kernel works on a BATCH of maps: task = blockIdx.y names one (scan, map) pair whose pointers and sizes sit in a
// device-resident task record, so the maps of all sequences of a step are maintained by one launch per phase instead of a dozen
// launches and three host synchronisations per map (the single-map entry points are batches of one).
#include <hip/hip_runtime.h>

#include "launch.hpp"
#pragma clang fp contract(off)
#include <stdint.h>

#include "lidar_device.hpp"
#include "lidar_device_fn.hpp"

namespace tc2li {

// ---- map_incremental: insertion class of every down-sampled scan point -----------------------------------------------------------
// 0 = not added, 1 = PointToAdd (down-sampled insertion), 2 = PointNoNeedDownsample; world coordinates at the (possibly updated) state.
__global__ __launch_bounds__(256) void k_mapinc_classify(const MapIncTask* __restrict__ tasks) {
    const MapIncTask T = global_record(tasks[blockIdx.y]);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= T.n) return;
    const double fs = T.fs;
    const PointXYZINormal pw = body_to_world(T.body[i], T.st);
    T.world[i] = pw;
    uint8_t c = 1;
    const int nf = T.nfound[i];
    if (nf > 0 && T.ekf_inited) {
        const float mx = (float)(floor((double)pw.x / fs) * fs + 0.5 * fs), my = (float)(floor((double)pw.y / fs) * fs + 0.5 * fs),
                    mz = (float)(floor((double)pw.z / fs) * fs + 0.5 * fs);
        const float dist = calc_dist3(pw.x, pw.y, pw.z, mx, my, mz);
        const PointXYZINormal n0 = T.grid.points[T.nearest_idx[(size_t)i * 5]];
        if ((double)fabsf(n0.x - mx) > 0.5 * fs && (double)fabsf(n0.y - my) > 0.5 * fs && (double)fabsf(n0.z - mz) > 0.5 * fs) {
            c = 2;
        } else if (nf >= 5) {
            for (int r = 0; r < 5; ++r) {
                const PointXYZINormal q = T.grid.points[T.nearest_idx[(size_t)i * 5 + r]];
                if (calc_dist3(q.x, q.y, q.z, mx, my, mz) < dist) { c = 0; break; }
            }
        }
    }
    T.cls[i] = c;
}

// One workgroup per task: the PointToAdd list in scan order, sorted by (map voxel, scan order) so that every voxel's candidates
// are consecutive and keep their order; group starts are flagged.  The PointNoNeedDownsample list is compacted in order.
// counts: [0] n_add [1] n_groups [2] n_noneed [3] overflow
__global__ __launch_bounds__(1024) void k_mapinc_group(const MapIncTask* __restrict__ tasks) {
    const MapIncTask T = global_record(tasks[blockIdx.x]);
    extern __shared__ unsigned long long s_key[];  // kMapIncMax keys, then kMapIncMax indices
    int* s_idx = reinterpret_cast<int*>(s_key + kMapIncMax);
    __shared__ int s_wave[16], s_base[3];
    const int tid = threadIdx.x, n = T.n;
    const float ds = T.ds;
    if (tid < 3) s_base[tid] = 0;
    if (tid == 0) { T.out[14] = 0; T.out[15] = 0; }  // the deletion list of a lean task (k_mapinc_apply, the next launch, appends to it)
    __syncthreads();
    // ordered compaction of both lists, 1024 points at a time
    for (int b0 = 0; b0 < n; b0 += 1024) {
        const int i = b0 + tid;
        const uint8_t c = i < n ? T.cls[i] : 0;
        int total;
        int pos = block_flag_scan(c == 1, s_wave, total);
        if (c == 1) {
            const int o = s_base[0] + pos;
            if (o < kMapIncMax) {
                const PointXYZINormal p = T.world[i];
                const long long ix = (long long)floorf(p.x / ds), iy = (long long)floorf(p.y / ds), iz = (long long)floorf(p.z / ds);
                s_key[o] = ((unsigned long long)((ix + (1 << 20)) & 0x1fffff) << 42) | ((unsigned long long)((iy + (1 << 20)) & 0x1fffff) << 21) |
                           (unsigned long long)((iz + (1 << 20)) & 0x1fffff);
                s_idx[o] = i;
            }
        }
        __syncthreads();
        if (tid == 0) s_base[0] += total;
        __syncthreads();
        pos = block_flag_scan(c == 2, s_wave, total);
        if (c == 2) T.noneed[s_base[2] + pos] = i;
        __syncthreads();
        if (tid == 0) s_base[2] += total;
        __syncthreads();
    }
    const int m = min(s_base[0], kMapIncMax);
    int P = 1;
    while (P < m) P <<= 1;
    for (int k = m + tid; k < P; k += 1024) { s_key[k] = ~0ull; s_idx[k] = 0x7fffffff; }
    __syncthreads();
    for (int size = 2; size <= P; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < (P >> 1); t += 1024) {
                const int lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
                const bool up = (lo & size) == 0;
                const unsigned long long ka = s_key[lo], kb = s_key[hi];
                const int ia = s_idx[lo], ib = s_idx[hi];
                const bool a_gt_b = ka > kb || (ka == kb && ia > ib);
                if (a_gt_b == up) { s_key[lo] = kb; s_key[hi] = ka; s_idx[lo] = ib; s_idx[hi] = ia; }
            }
            __syncthreads();
        }
    // group starts (ordered compaction of the flags)
    if (tid == 0) s_base[1] = 0;
    __syncthreads();
    for (int b0 = 0; b0 < m; b0 += 1024) {
        const int k = b0 + tid;
        const bool start = k < m && (k == 0 || s_key[k] != s_key[k - 1]);
        if (k < m) { T.recs[k].key = s_key[k]; T.recs[k].idx = s_idx[k]; T.recs[k].pad = 0; }
        int total;
        const int pos = block_flag_scan(start, s_wave, total);
        if (start) T.group_start[s_base[1] + pos] = k;
        __syncthreads();
        if (tid == 0) s_base[1] += total;
        __syncthreads();
    }
    if (tid == 0) {
        T.group_start[s_base[1]] = m;
        T.out[0] = m; T.out[1] = s_base[1]; T.out[2] = s_base[2]; T.out[3] = s_base[0] > kMapIncMax ? 1 : 0;
        if (s_base[0] > kMapIncMax) atomicExch(T.batch_overflow, 1);
    }
}

// One thread per map voxel that receives candidates: the sequence of KD_TREE::Add_Points(downsample_on) calls for that
// voxel.  The voxel's content is either the stored points E (untouched so far) or a single point c; a candidate p replaces
// the content by the point closest to the voxel centre among content + p whenever the content has more than one point or
// p itself is that closest point.
__global__ __launch_bounds__(128) void k_mapinc_apply(const MapIncTask* __restrict__ tasks) {
    const MapIncTask T = global_record(tasks[blockIdx.y]);
    const int g = blockIdx.x * 128 + threadIdx.x;
    if (g >= T.out[1]) return;
    if (*T.batch_overflow) return;  // a failing batch marks nothing (k_mapinc_group has raised the word: no map is touched)
    const MapGrid& grid = T.grid;
    const float ds = T.ds;
    const int k0 = T.group_start[g], k1 = T.group_start[g + 1];
    const PointXYZINormal first = T.world[T.recs[k0].idx];
    float bmin[3], bmax[3], mid[3];
    const float c[3] = {first.x, first.y, first.z};
    for (int a = 0; a < 3; ++a) {
        bmin[a] = (float)(floor((double)(c[a] / ds)) * (double)ds);
        bmax[a] = bmin[a] + ds;
        mid[a] = (float)((double)bmin[a] + (double)(bmax[a] - bmin[a]) / 2.0);
    }
    // stored points inside [bmin, bmax): count and the one closest to the centre
    int s = 0;
    float e_dist = 0.f;
    PointXYZINormal e_best = first;
    auto for_each_stored = [&](auto&& fn) {
        if (grid.n_points == 0) return;
        const int xa = max((int)floorf(bmin[0] * grid.inv_cell) - grid.x0, 0), xb = min((int)floorf(bmax[0] * grid.inv_cell) - grid.x0, grid.nx - 1);
        const int ya = max((int)floorf(bmin[1] * grid.inv_cell) - grid.y0, 0), yb = min((int)floorf(bmax[1] * grid.inv_cell) - grid.y0, grid.ny - 1);
        const int za = max((int)floorf(bmin[2] * grid.inv_cell) - grid.z0, 0), zb = min((int)floorf(bmax[2] * grid.inv_cell) - grid.z0, grid.nz - 1);
        if (xa > xb) return;
        for (int qz = za; qz <= zb; ++qz)
            for (int qy = ya; qy <= yb; ++qy) {
                const int row = qz * grid.ny + qy;
                for (int qx = xa; qx <= xb; ++qx) {  // cell by cell: a voxel's box touches one or two cells along x
                    const int si = map_start_index(grid, row, qx);
                    for (int k = grid.bucket_start[si]; k < grid.bucket_start[si + 1]; ++k) {
                        const float4 m = grid.pts[k];
                        if (__float_as_int(m.w) < 0) continue;  // tombstone
                        if (bmin[0] <= m.x && bmax[0] > m.x && bmin[1] <= m.y && bmax[1] > m.y && bmin[2] <= m.z && bmax[2] > m.z) fn(__float_as_int(m.w), k);
                    }
                }
            }
    };
    for_each_stored([&](int idx, int) {
        const PointXYZINormal q = grid.points[idx];
        const float d = calc_dist3(q.x, q.y, q.z, mid[0], mid[1], mid[2]);
        if (s == 0 || d < e_dist) { e_dist = d; e_best = q; }
        ++s;
    });
    bool intact = true;
    PointXYZINormal cur = first;  // the single point of a replaced content
    float cur_dist = 0.f;
    for (int k = k0; k < k1; ++k) {
        const PointXYZINormal p = T.world[T.recs[k].idx];
        const float dp = calc_dist3(p.x, p.y, p.z, mid[0], mid[1], mid[2]);
        const int size = intact ? s : 1;
        const float stored_dist = intact ? e_dist : cur_dist;
        const bool stored_wins = size > 0 && stored_dist < dp;
        const PointXYZINormal best = stored_wins ? (intact ? e_best : cur) : p;
        const bool same = fabsf(p.x - best.x) < 1e-6f && fabsf(p.y - best.y) < 1e-6f && fabsf(p.z - best.z) < 1e-6f;
        if (size > 1 || same) {  // Delete_by_range(box) + Add_by_point(best)
            cur_dist = stored_wins ? stored_dist : dp;
            cur = best;
            intact = false;
        }
    }
    if (!intact) {
        // Delete_by_range: the stored points of the voxel leave the map; with the grid maintained in place their entries become tombstones
        // here (voxels are disjoint boxes: no other thread looks at these entries)
        float4* const entries = const_cast<float4*>(grid.pts);
        const int fix = T.fix_grid;
        if (T.lean) {  // listed, not flagged: voxels are disjoint boxes, so every point is listed once; the list is sorted before it is used
            // (the flag is set as well: a map with more than kMapDelMax deletions in one step goes through the flag passes after all)
            for_each_stored([&](int idx, int k) {
                const int at = atomicAdd(&T.out[14], 1);
                if (at < kMapDelMax) T.holes[at] = idx; else T.out[15] = 1;
                T.deleted[idx] = 1;
                if (fix) entries[k].w = __int_as_float(-1);
            });
        } else {
            for_each_stored([&](int idx, int k) { T.deleted[idx] = 1; if (fix) entries[k].w = __int_as_float(-1); });
        }
        T.appended[g] = cur;
        T.has_append[g] = 1;
    } else {
        T.has_append[g] = 0;
    }
}

// deleted[i] = 1 for every map point inside one of the boxes [min, max) (KD_TREE::Delete_Point_Boxes)
__global__ __launch_bounds__(256) void k_map_mark_boxes(const MapIncTask* __restrict__ tasks) {
    const MapIncTask T = global_record(tasks[blockIdx.y]);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= T.n_map) return;
    const PointXYZINormal p = T.grid.points[i];
    uint8_t d = T.deleted[i];
    for (int b = 0; b < T.n_boxes; ++b) {
        const float* q = T.boxes + 6 * b;
        if (q[0] <= p.x && q[3] > p.x && q[1] <= p.y && q[4] > p.y && q[2] <= p.z && q[5] > p.z) d = 1;
    }
    T.deleted[i] = d;
}

// ---- compaction of the map after deletions, in place: with K points kept, the deleted places below K are filled with the kept points
// from K on (h-th hole <- h-th such point, both in index order: deterministic); only the moved points are copied, not the map.  The map
// is a set (an ikd-Tree has no order), so the order of its flat array is ours to choose.  Per-block kept counts, their scan (one
// workgroup per map), the hole / filler lists + the old -> new index map, the moves; then the appended voxel representatives (group
// order) and the PointNoNeedDownsample points (scan order) follow from K on -------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_map_keep_count(const MapIncTask* __restrict__ tasks) {
    const MapIncTask T = global_record(tasks[blockIdx.y]);
    if (T.lean || (int)blockIdx.x >= T.keep_blocks) return;
    const int i = blockIdx.x * 1024 + threadIdx.x;
    const int c = __syncthreads_count(i < T.n_map && !T.deleted[i]);
    if (threadIdx.x == 0) T.keep_counts[blockIdx.x] = c;
}
// out: [4] kept [5] appended [6..11] bounding box of what is added (encoded floats), initialised here
__global__ __launch_bounds__(1024) void k_map_keep_scan(const MapIncTask* __restrict__ tasks) {
    const MapIncTask T = global_record(tasks[blockIdx.x]);
    if (T.lean) return;
    __shared__ int s_part[1024];
    __shared__ int s_wave[16];
    const int tid = threadIdx.x, nblocks = T.keep_blocks;
    // chunks of 1024 block counts, coalesced; the running offset carried from chunk to chunk
    int carry = 0;
    for (int b0 = 0; b0 < nblocks; b0 += 1024) {
        const int k = b0 + tid;
        const int v = k < nblocks ? T.keep_counts[k] : 0;
        s_part[tid] = v;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            const int u = tid >= o ? s_part[tid - o] : 0;
            __syncthreads();
            s_part[tid] += u;
            __syncthreads();
        }
        if (k < nblocks) T.keep_counts[k] = carry + s_part[tid] - v;
        carry += s_part[1023];
        __syncthreads();
    }
    // appended representatives: has_append over the groups
    int a = 0;
    const int ng = T.has_inc ? T.out[1] : 0;
    for (int g = tid; g < ng; g += 1024) a += T.has_append[g];
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
    if ((tid & 63) == 0) s_wave[tid >> 6] = a;
    __syncthreads();
    if (tid == 0) {
        int tot = 0;
        for (int k = 0; k < 16; ++k) tot += s_wave[k];
        T.out[4] = carry;
        T.out[5] = tot;
        T.out[13] = 0;  // set by k_map_fill when it finds the grid inconsistent
        for (int k = 0; k < 3; ++k) { T.out[6 + k] = 0x7fffffff; T.out[9 + k] = (int)0x80000000; }
    }
    // kept points among the first K = carry places (the others are the holes to fill)
    __syncthreads();
    const int K = carry, bK = K / 1024;
    const int i = bK * 1024 + tid;
    const int c = __syncthreads_count(i < K && !T.deleted[i]);
    if (tid == 0) T.out[12] = (bK < nblocks ? T.keep_counts[bK] : K) + c;
}
__global__ __launch_bounds__(1024) void k_map_holes(const MapIncTask* __restrict__ tasks) {
    const MapIncTask T = global_record(tasks[blockIdx.y]);
    if (T.lean || (int)blockIdx.x >= T.keep_blocks) return;
    if (*T.batch_overflow) return;  // nothing is touched when the batch fails (the flags are reset by the host's next call)
    __shared__ int s_wave[16];
    const int i = blockIdx.x * 1024 + threadIdx.x;
    const bool in = i < T.n_map;
    const bool was_deleted = in && T.deleted[i];
    const bool keep = in && !was_deleted;
    if (was_deleted) T.deleted[i] = 0;  // the flags are all zero again when the call ends (they belong to the old numbering); few are set
    int total;
    const int kp = T.keep_counts[blockIdx.x] + block_flag_scan(keep, s_wave, total);  // kept points before i
    const int K = T.out[4], kpK = T.out[12];
    if (!in) return;
    T.remap[i] = keep ? i : -1;
    if (!keep && i < K) T.holes[i - kp] = i;
    if (keep && i >= K) T.holes[T.n_map + (kp - kpK)] = i;
}
// Round 5: the compaction of a lean task from its deletion LIST -- one workgroup per map.  The three kernels above pass over every point of
// every map (190 k points x 512 maps: 97 M flags read twice, 390 MB of index map written, 3 M wavefronts launched per step) to find the
// ~1 600 points per map a step deletes; k_mapinc_apply knows them.  The list is sorted (bitonic, LDS); with d deletions and K = n - d kept,
// the holes are the listed places below K and the fillers the places from K on that are NOT listed -- both in index order, h-th hole <-
// h-th filler: the assignment of k_map_holes / k_map_fill, so the map's point array is the same bit for bit.  The index map (remap) is
// not written: a lean task maintains its grid in place (fix_grid) and nothing reads it.  Also k_map_keep_scan's outputs: out[4] kept,
// out[5] appended representatives, out[6..11] bounding box words, out[12] kept among the first K, out[13] = 0.
__global__ __launch_bounds__(1024) void k_map_compact_list(const MapIncTask* __restrict__ tasks) {
    const MapIncTask T = global_record(tasks[blockIdx.x]);
    if (!T.lean) return;
    __shared__ int s_del[kMapDelMax], s_mov[kMapDelMax];
    __shared__ int s_wave[16], s_tot;
    const int tid = threadIdx.x;
    if (*T.batch_overflow != 0) return;  // a failing batch touches no map (the host reports it and takes the deletion marks back)
    if (T.out[15] != 0) return;          // more deletions than the list holds: the host sends this map through the flag passes (lean = 0)
    const int d = min(T.out[14], kMapDelMax), n = T.n_map, K = n - d;
    // appended representatives: has_append over the groups (k_map_keep_scan's sum)
    int a = 0;
    const int ng = T.has_inc ? T.out[1] : 0;
    for (int g = tid; g < ng; g += 1024) a += T.has_append[g];
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
    if ((tid & 63) == 0) s_wave[tid >> 6] = a;
    // the list, padded to a power of two with the largest int, sorted ascending
    int m = 1;
    while (m < d) m <<= 1;
    for (int k = tid; k < m; k += 1024) s_del[k] = k < d ? T.holes[k] : 0x7fffffff;
    for (int k = tid; k < d; k += 1024) T.deleted[T.holes[k]] = 0;  // the flags are all zero again when the call ends
    __syncthreads();
    if (tid == 0) {
        int tot = 0;
        for (int k = 0; k < 16; ++k) tot += s_wave[k];
        T.out[4] = K;
        T.out[5] = tot;
        T.out[13] = 0;
        for (int k = 0; k < 3; ++k) { T.out[6 + k] = 0x7fffffff; T.out[9 + k] = (int)0x80000000; }
    }
    for (int size = 2; size <= m; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int k = tid; k < m / 2; k += 1024) {
                const int lo = 2 * k - (k & (stride - 1)), hi = lo + stride;  // the pair (lo, lo + stride) of the bitonic network
                const bool up = (lo & size) == 0;
                const int x = s_del[lo], y = s_del[hi];
                if ((x > y) == up) { s_del[lo] = y; s_del[hi] = x; }
            }
            __syncthreads();
        }
    // holes: the listed places below K = the first h entries of the sorted list
    int h = 0;
    {   // h = number of entries < K (binary search, every thread alike)
        int lo = 0, hi = d;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_del[mid] < K) lo = mid + 1; else hi = mid; }
        h = lo;
    }
    if (tid == 0) { T.out[12] = K - h; s_tot = 0; }
    __syncthreads();
    // fillers: the places K .. n - 1 that are not listed, in index order (there are d candidates, d - h of them listed: h fillers)
    for (int j0 = 0; j0 < d; j0 += 1024) {
        const int j = j0 + tid, i = K + j;
        bool keep = false;
        if (j < d) {
            int lo = h, hi = d;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_del[mid] < i) lo = mid + 1; else hi = mid; }
            keep = !(lo < d && s_del[lo] == i);
        }
        const unsigned long long bal = __ballot(keep);
        const int lane = tid & 63, wave = wave_in_block();
        if (lane == 0) s_wave[wave] = __popcll(bal);
        __syncthreads();
        int off = 0, tot = 0;
        for (int k = 0; k < 16; ++k) { const int c = s_wave[k]; off += k < wave ? c : 0; tot += c; }
        const int base = s_tot;
        if (keep) s_mov[base + off + __popcll(bal & ((1ull << lane) - 1ull))] = i;
        __syncthreads();
        if (tid == 0) s_tot = base + tot;
        __syncthreads();
    }
    // the moves (k_map_fill's body)
    PointXYZINormal* pts = T.dst;
    const MapGrid& g = T.grid;
    float4* const entries = const_cast<float4*>(g.pts);
    for (int t = tid; t < h; t += 1024) {
        const int dst = s_del[t], src = s_mov[t];
        const PointXYZINormal p = pts[src];
        pts[dst] = p;
        if (T.fix_grid) {  // the moved point's grid entry carries its index: renumber it where it stands (its cell is a few entries)
            const int c = map_cell(g, p.x, p.y, p.z), row = c / g.nx, ix = c - row * g.nx;
            if (c < 0 || row >= g.ny * g.nz) { atomicOr(&T.out[13], 1); continue; }
            const int si = map_start_index(g, row, ix);
            const int k0 = g.bucket_start[si], k1 = g.bucket_start[si + 1];
            if (k0 < 0 || k1 < k0 || k1 > g.n_slots) { atomicOr(&T.out[13], 2); continue; }
            for (int k = k0; k < k1; ++k)
                if (__float_as_int(entries[k].w) == src) { entries[k].w = __int_as_float(dst); break; }
        }
    }
}
__global__ __launch_bounds__(256) void k_map_fill(const MapIncTask* __restrict__ tasks) {
    const MapIncTask T = global_record(tasks[blockIdx.y]);
    if (T.lean || *T.batch_overflow) return;
    const int H = T.out[4] - T.out[12];
    PointXYZINormal* pts = T.dst;
    const MapGrid& g = T.grid;
    float4* const entries = const_cast<float4*>(g.pts);
    for (int h = blockIdx.x * 256 + threadIdx.x; h < H; h += gridDim.x * 256) {
        const int dst = T.holes[h], src = T.holes[T.n_map + h];
        const PointXYZINormal p = pts[src];
        pts[dst] = p;
        T.remap[src] = dst;
        if (T.fix_grid) {  // the moved point's grid entry carries its index: renumber it where it stands (its cell is a few entries)
            const int c = map_cell(g, p.x, p.y, p.z), row = c / g.nx, ix = c - row * g.nx;
            if (c < 0 || row >= g.ny * g.nz) { atomicOr(&T.out[13], 1); continue; }
            const int si = map_start_index(g, row, ix);
            const int k0 = g.bucket_start[si], k1 = g.bucket_start[si + 1];
            if (k0 < 0 || k1 < k0 || k1 > g.n_slots) { atomicOr(&T.out[13], 2); continue; }
            for (int k = k0; k < k1; ++k)
                if (__float_as_int(entries[k].w) == src) { entries[k].w = __int_as_float(dst); break; }
        }
    }
}
// dst[kept ...] <- appended representatives, then the no-need points; also the bounding box of what was added
// the bounding box of the points a wavefront holds (lanes without one pass have = false): shuffle minima / maxima, then six atomics per
// wavefront instead of six per point on the map's six words
__device__ __forceinline__ void wave_bbox(int* bbox_enc, const PointXYZINormal& p, bool have) {
    int mn[3], mx[3];
    mn[0] = have ? enc_float(p.x) : 0x7fffffff; mn[1] = have ? enc_float(p.y) : 0x7fffffff; mn[2] = have ? enc_float(p.z) : 0x7fffffff;
    mx[0] = have ? enc_float(p.x) : (int)0x80000000; mx[1] = have ? enc_float(p.y) : (int)0x80000000; mx[2] = have ? enc_float(p.z) : (int)0x80000000;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            mn[a] = min(mn[a], __shfl_xor(mn[a], o, 64));
            mx[a] = max(mx[a], __shfl_xor(mx[a], o, 64));
        }
    if ((threadIdx.x & 63) == 0 && mn[0] != 0x7fffffff) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { atomicMin(&bbox_enc[a], mn[a]); atomicMax(&bbox_enc[3 + a], mx[a]); }
    }
}

__global__ __launch_bounds__(256) void k_map_append(const MapIncTask* __restrict__ tasks) {
    const MapIncTask T = global_record(tasks[blockIdx.x]);
    if (!T.has_inc || *T.batch_overflow) return;
    if (T.lean && T.out[15]) return;  // its compaction has not run: the host repeats the task through the flag passes
    __shared__ int s_wave[4];
    __shared__ int s_base;
    const int tid = threadIdx.x, ng = T.out[1], nn = T.out[2], kept = T.out[4];
    int* bbox_enc = T.out + 6;
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int b0 = 0; b0 < ng; b0 += 256) {
        const int g = b0 + tid;
        const bool f = g < ng && T.has_append[g];
        const unsigned long long bal = __ballot(f);
        const int lane = tid & 63, wave = wave_in_block();
        if (lane == 0) s_wave[wave] = __popcll(bal);
        __syncthreads();
        int off = 0, tot = 0;
        for (int k = 0; k < 4; ++k) { off += k < wave ? s_wave[k] : 0; tot += s_wave[k]; }
        PointXYZINormal p{};
        if (f) {
            p = T.appended[g];
            T.dst[kept + s_base + off + __popcll(bal & ((1ull << lane) - 1ull))] = p;
        }
        wave_bbox(bbox_enc, p, f);
        __syncthreads();
        if (tid == 0) s_base += tot;
        __syncthreads();
    }
    const int base = kept + T.out[5];
    for (int k0 = 0; k0 < nn; k0 += 256) {  // whole wavefronts stay in the loop: the shuffles of wave_bbox need all lanes
        const int k = k0 + tid;
        PointXYZINormal p{};
        if (k < nn) {
            p = T.world[T.noneed[k]];
            T.dst[base + k] = p;
        }
        wave_bbox(bbox_enc, p, k < nn);
    }
}

// ---- dense grid (re)build: counting sort of the map points into 1 m cells -----------------------------------------------------------
constexpr int kScanTile = 4096;  // cells per tile of the two-level prefix sum (1024 threads x 4)
// The cell counters are zero outside a build: allocated zeroed (lidar_host.cpp grid_prepare), raised by k_map_count, read by the scans,
// and taken down again by k_map_scatter, which hands out a cell's places from the back (atomicSub).  Rounds 1-2 cleared them and a second
// counter array per build (k_map_zero: 0.5 ms per step for 512 maps of ~0.9 M cells).
// Work item u of a build: u < n_old: entry u of the old grid (skipped when its point was deleted); then the added points.
// Returns the point's cell in the NEW geometry, or a negative key no neighbouring lane shares; xyzi = position and NEW index.
__device__ __forceinline__ int map_build_item(const MapGridTask& T, int u, float4& xyzi) {
    const int lane_key = -1 - (int)(threadIdx.x & 63);
    if (u < T.n_old) {
        const float4 e = T.old_sorted[u];
        const int oi = __float_as_int(e.w);
        if (oi < 0) return lane_key;  // tombstone / unused room of a row
        const int ni = T.remap ? T.remap[oi] : oi;
        if (ni < 0) return lane_key;
        xyzi = make_float4(e.x, e.y, e.z, __int_as_float(ni));
        return map_cell(T.g, e.x, e.y, e.z);
    }
    const int i = T.n_kept + (u - T.n_old);
    if (i >= T.g.n_points) return lane_key;
    const PointXYZINormal p = T.g.points[i];
    xyzi = make_float4(p.x, p.y, p.z, __int_as_float(i));
    return map_cell(T.g, p.x, p.y, p.z);
}
__global__ __launch_bounds__(256) void k_map_count(const MapGridTask* __restrict__ tasks) {
    const MapGridTask T = global_record(tasks[blockIdx.y]);
    const int u = blockIdx.x * 256 + threadIdx.x;
    if ((int)(blockIdx.x * 256) >= T.n_old + (T.g.n_points - T.n_kept)) return;  // whole workgroup
    float4 q;
    const int c = map_build_item(T, u, q);
    const RunInfo run = wave_runs(c);
    if (run.head && c >= 0) atomicAdd(&T.counts[c], run.length);
}
// level 1: sums of tiles of 4096 consecutive cells (coalesced int4 loads)
__global__ __launch_bounds__(1024) void k_map_scan_tiles(const MapGridTask* __restrict__ tasks) {
    const MapGridTask T = global_record(tasks[blockIdx.y]);
    const int tile = blockIdx.x, n_tiles = (T.n_cells + kScanTile - 1) / kScanTile;
    if (tile >= n_tiles) return;
    __shared__ int s_wave[16];
    const int tid = threadIdx.x, base = tile * kScanTile + 4 * tid;
    int v = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) v += base + k < T.n_cells ? T.counts[base + k] : 0;
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((tid & 63) == 0) s_wave[tid >> 6] = v;
    __syncthreads();
    if (tid == 0) {
        int tot = 0;
        for (int k = 0; k < 16; ++k) tot += s_wave[k];
        T.tile_sums[tile] = tot;
    }
}
// level 2: exclusive scan of the tile sums (<= 1024 tiles = 4 M cells), one workgroup per map
__global__ __launch_bounds__(1024) void k_map_scan_tops(const MapGridTask* __restrict__ tasks) {
    const MapGridTask T = global_record(tasks[blockIdx.x]);
    __shared__ int s_part[1024];
    const int tid = threadIdx.x, n_tiles = (T.n_cells + kScanTile - 1) / kScanTile;
    const int v = tid < n_tiles ? T.tile_sums[tid] : 0;
    s_part[tid] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int u = tid >= o ? s_part[tid - o] : 0;
        __syncthreads();
        s_part[tid] += u;
        __syncthreads();
    }
    if (tid < n_tiles) T.tile_sums[tid] = s_part[tid] - v;
    if (tid == 1023) T.start[T.n_cells] = s_part[1023];
}
// level 3: exclusive scan inside every tile + the tile's offset
__global__ __launch_bounds__(1024) void k_map_scan_cells(const MapGridTask* __restrict__ tasks) {
    const MapGridTask T = global_record(tasks[blockIdx.y]);
    const int tile = blockIdx.x, n_tiles = (T.n_cells + kScanTile - 1) / kScanTile;
    if (tile >= n_tiles) return;
    __shared__ int s_wave[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_in_block(), base = tile * kScanTile + 4 * tid;
    int c[4], sum = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { c[k] = base + k < T.n_cells ? T.counts[base + k] : 0; sum += c[k]; }
    int incl = sum;  // inclusive scan over the wavefront
    for (int o = 1; o < 64; o <<= 1) {
        const int u = __shfl_up(incl, o);
        if (lane >= o) incl += u;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int off = T.tile_sums[tile];
    for (int k = 0; k < wave; ++k) off += s_wave[k];
    int run = off + incl - sum;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (base + k < T.n_cells) T.start[base + k] = run;
        run += c[k];
    }
}
__global__ __launch_bounds__(256) void k_map_scatter(const MapGridTask* __restrict__ tasks) {
    const MapGridTask T = global_record(tasks[blockIdx.y]);
    const int u = blockIdx.x * 256 + threadIdx.x;
    if ((int)(blockIdx.x * 256) >= T.n_old + (T.g.n_points - T.n_kept)) return;  // whole workgroup
    float4 q;
    const int c = map_build_item(T, u, q);
    const RunInfo run = wave_runs(c);
    int first = 0;
    // the run's lanes take consecutive places; the cell's places go from the back, which leaves its counter at zero for the next build
    if (run.head && c >= 0) {
        const int row = c / T.g.nx;
        first = T.row_start[map_start_index(T.g, row, c - row * T.g.nx)] + atomicSub(&T.counts[c], run.length) - run.length;
    }
    first = __shfl(first, run.head_lane, 64);
    if (c >= 0) T.sorted[first + ((int)(threadIdx.x & 63) - run.head_lane)] = q;
}

// The segments' starts from the plain exclusive prefix E over the cells: the segment with first cell c0 and number q begins at
// 2 E[c0] + seg_slack q (room for as many entries again + seg_slack behind every segment), its cell j at E[c0 + j] + E[c0] + seg_slack q
// (cells beyond the row's last: the end of the segment's entries); entry 16 of a segment = the end of its entries, the last entry of
// the array (segment = number of segments, j = 0) = n_slots.
__global__ __launch_bounds__(256) void k_map_row_starts(const MapGridTask* __restrict__ tasks) {
    const MapGridTask T = global_record(tasks[blockIdx.y]);
    const int nseg = T.g.ny * T.g.nz * T.g.nsx;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx > nseg * kMapSegStride) return;
    const int seg = idx / kMapSegStride, j = idx - seg * kMapSegStride;
    const int row = seg / T.g.nsx, sx = seg - row * T.g.nsx;
    const int c0 = row * T.g.nx + sx * kMapSegCells;                      // (row = rows, sx = 0: n_cells)
    const int c = row * T.g.nx + min(sx * kMapSegCells + j, T.g.nx);      // past the row's last cell: the row's end
    T.row_start[idx] = T.start[c] + T.start[c0] + T.seg_slack * seg;
}

// ---- in-place insertion (round 4): the added points of a step are merged into the rows they fall into, nothing else is touched -----
// (KD_TREE::Add_Points inserts the points into the existing tree, ikd_Tree.cpp:478-584; rounds 1-3 rebuilt the whole grid instead:
// 13 GB of traffic per step of 512 maps to add 40 MB of points.)
// One workgroup per map: (cell, point) keys of the added points, sorted; the rows that receive points.
__global__ __launch_bounds__(1024) void k_map_ins_sort(const MapInsTask* __restrict__ tasks) {
    const MapInsTask T = global_record(tasks[blockIdx.x]);
    extern __shared__ unsigned long long s_keys[];  // kMapInsMax
    __shared__ int s_wave[16], s_base, s_bad;
    const int tid = threadIdx.x, n = T.count;
    const MapGrid& g = T.g;
    if (tid == 0) { s_base = 0; s_bad = n > kMapInsMax || g.n_slots <= 0; }
    __syncthreads();
    if (!s_bad) {
        for (int i = tid; i < n; i += 1024) {
            const PointXYZINormal p = g.points[T.first + i];
            const int cx = (int)floorf(p.x * g.inv_cell) - g.x0, cy = (int)floorf(p.y * g.inv_cell) - g.y0, cz = (int)floorf(p.z * g.inv_cell) - g.z0;
            if (cx < 0 || cx >= g.nx || cy < 0 || cy >= g.ny || cz < 0 || cz >= g.nz) { s_bad = 1; s_keys[i] = ~0ull; }  // outside the grid's box
            else s_keys[i] = (unsigned long long)(unsigned)((((cz * g.ny + cy) * g.nsx + (cx >> 4)) << 4) | (cx & 15)) << 32 | (unsigned)(T.first + i);
        }
    }
    __syncthreads();
    if (s_bad) { if (tid == 0) { T.out[0] = 0; T.out[1] = 1; T.out[2] = 0; T.out[3] = 0; } return; }
    int P = 1;
    while (P < n) P <<= 1;
    for (int k = n + tid; k < P; k += 1024) s_keys[k] = ~0ull;
    __syncthreads();
    for (int size = 2; size <= P; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < (P >> 1); t += 1024) {
                const int lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
                const bool up = (lo & size) == 0;
                const unsigned long long ka = s_keys[lo], kb = s_keys[hi];
                if ((ka > kb) == up) { s_keys[lo] = kb; s_keys[hi] = ka; }
            }
            __syncthreads();
        }
    for (int b0 = 0; b0 < n; b0 += 1024) {
        const int k = b0 + tid;
        bool start = false;
        if (k < n) {
            T.keys[k] = s_keys[k];
            start = k == 0 || (int)(s_keys[k] >> 36) != (int)(s_keys[k - 1] >> 36);  // another segment
        }
        int total;
        const int pos = block_flag_scan(start, s_wave, total);
        if (start) T.row_list[s_base + pos] = k;
        __syncthreads();
        if (tid == 0) s_base += total;
        __syncthreads();
    }
    if (tid == 0) { T.row_list[s_base] = n; T.out[0] = s_base; T.out[1] = 0; T.out[2] = 0; T.out[3] = 0; }
}
// One WAVEFRONT per segment that receives points (a map's segments dealt over blockIdx.x): the segment's entries without their tombstones
// and the new points, merged by cell (the old entries of a cell first, then the new ones by index), written back from the segment's
// first place; its 17 starts follow.  A segment without room, or with more entries than the merge holds, raises out[1]: the host rebuilds
// that map's grid (segments written before stay valid or not -- the rebuild starts from the points).  A segment is some tens of entries:
// a 256-thread workgroup per segment spent its time in barriers (1.6 ms per 512 maps; this form: see DESIGN.md).
__global__ __launch_bounds__(64) void k_map_ins_rows(const MapInsTask* __restrict__ tasks) {
    const MapInsTask T = global_record(tasks[blockIdx.y]);
    __shared__ float4 s_ent[kMapRowMax];
    __shared__ int s_rank[kMapRowMax + 1];
    __shared__ int s_cs[kMapSegStride + 1];
    // (no early exit on out[1]: other workgroups raise it while this one runs; a map whose keys could not be made has no segments, out[0] = 0)
    const MapGrid& g = T.g;
    constexpr int nx = kMapSegCells;
    const int lane = threadIdx.x, n_rows = T.out[0];
    const unsigned long long below = (1ull << lane) - 1ull;
    for (int rr = blockIdx.x; rr < n_rows; rr += gridDim.x) {
        __syncthreads();  // one wavefront: orders this iteration's LDS writes behind the last one's reads
        const int k0 = T.row_list[rr], k1 = T.row_list[rr + 1], N = k1 - k0;
        const int seg = (int)(T.keys[k0] >> 36);
        int* const cs = T.row_start + (size_t)seg * kMapSegStride;
        if (lane <= kMapSegStride) s_cs[lane] = cs[lane];  // the segment's 17 starts + the next segment's first = where its room ends
        __syncthreads();
        const int b = s_cs[0], e = s_cs[nx], limit = s_cs[kMapSegStride], M = e - b;
        bool sane = b >= 0 && e >= b && limit >= e && limit <= g.n_slots && seg >= 0 && seg < g.ny * g.nz * g.nsx;
        for (int j = 0; j < nx && sane; ++j) sane = s_cs[j] <= s_cs[j + 1];
        if (!sane) {  // uniform; a grid that is not what the host says it is
            if (lane == 0) { atomicExch(&T.out[1], 1); atomicOr(&T.out[3], 1); }
            continue;
        }
        if (M + N > kMapRowMax) { if (lane == 0) atomicExch(&T.out[1], 1); continue; }  // uniform
        // the old entries into LDS; s_rank[j] = live entries before j
        int V = 0;
        for (int j0 = 0; j0 < M; j0 += 64) {
            const int j = j0 + lane;
            float4 en = make_float4(0.f, 0.f, 0.f, __int_as_float(-1));
            if (j < M) { en = T.pts[b + j]; s_ent[j] = en; }
            const unsigned long long live = __ballot(__float_as_int(en.w) >= 0);
            if (j < M) s_rank[j] = V + __popcll(live & below);
            V += __popcll(live);
        }
        if (lane == 0) s_rank[M] = V;
        __syncthreads();
        if (b + V + N > limit) { if (lane == 0) atomicExch(&T.out[1], 1); continue; }  // uniform
        if (lane == 0 && M > V) atomicAdd(&T.out[2], M - V);  // the tombstones this rewrite drops
        // new entries with a cell below `cell` (the keys of the segment are sorted)
        auto new_below = [&](int cell) {
            int lo = k0, hi = k1;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if ((int)(T.keys[mid] >> 32) < cell) lo = mid + 1; else hi = mid; }
            return lo - k0;
        };
        // the segment's new starts (from the old ones, which stay in s_cs)
        if (lane <= nx) cs[lane] = lane == nx ? b + V + N : b + s_rank[s_cs[lane] - b] + new_below(seg * nx + lane);
        // the old entries: live rank + the new points of the cells before theirs
        for (int j = lane; j < M; j += 64) {
            const float4 en = s_ent[j];
            if (__float_as_int(en.w) < 0) continue;
            const int cell = seg * nx + (((int)floorf(en.x * g.inv_cell) - g.x0) & (kMapSegCells - 1));
            const int d = b + s_rank[j] + new_below(cell);
            if (d < b || d >= limit) atomicOr(&T.out[3], 8); else T.pts[d] = en;
        }
        // the new points: behind the live old entries of the cells up to their own
        for (int i = lane; i < N; i += 64) {
            const unsigned long long key = T.keys[k0 + i];
            const int ix = (int)(key >> 32) - seg * nx, idx = (int)(unsigned)(key & 0xffffffffull);
            const int d = b + i + s_rank[s_cs[ix + 1] - b];
            if (idx < 0 || idx >= g.n_points || d < b || d >= limit) { atomicOr(&T.out[3], 16); continue; }
            const PointXYZINormal p = g.points[idx];
            T.pts[d] = make_float4(p.x, p.y, p.z, __int_as_float(idx));
        }
        for (int j = b + V + N + lane; j < e; j += 64) T.pts[j] = make_float4(0.f, 0.f, 0.f, __int_as_float(-1));  // a segment that shrank: tombstones behind it
    }
}

// ---- launchers ---------------------------------------------------------------------------------------------------------------------
void launch_map_insert(const MapInsTask* tasks, int n_tasks, hipStream_t st) {
    if (!n_tasks) return;
    (void)ensure_dynamic_lds(reinterpret_cast<const void*>(k_map_ins_sort), kMapInsMax * (int)sizeof(unsigned long long));
    TC2LI_LAUNCH(k_map_ins_sort, dim3(n_tasks), dim3(1024), kMapInsMax * sizeof(unsigned long long), st, tasks);
    TC2LI_LAUNCH(k_map_ins_rows, dim3(256, n_tasks), dim3(64), 0, st, tasks);
}
void launch_mapinc_lists(const MapIncTask* tasks, int n_tasks, int max_points, hipStream_t st) {
    if (!n_tasks || !max_points) return;
    TC2LI_LAUNCH(k_mapinc_classify, dim3((max_points + 255) / 256, n_tasks), dim3(256), 0, st, tasks);
    const size_t lds = (size_t)kMapIncMax * (sizeof(unsigned long long) + sizeof(int));
    (void)ensure_dynamic_lds(reinterpret_cast<const void*>(k_mapinc_group), (int)lds);  // 96 KB of dynamic LDS: above the default limit of a kernel
    TC2LI_LAUNCH(k_mapinc_group, dim3(n_tasks), dim3(1024), lds, st, tasks);
    // a task has at most as many voxel groups as its scan has points (and at most kMapIncMax): no workgroups beyond that
    TC2LI_LAUNCH(k_mapinc_apply, dim3((std::min(max_points, kMapIncMax) + 127) / 128, n_tasks), dim3(128), 0, st, tasks);
}
void launch_map_mark_boxes(const MapIncTask* tasks, int n_tasks, int max_map_points, hipStream_t st) {
    if (n_tasks && max_map_points) TC2LI_LAUNCH(k_map_mark_boxes, dim3((max_map_points + 255) / 256, n_tasks), dim3(256), 0, st, tasks);
}
// max_map_points: the largest map among the tasks that are NOT lean; any_lean: some task works from its deletion list; any_flagged: some
// task does not.  k_map_keep_scan initialises out[4..13] of a task that is not lean -- also of one whose map is still EMPTY (no block to
// count: max_map_points == 0), so it is launched for any_flagged, not for nb != 0 (such a task read the kept count and the
// bounding box an earlier call had left in its slot).
void launch_map_compact(const MapIncTask* tasks, int n_tasks, int max_map_points, bool any_lean, bool any_flagged, hipStream_t st) {
    if (!n_tasks) return;
    if (any_lean) TC2LI_LAUNCH(k_map_compact_list, dim3(n_tasks), dim3(1024), 0, st, tasks);
    const int nb = (max_map_points + 1023) / 1024;
    if (nb) TC2LI_LAUNCH(k_map_keep_count, dim3(nb, n_tasks), dim3(1024), 0, st, tasks);
    if (any_flagged) TC2LI_LAUNCH(k_map_keep_scan, dim3(n_tasks), dim3(1024), 0, st, tasks);
    if (nb) TC2LI_LAUNCH(k_map_holes, dim3(nb, n_tasks), dim3(1024), 0, st, tasks);
    if (nb) TC2LI_LAUNCH(k_map_fill, dim3(std::min(nb * 4, 64), n_tasks), dim3(256), 0, st, tasks);
    TC2LI_LAUNCH(k_map_append, dim3(n_tasks), dim3(256), 0, st, tasks);
}
void launch_map_grid_build(const MapGridTask* tasks, int n_tasks, int max_work, int max_cells, int max_row_entries, hipStream_t st) {
    const int max_points = max_work;
    if (!n_tasks) return;
    const int tiles = (max_cells + kScanTile - 1) / kScanTile;
    if (max_points) TC2LI_LAUNCH(k_map_count, dim3((max_points + 255) / 256, n_tasks), dim3(256), 0, st, tasks);
    TC2LI_LAUNCH(k_map_scan_tiles, dim3(tiles, n_tasks), dim3(1024), 0, st, tasks);
    TC2LI_LAUNCH(k_map_scan_tops, dim3(n_tasks), dim3(1024), 0, st, tasks);
    TC2LI_LAUNCH(k_map_scan_cells, dim3(tiles, n_tasks), dim3(1024), 0, st, tasks);
    TC2LI_LAUNCH(k_map_row_starts, dim3((max_row_entries + 255) / 256, n_tasks), dim3(256), 0, st, tasks);
    if (max_points) TC2LI_LAUNCH(k_map_scatter, dim3((max_points + 255) / 256, n_tasks), dim3(256), 0, st, tasks);
}

}  // namespace tc2li
