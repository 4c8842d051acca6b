// gfx950 kernels of the LiDAR front end on the camera-LiDAR path:
// preprocess (stride decimation + blind cut, ordered compaction), PCL-style voxel-grid centroid filter, uniform
// hash-grid map with exact 5-nearest-neighbour search, 5-point plane fit with the reference's gates, ordered
// compaction of the selected points.  All passes are segmented over the scans of a batch; per-scan counts stay in
// device memory so that the stages chain without host synchronisation.  Float work uses individually rounded IEEE
// operations (the library is built with -ffp-contract=off), sums are taken in the CPU order: results are bit-equal
// to the CPU path.
#include <hip/hip_runtime.h>

#include "launch.hpp"
// Bit-exactness with the CPU path needs every float operation rounded on its own: no FMA contraction (the HIP
// `__fmul_rn`-style intrinsics are plain operators unless OCML_BASIC_ROUNDED_OPERATIONS is defined, and `__fsqrt_rn` is
// the approximate native square root -- use sqrtf(), which hipcc rounds correctly by default).
#pragma clang fp contract(off)
#include <stdint.h>

#include <algorithm>
#include <cstdlib>

#include "lidar_device.hpp"
#include "intro_sort.hpp"
#include "lidar_device_fn.hpp"

namespace tc2li {

// ---- helpers ------------------------------------------------------------------------------------------------------
// Orders the LDS accesses of the lanes of ONE wavefront (waves of a block run different trip counts in the kernels
// below, so a block barrier cannot be used): release + acquire fence at workgroup scope drains lgkmcnt and stops the
// compiler from moving LDS accesses across this point.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

__global__ void k_fill_int(int* p, size_t n, int v) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
void launch_fill_int(int* p, size_t n, int v, hipStream_t st) {
    if (n == 0) return;
    const int grid = (int)std::min<size_t>((n + 255) / 256, 2048);
    TC2LI_LAUNCH(k_fill_int, dim3(grid), dim3(256), 0, st, p, n, v);
}

// ---- b1: Preprocess::velodyne_handler, non-feature branch (preprocess.cpp:145-166) ---------------------------------
__device__ __forceinline__ bool pre_keep(const VelodynePoint& p, int i, const PreprocessParams& prm) {
    if (i % prm.point_filter_num != 0) return false;
    const float r2 = p.x * p.x + p.y * p.y + p.z * p.z;
    return (double)r2 > prm.blind_sq;
}

__global__ __launch_bounds__(kSegBlock) void k_pre_count(const VelodynePoint* __restrict__ raw, const int* __restrict__ raw_count,
                                                         const ScanSlot* __restrict__ slots, const SegBlock* __restrict__ blocks,
                                                         PreprocessParams prm, int* __restrict__ block_counts) {
    const SegBlock b = blocks[blockIdx.x];
    const ScanSlot sl = slots[b.scan];
    const int i = b.start + threadIdx.x;
    bool f = false;
    if (i < raw_count[b.scan]) f = pre_keep(raw[sl.raw_base + i], i, prm);
    const int c = __syncthreads_count(f);
    if (threadIdx.x == 0) block_counts[blockIdx.x] = c;
}

// Exclusive scan of the block counts of each scan (one workgroup per scan).
__global__ __launch_bounds__(256) void k_seg_scan(const ScanSlot* __restrict__ slots, const int* __restrict__ block_counts,
                                                  int* __restrict__ block_offsets, int* __restrict__ totals) {
    __shared__ int s_part[256];
    const ScanSlot sl = slots[blockIdx.x];
    const int nb = sl.n_blocks, per = (nb + 255) / 256, tid = threadIdx.x;
    const int lo = tid * per, hi = min(lo + per, nb);
    int sum = 0;
    for (int k = lo; k < hi; ++k) sum += block_counts[sl.first_block + k];
    s_part[tid] = sum;
    __syncthreads();
    // Hillis-Steele over 256 partials
    for (int o = 1; o < 256; o <<= 1) {
        const int v = tid >= o ? s_part[tid - o] : 0;
        __syncthreads();
        s_part[tid] += v;
        __syncthreads();
    }
    int run = s_part[tid] - sum;
    for (int k = lo; k < hi; ++k) {
        block_offsets[sl.first_block + k] = run;
        run += block_counts[sl.first_block + k];
    }
    if (tid == 255) totals[blockIdx.x] = s_part[255];
}

__global__ __launch_bounds__(kSegBlock) void k_pre_scatter(const VelodynePoint* __restrict__ raw, const int* __restrict__ raw_count,
                                                           const ScanSlot* __restrict__ slots, const SegBlock* __restrict__ blocks,
                                                           PreprocessParams prm, const int* __restrict__ block_offsets,
                                                           PointXYZINormal* __restrict__ out) {
    __shared__ int s_wave[kSegBlock / 64];
    const SegBlock b = blocks[blockIdx.x];
    const ScanSlot sl = slots[b.scan];
    const int i = b.start + threadIdx.x;
    bool f = false;
    VelodynePoint p;
    if (i < raw_count[b.scan]) { p = raw[sl.raw_base + i]; f = pre_keep(p, i, prm); }
    int total;
    const int pos = block_flag_scan(f, s_wave, total);
    if (f) {
        PointXYZINormal o;
        o.x = p.x; o.y = p.y; o.z = p.z; o.pad0 = 1.0f;
        o.normal_x = 0; o.normal_y = 0; o.normal_z = 0; o.pad1 = 0;
        o.intensity = p.intensity;
        o.curvature = p.time * prm.time_unit_scale;  // milliseconds (preprocess.cpp:157)
        o.pad2 = 0; o.pad3 = 0;
        out[sl.base + block_offsets[blockIdx.x] + pos] = o;
    }
}

// ---- b3: pcl::VoxelGrid<PointXYZINormal>::filter (LidarFrontEnd.cpp:712-714, 913-915) -------------------------------
__device__ __forceinline__ bool finite3(const PointXYZINormal& p) { return isfinite(p.x) && isfinite(p.y) && isfinite(p.z); }

// b1 for a batch of scans in ONE pass over the raw points (round 4; k_pre_count + k_seg_scan + k_pre_scatter read every raw scan twice):
// one workgroup per scan walks it 1024 points at a time with a running count of the points kept so far -- the output order is the
// scan's -- and leaves, in the same pass, the bounding box of the kept finite points where the voxel filter's k_voxel_bbox would put it
// (one more read of the preprocessed cloud saved when the filter follows directly).  Only every point_filter_num-th record can be kept:
// the others are not even loaded.  The next chunk's records are requested before this chunk's barriers.
__global__ __launch_bounds__(kSegBlock) void k_pre_stream(const VelodynePoint* __restrict__ raw, const int* __restrict__ raw_count,
                                                          const ScanSlot* __restrict__ slots, PreprocessParams prm, PointXYZINormal* __restrict__ out,
                                                          int* __restrict__ out_count, int* __restrict__ bbox_enc, float* __restrict__ time_out,
                                                          uint32_t* __restrict__ vkey_out, float inv_leaf, int* __restrict__ vk_ok) {
    // vkey_out (may be NULL): the kept points' voxel coordinates floor(p / leaf), packed 11 | 11 | 10 bits (x | y | z, two's complement), for the
    // voxel filter that follows at that leaf -- its sort otherwise opens every 48-byte record again for the 12 bytes of the position
    // (1.44 GB of the 3.4 GB k_voxel_sort_points moved per 512 scans).  vk_ok[scan] = 0 when a point does not fit (not finite, or beyond
    // +-1024 / +-512 leaves): the filter then reads the points as before.
    // time_out (may be NULL): the kept points' time stamps (curvature) as an array of their own, for UndistortPcl's time sort -- which
    // otherwise opens every 48-byte record again for 4 bytes of it (1.44 GB of its 2.4 GB per 512 scans, r05_pmc_traffic_inertial.json)
    __shared__ int s_wave[kSegBlock / 64];
    __shared__ int s_min[3], s_max[3], s_bad;
    const int s = blockIdx.x, tid = threadIdx.x;
    const ScanSlot sl = slots[s];
    const int n = raw_count[s];
    if (tid < 3) { s_min[tid] = 0x7fffffff; s_max[tid] = (int)0x80000000; }
    if (tid == 0) s_bad = 0;
    bool bad = false;
    const float4* __restrict__ src = reinterpret_cast<const float4*>(raw + sl.raw_base);
    auto wanted = [&](int i) { return i < n && i % prm.point_filter_num == 0; };
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    if (wanted(tid)) { a = src[2 * (size_t)tid]; b = src[2 * (size_t)tid + 1]; }
    int mn[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, mx[3] = {(int)0x80000000, (int)0x80000000, (int)0x80000000};
    int kept = 0;
    for (int c0 = 0; c0 < n; c0 += kSegBlock) {
        const int i = c0 + tid;
        float4 na = make_float4(0.f, 0.f, 0.f, 0.f), nb = na;
        if (wanted(i + kSegBlock)) { na = src[2 * (size_t)(i + kSegBlock)]; nb = src[2 * (size_t)(i + kSegBlock) + 1]; }
        // VelodynePoint: x y z pad0 | intensity time (ring, pad1) pad2
        const bool f = wanted(i) && (double)(a.x * a.x + a.y * a.y + a.z * a.z) > prm.blind_sq;
        int total;
        const int pos = block_flag_scan(f, s_wave, total);
        if (f) {
            PointXYZINormal o;
            o.x = a.x; o.y = a.y; o.z = a.z; o.pad0 = 1.0f;
            o.normal_x = 0; o.normal_y = 0; o.normal_z = 0; o.pad1 = 0;
            o.intensity = b.x;
            o.curvature = b.y * prm.time_unit_scale;  // milliseconds (preprocess.cpp:157)
            o.pad2 = 0; o.pad3 = 0;
            out[sl.base + kept + pos] = o;
            if (time_out) time_out[sl.base + kept + pos] = o.curvature;
            if (vkey_out) {
                const float fx = floorf(o.x * inv_leaf), fy = floorf(o.y * inv_leaf), fz = floorf(o.z * inv_leaf);  // voxel_index's own terms
                const bool fits = finite3(o) && fx >= -1024.f && fx <= 1023.f && fy >= -1024.f && fy <= 1023.f && fz >= -512.f && fz <= 511.f;
                bad |= !fits;
                const int ix = fits ? (int)fx : 0, iy = fits ? (int)fy : 0, iz = fits ? (int)fz : 0;
                vkey_out[sl.base + kept + pos] = ((uint32_t)(iz & 0x3ff) << 22) | ((uint32_t)(iy & 0x7ff) << 11) | (uint32_t)(ix & 0x7ff);
            }
            if (finite3(o)) {
                mn[0] = min(mn[0], enc_float(o.x)); mx[0] = max(mx[0], enc_float(o.x));
                mn[1] = min(mn[1], enc_float(o.y)); mx[1] = max(mx[1], enc_float(o.y));
                mn[2] = min(mn[2], enc_float(o.z)); mx[2] = max(mx[2], enc_float(o.z));
            }
        }
        kept += total;
        a = na; b = nb;
    }
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            mn[ax] = min(mn[ax], __shfl_xor(mn[ax], o, 64));
            mx[ax] = max(mx[ax], __shfl_xor(mx[ax], o, 64));
        }
    }
    if ((tid & 63) == 0) {
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) { atomicMin(&s_min[ax], mn[ax]); atomicMax(&s_max[ax], mx[ax]); }
    }
    __syncthreads();
    if (tid < 3) { bbox_enc[s * 6 + tid] = s_min[tid]; bbox_enc[s * 6 + 3 + tid] = s_max[tid]; }
    if (tid == 0) out_count[s] = kept;
    if (vk_ok) {
        if (bad) atomicOr(&s_bad, 1);
        __syncthreads();
        if (tid == 0) vk_ok[s] = s_bad ? 0 : 1;
    }
}

__global__ __launch_bounds__(kSegBlock) void k_voxel_bbox(const PointXYZINormal* __restrict__ pts, const int* __restrict__ count,
                                                          const ScanSlot* __restrict__ slots, const SegBlock* __restrict__ blocks,
                                                          int* __restrict__ bbox_enc) {
    __shared__ int s_min[3], s_max[3];
    const SegBlock b = blocks[blockIdx.x];
    if (b.start >= count[b.scan]) return;
    const ScanSlot sl = slots[b.scan];
    if (threadIdx.x < 3) { s_min[threadIdx.x] = 0x7fffffff; s_max[threadIdx.x] = (int)0x80000000; }
    __syncthreads();
    const int i = b.start + threadIdx.x;
    int mn[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, mx[3] = {(int)0x80000000, (int)0x80000000, (int)0x80000000};
    if (i < count[b.scan]) {
        const PointXYZINormal p = pts[sl.base + i];
        if (finite3(p)) {
            mn[0] = mx[0] = enc_float(p.x);
            mn[1] = mx[1] = enc_float(p.y);
            mn[2] = mx[2] = enc_float(p.z);
        }
    }
    // one LDS atomic per wavefront and axis instead of one per point (they all hit the same six words)
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            mn[a] = min(mn[a], __shfl_xor(mn[a], o, 64));
            mx[a] = max(mx[a], __shfl_xor(mx[a], o, 64));
        }
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { atomicMin(&s_min[a], mn[a]); atomicMax(&s_max[a], mx[a]); }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        atomicMin(&bbox_enc[b.scan * 6 + threadIdx.x], s_min[threadIdx.x]);
        atomicMax(&bbox_enc[b.scan * 6 + 3 + threadIdx.x], s_max[threadIdx.x]);
    }
}

__global__ void k_voxel_params(const int* __restrict__ bbox_enc, const int* __restrict__ count, int nscans, float leaf,
                               VoxelParams* __restrict__ vp) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nscans) return;
    VoxelParams v = vp[s];  // table_base / table_mask were filled by the host
    const float inv = 1.0f / leaf;
    int div_b[3];
    long long cells = 1;
    bool any = count[s] > 0 && bbox_enc[s * 6] != 0x7fffffff;
    for (int a = 0; a < 3; ++a) {
        const float mn = any ? dec_float(bbox_enc[s * 6 + a]) : 0.f, mx = any ? dec_float(bbox_enc[s * 6 + 3 + a]) : 0.f;
        cells *= (long long)((mx - mn) * inv) + 1;
        v.min_b[a] = (int)floorf(mn * inv);
        div_b[a] = (int)floorf(mx * inv) - v.min_b[a] + 1;
    }
    v.passthrough = cells > 2147483647ll ? 1 : 0;
    v.mul[0] = 1; v.mul[1] = div_b[0]; v.mul[2] = div_b[0] * div_b[1];
    vp[s] = v;
}

__device__ __forceinline__ int voxel_index(const PointXYZINormal& p, float inv, const VoxelParams& v) {
    const int i0 = (int)(floorf(p.x * inv) - (float)v.min_b[0]);
    const int i1 = (int)(floorf(p.y * inv) - (float)v.min_b[1]);
    const int i2 = (int)(floorf(p.z * inv) - (float)v.min_b[2]);
    return i0 * v.mul[0] + i1 * v.mul[1] + i2 * v.mul[2];
}
// Launch-order -> work-list index so that XCD k (workgroups reach the XCDs round-robin by linear index; grid.x is rounded up to a
// multiple of 8) takes the k-th contiguous eighth of the list: the blocks of a scan, and with them its hash table, member lists and map
// grid, meet in one L2 instead of eight.  -1: padding block.
__device__ __forceinline__ int xcd_contiguous(int bid, int n) {
    const int per = (n + 7) >> 3, l = (bid & 7) * per + (bid >> 3);
    return l < n ? l : -1;
}
__device__ __forceinline__ uint32_t hash_u32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// Neighbouring returns of a scan mostly fall into the same voxel: lanes that continue their left neighbour's voxel (a "run")
// leave the table work to the run's first lane -- one CAS probe and one counter atomic per run instead of per point -- and take
// its slot by shuffle.  pt_slot keeps every point's table slot for the later passes (-1: point not filtered).
constexpr int kMaxVoxelsPerScan = 32768;  // voxels of one scan the LDS sort holds
constexpr int kInsertThreads = 256;  // a block of the list = kSegBlock points = kSegBlock / kInsertThreads workgroups (blockIdx.y): the barrier
// below waits for the slowest probe chain of the workgroup, and 1024 lanes wait longer than 256
__global__ __launch_bounds__(kInsertThreads) void k_voxel_insert(const PointXYZINormal* __restrict__ pts, const int* __restrict__ count,
                                                            const ScanSlot* __restrict__ slots, const SegBlock* __restrict__ blocks,
                                                            float leaf, const VoxelParams* __restrict__ vp,
                                                            int* __restrict__ table_keys, int* __restrict__ table_counts,
                                                            int* __restrict__ pt_slot, int* __restrict__ n_vox, int* __restrict__ vox_keys, int nblocks) {
    const int bi = xcd_contiguous((int)blockIdx.x, nblocks);
    if (bi < 0) return;
    const SegBlock b = blocks[bi];
    const int first = b.start + (int)blockIdx.y * kInsertThreads;
    const int i = first + (int)threadIdx.x, n = count[b.scan];
    if (first >= n) return;  // whole workgroup
    const VoxelParams v = vp[b.scan];
    if (v.passthrough) return;
    const int base = slots[b.scan].base;
    bool valid = i < n;
    int idx = -1 - (int)(threadIdx.x & 63);  // lanes without a point: keys no neighbour shares
    if (valid) {
        const PointXYZINormal p = pts[base + i];
        valid = finite3(p);
        if (valid) idx = voxel_index(p, 1.0f / leaf, v);
    }
    __shared__ int s_new[kInsertThreads], s_n_new, s_first;
    if (threadIdx.x == 0) s_n_new = 0;
    __syncthreads();
    const RunInfo run = wave_runs(idx);
    int slot = -1;
    if (run.head && valid) {
        uint32_t h = hash_u32((uint32_t)idx) & (uint32_t)v.table_mask;
        for (;;) {
            const int prev = atomicCAS(&table_keys[v.table_base + h], -1, idx);
            if (prev == -1) { s_new[atomicAdd(&s_n_new, 1)] = idx; break; }  // a voxel nobody had seen: listed below
            if (prev == idx) break;
            h = (h + 1) & (uint32_t)v.table_mask;
        }
        slot = v.table_base + (int)h;
        atomicAdd(&table_counts[slot], run.length);
    }
    slot = __shfl(slot, run.head_lane, 64);
    if (i < n) pt_slot[base + i] = valid ? slot : -1;
    // the keys of the scan's voxels in arrival order: the sort reads this list, not the 2 x capacity table; one global atomic per workgroup
    __syncthreads();
    const int n_new = s_n_new;
    if (threadIdx.x == 0 && n_new > 0) s_first = atomicAdd(&n_vox[b.scan], n_new);
    __syncthreads();
    if ((int)threadIdx.x < n_new) {
        const int pos = s_first + (int)threadIdx.x;
        if (pos < kMaxVoxelsPerScan) vox_keys[base + pos] = s_new[threadIdx.x];
    }
}

__device__ __forceinline__ int table_find(const int* __restrict__ table_keys, const VoxelParams& v, int idx) {
    uint32_t h = hash_u32((uint32_t)idx) & (uint32_t)v.table_mask;
    while (table_keys[v.table_base + h] != idx) h = (h + 1) & (uint32_t)v.table_mask;
    return v.table_base + (int)h;
}

// One workgroup per scan: gather the occupied voxel keys, bitonic-sort them ascending in LDS (PCL emits voxels in
// ascending index order), publish rank and member offsets.
__global__ __launch_bounds__(1024) void k_voxel_sort(const ScanSlot* __restrict__ slots, const VoxelParams* __restrict__ vp,
                                                     const int* __restrict__ count, const int* __restrict__ table_keys,
                                                     const int* __restrict__ table_counts, int* __restrict__ table_rank,
                                                     int* __restrict__ vox_keys, int* __restrict__ vox_member_off,
                                                     int* __restrict__ vox_fill, int* __restrict__ vox_count, int* __restrict__ n_vox, int* __restrict__ status) {
    extern __shared__ int s_keys[];  // kMaxVoxelsPerScan
    __shared__ int s_n;
    __shared__ int s_part[1024];
    const int s = blockIdx.x, tid = threadIdx.x;
    const ScanSlot sl = slots[s];
    const VoxelParams v = vp[s];
    if (v.passthrough) { if (tid == 0) n_vox[s] = count[s]; return; }
    if (tid == 0) s_n = n_vox[s];  // the insert pass counted the scan's voxels and listed their keys (in arrival order)
    __syncthreads();
    int n = s_n;
    if (n > kMaxVoxelsPerScan) { if (tid == 0) { atomicExch(status, 1); n_vox[s] = 0; } return; }
    for (int k = tid; k < n; k += 1024) s_keys[k] = vox_keys[sl.base + k];
    __syncthreads();
    int np2 = 1;
    while (np2 < n) np2 <<= 1;
    for (int k = n + tid; k < np2; k += 1024) s_keys[k] = 0x7fffffff;
    __syncthreads();
    for (int size = 2; size <= np2; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int k = tid; k < (np2 >> 1); k += 1024) {
                const int lo = (k / stride) * (stride << 1) + (k % stride), hi = lo + stride;
                const bool up = ((lo & size) == 0);
                const int a = s_keys[lo], b = s_keys[hi];
                if ((a > b) == up) { s_keys[lo] = b; s_keys[hi] = a; }
            }
            __syncthreads();
        }
    // rank -> table, member counts -> exclusive offsets (scan in global memory, this block only)
    const int per = (n + 1023) / 1024, lo = tid * per, hi = min(lo + per, n);
    int sum = 0;
    for (int r = lo; r < hi; ++r) {
        const int key = s_keys[r];
        const int slot = table_find(table_keys, v, key);
        table_rank[slot] = r;
        vox_keys[sl.base + r] = key;
        sum += table_counts[slot];
    }
    s_part[tid] = sum;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int t = tid >= o ? s_part[tid - o] : 0;
        __syncthreads();
        s_part[tid] += t;
        __syncthreads();
    }
    int run = s_part[tid] - sum;
    for (int r = lo; r < hi; ++r) {
        vox_member_off[sl.base + r] = run;
        vox_fill[sl.base + r] = 0;  // the fill pass counts the voxel's runs here: only the scan's voxels need clearing
        const int cnt = table_counts[table_find(table_keys, v, s_keys[r])];
        vox_count[sl.base + r] = cnt;
        run += cnt;
    }
    if (tid == 0) n_vox[s] = n;
}

// A voxel's members are recorded as RUNS: lanes that continue their left neighbour's voxel are consecutive point indices, so one word
// per run (first index | (length - 1) << 24) says everything the ranking pass needs, and only the run's first lane touches memory:
// one counter atomic hands out the run's place.
__global__ __launch_bounds__(kSegBlock) void k_voxel_fill(const int* __restrict__ count, const ScanSlot* __restrict__ slots,
                                                          const SegBlock* __restrict__ blocks, const VoxelParams* __restrict__ vp,
                                                          const int* __restrict__ pt_slot, const int* __restrict__ table_rank,
                                                          const int* __restrict__ vox_member_off, int* __restrict__ vox_fill,
                                                          int* __restrict__ members, int nblocks) {
    if ((int)blockIdx.x >= nblocks) return;  // launch order kept: the XCD-contiguous order doubled this kernel's time (its position atomics)
    const SegBlock b = blocks[blockIdx.x];
    const int i = b.start + threadIdx.x, n = count[b.scan];
    if (b.start >= n) return;
    if (vp[b.scan].passthrough) return;
    const ScanSlot sl = slots[b.scan];
    const int slot = i < n ? pt_slot[sl.base + i] : -1;
    const RunInfo run = wave_runs(slot >= 0 ? slot : -1 - (int)(threadIdx.x & 63));
    if (run.head && slot >= 0) {
        const int r = table_rank[slot];
        const int k = atomicAdd(&vox_fill[sl.base + r], 1);
        members[sl.base + vox_member_off[sl.base + r] + k] = i | ((run.length - 1) << 24);
    }
}

// PCL sums a voxel's points in the order of its sorted index vector, i.e. by ascending point index, in float.
// k_voxel_rank: one thread per POINT finds its rank inside its voxel (count of member indices below its own; the members
// were filled in atomic order) and writes the point's fields to that position of a record array -- the quadratic part
// of the work is spread evenly over the whole grid whatever the voxel populations are (dense voxels near the sensor are
// neighbours in voxel order: a block-per-voxel-range layout left a few workgroups with most of the work).
// k_voxel_centroid: one thread per voxel adds its now contiguous records sequentially -- the exact summation order of
// pcl::CentroidPoint -- and divides by the count.
struct CentroidRec { float4 lo, hi; };  // x y z normal_x | normal_y normal_z intensity curvature

__global__ __launch_bounds__(256) void k_voxel_rank(const PointXYZINormal* __restrict__ pts, const int* __restrict__ count,
                                                    const ScanSlot* __restrict__ slots, const SegBlock* __restrict__ blocks, float leaf,
                                                    const VoxelParams* __restrict__ vp, const int* __restrict__ pt_slot,
                                                    const int* __restrict__ table_rank, const int* __restrict__ vox_member_off,
                                                    const int* __restrict__ vox_fill, const int* __restrict__ members,
                                                    CentroidRec* __restrict__ recs, int nblocks) {
    const int bi = xcd_contiguous((int)blockIdx.x, nblocks);
    if (bi < 0) return;
    const SegBlock b = blocks[bi];
    const int i = b.start + (int)blockIdx.y * 256 + threadIdx.x;
    if (i >= count[b.scan]) return;
    const VoxelParams v = vp[b.scan];
    if (v.passthrough) return;
    const ScanSlot sl = slots[b.scan];
    const int slot = pt_slot[sl.base + i];
    if (slot < 0) return;
    const PointXYZINormal p = pts[sl.base + i];
    const int r = table_rank[slot];
    const int off = vox_member_off[sl.base + r], n = vox_fill[sl.base + r];  // runs of the voxel
    const int* m = members + sl.base + off;
    // points of the voxel with a smaller index: of a run [s, s + L) that is clamp(i - s, 0, L)
    int rank = 0, k = 0;
    for (; k + 4 <= n; k += 4) {  // four independent loads in flight
        const int a0 = m[k], a1 = m[k + 1], a2 = m[k + 2], a3 = m[k + 3];
        rank += min(max(i - (a0 & 0xffffff), 0), (a0 >> 24) + 1) + min(max(i - (a1 & 0xffffff), 0), (a1 >> 24) + 1) +
                min(max(i - (a2 & 0xffffff), 0), (a2 >> 24) + 1) + min(max(i - (a3 & 0xffffff), 0), (a3 >> 24) + 1);
    }
    for (; k < n; ++k) { const int a = m[k]; rank += min(max(i - (a & 0xffffff), 0), (a >> 24) + 1); }
    CentroidRec rec;
    rec.lo = make_float4(p.x, p.y, p.z, p.normal_x);
    rec.hi = make_float4(p.normal_y, p.normal_z, p.intensity, p.curvature);
    recs[sl.base + off + rank] = rec;
}

__global__ __launch_bounds__(256) void k_voxel_centroid(const PointXYZINormal* __restrict__ pts, const int* __restrict__ count,
                                                        const ScanSlot* __restrict__ slots, const SegBlock* __restrict__ blocks,
                                                        const VoxelParams* __restrict__ vp, const int* __restrict__ n_vox,
                                                        const int* __restrict__ vox_member_off, const int* __restrict__ vox_count,
                                                        const CentroidRec* __restrict__ recs, PointXYZINormal* __restrict__ out,
                                                        int* __restrict__ out_count, int nblocks) {
    const int bi = xcd_contiguous((int)blockIdx.x, nblocks);
    if (bi < 0) return;
    const SegBlock b = blocks[bi];
    const ScanSlot sl = slots[b.scan];
    const int nv = n_vox[b.scan];
    if (b.start == 0 && blockIdx.y == 0 && threadIdx.x == 0) out_count[b.scan] = nv;
    const int r = b.start + (int)blockIdx.y * 256 + threadIdx.x;
    if (r >= nv) return;
    if (vp[b.scan].passthrough != 0) { out[sl.base + r] = pts[sl.base + r]; return; }
    const int n = vox_count[sl.base + r];
    const CentroidRec* __restrict__ q = recs + sl.base + vox_member_off[sl.base + r];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f, a5 = 0.f, a6 = 0.f, a7 = 0.f;
    int k = 0;
    for (; k + 4 <= n; k += 4) {  // loads of four records issued together, additions strictly in order
        const CentroidRec r0 = q[k], r1 = q[k + 1], r2 = q[k + 2], r3 = q[k + 3];
        a0 += r0.lo.x; a1 += r0.lo.y; a2 += r0.lo.z; a3 += r0.lo.w; a4 += r0.hi.x; a5 += r0.hi.y; a6 += r0.hi.z; a7 += r0.hi.w;
        a0 += r1.lo.x; a1 += r1.lo.y; a2 += r1.lo.z; a3 += r1.lo.w; a4 += r1.hi.x; a5 += r1.hi.y; a6 += r1.hi.z; a7 += r1.hi.w;
        a0 += r2.lo.x; a1 += r2.lo.y; a2 += r2.lo.z; a3 += r2.lo.w; a4 += r2.hi.x; a5 += r2.hi.y; a6 += r2.hi.z; a7 += r2.hi.w;
        a0 += r3.lo.x; a1 += r3.lo.y; a2 += r3.lo.z; a3 += r3.lo.w; a4 += r3.hi.x; a5 += r3.hi.y; a6 += r3.hi.z; a7 += r3.hi.w;
    }
    for (; k < n; ++k) {
        const CentroidRec r0 = q[k];
        a0 += r0.lo.x; a1 += r0.lo.y; a2 += r0.lo.z; a3 += r0.lo.w; a4 += r0.hi.x; a5 += r0.hi.y; a6 += r0.hi.z; a7 += r0.hi.w;
    }
    const float fn = (float)n;
    PointXYZINormal o;
    o.x = a0 / fn; o.y = a1 / fn; o.z = a2 / fn; o.pad0 = 1.0f;
    float snx = a3, sny = a4, snz = a5;
    const float nn = snx * snx + sny * sny + snz * snz;
    if (nn > 0) { const float rt = sqrtf(nn); snx /= rt; sny /= rt; snz /= rt; }
    o.normal_x = snx; o.normal_y = sny; o.normal_z = snz; o.pad1 = 0;
    o.intensity = a6 / fn; o.curvature = a7 / fn; o.pad2 = 0; o.pad3 = 0;
    out[sl.base + r] = o;
}

// ---- b3, sorted form (batches of scans): the index vector of VoxelGrid::applyFilter sorted for real ----------------------
// The hash form above finds a point's voxel through a table (CAS probes), lists the voxels, sorts them, hands out member places with
// counter atomics and restores the order inside a voxel by ranking: five passes over the points, two table clears of 2 x capacity per
// scan.  applyFilter itself does one thing: it sorts (voxel index, point index).  Here ONE workgroup of 1024 threads per scan does that
// with a stable LSD radix sort, eight bits per pass, only as many passes as the scan's largest voxel index has digits (3 for a
// 100 m scan at 0.5 m), a pass whose digit is the same for every point skipped:
//   * first sweep: voxel indices of the finite points, compacted in point order; the digit histograms of all passes (a wavefront's
//     equal digits found by eight ballots, one LDS atomic per group);
//   * a pass walks the list in chunks of 2048 keys (two per lane): equal digits inside a 64-key group by ballots, group totals to LDS,
//     256 lanes lay the 32 groups of the chunk behind each other per digit, every key goes to (digit's running place + groups before
//     its own + equal keys before it in its group): stable, so points of a voxel stay in ascending index order;
//   * last sweep: the first entry of every voxel by a running prefix count -- voxels in ascending index order, as PCL emits them.
// k_voxel_gather_sorted writes the points' fields in that order, k_voxel_centroid_sorted adds every voxel's records (pcl::CentroidPoint's order).
// Same centroids bit for bit as the hash form (tests/test_lidar_gpu.py runs both); the hash form stays for calls with few scans,
// where one workgroup per scan would leave the GPU empty.
constexpr int kVsThreads = 1024, kVsChunk = 2 * kVsThreads;

__device__ __forceinline__ unsigned long long match8(uint32_t d, unsigned long long valid) {  // the lanes of `valid` with the same 8-bit digit
    unsigned long long m = valid;
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const bool bit = (d >> b) & 1;
        const unsigned long long bal = __ballot(bit);
        m &= bit ? bal : ~bal;
    }
    return m;
}
__device__ __forceinline__ int vs_block_scan(int v, int* s_wave, int& total) {  // exclusive prefix over the 1024 threads; two barriers
    const int lane = threadIdx.x & 63, wave = wave_in_block();
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int before = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < kVsThreads / 64; ++k) { const int w = s_wave[k]; before += k < wave ? w : 0; tot += w; }
    total = tot;
    __syncthreads();
    return before + incl - v;
}

// vox_info[base]: finite points of the scan, vox_info[base + 1]: which buffer holds the sorted lists (0: A, 1: B)
__global__ __launch_bounds__(kVsThreads) void k_voxel_sort_points(const PointXYZINormal* __restrict__ pts, const int* __restrict__ count,
                                                                  const ScanSlot* __restrict__ slots, const VoxelParams* __restrict__ vp, float leaf,
                                                                  uint32_t* __restrict__ key_a, int* __restrict__ idx_a, uint32_t* __restrict__ key_b,
                                                                  int* __restrict__ idx_b, int* __restrict__ vox_start, int* __restrict__ vox_info,
                                                                  int* __restrict__ n_vox, const int* __restrict__ vk_ok) {
    // vk_ok (may be NULL): vk_ok[scan] != 0 -- key_b holds the points' packed voxel coordinates, left by k_pre_stream for this leaf
    __shared__ int s_hist[4][256], s_place[256], s_cstart[256], s_wave[kVsThreads / 64];
    __shared__ unsigned short s_wcnt[32][256], s_woff[32][256];
    __shared__ unsigned int s_max;
    const int s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = wave_in_block();
    const int n = count[s], base = slots[s].base;
    const VoxelParams v = vp[s];
    if (v.passthrough) { if (tid == 0) n_vox[s] = n; return; }
    for (int k = tid; k < 4 * 256; k += kVsThreads) (&s_hist[0][0])[k] = 0;
    if (tid == 0) s_max = 0;
    __syncthreads();
    // ---- first sweep: keys of the finite points in point order, digit histograms ----
    const float inv = 1.0f / leaf;
    const bool use_keys = vk_ok != nullptr && vk_ok[s] != 0;  // uniform
    const unsigned long long lt = (1ull << lane) - 1ull;
    int n_pts = 0;
    unsigned int my_max = 0;
    for (int c0 = 0; c0 < n; c0 += kVsThreads) {
        const int i = c0 + tid;
        bool valid = false;
        uint32_t key = 0;
        if (i < n && use_keys) {  // floor(p / leaf) per axis as k_pre_stream packed it: voxel_index's arithmetic on the same float values
            const uint32_t pk = key_b[base + i];
            const int ix = (int)(pk << 21) >> 21, iy = (int)(pk << 10) >> 21, iz = (int)pk >> 22;
            const int i0 = (int)((float)ix - (float)v.min_b[0]), i1 = (int)((float)iy - (float)v.min_b[1]), i2 = (int)((float)iz - (float)v.min_b[2]);
            valid = true;
            key = (uint32_t)(i0 * v.mul[0] + i1 * v.mul[1] + i2 * v.mul[2]);
        } else if (i < n) {
            const float4 xyz = *reinterpret_cast<const float4*>(&pts[base + i]);
            PointXYZINormal p;
            p.x = xyz.x; p.y = xyz.y; p.z = xyz.z;
            valid = finite3(p);
            if (valid) key = (uint32_t)voxel_index(p, inv, v);
        }
        int tot;
        const int ex = vs_block_scan(valid ? 1 : 0, s_wave, tot);
        if (valid) { key_a[base + n_pts + ex] = key; idx_a[base + n_pts + ex] = i; my_max = max(my_max, key); }
        const unsigned long long vb = __ballot(valid);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const uint32_t d = (key >> (8 * p)) & 255;
            const unsigned long long m = match8(d, vb);
            if (valid && (m & lt) == 0) atomicAdd(&s_hist[p][d], __popcll(m));  // the group's first lane
        }
        n_pts += tot;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) my_max = max(my_max, (unsigned int)__shfl_xor((int)my_max, o, 64));
    if (lane == 0) atomicMax(&s_max, my_max);
    __syncthreads();
    int passes = (32 - __clz((int)(s_max | 1u)) + 7) / 8;
    uint32_t *src_k = key_a + base, *dst_k = key_b + base;
    int *src_i = idx_a + base, *dst_i = idx_b + base;
    int which = 0;
    for (int p = 0; p < passes; ++p) {
        const int sh = 8 * p;
        // ---- the digits' first places; a pass in which every key has the same digit moves nothing ----
        const int hv = tid < 256 ? s_hist[p][tid] : 0;
        const bool skip = __syncthreads_or(tid < 256 && hv == n_pts && n_pts > 0) != 0;
        int tot;
        const int ex = vs_block_scan(hv, s_wave, tot);
        if (tid < 256) s_place[tid] = ex;
        __syncthreads();
        if (skip) continue;
        for (int c0 = 0; c0 < n_pts; c0 += kVsChunk) {
            for (int k = tid; k < 32 * 256 / 2; k += kVsThreads) reinterpret_cast<uint32_t*>(&s_wcnt[0][0])[k] = 0u;
            __syncthreads();
            uint32_t key[2], dig[2];
            int id[2], rank[2];
            bool val[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int i = c0 + wave * 128 + j * 64 + lane;
                val[j] = i < n_pts;
                key[j] = val[j] ? src_k[i] : 0u;
                id[j] = val[j] ? src_i[i] : 0;
                dig[j] = (key[j] >> sh) & 255;
                const unsigned long long m = match8(dig[j], __ballot(val[j]));
                rank[j] = __popcll(m & lt);
                if (val[j] && rank[j] == 0) s_wcnt[2 * wave + j][dig[j]] = (unsigned short)__popcll(m);
            }
            __syncthreads();
            if (tid < 256) {  // the chunk's 32 groups behind each other, per digit
                int run = 0;
#pragma unroll 8
                for (int w = 0; w < 32; ++w) { const int c = s_wcnt[w][tid]; s_woff[w][tid] = (unsigned short)run; run += c; }
                const int at = s_place[tid];
                s_cstart[tid] = at;
                s_place[tid] = at + run;
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 2; ++j)
                if (val[j]) {
                    const int pos = s_cstart[dig[j]] + (int)s_woff[2 * wave + j][dig[j]] + rank[j];
                    dst_k[pos] = key[j];
                    dst_i[pos] = id[j];
                }
        }
        { uint32_t* t = src_k; src_k = dst_k; dst_k = t; }
        { int* t = src_i; src_i = dst_i; dst_i = t; }
        which ^= 1;
        __syncthreads();  // (device scope not needed: the workgroup reads its own writes after the barrier)
    }
    // ---- the voxels: first entry of every run of equal keys, in order ----
    __threadfence_block();
    int nv = 0;
    for (int c0 = 0; c0 < n_pts; c0 += kVsThreads) {
        const int i = c0 + tid;
        const bool head = i < n_pts && (i == 0 || src_k[i] != src_k[i - 1]);
        int tot;
        const int ex = vs_block_scan(head ? 1 : 0, s_wave, tot);
        if (head) vox_start[base + nv + ex] = i;
        nv += tot;
    }
    if (tid == 0) { n_vox[s] = nv; vox_info[base] = n_pts; vox_info[base + 1] = which; }
}

// the points' fields in sorted order: one thread per entry, so every gather of the scan is in flight at once (a thread per voxel walking
// its members paid one round trip per member: 0.28 ms per 64 scans against 0.1 for both kernels below)
__global__ __launch_bounds__(256) void k_voxel_gather_sorted(const PointXYZINormal* __restrict__ pts, const ScanSlot* __restrict__ slots,
                                                             const SegBlock* __restrict__ blocks, const VoxelParams* __restrict__ vp,
                                                             const int* __restrict__ vox_info, const int* __restrict__ idx_a,
                                                             const int* __restrict__ idx_b, CentroidRec* __restrict__ recs, int nblocks) {
    const int bi = xcd_contiguous((int)blockIdx.x, nblocks);
    if (bi < 0) return;
    const SegBlock b = blocks[bi];
    if (vp[b.scan].passthrough != 0) return;
    const int base = slots[b.scan].base;
    const int i = b.start + (int)blockIdx.y * 256 + threadIdx.x;
    if (i >= vox_info[base]) return;
    const PointXYZINormal p = pts[base + (vox_info[base + 1] ? idx_b : idx_a)[base + i]];
    CentroidRec rec;
    rec.lo = make_float4(p.x, p.y, p.z, p.normal_x);
    rec.hi = make_float4(p.normal_y, p.normal_z, p.intensity, p.curvature);
    recs[base + i] = rec;
}

__global__ __launch_bounds__(256) void k_voxel_centroid_sorted(const PointXYZINormal* __restrict__ pts, const ScanSlot* __restrict__ slots,
                                                               const SegBlock* __restrict__ blocks, const VoxelParams* __restrict__ vp,
                                                               const int* __restrict__ n_vox, const int* __restrict__ vox_start,
                                                               const int* __restrict__ vox_info, const CentroidRec* __restrict__ recs,
                                                               PointXYZINormal* __restrict__ out, int* __restrict__ out_count, int nblocks) {
    const int bi = xcd_contiguous((int)blockIdx.x, nblocks);
    if (bi < 0) return;
    const SegBlock b = blocks[bi];
    const ScanSlot sl = slots[b.scan];
    const int nv = n_vox[b.scan];
    if (b.start == 0 && blockIdx.y == 0 && threadIdx.x == 0) out_count[b.scan] = nv;
    const int r = b.start + (int)blockIdx.y * 256 + threadIdx.x;
    if (r >= nv) return;
    if (vp[b.scan].passthrough != 0) { out[sl.base + r] = pts[sl.base + r]; return; }
    const int first = vox_start[sl.base + r], last = r + 1 < nv ? vox_start[sl.base + r + 1] : vox_info[sl.base];
    const CentroidRec* __restrict__ q = recs + sl.base;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f, a5 = 0.f, a6 = 0.f, a7 = 0.f;
    int k = first;
    for (; k + 4 <= last; k += 4) {  // loads of four records issued together, additions strictly in order
        const CentroidRec r0 = q[k], r1 = q[k + 1], r2 = q[k + 2], r3 = q[k + 3];
        a0 += r0.lo.x; a1 += r0.lo.y; a2 += r0.lo.z; a3 += r0.lo.w; a4 += r0.hi.x; a5 += r0.hi.y; a6 += r0.hi.z; a7 += r0.hi.w;
        a0 += r1.lo.x; a1 += r1.lo.y; a2 += r1.lo.z; a3 += r1.lo.w; a4 += r1.hi.x; a5 += r1.hi.y; a6 += r1.hi.z; a7 += r1.hi.w;
        a0 += r2.lo.x; a1 += r2.lo.y; a2 += r2.lo.z; a3 += r2.lo.w; a4 += r2.hi.x; a5 += r2.hi.y; a6 += r2.hi.z; a7 += r2.hi.w;
        a0 += r3.lo.x; a1 += r3.lo.y; a2 += r3.lo.z; a3 += r3.lo.w; a4 += r3.hi.x; a5 += r3.hi.y; a6 += r3.hi.z; a7 += r3.hi.w;
    }
    for (; k < last; ++k) {
        const CentroidRec r0 = q[k];
        a0 += r0.lo.x; a1 += r0.lo.y; a2 += r0.lo.z; a3 += r0.lo.w; a4 += r0.hi.x; a5 += r0.hi.y; a6 += r0.hi.z; a7 += r0.hi.w;
    }
    const float fn = (float)(last - first);
    PointXYZINormal o;
    o.x = a0 / fn; o.y = a1 / fn; o.z = a2 / fn; o.pad0 = 1.0f;
    float snx = a3, sny = a4, snz = a5;
    const float nn = snx * snx + sny * sny + snz * snz;
    if (nn > 0) { const float rt = sqrtf(nn); snx /= rt; sny /= rt; snz /= rt; }
    o.normal_x = snx; o.normal_y = sny; o.normal_z = snz; o.pad1 = 0;
    o.intensity = a6 / fn; o.curvature = a7 / fn; o.pad2 = 0; o.pad3 = 0;
    out[sl.base + r] = o;
}

// gather + centroid in one pass (round 4): a thread per voxel takes its members' indices four at a time, then their points, and adds them
// in the members' order -- the sums k_voxel_centroid_sorted forms, without the 32-byte records' trip through memory (64 B per point).
__global__ __launch_bounds__(256) void k_voxel_centroid_fused(const PointXYZINormal* __restrict__ pts, const ScanSlot* __restrict__ slots,
                                                              const SegBlock* __restrict__ blocks, const VoxelParams* __restrict__ vp,
                                                              const int* __restrict__ n_vox, const int* __restrict__ vox_start,
                                                              const int* __restrict__ vox_info, const int* __restrict__ idx_a, const int* __restrict__ idx_b,
                                                              PointXYZINormal* __restrict__ out, int* __restrict__ out_count, int nblocks) {
    const int bi = xcd_contiguous((int)blockIdx.x, nblocks);
    if (bi < 0) return;
    const SegBlock b = blocks[bi];
    const ScanSlot sl = slots[b.scan];
    const int nv = n_vox[b.scan];
    if (b.start == 0 && blockIdx.y == 0 && threadIdx.x == 0) out_count[b.scan] = nv;
    const int r = b.start + (int)blockIdx.y * 256 + threadIdx.x;
    if (r >= nv) return;
    if (vp[b.scan].passthrough != 0) { out[sl.base + r] = pts[sl.base + r]; return; }
    const int first = vox_start[sl.base + r], last = r + 1 < nv ? vox_start[sl.base + r + 1] : vox_info[sl.base];
    const int* __restrict__ idx = (vox_info[sl.base + 1] ? idx_b : idx_a) + sl.base;
    const float4* __restrict__ q = reinterpret_cast<const float4*>(pts + sl.base);  // three float4 per point
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f, a5 = 0.f, a6 = 0.f, a7 = 0.f;
    int k = first;
    for (; k + 4 <= last; k += 4) {  // the four indices, then the twelve loads, issued together; additions strictly in order
        const int i0 = idx[k], i1 = idx[k + 1], i2 = idx[k + 2], i3 = idx[k + 3];
        const float4 x0 = q[3 * i0], n0 = q[3 * i0 + 1], c0 = q[3 * i0 + 2];
        const float4 x1 = q[3 * i1], n1 = q[3 * i1 + 1], c1 = q[3 * i1 + 2];
        const float4 x2 = q[3 * i2], n2 = q[3 * i2 + 1], c2 = q[3 * i2 + 2];
        const float4 x3 = q[3 * i3], n3 = q[3 * i3 + 1], c3 = q[3 * i3 + 2];
        a0 += x0.x; a1 += x0.y; a2 += x0.z; a3 += n0.x; a4 += n0.y; a5 += n0.z; a6 += c0.x; a7 += c0.y;
        a0 += x1.x; a1 += x1.y; a2 += x1.z; a3 += n1.x; a4 += n1.y; a5 += n1.z; a6 += c1.x; a7 += c1.y;
        a0 += x2.x; a1 += x2.y; a2 += x2.z; a3 += n2.x; a4 += n2.y; a5 += n2.z; a6 += c2.x; a7 += c2.y;
        a0 += x3.x; a1 += x3.y; a2 += x3.z; a3 += n3.x; a4 += n3.y; a5 += n3.z; a6 += c3.x; a7 += c3.y;
    }
    for (; k < last; ++k) {
        const int i0 = idx[k];
        const float4 x0 = q[3 * i0], n0 = q[3 * i0 + 1], c0 = q[3 * i0 + 2];
        a0 += x0.x; a1 += x0.y; a2 += x0.z; a3 += n0.x; a4 += n0.y; a5 += n0.z; a6 += c0.x; a7 += c0.y;
    }
    const float fn = (float)(last - first);
    PointXYZINormal o;
    o.x = a0 / fn; o.y = a1 / fn; o.z = a2 / fn; o.pad0 = 1.0f;
    float snx = a3, sny = a4, snz = a5;
    const float nn = snx * snx + sny * sny + snz * snz;
    if (nn > 0) { const float rt = sqrtf(nn); snx /= rt; sny /= rt; snz /= rt; }
    o.normal_x = snx; o.normal_y = sny; o.normal_z = snz; o.pad1 = 0;
    o.intensity = a6 / fn; o.curvature = a7 / fn; o.pad2 = 0; o.pad3 = 0;
    out[sl.base + r] = o;
}

// ---- b2: ImuProcess::UndistortPcl, backward propagation (IMU_Processing.cpp:236-276) --------------------------------
// Points are in time order.  A point with time t belongs to the interval whose head pose is the last one earlier than t;
// it is rotated / translated with the constant-rate model of that interval into the scan-end frame, all in double like the
// reference (V3D / M3D), then rounded to float.  The first point of the sorted scan is compensated once per interval from
// its own down to the first (the reference's loop re-enters it after the break at pcl_out.points.begin()).
__device__ __forceinline__ void undistort_once(const Pose6DDev& head, const Pose6DDev& tail, double dt, const LidarStateDev& e, double P[3]) {
    const double* w = tail.gyr;
    const double n = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    double E[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (n > 0.0000001) {  // so3_math.h Exp(ang_vel, dt)
        const double a[3] = {w[0] / n, w[1] / n, w[2] / n};
        const double K[9] = {0, -a[2], a[1], a[2], 0, -a[0], -a[1], a[0], 0};
        const double ang = n * dt, s = sin(ang), c1 = 1.0 - cos(ang);
        double cK[9];
        for (int k = 0; k < 9; ++k) cK[k] = c1 * K[k];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) {
                const double kk = cK[3 * r] * K[c] + cK[3 * r + 1] * K[3 + c] + cK[3 * r + 2] * K[6 + c];
                E[3 * r + c] = (E[3 * r + c] + s * K[3 * r + c]) + kk;
            }
    }
    double R_i[9], a[3], b[3], c[3];
    for (int r = 0; r < 3; ++r)
        for (int q = 0; q < 3; ++q) R_i[3 * r + q] = head.rot[3 * r] * E[q] + head.rot[3 * r + 1] * E[3 + q] + head.rot[3 * r + 2] * E[6 + q];
    for (int r = 0; r < 3; ++r) a[r] = (e.off_r[3 * r] * P[0] + e.off_r[3 * r + 1] * P[1] + e.off_r[3 * r + 2] * P[2]) + e.off_t[r];
    for (int r = 0; r < 3; ++r) {
        const double T_ei = ((head.pos[r] + head.vel[r] * dt) + ((0.5 * tail.acc[r]) * dt) * dt) - e.pos[r];
        b[r] = (R_i[3 * r] * a[0] + R_i[3 * r + 1] * a[1] + R_i[3 * r + 2] * a[2]) + T_ei;
    }
    for (int r = 0; r < 3; ++r) c[r] = (e.rot[r] * b[0] + e.rot[3 + r] * b[1] + e.rot[6 + r] * b[2]) - e.off_t[r];
    for (int r = 0; r < 3; ++r) P[r] = e.off_r[r] * c[0] + e.off_r[3 + r] * c[1] + e.off_r[6 + r] * c[2];
}

__global__ __launch_bounds__(256) void k_undistort(const PointXYZINormal* __restrict__ in, const int* __restrict__ perm, int n,
                                                   const Pose6DDev* __restrict__ poses, int n_poses,
                                                   const LidarStateDev* __restrict__ end, PointXYZINormal* __restrict__ out) {
    __shared__ Pose6DDev s_pose[kMaxImuPoses];
    __shared__ LidarStateDev s_end;
    for (int k = threadIdx.x; k < n_poses * (int)(sizeof(Pose6DDev) / 8); k += 256) ((double*)s_pose)[k] = ((const double*)poses)[k];
    for (int k = threadIdx.x; k < (int)(sizeof(LidarStateDev) / 8); k += 256) ((double*)&s_end)[k] = ((const double*)end)[k];
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    PointXYZINormal p = in[perm[i]];
    const double t = (double)p.curvature / double(1000);
    int k = 0;  // interval: head = pose k - 1, tail = pose k; 0 = not compensated
    for (int j = 1; j < n_poses; ++j) if (t > s_pose[j - 1].offset_time) k = j;
    if (k > 0) {
        double P[3] = {(double)p.x, (double)p.y, (double)p.z};
        const int k_last = i == 0 ? 1 : k;
        for (int j = k; j >= k_last; --j) {
            undistort_once(s_pose[j - 1], s_pose[j], t - s_pose[j - 1].offset_time, s_end, P);
            P[0] = (double)(float)P[0]; P[1] = (double)(float)P[1]; P[2] = (double)(float)P[2];  // stored in the float point between passes
        }
        p.x = (float)P[0]; p.y = (float)P[1]; p.z = (float)P[2];
    }
    out[i] = p;
}

void launch_undistort(const PointXYZINormal* in, const int* perm, int n, const Pose6DDev* poses, int n_poses, const LidarStateDev* end,
                      PointXYZINormal* out, hipStream_t st) {
    if (n) TC2LI_LAUNCH(k_undistort, dim3((n + 255) / 256), dim3(256), 0, st, in, perm, n, poses, n_poses, end, out);
}

// ---- b2, batched: the time sort of ImuProcess::UndistortPcl (IMU_Processing.cpp:170-172) on the device ------------------------------
// sort(pcl_out.points.begin(), pcl_out.points.end(), time_list) is std::sort: not stable, and the order it leaves equal time stamps in
// is part of the reference's result (the voxel filter sums in point order; the 64 beams of a spinning LiDAR share every time stamp).
// libstdc++'s std::sort = introsort: while a range is longer than 16 -- median of (first + 1, middle, last - 1) swapped to `first`, an
// unguarded Hoare partition of (first, last) around it, recursion into [cut, last), loop on [first, cut) -- with a depth limit of
// 2 floor(log2 n) (then heap sort), and one insertion sort over everything at the end.  Which elements end where depends only on the
// outcomes of the comparisons, so the moves can be replayed level by level, all ranges of a recursion depth at once:
//   * the k-th swap of a partition exchanges the k-th element from the left that is not smaller than the pivot with the k-th from the
//     right that is not larger, while the former lies left of the latter: two prefix counts along the array give every element its k,
//     a scatter by k pairs them, and the cut is min(first unswapped left candidate, last swapped right candidate);
//   * the closing insertion sort never moves an element past an equal one, and no element has to cross a range boundary (left of a cut
//     everything is <= the pivot <= everything right of it): it is a stable sort inside every final range of at most 16 elements.
// Two kernels.  k_time_sort: one workgroup of 1024 threads per scan replays the levels on arrays in global memory while a range is
// longer than kSortLds elements (about five levels of a 65 k scan), then lists the ranges.  k_time_sort_lds: one workgroup of 512 threads (256: 4.95 ms per 512 scans, 512: 3.5, 1024: 3.7)
// per listed range finishes it in LDS -- the same level loop on 16-bit range-local indices -- down to the stable sort of the final
// ranges, and writes the permutation.  A range whose recursion reaches the depth limit (std::sort would heap-sort it) flags its scan;
// the host sorts that scan itself.
constexpr int kSortThreads = 1024, kSortLdsThreads = 512, kSortLds = 2048;
struct TimeSortArrays { float* key; int *idx, *sf, *sl, *cl, *cr, *lp, *rp, *cut; uint8_t* flag; };

// ranges[scan slot base ..]: first index of every range the first kernel left for the second; n_ranges[scan]
//
// Round 5: the levels in global memory no longer run over per-element bookkeeping arrays.  Rounds 3-4 replayed a level
// with nine per-element arrays (range first / last, two prefix counts, two rank tables, cut, flags) passed over four times: ~110 B per point and
// level, 14.2 GB per launch of 512 scans for 0.24 GB of keys (r04_pmc_traffic_inertial.json).  Now the ranges longer than kSortLds are a short
// LIST in LDS and the workgroup takes them one after the other; a range's partition is two passes over its keys:
//   (1) per 64-element block the numbers of left candidates (not smaller than the pivot) and right candidates (not larger), block prefix sums
//       in LDS;
//   (2) every candidate finds out ALONE whether Hoare's loop swaps it: the k-th left candidate L_k is exchanged with the k-th right candidate
//       from the right R_k while L_k < R_k, i.e. a left candidate at p with A(p) left candidates up to it and Bc(p) right candidates behind it
//       takes part iff Bc(p) >= A(p) (as swap number A(p) - 1), a right candidate at q, the j-th from the right, iff at least j left candidates
//       lie before it.  Only those are written to the rank tables; the cut is min(first left candidate that takes no part, last right
//       candidate that does) -- __unguarded_partition's return value.  (tools: the rule checked against a literal Hoare loop on 20 000 random
//       ranges with ties.)
// A scan that arrives nearly in time order -- the usual case -- swaps a handful of elements per partition: the traffic is the two reads of
// the keys, 8 B per point and level.
constexpr int kTsMaxActive = 128, kTsMaxBlocks = 4096;   // ranges longer than kSortLds at one level (n / kSortLds); 64-element blocks of a range (n <= 262144)
struct TsRange { int f, l, d; };
__global__ __launch_bounds__(kSortThreads) void k_time_sort(const PointXYZINormal* __restrict__ pts, const int* __restrict__ count,
                                                            const ScanSlot* __restrict__ slots, TimeSortArrays A, int* __restrict__ perm,
                                                            int* __restrict__ fallback, int* __restrict__ ranges, int* __restrict__ n_ranges, int depth_override,
                                                            int keys_ready) {
    __shared__ TsRange s_act[2][kTsMaxActive];
    __shared__ int s_nact[2], s_nfin, s_k, s_minL, s_minR, s_fail, s_cut;
    __shared__ int s_prefL[kTsMaxBlocks + 1], s_prefR[kTsMaxBlocks + 1];
    __shared__ int s_wave[kSortThreads / 64];
    static_assert(kSortThreads == kVsThreads, "vs_block_scan is written for this workgroup size");
    const int scan = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = wave_in_block();
    const ScanSlot sl_ = slots[scan];
    const int n = count[scan], B = sl_.base;
    if (tid == 0) { fallback[scan] = 0; n_ranges[scan] = 0; }
    if (n <= 0) return;
    if (n > 64 * kTsMaxBlocks) { if (tid == 0) fallback[scan] = 1; return; }  // (the host sorts such a scan)
    float* const key = A.key + B;
    int *const idx = A.idx + B, *const lp = A.lp + B, *const rp = A.rp + B, *const last_of = A.sl + B;
    if (keys_ready) { for (int x = tid; x < n; x += kSortThreads) idx[x] = x; }  // A.key holds the time stamps already (k_pre_stream's time_out)
    else for (int x = tid; x < n; x += kSortThreads) { key[x] = pts[B + x].curvature; idx[x] = x; }
    const int depth0 = depth_override >= 0 ? depth_override : 2 * (31 - __clz(n));
    if (tid == 0) {
        s_nact[0] = s_nact[1] = 0; s_nfin = 0; s_fail = 0; s_k = 0; s_minL = s_minR = 0x7fffffff;
        if (n > kSortLds) { s_act[0][0] = TsRange{0, n, depth0}; s_nact[0] = 1; }
        else { ranges[B] = 0; last_of[0] = n; rp[0] = depth0; s_nfin = 1; }
    }
    __syncthreads();
    const unsigned long long le = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);  // lanes up to and including this one
    int cur = 0;
    while (s_nact[cur] > 0) {
        const int na = s_nact[cur];
        for (int r = 0; r < na; ++r) {
            const TsRange R = s_act[cur][r];
            const int f = R.f, l = R.l;
            if (R.d == 0) { if (tid == 0) s_fail = 1; continue; }  // std::sort would heap-sort this range: the host sorts the scan
            // ---- __move_median_to_first(first, first + 1, mid, last - 1) ----
            if (tid == 0) {
                const int a = f + 1, b = f + (l - f) / 2, c = l - 1;
                const float ka = key[a], kb = key[b], kc = key[c];
                int m;
                if (ka < kb) m = kb < kc ? b : (ka < kc ? c : a);
                else m = ka < kc ? a : (kb < kc ? c : b);
                const float kf = key[f], km = key[m];
                const int jf = idx[f], jm = idx[m];
                key[f] = km; idx[f] = jm; key[m] = kf; idx[m] = jf;
            }
            __syncthreads();
            const float pv = key[f];
            const int nb = (l - (f + 1) + 63) / 64;
            // ---- pass 1: candidates per block ----
            for (int b = wave; b < nb; b += kSortThreads / 64) {
                const int x = f + 1 + 64 * b + lane;
                bool fl = false, fr = false;
                if (x < l) { const float k = key[x]; fl = !(k < pv); fr = !(pv < k); }
                const unsigned long long bl = __ballot(fl), br = __ballot(fr);
                if (lane == 0) { s_prefL[b] = __popcll(bl); s_prefR[b] = __popcll(br); }
            }
            __syncthreads();
            // exclusive prefix sums over the blocks: a thread owns q consecutive blocks
            int totL, totR;
            {
                const int q = (nb + kSortThreads - 1) / kSortThreads, b0 = tid * q, b1 = min(b0 + q, nb);
                int sL = 0, sR = 0;
                for (int b = b0; b < b1; ++b) { sL += s_prefL[b]; sR += s_prefR[b]; }
                int eL = vs_block_scan(sL, s_wave, totL);
                int eR = vs_block_scan(sR, s_wave, totR);
                for (int b = b0; b < b1; ++b) { const int cL = s_prefL[b], cR = s_prefR[b]; s_prefL[b] = eL; s_prefR[b] = eR; eL += cL; eR += cR; }
            }
            __syncthreads();
            // ---- pass 2: who is swapped, with whom; the first left candidate that is not, the last right candidate that is ----
            for (int b = wave; b < nb; b += kSortThreads / 64) {
                const int x = f + 1 + 64 * b + lane;
                bool fl = false, fr = false;
                if (x < l) { const float k = key[x]; fl = !(k < pv); fr = !(pv < k); }
                const unsigned long long bl = __ballot(fl), br = __ballot(fr);
                const int incL = s_prefL[b] + __popcll(bl & le), incR = s_prefR[b] + __popcll(br & le);
                bool partL = false, partR = false;
                if (fl) {
                    partL = totR - incR >= incL;            // right candidates behind x >= left candidates up to x
                    if (partL) lp[f + 1 + (incL - 1)] = x;
                }
                if (fr) {
                    const int j = totR - (incR - 1);        // x is the j-th right candidate from the right
                    partR = incL - (fl ? 1 : 0) >= j;       // at least j left candidates before x
                    if (partR) rp[f + 1 + (j - 1)] = x;
                }
                const unsigned long long bpl = __ballot(partL), bnl = __ballot(fl && !partL), bpr = __ballot(partR);
                if (lane == 0) {
                    if (bpl) atomicAdd(&s_k, __popcll(bpl));
                    if (bnl) atomicMin(&s_minL, f + 1 + 64 * b + __builtin_ctzll(bnl));
                    if (bpr) atomicMin(&s_minR, f + 1 + 64 * b + __builtin_ctzll(bpr));
                }
            }
            __syncthreads();
            // ---- the swaps; the two ranges of the partition ----
            const int ks = s_k;
            for (int k = tid; k < ks; k += kSortThreads) {
                const int Lx = lp[f + 1 + k], Rx = rp[f + 1 + k];
                const float k1 = key[Lx], k2 = key[Rx];
                const int j1 = idx[Lx], j2 = idx[Rx];
                key[Lx] = k2; idx[Lx] = j2; key[Rx] = k1; idx[Rx] = j1;
            }
            if (tid == 0) s_cut = min(s_minL, ks >= 1 ? s_minR : 0x7fffffff);
            __syncthreads();
            if (tid == 0) {
                const int c = s_cut, d = R.d - 1;
                const int lo[2] = {f, c}, hi[2] = {c, l};
                for (int h = 0; h < 2; ++h) {
                    if (hi[h] <= lo[h]) continue;
                    if (hi[h] - lo[h] > kSortLds) {
                        const int at = s_nact[cur ^ 1];
                        if (at < kTsMaxActive) { s_act[cur ^ 1][at] = TsRange{lo[h], hi[h], d}; s_nact[cur ^ 1] = at + 1; } else s_fail = 1;
                    } else {
                        ranges[B + s_nfin] = lo[h]; last_of[lo[h]] = hi[h]; rp[lo[h]] = d; ++s_nfin;  // rp[first of a range]: no rank-table slot (they start at first + 1)
                    }
                }
                s_k = 0; s_minL = s_minR = 0x7fffffff;
            }
            __syncthreads();
        }
        // every wavefront has read this level's count and ranges before they are cleared (a level whose depth budget is spent passes no
        // other barrier: all its ranges share d and take the `continue` above)
        __syncthreads();
        if (tid == 0) s_nact[cur] = 0;
        cur ^= 1;
        __syncthreads();
    }
    if (tid == 0) {
        if (s_fail) fallback[scan] = 1; else n_ranges[scan] = s_nfin;
    }
}

__global__ __launch_bounds__(kSortLdsThreads) void k_time_sort_lds(const int* __restrict__ count, const ScanSlot* __restrict__ slots, TimeSortArrays A,
                                                                   int* __restrict__ perm, int* __restrict__ fallback, const int* __restrict__ ranges,
                                                                   const int* __restrict__ n_ranges) {
    __shared__ float s_key[kSortLds];
    __shared__ int s_idx[kSortLds];
    __shared__ unsigned short s_sf[kSortLds], s_sl[kSortLds], s_cl[kSortLds], s_cr[kSortLds], s_lp[kSortLds + 1], s_rp[kSortLds + 1], s_cut[kSortLds];
    __shared__ uint8_t s_flag[kSortLds];
    __shared__ int s_tot[2][kSortLdsThreads / 64], s_base[2][kSortLdsThreads / 64 + 1], s_any;
    const int scan = blockIdx.y, tid = threadIdx.x;
    if (count[scan] <= 0 || fallback[scan]) return;
    const int B = slots[scan].base, nr = n_ranges[scan];
    for (int r = blockIdx.x; r < nr; r += gridDim.x) {
        const int f = ranges[B + r], l = A.sl[B + f], m = l - f, depth = A.rp[B + f];
        __syncthreads();  // the previous range's arrays are done with
        for (int x = tid; x < m; x += kSortLdsThreads) { s_key[x] = A.key[B + f + x]; s_idx[x] = A.idx[B + f + x]; s_sf[x] = 0; s_sl[x] = (unsigned short)m; }
        const bool ok = sort_levels<kSortLdsThreads, unsigned short, float>(s_key, s_idx, s_sf, s_sl, s_cl, s_cr, s_lp, s_rp, s_cut, s_flag, nullptr, m, depth, 16,
                                                                     s_tot, s_base, &s_any);
        if (!ok) { if (tid == 0) fallback[scan] = 1; continue; }  // uniform; the host sorts the scan
        __syncthreads();
        sort_final<kSortLdsThreads, unsigned short, float>(s_key, s_idx, s_sf, s_sl, m, perm + B + f);
    }
}

// points[i] = in[perm[i]] compensated into the scan-end frame, every scan of a batch (ImuProcess::UndistortPcl, point part)
__global__ __launch_bounds__(256) void k_undistort_batch(const PointXYZINormal* __restrict__ in, const int* __restrict__ perm, const int* __restrict__ count,
                                                         const ScanSlot* __restrict__ slots, const SegBlock* __restrict__ blocks,
                                                         const Pose6DDev* __restrict__ poses, const int* __restrict__ n_poses,
                                                         const LidarStateDev* __restrict__ ends, PointXYZINormal* __restrict__ out) {
    __shared__ Pose6DDev s_pose[kMaxImuPoses];
    __shared__ LidarStateDev s_end;
    const SegBlock b = blocks[blockIdx.x];
    const ScanSlot sl = slots[b.scan];
    const int n = count[b.scan], np_ = n_poses[b.scan];
    const int i = b.start + blockIdx.y * 256 + threadIdx.x;
    if (b.start + (int)blockIdx.y * 256 >= n) return;
    for (int k = threadIdx.x; k < np_ * (int)(sizeof(Pose6DDev) / 8); k += 256) ((double*)s_pose)[k] = ((const double*)(poses + (size_t)b.scan * kMaxImuPoses))[k];
    for (int k = threadIdx.x; k < (int)(sizeof(LidarStateDev) / 8); k += 256) ((double*)&s_end)[k] = ((const double*)(ends + b.scan))[k];
    __syncthreads();
    if (i >= n) return;
    PointXYZINormal p = in[sl.base + perm[sl.base + i]];
    const double t = (double)p.curvature / double(1000);
    int k = 0;  // interval: head = pose k - 1, tail = pose k; 0 = not compensated
    for (int j = 1; j < np_; ++j) if (t > s_pose[j - 1].offset_time) k = j;
    if (k > 0) {
        double P[3] = {(double)p.x, (double)p.y, (double)p.z};
        const int k_last = i == 0 ? 1 : k;
        for (int j = k; j >= k_last; --j) {
            undistort_once(s_pose[j - 1], s_pose[j], t - s_pose[j - 1].offset_time, s_end, P);
            P[0] = (double)(float)P[0]; P[1] = (double)(float)P[1]; P[2] = (double)(float)P[2];  // stored in the float point between passes
        }
        p.x = (float)P[0]; p.y = (float)P[1]; p.z = (float)P[2];
    }
    out[sl.base + i] = p;
}

void launch_time_sort(const PointXYZINormal* pts, const int* count, const ScanSlot* slots, int n_scans, float* key, int* ints5, int* ints3, uint8_t* flag,
                      size_t total, int* perm, int* fallback, int* ranges, int* n_ranges, int depth_override, bool keys_ready, hipStream_t st) {
    if (!n_scans) return;
    TimeSortArrays A{key, ints5, ints5 + total, ints5 + 2 * total, ints5 + 3 * total, ints5 + 4 * total, ints3, ints3 + total, ints3 + 2 * total, flag};
    TC2LI_LAUNCH(k_time_sort, dim3(n_scans), dim3(kSortThreads), 0, st, pts, count, slots, A, perm, fallback, ranges, n_ranges, depth_override, keys_ready ? 1 : 0);
    // ranges per scan: a 65 k scan leaves some tens to a few hundred; the workgroups of a scan take them in turn
    // (a batch: 8 workgroups per scan, each a dozen ranges one after the other -- a workgroup holds 47 KB of LDS, and beside the other stages'
    // kernels a launch pays for every workgroup it has placed: 64 per scan read 12-13 ms per 512 scans in the loop against 3.2 alone, 16: 7.3-7.8, 8: 5.8; the LiDAR-inertial stage alone 22.1 / 21.4 / 20.9 ms)
    static const int kGroups = getenv("TC2LI_TIME_SORT_GROUPS") ? std::max(1, std::min(atoi(getenv("TC2LI_TIME_SORT_GROUPS")), 256)) : 8;
    TC2LI_LAUNCH(k_time_sort_lds, dim3(n_scans >= 64 ? kGroups : 256, n_scans), dim3(kSortLdsThreads), 0, st, count, slots, A, perm, fallback, ranges, n_ranges);
}
void launch_undistort_batch(const PointXYZINormal* in, const int* perm, const int* count, const ScanSlot* slots, const SegBlock* blocks, int nblocks,
                            const Pose6DDev* poses, const int* n_poses, const LidarStateDev* ends, PointXYZINormal* out, hipStream_t st) {
    if (nblocks) TC2LI_LAUNCH(k_undistort_batch, dim3(nblocks, kSegBlock / 256), dim3(256), 0, st, in, perm, count, slots, blocks, poses, n_poses, ends, out);
}

// ---- b5: map spatial index (replaces the ikd-Tree as a dense uniform grid over the map's bounding box) ------------
// Cells are ordered x-fastest, so the points of a run of cells along x are one contiguous range of the sorted array.
// ---- b4 + b5 + b6: pointBodyToWorld, 5-NN, EstiPlane and the gates of feature_extraction --------------------------
// Least squares of the 5x3 system by Householder QR with column pivoting (what colPivHouseholderQr().solve() does),
// same operation order as the CPU statement.
__device__ void qr_solve_5x3(float (&A)[5][3], float (&b)[5], float (&x)[3]) {
    constexpr int R = 5, C = 3;
    constexpr float kEps = 1.1920928955078125e-07f, kMin = 1.17549435082228750797e-38f;
    float normU[3], normD[3];
#pragma unroll
    for (int j = 0; j < C; ++j) {
        float s = 0;
#pragma unroll
        for (int i = 0; i < R; ++i) s += A[i][j] * A[i][j];
        normU[j] = normD[j] = sqrtf(s);
    }
    const float maxn = fmaxf(normU[0], fmaxf(normU[1], normU[2]));
    const float thr_helper = __fdiv_rn((maxn * kEps) * (maxn * kEps), (float)R);
    const float downdate_thr = sqrtf(kEps);
    int perm[3] = {0, 1, 2};
    int nonzero = C;
    float tau[3] = {0, 0, 0};
#pragma unroll
    for (int k = 0; k < C; ++k) {
        int big = k;
#pragma unroll
        for (int j = k + 1; j < C; ++j) if (normU[j] > normU[big]) big = j;
        const float big_sq = normU[big] * normU[big];
        if (nonzero == C && big_sq < thr_helper * (float)(R - k)) nonzero = k;
        if (big != k) {
#pragma unroll
            for (int j = k + 1; j < C; ++j)
                if (j == big) {
#pragma unroll
                    for (int i = 0; i < R; ++i) { const float t = A[i][k]; A[i][k] = A[i][j]; A[i][j] = t; }
                    float t = normU[k]; normU[k] = normU[j]; normU[j] = t;
                    t = normD[k]; normD[k] = normD[j]; normD[j] = t;
                    const int ti = perm[k]; perm[k] = perm[j]; perm[j] = ti;
                }
        }
        float tail = 0;
#pragma unroll
        for (int i = k + 1; i < R; ++i) tail += A[i][k] * A[i][k];
        const float c0 = A[k][k];
        float beta;
        if (tail <= kMin) {
            tau[k] = 0;
            beta = c0;
#pragma unroll
            for (int i = k + 1; i < R; ++i) A[i][k] = 0;
        } else {
            beta = sqrtf(c0 * c0 + tail);
            if (c0 >= 0) beta = -beta;
#pragma unroll
            for (int i = k + 1; i < R; ++i) A[i][k] = __fdiv_rn(A[i][k], c0 - beta);
            tau[k] = __fdiv_rn(beta - c0, beta);
        }
        A[k][k] = beta;
        if (tau[k] != 0) {
#pragma unroll
            for (int j = k + 1; j < C; ++j) {
                float t = 0;
#pragma unroll
                for (int i = k + 1; i < R; ++i) t += A[i][k] * A[i][j];
                t += A[k][j];
                A[k][j] -= tau[k] * t;
#pragma unroll
                for (int i = k + 1; i < R; ++i) A[i][j] -= tau[k] * A[i][k] * t;
            }
        }
#pragma unroll
        for (int j = k + 1; j < C; ++j) {
            if (normU[j] != 0) {
                float temp = __fdiv_rn(fabsf(A[k][j]), normU[j]);
                temp = (1.0f + temp) * (1.0f - temp);
                temp = temp < 0 ? 0 : temp;
                const float r = __fdiv_rn(normU[j], normD[j]);
                const float temp2 = temp * (r * r);
                if (temp2 <= downdate_thr) {
                    float s = 0;
#pragma unroll
                    for (int i = k + 1; i < R; ++i) s += A[i][j] * A[i][j];
                    normD[j] = sqrtf(s);
                    normU[j] = normD[j];
                } else {
                    normU[j] *= sqrtf(temp);
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < C; ++k) {
        if (k < nonzero && tau[k] != 0) {
            float t = 0;
#pragma unroll
            for (int i = k + 1; i < R; ++i) t += A[i][k] * b[i];
            t += b[k];
            b[k] -= tau[k] * t;
#pragma unroll
            for (int i = k + 1; i < R; ++i) b[i] -= tau[k] * A[i][k] * t;
        }
    }
    float c[3] = {0, 0, 0};
#pragma unroll
    for (int i = C - 1; i >= 0; --i) {
        if (i < nonzero) {
            float s = b[i];
#pragma unroll
            for (int j = i + 1; j < C; ++j) if (j < nonzero) s -= A[i][j] * c[j];
            c[i] = __fdiv_rn(s, A[i][i]);
        }
    }
    x[0] = x[1] = x[2] = 0;
#pragma unroll
    for (int i = 0; i < C; ++i)
        if (i < nonzero) {
#pragma unroll
            for (int j = 0; j < C; ++j) if (perm[i] == j) x[j] = c[i];
        }
}

// Candidate with its position in the cell-sorted array: among candidates that compare equal under PointType_CMP the
// sequential search keeps the one met first, and every scan below visits positions in ascending order.
struct CandK { float d, x; int idx, k; };
constexpr int kNoCand = 0x7fffffff;
__device__ __forceinline__ bool candk_less(const CandK& a, const CandK& b) {
    if (b.k == kNoCand) return a.k != kNoCand;
    if (a.k == kNoCand) return false;
    if ((double)fabsf(a.d - b.d) < 1e-10) { if (a.x != b.x) return a.x < b.x; return a.k < b.k; }
    return a.d < b.d;
}
__device__ __forceinline__ CandK candk_empty() { CandK c; c.d = 0.f; c.x = 0.f; c.idx = -1; c.k = kNoCand; return c; }
__device__ __forceinline__ CandK candk_sel(bool f, const CandK& a, const CandK& b) {
    CandK o;
    o.d = f ? a.d : b.d; o.x = f ? a.x : b.x; o.idx = f ? a.idx : b.idx; o.k = f ? a.k : b.k;
    return o;
}
// The five best candidates, ascending; free slots hold the empty candidate (which compares greater than any other).
// Named members and select-based updates: the list must live in registers (an indexed array ends up in scratch memory
// and made this search memory-latency bound on its own bookkeeping).
struct Top5 {
    CandK b0, b1, b2, b3, b4;
    __device__ __forceinline__ void clear() { b0 = b1 = b2 = b3 = b4 = candk_empty(); }
    __device__ __forceinline__ int count() const {
        return (b0.k != kNoCand) + (b1.k != kNoCand) + (b2.k != kNoCand) + (b3.k != kNoCand) + (b4.k != kNoCand);
    }
    __device__ __forceinline__ void insert(const CandK& c) {
        if (!candk_less(c, b4)) return;
        const bool l3 = candk_less(c, b3), l2 = candk_less(c, b2), l1 = candk_less(c, b1), l0 = candk_less(c, b0);
        b4 = candk_sel(l3, b3, c);
        b3 = candk_sel(l3, candk_sel(l2, b2, c), b3);
        b2 = candk_sel(l2, candk_sel(l1, b1, c), b2);
        b1 = candk_sel(l1, candk_sel(l0, b0, c), b1);
        b0 = candk_sel(l0, c, b0);
    }
    __device__ __forceinline__ void pop() { b0 = b1; b1 = b2; b2 = b3; b3 = b4; b4 = candk_empty(); }
};
__device__ __forceinline__ CandK candk_shfl_xor(const CandK& c, int m) {
    CandK o;
    o.d = __shfl_xor(c.d, m, 64); o.x = __shfl_xor(c.x, m, 64); o.idx = __shfl_xor(c.idx, m, 64); o.k = __shfl_xor(c.k, m, 64);
    return o;
}
__device__ __forceinline__ CandK candk_shfl(const CandK& c, int src) {
    CandK o;
    o.d = __shfl(c.d, src, 64); o.x = __shfl(c.x, src, 64); o.idx = __shfl(c.idx, src, 64); o.k = __shfl(c.k, src, 64);
    return o;
}
// Smallest head among the GROUP lanes that share a query (GROUP = 4: quad, 64: wavefront); the lane that owns it pops.
template <int GROUP>
__device__ __forceinline__ CandK merge_step(Top5& t) {
    const int lane = threadIdx.x & 63, leader = lane & ~(GROUP - 1);
    const CandK head = t.b0;
    CandK w = head;
#pragma unroll
    for (int m = 1; m < GROUP; m <<= 1) {
        const CandK o = candk_shfl_xor(w, m);
        w = candk_sel(candk_less(o, w), o, w);
    }
    w = candk_shfl(w, leader);  // one opinion per group even where the tolerance compare is not transitive
    if (head.k != kNoCand && head.k == w.k) t.pop();
    return w;
}
// Merges the sorted lists of the group; afterwards every lane of the group holds the same five best candidates.
template <int GROUP>
__device__ __forceinline__ void merge_top5(Top5& t) {
    Top5 o;
    o.b0 = merge_step<GROUP>(t); o.b1 = merge_step<GROUP>(t); o.b2 = merge_step<GROUP>(t); o.b3 = merge_step<GROUP>(t);
    o.b4 = merge_step<GROUP>(t);
    t = o;
}

__device__ __forceinline__ void scan_run(const float4* __restrict__ pts, int s, int e, float qx, float qy, float qz, Top5& t) {
    for (int k = s; k < e; k += 2) {
        const float4 m0 = pts[k];
        const float4 m1 = pts[min(k + 1, e - 1)];  // issued together with m0: half the dependent-load stalls
        CandK c;
        c.d = (qx - m0.x) * (qx - m0.x) + (qy - m0.y) * (qy - m0.y) + (qz - m0.z) * (qz - m0.z);
        c.x = m0.x; c.idx = __float_as_int(m0.w); c.k = k;
        if (c.idx >= 0) t.insert(c);  // < 0: a tombstone (deleted point, lidar_device.hpp MapGrid)
        if (k + 1 < e) {
            c.d = (qx - m1.x) * (qx - m1.x) + (qy - m1.y) * (qy - m1.y) + (qz - m1.z) * (qz - m1.z);
            c.x = m1.x; c.idx = __float_as_int(m1.w); c.k = k + 1;
            if (c.idx >= 0) t.insert(c);
        }
    }
}

// The cube of cells [c - RING, c + RING]^3 as (2 RING + 1)^2 runs along x, the runs dealt round-robin to the GROUP lanes
// of the query; the bucket bounds of all of a lane's runs are fetched before the first point is read.
template <int RING, int GROUP>
__device__ __forceinline__ void scan_cube(const MapGrid& g, int cx, int cy, int cz, float qx, float qy, float qz, int l, Top5& t) {
    constexpr int W = 2 * RING + 1, NR = W * W, PER = (NR + GROUP - 1) / GROUP;
    static_assert(W <= kMapSegCells, "a run of the cube touches at most two segments");
    t.clear();
    const int xa = max(cx - RING - g.x0, 0), xb = min(cx + RING - g.x0, g.nx - 1);
    // the run's cells lie in one segment of the row, or in two: [xa, end of xa's segment] and [start of xb's segment, xb]
    const bool two = (xa >> 4) != (xb >> 4);
    const int xa_end = two ? (xa | (kMapSegCells - 1)) : xb, xb_begin = xb & ~(kMapSegCells - 1);
    int rs[PER], re[PER], rs2[PER], re2[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int r = l + GROUP * j;
        const int iz = cz - RING + r / W - g.z0, iy = cy - RING + r % W - g.y0;
        const bool ok = r < NR && xa <= xb && iz >= 0 && iz < g.nz && iy >= 0 && iy < g.ny;
        const int row = iz * g.ny + iy;
        rs[j] = ok ? g.bucket_start[map_start_index(g, row, xa)] : 0;
        re[j] = ok ? g.bucket_start[map_start_index(g, row, xa_end) + 1] : 0;
        rs2[j] = ok && two ? g.bucket_start[map_start_index(g, row, xb_begin)] : 0;
        re2[j] = ok && two ? g.bucket_start[map_start_index(g, row, xb) + 1] : 0;
    }
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        scan_run(g.pts, rs[j], re[j], qx, qy, qz, t);
        scan_run(g.pts, rs2[j], re2[j], qx, qy, qz, t);
    }
    merge_top5<GROUP>(t);
}

__device__ __forceinline__ bool knn_complete(const MapGrid& g, int ring, int cx, int cy, int cz, float qx, float qy, float qz, const Top5& t) {
    if (t.b4.k == kNoCand) return false;
    // every point outside the cube is at least `edge` away from the query
    const float lo = (float)ring * g.cell, hi = (float)(ring + 1) * g.cell;
    const float fx = qx - (float)cx * g.cell, fy = qy - (float)cy * g.cell, fz = qz - (float)cz * g.cell;
    float edge = fminf(fminf(fx + lo, hi - fx), fminf(fminf(fy + lo, hi - fy), fminf(fz + lo, hi - fz)));
    edge = fmaxf(edge, 0.f);
    return t.b4.d < edge * edge * 0.999f;
}

// esti_plane + the residual gate of feature_extraction / h_share_model (LidarFrontEnd.cpp:453-482, 1040-1055, 529-545) for one
// query and its five neighbours: returns the selection flag and fills nv (normal, intensity = pd2)
__device__ __forceinline__ uint8_t plane_gate(const PointXYZINormal* __restrict__ map_pts, const int (&ids)[5], const PointXYZINormal& pw,
                                              double bx, double by, double bz, PointXYZINormal& nv) {
    nv.x = 0; nv.y = 0; nv.z = 0; nv.pad0 = 1.0f; nv.normal_x = 0; nv.normal_y = 0; nv.normal_z = 0; nv.pad1 = 0;
    nv.intensity = 0; nv.curvature = 0; nv.pad2 = 0; nv.pad3 = 0;
    float A[5][3], rhs[5], sol[3], px[5], py[5], pz[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const PointXYZINormal m = map_pts[ids[j]];
        px[j] = m.x; py[j] = m.y; pz[j] = m.z;
        A[j][0] = m.x; A[j][1] = m.y; A[j][2] = m.z;
        rhs[j] = -1.0f;
    }
    qr_solve_5x3(A, rhs, sol);
    const float nrm = sqrtf(sol[0] * sol[0] + sol[1] * sol[1] + sol[2] * sol[2]);
    const float pa = __fdiv_rn(sol[0], nrm), pbn = __fdiv_rn(sol[1], nrm), pc = __fdiv_rn(sol[2], nrm);
    const float pd = (float)(1.0 / (double)nrm);
    bool plane = true;
#pragma unroll
    for (int j = 0; j < 5; ++j)
        if (fabsf(pa * px[j] + pbn * py[j] + pc * pz[j] + pd) > 0.1f) plane = false;
    if (!plane) return 0;
    const float pd2 = pa * pw.x + pbn * pw.y + pc * pw.z + pd;
    const double pnorm = sqrt(bx * bx + by * by + bz * bz);
    const float s = (float)(1 - 0.9 * (double)fabsf(pd2) / sqrt(pnorm));
    if (!((double)s > 0.9)) return 0;
    nv.x = pa; nv.y = pbn; nv.z = pc; nv.intensity = pd2;
    return 1;
}

// feature_extraction gates (LidarFrontEnd.cpp:1032-1055) and the result records of one query
__device__ __forceinline__ void knn_finish(const MapGrid& grid, const Top5& t, const PointXYZINormal& pw, double bx, double by,
                                           double bz, bool write, int o, uint8_t* __restrict__ selected, PointXYZINormal* __restrict__ normvec,
                                           int* __restrict__ nearest_idx, float* __restrict__ nearest_d, int* __restrict__ nfound) {
    const int nb = t.count();
    uint8_t sel = 0;
    PointXYZINormal nv;
    nv.x = 0; nv.y = 0; nv.z = 0; nv.pad0 = 1.0f; nv.normal_x = 0; nv.normal_y = 0; nv.normal_z = 0; nv.pad1 = 0;
    nv.intensity = 0; nv.curvature = 0; nv.pad2 = 0; nv.pad3 = 0;
    if (nb == 5 && !(t.b4.d > 5.f)) {
        const int ids[5] = {t.b0.idx, t.b1.idx, t.b2.idx, t.b3.idx, t.b4.idx};
        sel = plane_gate(grid.points, ids, pw, bx, by, bz, nv);
    }
    if (!write) return;
    nfound[o] = nb;
    int* ni = nearest_idx + (size_t)o * 5;
    float* nd = nearest_d + (size_t)o * 5;
    ni[0] = t.b0.idx; ni[1] = t.b1.idx; ni[2] = t.b2.idx; ni[3] = t.b3.idx; ni[4] = t.b4.idx;  // idx of an empty slot is -1, its d is 0
    nd[0] = t.b0.d; nd[1] = t.b1.d; nd[2] = t.b2.d; nd[3] = t.b3.d; nd[4] = t.b4.d;
    selected[o] = sel;
    normvec[o] = nv;
}

// Pass 1: four lanes per query search the 3^3 and, if needed, the 5^3 cube of map cells (the exact 5 nearest map points
// of almost every query of a scan that overlaps the map lie there); queries that need a wider search are queued.
__global__ __launch_bounds__(256) void k_knn_plane(const MapGrid* __restrict__ grids,
                                                   const PointXYZINormal* __restrict__ body, const int* __restrict__ count,
                                                   const ScanSlot* __restrict__ slots, const SegBlock* __restrict__ blocks,
                                                   const LidarStateDev* __restrict__ states, PointXYZINormal* __restrict__ world,
                                                   uint8_t* __restrict__ selected, PointXYZINormal* __restrict__ normvec,
                                                   int* __restrict__ nearest_idx, float* __restrict__ nearest_d,
                                                   int* __restrict__ nfound, int* __restrict__ hard_count, int2* __restrict__ hard_list, int nblocks) {
    const int bi = xcd_contiguous((int)blockIdx.x, nblocks);
    if (bi < 0) return;
    const SegBlock b = blocks[bi];
    const ScanSlot sl = slots[b.scan];
    const int n = count[b.scan];
    const int i = b.start + blockIdx.y * 64 + (threadIdx.x >> 2), l = threadIdx.x & 3;
    if (i >= n) return;  // whole quads leave together
    const MapGrid grid = global_record(grids[b.scan]);
    const PointXYZINormal pb = body[sl.base + i];
    const PointXYZINormal pw = body_to_world(pb, states[b.scan]);
    if (l == 0) world[sl.base + i] = pw;
    Top5 t;
    t.clear();
    const int cx = (int)floorf(pw.x * grid.inv_cell), cy = (int)floorf(pw.y * grid.inv_cell), cz = (int)floorf(pw.z * grid.inv_cell);
    bool done = grid.n_points == 0;
    if (!done) {
        scan_cube<1, 4>(grid, cx, cy, cz, pw.x, pw.y, pw.z, l, t);
        done = knn_complete(grid, 1, cx, cy, cz, pw.x, pw.y, pw.z, t);
    }
    if (!done) {
        scan_cube<2, 4>(grid, cx, cy, cz, pw.x, pw.y, pw.z, l, t);
        done = knn_complete(grid, 2, cx, cy, cz, pw.x, pw.y, pw.z, t);
    }
    if (!done) {
        if (l == 0) hard_list[atomicAdd(hard_count, 1)] = make_int2(b.scan, i);
        return;
    }
    knn_finish(grid, t, pw, (double)pb.x, (double)pb.y, (double)pb.z, l == 0, sl.base + i, selected, normvec, nearest_idx, nearest_d, nfound);
}

// Pass 2: one wavefront per queued query; wider cubes (runs dealt to the 64 lanes), finally the whole map.
template <int RING>
__device__ __forceinline__ bool hard_ring(const MapGrid& g, int cx, int cy, int cz, float qx, float qy, float qz, int lane, Top5& t) {
    constexpr int W = 2 * RING + 1;
    t.clear();
    const int xa = max(cx - RING - g.x0, 0), xb = min(cx + RING - g.x0, g.nx - 1);
    if (xa <= xb)
        for (int r = lane; r < W * W; r += 64) {
            const int iz = cz - RING + r / W - g.z0, iy = cy - RING + r % W - g.y0;
            if (iz < 0 || iz >= g.nz || iy < 0 || iy >= g.ny) continue;
            const int row = iz * g.ny + iy;
            for (int x0 = xa; x0 <= xb; x0 = (x0 | (kMapSegCells - 1)) + 1) {  // segment by segment
                const int x1 = min(x0 | (kMapSegCells - 1), xb);
                scan_run(g.pts, g.bucket_start[map_start_index(g, row, x0)], g.bucket_start[map_start_index(g, row, x1) + 1], qx, qy, qz, t);
            }
        }
    merge_top5<64>(t);
    return knn_complete(g, RING, cx, cy, cz, qx, qy, qz, t);
}

__global__ __launch_bounds__(256) void k_knn_hard(const MapGrid* __restrict__ grids, const PointXYZINormal* __restrict__ body,
                                                  const ScanSlot* __restrict__ slots, const LidarStateDev* __restrict__ states,
                                                  uint8_t* __restrict__ selected, PointXYZINormal* __restrict__ normvec,
                                                  int* __restrict__ nearest_idx, float* __restrict__ nearest_d, int* __restrict__ nfound,
                                                  const int* __restrict__ hard_count, const int2* __restrict__ hard_list) {
    const int lane = threadIdx.x & 63, wave = (int)blockIdx.x * 4 + wave_in_block(), n_waves = gridDim.x * 4;
    const int total = *hard_count;
    for (int h = wave; h < total; h += n_waves) {
        const int2 q = hard_list[h];
        const ScanSlot sl = slots[q.x];
        const MapGrid grid = global_record(grids[q.x]);
        const PointXYZINormal pb = body[sl.base + q.y];
        const PointXYZINormal pw = body_to_world(pb, states[q.x]);
        const int cx = (int)floorf(pw.x * grid.inv_cell), cy = (int)floorf(pw.y * grid.inv_cell), cz = (int)floorf(pw.z * grid.inv_cell);
        Top5 t;
        t.clear();
        // a cube is worth visiting only while it has fewer runs than a third of the map has points
        const int np3 = grid.n_points / 3;
        bool done = false;
        if (!done && 9 * 9 < np3) done = hard_ring<4>(grid, cx, cy, cz, pw.x, pw.y, pw.z, lane, t);
        if (!done && 17 * 17 < np3) done = hard_ring<8>(grid, cx, cy, cz, pw.x, pw.y, pw.z, lane, t);
        if (!done && 33 * 33 < np3) done = hard_ring<16>(grid, cx, cy, cz, pw.x, pw.y, pw.z, lane, t);
        if (!done && 65 * 65 < np3) done = hard_ring<32>(grid, cx, cy, cz, pw.x, pw.y, pw.z, lane, t);
        if (!done) {  // isolated query: the whole map keeps the result exact
            t.clear();
            const int per = (grid.n_slots + 63) / 64;  // every entry, the rows' unused room included (tombstones)
            scan_run(grid.pts, min(lane * per, grid.n_slots), min((lane + 1) * per, grid.n_slots), pw.x, pw.y, pw.z, t);
            merge_top5<64>(t);
        }
        knn_finish(grid, t, pw, (double)pb.x, (double)pb.y, (double)pb.z, lane == 0, sl.base + q.y, selected, normvec, nearest_idx, nearest_d,
                   nfound);
    }
}

// ---- iterated ESKF (esekf::update_iterated_dyn_share_modified with h_share_model, row b7) -----------------------------------
// An iteration that does not follow a converged one keeps the neighbours and the selection of the previous iteration and only
// re-evaluates the plane residual at the new state (LidarFrontEnd.cpp:519-545).
__global__ __launch_bounds__(256) void k_eskf_refit(const MapGrid grid, const PointXYZINormal* __restrict__ body, int n, const LidarStateDev* __restrict__ state,
                                                    const int* __restrict__ nearest_idx, PointXYZINormal* __restrict__ world,
                                                    uint8_t* __restrict__ selected, PointXYZINormal* __restrict__ normvec) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const PointXYZINormal pb = body[i];
    const PointXYZINormal pw = body_to_world(pb, *state);
    world[i] = pw;
    if (!selected[i]) return;
    const int* ni = nearest_idx + (size_t)i * 5;
    const int ids[5] = {ni[0], ni[1], ni[2], ni[3], ni[4]};
    PointXYZINormal nv;
    const uint8_t sel = plane_gate(grid.points, ids, pw, (double)pb.x, (double)pb.y, (double)pb.z, nv);
    selected[i] = sel;
    if (sel) normvec[i] = nv;
}

// Rows of the measurement Jacobian (LidarFrontEnd.cpp:566-600) and their normal equations: per workgroup the partial sums of
// H^T H (144), H^T h (12), sum |pd2| and the number of rows; summed over the workgroups in index order by k_eskf_reduce.
constexpr int kEskfOut = 144 + 12 + 2;
__device__ __forceinline__ void eskf_normal_rows(const PointXYZINormal* __restrict__ body, int n, const LidarStateDev* __restrict__ state,
                                                 const uint8_t* __restrict__ selected, const PointXYZINormal* __restrict__ normvec, int extrinsic_est_en,
                                                 int i, double (*s_row)[13], float* s_abs, int* s_cnt, double* __restrict__ out) {
    const int tid = threadIdx.x;
    double row[13];
#pragma unroll
    for (int k = 0; k < 13; ++k) row[k] = 0.0;
    float ares = 0.f;
    int cnt = 0;
    if (i < n && selected[i]) {
        const LidarStateDev st = *state;
        const PointXYZINormal pb = body[i];
        const PointXYZINormal nv = normvec[i];
        const double pbe[3] = {(double)pb.x, (double)pb.y, (double)pb.z}, nrm[3] = {(double)nv.x, (double)nv.y, (double)nv.z};
        double pt[3], C[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) pt[r] = (st.off_r[3 * r] * pbe[0] + st.off_r[3 * r + 1] * pbe[1] + st.off_r[3 * r + 2] * pbe[2]) + st.off_t[r];
#pragma unroll
        for (int r = 0; r < 3; ++r) C[r] = st.rot[r] * nrm[0] + st.rot[3 + r] * nrm[1] + st.rot[6 + r] * nrm[2];  // rot^T n
        row[0] = nrm[0]; row[1] = nrm[1]; row[2] = nrm[2];
        row[3] = pt[1] * C[2] - pt[2] * C[1];  // hat(point_this) * C
        row[4] = pt[2] * C[0] - pt[0] * C[2];
        row[5] = pt[0] * C[1] - pt[1] * C[0];
        if (extrinsic_est_en) {
            double D[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) D[r] = st.off_r[r] * C[0] + st.off_r[3 + r] * C[1] + st.off_r[6 + r] * C[2];  // offset_R^T C
            row[6] = pbe[1] * D[2] - pbe[2] * D[1];
            row[7] = pbe[2] * D[0] - pbe[0] * D[2];
            row[8] = pbe[0] * D[1] - pbe[1] * D[0];
            row[9] = C[0]; row[10] = C[1]; row[11] = C[2];
        }
        row[12] = -(double)nv.intensity;
        ares = fabsf(nv.intensity);
        cnt = 1;
    }
#pragma unroll
    for (int k = 0; k < 13; ++k) s_row[tid][k] = row[k];
    s_abs[tid] = ares;
    s_cnt[tid] = cnt;
    __syncthreads();
    if (tid < kEskfOut) {
        double acc = 0.0;
        if (tid < 144) {
            const int r = tid / 12, c = tid % 12;
            for (int k = 0; k < 256; ++k) acc += s_row[k][r] * s_row[k][c];
        } else if (tid < 156) {
            const int r = tid - 144;
            for (int k = 0; k < 256; ++k) acc += s_row[k][r] * s_row[k][12];
        } else if (tid == 156) {
            for (int k = 0; k < 256; ++k) acc += (double)s_abs[k];
        } else {
            for (int k = 0; k < 256; ++k) acc += (double)s_cnt[k];
        }
        out[tid] = acc;
    }
}
__global__ __launch_bounds__(256) void k_eskf_normal(const PointXYZINormal* __restrict__ body, int n, const LidarStateDev* __restrict__ state,
                                                     const uint8_t* __restrict__ selected, const PointXYZINormal* __restrict__ normvec,
                                                     int extrinsic_est_en, double* __restrict__ partial) {
    __shared__ double s_row[256][13];
    __shared__ float s_abs[256];
    __shared__ int s_cnt[256];
    eskf_normal_rows(body, n, state, selected, normvec, extrinsic_est_en, blockIdx.x * 256 + threadIdx.x, s_row, s_abs, s_cnt,
                     partial + (size_t)blockIdx.x * kEskfOut);
}
__global__ __launch_bounds__(256) void k_eskf_reduce(const double* __restrict__ partial, int nblocks, double* __restrict__ out) {
    const int tid = threadIdx.x;
    if (tid >= kEskfOut) return;
    double acc = 0.0;
    for (int b = 0; b < nblocks; ++b) acc += partial[(size_t)b * kEskfOut + tid];
    out[tid] = acc;
}

// ---- the same two kernels for the scans of a batch (tc2li_lidar_inertial_frontend_batch): a block list names the (scan, 1024-point
// block) pairs taking part; partial sums of the normal equations at row (slot base + point) / 256, so that a scan's rows are summed in
// the order of the one-scan kernels above (same bits) ----
__global__ __launch_bounds__(256) void k_eskf_refit_b(const MapGrid* __restrict__ grids, const PointXYZINormal* __restrict__ body, const int* __restrict__ count,
                                                      const ScanSlot* __restrict__ slots, const SegBlock* __restrict__ blocks,
                                                      const LidarStateDev* __restrict__ states, const int* __restrict__ nearest_idx,
                                                      PointXYZINormal* __restrict__ world, uint8_t* __restrict__ selected, PointXYZINormal* __restrict__ normvec) {
    const SegBlock b = blocks[blockIdx.x];
    const int i = b.start + blockIdx.y * 256 + threadIdx.x;
    if (i >= count[b.scan]) return;
    const size_t o = (size_t)slots[b.scan].base + i;
    const PointXYZINormal pb = body[o];
    const PointXYZINormal pw = body_to_world(pb, states[b.scan]);
    world[o] = pw;
    if (!selected[o]) return;
    const int* ni = nearest_idx + o * 5;
    const int ids[5] = {ni[0], ni[1], ni[2], ni[3], ni[4]};
    PointXYZINormal nv;
    const uint8_t sel = plane_gate(grids[b.scan].points, ids, pw, (double)pb.x, (double)pb.y, (double)pb.z, nv);
    selected[o] = sel;
    if (sel) normvec[o] = nv;
}
__global__ __launch_bounds__(256) void k_eskf_normal_b(const PointXYZINormal* __restrict__ body, const int* __restrict__ count, const ScanSlot* __restrict__ slots,
                                                       const SegBlock* __restrict__ blocks, const LidarStateDev* __restrict__ states,
                                                       const uint8_t* __restrict__ selected, const PointXYZINormal* __restrict__ normvec,
                                                       int extrinsic_est_en, double* __restrict__ partial) {
    __shared__ double s_row[256][13];
    __shared__ float s_abs[256];
    __shared__ int s_cnt[256];
    const SegBlock b = blocks[blockIdx.x];
    const int n = count[b.scan], i0 = b.start + blockIdx.y * 256;
    if (i0 >= n) return;
    const int base = slots[b.scan].base;
    eskf_normal_rows(body + base, n, states + b.scan, selected + base, normvec + base, extrinsic_est_en, i0 + threadIdx.x, s_row, s_abs, s_cnt,
                     partial + (size_t)((base + i0) / 256) * kEskfOut);
}
// one workgroup per listed scan: the scan's rows of `partial` in index order -> out[scan][kEskfOut] (pinned host memory)
__global__ __launch_bounds__(256) void k_eskf_reduce_b(const double* __restrict__ partial, const int* __restrict__ count, const ScanSlot* __restrict__ slots,
                                                       const int* __restrict__ scans, double* __restrict__ out) {
    const int tid = threadIdx.x, scan = scans[blockIdx.x];
    if (tid >= kEskfOut) return;
    const int n = count[scan], r0 = slots[scan].base / 256, nb = (n + 255) / 256;
    double acc = 0.0;
    for (int b = 0; b < nb; ++b) acc += partial[(size_t)(r0 + b) * kEskfOut + tid];
    out[(size_t)scan * kEskfOut + tid] = acc;
}

__global__ __launch_bounds__(kSegBlock) void k_sel_count(const uint8_t* __restrict__ selected, const int* __restrict__ count,
                                                         const ScanSlot* __restrict__ slots, const SegBlock* __restrict__ blocks,
                                                         int* __restrict__ block_counts) {
    const SegBlock b = blocks[blockIdx.x];
    const int i = b.start + threadIdx.x;
    const bool f = i < count[b.scan] && selected[slots[b.scan].base + i];
    const int c = __syncthreads_count(f);
    if (threadIdx.x == 0) block_counts[blockIdx.x] = c;
}

__global__ __launch_bounds__(kSegBlock) void k_sel_scatter(const uint8_t* __restrict__ selected, const int* __restrict__ count,
                                                           const ScanSlot* __restrict__ slots, const SegBlock* __restrict__ blocks,
                                                           const int* __restrict__ block_offsets, const PointXYZINormal* __restrict__ body,
                                                           const PointXYZINormal* __restrict__ normvec,
                                                           PointXYZINormal* __restrict__ cloud_ori, PointXYZINormal* __restrict__ corr_norm) {
    __shared__ int s_wave[kSegBlock / 64];
    const SegBlock b = blocks[blockIdx.x];
    const ScanSlot sl = slots[b.scan];
    const int i = b.start + threadIdx.x;
    const bool f = i < count[b.scan] && selected[sl.base + i];
    int total;
    const int pos = block_flag_scan(f, s_wave, total);
    if (f) {
        const int o = sl.base + block_offsets[blockIdx.x] + pos;
        cloud_ori[o] = body[sl.base + i];
        corr_norm[o] = normvec[sl.base + i];
    }
}

// ---- launch wrappers ----------------------------------------------------------------------------------------------
void launch_pre_count(const VelodynePoint* raw, const int* raw_count, const ScanSlot* slots, const SegBlock* blocks, int nblocks,
                      PreprocessParams prm, int* block_counts, hipStream_t st) {
    if (nblocks) TC2LI_LAUNCH(k_pre_count, dim3(nblocks), dim3(kSegBlock), 0, st, raw, raw_count, slots, blocks, prm, block_counts);
}
void launch_pre_stream(const VelodynePoint* raw, const int* raw_count, const ScanSlot* slots, int nscans, PreprocessParams prm, PointXYZINormal* out,
                       int* out_count, int* bbox_enc, float* time_out, int* vkey_out, float leaf, int* vk_ok, hipStream_t st) {
    if (nscans) TC2LI_LAUNCH(k_pre_stream, dim3(nscans), dim3(kSegBlock), 0, st, raw, raw_count, slots, prm, out, out_count, bbox_enc, time_out,
                             reinterpret_cast<uint32_t*>(vkey_out), vkey_out ? 1.0f / leaf : 0.f, vkey_out ? vk_ok : nullptr);
}
void launch_seg_scan(const ScanSlot* slots, int nscans, const int* block_counts, int* block_offsets, int* totals, hipStream_t st) {
    if (nscans) TC2LI_LAUNCH(k_seg_scan, dim3(nscans), dim3(256), 0, st, slots, block_counts, block_offsets, totals);
}
void launch_pre_scatter(const VelodynePoint* raw, const int* raw_count, const ScanSlot* slots, const SegBlock* blocks, int nblocks,
                        PreprocessParams prm, const int* block_offsets, PointXYZINormal* out, hipStream_t st) {
    if (nblocks) TC2LI_LAUNCH(k_pre_scatter, dim3(nblocks), dim3(kSegBlock), 0, st, raw, raw_count, slots, blocks, prm, block_offsets, out);
}
void launch_voxel_bbox(const PointXYZINormal* pts, const int* count, const ScanSlot* slots, const SegBlock* blocks, int nblocks,
                       int* bbox_enc, hipStream_t st) {
    if (nblocks) TC2LI_LAUNCH(k_voxel_bbox, dim3(nblocks), dim3(kSegBlock), 0, st, pts, count, slots, blocks, bbox_enc);
}
void launch_voxel_params(const int* bbox_enc, const int* count, const ScanSlot* slots, int nscans, float leaf, VoxelParams* vp,
                         hipStream_t st) {
    (void)slots;
    if (nscans) TC2LI_LAUNCH(k_voxel_params, dim3((nscans + 63) / 64), dim3(64), 0, st, bbox_enc, count, nscans, leaf, vp);
}
void launch_voxel_insert(const PointXYZINormal* pts, const int* count, const ScanSlot* slots, const SegBlock* blocks, int nblocks,
                         float leaf, const VoxelParams* vp, int* table_keys, int* table_counts, int* pt_slot, int* n_vox, int* vox_keys, hipStream_t st) {
    if (nblocks) TC2LI_LAUNCH(k_voxel_insert, dim3((nblocks + 7) / 8 * 8, kSegBlock / kInsertThreads), dim3(kInsertThreads), 0, st, pts, count, slots, blocks, leaf, vp, table_keys, table_counts, pt_slot, n_vox, vox_keys, nblocks);
}
void launch_voxel_sort(const ScanSlot* slots, int nscans, const VoxelParams* vp, const int* count, const int* table_keys,
                       const int* table_counts, int* table_rank, int* vox_keys, int* vox_member_off, int* vox_fill, int* vox_count, int* n_vox,
                       int* status, hipStream_t st) {
    if (!nscans) return;
    (void)ensure_dynamic_lds((const void*)k_voxel_sort, kMaxVoxelsPerScan * 4);
    TC2LI_LAUNCH(k_voxel_sort, dim3(nscans), dim3(1024), kMaxVoxelsPerScan * 4, st, slots, vp, count, table_keys, table_counts,
                       table_rank, vox_keys, vox_member_off, vox_fill, vox_count, n_vox, status);
}
void launch_voxel_fill(const int* count, const ScanSlot* slots, const SegBlock* blocks, int nblocks, const VoxelParams* vp, const int* pt_slot,
                       const int* table_rank, const int* vox_member_off, int* vox_fill, int* members, hipStream_t st) {
    if (nblocks) TC2LI_LAUNCH(k_voxel_fill, dim3((nblocks + 7) / 8 * 8), dim3(kSegBlock), 0, st, count, slots, blocks, vp, pt_slot, table_rank, vox_member_off, vox_fill, members, nblocks);
}
void launch_voxel_centroid(const PointXYZINormal* pts, const int* count, const ScanSlot* slots, const SegBlock* blocks, int nblocks,
                           float leaf, const VoxelParams* vp, const int* pt_slot, const int* table_rank, const int* n_vox,
                           const int* vox_member_off, const int* vox_fill, const int* vox_count, const int* members, void* recs, PointXYZINormal* out,
                           int* out_count, hipStream_t st) {
    if (!nblocks) return;
    TC2LI_LAUNCH(k_voxel_rank, dim3((nblocks + 7) / 8 * 8, kSegBlock / 256), dim3(256), 0, st, pts, count, slots, blocks, leaf, vp, pt_slot, table_rank,
                       vox_member_off, vox_fill, members, (CentroidRec*)recs, nblocks);
    TC2LI_LAUNCH(k_voxel_centroid, dim3((nblocks + 7) / 8 * 8, kSegBlock / 256), dim3(256), 0, st, pts, count, slots, blocks, vp, n_vox, vox_member_off,
                       vox_count, (const CentroidRec*)recs, out, out_count, nblocks);
}
void launch_voxel_sort_points(const PointXYZINormal* pts, const int* count, const ScanSlot* slots, int nscans, float leaf, const VoxelParams* vp, int* key_a,
                              int* idx_a, int* key_b, int* idx_b, int* vox_start, int* vox_info, int* n_vox, const int* vk_ok, hipStream_t st) {
    if (!nscans) return;
    TC2LI_LAUNCH(k_voxel_sort_points, dim3(nscans), dim3(kVsThreads), 0, st, pts, count, slots, vp, leaf, reinterpret_cast<uint32_t*>(key_a), idx_a,
                 reinterpret_cast<uint32_t*>(key_b), idx_b, vox_start, vox_info, n_vox, vk_ok);
}
// (slots, blocks, nblocks): a block list that covers the input points; (vslots, vblocks, nvblocks): one that covers the voxels (n_vox per scan)
void launch_voxel_sums(const PointXYZINormal* pts, const ScanSlot* slots, const SegBlock* blocks, int nblocks, const ScanSlot* vslots, const SegBlock* vblocks,
                       int nvblocks, const VoxelParams* vp, const int* idx_a, const int* idx_b, const int* vox_start, const int* vox_info, const int* n_vox,
                       void* recs, PointXYZINormal* out, int* out_count, hipStream_t st) {
    if (!nvblocks) return;
    // TC2LI_VOXEL_FUSED=0: the two-pass form (the points' fields written in sorted order, then summed)
    const char* fused_env = getenv("TC2LI_VOXEL_FUSED");  // (read per call: the tests switch it)
    const bool fused = !(fused_env && atoi(fused_env) == 0);
    if (fused) {
        TC2LI_LAUNCH(k_voxel_centroid_fused, dim3((nvblocks + 7) / 8 * 8, kSegBlock / 256), dim3(256), 0, st, pts, vslots, vblocks, vp, n_vox, vox_start, vox_info,
                     idx_a, idx_b, out, out_count, nvblocks);
        return;
    }
    if (nblocks)
        TC2LI_LAUNCH(k_voxel_gather_sorted, dim3((nblocks + 7) / 8 * 8, kSegBlock / 256), dim3(256), 0, st, pts, slots, blocks, vp, vox_info, idx_a, idx_b,
                     (CentroidRec*)recs, nblocks);
    TC2LI_LAUNCH(k_voxel_centroid_sorted, dim3((nvblocks + 7) / 8 * 8, kSegBlock / 256), dim3(256), 0, st, pts, vslots, vblocks, vp, n_vox, vox_start, vox_info,
                 (const CentroidRec*)recs, out, out_count, nvblocks);
}
void launch_knn_plane(const MapGrid* grids, const PointXYZINormal* body, const int* count,
                      const ScanSlot* slots, const SegBlock* blocks, int nblocks, const LidarStateDev* states, PointXYZINormal* world,
                      uint8_t* selected, PointXYZINormal* normvec, int* nearest_idx, float* nearest_d, int* nfound, int* hard_count,
                      int2* hard_list, hipStream_t st, hipEvent_t after_first) {
    if (!nblocks) return;
    (void)hipMemsetAsync(hard_count, 0, sizeof(int), st);
    TC2LI_LAUNCH(k_knn_plane, dim3((nblocks + 7) / 8 * 8, kSegBlock / 64), dim3(256), 0, st, grids, body, count, slots, blocks, states, world, selected,
                       normvec, nearest_idx, nearest_d, nfound, hard_count, hard_list, nblocks);
    if (after_first) (void)hipEventRecord(after_first, st);
    TC2LI_LAUNCH(k_knn_hard, dim3(512), dim3(256), 0, st, grids, body, slots, states, selected, normvec, nearest_idx, nearest_d, nfound,
                       hard_count, hard_list);
}
void launch_sel_count(const uint8_t* selected, const int* count, const ScanSlot* slots, const SegBlock* blocks, int nblocks,
                      int* block_counts, hipStream_t st) {
    if (nblocks) TC2LI_LAUNCH(k_sel_count, dim3(nblocks), dim3(kSegBlock), 0, st, selected, count, slots, blocks, block_counts);
}
void launch_sel_scatter(const uint8_t* selected, const int* count, const ScanSlot* slots, const SegBlock* blocks, int nblocks,
                        const int* block_offsets, const PointXYZINormal* body, const PointXYZINormal* normvec,
                        PointXYZINormal* cloud_ori, PointXYZINormal* corr_norm, hipStream_t st) {
    if (nblocks) TC2LI_LAUNCH(k_sel_scatter, dim3(nblocks), dim3(kSegBlock), 0, st, selected, count, slots, blocks, block_offsets, body, normvec, cloud_ori, corr_norm);
}

void launch_eskf_refit(const MapGrid& grid, const PointXYZINormal* body, int n, const LidarStateDev* state, const int* nearest_idx,
                       PointXYZINormal* world, uint8_t* selected, PointXYZINormal* normvec, hipStream_t st) {
    if (n) TC2LI_LAUNCH(k_eskf_refit, dim3((n + 255) / 256), dim3(256), 0, st, grid, body, n, state, nearest_idx, world, selected, normvec);
}
void launch_eskf_refit_batch(const MapGrid* grids, const PointXYZINormal* body, const int* count, const ScanSlot* slots, const SegBlock* blocks, int nblocks,
                             const LidarStateDev* states, const int* nearest_idx, PointXYZINormal* world, uint8_t* selected, PointXYZINormal* normvec,
                             hipStream_t st) {
    if (nblocks) TC2LI_LAUNCH(k_eskf_refit_b, dim3(nblocks, kSegBlock / 256), dim3(256), 0, st, grids, body, count, slots, blocks, states, nearest_idx, world, selected, normvec);
}
void launch_eskf_normal_batch(const PointXYZINormal* body, const int* count, const ScanSlot* slots, const SegBlock* blocks, int nblocks,
                              const LidarStateDev* states, const uint8_t* selected, const PointXYZINormal* normvec, int extrinsic_est_en, double* partial,
                              const int* scans, int n_scans, double* out, hipStream_t st) {
    if (nblocks) TC2LI_LAUNCH(k_eskf_normal_b, dim3(nblocks, kSegBlock / 256), dim3(256), 0, st, body, count, slots, blocks, states, selected, normvec, extrinsic_est_en, partial);
    if (n_scans) TC2LI_LAUNCH(k_eskf_reduce_b, dim3(n_scans), dim3(256), 0, st, partial, count, slots, scans, out);
}
void launch_eskf_normal(const PointXYZINormal* body, int n, const LidarStateDev* state, const uint8_t* selected, const PointXYZINormal* normvec,
                        int extrinsic_est_en, double* partial, double* out, hipStream_t st) {
    const int nb = (n + 255) / 256;
    if (nb) TC2LI_LAUNCH(k_eskf_normal, dim3(nb), dim3(256), 0, st, body, n, state, selected, normvec, extrinsic_est_en, partial);
    TC2LI_LAUNCH(k_eskf_reduce, dim3(1), dim3(256), 0, st, partial, nb, out);
}


// ---- b4: LidarFrontEndTools::transformPointCloud for a batch of clouds (one 48-byte read and one 48-byte write per point) --------------
__global__ __launch_bounds__(256) void k_transform_points(const TransformTask* __restrict__ tasks) {
    const TransformTask& T = tasks[blockIdx.y];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= T.n) return;
    const PointXYZINormal p = T.in[i];
    PointXYZINormal o;
    o.x = (T.R[0] * p.x + T.R[1] * p.y + T.R[2] * p.z) + T.t[0];
    o.y = (T.R[3] * p.x + T.R[4] * p.y + T.R[5] * p.z) + T.t[1];
    o.z = (T.R[6] * p.x + T.R[7] * p.y + T.R[8] * p.z) + T.t[2];
    o.pad0 = 1.0f;
    o.normal_x = 0; o.normal_y = 0; o.normal_z = 0; o.pad1 = 0;
    o.intensity = p.intensity; o.curvature = 0; o.pad2 = 0; o.pad3 = 0;
    T.out[i] = o;
}
void launch_transform_points(const TransformTask* tasks, int n_tasks, int max_points, hipStream_t st) {
    if (n_tasks > 0 && max_points > 0) TC2LI_LAUNCH(k_transform_points, dim3((max_points + 255) / 256, n_tasks), dim3(256), 0, st, tasks);
}

}  // namespace tc2li
