// Device helpers shared by the LiDAR kernel files (lidar_kernels.hip, map_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lidar_device.hpp"

namespace tc2li {

// Exclusive prefix of a per-thread flag over a 1024-thread block (16 wavefronts); returns the block total in `total`.
__device__ __forceinline__ int block_flag_scan(bool f, int* s_wave, int& total) {
    const int lane = threadIdx.x & 63, wave = wave_in_block();
    const unsigned long long bal = __ballot(f);
    if (lane == 0) s_wave[wave] = __popcll(bal);
    __syncthreads();
    int off = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < kSegBlock / 64; ++k) {
        const int c = s_wave[k];
        off += k < wave ? c : 0;
        tot += c;
    }
    total = tot;
    __syncthreads();
    return off + __popcll(bal & ((1ull << lane) - 1ull));
}

// Runs of equal keys along a wavefront: lanes that continue their left neighbour's key leave the shared-counter work to the run's
// first lane (one atomic per run instead of per lane) and take its result by shuffle.  Lanes without work pass keys no neighbour shares.
struct RunInfo { bool head; int head_lane, length; };
__device__ __forceinline__ RunInfo wave_runs(int key) {
    const int lane = threadIdx.x & 63;
    const int left = __shfl_up(key, 1, 64);
    RunInfo r;
    r.head = lane == 0 || left != key;
    const unsigned long long heads = __ballot(r.head);
    const unsigned long long upto = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);
    r.head_lane = 63 - __clzll(heads & upto);
    const unsigned long long above = lane == 63 ? 0ull : (heads >> (lane + 1));
    r.length = above ? __ffsll((long long)above) : 64 - lane;  // for a head: lanes up to the next head
    return r;
}


__device__ __forceinline__ int enc_float(float f) { const int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7fffffff; }
__device__ __forceinline__ float dec_float(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); }
// Cells are ordered x-fastest, so the points of a run of cells along x are one contiguous range of the sorted array.
__device__ __forceinline__ int map_cell(const MapGrid& g, float x, float y, float z) {
    const int cx = (int)floorf(x * g.inv_cell) - g.x0, cy = (int)floorf(y * g.inv_cell) - g.y0, cz = (int)floorf(z * g.inv_cell) - g.z0;
    return (cz * g.ny + cy) * g.nx + cx;
}

// index into MapGrid::bucket_start of cell cx of row `row` (= iz * ny + iy); + 1 = the end of that cell's entries
__device__ __forceinline__ int map_start_index(const MapGrid& g, int row, int cx) {
    return (row * g.nsx + (cx >> 4)) * kMapSegStride + (cx & (kMapSegCells - 1));
}

// pointBodyToWorld (LidarFrontEnd.cpp:130-139): double arithmetic, float result
__device__ __forceinline__ PointXYZINormal body_to_world(const PointXYZINormal& pb, const LidarStateDev& st) {
    const double bx = pb.x, by = pb.y, bz = pb.z;
    double t[3], g[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) t[r] = (st.off_r[3 * r] * bx + st.off_r[3 * r + 1] * by + st.off_r[3 * r + 2] * bz) + st.off_t[r];
#pragma unroll
    for (int r = 0; r < 3; ++r) g[r] = (st.rot[3 * r] * t[0] + st.rot[3 * r + 1] * t[1] + st.rot[3 * r + 2] * t[2]) + st.pos[r];
    PointXYZINormal pw;
    pw.x = (float)g[0]; pw.y = (float)g[1]; pw.z = (float)g[2]; pw.pad0 = 1.0f;
    pw.normal_x = 0; pw.normal_y = 0; pw.normal_z = 0; pw.pad1 = 0;
    pw.intensity = pb.intensity; pw.curvature = 0; pw.pad2 = 0; pw.pad3 = 0;
    return pw;
}

__device__ __forceinline__ float calc_dist3(float ax, float ay, float az, float bx, float by, float bz) {
    return (ax - bx) * (ax - bx) + (ay - by) * (ay - by) + (az - bz) * (az - bz);
}

}  // namespace tc2li
