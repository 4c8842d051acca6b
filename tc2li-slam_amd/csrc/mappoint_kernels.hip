// Per-map-point refresh of local mapping, one wavefront per point:
//   MapPoint::ComputeDistinctiveDescriptors   SF/src/MapPoint.cc:338-412
//   MapPoint::UpdateNormalAndDepth            SF/src/MapPoint.cc:444-503
// The N observed descriptors of a point are staged in LDS; lane i owns row i of the N x N Hamming-distance table (kept in LDS as
// 16-bit values) and finds the row's median -- the value sorted[(int)(0.5 (N - 1))] of the reference -- by bisection on the
// value range 0..256 (the smallest v with at least k + 1 entries <= v); the first row with the least median wins, as in the
// reference's loop.  The mean viewing direction is accumulated by one lane in observation order (float, like the reference).
#include <hip/hip_runtime.h>

#include "launch.hpp"
#pragma clang fp contract(off)
#include <stdint.h>

#include "mappoint_device.hpp"

namespace tc2li {

__global__ __launch_bounds__(64) void k_map_points_refresh(MapPointRefresh a) {
    __shared__ uint32_t s_desc[kMaxObservations * 8];
    __shared__ uint16_t s_dist[kMaxObservations * kMaxObservations];
    const int p = blockIdx.x, lane = threadIdx.x;
    const int b = a.obs_off[p], N = a.obs_off[p + 1] - b;
    if (N <= 0) { if (lane == 0) a.best_obs[p] = -1; return; }
    if (N > kMaxObservations) { if (lane == 0) a.best_obs[p] = -2; return; }  // the host entry takes these points
    const uint32_t* D = reinterpret_cast<const uint32_t*>(a.descriptors) + (size_t)b * 8;
    for (int k = lane; k < N * 8; k += 64) s_desc[k] = D[k];
    __syncthreads();
    for (int i = lane; i < N; i += 64) {
        uint32_t mine[8];
#pragma unroll
        for (int w = 0; w < 8; ++w) mine[w] = s_desc[i * 8 + w];
        for (int j = 0; j < N; ++j) {
            int d = 0;
#pragma unroll
            for (int w = 0; w < 8; ++w) d += __popc(mine[w] ^ s_desc[j * 8 + w]);
            s_dist[i * N + j] = (uint16_t)d;
        }
    }
    __syncthreads();
    const int kth = (int)(0.5 * (N - 1));
    int best_median = 0x7fffffff, best_row = 0x7fffffff;
    for (int i = lane; i < N; i += 64) {
        int lo = 0, hi = 256;  // smallest v with count(row <= v) >= kth + 1
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            int cnt = 0;
            for (int j = 0; j < N; ++j) cnt += s_dist[i * N + j] <= mid ? 1 : 0;
            if (cnt >= kth + 1) hi = mid; else lo = mid + 1;
        }
        if (lo < best_median) { best_median = lo; best_row = i; }  // rows of a lane ascend: the first minimum is kept
    }
    for (int o = 32; o >= 1; o >>= 1) {
        const int m = __shfl_xor(best_median, o, 64), r = __shfl_xor(best_row, o, 64);
        if (m < best_median || (m == best_median && r < best_row)) { best_median = m; best_row = r; }
    }
    if (lane == 0) {
        a.best_obs[p] = best_row;
        const float* c = a.centres + 3 * (size_t)b;
        const float px = a.positions[3 * p], py = a.positions[3 * p + 1], pz = a.positions[3 * p + 2];
        float nx = 0, ny = 0, nz = 0;
        for (int k = 0; k < N; ++k) {
            const float vx = px - c[3 * k], vy = py - c[3 * k + 1], vz = pz - c[3 * k + 2];
            const float nr = sqrtf(vx * vx + vy * vy + vz * vz);
            nx = nx + vx / nr; ny = ny + vy / nr; nz = nz + vz / nr;
        }
        const float rx = px - a.ref_centres[3 * p], ry = py - a.ref_centres[3 * p + 1], rz = pz - a.ref_centres[3 * p + 2];
        const float dist = sqrtf(rx * rx + ry * ry + rz * rz);
        const float mx = dist * a.level_scale[p];
        a.max_dist[p] = mx;
        a.min_dist[p] = mx / a.last_scale;
        a.normals[3 * p] = nx / (float)N; a.normals[3 * p + 1] = ny / (float)N; a.normals[3 * p + 2] = nz / (float)N;
    }
}

void launch_map_points_refresh(const MapPointRefresh& a, int n_points, hipStream_t st) {
    if (n_points > 0) TC2LI_LAUNCH(k_map_points_refresh, dim3(n_points), dim3(64), 0, st, a);
}

}  // namespace tc2li
