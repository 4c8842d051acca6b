// Sum of K <= 32 doubles per lane over the 64 lanes of a wavefront -- the result of the plain xor butterfly (x += shfl_xor(x, 32), 16, 8, 4, 2, 1
// for every value), bit for bit, with a sixth of its shuffles.  The butterfly forms, for every value, the same tree in all 64 lanes; here
// a lane keeps only the values it is responsible for: at offset 32 the lanes of the lower half keep values [0, 16) and receive the partner's
// copies of those, the upper half keeps [16, 32); at offset 16 half of what is left, and so on -- 16 + 8 + 4 + 2 + 1 exchanges and one plain
// step at offset 1.  Every partial sum adds the same two operands as the butterfly's (floating-point addition commutes exactly), in the same
// tree, so the result is the butterfly's.  Value k ends in the lanes with (lane >> 1) bit-reversed over 5 bits == k (wave_reduce_index).
#pragma once
#include <hip/hip_runtime.h>

namespace tc2li {

__device__ __forceinline__ int wave_reduce_index(int lane) {  // the value whose sum wave_reduce_32 leaves in this lane
    return ((lane >> 5) & 1) * 16 + ((lane >> 4) & 1) * 8 + ((lane >> 3) & 1) * 4 + ((lane >> 2) & 1) * 2 + ((lane >> 1) & 1);
}
template <int K>
__device__ __forceinline__ double wave_reduce_32(const double (&v)[K]) {
    static_assert(K <= 32, "at most 32 values");
    const int lane = threadIdx.x & 63;
    double a[16], b[8], c[4], d[2], e;
    {
        const bool hi = (lane & 32) != 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const double lo_v = i < K ? v[i] : 0.0, hi_v = 16 + i < K ? v[16 + i] : 0.0;
            const double r = __shfl_xor(hi ? lo_v : hi_v, 32, 64);
            a[i] = (hi ? hi_v : lo_v) + r;
        }
    }
    {
        const bool hi = (lane & 16) != 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) { const double r = __shfl_xor(hi ? a[i] : a[8 + i], 16, 64); b[i] = (hi ? a[8 + i] : a[i]) + r; }
    }
    {
        const bool hi = (lane & 8) != 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) { const double r = __shfl_xor(hi ? b[i] : b[4 + i], 8, 64); c[i] = (hi ? b[4 + i] : b[i]) + r; }
    }
    {
        const bool hi = (lane & 4) != 0;
#pragma unroll
        for (int i = 0; i < 2; ++i) { const double r = __shfl_xor(hi ? c[i] : c[2 + i], 4, 64); d[i] = (hi ? c[2 + i] : c[i]) + r; }
    }
    {
        const bool hi = (lane & 2) != 0;
        const double r = __shfl_xor(hi ? d[0] : d[1], 2, 64);
        e = (hi ? d[1] : d[0]) + r;
    }
    e += __shfl_xor(e, 1, 64);
    return e;
}

}  // namespace tc2li
