// Kernel launch macro of the library: hipLaunchKernelGGL, or -- while tc2li_profile_enable(1) -- hipExtLaunchKernelGGL with a start and a
// stop event bound to the dispatch itself (measurement).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <atomic>

namespace tc2li {
// The wavefront's number inside its workgroup as a SCALAR: threadIdx.x >> 6 is the same in all 64 lanes, but the compiler cannot know, and
// everything derived from it (the row / keypoint / problem a wavefront owns, its addresses) then lives in vector registers and is computed
// 64 times over; through readfirstlane it is a scalar register, the addresses are scalar arithmetic and the loads take a lane offset
// (round 6, k_blur7_strips: 78 -> 52 VGPRs from this line alone).
__device__ __forceinline__ int wave_in_block() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }
// Integer sums / minima over a wavefront without the LDS: a __shfl_xor is an LDS permute (10 ns per wavefront instruction per SIMD where a
// vector instruction is 1.2-2.7, tools/probes/valu_rate.hip), and a butterfly of six of them per value was most of what k_stereo_match's
// eleven window sums cost.  Four DPP steps inside each row of 16 lanes (quad_perm [1,0,3,2] and [2,3,0,1], row_half_mirror, row_mirror: after
// each the partner group's lanes all hold the same partial result, so mirroring is as good as the xor), then the four rows' results are
// read as scalars.  ALL 64 lanes must be active.  Integer arithmetic: any order gives the same bits.
template <int CTRL>
__device__ __forceinline__ int dpp_take(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
__device__ __forceinline__ int row_sum_i32(int v) {   // the sum of the lane's row of 16, in every lane of the row
    v += dpp_take<0xB1>(v); v += dpp_take<0x4E>(v); v += dpp_take<0x141>(v); v += dpp_take<0x140>(v);
    return v;
}
__device__ __forceinline__ int wave_sum_i32(int v) {  // -> the same scalar in all lanes
    v = row_sum_i32(v);
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
    v = min(v, (unsigned)dpp_take<0xB1>((int)v)); v = min(v, (unsigned)dpp_take<0x4E>((int)v));
    v = min(v, (unsigned)dpp_take<0x141>((int)v)); v = min(v, (unsigned)dpp_take<0x140>((int)v));
    return min(min((unsigned)__builtin_amdgcn_readlane((int)v, 0), (unsigned)__builtin_amdgcn_readlane((int)v, 16)),
               min((unsigned)__builtin_amdgcn_readlane((int)v, 32), (unsigned)__builtin_amdgcn_readlane((int)v, 48)));
}
// the sum over the lane's half of the wavefront (lanes 0..31 / 32..63), in every lane of that half
__device__ __forceinline__ int half_wave_sum_i32(int v) {
    v = row_sum_i32(v);
    const int lo = __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16), hi = __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
    return (threadIdx.x & 32) ? hi : lo;
}
// ---- per-launch timing (measurement only; include/tc2li_hip.h "tc2li_profile_*") ------------------------------------------------
// While enabled, every kernel launch of the library carries two HIP events that the runtime stamps when the dispatch starts and when
// it completes (hipExtLaunchKernel's startEvent / stopEvent): their distance is the kernel's own execution time -- what rocprofv3's
// kernel trace reports -- not the span between two markers on a stream that waits its turn on a shared GPU.  The report sums the
// durations per kernel name.  Off (the default) a launch costs one relaxed atomic load more.
namespace prof {
extern std::atomic<int> g_enabled;
bool acquire(const char* name, hipEvent_t* start, hipEvent_t* stop);  // false: no events (limit reached, creation failed): launch plainly
}  // namespace prof
// hipFuncAttributeMaxDynamicSharedMemorySize (a kernel gets 64 KB of dynamic LDS unless it asks for more) for kernel `fn` on the calling
// thread's current device: set once per (kernel, device), thread-safe.  false: the runtime refused -- the caller takes its fall-back
// kernel, or launches anyway and reports the launch error.
bool ensure_dynamic_lds(const void* fn, int bytes);
#define TC2LI_LAUNCH(kernel, grid, block, shmem, stream, ...)                                                                  \
    do {                                                                                                                       \
        hipEvent_t prof_a_ = nullptr, prof_b_ = nullptr;                                                                       \
        if (::tc2li::prof::g_enabled.load(std::memory_order_relaxed) && ::tc2li::prof::acquire(#kernel, &prof_a_, &prof_b_))   \
            hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, prof_a_, prof_b_, 0, __VA_ARGS__);                       \
        else                                                                                                                   \
            hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);                                               \
    } while (0)

}  // namespace tc2li
