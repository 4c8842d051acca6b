// Many small copies / fills as ONE launch.  A lock-step group of local-BA windows used to queue ~9 hipMemcpyAsync / hipMemsetAsync calls
// per window (input block, operand fill, plane clusters, results): ~1100 blit kernels of ~20 us per batch of 128 windows, all of them
// links of the group's dependency chain.  Here the host only writes a task list (pinned memory, read by the kernel in place), and the
// kernel moves the bytes: uploads read pinned host memory over the bus, results are written to it.
#include <hip/hip_runtime.h>

#include "common.hpp"
#include "launch.hpp"

namespace tc2li {

__global__ __launch_bounds__(256) void k_copy_tasks(const CopyTask* __restrict__ tasks) {
    const CopyTask T = tasks[blockIdx.y];
    const size_t stride = (size_t)gridDim.x * 256, first = (size_t)blockIdx.x * 256 + threadIdx.x;
    uint8_t* dst = static_cast<uint8_t*>(T.dst);
    const uint8_t* src = static_cast<const uint8_t*>(T.src);
    const bool wide = ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15) == 0;
    if (wide) {
        const size_t n16 = T.bytes / 16;
        uint4* d = reinterpret_cast<uint4*>(dst);
        const uint4* s = reinterpret_cast<const uint4*>(src);
        if (src) for (size_t i = first; i < n16; i += stride) d[i] = s[i];
        else for (size_t i = first; i < n16; i += stride) d[i] = uint4{0, 0, 0, 0};
        for (size_t i = n16 * 16 + first; i < T.bytes; i += stride) dst[i] = src ? src[i] : (uint8_t)0;
    } else {
        for (size_t i = first; i < T.bytes; i += stride) dst[i] = src ? src[i] : (uint8_t)0;
    }
}

void launch_copy_tasks(const CopyTask* tasks, int n, size_t max_bytes, hipStream_t st) {
    if (n <= 0) return;
    const unsigned gx = (unsigned)std::max<size_t>(1, std::min<size_t>(64, (max_bytes / 16 + 1023) / 1024));
    TC2LI_LAUNCH(k_copy_tasks, dim3(gx, n), dim3(256), 0, st, tasks);
}

}  // namespace tc2li
