// Many small copies / fills as ONE launch.  A lock-step group of local-BA windows used to queue ~9 hipMemcpyAsync / hipMemsetAsync calls
// per window (input block, operand fill, plane clusters, results): ~1100 blit kernels of ~20 us per batch of 128 windows, all of them
// links of the group's dependency chain.  Here the host only writes a task list (pinned memory, read by the kernel in place), and the
// kernel moves the bytes: uploads read pinned host memory over the bus, results are written to it.  (Round 3: the few LARGE uploads
// of the list are handed to hipMemcpyAsync after all, see launch_copy_tasks.)
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "common.hpp"
#include "global_ptr.hpp"
#include "launch.hpp"

namespace tc2li {

// A fixed, small grid: every workgroup takes its slice of every task.  (One workgroup per 16 KB of every task, as it was first written,
// put ~1.5 million threads on the GPU for a batch of 43 windows: they filled every wavefront slot for the 1.4 ms the bus needs for the
// 60 MB, and the kernels of the other streams waited for slots behind them.)  kCopyGroups x 256 threads x 4 x 16 B in flight is enough
// to keep the bus busy.
constexpr int kCopyGroups = 128, kCopyBatch = 256;
__global__ __launch_bounds__(256) void k_copy_tasks(const CopyTask* __restrict__ tasks, int n) {
    __shared__ CopyTask s_tasks[kCopyBatch];  // the descriptors live in pinned host memory: fetched once, not once per task and thread
    const int tid = threadIdx.x, b = blockIdx.x, G = gridDim.x;
    for (int base = 0; base < n; base += kCopyBatch) {
        const int m = min(kCopyBatch, n - base);
        if (tid < m) s_tasks[tid] = tasks[base + tid];
        __syncthreads();
        for (int t = 0; t < m; ++t) {
            const CopyTask T = s_tasks[t];
            uint8_t* dst = global_ptr(static_cast<uint8_t*>(T.dst));  // device or pinned host memory: global, not flat, accesses
            const uint8_t* src = global_ptr(static_cast<const uint8_t*>(T.src));
            const bool wide = ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15) == 0;
            if (wide) {
                const size_t n16 = T.bytes / 16, per = (n16 + G - 1) / G, lo = (size_t)b * per, hi = min(n16, lo + per);
                uint4* d = reinterpret_cast<uint4*>(dst);
                const uint4* s = reinterpret_cast<const uint4*>(src);
                if (src) {
                    size_t i = lo + tid;
                    for (; i + 768 < hi; i += 1024) {  // four loads in flight per thread
                        const uint4 v0 = s[i], v1 = s[i + 256], v2 = s[i + 512], v3 = s[i + 768];
                        d[i] = v0; d[i + 256] = v1; d[i + 512] = v2; d[i + 768] = v3;
                    }
                    for (; i < hi; i += 256) d[i] = s[i];
                } else {
                    for (size_t i = lo + tid; i < hi; i += 256) d[i] = uint4{0, 0, 0, 0};
                }
                if (b == 0) for (size_t i = n16 * 16 + tid; i < T.bytes; i += 256) dst[i] = src ? src[i] : (uint8_t)0;
            } else {
                const size_t per = (T.bytes + G - 1) / G, lo = (size_t)b * per, hi = min(T.bytes, lo + per);
                for (size_t i = lo + tid; i < hi; i += 256) dst[i] = src ? src[i] : (uint8_t)0;
            }
        }
        __syncthreads();
    }
}

void launch_copy_tasks(const CopyTask* tasks, int n, size_t max_bytes, hipStream_t st) {
    if (n <= 0) return;
    // Copies of a megabyte or more (the windows' input blocks, 1.4 MB each) go to the runtime's copy path -- the SDMA engines -- instead
    // of the kernel: the same bytes cross the bus, but no wavefront waits on them.  With all stages running the step went from 33.5
    // to 30.0 ms (15.2 -> 17.0 k frames/s); with the threshold at 64 KB or 200 KB (results and plane clusters too: three times the
    // calls) 31.0-31.9 ms.  The list lives in pinned memory, so the host takes those entries out of it.  TC2LI_COPY_ENGINE_BYTES
    // sets the threshold, 0 leaves everything to the kernel.
    static const long engine_bytes = getenv("TC2LI_COPY_ENGINE_BYTES") ? atol(getenv("TC2LI_COPY_ENGINE_BYTES")) : 1000000;
    if (engine_bytes > 0) {
        CopyTask* list = const_cast<CopyTask*>(tasks);
        size_t left_max = 0;
        int left = 0;
        for (int i = 0; i < n; ++i) {
            if (list[i].src && (long)list[i].bytes >= engine_bytes) {
                (void)hipMemcpyAsync(list[i].dst, list[i].src, list[i].bytes, hipMemcpyDefault, st);
                list[i].bytes = 0;
            } else if (list[i].bytes) {
                left_max = std::max(left_max, list[i].bytes);
                ++left;
            }
        }
        if (!left) return;
        max_bytes = left_max;
    }
    // small batches: no more workgroups than 4 KB slices of the largest task
    static const int groups = getenv("TC2LI_COPY_GROUPS") ? atoi(getenv("TC2LI_COPY_GROUPS")) : kCopyGroups;
    const unsigned g = (unsigned)std::max<size_t>(1, std::min<size_t>(groups, (max_bytes + 4095) / 4096));
    TC2LI_LAUNCH(k_copy_tasks, dim3(g), dim3(256), 0, st, tasks, n);
}

}  // namespace tc2li
