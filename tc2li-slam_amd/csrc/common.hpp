// Host-side helpers shared by the C-ABI translation units: error reporting, HIP call checking, a small
// persistent worker pool for the sequential per-(image, level) host stages, RAII device/pinned buffers.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <map>
#include <vector>

#include "../../include/tc2li_hip.h"

#include "launch.hpp"

namespace tc2li {

void set_error(const char* fmt, ...);
bool device_ready();
// hipStreamSynchronize for waits that last milliseconds (the front-end stages): the calling thread sleeps on an event made with
// sleeping between looks at an event instead of spinning on the queue -- several host threads wait on the GPU at once and a spinning
// thread costs a whole core (the GPU boxes give a process 16); see common.cpp.  The host-driven lock-step BA loop (TC2LI_BA_DEVICE_LM=0)
// keeps spinning on its short phases: a wake-up costs more than one of its kernels.
hipError_t stream_wait_blocking(hipStream_t st);
// the wait itself, on an event the caller has recorded: a short look, then sleeps between looks (common.cpp)
hipError_t event_wait_sleeping(hipEvent_t ev);
// The stream of the entry points that take none (single-scan / single-frame calls): one non-blocking stream per host thread and
// device, so such a call never touches the NULL stream -- work on the NULL stream serialises against every blocking stream of the
// process (the other stage threads of the caller).  nullptr (= the NULL stream) only when the stream cannot be created.
hipStream_t private_stream();
// hipMemcpy on a stream of our own: asynchronous copy + wait for that stream only.
hipError_t copy_sync(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t st);
hipError_t memset_sync(void* dst, int value, size_t bytes, hipStream_t st);

// One copy (src: device memory or PINNED host memory) or zero fill (src == NULL) of a batch executed by a single launch (copy_kernels.hip).
struct CopyTask { void* dst; const void* src; size_t bytes; };
void launch_copy_tasks(const CopyTask* tasks /* pinned or device memory */, int n, size_t max_bytes, hipStream_t st);
// While a CopySink is alive on a thread, upload_or_defer / zero_or_defer append to its list instead of queueing a copy of their own; the
// owner of the list launches it once for the whole batch.  Sources handed to upload_or_defer must then be pinned and stay valid until
// that launch has run.
struct CopySink {
    explicit CopySink(std::vector<CopyTask>* list);
    ~CopySink();
    CopySink(const CopySink&) = delete;
    CopySink& operator=(const CopySink&) = delete;
    std::vector<CopyTask>* prev_;
};
bool copy_sink_active();
hipError_t upload_or_defer(void* dst, const void* src, size_t bytes, hipStream_t st);
hipError_t zero_or_defer(void* dst, size_t bytes, hipStream_t st);

#define TC2LI_HIP_CHECK(call)                                                                       \
    do {                                                                                            \
        hipError_t e_ = (call);                                                                     \
        if (e_ != hipSuccess) {                                                                     \
            ::tc2li::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return TC2LI_ERR_HIP;                                                                   \
        }                                                                                           \
    } while (0)

extern std::atomic<long> g_buffer_allocs;  // hipMalloc / hipHostMalloc calls of DevBuf / PinnedBuf so far (TC2LI_BA_TIMING reports them)

// Growing a DevBuf / PinnedBuf is hipFree + hipMalloc, and hipFree waits for the whole device.  A one-off for the batch calls (a window
// always meets the same work space); in the bundle-adjustment engine windows of every size pass through every work space, and a
// synchronisation per growth stalls every stream of the process.  While a BufferCacheScope is alive on a thread, buffers released there go
// to the cache instead of back to the runtime, and allocations are served from it when a block of at least (and at most twice) the size is
// there.  The owner guarantees what hipFree's wait did: a buffer is released only when no queued work uses it.
struct BufferCache {
    std::mutex mu;
    std::multimap<size_t, void*> device, pinned;   // bytes -> block
    ~BufferCache();
    void* take(bool is_pinned, size_t bytes, size_t* real_bytes);
    void give(bool is_pinned, void* p, size_t bytes);
};
extern thread_local BufferCache* tl_buffer_cache;
struct BufferCacheScope {
    BufferCache* prev_;
    explicit BufferCacheScope(BufferCache* c) : prev_(tl_buffer_cache) { tl_buffer_cache = c; }
    ~BufferCacheScope() { tl_buffer_cache = prev_; }
    BufferCacheScope(const BufferCacheScope&) = delete;
    BufferCacheScope& operator=(const BufferCacheScope&) = delete;
};

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    DevBuf() {}
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
    ~DevBuf() { release(); }
    void release() {
        if (p) { if (tl_buffer_cache) tl_buffer_cache->give(false, p, n * sizeof(T)); else (void)hipFree(p); }
        p = nullptr; n = 0;
    }
    hipError_t alloc(size_t count) {
        release();
        if (count == 0) return hipSuccess;
        if (tl_buffer_cache) {
            size_t real = 0;
            if (void* q = tl_buffer_cache->take(false, count * sizeof(T), &real)) { p = (T*)q; n = real / sizeof(T); return hipSuccess; }
        }
        g_buffer_allocs.fetch_add(1, std::memory_order_relaxed);
        hipError_t e = hipMalloc((void**)&p, count * sizeof(T));
        if (e == hipSuccess) n = count;
        return e;
    }
    hipError_t ensure(size_t count) { return count <= n ? hipSuccess : alloc(count + count / 4); }
    hipError_t upload(const std::vector<T>& v) {
        hipError_t e = alloc(v.size());
        if (e != hipSuccess || v.empty()) return e;
        return hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
    }
};

template <typename T>
struct PinnedBuf {
    T* p = nullptr;
    size_t n = 0;
    PinnedBuf() {}
    PinnedBuf(const PinnedBuf&) = delete;
    PinnedBuf& operator=(const PinnedBuf&) = delete;
    PinnedBuf(PinnedBuf&& o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
    ~PinnedBuf() { release(); }
    void release() {
        if (p) { if (tl_buffer_cache) tl_buffer_cache->give(true, p, n * sizeof(T)); else (void)hipHostFree(p); }
        p = nullptr; n = 0;
    }
    hipError_t alloc(size_t count) {
        release();
        if (count == 0) return hipSuccess;
        if (tl_buffer_cache) {
            size_t real = 0;
            if (void* q = tl_buffer_cache->take(true, count * sizeof(T), &real)) { p = (T*)q; n = real / sizeof(T); return hipSuccess; }
        }
        g_buffer_allocs.fetch_add(1, std::memory_order_relaxed);
        hipError_t e = hipHostMalloc((void**)&p, count * sizeof(T), hipHostMallocDefault);
        if (e == hipSuccess) n = count;
        return e;
    }
    hipError_t ensure(size_t count) { return count <= n ? hipSuccess : alloc(count + count / 4); }
};

// Runs fn(i) for i in [0, n) on a fixed set of worker threads (the caller participates).
class WorkerPool {
public:
    explicit WorkerPool(int nthreads, const char* name = nullptr);
    ~WorkerPool();
    void parallel_for(int n, const std::function<void(int)>& fn);
    int size() const { return (int)workers_.size() + 1; }

private:
    void loop();
    std::vector<std::thread> workers_;
    std::mutex mu_, call_mu_;  // call_mu_ serialises concurrent parallel_for callers
    std::condition_variable cv_, done_cv_;
    const std::function<void(int)>* fn_ = nullptr;
    std::atomic<int> next_{0};
    int n_ = 0, generation_ = 0, active_ = 0;
    int device_ = -1;  // HIP device of the thread that called parallel_for: the current device is per thread, and a new thread starts on device 0
    bool stop_ = false;
};

// The library's pools, created on first use and destroyed (threads joined) by tc2li_shutdown; sizes from the host-thread budget
// (tc2li_set_host_thread_budget / TC2LI_HOST_THREAD_BUDGET / the cores this process may run on), common.cpp pool_threads().
constexpr int kMaxLockstepGroups = 8;
enum PoolId {
    kPoolGlobal = 0,   // the ORB extractor's host stages
    kPoolTracking,     // the tracking thread's host steps
    kPoolLidar,        // per-scan host steps of the LiDAR batch calls
    kPoolBaTop,        // one thread per lock-step group of tc2li_local_bundle_adjustment_batch
    kPoolLviTop,       // the same for tc2li_local_lvi_bundle_adjustment_batch
    kPoolBaGroup0,     // + g: host threads of lock-step group g (LV-BA)
    kPoolLviGroup0 = kPoolBaGroup0 + kMaxLockstepGroups,
    kPoolCount = kPoolLviGroup0 + kMaxLockstepGroups
};
WorkerPool& named_pool(int id);
// fn runs inside tc2li_shutdown (after the pools' threads have been joined, while the HIP runtime is alive); hooks stay registered
void at_shutdown(std::function<void()> fn);
// The one process-wide instance of a work space T (Tag tells apart two of the same type): made on first use, destroyed by
// tc2li_shutdown and made again by the next use; never destroyed at process exit, where its hipFree calls would run after the HIP
// runtime's own teardown.
template <typename T, int Tag = 0>
T& shutdown_owned() {
    static T* p = nullptr;
    static std::mutex mu;
    static bool registered = false;
    std::lock_guard<std::mutex> lk(mu);
    if (!registered) {
        registered = true;
        at_shutdown([] { std::lock_guard<std::mutex> lk2(mu); delete p; p = nullptr; });
    }
    if (!p) p = new T();
    return *p;
}
int pool_threads(int id);
int host_thread_budget();
void note_hip_touched();  // tc2li_set_hardware_queues refuses once the library has called into HIP
WorkerPool& global_pool();    // the ORB extractor's host stages (quadtree per image and level)
// A second pool for the tracking thread's host steps (stereo / matcher / tracking entry points): WorkerPool::parallel_for serialises
// its callers, so on one shared pool the milliseconds-long quadtree phase of the extraction thread would stall every short
// parallel_for of the tracking thread that runs beside it.
WorkerPool& tracking_pool();

}  // namespace tc2li
