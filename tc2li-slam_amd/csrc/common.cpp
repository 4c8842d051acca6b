#include "common.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <ctime>
#if defined(__linux__)
#include <pthread.h>
#include <sched.h>
#include <sys/prctl.h>
#endif

namespace tc2li {

static thread_local std::string g_last_error;

void set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

bool device_ready() {
    int n = 0;
    note_hip_touched();
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_error("no HIP device available (%s); this library has no CPU fallback",
                  e == hipSuccess ? "device count 0" : hipGetErrorString(e));
        (void)hipGetLastError();
        return false;
    }
    return true;
}

// Waiting for the GPU without holding a core (round 6).  hipEventSynchronize on a hipEventBlockingSync event -- what this function did in
// rounds 2-5 -- does not sleep on this runtime: every waiting thread showed 0.6-1.0 CPU-seconds per second in /proc (the five stage threads
// and the three lock-step groups of the bench: 6 of the 13 CPUs the loop kept busy).  Now: record an event, look at it (hipEventQuery) for
// the first 250 us -- the waits of a one-sequence call end there: with 40 us one host-fed sequence read 547 frames/s against 826 spinning -- and then between real sleeps that grow from 30 to
// 200 us (the thread's timer slack set to 1 us once, so that a 30 us sleep is not rounded up to 80).  TC2LI_SPIN_WAIT=1: hipStreamSynchronize.
hipError_t event_wait_sleeping(hipEvent_t ev) {
#if defined(__linux__)
    static thread_local bool slack_set = false;
    if (!slack_set) { (void)prctl(PR_SET_TIMERSLACK, 1000UL, 0UL, 0UL, 0UL); slack_set = true; }
#endif
    static const int spin_us = getenv("TC2LI_WAIT_SPIN_US") ? std::max(0, atoi(getenv("TC2LI_WAIT_SPIN_US"))) : 250;
    const auto t0 = std::chrono::steady_clock::now();
    long nap_ns = 30000;
    for (;;) {
        const hipError_t q = hipEventQuery(ev);
        if (q != hipErrorNotReady) return q;
        (void)hipGetLastError();  // (hipErrorNotReady is sticky in hipGetLastError otherwise)
        if (std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() < spin_us) {
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
            continue;
        }
        struct timespec ts = {0, nap_ns};
        (void)nanosleep(&ts, nullptr);
        nap_ns = std::min(200000L, nap_ns + nap_ns / 2);
    }
}
hipError_t stream_wait_blocking(hipStream_t st) {
    struct Ev {
        hipEvent_t e = nullptr;
        ~Ev() { if (e) (void)hipEventDestroy(e); }
    };
    static thread_local Ev ev;
    static const bool spin = getenv("TC2LI_SPIN_WAIT") != nullptr;  // A/B switch for measurements
    if (spin) return hipStreamSynchronize(st);
    if (!ev.e) {
        hipError_t e = hipEventCreateWithFlags(&ev.e, hipEventDisableTiming);
        if (e != hipSuccess) { ev.e = nullptr; return hipStreamSynchronize(st); }
    }
    hipError_t e = hipEventRecord(ev.e, st);
    if (e != hipSuccess) return e;
    return event_wait_sleeping(ev.e);
}

hipStream_t private_stream() {
    constexpr int kMaxDev = 16;
    struct Streams {
        hipStream_t s[kMaxDev] = {};  // never destroyed: a thread may end after the HIP runtime has been torn down
    };
    static thread_local Streams ts;
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= kMaxDev) { (void)hipGetLastError(); return nullptr; }
    if (!ts.s[d] && hipStreamCreateWithFlags(&ts.s[d], hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); ts.s[d] = nullptr; }
    return ts.s[d];
}

hipError_t copy_sync(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t st) {
    if (bytes == 0) return hipSuccess;
    hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, st);
    return e != hipSuccess ? e : hipStreamSynchronize(st);
}

namespace { thread_local std::vector<CopyTask>* tl_copy_sink = nullptr; }
CopySink::CopySink(std::vector<CopyTask>* list) : prev_(tl_copy_sink) { tl_copy_sink = list; }
CopySink::~CopySink() { tl_copy_sink = prev_; }
bool copy_sink_active() { return tl_copy_sink != nullptr; }
hipError_t upload_or_defer(void* dst, const void* src, size_t bytes, hipStream_t st) {
    if (bytes == 0) return hipSuccess;
    if (tl_copy_sink) { tl_copy_sink->push_back(CopyTask{dst, src, bytes}); return hipSuccess; }
    return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st);
}
hipError_t zero_or_defer(void* dst, size_t bytes, hipStream_t st) {
    if (bytes == 0) return hipSuccess;
    if (tl_copy_sink) { tl_copy_sink->push_back(CopyTask{dst, nullptr, bytes}); return hipSuccess; }
    return hipMemsetAsync(dst, 0, bytes, st);
}

hipError_t memset_sync(void* dst, int value, size_t bytes, hipStream_t st) {
    if (bytes == 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(dst, value, bytes, st);
    return e != hipSuccess ? e : hipStreamSynchronize(st);
}

bool ensure_dynamic_lds(const void* fn, int bytes) {
    struct Entry { const void* fn; int dev, bytes; bool ok; };
    static std::mutex mu;
    static std::vector<Entry>* done = new std::vector<Entry>();
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return false; }
    std::lock_guard<std::mutex> lk(mu);
    // only a GRANTED size answers for the smaller ones; a refusal is remembered for requests at least as large (a refused 156 KB must not
    // turn a later 38 KB request for the same kernel into a silent "no": ADVICE r4)
    for (const Entry& e : *done) {
        if (e.fn != fn || e.dev != dev) continue;
        if (e.ok && e.bytes >= bytes) return true;
        if (!e.ok && e.bytes <= bytes) return false;
    }
    const hipError_t err = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (err != hipSuccess) { (void)hipGetLastError(); set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize = %d) failed on device %d: %s", bytes, dev, hipGetErrorString(err)); }
    done->push_back(Entry{fn, dev, bytes, err == hipSuccess});
    return err == hipSuccess;
}

namespace prof {
std::atomic<int> g_enabled{0};
namespace {
struct Rec { const char* name; hipEvent_t a, b; };
std::mutex g_mu;
std::vector<Rec> g_recs;                              // launches since the last report
std::vector<std::pair<hipEvent_t, hipEvent_t>> g_free;  // event pairs to reuse
constexpr size_t kMaxRecs = 1u << 20;
}  // namespace
bool acquire(const char* name, hipEvent_t* start, hipEvent_t* stop) {
    std::pair<hipEvent_t, hipEvent_t> ev{nullptr, nullptr};
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (g_recs.size() >= kMaxRecs) return false;
        if (!g_free.empty()) { ev = g_free.back(); g_free.pop_back(); }
    }
    if (!ev.first && (hipEventCreate(&ev.first) != hipSuccess || hipEventCreate(&ev.second) != hipSuccess)) { (void)hipGetLastError(); return false; }
    std::lock_guard<std::mutex> lk(g_mu);
    g_recs.push_back(Rec{name, ev.first, ev.second});
    *start = ev.first; *stop = ev.second;
    return true;
}
}  // namespace prof

WorkerPool::WorkerPool(int nthreads, const char* name) {
    for (int i = 0; i < nthreads - 1; ++i) {
        workers_.emplace_back([this] { loop(); });
#if defined(__linux__)
        if (name) (void)pthread_setname_np(workers_.back().native_handle(), name);  // <= 15 characters: visible in /proc/<pid>/task/*/comm, top -H
#endif
    }
}

WorkerPool::~WorkerPool() {
    {
        std::lock_guard<std::mutex> lk(mu_);
        stop_ = true;
        ++generation_;
    }
    cv_.notify_all();
    for (auto& t : workers_) t.join();
}

void WorkerPool::loop() {
    int seen = 0, dev = -1, my_dev = -1;
    for (;;) {
        const std::function<void(int)>* fn;
        int n;
        {
            std::unique_lock<std::mutex> lk(mu_);
            cv_.wait(lk, [&] { return generation_ != seen; });
            seen = generation_;
            if (stop_) return;
            fn = fn_;
            n = n_;
            if (!fn) continue;  // woke after that generation had already been drained
            ++active_;
            dev = device_;
        }
        // the work runs on the caller's GPU (one process may see several; a worker thread would otherwise launch on device 0)
        if (dev >= 0 && dev != my_dev && hipSetDevice(dev) == hipSuccess) my_dev = dev;
        for (int i; (i = next_.fetch_add(1)) < n;) (*fn)(i);
        {
            std::lock_guard<std::mutex> lk(mu_);
            --active_;
        }
        done_cv_.notify_all();
    }
}

void WorkerPool::parallel_for(int n, const std::function<void(int)>& fn) {
    if (n <= 0) return;
    if (workers_.empty() || n == 1) {
        for (int i = 0; i < n; ++i) fn(i);
        return;
    }
    std::lock_guard<std::mutex> call(call_mu_);
    {
        std::lock_guard<std::mutex> lk(mu_);
        fn_ = &fn;
        n_ = n;
        next_.store(0);
        int d = -1;
        device_ = hipGetDevice(&d) == hipSuccess ? d : -1;
        if (device_ < 0) (void)hipGetLastError();  // no device: host-only work
        ++generation_;
    }
    cv_.notify_all();
    for (int i; (i = next_.fetch_add(1)) < n;) fn(i);
    // wait until every worker that picked this generation up has drained
    std::unique_lock<std::mutex> lk(mu_);
    done_cv_.wait(lk, [&] { return active_ == 0 && next_.load() >= n; });
    // workers that have not woken yet will find next_ >= n and do nothing with a still-valid n_
    fn_ = nullptr;
}

// ---- the library's host threads --------------------------------------------------------------------------------------------
// Every pool is created on first use and owned by this table, so that tc2li_shutdown can join the threads (whose thread-local work
// spaces -- device and pinned buffers, streams -- are released by their destructors while the HIP runtime is still alive).
namespace {
std::mutex g_pools_mu;
WorkerPool* g_pools[kPoolCount] = {};
std::atomic<int> g_thread_budget{0};  // 0: not set -> derived from the cores this process may run on
std::atomic<bool> g_hip_touched{false};
std::mutex g_hooks_mu;
std::vector<std::function<void()>>& hooks() { static auto* v = new std::vector<std::function<void()>>(); return *v; }

int cores_of_this_process() {
    int n = 0;
#if defined(__linux__)
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = CPU_COUNT(&set);
#endif
    if (n < 1) n = (int)std::thread::hardware_concurrency();
    return n < 1 ? 1 : n;
}
}  // namespace

void at_shutdown(std::function<void()> fn) {
    std::lock_guard<std::mutex> lk(g_hooks_mu);
    hooks().push_back(std::move(fn));
}

void note_hip_touched() { g_hip_touched.store(true, std::memory_order_relaxed); }

int host_thread_budget() {
    int b = g_thread_budget.load();
    if (b > 0) return b;
    if (const char* s = getenv("TC2LI_HOST_THREAD_BUDGET")) { b = atoi(s); if (b > 0) return b; }
    return cores_of_this_process();
}

// Threads of pool `id` under the budget B of this process.  B counts the CPUs the process may really use (bench.py / a caller passes
// min(affinity, cgroup quota) / ranks on the node); a pool thread spends part of its time waiting for its stream, so the pools may hold
// TC2LI_HOST_THREADS_PER_CPU (default 3 since round 6; 8 in round 5) threads per CPU in all.  Round 5 measured 2 per CPU -- 3 threads per
// lock-step group -- at 27.3 / 28.7 ms per step of the whole loop against 26.1 / 26.0 with 16: the per-window host steps between the phases of
// a group's 43 windows wanted more threads than CPUs.  Those steps are gone (the LM loop runs on the device, ba_device.hpp: BaLmState): on the
// 16-CPU box 8 / 3 / 2 per CPU now read 19.97-20.40 / 19.96-20.06 / 19.74-19.97 k frames/s with 110 / 40 / 26 threads and 6.0 / 4.8-5.1 / 4.0
// CPU-seconds per second, confined to 8 CPUs 18.8-19.0 k either way (15.4 k with the host-driven loop).  The stage threads of a caller like the reference (tracking, LiDAR, local
// mapping: counted as 5) come off first; of the rest the extractor pool may take a quarter, the tracking pool, the LiDAR pool and each
// lock-step BA group an eighth -- the one-GPU box's 16 CPUs give 10 / 5 / 5 / 5 per group (round 5: 30 / 15 / 15 / 15), 8 ranks on a node whose
// cgroup grants 16 CPUs in all (B = 2) 1 + 1 + 1 + 3 x 1.
std::atomic<long> g_buffer_allocs{0};
thread_local BufferCache* tl_buffer_cache = nullptr;
BufferCache::~BufferCache() {
    for (auto& kv : device) (void)hipFree(kv.second);
    for (auto& kv : pinned) (void)hipHostFree(kv.second);
}
void* BufferCache::take(bool is_pinned, size_t bytes, size_t* real_bytes) {
    std::lock_guard<std::mutex> lk(mu);
    auto& m = is_pinned ? pinned : device;
    auto it = m.lower_bound(bytes);
    if (it == m.end() || it->first > 2 * bytes + 4096) return nullptr;
    void* p = it->second;
    *real_bytes = it->first;
    m.erase(it);
    return p;
}
void BufferCache::give(bool is_pinned, void* p, size_t bytes) {
    std::lock_guard<std::mutex> lk(mu);
    (is_pinned ? pinned : device).emplace(bytes, p);
}

int pool_threads(int id) {
    int per_cpu = 3;
    if (const char* s = getenv("TC2LI_HOST_THREADS_PER_CPU")) per_cpu = std::max(1, std::min(16, atoi(s)));
    const int B = std::max(1, per_cpu * host_thread_budget() - 5);
    auto share = [&](int cap, int den) { return std::max(1, std::min(cap, B / den)); };
    if (id == kPoolGlobal) {
        if (const char* s = getenv("TC2LI_HOST_THREADS")) return std::max(1, std::min(32, atoi(s)));
        return share(32, 4);
    }
    if (id == kPoolTracking) {
        if (const char* s = getenv("TC2LI_TRACKING_THREADS")) return std::max(1, std::min(16, atoi(s)));
        return share(16, 8);
    }
    if (id == kPoolLidar) return share(16, 8);
    if (id == kPoolBaTop || id == kPoolLviTop) return kMaxLockstepGroups;  // one thread per lock-step group: they wait on their streams
    // lock-step BA groups: the setup of a group (graph structure, staging, plane extraction) and the per-window host steps between the phases
    if (const char* s = getenv("TC2LI_BA_GROUP_THREADS")) return std::max(1, atoi(s));
    return share(16, 8);
}

WorkerPool& named_pool(int id) {
    std::lock_guard<std::mutex> lk(g_pools_mu);
    if (!g_pools[id]) {
        const char* name = id == kPoolGlobal ? "tc2li-orb" : id == kPoolTracking ? "tc2li-track" : id == kPoolLidar ? "tc2li-lidar" :
                           id == kPoolBaTop || id == kPoolLviTop ? "tc2li-ba-top" : "tc2li-ba-group";
        g_pools[id] = new WorkerPool(pool_threads(id), name);
    }
    return *g_pools[id];
}

WorkerPool& global_pool() { return named_pool(kPoolGlobal); }
WorkerPool& tracking_pool() { return named_pool(kPoolTracking); }

}  // namespace tc2li

extern "C" {
int tc2li_profile_enable(int on) {
    tc2li::prof::g_enabled.store(on ? 1 : 0);
    return TC2LI_OK;
}
// text: one line per kernel, "name<TAB>launches<TAB>total_ms<NL>", sorted by total time; returns the number of bytes the whole report needs
int tc2li_profile_report(char* text, int capacity) {
    using namespace tc2li::prof;
    std::vector<Rec> recs;
    { std::lock_guard<std::mutex> lk(g_mu); recs.swap(g_recs); }
    struct Acc { std::string name; long calls = 0; double ms = 0; };
    std::vector<Acc> acc;
    for (const Rec& r : recs) {
        float ms = 0;
        if (hipEventSynchronize(r.b) != hipSuccess || hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) { (void)hipGetLastError(); ms = 0; }
        size_t k = 0;
        while (k < acc.size() && acc[k].name != r.name) ++k;
        if (k == acc.size()) { acc.emplace_back(); acc[k].name = r.name; }
        acc[k].calls++; acc[k].ms += ms;
    }
    { std::lock_guard<std::mutex> lk(g_mu); for (const Rec& r : recs) g_free.emplace_back(r.a, r.b); }
    std::sort(acc.begin(), acc.end(), [](const Acc& x, const Acc& y) { return x.ms > y.ms; });
    std::string out;
    char line[512];
    for (const Acc& a : acc) { snprintf(line, sizeof(line), "%s\t%ld\t%.6f\n", a.name.c_str(), a.calls, a.ms); out += line; }
    if (text && capacity > 0) { const size_t n = std::min((size_t)capacity - 1, out.size()); memcpy(text, out.data(), n); text[n] = 0; }
    return (int)out.size() + 1;
}
const char* tc2li_last_error(void) { return tc2li::g_last_error.c_str(); }
int tc2li_abi_version(void) { return 1; }
int tc2li_device_count(void) {
    int n = 0;
    tc2li::note_hip_touched();
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}
int tc2li_set_hardware_queues(int n) {
    if (n < 1 || n > 32) { tc2li::set_error("tc2li_set_hardware_queues: %d queues", n); return TC2LI_ERR_INVALID; }
    if (tc2li::g_hip_touched.load()) {
        tc2li::set_error("tc2li_set_hardware_queues: called after this library's first HIP call -- the runtime has read GPU_MAX_HW_QUEUES already");
        return TC2LI_ERR_INVALID;
    }
    char buf[16];
    snprintf(buf, sizeof(buf), "%d", n);
    setenv("GPU_MAX_HW_QUEUES", buf, 1);  // read by the HIP runtime when it initialises; setenv: call it before the process has other threads
    return TC2LI_OK;
}
int tc2li_set_host_thread_budget(int threads) {
    if (threads < 1) { tc2li::set_error("tc2li_set_host_thread_budget: %d threads", threads); return TC2LI_ERR_INVALID; }
    std::lock_guard<std::mutex> lk(tc2li::g_pools_mu);
    for (int i = 0; i < tc2li::kPoolCount; ++i)
        if (tc2li::g_pools[i]) { tc2li::set_error("tc2li_set_host_thread_budget: the library's worker pools exist already (call it first, or after tc2li_shutdown)"); return TC2LI_ERR_INVALID; }
    tc2li::g_thread_budget.store(threads);
    return TC2LI_OK;
}
int tc2li_host_threads(int32_t* counts, int capacity) {
    const int ids[5] = {tc2li::kPoolGlobal, tc2li::kPoolTracking, tc2li::kPoolLidar, tc2li::kPoolBaGroup0, tc2li::kPoolBaTop};
    if (!counts || capacity < 6) { tc2li::set_error("tc2li_host_threads: invalid argument"); return TC2LI_ERR_INVALID; }
    counts[0] = tc2li::host_thread_budget();
    for (int i = 0; i < 5; ++i) counts[1 + i] = tc2li::pool_threads(ids[i]);
    return 6;
}
int tc2li_shutdown(void) {
    // the caller guarantees that no other thread is inside the library; the pools' threads are idle then
    tc2li::WorkerPool* pools[tc2li::kPoolCount];
    {
        std::lock_guard<std::mutex> lk(tc2li::g_pools_mu);
        for (int i = 0; i < tc2li::kPoolCount; ++i) { pools[i] = tc2li::g_pools[i]; tc2li::g_pools[i] = nullptr; }
    }
    for (int i = 0; i < tc2li::kPoolCount; ++i) delete pools[i];  // joins: every worker's thread_local work spaces are freed by its own exit
    {
        std::vector<std::function<void()>> todo;
        { std::lock_guard<std::mutex> lk(tc2li::g_hooks_mu); todo = tc2li::hooks(); }
        for (auto& fn : todo) fn();  // the process-wide work spaces (lock-step BA contexts, mapping / map-point work spaces)
    }
    if (tc2li::g_hip_touched.load()) {
        int n = 0;
        if (hipGetDeviceCount(&n) == hipSuccess && n > 0) (void)hipDeviceSynchronize(); else (void)hipGetLastError();
    }
    return TC2LI_OK;
}
}
