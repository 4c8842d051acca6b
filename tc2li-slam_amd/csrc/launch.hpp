// Kernel launch macro of the library: hipLaunchKernelGGL plus the optional per-launch event pair of tc2li_profile_* (measurement).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>

namespace tc2li {
// ---- per-launch timing (measurement only; include/tc2li_hip.h "tc2li_profile_*") ------------------------------------------------
// While enabled, every kernel launch of the library is bracketed by two HIP events on the stream it is launched on; the report sums
// the event spans per kernel name.  Off (the default) it costs one relaxed atomic load per launch.
namespace prof {
extern std::atomic<int> g_enabled;
struct Scope {
    int slot = -1;
    hipStream_t st;
    Scope(const char* name, hipStream_t stream) : st(stream) { if (g_enabled.load(std::memory_order_relaxed)) begin(name); }
    ~Scope() { if (slot >= 0) end(); }
    void begin(const char* name);
    void end();
};
}  // namespace prof
#define TC2LI_LAUNCH(kernel, grid, block, shmem, stream, ...)                    \
    do {                                                                         \
        ::tc2li::prof::Scope prof_scope_(#kernel, stream);                       \
        hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);     \
    } while (0)

}  // namespace tc2li
