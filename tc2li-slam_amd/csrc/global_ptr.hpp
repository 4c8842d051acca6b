// Address space of pointers that kernels read out of records in memory.
#pragma once
#include <hip/hip_runtime.h>

namespace tc2li {

#if defined(__HIPCC__)
// A pointer a kernel reads out of a record in memory (a task table, a slot of the lock-step BA) could point anywhere as far as the
// compiler knows, so every access through it is a FLAT instruction: flat accesses count on the LDS counter as well as on the memory
// counter -- a wait for an LDS read then also waits for every load in flight (the Schur kernel's prefetch of the next slice ended at the
// first LDS wait of the task loop) -- and carry their 64-bit address per lane.  Our records only ever hold device or pinned host
// addresses: through this cast the accesses are global_load / global_store with the base in scalar registers, as in kernels that get the
// same pointers as arguments.  (Through the integer: a direct generic -> global -> generic cast pair is folded away before the
// address-space inference sees it.)
template <typename T>
__device__ __forceinline__ T* global_ptr(T* p) {
    typedef __attribute__((address_space(1))) T* global_t;
    return (T*)(global_t)(unsigned long long)p;
}
#define TC2LI_GLOBAL_FIELD(rec, f) (rec).f = ::tc2li::global_ptr((rec).f)
#endif

}  // namespace tc2li
