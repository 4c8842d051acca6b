// RCCL behind the tc2li_allreduce_fn of a sharded bundle-adjustment window (include/tc2li_hip.h): one process per GPU, the
// communicator made from a unique id the launcher hands round.  librccl is opened on first use -- the library itself carries
// no link-time dependency on it, and inside a PyTorch process the copy PyTorch already loaded is the one that answers.
// Only the four entry points used are declared here (rccl.h: ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclAllReduce).
#include <dlfcn.h>

#include <cstring>
#include <mutex>

#include "common.hpp"

using namespace tc2li;

namespace {

struct UniqueId { char internal[128]; };  // NCCL_UNIQUE_ID_BYTES
constexpr int kSum = 0, kMax = 2, kFloat64 = 8;  // ncclRedOp_t / ncclDataType_t values of rccl.h

struct Rccl {
    void* lib = nullptr;
    int (*get_unique_id)(UniqueId*) = nullptr;
    int (*comm_init_rank)(void**, int, UniqueId, int) = nullptr;
    int (*comm_destroy)(void*) = nullptr;
    int (*all_reduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    const char* (*error_string)(int) = nullptr;
    bool ok = false;
};

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.lib) break;
        }
        if (!r.lib) return;
        r.get_unique_id = (decltype(r.get_unique_id))dlsym(r.lib, "ncclGetUniqueId");
        r.comm_init_rank = (decltype(r.comm_init_rank))dlsym(r.lib, "ncclCommInitRank");
        r.comm_destroy = (decltype(r.comm_destroy))dlsym(r.lib, "ncclCommDestroy");
        r.all_reduce = (decltype(r.all_reduce))dlsym(r.lib, "ncclAllReduce");
        r.error_string = (decltype(r.error_string))dlsym(r.lib, "ncclGetErrorString");
        r.ok = r.get_unique_id && r.comm_init_rank && r.comm_destroy && r.all_reduce;
    });
    return r;
}

int fail(const char* what, int code) {
    Rccl& r = rccl();
    set_error("%s failed: %s (%d)", what, r.error_string ? r.error_string(code) : "rccl error", code);
    return TC2LI_ERR_COMM;
}

bool ready() {
    if (rccl().ok) return true;
    set_error("librccl could not be loaded (%s)", dlerror() ? "dlopen/dlsym failed" : "missing symbols");
    return false;
}

}  // namespace

extern "C" {

int tc2li_rccl_unique_id(void* unique_id_128) {
    if (!unique_id_128) { set_error("tc2li_rccl_unique_id: invalid argument"); return TC2LI_ERR_INVALID; }
    if (!ready()) return TC2LI_ERR_COMM;
    UniqueId id;
    if (int rc = rccl().get_unique_id(&id)) return fail("ncclGetUniqueId", rc);
    memcpy(unique_id_128, &id, sizeof(id));
    return TC2LI_OK;
}

int tc2li_rccl_comm_create(const void* unique_id_128, int rank, int world, void** comm) {
    if (!unique_id_128 || !comm || world < 1 || rank < 0 || rank >= world) { set_error("tc2li_rccl_comm_create: invalid argument"); return TC2LI_ERR_INVALID; }
    if (!device_ready()) return TC2LI_ERR_NO_DEVICE;
    if (!ready()) return TC2LI_ERR_COMM;
    UniqueId id;
    memcpy(&id, unique_id_128, sizeof(id));
    *comm = nullptr;
    if (int rc = rccl().comm_init_rank(comm, world, id, rank)) return fail("ncclCommInitRank", rc);
    return TC2LI_OK;
}

int tc2li_rccl_comm_destroy(void* comm) {
    if (!comm) return TC2LI_OK;
    if (!ready()) return TC2LI_ERR_COMM;
    if (int rc = rccl().comm_destroy(comm)) return fail("ncclCommDestroy", rc);
    return TC2LI_OK;
}

int tc2li_rccl_allreduce(void* comm, double* device_buf, size_t count, int op, void* stream) {
    if (!comm || (!device_buf && count) || (op != TC2LI_REDUCE_SUM && op != TC2LI_REDUCE_MAX)) { set_error("tc2li_rccl_allreduce: invalid argument"); return TC2LI_ERR_INVALID; }
    if (count == 0) return TC2LI_OK;
    if (!ready()) return TC2LI_ERR_COMM;
    if (int rc = rccl().all_reduce(device_buf, device_buf, count, kFloat64, op == TC2LI_REDUCE_SUM ? kSum : kMax, comm, (hipStream_t)stream))
        return fail("ncclAllReduce", rc);
    return TC2LI_OK;
}

}  // extern "C"
