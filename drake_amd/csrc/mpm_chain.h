// Multi-GPU chain driven from the host library itself: ranks tile the domain along x (see
// DESIGN.md section 5), the halo buffers travel with RCCL point-to-point calls issued ON THE
// ENGINE'S STREAM.  Measured on this stack: a cross-stream dependency costs ~20 us each way, which
// is what a substep pays twice when the transfer is posted through torch.distributed (its NCCL
// work runs on a stream of its own); here pack -> send/recv -> add are stream-ordered with no
// event in between, and a batch of substeps is one host call.
//
// RCCL is bound at run time (dlopen) so that single-GPU users do not need it.
#pragma once
#include <dlfcn.h>

#include <cstring>

#include "mpm_host.h"

namespace rccl_rt {
constexpr int kIdBytes = 128;                       // NCCL_UNIQUE_ID_BYTES
struct UniqueId { char internal[kIdBytes]; };       // ncclUniqueId
using Comm = void*;                                 // ncclComm_t
using GetUniqueId = int (*)(UniqueId*);
using CommInitRank = int (*)(Comm*, int, UniqueId, int);
using CommDestroy = int (*)(Comm);
using GroupStart = int (*)();
using GroupEnd = int (*)();
using Send = int (*)(const void*, size_t, int, int, Comm, hipStream_t);
using Recv = int (*)(void*, size_t, int, int, Comm, hipStream_t);
using AllReduce = int (*)(const void*, void*, size_t, int, int, Comm, hipStream_t);
using GetErrorString = const char* (*)(int);
using GetLastError = const char* (*)(Comm);
struct Api {
    void* lib = nullptr;
    GetUniqueId get_unique_id = nullptr;
    CommInitRank comm_init_rank = nullptr;
    CommDestroy comm_destroy = nullptr;
    GroupStart group_start = nullptr;
    GroupEnd group_end = nullptr;
    Send send = nullptr;
    Recv recv = nullptr;
    AllReduce all_reduce = nullptr;
    GetErrorString error_string = nullptr;
    GetLastError last_error = nullptr;   // ncclGetLastError: RCCL's own words for what went wrong (optional symbol)
};
// The process may already hold an RCCL (PyTorch ships one): use that instance, never a second one --
// two copies of RCCL's dependencies (rocm_smi) in one process end in a double free at exit.  A process
// that will load PyTorch later names its copy in MPM_RCCL_LIBRARY (drake_amd/capi.py does).
static const Api* api() {
    static Api a;
    static bool tried = false;
    if (tried) return a.lib ? &a : nullptr;
    tried = true;
    const char* names[] = {"librccl.so.1", "librccl.so"};
    for (const char* n : names)
        if ((a.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
    if (!a.lib)
        if (const char* path = getenv("MPM_RCCL_LIBRARY")) a.lib = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
    if (!a.lib)
        for (const char* n : names)
            if ((a.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!a.lib) return nullptr;
    a.get_unique_id = (GetUniqueId)dlsym(a.lib, "ncclGetUniqueId");
    a.comm_init_rank = (CommInitRank)dlsym(a.lib, "ncclCommInitRank");
    a.comm_destroy = (CommDestroy)dlsym(a.lib, "ncclCommDestroy");
    a.group_start = (GroupStart)dlsym(a.lib, "ncclGroupStart");
    a.group_end = (GroupEnd)dlsym(a.lib, "ncclGroupEnd");
    a.send = (Send)dlsym(a.lib, "ncclSend");
    a.recv = (Recv)dlsym(a.lib, "ncclRecv");
    a.all_reduce = (AllReduce)dlsym(a.lib, "ncclAllReduce");
    a.error_string = (GetErrorString)dlsym(a.lib, "ncclGetErrorString");
    a.last_error = (GetLastError)dlsym(a.lib, "ncclGetLastError");
    if (!a.get_unique_id || !a.comm_init_rank || !a.comm_destroy || !a.group_start || !a.group_end || !a.send || !a.recv) {
        a.lib = nullptr;
        return nullptr;
    }
    return &a;
}
}  // namespace rccl_rt

#define RCCL_TRY(expr)                                                                              \
    do {                                                                                            \
        const int rc_ = (expr);                                                                     \
        if (rc_ != 0) {                                                                             \
            const rccl_rt::Api* a_ = rccl_rt::api();                                                \
            const char* why_ = a_ && a_->last_error ? a_->last_error(nullptr) : nullptr;            \
            return fail(MPM_ERR_HIP, std::string("RCCL: ") + #expr + ": " +                        \
                                         (a_ && a_->error_string ? a_->error_string(rc_) : "error") + \
                                         (why_ && *why_ ? std::string(" -- ") + why_ : std::string())); \
        }                                                                                           \
    } while (0)
