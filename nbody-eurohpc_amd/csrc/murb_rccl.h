// RCCL bound at run time (dlopen by soname): only the sharded / one-process-per-GPU modes need it, and a
// Python host may already have loaded its own build of it.  Included by murbhip.hip only.
#ifndef MURB_RCCL_H_
#define MURB_RCCL_H_

#include <dlfcn.h>

#include <cstdlib>
#include <string>
#include <hip/hip_runtime.h>

#include "../../include/murbhip.h"

namespace {

typedef struct { char internal[MURBHIP_UNIQUE_ID_BYTES]; } rccl_id_t;
typedef void* rccl_comm_t;
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(rccl_id_t*) = nullptr;
    int (*CommInitRank)(rccl_comm_t*, int, rccl_id_t, int) = nullptr;
    int (*CommInitAll)(rccl_comm_t*, int, const int*) = nullptr;
    int (*CommDestroy)(rccl_comm_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, rccl_comm_t, hipStream_t) = nullptr;
    int (*ReduceScatter)(const void*, void*, size_t, int, int, rccl_comm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void*, size_t, int, int, rccl_comm_t, hipStream_t) = nullptr;   // point-to-point form of the exchange
    int (*Recv)(void*, size_t, int, int, rccl_comm_t, hipStream_t) = nullptr;         // ("exchange_p2p"), optional
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false;
};
constexpr int kRcclFloat = 7;   // ncclFloat32
constexpr int kRcclSum = 0;     // ncclSum

Rccl& rccl()
{
    static Rccl r;
    if (r.lib || r.ok) return r;
    // MURBHIP_RCCL_LIBRARY: path of the library to bind instead (a site build of RCCL; the tests' mock).
    // RTLD_LOCAL for it: its ncclXxx symbols must not shadow those of an RCCL the host already loaded.
    // "none": behave as on a machine without RCCL (MURBHIP_E_NO_RCCL; callers fall back to peer copies).
    const char* override_path = std::getenv("MURBHIP_RCCL_LIBRARY");
    if (override_path && std::string(override_path) == "none") return r;
    if (override_path && *override_path) r.lib = dlopen(override_path, RTLD_NOW | RTLD_LOCAL);
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* nm : names) {
        if (r.lib) break;
        r.lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
    }
    if (!r.lib) return r;
    r.GetUniqueId = (int (*)(rccl_id_t*))dlsym(r.lib, "ncclGetUniqueId");
    r.CommInitRank = (int (*)(rccl_comm_t*, int, rccl_id_t, int))dlsym(r.lib, "ncclCommInitRank");
    r.CommInitAll = (int (*)(rccl_comm_t*, int, const int*))dlsym(r.lib, "ncclCommInitAll");
    r.CommDestroy = (int (*)(rccl_comm_t))dlsym(r.lib, "ncclCommDestroy");
    r.AllGather = (int (*)(const void*, void*, size_t, int, rccl_comm_t, hipStream_t))dlsym(r.lib, "ncclAllGather");
    r.ReduceScatter =
        (int (*)(const void*, void*, size_t, int, int, rccl_comm_t, hipStream_t))dlsym(r.lib, "ncclReduceScatter");
    r.GroupStart = (int (*)())dlsym(r.lib, "ncclGroupStart");
    r.GroupEnd = (int (*)())dlsym(r.lib, "ncclGroupEnd");
    r.Send = (int (*)(const void*, size_t, int, int, rccl_comm_t, hipStream_t))dlsym(r.lib, "ncclSend");
    r.Recv = (int (*)(void*, size_t, int, int, rccl_comm_t, hipStream_t))dlsym(r.lib, "ncclRecv");
    r.GetErrorString = (const char* (*)(int))dlsym(r.lib, "ncclGetErrorString");
    // ncclGroupStart/End are not needed for the collectives: every communicator is driven by a thread of its own (ShardCrew);
    // only the optional point-to-point exchange groups its sends and receives
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommInitAll && r.CommDestroy && r.AllGather && r.ReduceScatter;
    return r;
}

}  // namespace

#endif
