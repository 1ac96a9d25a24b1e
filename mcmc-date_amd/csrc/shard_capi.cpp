// shard_capi.cpp -- the one exchange step of the path across GPUs: an all-gather of per-chain values (the ln
// posterior the MC3 swap phase needs -- `mc3 (MC3Settings (NChains 4) (SwapPeriod 2) (NSwaps 3))`, app/Main.hs:476-478 -- and
// diagnostics) over the ranks' contiguous chain shards, as RCCL's ncclAllGather over xGMI.  The values are a few KB per rank:
// latency bound, hence ONE direct all-gather per swap period and nothing else.
//
// RCCL is bound at run time (dlopen): a single-GPU user of the library does not need it, and a host written in the
// reference's language needs no RCCL binding of its own -- the communicator is created through the three thin wrappers below
// (the unique id travels between the ranks' processes by whatever means the host has: a file, MPI, a socket).
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cstdio>
#include <cstring>
#include <mutex>

#include "../../include/mcmcdate_mvn.h"

extern "C" int mcd_set_last_error_(int code, const char* msg);

namespace {

// the part of rccl.h that is used (ROCm 7.2: rccl/rccl.h:40-43, 187, 220, 260, 467, 678)
struct NcclUniqueId {
    char internal[128];
};
typedef void* NcclComm;
typedef int (*GetUniqueIdFn)(NcclUniqueId*);
typedef int (*CommInitRankFn)(NcclComm*, int, NcclUniqueId, int);
typedef int (*CommDestroyFn)(NcclComm);
typedef int (*AllGatherFn)(const void*, void*, size_t, int, NcclComm, hipStream_t);
typedef int (*CommCountFn)(NcclComm, int*);
typedef const char* (*GetErrorStringFn)(int);
constexpr int kNcclDouble = 8;

struct Rccl {
    void* lib = nullptr;
    GetUniqueIdFn get_unique_id = nullptr;
    CommInitRankFn comm_init_rank = nullptr;
    CommDestroyFn comm_destroy = nullptr;
    AllGatherFn all_gather = nullptr;
    CommCountFn comm_count = nullptr;
    GetErrorStringFn error_string = nullptr;
    bool ok = false;
};

Rccl& rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char* n : names) {
            r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (r.lib) break;
        }
        if (!r.lib) return;
        r.get_unique_id = (GetUniqueIdFn)dlsym(r.lib, "ncclGetUniqueId");
        r.comm_init_rank = (CommInitRankFn)dlsym(r.lib, "ncclCommInitRank");
        r.comm_destroy = (CommDestroyFn)dlsym(r.lib, "ncclCommDestroy");
        r.all_gather = (AllGatherFn)dlsym(r.lib, "ncclAllGather");
        r.comm_count = (CommCountFn)dlsym(r.lib, "ncclCommCount");
        r.error_string = (GetErrorStringFn)dlsym(r.lib, "ncclGetErrorString");
        r.ok = r.get_unique_id && r.comm_init_rank && r.comm_destroy && r.all_gather;
    });
    return r;
}

int fail(int code, const char* what, int nccl_rc = 0)
{
    char buf[256];
    if (nccl_rc != 0 && rccl().error_string)
        snprintf(buf, sizeof buf, "%s: %s", what, rccl().error_string(nccl_rc));
    else
        snprintf(buf, sizeof buf, "%s", what);
    return mcd_set_last_error_(code, buf);
}

}  // namespace

extern "C" {

int mcd_shard_unique_id(char id[MCD_SHARD_ID_BYTES])
{
    if (!id) return fail(MCD_ERR_INVALID_ARG, "mcd_shard_unique_id: NULL argument");
    if (!rccl().ok) return fail(MCD_ERR_UNSUPPORTED, "mcd_shard_unique_id: librccl.so could not be loaded");
    NcclUniqueId u;
    if (int rc = rccl().get_unique_id(&u)) return fail(MCD_ERR_HIP, "ncclGetUniqueId", rc);
    static_assert(sizeof u.internal == MCD_SHARD_ID_BYTES, "unique id size");
    memcpy(id, u.internal, sizeof u.internal);
    return MCD_OK;
}

int mcd_shard_comm_create(void** comm, int world_size, int rank, const char id[MCD_SHARD_ID_BYTES], int device_id)
{
    if (!comm || !id) return fail(MCD_ERR_INVALID_ARG, "mcd_shard_comm_create: NULL argument");
    *comm = nullptr;
    if (world_size < 1 || rank < 0 || rank >= world_size) return fail(MCD_ERR_INVALID_ARG, "mcd_shard_comm_create: need 0 <= rank < world_size");
    if (!rccl().ok) return fail(MCD_ERR_UNSUPPORTED, "mcd_shard_comm_create: librccl.so could not be loaded");
    if (hipError_t e = hipSetDevice(device_id)) return fail(MCD_ERR_HIP, hipGetErrorString(e));
    NcclUniqueId u;
    memcpy(u.internal, id, sizeof u.internal);
    NcclComm c = nullptr;
    if (int rc = rccl().comm_init_rank(&c, world_size, u, rank)) return fail(MCD_ERR_HIP, "ncclCommInitRank", rc);
    *comm = c;
    return MCD_OK;
}

void mcd_shard_comm_destroy(void* comm)
{
    if (comm && rccl().ok) (void)rccl().comm_destroy((NcclComm)comm);
}

int mcd_shard_comm_count(void* comm, int* n_ranks)
{
    if (!comm || !n_ranks) return fail(MCD_ERR_INVALID_ARG, "mcd_shard_comm_count: NULL argument");
    if (!rccl().ok || !rccl().comm_count) return fail(MCD_ERR_UNSUPPORTED, "mcd_shard_comm_count: librccl.so could not be loaded");
    if (int rc = rccl().comm_count((NcclComm)comm, n_ranks)) return fail(MCD_ERR_HIP, "ncclCommCount", rc);
    return MCD_OK;
}

int mcd_shard_allgather(void* comm, const double* send, double* recv, int64_t count, void* stream)
{
    if (!comm || !send || !recv) return fail(MCD_ERR_INVALID_ARG, "mcd_shard_allgather: NULL argument");
    if (count < 0) return fail(MCD_ERR_INVALID_ARG, "mcd_shard_allgather: negative count");
    if (count == 0) return MCD_OK;
    if (!rccl().ok) return fail(MCD_ERR_UNSUPPORTED, "mcd_shard_allgather: librccl.so could not be loaded");
    if (int rc = rccl().all_gather(send, recv, (size_t)count, kNcclDouble, (NcclComm)comm, (hipStream_t)stream))
        return fail(MCD_ERR_HIP, "ncclAllGather", rc);
    return MCD_OK;
}

}  // extern "C"
