// Gradient exchange of the data-parallel train step: RCCL all-reduce over xGMI behind the C-ABI.
//
// The reference is single-GPU and has no collective; the exchange points are after trainer.py:98 (D's gradients, before
// optimizer_d.step() at :100) and after trainer.py:111 (G's gradients, before :112).  One process per GPU; the
// communicator is bootstrapped from Python (rank 0 calls pg_comm_unique_id, the 128 bytes travel through the
// torch.distributed store / a gloo broadcast, every rank calls pg_comm_init_rank) and every exchange is ONE in-place
// ncclAllReduce(sum, fp32) of a span of the network's flat gradient buffer on the caller's stream.
//
// RCCL is bound at run time (dlopen), not at link time: a process that already holds an RCCL instance (PyTorch ships
// its own librccl.so without a SONAME) must not get a second one with its own transports, and the kernels of this
// library stay loadable on a box without RCCL.  Resolution order: an already-loaded "librccl.so" / "librccl.so.1",
// then a fresh load of either.  The table of entry points is immutable after the first successful resolution.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdint.h>
#include <string.h>
#include <mutex>
#include "pggan_hip.h"

namespace {

// the part of rccl.h this file needs (the header is not required to build the library)
struct RcclUniqueId { char internal[PG_COMM_ID_BYTES]; };
typedef void* rcclComm_t;
enum { RCCL_FLOAT32 = 7, RCCL_SUM = 0 };                     // ncclFloat32, ncclSum (rccl.h ncclDataType_t / ncclRedOp_t)

struct RcclApi {
    int (*GetVersion)(int*);
    int (*GetUniqueId)(RcclUniqueId*);
    int (*CommInitRank)(rcclComm_t*, int, RcclUniqueId, int);
    int (*CommDestroy)(rcclComm_t);
    int (*CommCount)(rcclComm_t, int*);
    int (*CommUserRank)(rcclComm_t, int*);
    int (*AllReduce)(const void*, void*, size_t, int, int, rcclComm_t, hipStream_t);
    bool ok;
};

RcclApi g_api;
std::once_flag g_once;

void resolve()
{
    void* h = nullptr;
    const char* names[] = {"librccl.so", "librccl.so.1"};
    for (int pass = 0; pass < 2 && !h; ++pass)
        for (const char* n : names) {
            h = dlopen(n, RTLD_NOW | RTLD_LOCAL | (pass == 0 ? RTLD_NOLOAD : 0));
            if (h) break;
        }
    g_api.ok = false;
    if (!h) return;
#define PG_SYM(field, name) *reinterpret_cast<void**>(&g_api.field) = dlsym(h, name); if (!g_api.field) return;
    PG_SYM(GetVersion, "ncclGetVersion")
    PG_SYM(GetUniqueId, "ncclGetUniqueId")
    PG_SYM(CommInitRank, "ncclCommInitRank")
    PG_SYM(CommDestroy, "ncclCommDestroy")
    PG_SYM(CommCount, "ncclCommCount")
    PG_SYM(CommUserRank, "ncclCommUserRank")
    PG_SYM(AllReduce, "ncclAllReduce")
#undef PG_SYM
    g_api.ok = true;
}

const RcclApi* api()
{
    std::call_once(g_once, resolve);
    return g_api.ok ? &g_api : nullptr;
}

inline int rccl_rc(int r) { return r == 0 ? 0 : PG_E_RCCL_BASE - r; }

}  // namespace

extern "C" int pg_rccl_version(int* version)
{
    const RcclApi* a = api();
    if (!a) return PG_E_NOLIB;
    if (!version) return PG_E_ARG;
    return rccl_rc(a->GetVersion(version));
}

extern "C" int pg_comm_unique_id(void* id_out)
{
    const RcclApi* a = api();
    if (!a) return PG_E_NOLIB;
    if (!id_out) return PG_E_ARG;
    RcclUniqueId id;
    const int r = a->GetUniqueId(&id);
    if (r == 0) memcpy(id_out, id.internal, PG_COMM_ID_BYTES);
    return rccl_rc(r);
}

extern "C" int pg_comm_init_rank(void** comm_out, int nranks, const void* id, int rank)
{
    const RcclApi* a = api();
    if (!a) return PG_E_NOLIB;
    if (!comm_out || !id || nranks <= 0 || rank < 0 || rank >= nranks) return PG_E_ARG;
    RcclUniqueId uid;
    memcpy(uid.internal, id, PG_COMM_ID_BYTES);
    rcclComm_t c = nullptr;
    const int r = a->CommInitRank(&c, nranks, uid, rank);
    if (r == 0) *comm_out = c;
    return rccl_rc(r);
}

extern "C" int pg_comm_destroy(void* comm)
{
    const RcclApi* a = api();
    if (!a) return PG_E_NOLIB;
    if (!comm) return PG_E_ARG;
    return rccl_rc(a->CommDestroy(comm));
}

extern "C" int pg_comm_info(void* comm, int* nranks, int* rank)
{
    const RcclApi* a = api();
    if (!a) return PG_E_NOLIB;
    if (!comm || !nranks || !rank) return PG_E_ARG;
    int r = a->CommCount(comm, nranks);
    if (r == 0) r = a->CommUserRank(comm, rank);
    return rccl_rc(r);
}

extern "C" int pg_allreduce_sum_f32(void* comm, float* buf, int64_t count, pg_stream_t stream)
{
    const RcclApi* a = api();
    if (!a) return PG_E_NOLIB;
    if (!comm || !buf || count < 0) return PG_E_ARG;
    if (count == 0) return 0;
    return rccl_rc(a->AllReduce(buf, buf, (size_t)count, RCCL_FLOAT32, RCCL_SUM, comm, (hipStream_t)stream));
}
