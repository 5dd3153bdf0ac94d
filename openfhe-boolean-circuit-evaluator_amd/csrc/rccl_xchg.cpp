// rccl_xchg.cpp -- the one collective of the multi-GPU path issued from inside the library: an RCCL all-gather of
// boundary ciphertexts on the ENGINE'S OWN HIP STREAM, between the pack kernel that fills the send buffer and the
// scatter kernels that consume the receive buffer.  No host synchronisation, no interpreter in the per-level loop
// (the first version called back into Python / torch.distributed after a hipStreamSynchronize, dist.py).
//
// RCCL is loaded at run time (dlopen), so the library has no link-time dependency on it: a process that already
// holds an RCCL (PyTorch's) shares it, a single-GPU host never needs it, and a missing library is reported as a
// status code + message by the call that wanted it.  Rendezvous stays outside: rank 0 obtains the 128-byte unique
// id (bce_rccl_unique_id) and the host program distributes it to the other ranks by whatever channel it has
// (torch.distributed broadcast in dist.py, MPI, a file).
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <cstring>
#include <string>
#include <type_traits>

#include "../../include/bce_circuit.h"
#include "../../include/bce_gpu.h"

extern "C" int bce_set_error(bce_ctx* c, int code, const char* msg);                  // engine.cpp
extern "C" hipStream_t bce_internal_stream(bce_ctx* c);                                // engine.cpp
extern "C" void** bce_internal_comm_slot(bce_ctx* c);                                  // engine.cpp: where the communicator lives
extern "C" int bce_internal_device(bce_ctx* c);                                        // engine.cpp: the context's HIP device

// Prototypes and constants come from the installed header (compile time); the LIBRARY is still bound at run time with
// dlopen / dlsym -- no link-time dependency, a single-GPU host never loads it.  decltype() of the header's declarations keeps
// every pointer type below in step with rccl.h, and the static_asserts pin the few ABI facts the call sites rely on.
#include <rccl/rccl.h>

namespace {

using GetUniqueIdFn = decltype(&ncclGetUniqueId);
using CommInitRankFn = decltype(&ncclCommInitRank);
using CommDestroyFn = decltype(&ncclCommDestroy);
using CommCountFn = decltype(&ncclCommCount);
using CommCuDeviceFn = decltype(&ncclCommCuDevice);
using CommUserRankFn = decltype(&ncclCommUserRank);
using AllGatherFn = decltype(&ncclAllGather);
using ErrStrFn = decltype(&ncclGetErrorString);
using GetVersionFn = decltype(&ncclGetVersion);
static_assert(sizeof(ncclUniqueId) == 128 && NCCL_UNIQUE_ID_BYTES == 128, "bce_rccl_unique_id hands out 128 bytes");
static_assert(ncclUint8 == 1, "payloads are sent as ncclUint8");
static_assert(ncclSuccess == 0, "status codes: 0 = success");
static_assert(std::is_same<ncclComm_t, ncclComm*>::value && sizeof(ncclComm_t) == sizeof(void*), "the context keeps the communicator in a void* slot");
static_assert(NCCL_MAJOR == 2, "written against the NCCL 2.x API of rccl.h");

struct Rccl {
    void* handle = nullptr;
    GetUniqueIdFn get_unique_id = nullptr;
    CommInitRankFn comm_init_rank = nullptr;
    CommDestroyFn comm_destroy = nullptr;
    CommCountFn comm_count = nullptr;
    CommCuDeviceFn comm_device = nullptr;
    CommUserRankFn comm_rank = nullptr;
    AllGatherFn all_gather = nullptr;
    ErrStrFn err_str = nullptr;
    int version = 0;   // ncclGetVersion of the library actually loaded
    std::string why;
};

Rccl& rccl() {
    static Rccl r = [] {
        Rccl x;
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            x.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (x.handle) break;
        }
        if (!x.handle) {
            const char* e = dlerror();   // one call: dlerror() clears the message it returns
            x.why = std::string("RCCL not found: ") + (e ? e : "dlopen failed");
            return x;
        }
        x.get_unique_id = (GetUniqueIdFn)dlsym(x.handle, "ncclGetUniqueId");
        x.comm_init_rank = (CommInitRankFn)dlsym(x.handle, "ncclCommInitRank");
        x.comm_destroy = (CommDestroyFn)dlsym(x.handle, "ncclCommDestroy");
        x.all_gather = (AllGatherFn)dlsym(x.handle, "ncclAllGather");
        x.comm_count = (CommCountFn)dlsym(x.handle, "ncclCommCount");
        x.comm_device = (CommCuDeviceFn)dlsym(x.handle, "ncclCommCuDevice");
        x.comm_rank = (CommUserRankFn)dlsym(x.handle, "ncclCommUserRank");
        x.err_str = (ErrStrFn)dlsym(x.handle, "ncclGetErrorString");
        GetVersionFn get_version = (GetVersionFn)dlsym(x.handle, "ncclGetVersion");
        if (!x.get_unique_id || !x.comm_init_rank || !x.comm_destroy || !x.all_gather || !x.err_str || !get_version || !x.comm_count ||
            !x.comm_device || !x.comm_rank) {
            x.why = "RCCL library lacks an expected entry point";
            dlclose(x.handle);
            x.handle = nullptr;
            return x;
        }
        // the prototypes are those of the header this file was compiled against (2.x): refuse a library that reports another
        // major version instead of calling into it with the wrong ABI
        if (get_version(&x.version) != ncclSuccess || x.version < 20000 || x.version >= 30000) {
            x.why = "RCCL library reports version " + std::to_string(x.version) + ": the in-library exchange is written against the 2.x ABI";
            dlclose(x.handle);
            x.handle = nullptr;
        }
        return x;
    }();
    return r;
}

int fail_nccl(bce_ctx* c, const char* what, ncclResult_t rc) {
    const std::string m = std::string(what) + ": " + (rccl().err_str ? rccl().err_str(rc) : "RCCL error");
    return bce_set_error(c, BCE_ERR_HIP, m.c_str());
}

}  // namespace

extern "C" {

int bce_rccl_available(void) { return rccl().handle ? 1 : 0; }
int bce_rccl_version(void) { return rccl().handle ? rccl().version : 0; }

int bce_rccl_unique_id(uint8_t out[128]) {
    if (!out) return BCE_ERR_ARG;
    Rccl& r = rccl();
    if (!r.handle) return BCE_ERR_UNSUPPORTED;
    ncclUniqueId id;
    if (r.get_unique_id(&id) != ncclSuccess) return BCE_ERR_HIP;
    std::memcpy(out, id.internal, 128);
    return BCE_OK;
}

int bce_rccl_init(bce_ctx* c, const uint8_t uid[128], int rank, int world) {
    if (!c || !uid || world < 1 || rank < 0 || rank >= world) return BCE_ERR_ARG;
    Rccl& r = rccl();
    if (!r.handle) return bce_set_error(c, BCE_ERR_UNSUPPORTED, r.why.c_str());
    void** slot = bce_internal_comm_slot(c);
    if (*slot) { r.comm_destroy(static_cast<ncclComm_t>(*slot)); *slot = nullptr; }
    ncclUniqueId id;
    std::memcpy(id.internal, uid, 128);
    if (hipSetDevice(bce_internal_device(c)) != hipSuccess) return bce_set_error(c, BCE_ERR_HIP, "hipSetDevice failed");
    ncclComm_t comm = nullptr;
    const ncclResult_t rc = r.comm_init_rank(&comm, world, id, rank);   // binds to the device current on this thread: the engine's
    if (rc != ncclSuccess) { *slot = nullptr; return fail_nccl(c, "ncclCommInitRank", rc); }
    *slot = comm;
    return BCE_OK;
}

int bce_rccl_allgather(bce_ctx* c, const void* dev_send, void* dev_recv, uint64_t bytes) {
    if (!c || !dev_send || !dev_recv) return BCE_ERR_ARG;
    void** slot = bce_internal_comm_slot(c);
    if (!*slot) return bce_set_error(c, BCE_ERR_STATE, "bce_rccl_allgather before bce_rccl_init");
    const ncclResult_t rc = rccl().all_gather(dev_send, dev_recv, (size_t)bytes, ncclUint8, static_cast<ncclComm_t>(*slot), bce_internal_stream(c));
    return rc == ncclSuccess ? BCE_OK : fail_nccl(c, "ncclAllGather", rc);
}

int bce_rccl_shutdown(bce_ctx* c) {
    if (!c) return BCE_ERR_ARG;
    void** slot = bce_internal_comm_slot(c);
    if (*slot && rccl().handle) rccl().comm_destroy(static_cast<ncclComm_t>(*slot));
    *slot = nullptr;
    return BCE_OK;
}

// what the communicator of this context says about itself: out[0] = ncclCommCount (ranks RCCL sees), out[1] = ncclCommUserRank,
// out[2] = ncclCommCuDevice (the HIP device it is bound to).  For bench lines and tests: proof of what the collective spans.
int bce_rccl_comm_info(bce_ctx* c, int out[3]) {
    if (!c || !out) return BCE_ERR_ARG;
    void** slot = bce_internal_comm_slot(c);
    if (!*slot) return bce_set_error(c, BCE_ERR_STATE, "bce_rccl_comm_info before bce_rccl_init");
    Rccl& r = rccl();
    ncclComm_t comm = static_cast<ncclComm_t>(*slot);
    ncclResult_t rc;
    if ((rc = r.comm_count(comm, &out[0])) != ncclSuccess) return fail_nccl(c, "ncclCommCount", rc);
    if ((rc = r.comm_rank(comm, &out[1])) != ncclSuccess) return fail_nccl(c, "ncclCommUserRank", rc);
    if ((rc = r.comm_device(comm, &out[2])) != ncclSuccess) return fail_nccl(c, "ncclCommCuDevice", rc);
    return BCE_OK;
}

}  // extern "C"
