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

#include "../../include/bce_circuit.h"
#include "../../include/bce_gpu.h"

extern "C" int bce_set_error(bce_ctx* c, int code, const char* msg);                  // engine.cpp
extern "C" hipStream_t bce_internal_stream(bce_ctx* c);                                // engine.cpp
extern "C" void** bce_internal_comm_slot(bce_ctx* c);                                  // engine.cpp: where the communicator lives
extern "C" int bce_internal_device(bce_ctx* c);                                        // engine.cpp: the context's HIP device

namespace {

// the few RCCL entry points used, with the signatures of rccl.h (ncclUniqueId is a 128-byte struct passed by value)
struct UniqueId { char internal[128]; };
using GetUniqueIdFn = int (*)(UniqueId*);
using CommInitRankFn = int (*)(void**, int, UniqueId, int);
using CommDestroyFn = int (*)(void*);
using AllGatherFn = int (*)(const void*, void*, size_t, int, void*, hipStream_t);
using ErrStrFn = const char* (*)(int);
using GetVersionFn = int (*)(int*);

struct Rccl {
    void* handle = nullptr;
    GetUniqueIdFn get_unique_id = nullptr;
    CommInitRankFn comm_init_rank = nullptr;
    CommDestroyFn comm_destroy = nullptr;
    AllGatherFn all_gather = nullptr;
    ErrStrFn err_str = nullptr;
    int version = 0;   // ncclGetVersion: the prototypes above are those of the 2.x ABI (rccl.h of ROCm 6 / 7)
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
        x.err_str = (ErrStrFn)dlsym(x.handle, "ncclGetErrorString");
        GetVersionFn get_version = (GetVersionFn)dlsym(x.handle, "ncclGetVersion");
        if (!x.get_unique_id || !x.comm_init_rank || !x.comm_destroy || !x.all_gather || !x.err_str || !get_version) {
            x.why = "RCCL library lacks an expected entry point";
            dlclose(x.handle);
            x.handle = nullptr;
            return x;
        }
        // the hand-declared prototypes are the NCCL 2.x ones (ncclUniqueId by value, ncclDataType_t ncclUint8 = 1): refuse
        // a library that reports another major version instead of calling into it with the wrong ABI
        if (get_version(&x.version) != 0 || x.version < 20000 || x.version >= 30000) {
            x.why = "RCCL library reports version " + std::to_string(x.version) + ": the in-library exchange is written against the 2.x ABI";
            dlclose(x.handle);
            x.handle = nullptr;
        }
        return x;
    }();
    return r;
}

int fail_nccl(bce_ctx* c, const char* what, int rc) {
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
    UniqueId id;
    if (r.get_unique_id(&id) != 0) return BCE_ERR_HIP;
    std::memcpy(out, id.internal, 128);
    return BCE_OK;
}

int bce_rccl_init(bce_ctx* c, const uint8_t uid[128], int rank, int world) {
    if (!c || !uid || world < 1 || rank < 0 || rank >= world) return BCE_ERR_ARG;
    Rccl& r = rccl();
    if (!r.handle) return bce_set_error(c, BCE_ERR_UNSUPPORTED, r.why.c_str());
    void** slot = bce_internal_comm_slot(c);
    if (*slot) { r.comm_destroy(*slot); *slot = nullptr; }
    UniqueId id;
    std::memcpy(id.internal, uid, 128);
    if (hipSetDevice(bce_internal_device(c)) != hipSuccess) return bce_set_error(c, BCE_ERR_HIP, "hipSetDevice failed");
    const int rc = r.comm_init_rank(slot, world, id, rank);   // binds to the device current on this thread: the engine's
    if (rc != 0) { *slot = nullptr; return fail_nccl(c, "ncclCommInitRank", rc); }
    return BCE_OK;
}

int bce_rccl_allgather(bce_ctx* c, const void* dev_send, void* dev_recv, uint64_t bytes) {
    if (!c || !dev_send || !dev_recv) return BCE_ERR_ARG;
    void** slot = bce_internal_comm_slot(c);
    if (!*slot) return bce_set_error(c, BCE_ERR_STATE, "bce_rccl_allgather before bce_rccl_init");
    const int rc = rccl().all_gather(dev_send, dev_recv, (size_t)bytes, /*ncclUint8*/ 1, *slot, bce_internal_stream(c));
    return rc == 0 ? BCE_OK : fail_nccl(c, "ncclAllGather", rc);
}

int bce_rccl_shutdown(bce_ctx* c) {
    if (!c) return BCE_ERR_ARG;
    void** slot = bce_internal_comm_slot(c);
    if (*slot && rccl().handle) rccl().comm_destroy(*slot);
    *slot = nullptr;
    return BCE_OK;
}

}  // extern "C"
