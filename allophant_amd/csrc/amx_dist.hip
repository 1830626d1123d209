// Data-parallel gather of the C ABI (include/allophant_amx.h: amx_gather_outputs): the exchange step of utterance-level data
// parallelism for hosts that do not go through allophant_amd/parallel.py (torch.distributed) -- every rank sends the output
// block of its shard and its frame lengths to one root rank over an RCCL communicator the CALLER created (one process per
// GPU, ncclCommInitRank; RCCL moves the blocks over xGMI, each GPU -> root on its own link).
//
// The library is NOT linked against RCCL: the five entry points used here are looked up in the process at the first call
// (dlsym(RTLD_DEFAULT, ...), or in the library AMX_RCCL_LIBRARY names), i.e. in the very RCCL the caller's communicator belongs
// to -- a torch process carries its own bundled librccl, a C host links /opt/rocm/lib/librccl.so, and mixing two copies on one
// communicator is not an option.
#include "../../include/allophant_amx.h"

#include <dlfcn.h>
#include <stdlib.h>
#include <hip/hip_runtime.h>

#include <mutex>
#include <string>

namespace {

// the slice of rccl.h this file needs (ABI-stable since NCCL 2.7: ncclSend / ncclRecv)
typedef void* comm_t;
typedef int result_t;  // ncclResult_t, 0 = ncclSuccess
constexpr int kFloat32 = 7, kInt64 = 4;  // ncclFloat32, ncclInt64 (rccl.h:463-466)

struct Rccl {
    result_t (*group_start)() = nullptr;
    result_t (*group_end)() = nullptr;
    result_t (*send)(const void*, size_t, int, int, comm_t, hipStream_t) = nullptr;
    result_t (*recv)(void*, size_t, int, int, comm_t, hipStream_t) = nullptr;
    const char* (*error_string)(result_t) = nullptr;
    bool ok = false;
    std::string why;  // why `ok` is false
};

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // AMX_RCCL_LIBRARY names the RCCL the caller's communicators come from when the process holds more than one copy or
        // holds it outside the global symbol scope (a Python process: torch bundles its own librccl and loads it locally)
        void* where = RTLD_DEFAULT;
        if (const char* path = getenv("AMX_RCCL_LIBRARY")) {
            // an explicitly named library that does not load is an error, not a reason to bind whatever librccl the global
            // scope happens to hold: a second copy would be handed a communicator it did not create
            void* h = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
            if (!h) {
                const char* e = dlerror();
                r.why = std::string("AMX_RCCL_LIBRARY=") + path + " could not be loaded: " + (e ? e : "unknown dlopen error");
                return;
            }
            where = h;
        }
        r.group_start = (result_t (*)())dlsym(where, "ncclGroupStart");
        r.group_end = (result_t (*)())dlsym(where, "ncclGroupEnd");
        r.send = (result_t (*)(const void*, size_t, int, int, comm_t, hipStream_t))dlsym(where, "ncclSend");
        r.recv = (result_t (*)(void*, size_t, int, int, comm_t, hipStream_t))dlsym(where, "ncclRecv");
        r.error_string = (const char* (*)(result_t))dlsym(where, "ncclGetErrorString");
        r.ok = r.group_start && r.group_end && r.send && r.recv;
        if (!r.ok)
            r.why = "no RCCL in this process: amx_gather_outputs uses the ncclSend / ncclRecv of the library the caller's communicator "
                    "was created with (link or load librccl before the first call, or name it in AMX_RCCL_LIBRARY)";
    });
    return r;
}

thread_local std::string g_dist_error;

int fail(int code, const std::string& msg) {
    g_dist_error = msg;
    return code;
}

}  // namespace

extern "C" const char* amx_dist_last_error(void) { return g_dist_error.c_str(); }

extern "C" int amx_gather_outputs(void* nccl_comm, int rank, int world, int root, const float* send, int64_t count, float* recv,
                                  const int64_t* send_lengths, int n_local, int64_t* recv_lengths, void* stream) {
    if (!nccl_comm || world < 1 || rank < 0 || rank >= world || root < 0 || root >= world || count < 0 || n_local < 0)
        return fail(AMX_EINVAL, "bad gather arguments");
    if ((count > 0 && !send) || (n_local > 0 && !send_lengths)) return fail(AMX_EINVAL, "null send buffer");
    if (rank == root && ((count > 0 && !recv) || (n_local > 0 && !recv_lengths))) return fail(AMX_EINVAL, "the root rank needs receive buffers");
    Rccl& r = rccl();
    if (!r.ok) return fail(AMX_ESTATE, r.why);
    hipStream_t s = (hipStream_t)stream;
    auto check = [&](result_t rc, const char* what) {
        if (rc == 0) return AMX_OK;
        return fail(AMX_EHIP, std::string(what) + ": " + (r.error_string ? r.error_string(rc) : "RCCL error " + std::to_string(rc)));
    };
    // one group: the root posts a receive per rank (its own block included: a send to self inside a group is a copy), every
    // rank posts its two sends; equal block sizes on all ranks (equal shards -- the caller pads the last one)
    int rc = check(r.group_start(), "ncclGroupStart");
    if (rc) return rc;
    if (rank == root) {
        for (int peer = 0; peer < world && !rc; ++peer) {
            if (count > 0) rc = check(r.recv(recv + (int64_t)peer * count, (size_t)count, kFloat32, peer, nccl_comm, s), "ncclRecv");
            if (!rc && n_local > 0) rc = check(r.recv(recv_lengths + (int64_t)peer * n_local, (size_t)n_local, kInt64, peer, nccl_comm, s), "ncclRecv");
        }
    }
    if (!rc && count > 0) rc = check(r.send(send, (size_t)count, kFloat32, root, nccl_comm, s), "ncclSend");
    if (!rc && n_local > 0) rc = check(r.send(send_lengths, (size_t)n_local, kInt64, root, nccl_comm, s), "ncclSend");
    const int rc_end = check(r.group_end(), "ncclGroupEnd");
    return rc ? rc : rc_end;
}
