// spl_comm.h -- RCCL bound at run time for the multi-GPU entry points of the C ABI (spl_comm_*,
// spl_allgather_slabs, spl_allgatherv_csr; include/splintr_hip.h).
//
// north_star: "large batches shard by document across the 8 GPUs of one node with an RCCL all-gatherv over
// xGMI to reassemble the ragged token-id output".  The reference has nothing distributed
// (src/core/tokenizer.rs:932-934 is a Rayon par_iter on one host), so there is no interface to mirror: the
// entry points are this build's own, one process per GPU, the communicator created from a 128-byte id the
// caller distributes by whatever it has (MPI, a file, torch's store).
//
// RCCL is resolved with dlopen / dlsym instead of a link-time dependency: a process that already holds a copy
// (PyTorch ships its own librccl.so with the same SONAME, librccl.so.1) must not get a second one, and a
// single-GPU user of the library needs no RCCL at all.
#pragma once
#include <dlfcn.h>
#include <stdlib.h>
#include <rccl/rccl.h>

#include <mutex>
#include <string>

namespace spl {

struct Rccl {
    void* lib = nullptr;
    std::string err;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;

    bool load() {
        if (lib) return true;
        err.clear();                                  // a failed attempt must not poison the next one
        // an already loaded copy first (same SONAME: the loader hands back the one the process has).
        // SPL_RCCL_LIB names ONE library to try instead (tests point it at a path that does not exist).
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        const char* only = getenv("SPL_RCCL_LIB");
        if (only && *only) {
            lib = dlopen(only, RTLD_NOW | RTLD_GLOBAL);
        } else {
            for (const char* n : names) {
                lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
                if (lib) break;
            }
        }
        if (!lib) {
            const char* e = dlerror();                // ONE call: dlerror() clears the message it returns
            err = std::string("librccl not found: ") + (e ? e : "");
            return false;
        }
        auto sym = [&](const char* s) -> void* {
            void* p = dlsym(lib, s);
            if (!p && err.empty()) err = std::string("librccl lacks ") + s;
            return p;
        };
        GetUniqueId = (decltype(GetUniqueId))sym("ncclGetUniqueId");
        CommInitRank = (decltype(CommInitRank))sym("ncclCommInitRank");
        CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
        GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
        AllGather = (decltype(AllGather))sym("ncclAllGather");
        Send = (decltype(Send))sym("ncclSend");
        Recv = (decltype(Recv))sym("ncclRecv");
        GroupStart = (decltype(GroupStart))sym("ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
        if (!err.empty()) { lib = nullptr; return false; }
        return true;
    }
};

inline Rccl& rccl() {
    static Rccl r;
    static std::mutex mu;
    std::lock_guard<std::mutex> g(mu);
    r.load();
    return r;
}

}  // namespace spl
