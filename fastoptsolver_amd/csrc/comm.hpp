// Communicator of a row-sharded problem: one process per GPU, the partial results of every pass over this rank's rows of
// A (gradient + ||r||^2, residual norms of trial points, the L-BFGS fg pair) are summed over the ranks ON THE HANDLE'S
// STREAM, between the kernels that produce and consume them - a sharded fos_fista_run enqueues `iters` iterations
// without host involvement, like the single-GPU one (SURVEY.md 8b "fos_mg_*", 8e).
//
// Transport: RCCL (ncclAllReduce over xGMI).  The library is resolved at run time with dlopen - the process normally has
// one loaded already (torch's), and that copy shares the HIP runtime that owns the caller's streams and pointers; a CPU
// box without RCCL can still load libfos_hip.so.  The reference has no distributed code; nothing here restates it.
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <mutex>
#include <string>

namespace fos {

struct RcclApi {
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string origin;      // which library the symbols came from
};

// nullptr (and *err filled) when no RCCL can be found.  Thread-safe, resolved once.
inline const RcclApi* rccl_api(std::string* err) {
  static RcclApi api;
  static std::string failure;
  static std::once_flag once;
  std::call_once(once, [] {
    // a copy that is already in the process first (RTLD_NOLOAD), then the system one
    const char* names[] = {"librccl.so", "librccl.so.1"};
    void* h = nullptr;
    for (const char* nm : names)
      if ((h = dlopen(nm, RTLD_NOW | RTLD_NOLOAD))) { api.origin = std::string(nm) + " (already loaded)"; break; }
    if (!h) {
      const char* paths[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
      for (const char* nm : paths)
        if ((h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL))) { api.origin = nm; break; }
    }
    if (!h) { failure = "RCCL not found (dlopen librccl.so / librccl.so.1)"; return; }
    api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(dlsym(h, "ncclAllReduce"));
    api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllReduce || !api.GetErrorString)
      failure = "RCCL library lacks an expected symbol: " + api.origin;
  });
  if (!failure.empty()) {
    if (err) *err = failure;
    return nullptr;
  }
  return &api;
}

}  // namespace fos

struct fos_comm {
  int nranks = 1, rank = 0;
  ncclComm_t nccl = nullptr;
  const fos::RcclApi* api = nullptr;
};
