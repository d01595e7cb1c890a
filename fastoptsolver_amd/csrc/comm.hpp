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

namespace fos {

// ---- one-shot full-mesh all-reduce ----------------------------------------------------------
// The exchange of a sharded iteration is n + 1 floats (64 KiB): a ring all-reduce spends 2(P-1) latency-bound steps on it.
// xGMI is a full mesh of point-to-point links, so every rank can instead WRITE its vector straight into an inbox on every
// peer (one hop, all links busy at once) and each rank sums the P inbox rows itself, in rank order - one kernel, one
// network latency, and bit-identical results on all ranks by construction (same numbers, same order).
//   inbox  [2 sets][P sources][cap bytes]   on every rank, IPC-mapped into the peers (hipIpcGetMemHandle)
//   flags  [2 sets][P sources][MESH_MAXWG]  sequence numbers: "source s has delivered slice w of all-reduce #seq"
// Workgroup w owns a slice of the vector: push the slice to all peers, fence (system scope), publish the flags, wait for
// the P flags of its own slice (bounded spin on the 100 MHz wall clock: a lost peer ends the kernel with an error flag
// instead of hanging the GPU), sum.  Two inbox sets alternate: a rank can be at most one all-reduce ahead of the slowest.
// Inboxes and flags are fine-grained allocations (coherent between devices while kernels run; round 2 had plain hipMalloc,
// which two processes on ONE GPU cannot tell apart but which is no architectural guarantee across xGMI): data travels as
// plain 16-byte stores behind a system-scope release fence, flags as system-scope release / acquire atomics.
constexpr int MESH_MAXWG = 32;
constexpr int MESH_MAXRANKS = 8;
constexpr int MESH_THREADS = 256;

struct MeshPeers {
  char* inbox[MESH_MAXRANKS];                 // inbox base of every rank (own entry: the local allocation)
  unsigned long long* flags[MESH_MAXRANKS];
};

template <typename T>
__global__ __launch_bounds__(MESH_THREADS) void mesh_allreduce_kernel(MeshPeers peers, int nranks, int rank, T* __restrict__ buf,
                                                                     long long count, unsigned long long seq,
                                                                     long long cap_elems, unsigned long long timeout_ticks,
                                                                     int* __restrict__ err) {
  const int set = (int)(seq & 1ull);
  const int w = blockIdx.x, tid = threadIdx.x;
  long long per = (count + gridDim.x - 1) / gridDim.x;
  per = (per + 3) & ~3ll;                     // slices start on 16-byte boundaries (floats and doubles)
  long long lo = (long long)w * per;
  if (lo > count) lo = count;
  long long hi = lo + per;
  if (hi > count) hi = count;
  // 1. push this slice into every rank's inbox row [set][rank].  The inboxes are FINE-GRAINED device memory
  // (hipExtMallocWithFlags(hipDeviceMallocFinegrained): coherent across devices during a kernel), so plain 16-byte stores
  // + a system-scope release fence before the flags are the architected way to hand the data over; slices start on
  // 16-byte boundaries of the inbox row, the vector path needs the caller's buffer aligned likewise.
  constexpr int VE = 16 / (int)sizeof(T);
  const bool vec = (reinterpret_cast<uintptr_t>(buf) & 15u) == 0 && (lo % VE) == 0;
  const long long nvec = vec ? (hi - lo) / VE : 0;
  for (int p = 0; p < nranks; ++p) {
    T* dst = reinterpret_cast<T*>(peers.inbox[p]) + ((long long)(set * nranks + rank)) * cap_elems;
    const uint4* s16 = reinterpret_cast<const uint4*>(buf + lo);
    uint4* d16 = reinterpret_cast<uint4*>(dst + lo);
    for (long long i = tid; i < nvec; i += MESH_THREADS) d16[i] = s16[i];
    for (long long i = lo + nvec * VE + tid; i < hi; i += MESH_THREADS)
      __hip_atomic_store(dst + i, buf[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __threadfence_system();
  __syncthreads();
  // 2. publish: "rank has delivered slice w of all-reduce #seq" to every rank
  if (tid < nranks)
    __hip_atomic_store(peers.flags[tid] + ((long long)(set * nranks + rank)) * MESH_MAXWG + w, seq, __ATOMIC_RELEASE,
                       __HIP_MEMORY_SCOPE_SYSTEM);
  // 3. wait for slice w of every source (bounded)
  __shared__ int bad;
  if (tid == 0) bad = 0;
  __syncthreads();
  if (tid < nranks) {
    const unsigned long long* f = peers.flags[rank] + ((long long)(set * nranks + tid)) * MESH_MAXWG + w;
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
      if (wall_clock64() - t0 > timeout_ticks) { bad = 1; break; }
      __builtin_amdgcn_s_sleep(8);
    }
  }
  __syncthreads();
  if (bad) {
    if (tid == 0) *err = 1;
    return;                                   // buf keeps this rank's partial: the host reports the failure
  }
  // 4. sum the P inbox rows of this slice in rank order (every thread acquires at system scope behind the pollers)
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  const T* rows = reinterpret_cast<const T*>(peers.inbox[rank]) + ((long long)set * nranks) * cap_elems;
  for (long long i = tid; i < nvec; i += MESH_THREADS) {
    T acc[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) acc[e] = (T)0;
    for (int src = 0; src < nranks; ++src) {
      const uint4 raw = reinterpret_cast<const uint4*>(rows + (long long)src * cap_elems + lo)[i];
      T v[VE];
      __builtin_memcpy(v, &raw, 16);
#pragma unroll
      for (int e = 0; e < VE; ++e) acc[e] += v[e];
    }
    uint4 o;
    __builtin_memcpy(&o, acc, 16);
    reinterpret_cast<uint4*>(buf + lo)[i] = o;
  }
  for (long long i = lo + nvec * VE + tid; i < hi; i += MESH_THREADS) {
    T s = (T)0;
    for (int src = 0; src < nranks; ++src)
      s += __hip_atomic_load(rows + (long long)src * cap_elems + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    buf[i] = s;
  }
}

}  // namespace fos

struct fos_comm {
  int nranks = 1, rank = 0;
  int kind = 0;                               // 0: RCCL, 1: full-mesh one-shot kernel
  ncclComm_t nccl = nullptr;
  const fos::RcclApi* api = nullptr;
  // mesh transport
  char* inbox = nullptr;                      // local [2][nranks][cap_bytes]
  unsigned long long* flags = nullptr;        // local [2][nranks][MESH_MAXWG]
  int* err = nullptr;                         // device flag raised by a timed-out wait
  bool fine_grained = false;                  // inbox / flags came from hipExtMallocWithFlags(hipDeviceMallocFinegrained)
  size_t cap_bytes = 0;
  unsigned long long seq = 0;
  fos::MeshPeers peers{};
  void* opened[2 * fos::MESH_MAXRANKS] = {};  // IPC mappings to close
  int n_opened = 0;
};
