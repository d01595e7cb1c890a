// libfos_hip.so, translation unit 2 of 4 - communicators of sharded problems (include/fos.h "row-sharded problems"):
// RCCL via dlopen, the one-shot full-mesh kernel (comm.hpp), and the in-place sums every other unit calls.
#include "fos_internal.hpp"

namespace fosapi {

// In-place sum over the ranks of a communicator on `st`: RCCL, or the one-shot full-mesh kernel (comm.hpp).
int comm_allreduce(fos_comm* c, void* buf, size_t count, bool f64, hipStream_t st) {
  if (c->kind == 0) {
    const ncclResult_t r = c->api->AllReduce(buf, buf, count, f64 ? ncclDouble : ncclFloat, ncclSum, c->nccl, st);
    if (r != ncclSuccess) return fail(FOS_ERR_HIP, std::string("ncclAllReduce: ") + c->api->GetErrorString(r));
    return FOS_OK;
  }
  const size_t esz = f64 ? 8 : 4;
  if (count * esz > c->cap_bytes)
    return fail(FOS_ERR_ARG, "mesh all-reduce: message larger than the inbox rows the communicator was created with");
  if (count == 0) return FOS_OK;
  c->seq += 1;
  const int nwg = (int)std::max<size_t>(1, std::min<size_t>(fos::MESH_MAXWG, (count * esz + 4095) / 4096));
  const unsigned long long timeout = 100000000ull * 20ull;      // 20 s of the 100 MHz wall clock
  if (f64)
    hipLaunchKernelGGL(fos::mesh_allreduce_kernel<double>, dim3(nwg), dim3(fos::MESH_THREADS), 0, st, c->peers, c->nranks,
                       c->rank, (double*)buf, (long long)count, c->seq, (long long)(c->cap_bytes / 8), timeout, c->err);
  else
    hipLaunchKernelGGL(fos::mesh_allreduce_kernel<float>, dim3(nwg), dim3(fos::MESH_THREADS), 0, st, c->peers, c->nranks,
                       c->rank, (float*)buf, (long long)count, c->seq, (long long)(c->cap_bytes / 4), timeout, c->err);
  LAUNCH_CHECK();
  return FOS_OK;
}

// Sum `count` floats / doubles over the ranks of a sharded problem, in place, on the handle's stream (no-op otherwise).
int reduce_across(fos_problem* p, void* buf, size_t count, bool f64) {
  if (!p->comm) return FOS_OK;
  return comm_allreduce(p->comm, buf, count, f64, p->stream);
}

}  // namespace fosapi

using namespace fosapi;

extern "C" {

// ---- communicator (comm.hpp) ------------------------------------------------------------------------------------
int fos_comm_unique_id(char id[128]) {
  if (!id) return fail(FOS_ERR_ARG, "fos_comm_unique_id: null");
  std::string err;
  const fos::RcclApi* api = fos::rccl_api(&err);
  if (!api) return fail(FOS_ERR_UNSUPPORTED, err);
  ncclUniqueId uid;
  const ncclResult_t r = api->GetUniqueId(&uid);
  if (r != ncclSuccess) return fail(FOS_ERR_HIP, std::string("ncclGetUniqueId: ") + api->GetErrorString(r));
  static_assert(sizeof(uid) == 128, "ncclUniqueId is 128 bytes");
  std::memcpy(id, &uid, sizeof(uid));
  return FOS_OK;
}

int fos_comm_create(fos_comm** out, const char id[128], int nranks, int rank) {
  if (!out || !id || nranks < 1 || rank < 0 || rank >= nranks) return fail(FOS_ERR_ARG, "fos_comm_create: bad argument");
  std::string err;
  const fos::RcclApi* api = fos::rccl_api(&err);
  if (!api) return fail(FOS_ERR_UNSUPPORTED, err);
  ncclUniqueId uid;
  std::memcpy(&uid, id, sizeof(uid));
  fos_comm* c = new fos_comm();
  c->nranks = nranks; c->rank = rank; c->api = api;
  const ncclResult_t r = api->CommInitRank(&c->nccl, nranks, uid, rank);
  if (r != ncclSuccess) {
    delete c;
    return fail(FOS_ERR_HIP, std::string("ncclCommInitRank: ") + api->GetErrorString(r));
  }
  *out = c;
  return FOS_OK;
}

int fos_comm_destroy(fos_comm* c) {
  if (!c) return FOS_OK;
  if (c->nccl) (void)c->api->CommDestroy(c->nccl);
  for (int i = 0; i < c->n_opened; ++i) (void)hipIpcCloseMemHandle(c->opened[i]);
  void* bufs[] = {c->inbox, c->flags, c->err};
  for (void* q : bufs)
    if (q) (void)hipFree(q);
  delete c;
  return FOS_OK;
}

// ---- full-mesh transport: local allocation + IPC handles, then the peers' handles -----------------------------------
int fos_comm_mesh_create(fos_comm** out, int nranks, int rank, int64_t cap_bytes, char handles[128]) {
  if (!out || !handles || nranks < 1 || nranks > fos::MESH_MAXRANKS || rank < 0 || rank >= nranks || cap_bytes < 16)
    return fail(FOS_ERR_ARG, "fos_comm_mesh_create: bad argument (1..8 ranks)");
  fos_comm* c = new fos_comm();
  c->kind = 1; c->nranks = nranks; c->rank = rank;
  c->cap_bytes = ((size_t)cap_bytes + 63) & ~(size_t)63;
  const size_t inbox_bytes = 2 * (size_t)nranks * c->cap_bytes;
  const size_t flag_bytes = 2 * (size_t)nranks * fos::MESH_MAXWG * sizeof(unsigned long long);
  // fine-grained (inter-device coherent) memory for everything a PEER reads or writes while kernels run; a runtime that
  // refuses the flag for IPC-shared memory falls back to plain device memory (reported by fos_comm_mesh_info)
  hipError_t e = hipExtMallocWithFlags((void**)&c->inbox, inbox_bytes, hipDeviceMallocFinegrained);
  if (e == hipSuccess) e = hipExtMallocWithFlags((void**)&c->flags, flag_bytes, hipDeviceMallocFinegrained);
  c->fine_grained = (e == hipSuccess);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    if (c->inbox) (void)hipFree(c->inbox);
    c->inbox = nullptr; c->flags = nullptr;
    e = hipMalloc(&c->inbox, inbox_bytes);
    if (e == hipSuccess) e = hipMalloc(&c->flags, flag_bytes);
  }
  if (e == hipSuccess) e = hipMalloc(&c->err, sizeof(int));
  if (e == hipSuccess) e = hipMemset(c->inbox, 0, inbox_bytes);
  if (e == hipSuccess) e = hipMemset(c->flags, 0, flag_bytes);
  if (e == hipSuccess) e = hipMemset(c->err, 0, sizeof(int));
  hipIpcMemHandle_t h0, h1;
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "two IPC handles travel in 128 bytes");
  if (e == hipSuccess) e = hipIpcGetMemHandle(&h0, c->inbox);
  if (e == hipSuccess) e = hipIpcGetMemHandle(&h1, c->flags);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) {
    fos_comm_destroy(c);
    return fail(FOS_ERR_HIP, std::string("fos_comm_mesh_create: ") + hipGetErrorString(e));
  }
  std::memcpy(handles, &h0, 64);
  std::memcpy(handles + 64, &h1, 64);
  c->peers.inbox[rank] = c->inbox;
  c->peers.flags[rank] = c->flags;
  *out = c;
  return FOS_OK;
}

int fos_comm_mesh_connect(fos_comm* c, const char* all_handles) {
  if (!c || c->kind != 1 || !all_handles) return fail(FOS_ERR_ARG, "fos_comm_mesh_connect: bad argument");
  for (int p = 0; p < c->nranks; ++p) {
    if (p == c->rank) continue;
    hipIpcMemHandle_t h0, h1;
    std::memcpy(&h0, all_handles + (size_t)p * 128, 64);
    std::memcpy(&h1, all_handles + (size_t)p * 128 + 64, 64);
    void *a = nullptr, *b = nullptr;
    HIP_TRY(hipIpcOpenMemHandle(&a, h0, hipIpcMemLazyEnablePeerAccess));
    c->opened[c->n_opened++] = a;
    HIP_TRY(hipIpcOpenMemHandle(&b, h1, hipIpcMemLazyEnablePeerAccess));
    c->opened[c->n_opened++] = b;
    c->peers.inbox[p] = (char*)a;
    c->peers.flags[p] = (unsigned long long*)b;
  }
  return FOS_OK;
}

int fos_comm_check(fos_comm* c, void* stream) {
  if (!c) return fail(FOS_ERR_ARG, "fos_comm_check: null");
  if (c->kind != 1) return FOS_OK;
  int bad = 0;
  HIP_TRY(hipMemcpyAsync(&bad, c->err, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  if (bad) return fail(FOS_ERR_STATE, "mesh all-reduce: a peer did not deliver within the time limit; results are invalid");
  return FOS_OK;
}

int fos_comm_mesh_info(const fos_comm* c, int* fine_grained, int64_t* cap_bytes) {
  if (!c || c->kind != 1) return fail(FOS_ERR_ARG, "fos_comm_mesh_info: not a mesh communicator");
  if (fine_grained) *fine_grained = c->fine_grained ? 1 : 0;
  if (cap_bytes) *cap_bytes = (int64_t)c->cap_bytes;
  return FOS_OK;
}

int fos_comm_info(const fos_comm* c, int* nranks, int* rank) {
  if (!c) return fail(FOS_ERR_ARG, "fos_comm_info: null");
  if (nranks) *nranks = c->nranks;
  if (rank) *rank = c->rank;
  return FOS_OK;
}

const char* fos_comm_transport(void) {
  static std::string text;
  std::string err;
  const fos::RcclApi* api = fos::rccl_api(&err);
  text = api ? ("rccl: " + api->origin) : ("none: " + err);
  return text.c_str();
}

int fos_comm_allreduce(fos_comm* c, void* buf, int64_t count, int is_f64, void* stream) {
  if (!c || !buf || count < 0) return fail(FOS_ERR_ARG, "fos_comm_allreduce: bad argument");
  return comm_allreduce(c, buf, (size_t)count, is_f64 != 0, (hipStream_t)stream);
}

}  // extern "C"
