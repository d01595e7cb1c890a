// Which loads-only kernel is the fastest reader on this box?  (tuning aid for fos_stream_read_probe, not part of the library)
//   hipcc -O3 --offload-arch=gfx950 -o tools/read_probe_sweep tools/read_probe_sweep.hip && ./tools/read_probe_sweep [GiB]
// Variants: threads per workgroup x workgroups per CU x loads in flight per thread x order (grid-stride: consecutive waves read
// consecutive KiB / blocks: every workgroup streams its own contiguous range) x temporal / non-temporal loads.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool NT> __device__ inline f32x4 ld(const f32x4* p) {
  if constexpr (NT) return __builtin_nontemporal_load(p);
  else return *p;
}

template <int THREADS, int UNR, bool BLOCKS, bool NT>
__global__ __launch_bounds__(THREADS) void read_kernel(const f32x4* __restrict__ src, size_t n16, float* __restrict__ sink) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if constexpr (BLOCKS) {
    const size_t per = ((n16 + gridDim.x - 1) / gridDim.x + THREADS * UNR - 1) / (THREADS * UNR) * (THREADS * UNR);
    const size_t lo = per * blockIdx.x, hi = std::min(n16, lo + per);
    size_t i = lo + threadIdx.x;
    for (; i + (size_t)(UNR - 1) * THREADS < hi; i += (size_t)UNR * THREADS) {
      f32x4 v[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) v[u] = ld<NT>(src + i + (size_t)u * THREADS);
#pragma unroll
      for (int u = 0; u < UNR; ++u) acc += v[u];
    }
    for (; i < hi; i += THREADS) acc += ld<NT>(src + i);
  } else {
    const size_t stride = (size_t)gridDim.x * THREADS;
    size_t i = (size_t)blockIdx.x * THREADS + threadIdx.x;
    for (; i + (UNR - 1) * stride < n16; i += UNR * stride) {
      f32x4 v[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) v[u] = ld<NT>(src + i + u * stride);
#pragma unroll
      for (int u = 0; u < UNR; ++u) acc += v[u];
    }
    for (; i < n16; i += stride) acc += ld<NT>(src + i);
  }
  const float t = (acc.x + acc.y) + (acc.z + acc.w);
  if (t == 123.456f) sink[blockIdx.x] = t;       // keeps the loads alive without a store per thread
}

typedef void (*Fn)(const f32x4*, size_t, float*);
struct Variant { const char* name; Fn fn; int threads; };
#define V(T, U, B, N) {#T " thr, " #U " in flight, " #B " blocks, " #N " nt", read_kernel<T, U, B, N>, T}

int main(int argc, char** argv) {
  const double gib = argc > 1 ? atof(argv[1]) : 8.0;
  const size_t bytes = (size_t)(gib * 1073741824.0) / 16 * 16;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  f32x4* buf;
  float* sink;
  CK(hipMalloc(&buf, bytes));
  CK(hipMemset(buf, 0, bytes));
  CK(hipMalloc(&sink, 1 << 20));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  std::vector<Variant> vs = {
      V(256, 4, false, true),  V(256, 8, false, true),  V(256, 16, false, true), V(512, 4, false, true), V(512, 8, false, true),
      V(512, 16, false, true), V(1024, 4, false, true), V(1024, 8, false, true), V(512, 8, false, false), V(1024, 4, false, false),
      V(256, 8, true, true),   V(512, 4, true, true),   V(512, 8, true, true),   V(1024, 4, true, true), V(1024, 8, true, true),
      V(512, 8, true, false),  V(1024, 4, true, false), V(1024, 8, true, false),
      V(64, 8, true, true),    V(64, 16, true, true),   V(64, 8, false, true),   // single-wave workgroups (narrow-row geometries)
  };
  printf("device %s, %d CUs, buffer %.2f GiB\n", prop.name, ncu, bytes / 1073741824.0);
  const size_t n16 = bytes / 16;
  const int iters = std::max(3, (int)(4e10 / bytes));
  for (int round = 0; round < 2; ++round)
    for (auto& v : vs)
      for (int per_cu : {1, 2, 4, 8, 16, 32}) {
        if (v.threads * per_cu > 2048 || (per_cu > 8 && v.threads > 64)) continue;
        const int grid = ncu * per_cu;
        v.fn<<<grid, v.threads>>>(buf, n16, sink);
        CK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
          CK(hipEventRecord(e0, 0));
          for (int i = 0; i < iters; ++i) v.fn<<<grid, v.threads>>>(buf, n16, sink);
          CK(hipEventRecord(e1, 0));
          CK(hipEventSynchronize(e1));
          float ms;
          CK(hipEventElapsedTime(&ms, e0, e1));
          best = std::min(best, ms / iters);
        }
        printf("r%d %-44s x%d/CU : %9.1f us  %7.0f GB/s\n", round, v.name, per_cu, best * 1e3, bytes / (best * 1e-3) / 1e9);
        fflush(stdout);
      }
  return 0;
}
