// Geometry experiment for the fp64-accumulating pass (the L-BFGS fg, fos_gemv_pair_dd) - not part of the product library.
//   hipcc -O3 --offload-arch=gfx950 -o tools/dd_bench tools/dd_bench.hip && ./tools/dd_bench <case> [iters]
//   case: tall1024 (1048576 x 1024 fp32), bf16wide (131072 x 16384 bf16), f32wide (131072 x 16384 fp32), cfg2 (65536 x 8192 fp32)
// Every variant is checked against the first one (relative difference of the reduced gradient).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../fastoptsolver_amd/csrc/gemv_pair.hpp"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void fill_normal(float* p, size_t n, unsigned seed) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    unsigned long long z = (i + 1) * 0x9E3779B97F4A7C15ull + seed * 0xD1B54A32D192ED03ull;
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31;
    float u1 = ((unsigned)(z & 0xffffffffu) + 1.0f) * 2.3283064e-10f;
    float u2 = (unsigned)(z >> 32) * 2.3283064e-10f;
    p[i] = sqrtf(-2.0f * logf(u1)) * cosf(6.2831853f * u2);
  }
}
__global__ void to_bf16(const float* in, unsigned short* out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    unsigned u = __float_as_uint(in[i]);
    out[i] = (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
  }
}
__global__ void to_f64(const float* in, double* out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (double)in[i] * 1.000000001;
}
__global__ void reduce_slabs(const double* slabs, int nslabs, int n, double* g) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  double acc = 0.0;
  for (int s = 0; s < nslabs; ++s) acc += slabs[(size_t)s * n + j];
  g[j] = acc;
}

typedef void (*Launch)(const void*, int64_t, const float*, int64_t, int, fos::YSource, int64_t, double*, double*, int, hipStream_t);
template <typename T, int THREADS, int K, int R, int MINW, int NBUF, bool IL, bool YLDS, bool KEEP = false>
void launch_dd(const void* A, int64_t lda, const float* b, int64_t m, int n, fos::YSource ys, int64_t rpw, double* slabs, double* rr,
               int nwg, hipStream_t st) {
  auto kern = fos::gemv_pair_kernel<T, THREADS, K, R, true, MINW, true, NBUF, IL, false, false, double, YLDS, false, false, KEEP>;
  constexpr size_t lds = YLDS ? (size_t)THREADS * K * fos::ElemTraits<T>::EPC * sizeof(double) : 0;
  if (lds > 65536) {
    static bool done = false;
    if (!done) { CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); done = true; }
  }
  hipLaunchKernelGGL(kern, dim3(nwg), dim3(THREADS), lds, st, (const T*)A, lda, b, m, n, ys, rpw, slabs, rr, (double*)nullptr);
}
struct Variant { std::string name; Launch fn; int wg_per_cu; };

int main(int argc, char** argv) {
  const std::string which = argc > 1 ? argv[1] : "tall1024";
  int iters = argc > 2 ? atoi(argv[2]) : 20;
  int64_t m; int n; bool bf16 = false;
  std::vector<Variant> vs;
  using fos::bf16_t;
  if (which == "tall1024") {
    m = 1048576; n = 1024;
    vs.push_back({"f32 t256 k1 r2 nb2 (product)", launch_dd<float, 256, 1, 2, 2, 2, false, false>, 2});
    vs.push_back({"f32 t256 k1 r4 nb2", launch_dd<float, 256, 1, 4, 2, 2, false, false>, 2});
    vs.push_back({"f32 t256 k1 r8 nb2", launch_dd<float, 256, 1, 8, 2, 2, false, false>, 2});
    vs.push_back({"f32 t256 k1 r4 nb3", launch_dd<float, 256, 1, 4, 2, 3, false, false>, 2});
    vs.push_back({"f32 t256 k1 r2 nb2 4wg/cu", launch_dd<float, 256, 1, 2, 2, 2, false, false>, 4});
    vs.push_back({"f32 t256 k1 r4 nb2 4wg/cu", launch_dd<float, 256, 1, 4, 2, 2, false, false>, 4});
    vs.push_back({"f32 t64 k4 r4 nb2 16wg/cu", launch_dd<float, 64, 4, 4, 2, 2, false, false>, 16});
    vs.push_back({"f32 t64 k4 r2 nb2 16wg/cu", launch_dd<float, 64, 4, 2, 2, 2, false, false>, 16});
    vs.push_back({"f32 t128 k2 r4 nb2 8wg/cu", launch_dd<float, 128, 2, 4, 2, 2, false, false>, 8});
  } else if (which == "bf16wide") {
    m = 131072; n = 16384; bf16 = true;
    vs.push_back({"bf16 t512 k4 r1 nb2 ylds (product blocks)", launch_dd<bf16_t, 512, 4, 1, 2, 2, false, true>, 1});
    vs.push_back({"bf16 t512 k4 r1 nb3 ylds IL", launch_dd<bf16_t, 512, 4, 1, 2, 3, true, true>, 1});
    vs.push_back({"bf16 t512 k4 r1 nb2 ylds IL", launch_dd<bf16_t, 512, 4, 1, 2, 2, true, true>, 1});
    vs.push_back({"bf16 t512 k4 r1 nb2 ylds KEEPCVT", launch_dd<bf16_t, 512, 4, 1, 2, 2, false, true, true>, 1});
    vs.push_back({"bf16 t512 k4 r1 nb2 ylds IL KEEPCVT", launch_dd<bf16_t, 512, 4, 1, 2, 2, true, true, true>, 1});
  } else if (which == "f32wide") {
    m = 131072; n = 16384;
    vs.push_back({"f32 t512 k8 r1 nb2 ylds (product)", launch_dd<float, 512, 8, 1, 2, 2, false, true>, 1});
    vs.push_back({"f32 t512 k8 r1 nb2 ylds IL", launch_dd<float, 512, 8, 1, 2, 2, true, true>, 1});
    vs.push_back({"f32 t512 k8 r1 nb2 ylds KEEPCVT", launch_dd<float, 512, 8, 1, 2, 2, false, true, true>, 1});
  } else if (which == "bf16mid") {
    m = 262144; n = 8192; bf16 = true;
    vs.push_back({"bf16 t256 k4 r1 nb3 ylds (product)", launch_dd<bf16_t, 256, 4, 1, 2, 3, false, true>, 2});
    vs.push_back({"bf16 t256 k4 r1 nb2 ylds KEEPCVT 1wg/cu", launch_dd<bf16_t, 256, 4, 1, 2, 2, false, true, true>, 1});
    vs.push_back({"bf16 t256 k4 r1 nb2 ylds KEEPCVT", launch_dd<bf16_t, 256, 4, 1, 2, 2, false, true, true>, 2});
    vs.push_back({"bf16 t512 k2 r1 nb3 ylds KEEPCVT", launch_dd<bf16_t, 512, 2, 1, 2, 3, false, true, true>, 1});
  } else if (which == "tall2048") {
    m = 524288; n = 2048;
    vs.push_back({"f32 t256 k2 r2 nb2 (product)", launch_dd<float, 256, 2, 2, 2, 2, false, false>, 2});
    vs.push_back({"f32 t256 k2 r2 nb2 4wg/cu", launch_dd<float, 256, 2, 2, 2, 2, false, false>, 4});
    vs.push_back({"f32 t256 k2 r4 nb2", launch_dd<float, 256, 2, 4, 2, 2, false, false>, 2});
    vs.push_back({"f32 t256 k2 r2 nb2 3wg/cu", launch_dd<float, 256, 2, 2, 2, 2, false, false>, 3});
  } else {
    m = 65536; n = 8192;
    vs.push_back({"f32 t512 k4 r1 nb3 (product)", launch_dd<float, 512, 4, 1, 2, 3, false, false>, 1});
    vs.push_back({"f32 t512 k4 r1 nb3 IL", launch_dd<float, 512, 4, 1, 2, 3, true, false>, 1});
    vs.push_back({"f32 t512 k4 r1 nb3 KEEPCVT", launch_dd<float, 512, 4, 1, 2, 3, false, false, true>, 1});
    vs.push_back({"f32 t512 k4 r1 nb3 ylds KEEPCVT", launch_dd<float, 512, 4, 1, 2, 3, false, true, true>, 1});
    vs.push_back({"f32 t512 k4 r1 nb2 ylds", launch_dd<float, 512, 4, 1, 2, 2, false, true>, 1});
  }
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  const int esz = bf16 ? 2 : 4;
  printf("device %s CUs %d  %s: m %lld n %d  A %.2f GiB\n", prop.name, ncu, which.c_str(), (long long)m, n, (double)m * n * esz / (1 << 30));
  float *A32, *b, *yf;
  double *yd, *slabs, *rr, *g;
  void* A;
  const int max_wg = 4096;
  CK(hipMalloc(&A32, (size_t)m * n * 4));
  CK(hipMalloc(&b, (size_t)m * 4));
  CK(hipMalloc(&yf, (size_t)n * 4));
  CK(hipMalloc(&yd, (size_t)n * 8));
  CK(hipMalloc(&slabs, (size_t)max_wg * n * 8));
  CK(hipMalloc(&g, (size_t)n * 8));
  CK(hipMalloc(&rr, max_wg * sizeof(double)));
  fill_normal<<<4096, 256>>>(A32, (size_t)m * n, 1);
  fill_normal<<<256, 256>>>(b, (size_t)m, 2);
  fill_normal<<<32, 256>>>(yf, (size_t)n, 3);
  to_f64<<<(n + 255) / 256, 256>>>(yf, yd, n);
  A = A32;
  if (bf16) {
    unsigned short* A16;
    CK(hipMalloc(&A16, (size_t)m * n * 2));
    to_bf16<<<4096, 256>>>(A32, A16, (size_t)m * n);
    CK(hipDeviceSynchronize());
    CK(hipFree(A32));
    A = A16;
  }
  CK(hipDeviceSynchronize());
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  fos::YSource ys{nullptr, nullptr, nullptr, nullptr, nullptr, 0.0, yd};
  const double bytes = (double)m * n * esz + 4.0 * m + 8.0 * n;
  std::vector<double> href, hg(n);
  for (int round = 0; round < 2; ++round)
    for (auto& v : vs) {
      int nwg = ncu * v.wg_per_cu;
      int64_t rpw = (m + nwg - 1) / nwg;
      nwg = (int)((m + rpw - 1) / rpw);
      v.fn(A, n, b, m, n, ys, rpw, slabs, rr, nwg, st);
      reduce_slabs<<<(n + 255) / 256, 256, 0, st>>>(slabs, nwg, n, g);
      if (hipStreamSynchronize(st) != hipSuccess) { printf("%s LAUNCH FAILED: %s\n", v.name.c_str(), hipGetErrorString(hipGetLastError())); continue; }
      CK(hipMemcpy(hg.data(), g, (size_t)n * 8, hipMemcpyDeviceToHost));
      if (href.empty()) href = hg;
      double err = 0, rn = 0;
      for (int j = 0; j < n; ++j) { err += (hg[j] - href[j]) * (hg[j] - href[j]); rn += href[j] * href[j]; }
      float best = 1e30f;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < iters; ++i) v.fn(A, n, b, m, n, ys, rpw, slabs, rr, nwg, st);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms / iters);
      }
      printf("  r%d %-44s nwg %4d : %8.1f us  %.0f GB/s (%.1f%% of 8 TB/s)  rel diff vs first %.1e\n", round, v.name.c_str(), nwg, best * 1e3,
             bytes / (best * 1e-3) / 1e9, bytes / (best * 1e-3) / 8e12 * 100, std::sqrt(err / rn));
      fflush(stdout);
    }
  return 0;
}
