// Does a matrix that fits the 256 MiB Infinity Cache run faster with TEMPORAL loads in the single-pass kernel?  (tuning aid)
//   hipcc -O3 --offload-arch=gfx950 -o tools/nt_bench tools/nt_bench.hip && ./tools/nt_bench
// For A of 16 ... 512 MiB at n = 2048 / 4096 / 8192: the product geometry with non-temporal (product) and temporal loads, 60
// back-to-back passes over the SAME matrix (a solver loop), best of 3.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../fastoptsolver_amd/csrc/gemv_pair.hpp"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef void (*Launch)(const float*, int64_t, const float*, int64_t, int, fos::YSource, int64_t, float*, double*, int, hipStream_t);
template <int THREADS, int K, int R, int MINW, int NBUF, bool NT>
void launch_variant(const float* A, int64_t lda, const float* b, int64_t m, int n, fos::YSource ys, int64_t rpw, float* slabs, double* rr,
                    int nwg, hipStream_t st) {
  hipLaunchKernelGGL((fos::gemv_pair_kernel<float, THREADS, K, R, NT, MINW, true, NBUF, false, false, false, float, false, false, false>),
                     dim3(nwg), dim3(THREADS), 0, st, A, lda, b, m, n, ys, rpw, slabs, rr, (double*)nullptr);
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  struct Geo { int n; Launch nt, tmp; const char* name; };
  const Geo geos[] = {
      {2048, launch_variant<256, 2, 4, 2, 2, true>, launch_variant<256, 2, 4, 2, 2, false>, "256x2 r4"},
      {4096, launch_variant<256, 4, 2, 2, 2, true>, launch_variant<256, 4, 2, 2, 2, false>, "256x4 r2"},
      {8192, launch_variant<512, 4, 1, 2, 3, true>, launch_variant<512, 4, 1, 2, 3, false>, "512x4 r1 3 tiles"},
  };
  for (const Geo& g : geos)
    for (int mib : {16, 32, 64, 96, 128, 192, 256, 512}) {
      const int n = g.n;
      const int64_t m = (int64_t)mib * 1048576 / (4 * n);
      float *A, *b, *y, *slabs;
      double* rr;
      CK(hipMalloc(&A, (size_t)m * n * 4));
      CK(hipMalloc(&b, (size_t)m * 4));
      CK(hipMalloc(&y, (size_t)n * 4));
      CK(hipMalloc(&slabs, (size_t)ncu * n * 4));
      CK(hipMalloc(&rr, ncu * sizeof(double)));
      CK(hipMemset(A, 0, (size_t)m * n * 4));
      CK(hipMemset(b, 0, (size_t)m * 4));
      CK(hipMemset(y, 0, (size_t)n * 4));
      fos::YSource ys{y, nullptr, nullptr, nullptr, nullptr, 0.0, nullptr};
      const int64_t rpw = (m + ncu - 1) / ncu;
      const int nwg = (int)((m + rpw - 1) / rpw);
      const double bytes = (double)m * n * 4;
      float res[2];
      int idx = 0;
      for (Launch fn : {g.nt, g.tmp}) {
        for (int i = 0; i < 5; ++i) fn(A, n, b, m, n, ys, rpw, slabs, rr, nwg, st);
        CK(hipStreamSynchronize(st));
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
          CK(hipEventRecord(e0, st));
          for (int i = 0; i < 60; ++i) fn(A, n, b, m, n, ys, rpw, slabs, rr, nwg, st);
          CK(hipEventRecord(e1, st));
          CK(hipEventSynchronize(e1));
          float ms;
          CK(hipEventElapsedTime(&ms, e0, e1));
          best = std::min(best, ms / 60);
        }
        res[idx++] = best;
      }
      printf("n %5d (%s)  A %4d MiB (m %7lld, %d wg): non-temporal %7.1f us %6.0f GB/s | temporal %7.1f us %6.0f GB/s | temporal/nt time %.3f\n",
             n, g.name, mib, (long long)m, nwg, res[0] * 1e3, bytes / (res[0] * 1e-3) / 1e9, res[1] * 1e3, bytes / (res[1] * 1e-3) / 1e9,
             res[1] / res[0]);
      fflush(stdout);
      CK(hipFree(A)); CK(hipFree(b)); CK(hipFree(y)); CK(hipFree(slabs)); CK(hipFree(rr));
    }
  return 0;
}
