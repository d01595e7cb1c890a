// Ragged tall-skinny pass (n = 5, LDS-staged copy): cached vs non-temporal loads of the staged block, workgroups per CU.
//   hipcc -O3 --offload-arch=gfx950 -o tools/tall_bench tools/tall_bench.hip && ./tools/tall_bench [m] [n]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "../fastoptsolver_amd/csrc/gemv_tall.hpp"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
__global__ void fill(float* p, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += st) p[i] = (float)((i * 2654435761u) & 0xffff) / 65536.f - 0.5f;
}
template <bool NT>
void launch(const float* A, const float* b, int64_t m, int n, fos::YSource ys, int64_t rpw, float* slabs, double* rr, int nwg, hipStream_t st) {
  hipLaunchKernelGGL((fos::gemv_tall_kernel<float, 8, fos::TL_STAGE4, true, false, float, NT>), dim3(nwg), dim3(fos::TL_THREADS), 0, st, A,
                     (int64_t)n, b, m, n, ys, rpw, slabs, rr, (double*)nullptr);
}
int main(int argc, char** argv) {
  int64_t m = argc > 1 ? atoll(argv[1]) : 32000000;
  int n = argc > 2 ? atoi(argv[2]) : 5;
  float *A, *b, *y, *slabs; double* rr;
  CK(hipMalloc(&A, (size_t)m * n * 4)); CK(hipMalloc(&b, (size_t)m * 4)); CK(hipMalloc(&y, 64 * 4));
  CK(hipMalloc(&slabs, (size_t)8192 * 8 * 4)); CK(hipMalloc(&rr, 8192 * 8));
  fill<<<4096, 256>>>(A, (size_t)m * n); fill<<<1024, 256>>>(b, (size_t)m); fill<<<1, 64>>>(y, 64);
  CK(hipDeviceSynchronize());
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  fos::YSource ys{y, nullptr, nullptr, nullptr, nullptr, 0.0, nullptr};
  const double bytes = (double)m * n * 4 + 4.0 * m;
  for (int round = 0; round < 2; ++round)
    for (int nt = 0; nt < 2; ++nt)
      for (int wpc : {2, 4, 6, 8}) {
        int nwg = 256 * wpc;
        int64_t rpw = ((m + nwg - 1) / nwg + 3) / 4 * 4;
        nwg = (int)((m + rpw - 1) / rpw);
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
          CK(hipEventRecord(e0, st));
          for (int i = 0; i < 10; ++i) { if (nt) launch<true>(A, b, m, n, ys, rpw, slabs, rr, nwg, st); else launch<false>(A, b, m, n, ys, rpw, slabs, rr, nwg, st); }
          CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
          float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms / 10);
        }
        printf("r%d nt=%d wg/cu %d nwg %5d : %8.1f us  %.0f GB/s (%.1f%% of 8 TB/s)\n", round, nt, wpc, nwg, best * 1e3, bytes / (best * 1e-3) / 1e9,
               bytes / (best * 1e-3) / 8e12 * 100);
        fflush(stdout);
      }
  return 0;
}
