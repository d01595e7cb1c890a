// Prices the parts of the one-read multi-lambda pass (cluster_pass.hpp) - not part of the product library.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -o tools/cluster_bench tools/cluster_bench.hip && ./tools/cluster_bench [m] [n]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../fastoptsolver_amd/csrc/cluster_pass.hpp"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void fill(float* p, size_t n, unsigned seed) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned long long z = (i + 1) * 0x9E3779B97F4A7C15ull + seed * 0xD1B54A32D192ED03ull;
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27;
    p[i] = ((float)(z & 0xffff) - 32768.f) * (1.f / 32768.f);
  }
}

template <int CS, int MODE>
static void run(const char* what, const float* A, const float* b, int64_t m, int n, const float* xp, float* xchg, unsigned* flags,
                float* slabs, int* err, unsigned& epoch, int ncu) {
  auto kern = fos::cluster_pass_kernel<CS, MODE>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)fos::CP_LDS_BYTES));
  const int clusters = ncu / CS;
  int64_t lda = n, rpc = ((m + clusters - 1) / clusters + 15) / 16 * 16, n_stride = n;
  int n_pad = (n + 63) / 64 * 64, xa = 1;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    void* args[] = {&A, &lda, &b, &m, &n, &n_pad, &xp, &rpc, &xa, &xchg, &flags, &epoch, &slabs, &n_stride, &err};
    CK(hipEventRecord(e0, 0));
    CK(hipLaunchCooperativeKernel(reinterpret_cast<const void*>(kern), dim3(clusters * CS), dim3(fos::CP_THREADS), args,
                                  (unsigned)fos::CP_LDS_BYTES, 0));
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best;
    epoch += (unsigned)(rpc / 16) + 16;
  }
  int herr = 0;
  CK(hipMemcpy(&herr, err, sizeof(int), hipMemcpyDeviceToHost));
  const double panels = (double)rpc / 16;
  printf("%-46s %8.1f us  = %.2f us per panel, %.2f TB/s on one read of A%s\n", what, best * 1e3, best * 1e3 / panels,
         (double)m * n * 4 / (best * 1e-3) / 1e12, herr ? "  [TIMEOUT FLAG SET]" : "");
  fflush(stdout);
}

int main(int argc, char** argv) {
  const int64_t m = argc > 1 ? atoll(argv[1]) : 65536;
  const int n = argc > 2 ? atoi(argv[2]) : 8192;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  float *A, *b, *xp, *xchg, *slabs; unsigned* flags; int* err;
  CK(hipMalloc(&A, (size_t)m * n * 4)); CK(hipMalloc(&b, m * 4)); CK(hipMalloc(&xp, (size_t)(n + 64) * 16 * 4));
  CK(hipMalloc(&xchg, (size_t)ncu * fos::CP_SLOTS * 256 * 4)); CK(hipMalloc(&flags, (size_t)ncu * 128));
  CK(hipMalloc(&slabs, (size_t)ncu * 16 * n * 4)); CK(hipMalloc(&err, 4));
  CK(hipMemset(flags, 0, (size_t)ncu * 128)); CK(hipMemset(err, 0, 4));
  fill<<<1024, 256>>>(A, (size_t)m * n, 1); fill<<<64, 256>>>(b, m, 2); fill<<<64, 256>>>(xp, (size_t)(n + 64) * 16, 3);
  CK(hipDeviceSynchronize());
  unsigned epoch = 1;
  printf("m %lld n %d, %d CUs\n", (long long)m, n, ncu);
  if (n <= 8192) {
    run<8, 0>("full", A, b, m, n, xp, xchg, flags, slabs, err, epoch, ncu);
    run<8, 1>("no hand-off", A, b, m, n, xp, xchg, flags, slabs, err, epoch, ncu);
    run<8, 2>("no matrix-core work", A, b, m, n, xp, xchg, flags, slabs, err, epoch, ncu);
    run<8, 3>("no hand-off, no matrix-core work (loads only)", A, b, m, n, xp, xchg, flags, slabs, err, epoch, ncu);
    run<8, 4>("no loads of A", A, b, m, n, xp, xchg, flags, slabs, err, epoch, ncu);
    run<8, 5>("no loads, no hand-off (matrix cores + LDS)", A, b, m, n, xp, xchg, flags, slabs, err, epoch, ncu);
    run<8, 6>("no loads, no matrix-core work (hand-off only)", A, b, m, n, xp, xchg, flags, slabs, err, epoch, ncu);
    run<8, 0>("full", A, b, m, n, xp, xchg, flags, slabs, err, epoch, ncu);
  } else {
    run<16, 0>("full", A, b, m, n, xp, xchg, flags, slabs, err, epoch, ncu);
    run<16, 1>("no hand-off", A, b, m, n, xp, xchg, flags, slabs, err, epoch, ncu);
    run<16, 3>("no hand-off, no matrix-core work (loads only)", A, b, m, n, xp, xchg, flags, slabs, err, epoch, ncu);
    run<16, 5>("no loads, no hand-off (matrix cores + LDS)", A, b, m, n, xp, xchg, flags, slabs, err, epoch, ncu);
    run<16, 6>("no loads, no matrix-core work (hand-off only)", A, b, m, n, xp, xchg, flags, slabs, err, epoch, ncu);
  }
  return 0;
}
