// Placement / phase experiment for the single-pass kernel (not part of the product library).
//   hipcc -O3 --offload-arch=gfx950 -o tools/skew_bench tools/skew_bench.hip && ./tools/skew_bench [m] [n] [iters] [trials]
// Question (round 3): the same binary measures 9.55 ms or 10.0-10.6 ms per cfg4 pass depending on box AND on the process
// (fresh allocation).  The workgroups' row blocks start a power of two apart (cfg4: 256 MiB, cfg2: 8 MiB), so all 256 CUs
// read the same offset of their block at the same time.  Variants: contiguous blocks (product), non-power-of-two block
// sizes (fewer workgroups), rotated start per workgroup (SKEW), rows dealt round-robin (IL).  Every trial re-allocates A
// behind a pad buffer of a different size so that the physical placement changes.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
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
__global__ void reduce_slabs(const float* slabs, int nslabs, int n, float* g) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  double acc = 0.0;
  for (int s = 0; s < nslabs; ++s) acc += slabs[(size_t)s * n + j];
  g[j] = (float)acc;
}

typedef void (*Launch)(const void*, int64_t, const float*, int64_t, int, fos::YSource, int64_t, float*, double*, int, hipStream_t);
template <int THREADS, int K, int MINW, int NBUF, bool IL, bool DRAIN, bool SKEW>
void launch_variant(const void* A, int64_t lda, const float* b, int64_t m, int n, fos::YSource ys, int64_t rpw, float* slabs,
                    double* rr, int nwg, hipStream_t st) {
  hipLaunchKernelGGL((fos::gemv_pair_kernel<float, THREADS, K, 1, true, MINW, true, NBUF, IL, false, DRAIN, float, false, false, SKEW>),
                     dim3(nwg), dim3(THREADS), 0, st, (const float*)A, lda, b, m, n, ys, rpw, slabs, rr, (double*)nullptr);
}
struct Variant { std::string name; Launch fn; int nwg; };

int main(int argc, char** argv) {
  int64_t m = argc > 1 ? atoll(argv[1]) : 65536;
  int n = argc > 2 ? atoi(argv[2]) : 8192;
  int iters = argc > 3 ? atoi(argv[3]) : 20;
  int trials = argc > 4 ? atoi(argv[4]) : 4;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  printf("device %s  CUs %d  m %lld n %d  A %.2f GiB\n", prop.name, ncu, (long long)m, n, (double)m * n * 4 / (1 << 30));
  float *b, *y, *slabs, *g;
  double* rr;
  const int max_wg = 1024;
  CK(hipMalloc(&b, (size_t)m * 4));
  CK(hipMalloc(&y, (size_t)n * 4));
  CK(hipMalloc(&slabs, (size_t)max_wg * n * 4));
  CK(hipMalloc(&g, (size_t)n * 4));
  CK(hipMalloc(&rr, max_wg * sizeof(double)));
  fill_normal<<<256, 256>>>(b, (size_t)m, 2);
  fill_normal<<<32, 256>>>(y, (size_t)n, 3);
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  fos::YSource ys{y, nullptr, nullptr, nullptr, nullptr, 0.0, nullptr};

  std::vector<Variant> vs;
  if (n <= 8192) {
    vs.push_back({"t512k4 nb3 blocks      ", launch_variant<512, 4, 2, 3, false, false, false>, ncu});
    vs.push_back({"t512k4 nb3 IL          ", launch_variant<512, 4, 2, 3, true, false, false>, ncu});
    vs.push_back({"t512k4 nb2 DRAIN       ", launch_variant<512, 4, 2, 2, false, true, false>, ncu});
    vs.push_back({"t512k4 nb3 DRAIN       ", launch_variant<512, 4, 2, 3, false, true, false>, ncu});
    vs.push_back({"t1024k2 r1 nb2 DRAIN   ", launch_variant<1024, 2, 4, 2, false, true, false>, ncu});
    vs.push_back({"t1024k2 r1 nb3         ", launch_variant<1024, 2, 4, 3, false, false, false>, ncu});
    vs.push_back({"t1024k2 r1 nb3 DRAIN   ", launch_variant<1024, 2, 4, 3, false, true, false>, ncu});
    vs.push_back({"t1024k2 r1 nb4 DRAIN   ", launch_variant<1024, 2, 4, 4, false, true, false>, ncu});
  } else {
    vs.push_back({"t1024k4 DRAIN blocks      ", launch_variant<1024, 4, 4, 2, false, true, false>, ncu});
    vs.push_back({"t1024k4 DRAIN blocks 255wg", launch_variant<1024, 4, 4, 2, false, true, false>, ncu - 1});
    vs.push_back({"t1024k4 DRAIN blocks 248wg", launch_variant<1024, 4, 4, 2, false, true, false>, ncu - 8});
    vs.push_back({"t1024k4 DRAIN SKEW        ", launch_variant<1024, 4, 4, 2, false, true, true>, ncu});
    vs.push_back({"t1024k4 DRAIN IL          ", launch_variant<1024, 4, 4, 2, true, true, false>, ncu});
    vs.push_back({"t1024k4 nb2 blocks        ", launch_variant<1024, 4, 4, 2, false, false, false>, ncu});
    vs.push_back({"t1024k4 nb2 SKEW          ", launch_variant<1024, 4, 4, 2, false, false, true>, ncu});
    vs.push_back({"t512k8 nb2 blocks         ", launch_variant<512, 8, 2, 2, false, false, false>, ncu});
    vs.push_back({"t512k8 nb2 SKEW           ", launch_variant<512, 8, 2, 2, false, false, true>, ncu});
  }
  const double bytes = (double)m * n * 4;
  std::vector<float> href, hg(n);
  for (int trial = 0; trial < trials; ++trial) {
    void* pad = nullptr;
    const size_t pad_bytes = (size_t)trial * ((1ull << 30) + (37ull << 20) + 4096 * 3);
    if (pad_bytes) CK(hipMalloc(&pad, pad_bytes));
    float* A;
    CK(hipMalloc(&A, (size_t)m * n * 4));
    fill_normal<<<4096, 256, 0, st>>>(A, (size_t)m * n, 1);
    CK(hipStreamSynchronize(st));
    printf("-- trial %d: A at %p (pad %.2f GiB)\n", trial, (void*)A, pad_bytes / 1073741824.0);
    for (int round = 0; round < 2; ++round) {
      for (auto& v : vs) {
        int64_t rpw = (m + v.nwg - 1) / v.nwg;
        int nwg = (int)((m + rpw - 1) / rpw);
        v.fn(A, n, b, m, n, ys, rpw, slabs, rr, nwg, st);
        reduce_slabs<<<(n + 255) / 256, 256, 0, st>>>(slabs, nwg, n, g);
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(hg.data(), g, (size_t)n * 4, hipMemcpyDeviceToHost));
        if (href.empty()) href = hg;
        double err = 0, rn = 0;
        for (int j = 0; j < n; ++j) { err += ((double)hg[j] - href[j]) * ((double)hg[j] - href[j]); rn += (double)href[j] * href[j]; }
        err = std::sqrt(err / rn);
        float best = 1e30f, worst = 0;
        for (int rep = 0; rep < 3; ++rep) {
          CK(hipEventRecord(e0, st));
          for (int i = 0; i < iters; ++i) v.fn(A, n, b, m, n, ys, rpw, slabs, rr, nwg, st);
          CK(hipEventRecord(e1, st));
          CK(hipEventSynchronize(e1));
          float ms;
          CK(hipEventElapsedTime(&ms, e0, e1));
          best = std::min(best, ms / iters);
          worst = std::max(worst, ms / iters);
        }
        printf("  r%d %s nwg %4d : best %8.1f us  worst %8.1f us  %.0f GB/s (%.1f%%)  relerr-vs-first %.1e\n", round, v.name.c_str(), nwg,
               best * 1e3, worst * 1e3, bytes / (best * 1e-3) / 1e9, bytes / (best * 1e-3) / 8e12 * 100, err);
        fflush(stdout);
      }
    }
    CK(hipFree(A));
    if (pad) CK(hipFree(pad));
  }
  return 0;
}
