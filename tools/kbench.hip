// Tuning microbench for the single-pass GEMV-pair kernel (not part of the product library).
//   hipcc -O3 --offload-arch=gfx950 -o tools/kbench tools/kbench.hip && ./tools/kbench [m] [n] [iters]
// Sweeps template variants of fos::gemv_pair_kernel in ONE process (guide rule 24: interleaved rounds),
// checks every variant against the two-pass fallback kernels, prints GB/s on the algorithmic bytes m*n*4.
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
    // counter-based hash -> Box-Muller
    unsigned long long z = (i + 1) * 0x9E3779B97F4A7C15ull + seed * 0xD1B54A32D192ED03ull;
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31;
    float u1 = ((unsigned)(z & 0xffffffffu) + 1.0f) * 2.3283064e-10f;
    float u2 = (unsigned)(z >> 32) * 2.3283064e-10f;
    p[i] = sqrtf(-2.0f * logf(u1)) * cosf(6.2831853f * u2);
  }
}

// plain streaming read (sum) = the box's read-bandwidth ceiling for this access pattern
template <bool NT>
__global__ __launch_bounds__(256) void stream_sum(const fos::u32x4* __restrict__ p, size_t nvec, float* out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  float acc = 0.f;
  for (; i + 3 * stride < nvec; i += 4 * stride) {
    fos::u32x4 a = fos::load16<NT>(p + i), b = fos::load16<NT>(p + i + stride), c = fos::load16<NT>(p + i + 2 * stride),
               d = fos::load16<NT>(p + i + 3 * stride);
    acc += __uint_as_float(a.x ^ b.y ^ c.z ^ d.w);
  }
  for (; i < nvec; i += stride) acc += __uint_as_float(fos::load16<NT>(p + i).x);
  if (acc == 123.456f) out[0] = acc;
}

__global__ void reduce_slabs(const float* slabs, int nslabs, int n, float* g) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  float acc = 0.f;
  for (int s = 0; s < nslabs; ++s) acc += slabs[(size_t)s * n + j];
  g[j] = acc;
}

struct Variant {
  std::string name;
  int threads, k, r, epc;
  void (*launch)(const void*, int64_t, const float*, int64_t, int, fos::YSource, int64_t, float*, double*, int, hipStream_t);
};

template <typename T, int THREADS, int K, int R, bool NT, int MINW, int NBUF = 2, bool IL = false, bool DRAIN = false>
void launch_variant(const void* A, int64_t lda, const float* b, int64_t m, int n, fos::YSource ys, int64_t rpw,
                    float* slabs, double* rr, int nwg, hipStream_t st) {
  hipLaunchKernelGGL((fos::gemv_pair_kernel<T, THREADS, K, R, NT, MINW, true, NBUF, IL, false, DRAIN>), dim3(nwg), dim3(THREADS), 0, st,
                     (const T*)A, lda, b, m, n, ys, rpw, slabs, rr, (double*)nullptr);
}

__global__ void to_bf16(const float* in, unsigned short* out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    unsigned u = __float_as_uint(in[i]);
    out[i] = (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
  }
}

int main(int argc, char** argv) {
  int64_t m = argc > 1 ? atoll(argv[1]) : 65536;
  int n = argc > 2 ? atoi(argv[2]) : 8192;
  int iters = argc > 3 ? atoi(argv[3]) : 20;
  const bool bf16 = argc > 4 && std::string(argv[4]) == "bf16";
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("device %s  CUs %d  m %lld n %d  A %.2f GiB\n", prop.name, prop.multiProcessorCount, (long long)m, n,
         (double)m * n * 4 / (1 << 30));
  const int ncu = prop.multiProcessorCount;
  float *A, *b, *y, *slabs, *g, *gref;
  double* rvec;
  double* rr;
  const int max_wg = 4096;
  CK(hipMalloc(&A, (size_t)m * n * 4));
  CK(hipMalloc(&b, (size_t)m * 4));
  CK(hipMalloc(&y, (size_t)n * 4));
  CK(hipMalloc(&rvec, (size_t)m * 8));
  CK(hipMalloc(&slabs, (size_t)max_wg * n * 4));
  CK(hipMalloc(&g, (size_t)n * 4));
  CK(hipMalloc(&gref, (size_t)n * 4));
  CK(hipMalloc(&rr, max_wg * sizeof(double)));
  fill_normal<<<4096, 256>>>(A, (size_t)m * n, 1);
  fill_normal<<<256, 256>>>(b, (size_t)m, 2);
  fill_normal<<<32, 256>>>(y, (size_t)n, 3);
  CK(hipDeviceSynchronize());
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  fos::YSource ys{y, nullptr, nullptr, nullptr, nullptr, 0.0, nullptr};
  unsigned short* A16 = nullptr;
  if (bf16) {
    CK(hipMalloc(&A16, (size_t)m * n * 2));
    to_bf16<<<4096, 256, 0, st>>>(A, A16, (size_t)m * n);
    CK(hipStreamSynchronize(st));
  }
  const void* Aany = bf16 ? (const void*)A16 : (const void*)A;

  // reference gradient via the two-pass fallback
  {
    int chunks = 64;
    int64_t rpc = (m + chunks - 1) / chunks;
    if (bf16) {
      hipLaunchKernelGGL(fos::residual_rows_kernel<fos::bf16_t>, dim3(2048), dim3(256), 0, st, (const fos::bf16_t*)A16,
                         (int64_t)n, b, m, n, ys, rvec, rr);
      hipLaunchKernelGGL(fos::transpose_rows_kernel<fos::bf16_t>, dim3((n + 255) / 256, chunks), dim3(256), 0, st,
                         (const fos::bf16_t*)A16, (int64_t)n, m, n, rvec, (const int*)nullptr, rpc, slabs);
    } else {
      hipLaunchKernelGGL(fos::residual_rows_kernel<float>, dim3(2048), dim3(256), 0, st, A, (int64_t)n, b, m, n, ys, rvec, rr);
      hipLaunchKernelGGL(fos::transpose_rows_kernel<float>, dim3((n + 255) / 256, chunks), dim3(256), 0, st, A, (int64_t)n, m,
                         n, rvec, (const int*)nullptr, rpc, slabs);
    }
    reduce_slabs<<<(n + 255) / 256, 256, 0, st>>>(slabs, chunks, n, gref);
    CK(hipStreamSynchronize(st));
  }
  std::vector<float> href(n), hg(n);
  CK(hipMemcpy(href.data(), gref, (size_t)n * 4, hipMemcpyDeviceToHost));
  double refnorm = 0;
  for (float v : href) refnorm += (double)v * v;
  refnorm = std::sqrt(refnorm);

  const double bytes = (double)m * n * (bf16 ? 2 : 4);
  // stream ceiling
  for (int nt = 0; nt < 2; ++nt) {
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < iters; ++i) {
        if (nt) stream_sum<true><<<ncu * 8, 256, 0, st>>>((const fos::u32x4*)A, (size_t)m * n / 4, g);
        else stream_sum<false><<<ncu * 8, 256, 0, st>>>((const fos::u32x4*)A, (size_t)m * n / 4, g);
      }
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep) printf("stream_sum nt=%d : %.1f us  %.0f GB/s\n", nt, ms * 1e3 / iters, bytes / (ms / iters * 1e-3) / 1e9);
    }
  }

  std::vector<Variant> vs;
#define V(T_, K_, R_, NT_, W_) vs.push_back({"t" #T_ "_k" #K_ "_r" #R_ "_nt" #NT_ "_w" #W_, T_, K_, R_, 4, launch_variant<float, T_, K_, R_, NT_, W_>})
#define VB(T_, K_, R_, NT_, W_) vs.push_back({"bf16_t" #T_ "_k" #K_ "_r" #R_ "_nt" #NT_ "_w" #W_, T_, K_, R_, 8, launch_variant<fos::bf16_t, T_, K_, R_, NT_, W_>})
#define VX(T_, K_, R_, W_, NB_, IL_) vs.push_back({"t" #T_ "_k" #K_ "_r" #R_ "_nbuf" #NB_ "_il" #IL_, T_, K_, R_, 4, launch_variant<float, T_, K_, R_, true, W_, NB_, IL_>})
  if (!bf16 && argc > 5) {
    // experiment set: pipeline depth / rows per step (after the counted-vmcnt restructure)
    if (n <= 8192) {
      VX(512, 4, 2, 2, 2, false); VX(512, 4, 2, 2, 3, false); VX(512, 4, 1, 2, 3, false); VX(512, 4, 1, 2, 4, false);
      VX(512, 4, 1, 2, 2, false); VX(1024, 2, 2, 4, 2, false); VX(1024, 2, 1, 4, 3, false); VX(256, 8, 1, 2, 2, false);
      VX(512, 4, 4, 2, 2, false); VX(1024, 2, 1, 4, 4, false);
    } else {
#define VXD(T_, K_, R_, W_, NB_, IL_) vs.push_back({"t" #T_ "_k" #K_ "_r" #R_ "_nbuf" #NB_ "_il" #IL_ "_DRAIN", T_, K_, R_, 4, launch_variant<float, T_, K_, R_, true, W_, NB_, IL_, true>})
      VX(512, 8, 1, 2, 2, false); VXD(512, 8, 1, 2, 2, false); VX(1024, 4, 1, 4, 2, false); VXD(1024, 4, 1, 4, 2, false);
      VX(512, 8, 1, 2, 2, true); VXD(512, 8, 1, 2, 2, true);
    }
  } else if (bf16 && argc > 5) {
#define VBX(T_, K_, R_, W_, NB_, IL_) vs.push_back({"bf16_t" #T_ "_k" #K_ "_r" #R_ "_nbuf" #NB_ "_il" #IL_, T_, K_, R_, 8, launch_variant<fos::bf16_t, T_, K_, R_, true, W_, NB_, IL_>})
    VBX(512, 4, 1, 2, 2, false); VBX(512, 4, 1, 2, 3, false); VBX(1024, 2, 1, 4, 2, false); VBX(1024, 2, 1, 4, 3, false);
    VBX(256, 4, 1, 2, 3, false);
  } else if (bf16) {
    if (n <= 8192) { VB(256, 4, 2, true, 2); VB(512, 2, 2, true, 2); VB(512, 2, 4, true, 2); VB(1024, 1, 4, true, 4); VB(1024, 1, 2, true, 4); }
    else { VB(512, 4, 2, true, 2); VB(512, 4, 1, true, 2); VB(512, 4, 1, true, 4); VB(1024, 2, 2, true, 4); VB(1024, 2, 1, true, 4); VB(1024, 2, 4, true, 4); VB(512, 4, 2, false, 2); }
  } else if (n <= 8192) {
    V(256, 8, 1, false, 2); V(256, 8, 2, false, 2); V(256, 8, 2, true, 2); V(256, 8, 1, true, 3); V(256, 8, 1, true, 4);
    V(512, 4, 2, true, 2); V(512, 4, 2, true, 4); V(512, 4, 4, true, 2); V(512, 4, 4, false, 2); V(512, 4, 1, true, 4);
    V(1024, 2, 4, true, 4); V(1024, 2, 2, true, 4); V(1024, 2, 4, false, 4); V(1024, 2, 1, true, 4);
  } else {
    V(512, 8, 1, true, 2); V(512, 8, 2, true, 2); V(512, 8, 2, false, 2);
    V(1024, 4, 2, true, 4); V(1024, 4, 1, true, 4); V(1024, 4, 1, false, 4);
  }

  // workgroups per CU to try
  const int wpc_list[] = {1, 2, 4};
  for (auto& v : vs) {
    if ((int64_t)v.threads * v.k * v.epc < n) continue;
    for (int wpc : wpc_list) {
      if (wpc * v.threads > 2048) continue;
      int nwg = ncu * wpc;
      if (nwg > max_wg) continue;
      int64_t rpw = (m + nwg - 1) / nwg;
      // correctness
      v.launch(Aany, n, b, m, n, ys, rpw, slabs, rr, nwg, st);
      reduce_slabs<<<(n + 255) / 256, 256, 0, st>>>(slabs, nwg, n, g);
      if (hipStreamSynchronize(st) != hipSuccess) { printf("%s wpc %d LAUNCH FAILED\n", v.name.c_str(), wpc); (void)hipGetLastError(); continue; }
      CK(hipMemcpy(hg.data(), g, (size_t)n * 4, hipMemcpyDeviceToHost));
      double err = 0;
      for (int j = 0; j < n; ++j) err += ((double)hg[j] - href[j]) * ((double)hg[j] - href[j]);
      err = std::sqrt(err) / refnorm;
      float best = 1e30f, tot = 0;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < iters; ++i) v.launch(Aany, n, b, m, n, ys, rpw, slabs, rr, nwg, st);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms / iters);
        tot += ms / iters;
      }
      printf("%-24s wg/cu %d nwg %5d : best %.1f us  %.0f GB/s (%.1f%% of 8TB/s)  mean %.1f us  relerr %.2e\n", v.name.c_str(), wpc,
             nwg, best * 1e3, bytes / (best * 1e-3) / 1e9, bytes / (best * 1e-3) / 8e12 * 100, tot / 3 * 1e3, err);
      fflush(stdout);
    }
  }
  return 0;
}
