// Probe (not part of the product library): what does one hand-off per row panel between the workgroups of a column-strip
// cluster cost on gfx950?  It prices the single-read multi-lambda design of DESIGN.md "Multi-lambda": CS workgroups share
// a row panel, each holds a column strip, publishes its 16 x 16 partial residual block (1 KiB), waits for the others'
// and sums them.  Hand-off per MI355X_MICROARCH.md "hand-offs measured with sc1 loads" row 1: sc1 stores by one wave,
// s_waitcnt vmcnt(0), sc1 flag store by one lane; the consumer wave polls the flags with sc1 loads and then loads the
// bytes with sc1 loads.  Every wait is bounded.
//   hipcc -O3 --offload-arch=gfx950 -o tools/cluster_probe tools/cluster_probe.hip && ./tools/cluster_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef unsigned long long u64;
constexpr int SLOTS = 4, PART = 256 /* floats per partial */, FLAG_STRIDE = 32 /* unsigned: one 128-B line per flag */;
constexpr unsigned SPIN_LIMIT = 1u << 18;

__device__ inline void st_sc1(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline u64 ld_sc1(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline unsigned ld_flag(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void st_flag(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ inline float payload(int cluster, int member, int p, int e) { return (float)((cluster * 31 + member * 7 + p * 3 + e) & 1023); }

// LAG: the consumer of panel p runs LAG panels behind the publisher (LAG = 0: publish p, then wait for p)
template <int CS, int LAG>
__global__ __launch_bounds__(256) void probe(float* X, unsigned* flags, int npanels, unsigned epoch, int delay, int xcd_aware,
                                            double* out, int* err) {
  extern __shared__ char dyn[];
  __shared__ int abort_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int cluster, member;
  if (xcd_aware) {                 // members of a cluster on one XCD (workgroup i runs on XCD i % 8)
    const int xcd = blockIdx.x % 8, idx = blockIdx.x / 8;
    cluster = xcd + 8 * (idx / CS);
    member = idx % CS;
  } else {
    cluster = blockIdx.x / CS;
    member = blockIdx.x % CS;
  }
  if (tid == 0) abort_s = 0;
  if (tid == 0) dyn[0] = 0;
  __syncthreads();
  double check = 0.0;
  for (int p = 0; p < npanels + LAG; ++p) {
    if (delay > 0) {               // stands for the matrix-core work of a panel
      for (int i = 0; i < delay / 64; ++i) __builtin_amdgcn_s_sleep(1);      // ~64 cycles each
    }
    __syncthreads();
    if (abort_s) break;
    if (wave == 0) {
      if (p < npanels) {           // publish panel p
        u64* dst = reinterpret_cast<u64*>(X + (((size_t)cluster * SLOTS + (p % SLOTS)) * CS + member) * PART + lane * 4);
        const float v0 = payload(cluster, member, p, lane * 4), v1 = payload(cluster, member, p, lane * 4 + 1);
        const float v2 = payload(cluster, member, p, lane * 4 + 2), v3 = payload(cluster, member, p, lane * 4 + 3);
        st_sc1(dst, ((u64)__float_as_uint(v1) << 32) | __float_as_uint(v0));
        st_sc1(dst + 1, ((u64)__float_as_uint(v3) << 32) | __float_as_uint(v2));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) st_flag(flags + ((size_t)cluster * CS + member) * FLAG_STRIDE, epoch + p + 1);
      }
      const int q = p - LAG;       // consume panel q
      if (q >= 0) {
        bool ok = true;
        if (lane < CS) {
          const unsigned* f = flags + ((size_t)cluster * CS + lane) * FLAG_STRIDE;
          unsigned spins = 0;
          while ((int)(ld_flag(f) - epoch) < q + 1) {
            if (++spins > SPIN_LIMIT) { ok = false; break; }
          }
        }
        if (__ballot(!ok) != 0) {
          if (lane == 0) { abort_s = 1; *err = 1; }
        } else {
          float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
          u64 a[CS], b[CS];
#pragma unroll
          for (int j = 0; j < CS; ++j) {
            const u64* src = reinterpret_cast<const u64*>(X + (((size_t)cluster * SLOTS + (q % SLOTS)) * CS + j) * PART + lane * 4);
            a[j] = ld_sc1(src);
            b[j] = ld_sc1(src + 1);
          }
#pragma unroll
          for (int j = 0; j < CS; ++j) {
            s0 += __uint_as_float((unsigned)a[j]); s1 += __uint_as_float((unsigned)(a[j] >> 32));
            s2 += __uint_as_float((unsigned)b[j]); s3 += __uint_as_float((unsigned)(b[j] >> 32));
          }
          check += (double)s0 + (double)s1 + (double)s2 + (double)s3;
        }
      }
    }
  }
  if (wave == 0) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) check += __shfl_xor(check, off, 64);
    if (lane == 0) out[blockIdx.x] = check;
  }
}

template <int CS, int LAG>
static void run(const char* name, int nwg, int npanels, int delay, int xcd_aware, float* X, unsigned* flags, double* out, int* err,
                unsigned& epoch) {
  auto kern = probe<CS, LAG>;
  const size_t lds = 100 * 1024;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipMemset(err, 0, sizeof(int)));
    void* args[] = {&X, &flags, &npanels, &epoch, &delay, &xcd_aware, &out, &err};
    CK(hipEventRecord(e0, 0));
    CK(hipLaunchCooperativeKernel(reinterpret_cast<const void*>(kern), dim3(nwg), dim3(256), args, lds, 0));
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best;
    epoch += (unsigned)npanels + 16;
  }
  int herr = 0;
  CK(hipMemcpy(&herr, err, sizeof(int), hipMemcpyDeviceToHost));
  std::vector<double> h(nwg);
  CK(hipMemcpy(h.data(), out, nwg * sizeof(double), hipMemcpyDeviceToHost));
  // expected checksum of a workgroup: sum over panels, members, elements of the payload
  int bad = 0;
  for (int w = 0; w < nwg; ++w) {
    int cluster = xcd_aware ? (w % 8) + 8 * ((w / 8) / CS) : w / CS;
    double ref = 0.0;
    for (int p = 0; p < npanels; ++p)
      for (int j = 0; j < CS; ++j)
        for (int e = 0; e < PART; ++e) ref += (double)((cluster * 31 + j * 7 + p * 3 + e) & 1023);
    if (h[w] != ref) ++bad;
  }
  printf("%-28s CS %2d lag %d delay %5d cycles xcd_aware %d: %8.3f us per panel (%d panels), timeouts %d, wrong sums %d of %d\n", name, CS,
         LAG, delay, xcd_aware, best * 1e3 / npanels, npanels, herr, bad, nwg);
  fflush(stdout);
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int nwg = prop.multiProcessorCount;
  printf("device %s, %d CUs, cooperative launch %d\n", prop.name, nwg, prop.cooperativeLaunch);
  float* X; unsigned* flags; double* out; int* err;
  CK(hipMalloc(&X, (size_t)nwg * SLOTS * PART * sizeof(float) * 2));
  CK(hipMalloc(&flags, (size_t)nwg * FLAG_STRIDE * sizeof(unsigned)));
  CK(hipMemset(flags, 0, (size_t)nwg * FLAG_STRIDE * sizeof(unsigned)));
  CK(hipMalloc(&out, nwg * sizeof(double)));
  CK(hipMalloc(&err, sizeof(int)));
  unsigned epoch = 1000;
  const int np = 2000;
  for (int xa = 1; xa >= 0; --xa) {
    run<8, 0>("publish+wait+gather", nwg, np, 0, xa, X, flags, out, err, epoch);
    run<8, 0>("  + 2400-cycle panel work", nwg, np, 2400, xa, X, flags, out, err, epoch);
    run<8, 1>("lag 1", nwg, np, 0, xa, X, flags, out, err, epoch);
    run<8, 1>("lag 1 + 2400 cycles", nwg, np, 2400, xa, X, flags, out, err, epoch);
    run<8, 2>("lag 2 + 2400 cycles", nwg, np, 2400, xa, X, flags, out, err, epoch);
    run<16, 0>("publish+wait+gather", nwg, np, 0, xa, X, flags, out, err, epoch);
    run<16, 1>("lag 1 + 2400 cycles", nwg, np, 2400, xa, X, flags, out, err, epoch);
    run<16, 2>("lag 2 + 2400 cycles", nwg, np, 2400, xa, X, flags, out, err, epoch);
  }
  return 0;
}
