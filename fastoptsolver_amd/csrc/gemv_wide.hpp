// Single pass over A for rows wider than the streaming kernel's register budget: 16384 < n <= 32768 fp32 columns, and
// (round 3) 24576 < n <= 32768 bf16 columns - 32 ... 64 KiB rows, which on the register-resident kernel would need y and
// the gradient slice of 8 bf16 chunks per thread (128 VGPRs) beside the row tiles and spill.
//
// gemv_pair_kernel keeps y, the gradient slice and the in-flight row tiles of a thread in registers; at 1024 threads
// (128 VGPRs) that ends at 4 chunks of 16 bytes per thread = 16384 columns, and wider rows used to take the two-pass
// kernels (A read twice, 4x slower).  Here y lives in LDS as fp32 (128 KiB of the CU's 160 KiB at n = 32768, dynamic
// shared memory, one workgroup per CU) and only the gradient slice and two row tiles stay in registers: the loads of
// row i+1 are issued before row i is reduced.  Same contract as gemv_pair_kernel (slabs[wg][n], rr_part[wg]); requires
// n % 4 == 0, lda % 4 == 0, 16-byte aligned A - the streaming layout.  No DUAL form.
#pragma once
#include "gemv_pair.hpp"

namespace fos {

constexpr int WD_THREADS = 512;
constexpr int WD_K = 16;                                  // 16-byte chunks per thread per row (fp32; bf16: 8)
constexpr int WD_MAX_N = WD_THREADS * WD_K * 4;           // 32768 columns (either storage type: 64 KiB of fp32 y in LDS... 128 KiB)

template <bool WITH_G, typename T = float>
__global__ __launch_bounds__(WD_THREADS) void gemv_wide_kernel(const T* __restrict__ A, int64_t lda,
                                                              const float* __restrict__ b, int64_t m, int n, YSource ys,
                                                              int64_t rows_per_wg, float* __restrict__ slabs,
                                                              double* __restrict__ rr_part) {
  using Tr = ElemTraits<T>;
  constexpr int EPC = Tr::EPC;                             // elements per 16-byte chunk
  constexpr int WD_K = WD_MAX_N / (WD_THREADS * EPC);      // (shadows the fp32 constant)
  extern __shared__ __attribute__((aligned(16))) float y_s[];         // n floats
  __shared__ float red[2][WD_THREADS / 64];
  if (ys.stopped != nullptr && *ys.stopped != 0) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const double beta = source_beta(ys);
  for (int j = tid; j < n; j += WD_THREADS) y_s[j] = (float)source_y(ys, j, beta);     // same form_y as everywhere
  __syncthreads();

  const int64_t row_lo = (int64_t)blockIdx.x * rows_per_wg;
  int64_t row_hi = row_lo + rows_per_wg;
  if (row_hi > m) row_hi = m;
  // chunk c of this thread covers columns (c*512 + tid)*EPC .. ; chunks beyond n re-read chunk 0 (branch-free loads,
  // see gemv_pair.hpp) and meet a zero weight
  bool live[WD_K];
  unsigned voff[WD_K];
#pragma unroll
  for (int c = 0; c < WD_K; ++c) {
    live[c] = (c * WD_THREADS + tid) * EPC < n;
    voff[c] = live[c] ? (unsigned)(c * WD_THREADS + tid) * 16u : (live[0] ? (unsigned)tid * 16u : 0u);
  }
  float gv[WITH_G ? WD_K : 1][EPC];
#pragma unroll
  for (int c = 0; c < (WITH_G ? WD_K : 1); ++c)
#pragma unroll
    for (int e = 0; e < EPC; ++e) gv[c][e] = 0.f;
  const char* base = reinterpret_cast<const char*>(A);
  const int64_t row_bytes = lda * (int64_t)sizeof(T);
  auto load_row = [&](int64_t row, u32x4 (&t)[WD_K]) {
    const char* rp = base + row * row_bytes;
#pragma unroll
    for (int c = 0; c < WD_K; ++c) t[c] = load16<true>(reinterpret_cast<const u32x4*>(rp + voff[c]));
  };
  double rr = 0.0;
  u32x4 cur[WD_K], nxt[WD_K];
  // b_i travels with its row (round 3): read after the barrier it put a memory latency on the critical path of every row
  float b_cur = 0.f, b_nxt = 0.f;
  if (row_lo < row_hi) {
    load_row(row_lo, cur);
    b_cur = b != nullptr ? b[row_lo] : 0.f;
  }
  for (int64_t row = row_lo; row < row_hi; ++row) {
    const int64_t rn = row + 1 < row_hi ? row + 1 : row;             // last row: re-read (L2 hit), keeps the loop uniform
    load_row(rn, nxt);
    b_nxt = b != nullptr ? b[rn] : 0.f;
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < WD_K; ++c) {
      if (live[c]) {
        float a[EPC];
        Tr::unpack(cur[c], a);
#pragma unroll
        for (int q = 0; q < EPC / 4; ++q) {
          const f32x4 yv = *reinterpret_cast<const f32x4*>(y_s + (c * WD_THREADS + tid) * EPC + 4 * q);
          acc += a[4 * q] * yv.x + a[4 * q + 1] * yv.y + a[4 * q + 2] * yv.z + a[4 * q + 3] * yv.w;
        }
      }
    }
    acc = wave_sum(acc);
    const int pb = (int)(row & 1);
    if (lane == 0) red[pb][wave] = acc;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < WD_THREADS / 64; ++w) s += red[pb][w];
    const float r = s - b_cur;
    rr += (double)r * (double)r;
    if constexpr (WITH_G) {
#pragma unroll
      for (int c = 0; c < WD_K; ++c) {
        float a[EPC];
        Tr::unpack(cur[c], a);
#pragma unroll
        for (int e = 0; e < EPC; ++e) gv[c][e] += a[e] * r;
      }
    }
#pragma unroll
    for (int c = 0; c < WD_K; ++c) cur[c] = nxt[c];
    b_cur = b_nxt;
  }
  if constexpr (WITH_G) {
#pragma unroll
    for (int c = 0; c < WD_K; ++c)
      if (live[c])
#pragma unroll
        for (int q = 0; q < EPC / 4; ++q)
          *reinterpret_cast<f32x4*>(slabs + (int64_t)blockIdx.x * n + (c * WD_THREADS + tid) * EPC + 4 * q) =
              f32x4{gv[c][4 * q], gv[c][4 * q + 1], gv[c][4 * q + 2], gv[c][4 * q + 3]};
  }
  if (tid == 0) rr_part[blockIdx.x] = rr;                  // every thread carries the same rr (same r values)
}

}  // namespace fos
