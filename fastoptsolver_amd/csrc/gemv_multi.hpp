// Multi-lambda single pass:  slab[w][v] = sum_{rows i of workgroup w} A_i^T (A_i . y_v - b_i)   for v = 0..NVEC-1.
//
// A regularisation path (the same A, b with several alpha1 / alpha2) runs NVEC FISTA state machines in lockstep; each
// iteration reads A from HBM ONCE for all of them ("A read once for many alpha1").  Structure
// identical to gemv_pair_kernel (R = 1 row per step, NBUF register tiles, counted vmcnt, branch-free loads); each thread
// keeps NVEC slices of y and of the running gradients in VGPRs, the per-row reduction carries NVEC partial dots.
// VALU work is 2*NVEC FMAs per element (NVEC = 4: a quarter of the VALU budget at the HBM rate).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gemv_pair.hpp"

namespace fos {

struct MultiY {
  const float* y[4];      // NVEC explicit fp32 vectors (16-byte aligned, n floats each)
  const int* stopped;     // optional device flag of the first state machine
};

template <typename T, int THREADS, int K, int NBUF, int NVEC, int MINW>
__global__ __launch_bounds__(THREADS, MINW) void gemv_multi_kernel(
    const T* __restrict__ A, int64_t lda, const float* __restrict__ b, int64_t m, int n, MultiY ys,
    int64_t rows_per_wg, float* __restrict__ slabs, double* __restrict__ rr_part) {
  using Tr = ElemTraits<T>;
  constexpr int EPC = Tr::EPC;
  constexpr int NW = THREADS / 64;
  static_assert(NVEC >= 1 && NVEC <= 4, "NVEC");
  __shared__ float red[2][NVEC][NW];

  if (ys.stopped != nullptr && *ys.stopped != 0) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t row_lo = (int64_t)blockIdx.x * rows_per_wg;
  int64_t row_hi = row_lo + rows_per_wg;
  if (row_hi > m) row_hi = m;

  float yv[NVEC][K][EPC];
  float gv[NVEC][K][EPC];
  bool live[K];
#pragma unroll
  for (int c = 0; c < K; ++c) {
    const int col = (c * THREADS + tid) * EPC;
    live[c] = col < n;
#pragma unroll
    for (int v = 0; v < NVEC; ++v) {
#pragma unroll
      for (int e = 0; e < EPC; ++e) { yv[v][c][e] = 0.f; gv[v][c][e] = 0.f; }
      if (live[c]) {
#pragma unroll
        for (int q = 0; q < EPC / 4; ++q) {
          const f32x4 t = *reinterpret_cast<const f32x4*>(ys.y[v] + col + 4 * q);
          yv[v][c][4 * q + 0] = t.x; yv[v][c][4 * q + 1] = t.y; yv[v][c][4 * q + 2] = t.z; yv[v][c][4 * q + 3] = t.w;
        }
      }
    }
  }
  const bool any_live = live[0];
  unsigned voff[K];
#pragma unroll
  for (int c = 0; c < K; ++c) voff[c] = (any_live ? (unsigned)tid * 16u : 0u) + (live[c] ? (unsigned)c * THREADS * 16u : 0u);

  const int64_t nsteps = row_hi > row_lo ? row_hi - row_lo : 0;
  const char* base = reinterpret_cast<const char*>(A);
  const int64_t row_bytes = lda * (int64_t)sizeof(T);
  double rr[NVEC];
#pragma unroll
  for (int v = 0; v < NVEC; ++v) rr[v] = 0.0;

  u32x4 tile[NBUF][K];
  float bval[NBUF];
  const float* b_src = b != nullptr ? b : reinterpret_cast<const float*>(A);
  auto issue = [&](int buf, int64_t step) {
    int64_t row = row_lo + step;
    if (row >= row_hi) row = row_hi - 1;
    bval[buf] = b_src[b != nullptr ? row : 0];
    const char* rp = base + row * row_bytes;
#pragma unroll
    for (int c = 0; c < K; ++c) tile[buf][c] = load16<true>(rp + voff[c]);
  };
  auto consume = [&](int buf, int64_t step) {
    float part[NVEC];
#pragma unroll
    for (int v = 0; v < NVEC; ++v) {
      float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
      for (int c = 0; c < K; ++c) {
        float a[EPC];
        Tr::unpack(tile[buf][c], a);
#pragma unroll
        for (int e = 0; e < EPC; e += 2) {
          acc0 = fmaf(a[e], yv[v][c][e], acc0);
          acc1 = fmaf(a[e + 1], yv[v][c][e + 1], acc1);
        }
      }
      part[v] = wave_sum(acc0 + acc1);
    }
    const int pb = (int)(step & 1);
    if (lane == 0) {
#pragma unroll
      for (int v = 0; v < NVEC; ++v) red[pb][v][wave] = part[v];
    }
    __syncthreads();
    const bool valid = row_lo + step < row_hi;
    const float bi = b != nullptr ? bval[buf] : 0.f;
    float res[NVEC];
#pragma unroll
    for (int v = 0; v < NVEC; ++v) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) s += red[pb][v][w];
      s = valid ? s - bi : 0.f;
      rr[v] += (double)s * (double)s;
      res[v] = s;
    }
#pragma unroll
    for (int c = 0; c < K; ++c) {
      float a[EPC];
      Tr::unpack(tile[buf][c], a);
#pragma unroll
      for (int v = 0; v < NVEC; ++v)
#pragma unroll
        for (int e = 0; e < EPC; ++e) gv[v][c][e] = fmaf(a[e], res[v], gv[v][c][e]);
    }
  };

  if (nsteps > 0) {
#pragma unroll
    for (int u = 0; u < NBUF - 1; ++u) issue(u, u);
    int64_t s = 0;
    for (; s + NBUF <= nsteps; s += NBUF) {
#pragma unroll
      for (int u = 0; u < NBUF; ++u) {
        issue((u + NBUF - 1) % NBUF, s + u + NBUF - 1);
        __builtin_amdgcn_sched_barrier(0);
        consume(u, s + u);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int u = 0; u < NBUF - 1; ++u)
      if (s + u < nsteps) consume(u, s + u);
  }

#pragma unroll
  for (int v = 0; v < NVEC; ++v) {
    float* slab = slabs + ((int64_t)blockIdx.x * NVEC + v) * n;
#pragma unroll
    for (int c = 0; c < K; ++c) {
      if (live[c]) {
        const int col = (c * THREADS + tid) * EPC;
#pragma unroll
        for (int q = 0; q < EPC / 4; ++q) {
          f32x4 o = {gv[v][c][4 * q + 0], gv[v][c][4 * q + 1], gv[v][c][4 * q + 2], gv[v][c][4 * q + 3]};
          *reinterpret_cast<f32x4*>(slab + col + 4 * q) = o;
        }
      }
    }
    if (tid == 0) rr_part[(int64_t)blockIdx.x * NVEC + v] = rr[v];
  }
}

}  // namespace fos
