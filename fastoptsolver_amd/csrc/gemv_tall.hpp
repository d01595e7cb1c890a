// Tall-skinny single-pass GEMV pair: n <= 64 columns, any m, any lda >= n, ragged n welcome.
//
// The streaming kernel (gemv_pair.hpp) gives a whole row to a workgroup: with rows of a few dozen bytes - regression
// data with many samples and few features, the reference's own domain (its generator is m x 5) - nearly every lane
// idles (1,000,000 x 5 ran at 0.1 % of the HBM roofline, 1.7 ms per iteration).  Here every THREAD owns whole rows:
// lane l of a wave reads row base+l, so a wave's loads cover 64 consecutive rows = one contiguous block of memory
// (A is row-major), the dot with y and the rank-1 update of the gradient happen in that thread's registers, and one
// block reduction per workgroup ends the pass.  y (and x_k for DUAL) sits in LDS as doubles (broadcast reads); the
// row dots, ||r||^2 and the gradient accumulators are fp64 (these problems are the ill-conditioned ones, cond 1e9 for
// the unstandardised Boston features), the slab written per workgroup is fp32 like every other path (fp64 in the ST = double form that serves the L-BFGS
// fg, fos_gemv_pair_dd).
// Output contract = gemv_pair_kernel except for the slab row stride: slabs[wg][tall_slab_stride(n)] (zero padded),
// rr_part[wg], rr2_part[wg] (DUAL), rows [wg*rows_per_wg, ...).
#pragma once
#include <type_traits>

#include "gemv_pair.hpp"

namespace fos {

constexpr int TL_THREADS = 256;
constexpr int TL_MAX_N = 64;
constexpr int TLR_MAX_N = 128;          // aligned rows, chunk per lane: up to 32 lanes per row (gemv_tall_rows_kernel)
__host__ __device__ inline int tall_slab_stride(int n) { return (n + 3) & ~3; }

// NC: compile-time column capacity (8/16/32/64; columns beyond n are zero).  LOAD: how a thread gets its row -
//   TL_DIRECT  scalar loads straight from global (any lda; a wave's 64 rows share cache lines, so HBM traffic stays 1x,
//              but every instruction touches all of them: 2.6 % of the roofline at n = 5)
//   TL_VEC     16-byte loads (fp32, n % 4 == 0, lda % 4 == 0, A 16-byte aligned)
//   TL_STAGE   contiguous matrix (lda == n), ragged n: the workgroup copies its 256 x n block of A - one contiguous
//              span of memory - into LDS with fully coalesced loads, and each thread then reads its row from LDS
//              (stride n words: conflict-free for odd n); NC <= 32 (LDS budget)
//   TL_STAGE4  TL_STAGE for fp32 with 16-byte aligned block starts (A aligned, rows per workgroup a multiple of 4): the
//              copy moves float4s - 3 load instructions per thread and block instead of 10 at n = 5
enum : int { TL_DIRECT = 0, TL_VEC = 1, TL_STAGE = 2, TL_STAGE4 = 3 };
template <typename T, int NC, int LOAD, bool WITH_G, bool DUAL, typename ST = float, bool STAGE_NT = false>
__global__ __launch_bounds__(TL_THREADS) void gemv_tall_kernel(const T* __restrict__ A, int64_t lda,
                                                              const float* __restrict__ b, int64_t m, int n, YSource ys,
                                                              int64_t rows_per_wg, ST* __restrict__ slabs,
                                                              double* __restrict__ rr_part, double* __restrict__ rr2_part) {
  constexpr int NW = TL_THREADS / 64;
  constexpr bool VEC = (LOAD == TL_VEC), STAGE = (LOAD == TL_STAGE || LOAD == TL_STAGE4), STAGE4 = (LOAD == TL_STAGE4);
  static_assert(!STAGE4 || sizeof(T) == 4, "TL_STAGE4 is the fp32 form");
  __shared__ __attribute__((aligned(16))) float tile_s[STAGE ? (NC <= 16 ? 2 : 1) * TL_THREADS * NC : 4];
  __shared__ double y_s[NC];
  __shared__ double x_s[DUAL ? NC : 1];
  __shared__ double red[NW][8];
  if (ys.stopped != nullptr && *ys.stopped != 0) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const double beta = source_beta(ys);
  const int sstride = tall_slab_stride(n);
  if (tid < NC) {
    y_s[tid] = tid < n ? source_y(ys, tid, beta) : 0.0;
    if constexpr (DUAL) x_s[tid] = tid < n ? ys.x_cur[tid] : 0.0;
  }
  __syncthreads();

  double g[WITH_G ? NC : 1];
#pragma unroll
  for (int j = 0; j < (WITH_G ? NC : 1); ++j) g[j] = 0.0;
  double rr = 0.0, rr2 = 0.0;
  const int64_t row_lo = (int64_t)blockIdx.x * rows_per_wg;
  int64_t row_hi = row_lo + rows_per_wg;
  if (row_hi > m) row_hi = m;
  // RPI rows per thread per loop trip: the loads of all of them are issued before the first is consumed (a thread has
  // no other memory-level parallelism than its own rows); 2 where the registers allow (NC <= 16).
  constexpr int RPI = NC <= 16 ? 2 : 1;
  constexpr int BLOCK_ROWS = RPI * TL_THREADS;
  // TL_STAGE: the block after the one being consumed is already on its way into registers (its global-load latency
  // overlaps the LDS reads and the fp64 arithmetic of this one); it moves into LDS between two barriers.
  constexpr int NPRE = STAGE4 ? (NC * RPI + 3) / 4 : NC * RPI;          // float4s / floats per thread and block
  typedef typename std::conditional<STAGE4, f32x4, float>::type PreT;
  PreT pre[STAGE ? NPRE : 1];
  // b of the prefetched block travels with it (round 3): loaded at consume time it put a whole global-memory latency on
  // the critical path of every 512-row block - the one thing in this loop that nothing overlapped
  float bpre[STAGE ? RPI : 1];
  auto prefetch = [&](int64_t row0) {
    if constexpr (STAGE) {
#pragma unroll
      for (int u = 0; u < RPI; ++u) {
        const int64_t row = row0 + u * TL_THREADS + tid;
        bpre[u] = (row < row_hi && b != nullptr) ? b[row] : 0.f;
      }
      const int64_t left = row_hi - row0;
      const int count = left <= 0 ? 0 : (int)(left < BLOCK_ROWS ? left : BLOCK_ROWS) * n;
      const T* src = A + row0 * (int64_t)n;                              // lda == n: the block is one contiguous span
#pragma unroll
      for (int u = 0; u < NPRE; ++u) {
        if constexpr (STAGE4) {
          const float* srcf = reinterpret_cast<const float*>(src);
          const int i = 4 * (u * TL_THREADS + tid);
          f32x4 v = {0.f, 0.f, 0.f, 0.f};
          if (i + 3 < count) {
            // (the staged copy reads one contiguous span, 1 KiB per wave instruction: unlike the row-wise 16-byte loads of
            // TL_VEC it can stream past the caches)
            if constexpr (STAGE_NT) v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(srcf + i));
            else v = *reinterpret_cast<const f32x4*>(srcf + i);
          } else if (i < count) {                                        // the block's last, partial quad
            v.x = srcf[i];
            if (i + 1 < count) v.y = srcf[i + 1];
            if (i + 2 < count) v.z = srcf[i + 2];
          }
          pre[u] = v;
        } else {
          const int i = u * TL_THREADS + tid;
          pre[u] = i < count ? elem_to_float<T>(src[i]) : 0.f;
        }
      }
    }
  };
  prefetch(row_lo);
  for (int64_t row0 = row_lo; row0 < row_hi; row0 += BLOCK_ROWS) {          // uniform loop: TL_STAGE needs the barriers
    float bcur[STAGE ? RPI : 1];
    if constexpr (STAGE) {
#pragma unroll
      for (int u = 0; u < RPI; ++u) bcur[u] = bpre[u];
      __syncthreads();                                                   // the previous block has been consumed
#pragma unroll
      for (int u = 0; u < NPRE; ++u) {
        if constexpr (STAGE4) {
          const int i = 4 * (u * TL_THREADS + tid);
          if (i < BLOCK_ROWS * n) *reinterpret_cast<f32x4*>(&tile_s[i]) = pre[u];      // BLOCK_ROWS * n is a multiple of 4
        } else {
          const int i = u * TL_THREADS + tid;
          if (i < BLOCK_ROWS * n) tile_s[i] = pre[u];
        }
      }
      __syncthreads();
      prefetch(row0 + BLOCK_ROWS);
    }
    float a[RPI][NC];
    double bi[RPI];
#pragma unroll
    for (int u = 0; u < RPI; ++u) {
      const int64_t row = row0 + u * TL_THREADS + tid;
      const bool in = row < row_hi;
      if constexpr (STAGE) bi[u] = (double)bcur[u];
      else bi[u] = (in && b != nullptr) ? (double)b[row] : 0.0;
      if constexpr (STAGE) {
        const float* ar = tile_s + (u * TL_THREADS + tid) * n;
#pragma unroll
        for (int j = 0; j < NC; ++j) a[u][j] = (in && j < n) ? ar[j] : 0.f;
      } else if constexpr (VEC) {
        const T* ar = A + row * lda;
#pragma unroll
        for (int c = 0; c < NC / 4; ++c) {
          f32x4 v = {0.f, 0.f, 0.f, 0.f};
          if (in && 4 * c < n) v = reinterpret_cast<const f32x4*>(ar)[c];   // cached: the 4 pieces of a line hit L1
          a[u][4 * c] = v.x; a[u][4 * c + 1] = v.y; a[u][4 * c + 2] = v.z; a[u][4 * c + 3] = v.w;
        }
      } else {
        const T* ar = A + row * lda;
#pragma unroll
        for (int j = 0; j < NC; ++j) a[u][j] = (in && j < n) ? elem_to_float<T>(ar[j]) : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < RPI; ++u) {            // rows beyond row_hi: a = 0, b = 0 -> r = 0, no contribution
      double acc = -bi[u], acc2 = -bi[u];
#pragma unroll
      for (int j = 0; j < NC; ++j) {
        acc += (double)a[u][j] * y_s[j];
        if constexpr (DUAL) acc2 += (double)a[u][j] * x_s[j];
      }
      rr += acc * acc;
      if constexpr (DUAL) rr2 += acc2 * acc2;
      if constexpr (WITH_G) {
#pragma unroll
        for (int j = 0; j < NC; ++j) g[j] += (double)a[u][j] * acc;
      }
    }
  }

  // ---- one block reduction per workgroup: 8 values at a time through the DPP ladder, 4 wave partials through LDS ----
  auto block_reduce8 = [&](double (&v)[8], auto&& sink) {
    wave_sum_n(v);
    if (lane == 0) {
#pragma unroll
      for (int c = 0; c < 8; ++c) red[wave][c] = v[c];
    }
    __syncthreads();
    if (tid < 8) sink(tid, (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]));
    __syncthreads();
  };
  if constexpr (WITH_G) {
#pragma unroll
    for (int c0 = 0; c0 < NC; c0 += 8) {
      double v[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) v[c] = g[c0 + c];
      block_reduce8(v, [&](int c, double tot) {      // slab rows are padded to a multiple of 4 floats (zeros): the
        if (c0 + c < sstride) slabs[(int64_t)blockIdx.x * sstride + c0 + c] = (ST)tot;   // float4 epilogues take any n
      });
    }
  }
  double tail[8] = {rr, rr2, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  block_reduce8(tail, [&](int c, double tot) {
    if (c == 0) rr_part[blockIdx.x] = tot;
    if (DUAL && c == 1) rr2_part[blockIdx.x] = tot;
  });
}

// 33..64 columns: a row per QUAD of lanes, 16 columns per lane.  The row-per-thread form needs y, the row and 64 fp64
// accumulators in one thread - 400 registers, one workgroup per CU, 37 % of the roofline at 2,000,000 x 64.  Here a
// lane keeps 16 columns of everything; the four partial dots of a row meet through two quad_perm DPP moves, and the
// per-lane gradient partials are folded through LDS once per workgroup.  Same output contract as gemv_tall_kernel.
constexpr int TQ_CPL = 16;                 // columns per lane
constexpr int TQ_ROWS = TL_THREADS / 4;    // rows per workgroup trip and per row slot (64)

__device__ inline double quad_sum(double v) {
  v += dpp_fetch<0xB1, 0xF>(v);            // quad_perm [1,0,3,2]
  v += dpp_fetch<0x4E, 0xF>(v);            // quad_perm [2,3,0,1]
  return v;
}

template <typename T, bool VEC, bool WITH_G, bool DUAL, typename ST = float>
__global__ __launch_bounds__(TL_THREADS) void gemv_tall_quad_kernel(const T* __restrict__ A, int64_t lda,
                                                                   const float* __restrict__ b, int64_t m, int n, YSource ys,
                                                                   int64_t rows_per_wg, ST* __restrict__ slabs,
                                                                   double* __restrict__ rr_part, double* __restrict__ rr2_part) {
  constexpr int NW = TL_THREADS / 64, RPI = 2;
  __shared__ double y_s[64];
  __shared__ double x_s[DUAL ? 64 : 1];
  __shared__ double gred[WITH_G ? TL_THREADS : 1][TQ_CPL];
  __shared__ double red[NW][8];
  if (ys.stopped != nullptr && *ys.stopped != 0) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int sub = tid & 3, rslot = tid >> 2;                 // column block of this lane, row slot within a trip
  const int cbase = sub * TQ_CPL;
  const double beta = source_beta(ys);
  const int sstride = tall_slab_stride(n);
  if (tid < 64) {
    y_s[tid] = tid < n ? source_y(ys, tid, beta) : 0.0;
    if constexpr (DUAL) x_s[tid] = tid < n ? ys.x_cur[tid] : 0.0;
  }
  __syncthreads();
  double yv[TQ_CPL], xv[DUAL ? TQ_CPL : 1];
#pragma unroll
  for (int c = 0; c < TQ_CPL; ++c) {
    yv[c] = y_s[cbase + c];
    if constexpr (DUAL) xv[c] = x_s[cbase + c];
  }
  double g[WITH_G ? TQ_CPL : 1];
#pragma unroll
  for (int c = 0; c < (WITH_G ? TQ_CPL : 1); ++c) g[c] = 0.0;
  double rr = 0.0, rr2 = 0.0;
  const int64_t row_lo = (int64_t)blockIdx.x * rows_per_wg;
  int64_t row_hi = row_lo + rows_per_wg;
  if (row_hi > m) row_hi = m;
  for (int64_t row0 = row_lo; row0 < row_hi; row0 += RPI * TQ_ROWS) {
    float a[RPI][TQ_CPL];
    double bi[RPI];
#pragma unroll
    for (int u = 0; u < RPI; ++u) {
      const int64_t row = row0 + u * TQ_ROWS + rslot;
      const bool in = row < row_hi;
      bi[u] = (in && b != nullptr) ? (double)b[row] : 0.0;
      const T* ar = A + row * lda + cbase;
      if constexpr (VEC) {
#pragma unroll
        for (int c = 0; c < TQ_CPL / 4; ++c) {
          f32x4 v = {0.f, 0.f, 0.f, 0.f};
          if (in && cbase + 4 * c < n) v = reinterpret_cast<const f32x4*>(ar)[c];       // cached: lines are shared
          a[u][4 * c] = v.x; a[u][4 * c + 1] = v.y; a[u][4 * c + 2] = v.z; a[u][4 * c + 3] = v.w;
        }
      } else {
#pragma unroll
        for (int c = 0; c < TQ_CPL; ++c) a[u][c] = (in && cbase + c < n) ? elem_to_float<T>(ar[c]) : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < RPI; ++u) {            // rows beyond row_hi: a = 0, b = 0 -> r = 0
      double acc = 0.0, acc2 = 0.0;
#pragma unroll
      for (int c = 0; c < TQ_CPL; ++c) {
        acc += (double)a[u][c] * yv[c];
        if constexpr (DUAL) acc2 += (double)a[u][c] * xv[c];
      }
      acc = quad_sum(acc) - bi[u];             // all four lanes of the quad hold r_i
      if (sub == 0) rr += acc * acc;
      if constexpr (DUAL) {
        acc2 = quad_sum(acc2) - bi[u];
        if (sub == 0) rr2 += acc2 * acc2;
      }
      if constexpr (WITH_G) {
#pragma unroll
        for (int c = 0; c < TQ_CPL; ++c) g[c] += (double)a[u][c] * acc;
      }
    }
  }
  if constexpr (WITH_G) {
#pragma unroll
    for (int c = 0; c < TQ_CPL; ++c) gred[tid][c] = g[c];
    __syncthreads();
    if (tid < 64) {                            // column tid: held by the lanes with sub == tid / 16, one per row slot
      const int sb = tid / TQ_CPL, c = tid % TQ_CPL;
      double tot = 0.0;
      for (int k = 0; k < TQ_ROWS; ++k) tot += gred[4 * k + sb][c];
      if (tid < sstride) slabs[(int64_t)blockIdx.x * sstride + tid] = tid < n ? (ST)tot : (ST)0;
    }
  }
  double tail[8] = {rr, rr2, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  wave_sum_n(tail);
  if (lane == 0) { red[wave][0] = tail[0]; red[wave][1] = tail[1]; }
  __syncthreads();
  if (tid == 0) {
    rr_part[blockIdx.x] = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
    if constexpr (DUAL) rr2_part[blockIdx.x] = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
  }
}

// ---------------------------------------------------------------------------------------------------------
// Aligned tall-skinny rows (n <= 128, n and lda multiples of 16 bytes, A 16-byte aligned): a row per LPR lanes, every
// lane owns ONE 16-byte chunk of the row (4 fp32 / 8 bf16 columns).  A wave's load instruction then reads 64 consecutive
// chunks = 1 KiB of contiguous memory when lda == n (the row-per-thread and row-per-quad forms above touch 16 bytes of 64
// different rows / a quarter of every line per instruction and reach 52 % / 53 % of the roofline at n = 32 / 64), the
// per-lane state is 4-8 columns of y and of the gradient instead of 16-64, and U row steps are in flight per lane.
// The row dot is a butterfly over the LPR lanes on the DPP path (quad_perm, row_half_mirror, row_mirror: every lane of
// the row ends with the total, no broadcast step).  fp64 accumulation like the other tall kernels.
// Output contract = gemv_tall_kernel.
// ---------------------------------------------------------------------------------------------------------
template <int LPR>
__device__ inline double lanes_sum(double v) {
  static_assert(LPR == 4 || LPR == 8 || LPR == 16 || LPR == 32, "LPR");
  v += dpp_fetch<0xB1, 0xF>(v);                         // quad_perm [1,0,3,2]
  v += dpp_fetch<0x4E, 0xF>(v);                         // quad_perm [2,3,0,1]
  if constexpr (LPR >= 8) v += dpp_fetch<0x141, 0xF>(v);   // row_half_mirror
  if constexpr (LPR >= 16) v += dpp_fetch<0x140, 0xF>(v);  // row_mirror
  if constexpr (LPR >= 32) v += __shfl_xor(v, 16);         // the other 16-lane row of the pair (no DPP pattern crosses rows
  return v;                                                //  both ways on this ISA: one ds_bpermute pair per row step)
}

template <int LPR>
__device__ inline float lanes_sum(float v) {
  static_assert(LPR == 4 || LPR == 8 || LPR == 16 || LPR == 32, "LPR");
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false));
  if constexpr (LPR >= 8) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, false));
  if constexpr (LPR >= 16) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, false));
  if constexpr (LPR >= 32) v += __shfl_xor(v, 16);
  return v;
}

// AT: accumulator type of the row dot and the gradient slice.  double for fp32 storage (the ill-conditioned regression data
// this family exists for) and for every fp64 pass; float for the fp32 pass over bf16 storage (round 3: eight 2-byte
// elements per chunk through fp64 conversion and accumulation made that pass instruction-bound at 42-54 % of the roofline;
// its streaming siblings accumulate in fp32 as well).  ||r||^2 stays fp64.
template <typename T, int LPR, bool WITH_G, bool DUAL, typename ST = float, typename AT = double>
__global__ __launch_bounds__(TL_THREADS) void gemv_tall_rows_kernel(const T* __restrict__ A, int64_t lda,
                                                                   const float* __restrict__ b, int64_t m, int n, YSource ys,
                                                                   int64_t rows_per_wg, ST* __restrict__ slabs,
                                                                   double* __restrict__ rr_part, double* __restrict__ rr2_part) {
  using Tr = ElemTraits<T>;
  constexpr int EPL = Tr::EPC;                  // columns per lane (one 16-byte chunk)
  constexpr int RPS = TL_THREADS / LPR;         // rows per step of the workgroup
  constexpr int U = 4;                          // row steps in flight per lane (8 measured the same: r02_sweep_wgs)
  constexpr int NW = TL_THREADS / 64;
  __shared__ AT gred[WITH_G ? TL_THREADS : 1][EPL];
  __shared__ double red[NW][2];
  if (ys.stopped != nullptr && *ys.stopped != 0) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int sub = tid % LPR, rslot = tid / LPR;
  const int cbase = sub * EPL;
  const bool live = cbase < n;                  // n % EPL == 0: a chunk is inside the row or outside it entirely
  const double beta = source_beta(ys);
  AT yv[EPL], xv[DUAL ? EPL : 1], g[WITH_G ? EPL : 1];
#pragma unroll
  for (int e = 0; e < EPL; ++e) {
    yv[e] = live ? (AT)source_y(ys, cbase + e, beta) : (AT)0;
    if constexpr (DUAL) xv[e] = live ? (AT)ys.x_cur[cbase + e] : (AT)0;
    if constexpr (WITH_G) g[e] = (AT)0;
  }
  double rr = 0.0, rr2 = 0.0;
  const int64_t row_lo = (int64_t)blockIdx.x * rows_per_wg;
  int64_t row_hi = row_lo + rows_per_wg;
  if (row_hi > m) row_hi = m;
  // Branch-free loads (rows past the end re-read the last row, lanes past n re-read chunk 0; both are masked when
  // consumed), two register sets: the loads of the next U row steps are in flight while this set is consumed, and the
  // compiler can wait with a counted vmcnt (gemv_pair.hpp "Software pipeline").
  const char* base = reinterpret_cast<const char*>(A) + (live ? (size_t)cbase * sizeof(T) : 0);
  const int64_t row_bytes = lda * (int64_t)sizeof(T);
  const float* b_src = b != nullptr ? b : reinterpret_cast<const float*>(A);
  constexpr int STEP = U * RPS;
  auto issue = [&](u32x4 (&raw)[U], float (&bi)[U], int64_t row0) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int64_t row = row0 + u * RPS + rslot;
      if (row >= row_hi) row = row_hi - 1;
      raw[u] = load16<true>(base + row * row_bytes);
      bi[u] = b_src[b != nullptr ? row : 0];
    }
  };
  auto consume = [&](const u32x4 (&raw)[U], const float (&bi)[U], int64_t row0) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool in = (row0 + u * RPS + rslot < row_hi) && live;
      float a[EPL];
      Tr::unpack(raw[u], a);
      AT acc = (AT)0, acc2 = (AT)0;
#pragma unroll
      for (int e = 0; e < EPL; ++e) {
        a[e] = in ? a[e] : 0.f;                  // masked rows / lanes: a = 0 -> no contribution to the dot or to g
        acc = fma((AT)a[e], yv[e], acc);
        if constexpr (DUAL) acc2 = fma((AT)a[e], xv[e], acc2);
      }
      const bool row_in = row0 + u * RPS + rslot < row_hi;
      const AT bv = (row_in && b != nullptr) ? (AT)bi[u] : (AT)0;
      acc = lanes_sum<LPR>(acc) - bv;
      if (sub == 0 && row_in) rr += (double)acc * (double)acc;
      if constexpr (DUAL) {
        acc2 = lanes_sum<LPR>(acc2) - bv;
        if (sub == 0 && row_in) rr2 += (double)acc2 * (double)acc2;
      }
      if constexpr (WITH_G) {
#pragma unroll
        for (int e = 0; e < EPL; ++e) g[e] = fma((AT)a[e], acc, g[e]);
      }
    }
  };
  if (row_lo < row_hi) {
    u32x4 rawA[U], rawB[U];
    float biA[U], biB[U];
    issue(rawA, biA, row_lo);
    for (int64_t row0 = row_lo; row0 < row_hi; row0 += 2 * STEP) {
      issue(rawB, biB, row0 + STEP);
      consume(rawA, biA, row0);
      issue(rawA, biA, row0 + 2 * STEP);
      consume(rawB, biB, row0 + STEP);
    }
  }
  const int sstride = tall_slab_stride(n);
  if constexpr (WITH_G) {
#pragma unroll
    for (int e = 0; e < EPL; ++e) gred[tid][e] = g[e];
    __syncthreads();
    if (tid < LPR * EPL) {                       // column tid: held by the lanes with sub == tid / EPL, one per row slot
      const int sb = tid / EPL, e = tid % EPL;
      double tot = 0.0;
      for (int k = 0; k < RPS; ++k) tot += (double)gred[k * LPR + sb][e];
      if (tid < sstride) slabs[(int64_t)blockIdx.x * sstride + tid] = tid < n ? (ST)tot : (ST)0;
    }
  }
  double tail[8] = {rr, rr2, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  wave_sum_n(tail);
  if (lane == 0) { red[wave][0] = tail[0]; red[wave][1] = tail[1]; }
  __syncthreads();
  if (tid == 0) {
    rr_part[blockIdx.x] = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
    if constexpr (DUAL) rr2_part[blockIdx.x] = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
  }
}

}  // namespace fos
