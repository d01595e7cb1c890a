// Batched Armijo trials: ||A dlt_j||^2 for up to 16 candidate steps in ONE pass over A, on the matrix cores.
//
// Spec: the backtracking loops iterative_solvers.py:187-194 / :302-309 / :96-104 evaluate g at one candidate per
// trial (two passes over A each).  With dlt_j = prox(y - t_j*grad) - y for t_j = t*eta^j, j = 0..NV-1, the test of
// candidate j is (1-C)*grad.dlt_j + 0.5*||A dlt_j||^2 + 0.5*alpha2*||dlt_j||^2 <= 0, so ONE kernel that
// returns q_j = ||A dlt_j||^2 for all j decides the whole line search.  SURVEY.md §8(f) rank 1.
//
// This is GEMM-shaped (A[m x n] times X[n x 16]), so it runs on MFMA: v_mfma_f32_16x16x4_f32 (exact fp32, k-ordered
// fma chain).  Per 64-row x 64-column tile the workgroup stages A through LDS with fully coalesced 16-byte loads
// (4 rows x 256 B per wave instruction); a fragment-shaped direct load would touch 16 rows x 64 B per instruction.
// Wave w owns rows 16w..16w+15 of the tile: lane l supplies A[row l&15][col 4(l>>4)+c] (one ds_read_b128, c = the
// four MFMA steps) and X[col 4(l>>4)+c][vector l&15] (one ds_read_b128 of the pre-permuted candidate block).
// D (4 VGPRs) holds 16 rows x 16 candidates.  Rate: 64 MFMAs of 32 cycles per 16 KiB tile per CU = 32 B/clk/CU of
// matrix-core capacity against the ~13 B/clk/CU the HBM stream delivers (~40 % MFMA busy) - with 16 right-hand
// sides the matrix cores are the right unit; with one (the main gradient kernel) they are not.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gemv_pair.hpp"
#include "reduce_update.hpp"

namespace fos {

constexpr int BT_ROWS = 64;          // rows per tile  (4 waves x 16)
#ifndef FOS_BT_COLS
#define FOS_BT_COLS 64
#endif
constexpr int BT_COLS = FOS_BT_COLS; // columns per tile (row segments of 4*BT_COLS bytes per load instruction group)
constexpr int BT_LDS_STRIDE = BT_COLS + 4;   // padded row stride (floats): rows land on distinct bank groups
constexpr int BT_THREADS = 256;
constexpr int BT_F4_ROW = BT_COLS / 4;                               // float4 per tile row
constexpr int BT_A_LOADS = BT_ROWS * BT_F4_ROW / BT_THREADS;         // float4 of A per thread per tile
constexpr int BT_X_LOADS = BT_COLS * BT_NV / 4 / BT_THREADS;         // float4 of the candidate block per thread per tile
constexpr int BT_W = 3 * BT_NV + 2;  // per-workgroup outputs of the candidate kernel

static_assert(BT_COLS % 16 == 0 && BT_A_LOADS >= 1 && BT_X_LOADS >= 1, "tile shape");

// Candidate generation: dlt_j (fp64 -> fp32, written in Xp layout) and per-candidate sums.
// part[wg] = { gd_j (16), dd_j (16), nnz_j (16), ||grad||^2, ||y||^2 }.
static __global__ __launch_bounds__(256) void fista_trial_batch_kernel(GradSrc gsrc, int n, int n_pad,
                                                               const double* __restrict__ x_cur,
                                                               const double* __restrict__ x_prev,
                                                               const FistaScalars* __restrict__ scal, FistaParams prm,
                                                               double t0, double eta, int nv, float* __restrict__ xp,
                                                               double* __restrict__ part, int t_from_state = 0) {
  __shared__ double red[4][BT_W];
  if (t_from_state) {                            // device-driven backtracking: the step lives in FistaScalars::tau
    if (scal->stopped != 0) return;
    t0 = scal->tau;
  }
  const double beta = scal->beta;
  double gd[BT_NV], dd[BT_NV], nz[BT_NV];
#pragma unroll
  for (int j = 0; j < BT_NV; ++j) { gd[j] = 0.0; dd[j] = 0.0; nz[j] = 0.0; }
  double g2 = 0.0, y2 = 0.0;
  for (int col = blockIdx.x * 256 + threadIdx.x; col < n_pad; col += gridDim.x * 256) {
    if (col < n) {
      const double y = form_y(x_cur[col], x_prev[col], beta);
      double gf = grad_at(gsrc, col);
      if (prm.prox_kind == PROX_L1 && prm.alpha2 > 0.0) gf += prm.alpha2 * y;
      g2 += gf * gf;
      y2 += y * y;
      double t = t0;
#pragma unroll
      for (int j = 0; j < BT_NV; ++j) {
        double d = 0.0;
        if (j < nv) {
          const double v = y - t * gf;
          double xt = prm.alpha1 > 0.0 ? soft_threshold(v, t * prm.alpha1) : v;
          if (prm.prox_kind == PROX_ENET) xt *= 1.0 / (1.0 + t * prm.alpha2);
          d = xt - y;
          gd[j] += gf * d;
          dd[j] += d * d;
          nz[j] += (d != 0.0) ? 1.0 : 0.0;
        }
        xp[xp_index(col, j)] = (float)d;
        t *= eta;
      }
    } else {
#pragma unroll
      for (int j = 0; j < BT_NV; ++j) xp[xp_index(col, j)] = 0.f;     // zero padding up to a multiple of 64 columns
    }
  }
  // block reduction of the BT_W values: wave butterflies, one LDS exchange, one barrier
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  auto put = [&](double v, int slot) {
    v = wave_sum(v);
    if (lane == 0) red[wave][slot] = v;
  };
#pragma unroll
  for (int j = 0; j < BT_NV; ++j) { put(gd[j], j); put(dd[j], BT_NV + j); put(nz[j], 2 * BT_NV + j); }
  put(g2, 3 * BT_NV);
  put(y2, 3 * BT_NV + 1);
  __syncthreads();
  if (threadIdx.x < BT_W)
    part[(int64_t)blockIdx.x * BT_W + threadIdx.x] =
        (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// q_part[wg][j] = sum over this workgroup's rows of (A_i . X_j - use_b * b_i)^2.
// Requirements (host-checked): n % 4 == 0, lda % 4 == 0, A 16-byte aligned, Xp zero-padded to n_pad = 64*ceil(n/64).
// STORE_R = true (multi-lambda gradient, gram_batch.hpp): the residuals themselves, rout[row][16 candidates] fp32,
// are kept for the second product G = A^T R.
template <int RB, bool STORE_R = false>
__global__ __launch_bounds__(BT_THREADS) void residual_batch_mfma_kernel(const float* __restrict__ A, int64_t lda,
                                                                        const float* __restrict__ b, int use_b,
                                                                        int64_t m, int n,
                                                                        const float* __restrict__ xp,
                                                                        int64_t groups_per_wg,
                                                                        double* __restrict__ q_part,
                                                                        float* __restrict__ rout = nullptr,
                                                                        const int* __restrict__ stopped = nullptr) {
  if (stopped != nullptr && *stopped != 0) return;          // parked pipeline (solver stopped / line search stalled)
  constexpr int ROWS = BT_ROWS * RB;              // RB 16-row blocks per wave share each candidate fragment read
  constexpr int A_LOADS = BT_A_LOADS * RB;
  __shared__ __attribute__((aligned(16))) float a_s[2][ROWS][BT_LDS_STRIDE];
  __shared__ __attribute__((aligned(16))) float x_s[2][BT_COLS * BT_NV];
  __shared__ double wsum[4][BT_NV];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t ngroups = (m + ROWS - 1) / ROWS;
  const int64_t g_lo = (int64_t)blockIdx.x * groups_per_wg;
  int64_t g_hi = g_lo + groups_per_wg;
  if (g_hi > ngroups) g_hi = ngroups;
  const int ktiles = (n + BT_COLS - 1) / BT_COLS;
  const int64_t ntiles = (g_hi > g_lo ? (g_hi - g_lo) : 0) * ktiles;

  // Two register sets: the loads of tile t+2 are issued while tile t is on the matrix cores and tile t+1 waits in
  // registers for its turn in LDS -> two 16 KiB tiles in flight per workgroup (x 3 workgroups per CU).
  f32x4 areg[2][A_LOADS];
  f32x4 xreg[2][BT_X_LOADS];
  auto load_tile = [&](int set, int64_t t) {
    const int64_t grp = g_lo + t / ktiles;
    const int kt = (int)(t % ktiles);
    const int64_t row0 = grp * ROWS;
    const int col0 = kt * BT_COLS;
#pragma unroll
    for (int u = 0; u < A_LOADS; ++u) {
      const int f = u * BT_THREADS + tid;
      int64_t row = row0 + f / BT_F4_ROW;
      int col = col0 + 4 * (f % BT_F4_ROW);
      if (row >= m) row = m - 1;                 // clamped rows are masked when the residual is formed
      if (col >= n) col = n - 4;                 // clamped columns meet zero rows of Xp
      areg[set][u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(A + row * lda + col));
    }
#pragma unroll
    for (int u = 0; u < BT_X_LOADS; ++u)
      xreg[set][u] = *reinterpret_cast<const f32x4*>(xp + (int64_t)kt * (BT_COLS * BT_NV) + 4 * (u * BT_THREADS + tid));
  };
  auto store_tile = [&](int set, int buf) {
#pragma unroll
    for (int u = 0; u < A_LOADS; ++u) {
      const int f = u * BT_THREADS + tid;
      *reinterpret_cast<f32x4*>(&a_s[buf][f / BT_F4_ROW][4 * (f % BT_F4_ROW)]) = areg[set][u];
    }
#pragma unroll
    for (int u = 0; u < BT_X_LOADS; ++u) *reinterpret_cast<f32x4*>(&x_s[buf][4 * (u * BT_THREADS + tid)]) = xreg[set][u];
  };
  // two accumulators per row block (even / odd MFMA steps): v_mfma_f32_16x16x4_f32 issues every 32 cycles but a
  // dependent one needs 40 - one chain would leave the pipe idle a fifth of the time
  f32x4 acc[RB], acc_odd[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) { acc[rb] = f32x4{0.f, 0.f, 0.f, 0.f}; acc_odd[rb] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  double qsum = 0.0;                              // this lane's candidate j = lane & 15, its 4 rows per row block
  auto compute_tile = [&](int buf, int64_t t) {
#pragma unroll
    for (int sub = 0; sub < BT_COLS / 16; ++sub) {
      const f32x4 x4 = *reinterpret_cast<const f32x4*>(&x_s[buf][(sub * 4 + (lane >> 4)) * 64 + (lane & 15) * 4]);
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        const f32x4 a4 =
            *reinterpret_cast<const f32x4*>(&a_s[buf][16 * (wave * RB + rb) + (lane & 15)][16 * sub + 4 * (lane >> 4)]);
        acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, x4.x, acc[rb], 0, 0, 0);
        acc_odd[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, x4.y, acc_odd[rb], 0, 0, 0);
        acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, x4.z, acc[rb], 0, 0, 0);
        acc_odd[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, x4.w, acc_odd[rb], 0, 0, 0);
      }
    }
    if ((t + 1) % ktiles == 0) {
      // row group complete: D[row = 4*(lane>>4)+reg][candidate = lane&15]
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        acc[rb] += acc_odd[rb];
        const int64_t row0 = (g_lo + t / ktiles) * ROWS + 16 * (wave * RB + rb) + 4 * (lane >> 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t row = row0 + r;
          if (row < m) {
            float v = acc[rb][r];
            if (use_b) v -= b[row];
            qsum += (double)v * (double)v;
            if constexpr (STORE_R) rout[row * BT_NV + (lane & 15)] = v;      // 16 lanes: one 64-byte row of R
          }
        }
        acc[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
        acc_odd[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  };

  // Straight-line pair loop: loads and LDS stores are unconditional (tile index clamped to the last tile: redundant
  // L2 hits at the tail), so the compiler can wait for a register set with a COUNTED vmcnt and leave the set it has
  // just issued in flight.  A conditional load_tile / store_tile forces vmcnt(0) (DESIGN.md "What the ISA showed").
  if (ntiles > 0) {
    const int64_t last = ntiles - 1;
    auto clampt = [&](int64_t t) { return t < last ? t : last; };
    load_tile(0, 0);
    store_tile(0, 0);
    load_tile(1, clampt(1));                      // tile 1 waits in register set 1
    __syncthreads();
    int64_t t = 0;
    for (; t + 2 <= ntiles; t += 2) {
      // even step: tile t in LDS buffer 0, tile t+1 in register set 1
      load_tile(0, clampt(t + 2));
      compute_tile(0, t);
      store_tile(1, 1);
      __syncthreads();
      // odd step: tile t+1 in LDS buffer 1, tile t+2 in register set 0
      load_tile(1, clampt(t + 3));
      compute_tile(1, t + 1);
      store_tile(0, 0);
      __syncthreads();
    }
    if (t < ntiles) compute_tile(0, t);   // odd count: the last tile sits in buffer 0
  }
  // lanes j, j+16, j+32, j+48 hold candidate j
  qsum += __shfl_xor(qsum, 16, 64);
  qsum += __shfl_xor(qsum, 32, 64);
  if (lane < BT_NV) wsum[wave][lane] = qsum;
  __syncthreads();
  if (tid < BT_NV) q_part[(int64_t)blockIdx.x * BT_NV + tid] = (wsum[0][tid] + wsum[1][tid]) + (wsum[2][tid] + wsum[3][tid]);
}


// =========================================================================================================
// bf16 A: the same batched residual on v_mfma_f32_16x16x32_bf16 (BASELINE config 5: "CDNA4 bf16 MFMA path").
//
// The MFMA needs bf16 operands on both sides, but rounding the candidate vectors to bf16 (8 mantissa bits) would put
// 4e-3 relative noise on ||A dlt||^2.  Each candidate is therefore split into THREE bf16 terms
//     d = hi + mid + lo,   hi = bf16(d), mid = bf16(d - hi), lo = bf16(d - hi - mid)      (24 mantissa bits)
// and the three products A*hi, A*mid, A*lo accumulate into the SAME fp32 D tile: fp32-equivalent accuracy at bf16
// MFMA rate (16 cycles per 16x16x32 step; 12 MFMAs per 16 KiB tile per wave against 16 f32 MFMAs of 32 cycles).
// Tile = 64 rows x 128 bf16 columns = the same 16 KiB / 256 B-per-row staging as the fp32 kernel.
//
// Candidate block layout ("Xq"), bf16:  for column k, candidate j, part p (0 hi, 1 mid, 2 lo)
//   Xq[ (((k/32)*3 + p)*4 + (k%32)/8) * 128 + j*8 + (k%8) ]
// i.e. per 32-column k-step and part a [q = 4][j = 16][e = 8] cube: one 16-byte read per lane, 1 KiB per wave.
// =========================================================================================================
constexpr int BQ_COLS = 128;                         // Xq is zero-padded to a multiple of this many columns
typedef short bf16x8 __attribute__((ext_vector_type(8)));

// Candidate generation for the bf16 path: same sums as fista_trial_batch_kernel, dlt_j written as three bf16 terms.
static __global__ __launch_bounds__(256) void fista_trial_batch_bf16_kernel(GradSrc gsrc, int n, int n_pad,
                                                                    const double* __restrict__ x_cur,
                                                                    const double* __restrict__ x_prev,
                                                                    const FistaScalars* __restrict__ scal,
                                                                    FistaParams prm, double t0, double eta, int nv,
                                                                    unsigned short* __restrict__ xq,
                                                                    double* __restrict__ part, int t_from_state = 0) {
  __shared__ double red[4][BT_W];
  if (t_from_state) {
    if (scal->stopped != 0) return;
    t0 = scal->tau;
  }
  const double beta = scal->beta;
  double gd[BT_NV], dd[BT_NV], nz[BT_NV];
#pragma unroll
  for (int j = 0; j < BT_NV; ++j) { gd[j] = 0.0; dd[j] = 0.0; nz[j] = 0.0; }
  double g2 = 0.0, y2 = 0.0;
  for (int col = blockIdx.x * 256 + threadIdx.x; col < n_pad; col += gridDim.x * 256) {
    double y = 0.0, gf = 0.0;
    if (col < n) {
      y = form_y(x_cur[col], x_prev[col], beta);
      gf = grad_at(gsrc, col);
      if (prm.prox_kind == PROX_L1 && prm.alpha2 > 0.0) gf += prm.alpha2 * y;
      g2 += gf * gf;
      y2 += y * y;
    }
    double t = t0;
#pragma unroll
    for (int j = 0; j < BT_NV; ++j) {
      double d = 0.0;
      if (col < n && j < nv) {
        const double v = y - t * gf;
        double xt = prm.alpha1 > 0.0 ? soft_threshold(v, t * prm.alpha1) : v;
        if (prm.prox_kind == PROX_ENET) xt *= 1.0 / (1.0 + t * prm.alpha2);
        d = xt - y;
        gd[j] += gf * d;
        dd[j] += d * d;
        nz[j] += (d != 0.0) ? 1.0 : 0.0;
      }
      const float df = (float)d;
      const unsigned short hi = f32_to_bf16_rn(df);
      const float r1 = df - bf16_to_f32(hi);
      const unsigned short mid = f32_to_bf16_rn(r1);
      const unsigned short lo = f32_to_bf16_rn(r1 - bf16_to_f32(mid));
      xq[xq_index(col, j, 0)] = hi;
      xq[xq_index(col, j, 1)] = mid;
      xq[xq_index(col, j, 2)] = lo;
      t *= eta;
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  auto put = [&](double v, int slot) {
    v = wave_sum(v);
    if (lane == 0) red[wave][slot] = v;
  };
#pragma unroll
  for (int j = 0; j < BT_NV; ++j) { put(gd[j], j); put(dd[j], BT_NV + j); put(nz[j], 2 * BT_NV + j); }
  put(g2, 3 * BT_NV);
  put(y2, 3 * BT_NV + 1);
  __syncthreads();
  if (threadIdx.x < BT_W)
    part[(int64_t)blockIdx.x * BT_W + threadIdx.x] =
        (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// Plain [n][16] fp32 candidates -> Xq (three bf16 terms each); stand-alone entry / tests.
static __global__ void xq_pack_kernel(const float* __restrict__ X, int n, int n_pad, int nv, unsigned short* __restrict__ xq) {
  for (int col = blockIdx.x * 256 + threadIdx.x; col < n_pad; col += gridDim.x * 256)
    for (int j = 0; j < BT_NV; ++j) {
      const float df = (col < n && j < nv) ? X[(int64_t)col * BT_NV + j] : 0.f;
      const unsigned short hi = f32_to_bf16_rn(df);
      const float r1 = df - bf16_to_f32(hi);
      const unsigned short mid = f32_to_bf16_rn(r1);
      xq[xq_index(col, j, 0)] = hi;
      xq[xq_index(col, j, 1)] = mid;
      xq[xq_index(col, j, 2)] = f32_to_bf16_rn(r1 - bf16_to_f32(mid));
    }
}

// q_part[wg][j] = sum over this workgroup's rows of (A_i . X_j - use_b*b_i)^2, A in bf16.
// RB = 16-row blocks per wave (the candidate fragments read from LDS are reused for RB row blocks: X is 96 bytes per
// column against 32 bytes of A per row block, so LDS traffic per byte of A falls with RB); COLS = bf16 columns per tile.
// Requirements: n % 8 == 0, lda % 8 == 0, A 16-byte aligned, Xq zero-padded to n_pad (a multiple of 128).
template <int RB, int COLS, bool STORE_R = false>
__global__ __launch_bounds__(BT_THREADS) void residual_batch_mfma_bf16_kernel(
    const bf16_t* __restrict__ A, int64_t lda, const float* __restrict__ b, int use_b, int64_t m, int n,
    const unsigned short* __restrict__ xq, int64_t groups_per_wg, double* __restrict__ q_part,
    float* __restrict__ rout = nullptr, const int* __restrict__ stopped = nullptr) {
  if (stopped != nullptr && *stopped != 0) return;
  constexpr int ROWS = 64 * RB;
  constexpr int STRIDE = COLS + 8;
  constexpr int CPR = COLS / 8;                               // 16-byte chunks per tile row
  constexpr int X_TILE = COLS * BT_NV * 3;
  constexpr int A_LOADS = ROWS * CPR / BT_THREADS;
  constexpr int X_CHUNKS = X_TILE / 8;
  constexpr int X_LOADS = (X_CHUNKS + BT_THREADS - 1) / BT_THREADS;
  __shared__ __attribute__((aligned(16))) unsigned short a_s[2][ROWS][STRIDE];
  __shared__ __attribute__((aligned(16))) unsigned short x_s[2][X_TILE];
  __shared__ double wsum[4][BT_NV];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t ngroups = (m + ROWS - 1) / ROWS;
  const int64_t g_lo = (int64_t)blockIdx.x * groups_per_wg;
  int64_t g_hi = g_lo + groups_per_wg;
  if (g_hi > ngroups) g_hi = ngroups;
  const int ktiles = (n + COLS - 1) / COLS;
  const int64_t ntiles = (g_hi > g_lo ? (g_hi - g_lo) : 0) * ktiles;

  u32x4 areg[2][A_LOADS];
  u32x4 xreg[2][X_LOADS];
  auto load_tile = [&](int set, int64_t t) {
    const int64_t grp = g_lo + t / ktiles;
    const int kt = (int)(t % ktiles);
    const int64_t row0 = grp * ROWS;
    const int col0 = kt * COLS;
#pragma unroll
    for (int u = 0; u < A_LOADS; ++u) {
      const int f = u * BT_THREADS + tid;
      int64_t row = row0 + f / CPR;
      int col = col0 + 8 * (f % CPR);
      if (row >= m) row = m - 1;
      if (col >= n) col = n - 8;                  // clamped columns meet zero rows of Xq
      areg[set][u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(A + row * lda + col));
    }
#pragma unroll
    for (int u = 0; u < X_LOADS; ++u) {
      int c = u * BT_THREADS + tid;
      if (c >= X_CHUNKS) c = X_CHUNKS - 1;        // branch-free; the surplus lanes do not store
      xreg[set][u] = *reinterpret_cast<const u32x4*>(xq + (int64_t)kt * X_TILE + 8 * c);
    }
  };
  auto store_tile = [&](int set, int buf) {
#pragma unroll
    for (int u = 0; u < A_LOADS; ++u) {
      const int f = u * BT_THREADS + tid;
      *reinterpret_cast<u32x4*>(&a_s[buf][f / CPR][8 * (f % CPR)]) = areg[set][u];
    }
#pragma unroll
    for (int u = 0; u < X_LOADS; ++u) {
      const int c = u * BT_THREADS + tid;
      if (c < X_CHUNKS) *reinterpret_cast<u32x4*>(&x_s[buf][8 * c]) = xreg[set][u];
    }
  };
  f32x4 acc[RB], acc_odd[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) { acc[rb] = f32x4{0.f, 0.f, 0.f, 0.f}; acc_odd[rb] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  double qsum = 0.0;
  auto compute_tile = [&](int buf, int64_t t) {
#pragma unroll
    for (int ks = 0; ks < COLS / 32; ++ks) {
      const int xo = (lane >> 4) * 128 + (lane & 15) * 8;
      const bf16x8 xh = *reinterpret_cast<const bf16x8*>(&x_s[buf][(ks * 3 + 0) * 512 + xo]);
      const bf16x8 xm = *reinterpret_cast<const bf16x8*>(&x_s[buf][(ks * 3 + 1) * 512 + xo]);
      const bf16x8 xl = *reinterpret_cast<const bf16x8*>(&x_s[buf][(ks * 3 + 2) * 512 + xo]);
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        const bf16x8 a8 =
            *reinterpret_cast<const bf16x8*>(&a_s[buf][16 * (wave * RB + rb) + (lane & 15)][32 * ks + 8 * (lane >> 4)]);
        acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, xh, acc[rb], 0, 0, 0);
        acc_odd[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, xm, acc_odd[rb], 0, 0, 0);
        acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, xl, acc[rb], 0, 0, 0);
      }
    }
    if ((t + 1) % ktiles == 0) {
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        acc[rb] += acc_odd[rb];
        const int64_t row0 = (g_lo + t / ktiles) * ROWS + 16 * (wave * RB + rb) + 4 * (lane >> 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t row = row0 + r;
          if (row < m) {
            float v = acc[rb][r];
            if (use_b) v -= b[row];
            qsum += (double)v * (double)v;
            if constexpr (STORE_R) rout[row * BT_NV + (lane & 15)] = v;
          }
        }
        acc[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
        acc_odd[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  };

  if (ntiles > 0) {
    const int64_t last = ntiles - 1;
    auto clampt = [&](int64_t t) { return t < last ? t : last; };
    load_tile(0, 0);
    store_tile(0, 0);
    load_tile(1, clampt(1));
    __syncthreads();
    int64_t t = 0;
    for (; t + 2 <= ntiles; t += 2) {
      load_tile(0, clampt(t + 2));
      compute_tile(0, t);
      store_tile(1, 1);
      __syncthreads();
      load_tile(1, clampt(t + 3));
      compute_tile(1, t + 1);
      store_tile(0, 0);
      __syncthreads();
    }
    if (t < ntiles) compute_tile(0, t);
  }
  qsum += __shfl_xor(qsum, 16, 64);
  qsum += __shfl_xor(qsum, 32, 64);
  if (lane < BT_NV) wsum[wave][lane] = qsum;
  __syncthreads();
  if (tid < BT_NV) q_part[(int64_t)blockIdx.x * BT_NV + tid] = (wsum[0][tid] + wsum[1][tid]) + (wsum[2][tid] + wsum[3][tid]);
}

}  // namespace fos
