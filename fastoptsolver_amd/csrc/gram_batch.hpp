// Multi-lambda gradient on the matrix cores: G[n x 16] = A^T (A Y - b 1^T) for 16 iterates at once (SURVEY.md 8f rank 3:
// "A read once for many alpha1" - a regularisation path advances 16 FISTA state machines in lockstep).
//
// The single-vector pass keeps y and the gradient slice of a whole row in one workgroup's registers; with 16 right-hand
// sides that state is n x 16 x 2 floats (1 MiB at n = 8192) - no workgroup can hold whole rows any more.  The pass is
// therefore two GEMM-shaped products over row PANELS small enough to stay in the 256 MiB Infinity Cache between them:
//   product 1  R[P x 16] = A_panel Y - b      residual_batch_mfma_kernel<.., STORE_R> (batch_trial.hpp): A from HBM
//   product 2  G[16 x n] (+)= R^T A_panel     gram_batch_mfma_kernel (this file): the panel again, from the Infinity Cache
// so A still crosses the HBM interface ONCE per iteration for all 16 weights.
//
// Product 2 on v_mfma_f32_16x16x4_f32 with M = lambda, K = row, N = column: lane l supplies R[row 4(l>>4)+c][lambda l&15]
// and A[row 4(l>>4)+c][column l&15] (four MFMAs per 16 rows), D holds G[lambda 4(l>>4)+r][column l&15]: 16 lanes write 64
// contiguous bytes of one slab row.  A workgroup owns a 64-column strip of a row split (4 waves x 16 columns), stages
// 64 x 64 tiles of A through LDS exactly like product 1 (coalesced 256-byte row segments, two register sets, straight-
// line pair loop) and converts bf16 storage to fp32 on the way in, so one kernel serves both element types: at 16
// right-hand sides the fp32 matrix pipe needs 32 flop per element, 40 % of its rate at the fp32 stream and 80 % at the
// bf16 stream (bf16 storage has its own product-2 kernel below, natively on the bf16 pipe).
// Slabs: slabs[split][lambda][n_stride]; panel 0 writes, later panels add (kernels of one stream: fixed order).
#pragma once
#include "batch_trial.hpp"

namespace fos {

constexpr int GB_ROWS = 64, GB_COLS = 64, GB_THREADS = 256;
constexpr int GB_ASTRIDE = GB_COLS + 4;        // floats: (4q rows apart) -> 16 banks apart, conflict-free operand reads
constexpr int GB_RSTRIDE = BT_NV + 4;

template <typename T, bool ACCUM>
__global__ __launch_bounds__(GB_THREADS) void gram_batch_mfma_kernel(const T* __restrict__ A, int64_t lda, int64_t m, int n,
                                                                    const float* __restrict__ R, int64_t rows_per_split,
                                                                    float* __restrict__ slabs, int64_t n_stride) {
  using Tr = ElemTraits<T>;
  constexpr int EPC = Tr::EPC;
  constexpr int CPR = GB_COLS / EPC;                              // 16-byte chunks per tile row
  constexpr int A_LOADS = GB_ROWS * CPR / GB_THREADS;             // 4 (fp32) / 2 (bf16)
  __shared__ __attribute__((aligned(16))) float a_s[2][GB_ROWS][GB_ASTRIDE];
  __shared__ __attribute__((aligned(16))) float r_s[2][GB_ROWS][GB_RSTRIDE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col0 = blockIdx.x * GB_COLS;
  const int64_t row_lo = (int64_t)blockIdx.y * rows_per_split;
  int64_t row_hi = row_lo + rows_per_split;
  if (row_hi > m) row_hi = m;
  const int64_t ntiles = row_hi > row_lo ? (row_hi - row_lo + GB_ROWS - 1) / GB_ROWS : 0;

  u32x4 areg[2][A_LOADS];
  f32x4 rreg[2];
  auto load_tile = [&](int set, int64_t t) {
    const int64_t row0 = row_lo + t * GB_ROWS;
#pragma unroll
    for (int u = 0; u < A_LOADS; ++u) {
      const int f = u * GB_THREADS + tid;
      int64_t row = row0 + f / CPR;
      int col = col0 + EPC * (f % CPR);
      if (row >= row_hi) row = row_hi - 1;       // clamped rows meet zero rows of R
      if (col >= n) col = n - EPC;               // clamped columns are not stored
      areg[set][u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(A + row * lda + col));
    }
    int64_t rrow = row0 + tid / 4;               // 64 rows x 16 floats: one float4 per thread
    const bool in = rrow < row_hi;
    if (!in) rrow = row_hi - 1;
    const f32x4 rv = *reinterpret_cast<const f32x4*>(R + rrow * BT_NV + 4 * (tid % 4));
    rreg[set] = in ? rv : f32x4{0.f, 0.f, 0.f, 0.f};
  };
  auto store_tile = [&](int set, int buf) {
#pragma unroll
    for (int u = 0; u < A_LOADS; ++u) {
      const int f = u * GB_THREADS + tid;
      float a[EPC];
      Tr::unpack(areg[set][u], a);
#pragma unroll
      for (int q = 0; q < EPC / 4; ++q)
        *reinterpret_cast<f32x4*>(&a_s[buf][f / CPR][EPC * (f % CPR) + 4 * q]) =
            f32x4{a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]};
    }
    *reinterpret_cast<f32x4*>(&r_s[buf][tid / 4][4 * (tid % 4)]) = rreg[set];
  };
  f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc_odd = {0.f, 0.f, 0.f, 0.f};
  auto compute_tile = [&](int buf) {
    const int kq = 4 * (lane >> 4), j = lane & 15;
#pragma unroll
    for (int ks = 0; ks < GB_ROWS / 16; ++ks) {
      const int r0 = 16 * ks + kq;
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(r_s[buf][r0][j], a_s[buf][r0][16 * wave + j], acc, 0, 0, 0);
      acc_odd = __builtin_amdgcn_mfma_f32_16x16x4f32(r_s[buf][r0 + 1][j], a_s[buf][r0 + 1][16 * wave + j], acc_odd, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(r_s[buf][r0 + 2][j], a_s[buf][r0 + 2][16 * wave + j], acc, 0, 0, 0);
      acc_odd = __builtin_amdgcn_mfma_f32_16x16x4f32(r_s[buf][r0 + 3][j], a_s[buf][r0 + 3][16 * wave + j], acc_odd, 0, 0, 0);
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
      compute_tile(0);
      store_tile(1, 1);
      __syncthreads();
      load_tile(1, clampt(t + 3));
      compute_tile(1);
      store_tile(0, 0);
      __syncthreads();
    }
    if (t < ntiles) compute_tile(0);
  }
  acc += acc_odd;
  const int col = col0 + 16 * wave + (lane & 15);
  if (col < n) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float* dst = slabs + ((int64_t)blockIdx.y * BT_NV + 4 * (lane >> 4) + r) * n_stride + col;
      if constexpr (ACCUM) *dst += acc[r];
      else *dst = acc[r];
    }
  }
}

// bf16 A: product 2 natively on v_mfma_f32_16x16x32_bf16 (K = 32 rows per instruction).  The A tile stays bf16 in LDS as
// it arrives ([row][col], 16-byte stores); the B operand of the MFMA needs 8 consecutive ROWS of one column per lane,
// i.e. the tile read transposed: two ds_read_b64_tr_b16 per operand (gfx950: a 16-lane group reads a 4-row x 16-column
// block and every lane receives one column of it).  R (fp32) is split into three bf16 terms when it is staged, stored
// transposed [term][lambda][row] so that the A operand (8 consecutive rows of one lambda) is one 16-byte read.
// 6 MFMAs of 16 cycles per 64 x 16 sub-tile against 16 of 32 cycles on the fp32 pipe: the product is HBM-bound again.
constexpr int GQ_COLS = 128;                    // columns per workgroup strip (4 waves x 32)
constexpr int GQ_ASTRIDE = GQ_COLS + 8;         // shorts: 272-byte rows (16-byte aligned, a multiple of 8 bytes for tr reads)
constexpr int GQ_RSTRIDE = GB_ROWS + 8;         // shorts
typedef short s16x4 __attribute__((ext_vector_type(4)));

template <bool ACCUM>
__global__ __launch_bounds__(GB_THREADS) void gram_batch_mfma_bf16_kernel(const bf16_t* __restrict__ A, int64_t lda, int64_t m,
                                                                         int n, const float* __restrict__ R,
                                                                         int64_t rows_per_split, float* __restrict__ slabs,
                                                                         int64_t n_stride) {
  constexpr int CPR = GQ_COLS / 8;                                // 16-byte chunks per tile row
  constexpr int A_LOADS = GB_ROWS * CPR / GB_THREADS;             // 4
  __shared__ __attribute__((aligned(16))) unsigned short a_s[2][GB_ROWS][GQ_ASTRIDE];
  __shared__ __attribute__((aligned(16))) unsigned short rT_s[2][3][BT_NV][GQ_RSTRIDE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col0 = blockIdx.x * GQ_COLS;
  const int64_t row_lo = (int64_t)blockIdx.y * rows_per_split;
  int64_t row_hi = row_lo + rows_per_split;
  if (row_hi > m) row_hi = m;
  const int64_t ntiles = row_hi > row_lo ? (row_hi - row_lo + GB_ROWS - 1) / GB_ROWS : 0;

  u32x4 areg[2][A_LOADS];
  f32x4 rreg[2];
  auto load_tile = [&](int set, int64_t t) {
    const int64_t row0 = row_lo + t * GB_ROWS;
#pragma unroll
    for (int u = 0; u < A_LOADS; ++u) {
      const int f = u * GB_THREADS + tid;
      int64_t row = row0 + f / CPR;
      int col = col0 + 8 * (f % CPR);
      if (row >= row_hi) row = row_hi - 1;       // clamped rows meet zero rows of R
      if (col >= n) col = n - 8;                 // clamped columns are not stored
      areg[set][u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(A + row * lda + col));
    }
    int64_t rrow = row0 + tid / 4;
    const bool in = rrow < row_hi;
    if (!in) rrow = row_hi - 1;
    const f32x4 rv = *reinterpret_cast<const f32x4*>(R + rrow * BT_NV + 4 * (tid % 4));
    rreg[set] = in ? rv : f32x4{0.f, 0.f, 0.f, 0.f};
  };
  auto store_tile = [&](int set, int buf) {
#pragma unroll
    for (int u = 0; u < A_LOADS; ++u) {
      const int f = u * GB_THREADS + tid;
      *reinterpret_cast<u32x4*>(&a_s[buf][f / CPR][8 * (f % CPR)]) = areg[set][u];
    }
    const int row = tid / 4, l4 = 4 * (tid % 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float v = rreg[set][e];
      const unsigned short hi = f32_to_bf16_rn(v);
      const float r1 = v - bf16_to_f32(hi);
      const unsigned short mid = f32_to_bf16_rn(r1);
      rT_s[buf][0][l4 + e][row] = hi;
      rT_s[buf][1][l4 + e][row] = mid;
      rT_s[buf][2][l4 + e][row] = f32_to_bf16_rn(r1 - bf16_to_f32(mid));
    }
  };
  f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  auto compute_tile = [&](int buf) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3;
#pragma unroll
    for (int ks = 0; ks < GB_ROWS / 32; ++ks) {
      bf16x8 ra[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) ra[p] = *reinterpret_cast<const bf16x8*>(&rT_s[buf][p][i][32 * ks + 8 * g]);
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        const int c0 = 32 * wave + 16 * ct;
        const int r0 = 32 * ks + 8 * g;
        typedef __attribute__((address_space(3))) s16x4* lds_v4;
        const s16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(&a_s[buf][r0 + q][c0 + 4 * pp]));
        const s16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(&a_s[buf][r0 + 4 + q][c0 + 4 * pp]));
        const bf16x8 b8 = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
#pragma unroll
        for (int p = 0; p < 3; ++p) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ra[p], b8, acc[ct], 0, 0, 0);
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
      compute_tile(0);
      store_tile(1, 1);
      __syncthreads();
      load_tile(1, clampt(t + 3));
      compute_tile(1);
      store_tile(0, 0);
      __syncthreads();
    }
    if (t < ntiles) compute_tile(0);
  }
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    const int col = col0 + 32 * wave + 16 * ct + (lane & 15);
    if (col < n) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float* dst = slabs + ((int64_t)blockIdx.y * BT_NV + 4 * (lane >> 4) + r) * n_stride + col;
        if constexpr (ACCUM) *dst += acc[ct][r];
        else *dst = acc[ct][r];
      }
    }
  }
}

// y_k of state machine `slot` into the candidate block of product 1: fp32 Xp layout, or three bf16 terms (Xq) for bf16 A.
static __global__ __launch_bounds__(256) void form_y_block_kernel(const double* __restrict__ x_cur, const double* __restrict__ x_prev,
                                                           double beta, int n, int slot, float* __restrict__ xp,
                                                           unsigned short* __restrict__ xq) {
  for (int col = blockIdx.x * 256 + threadIdx.x; col < n; col += gridDim.x * 256) {
    const float y = (float)form_y(x_cur[col], x_prev[col], beta);
    if (xp != nullptr) {
      xp[xp_index(col, slot)] = y;
    } else {
      const unsigned short hi = f32_to_bf16_rn(y);
      const float r1 = y - bf16_to_f32(hi);
      const unsigned short mid = f32_to_bf16_rn(r1);
      xq[xq_index(col, slot, 0)] = hi;
      xq[xq_index(col, slot, 1)] = mid;
      xq[xq_index(col, slot, 2)] = f32_to_bf16_rn(r1 - bf16_to_f32(mid));
    }
  }
}

}  // namespace fos
