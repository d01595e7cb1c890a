// Multi-lambda gradient with ONE read of A: G[n x 16] = A^T (A Y - b 1^T) for 16 iterates, both products in one launch.
//
// gram_batch.hpp forms the two GEMM-shaped products in two kernels, so A crosses the fabric twice per iteration (the
// Infinity Cache has HBM's bandwidth, a panel small enough for the L2s is too small for a launch): 678 us per iteration
// at config 2 against 309 us for one weight.  A workgroup cannot hold whole rows with 16 right-hand sides (the gradient
// slice alone is n x 16 floats), so here the rows are shared by a CLUSTER of workgroups:
//   * a cluster owns a block of rows and walks it in panels of 16 rows; member j owns the 1024-column strip j of every
//     panel (CS = n / 1024 members, 256 / CS clusters) and keeps, for the whole launch, its 1024 x 16 strip of Y as MFMA
//     operands in registers (64 VGPRs) and its 16 x 1024 strip of G as MFMA accumulators (64 VGPRs);
//   * per panel the member stages its 16 x 1024 tile of A in LDS ONCE (64 KiB, coalesced 16-byte loads, two LDS buffers
//     plus one tile in flight in registers) and uses it twice:
//       product 1   partial R = tile . Y_strip   (16 x 16, v_mfma_f32_16x16x4_f32, K = the strip's columns)
//       hand-off    the members' partials are summed -> R = A_panel Y - b   (16 rows x 16 weights)
//       product 2   G_strip += R^T . tile        (K = the panel's 16 rows)
//     The D fragment of product 1 IS the A-operand fragment of product 2 (row 4(l>>4)+c, weight l&15): no transposition.
//   * the hand-off is the only communication: 1 KiB per member and panel through L2 (MI355X_MICROARCH.md, "hand-offs
//     measured with sc1 loads", first row: sc1 stores by one wave, s_waitcnt vmcnt(0), sc1 flag store by one lane; the
//     consumer polls the flags with sc1 loads, then loads the bytes with sc1 loads; one workgroup per CU).  It costs
//     ~1.5 us of latency per panel (tools/cluster_probe.hip), so a FIFTH wave does nothing else and product 2 runs one
//     panel behind product 1: the matrix-core waves never wait for a flag, only at workgroup barriers.
//   * members of a cluster sit on one XCD (workgroup i runs on XCD i % 8), so the partials travel through that XCD's L2;
//     correctness does not depend on it (sc1 accesses are coherent across XCDs), only 0.3 us per panel.
// All members must be resident at once: the kernel is launched cooperatively (hipLaunchCooperativeKernel refuses a grid
// that does not fit), every wait is bounded and a timeout parks the workgroup and raises *error.
// Sums: product 1 in MFMA k-order per wave, the 4 waves and then the CS members in index order; product 2 in panel
// order -> deterministic, bit-reproducible.
#pragma once
#include "batch_trial.hpp"

namespace fos {

constexpr int CP_ROWS = 16;                 // rows per panel: MFMA M of product 1, K of product 2
constexpr int CP_W = 1024;                  // columns per strip (4 waves x 16 subtiles of 16)
constexpr int CP_ASTRIDE = CP_W + 4;        // floats: rows 4 banks apart
constexpr int CP_THREADS = 320;             // 4 matrix-core waves + the hand-off wave
constexpr int CP_SLOTS = 4;                 // ring of partials per member (a member runs at most 2 panels ahead)
constexpr int CP_FLAG_STRIDE = 32;          // unsigned: one 128-byte line per flag
constexpr unsigned CP_SPIN_LIMIT = 1u << 18;
constexpr size_t CP_LDS_BYTES = (size_t)2 * CP_ROWS * CP_ASTRIDE * 4 + 4 * 64 * 16 + 2 * 64 * 16 + 16;

typedef unsigned long long cp_u64;
__device__ inline void cp_store_sc1(cp_u64* p, cp_u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline cp_u64 cp_load_sc1(const cp_u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline cp_u64 cp_pack(float lo, float hi) { return ((cp_u64)__float_as_uint(hi) << 32) | __float_as_uint(lo); }

// grid = nclusters * CS workgroups of CP_THREADS; dynamic LDS = CP_LDS_BYTES; rows_per_cluster a multiple of CP_ROWS.
// xp: the 16 iterates in the Xp layout (reduce_update.hpp), zero-padded to n_pad columns.  n % 4 == 0, lda % 4 == 0.
// slabs[cluster][16][n_stride]: the cluster's partial gradients (its members write disjoint strips).
// MODE (tools/cluster_bench.hip prices the parts; the library runs MODE 0): bit 0 = no hand-off (R = this member's own
// partial), bit 1 = no matrix-core work, bit 2 = no loads of A after the first two tiles.
template <int CS, int MODE = 0>
__global__ __launch_bounds__(CP_THREADS) void cluster_pass_kernel(const float* __restrict__ A, int64_t lda,
                                                                 const float* __restrict__ b, int64_t m, int n, int n_pad,
                                                                 const float* __restrict__ xp, int64_t rows_per_cluster,
                                                                 int xcd_aware, float* __restrict__ xchg,
                                                                 unsigned* __restrict__ flags, unsigned epoch,
                                                                 float* __restrict__ slabs, int64_t n_stride,
                                                                 int* __restrict__ error) {
  extern __shared__ __attribute__((aligned(16))) char cp_lds[];
  float (*a_s)[CP_ROWS][CP_ASTRIDE] = reinterpret_cast<float (*)[CP_ROWS][CP_ASTRIDE]>(cp_lds);
  f32x4 (*red)[64] = reinterpret_cast<f32x4 (*)[64]>(cp_lds + (size_t)2 * CP_ROWS * CP_ASTRIDE * 4);
  f32x4 (*r_s)[64] = reinterpret_cast<f32x4 (*)[64]>(cp_lds + (size_t)2 * CP_ROWS * CP_ASTRIDE * 4 + 4 * 64 * 16);
  int* abort_s = reinterpret_cast<int*>(cp_lds + (size_t)2 * CP_ROWS * CP_ASTRIDE * 4 + 4 * 64 * 16 + 2 * 64 * 16);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool mc = wave < 4;                  // matrix-core wave (wave 4: hand-off)
  int cluster, member;
  if (xcd_aware) {
    const int xcd = blockIdx.x % 8, idx = blockIdx.x / 8;
    cluster = xcd + 8 * (idx / CS);
    member = idx % CS;
  } else {
    cluster = blockIdx.x / CS;
    member = blockIdx.x % CS;
  }
  const int col_strip = member * CP_W;
  const int64_t row_lo = (int64_t)cluster * rows_per_cluster;
  const int npanels = (int)(rows_per_cluster / CP_ROWS);
  if (tid == 0) *abort_s = 0;

  // ---- matrix-core waves: operands that live in registers for the whole launch ---------------------------------
  f32x4 xr[16];                              // Y strip: subtile s of this wave, B operand of product 1
  f32x4 G[16];                               // gradient strip: 16 weights x 16 columns per subtile
  f32x4 areg[CP_ROWS];                       // the tile in flight (one 16-byte chunk of every row)
  const int wcol = col_strip + 256 * (wave & 3);      // this wave's 256 columns
  if (mc) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int col = wcol + 16 * s;
      const f32x4 v = *reinterpret_cast<const f32x4*>(xp + ((int64_t)((col < n_pad ? col : 0) / 16) * 4 + (lane >> 4)) * 64 +
                                                      (lane & 15) * 4);
      xr[s] = col < n_pad ? v : f32x4{0.f, 0.f, 0.f, 0.f};
      G[s] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  int lcol = col_strip + 4 * tid;            // this thread's chunk of every tile row (matrix-core waves: tid < 256)
  if (lcol >= n) lcol = n - 4;               // clamped columns meet zero entries of Y and are never stored
  auto load_tile = [&](int pnl) {
    if ((MODE & 4) && pnl >= 2) return;
    if (pnl >= npanels) pnl = npanels - 1;
#pragma unroll
    for (int u = 0; u < CP_ROWS; ++u) {
      int64_t row = row_lo + (int64_t)pnl * CP_ROWS + u;
      if (row >= m) row = m - 1;             // clamped rows meet zero rows of R
      areg[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(A + row * lda + lcol));
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int u = 0; u < CP_ROWS; ++u) *reinterpret_cast<f32x4*>(&a_s[buf][u][4 * tid]) = areg[u];
  };

  if (mc) {
    load_tile(0);
    store_tile(0);
    load_tile(1);
  }
  __syncthreads();

  for (int i = 0; i <= npanels; ++i) {
    // ---- phase 1: product 1 of panel i | hand-off wave: gather panel i-1 ----------------------------------------
    if (mc) {
      if (i < npanels) {
        const int buf = i & 1;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc_odd = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < ((MODE & 2) ? 1 : 16); ++s) {
          const f32x4 a4 = *reinterpret_cast<const f32x4*>(&a_s[buf][lane & 15][256 * wave + 16 * s + 4 * (lane >> 4)]);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, xr[s].x, acc, 0, 0, 0);
          acc_odd = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, xr[s].y, acc_odd, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, xr[s].z, acc, 0, 0, 0);
          acc_odd = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, xr[s].w, acc_odd, 0, 0, 0);
        }
        red[wave][lane] = acc + acc_odd;
      }
    } else if (i >= 1 && (MODE & 1)) {
      r_s[(i - 1) & 1][lane] = red[0][lane];           // (stale by design: pricing switch only)
    } else if (i >= 1) {
      const int q = i - 1;
      bool ok = true;
      if (lane < CS) {
        const unsigned* f = flags + ((size_t)cluster * CS + lane) * CP_FLAG_STRIDE;
        unsigned spins = 0;
        while ((int)(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - epoch) < q + 1) {
          if (++spins > CP_SPIN_LIMIT) { ok = false; break; }
          __builtin_amdgcn_s_sleep(1);
        }
      }
      if (__ballot(!ok) != 0) {
        if (lane == 0) { *abort_s = 1; *error = 1; }
      } else {
        // the members' partials, summed in index order; at most 8 members' loads in flight (registers: the matrix-core
        // waves' operands are live here too)
        constexpr int GC = CS < 8 ? CS : 8;
        f32x4 r = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int j0 = 0; j0 < CS; j0 += GC) {
          cp_u64 lo[GC], hi[GC];
#pragma unroll
          for (int j = 0; j < GC; ++j) {
            const cp_u64* src = reinterpret_cast<const cp_u64*>(
                xchg + (((size_t)cluster * CP_SLOTS + (q % CP_SLOTS)) * CS + j0 + j) * 256 + lane * 4);
            lo[j] = cp_load_sc1(src);
            hi[j] = cp_load_sc1(src + 1);
          }
#pragma unroll
          for (int j = 0; j < GC; ++j) {
            r.x += __uint_as_float((unsigned)lo[j]); r.y += __uint_as_float((unsigned)(lo[j] >> 32));
            r.z += __uint_as_float((unsigned)hi[j]); r.w += __uint_as_float((unsigned)(hi[j] >> 32));
          }
        }
        // fragment element c of lane l = row 4(l>>4)+c of the panel, weight l&15
        const int64_t row0 = row_lo + (int64_t)q * CP_ROWS + 4 * (lane >> 4);
        float rv[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int64_t row = row0 + c;
          rv[c] = row < m ? (b != nullptr ? rv[c] - b[row] : rv[c]) : 0.f;
        }
        r_s[q & 1][lane] = f32x4{rv[0], rv[1], rv[2], rv[3]};
      }
    }
    __syncthreads();                                   // A: red(i) and R(i-1) are in LDS
    if (*abort_s) break;
    // ---- phase 2: product 2 of panel i-1 | hand-off wave: publish the partial of panel i -------------------------
    if (mc) {
      if (i >= 1) {
        const int buf = (i - 1) & 1;
        const f32x4 r4 = r_s[(i - 1) & 1][lane];
        const float rc[4] = {r4.x, r4.y, r4.z, r4.w};
#pragma unroll
        for (int c = 0; c < ((MODE & 2) ? 1 : 4); ++c) {
#pragma unroll
          for (int t = 0; t < 16; ++t)
            G[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(rc[c], a_s[buf][4 * (lane >> 4) + c][256 * wave + 16 * t + (lane & 15)],
                                                        G[t], 0, 0, 0);
        }
      }
    } else if (i < npanels && !(MODE & 1)) {
      const f32x4 s = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
      cp_u64* dst = reinterpret_cast<cp_u64*>(xchg + (((size_t)cluster * CP_SLOTS + (i % CP_SLOTS)) * CS + member) * 256 + lane * 4);
      cp_store_sc1(dst, cp_pack(s.x, s.y));
      cp_store_sc1(dst + 1, cp_pack(s.z, s.w));
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0)
        __hip_atomic_store(flags + ((size_t)cluster * CS + member) * CP_FLAG_STRIDE, epoch + (unsigned)i + 1u, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();                                   // B: the buffer of panel i-1 and red are free
    // ---- phase 3: tile i+1 moves from registers to LDS, tile i+2 is requested ------------------------------------
    if (mc && i + 1 < npanels) {
      store_tile((i + 1) & 1);
      load_tile(i + 2);
    }
    __syncthreads();                                   // C: tile i+1 is in LDS
  }

  if (mc && *abort_s == 0) {
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int col = wcol + 16 * t + (lane & 15);
      if (col < n) {
        const float gv[4] = {G[t].x, G[t].y, G[t].z, G[t].w};
#pragma unroll
        for (int r = 0; r < 4; ++r) slabs[((int64_t)cluster * BT_NV + 4 * (lane >> 4) + r) * n_stride + col] = gv[r];
      }
    }
  }
}

}  // namespace fos
