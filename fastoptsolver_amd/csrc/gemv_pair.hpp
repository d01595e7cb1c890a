// Single-pass GEMV pair for gfx950:  slab[w] = sum_{rows i of workgroup w} A_i^T (A_i . y - b_i)
//
// Spec (not reference code; the reference is NumPy):  grad = A.T @ (A @ y - b)
//   iterative_solvers.py:173 (fista), :292 (fista_delta), :54 (power iteration, b = 0), lbfgs.py:46-48 (fg).
//
// Design (HBM-bound, 1 flop/byte fp32): every A element is read from HBM exactly once.  A workgroup owns
// whole rows.  Thread t owns the same columns in every row (16-byte chunks, chunk c of thread t starts at
// column (c*THREADS + t) * EPC), so its slice of y and of the running gradient live in VGPRs for the whole
// kernel.  Per step the workgroup takes R rows: partial dots per thread -> wave reduction -> one LDS exchange
// between the waves -> r_i = A_i.y - b_i known to every thread -> g += A_i * r_i from the SAME registers.
// The next R rows are already in flight (second register tile) while this happens.  Each workgroup finally
// writes its n-float partial gradient ("slab"); a second tiny kernel (reduce_update.hpp) sums the slabs in
// fixed order (deterministic, no float atomics) and applies prox + momentum.
//
// Why VALU and not MFMA for fp32 (DESIGN.md "MFMA analysis"): with one right-hand side a 16x16x4 f32 MFMA
// retires 64 A elements per 32 cycles per SIMD = 16 B/clk/CU for the two products, v_fma_f32 retires them in
// 4 cycles (256 B/clk/CU); the HBM stream needs ~13 B/clk/CU.  MFMA would run at ~80 % pipe utilisation just to
// keep up and its 16x16 / 32x32 accumulator tile wastes 15/16 of the registers this design spends on in-flight rows.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fos {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

// a (an element of A, exact in either type) times y plus acc in the accumulator's type
__device__ inline float fma_acc(float a, float y, float acc) { return fmaf(a, y, acc); }
__device__ inline double fma_acc(float a, double y, double acc) { return fma((double)a, y, acc); }

__device__ inline double form_y(double xc, double xp, double beta) { return xc + beta * (xc - xp); }

// Where the kernel takes y from.  The FISTA iterate state (x_k, x_{k-1}) is kept in fp64 on the device: the
// n-vectors are negligible traffic, and an fp32 state alone costs 1e-4 of parity on ill-conditioned data
// (DESIGN.md "Precision").  y_k = x_k + beta (x_k - x_prev) is formed in fp64 and rounded ONCE to fp32 for the
// pass over A.
struct YSource {
  const float* y;        // explicit y (power iteration, L-BFGS, trial points) or nullptr
  const double* x_cur;   // x_k
  const double* x_prev;  // x_{k-1}
  const double* beta;    // device scalar, or nullptr: use beta_val (host-driven momentum, see fos_fista_run)
  const int* stopped;    // device flag: non-zero -> kernel is a no-op (solver already stopped)
  double beta_val;
  const double* yd;      // explicit y in fp64 (L-BFGS iterate); used when y == nullptr and x_cur == nullptr
  // Column-blocked passes over rows too wide for one workgroup's registers (fos_plan.hip "column blocks"); read by the
  // CB = true instantiations of gemv_pair_kernel only:
  float* res_out;        // nullable: the NEGATED residual of every row, res_out[i] = (res_accum ? res_out[i] : 0) - s_i
  int res_accum;
  int64_t slab_stride;   // floats between consecutive slab rows (0 = n): a block writes its columns of full-width slabs
  // nullable: the first thread of the grid leaves the constant-rate wall clock here when the kernel starts (the L-BFGS
  // driver times its evaluations with it instead of hipEvents or a stamp kernel of its own: each costs ~5 us of stream time)
  unsigned long long* t_stamp;
};

__device__ inline double source_beta(const YSource& ys) {
  return (ys.y != nullptr || ys.x_cur == nullptr) ? 0.0 : (ys.beta != nullptr ? *ys.beta : ys.beta_val);
}
// y_j in fp64 whatever the source
__device__ inline double source_y(const YSource& ys, int64_t j, double beta) {
  if (ys.y != nullptr) return (double)ys.y[j];
  if (ys.x_cur != nullptr) return form_y(ys.x_cur[j], ys.x_prev[j], beta);
  return ys.yd[j];
}



template <typename T> struct ElemTraits;
template <> struct ElemTraits<float> {
  static constexpr int EPC = 4;   // elements per 16-byte chunk
  __device__ static inline void unpack(const u32x4& raw, float (&f)[4]) {
    f[0] = __uint_as_float(raw.x); f[1] = __uint_as_float(raw.y);
    f[2] = __uint_as_float(raw.z); f[3] = __uint_as_float(raw.w);
  }
};
struct bf16_t { uint16_t bits; };
template <> struct ElemTraits<bf16_t> {
  static constexpr int EPC = 8;
  __device__ static inline void unpack(const u32x4& raw, float (&f)[8]) {
    // bf16 -> f32 is a 16-bit shift: exact.
    f[0] = __uint_as_float(raw.x << 16); f[1] = __uint_as_float(raw.x & 0xffff0000u);
    f[2] = __uint_as_float(raw.y << 16); f[3] = __uint_as_float(raw.y & 0xffff0000u);
    f[4] = __uint_as_float(raw.z << 16); f[5] = __uint_as_float(raw.z & 0xffff0000u);
    f[6] = __uint_as_float(raw.w << 16); f[7] = __uint_as_float(raw.w & 0xffff0000u);
  }
};

template <bool NT>
__device__ inline u32x4 load16(const void* p) {
  if constexpr (NT) return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
  else return *reinterpret_cast<const u32x4*>(p);
}

// Wave-wide fp64 sums on the DPP path (no LDS crossbar), for the latency-bound kernels (resident loop, L-BFGS two-loop):
// a __shfl_xor of a double is two ds_bpermute round trips of ~150 cycles, and the resident loop's reductions were 12
// such dependent round trips per iteration - most of its 4-5 us.  Here each step is two v_mov_b32_dpp and one
// v_add_f64: row_shr 1/2/4/8 leave every 16-lane row's total in its lane 15,
// row_bcast:15 (rows 1, 3) and row_bcast:31 (rows 2, 3) carry the totals up to lane 63, v_readlane broadcasts it.
// Lanes without a DPP source receive 0.  Fixed order -> deterministic.
template <int CTRL, int ROW_MASK>
__device__ inline double dpp_fetch(double v) {
  const long long bits = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)bits, CTRL, ROW_MASK, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(bits >> 32), CTRL, ROW_MASK, 0xF, false);
  return __longlong_as_double(((long long)hi << 32) | (long long)(unsigned)lo);
}
template <int N>
__device__ inline void wave_sum_n(double (&v)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] += dpp_fetch<0x111, 0xF>(v[i]);     // row_shr:1
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] += dpp_fetch<0x112, 0xF>(v[i]);     // row_shr:2
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] += dpp_fetch<0x114, 0xF>(v[i]);     // row_shr:4
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] += dpp_fetch<0x118, 0xF>(v[i]);     // row_shr:8
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] += dpp_fetch<0x142, 0xA>(v[i]);     // row_bcast:15 into rows 1 and 3
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] += dpp_fetch<0x143, 0xC>(v[i]);     // row_bcast:31 into rows 2 and 3
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const long long bits = __double_as_longlong(v[i]);
    const int lo = __builtin_amdgcn_readlane((int)bits, 63), hi = __builtin_amdgcn_readlane((int)(bits >> 32), 63);
    v[i] = __longlong_as_double(((long long)hi << 32) | (long long)(unsigned)lo);
  }
}
__device__ inline double wave_sum_dpp(double x) {
  double v[1] = {x};
  wave_sum_n(v);
  return v[0];
}

// Sum over the 64 lanes of a wave; every lane gets the total (fixed order -> deterministic).  All lanes must be active.
__device__ inline float wave_sum(float v) {
  // same DPP ladder in fp32 (one v_mov_b32_dpp + v_add_f32 per step instead of a ds_bpermute round trip): measured
  // 0.5 % on the default streaming geometries, 2-7 % on the latency-sensitive ones (tools/kbench, same box, interleaved)
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xF, 0xF, false));   // row_shr:1
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x112, 0xF, 0xF, false));   // row_shr:2
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x114, 0xF, 0xF, false));   // row_shr:4
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x118, 0xF, 0xF, false));   // row_shr:8
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xA, 0xF, false));   // row_bcast:15
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xC, 0xF, false));   // row_bcast:31
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ inline double wave_sum(double v) { return wave_sum_dpp(v); }   // fp64: DPP path (see above)

// T: element type of A.  THREADS: workgroup size.  K: 16-byte chunks per thread per row.  R: rows per step.
// Requirements (checked on the host): n % EPC == 0, lda % EPC == 0, A 16-byte aligned, n <= K*THREADS*EPC.
// Rows of workgroup w: [w*rows_per_wg, min(m, (w+1)*rows_per_wg)).
// WITH_G = false gives the residual-only pass (K5: ||A x - b||^2 for objectives / Armijo trials).
// NBUF register tiles (NBUF-1 steps in flight behind the one being consumed).  IL = true deals rows to the
// workgroups round-robin in R-row groups (all CUs stream one contiguous window) instead of one contiguous
// block per workgroup; rows_per_wg is then ignored.
// DUAL = true (FISTA source only) also dots every row with x_k itself and returns rr2_part[w] = sum (A_i.x_k - b_i)^2:
// the history objective f(x_k) (iterative_solvers.py:225-230, :321) comes out of the SAME pass over A that produces
// the gradient at y_k, instead of the extra pass the reference pays per iteration.
// DRAIN = true waits for ALL outstanding loads (also the tile just prefetched) before each consume: a bursty
// request pattern that measured 2-3 % faster than the continuously overlapped one for 64 KiB rows (n = 16384 fp32).
// ACC = double: the L-BFGS `fg` form (lbfgs.py:43-54 through SciPy's float64 optimiser).  y is taken in fp64 and never
// rounded, every product a_ij*y_j and a_ij*r_i is formed in fp64 (exact: 24 x 53 bits round once in the fma), row dots,
// the wave / workgroup reductions and the gradient slice accumulate in fp64 and the slab is written as doubles.  The pass
// stays HBM-bound: per 16-byte chunk it costs 4 v_cvt_f64_f32 + 8 v_fma_f64 against ~5 cycles of HBM time per CU
// (DESIGN.md "L-BFGS in fp64"); registers, not VALU, are what it pays with (yv and gv double in size).
// YLDS = true (ACC = double only) keeps y in LDS instead (THREADS*K*EPC doubles of dynamic shared memory, up to 128 KiB
// of the CU's 160 KiB): the geometries for 64 KiB rows would otherwise need y + gradient slice = 256 KiB of the CU's
// 512 KiB register file plus two row tiles, and spill.  LDS read traffic is 8 bytes per element of A, ~26 B/clk/CU.
// CB = true is the column-block instantiation (rows wider than any geometry, fos_plan.hip "column blocks"): it also
// stores / accumulates the negated residual of every row (YSource::res_out) and honours YSource::slab_stride.  Those are
// COMPILE-TIME so that the streaming instantiations' steady-state loop carries no exec-masked store block between the
// post-barrier LDS reads and the next tile's loads (round 2 had them as runtime fields: +52 ISA lines, +10 branches in
// the cfg2 loop, and the in-loop headline was 4.5 % slower; tests/test_hot_loop_isa.py guards the loop now).
template <typename T, int THREADS, int K, int R, bool NT, int MINW, bool WITH_G = true, int NBUF = 2, bool IL = false,
          bool DUAL = false, bool DRAIN = false, typename ACC = float, bool YLDS = false, bool CB = false, bool SKEW = false, bool KEEPCVT = false>
__global__ __launch_bounds__(THREADS, MINW) void gemv_pair_kernel(
    const T* __restrict__ A, int64_t lda, const float* __restrict__ b, int64_t m, int n, YSource ys,
    int64_t rows_per_wg, ACC* __restrict__ slabs, double* __restrict__ rr_part, double* __restrict__ rr2_part) {
  using Tr = ElemTraits<T>;
  constexpr int EPC = Tr::EPC;
  constexpr int NW = THREADS / 64;
  constexpr int NV = DUAL ? 2 : 1;
  __shared__ ACC red[2][R * NV][NW];
  extern __shared__ __attribute__((aligned(16))) double y_lds[];      // YLDS only: THREADS*K*EPC doubles
  static_assert(!YLDS || (sizeof(ACC) == 8 && !DUAL), "YLDS is the fp64 form without DUAL");

  if (ys.stopped != nullptr && *ys.stopped != 0) return;
  if (ys.t_stamp != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *ys.t_stamp = wall_clock64();

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int64_t row_lo = IL ? 0 : (int64_t)blockIdx.x * rows_per_wg;
  int64_t row_hi = IL ? m : row_lo + rows_per_wg;
  if (row_hi > m) row_hi = m;

  // ---- prologue: this thread's slice of y, straight into registers --------------------------------
  ACC yv[YLDS ? 1 : K][EPC];
  ACC gv[K][EPC];
  ACC xv[DUAL ? K : 1][EPC];
  bool live[K];
  const double beta = source_beta(ys);
#pragma unroll
  for (int c = 0; c < K; ++c) {
    const int col = (c * THREADS + tid) * EPC;
    live[c] = col < n;
#pragma unroll
    for (int e = 0; e < EPC; ++e) { gv[c][e] = (ACC)0; if constexpr (!YLDS) yv[c][e] = (ACC)0; }
    if constexpr (DUAL) {
#pragma unroll
      for (int e = 0; e < EPC; ++e) xv[c][e] = live[c] ? (ACC)ys.x_cur[col + e] : (ACC)0;
    }
    if constexpr (YLDS) {                                    // dead chunks read zeros
#pragma unroll
      for (int e = 0; e < EPC; ++e) y_lds[col + e] = live[c] ? source_y(ys, col + e, beta) : 0.0;
    } else if (live[c]) {
#pragma unroll
      for (int q = 0; q < EPC / 4; ++q) {
        if (ys.y != nullptr) {
          const f32x4 t = *reinterpret_cast<const f32x4*>(ys.y + col + 4 * q);
          yv[c][4 * q + 0] = t.x; yv[c][4 * q + 1] = t.y; yv[c][4 * q + 2] = t.z; yv[c][4 * q + 3] = t.w;
        } else {
          // same form_y as the update kernel -> both kernels see the same y_k
#pragma unroll
          for (int e = 0; e < 4; ++e) yv[c][4 * q + e] = (ACC)source_y(ys, col + 4 * q + e, beta);
        }
      }
    }
  }

  if constexpr (YLDS) __syncthreads();      // (every thread wrote only the y entries it reads itself; kept for clarity)

  // Branch-free loads: a chunk beyond n reads this thread's chunk 0 instead (always inside the row when the
  // workgroup has any live chunk at all; its y slice is zero and its gradient slice is never stored).  Predicating
  // the loads instead makes hipcc wait vmcnt(0) before every consume, i.e. also for the tile just prefetched.
  // Addressing: uniform row base (SGPR pair) + loop-invariant 32-bit per-lane offset -> saddr+voffset loads, no
  // 64-bit address VGPRs per load.
  const bool any_live = live[0];
  unsigned voff[K];
#pragma unroll
  for (int c = 0; c < K; ++c) voff[c] = (any_live ? (unsigned)tid * 16u : 0u) + (live[c] ? (unsigned)c * THREADS * 16u : 0u);

  const int64_t nrows = row_hi - row_lo;           // may be <= 0 for trailing workgroups
  const int64_t group = (int64_t)R * (IL ? gridDim.x : 1);   // rows between consecutive steps of this workgroup
  const int64_t first = IL ? (int64_t)blockIdx.x * R : 0;
  const int64_t nsteps = nrows > first ? (nrows - first + group - 1) / group : 0;
  // SKEW: workgroup w walks its block starting w/gridDim of the way through it (and wraps): the workgroups' blocks start
  // a power-of-two number of bytes apart, so without it all CUs sit at the same offset of their block at the same time
  // - the same DRAM banks of every channel, different pages.
  const int64_t skew = SKEW && !IL ? (nsteps * (int64_t)blockIdx.x) / (int64_t)gridDim.x : 0;
  auto rot = [&](int64_t step) -> int64_t {
    if constexpr (!SKEW) return step;
    int64_t t = step + skew;
    if (t >= nsteps) t -= nsteps;
    if (t >= nsteps) t = nsteps - 1;        // prefetches past the end (clamped re-reads)
    return t;
  };
  const char* base = reinterpret_cast<const char*>(A);   // threads with no live chunk at all read column 0
  const int64_t row_bytes = lda * (int64_t)sizeof(T);
  double rr = 0.0, rr2 = 0.0;

  u32x4 tile[NBUF][R][K];
  float bval[NBUF][R];
  // b_i travels with its row: loaded at issue time so that it is OLDER than the younger prefetches (vmcnt counts in
  // order - a b load issued at consume time could only be waited for with vmcnt(0), draining the whole pipeline).
  const float* b_src = b != nullptr ? b : reinterpret_cast<const float*>(A);   // dummy source, discarded by select
  auto issue = [&](int buf, int64_t step) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      int64_t row = row_lo + first + rot(step) * group + r;
      if (row >= row_hi) row = row_hi - 1;          // clamp: loaded but weighted by zero below
      bval[buf][r] = b_src[b != nullptr ? row : 0];
      const char* rp = base + row * row_bytes;
#pragma unroll
      for (int c = 0; c < K; ++c) tile[buf][r][c] = load16<NT>(rp + voff[c]);
    }
  };
  auto consume = [&](int buf, int64_t step) {
    ACC part[R * NV];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      ACC acc0 = (ACC)0, acc1 = (ACC)0, acc2 = (ACC)0, acc3 = (ACC)0;
#pragma unroll
      for (int c = 0; c < K; ++c) {
        float a[EPC];
        Tr::unpack(tile[buf][r][c], a);
        ACC yl[EPC];
        if constexpr (YLDS) {
#pragma unroll
          for (int q = 0; q < EPC / 2; ++q) {
            const f64x2 t = *reinterpret_cast<const f64x2*>(y_lds + (c * THREADS + tid) * EPC + 2 * q);
            yl[2 * q] = (ACC)t.x; yl[2 * q + 1] = (ACC)t.y;
          }
        }
#pragma unroll
        for (int e = 0; e < EPC; e += 2) {
          const ACC y0 = YLDS ? yl[e] : yv[c][e], y1 = YLDS ? yl[e + 1] : yv[c][e + 1];
          acc0 = fma_acc(a[e], y0, acc0);
          acc1 = fma_acc(a[e + 1], y1, acc1);
          if constexpr (DUAL) {
            acc2 = fma_acc(a[e], xv[c][e], acc2);
            acc3 = fma_acc(a[e + 1], xv[c][e + 1], acc3);
          }
        }
        if constexpr (KEEPCVT) __builtin_amdgcn_sched_barrier(0);
      }
      part[r] = wave_sum(acc0 + acc1);
      if constexpr (DUAL) part[R + r] = wave_sum(acc2 + acc3);
    }
    const int pb = (int)(step & 1);
    if constexpr (NW > 1) {                    // (one wave per row: the wave sum IS the row dot - no LDS trip, no barrier)
      if (lane == 0) {
#pragma unroll
        for (int r = 0; r < R * NV; ++r) red[pb][r][wave] = part[r];
      }
      __syncthreads();
    }
    ACC res[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      ACC s = (ACC)0;
      if constexpr (NW == 1) {
        s = part[r];
      } else if constexpr (KEEPCVT && NW >= 8) {      // one partial per lane + the DPP ladder: 2 VGPRs instead of 2*NW
        s = wave_sum(lane < NW ? red[pb][r][lane] : (ACC)0);
      } else {
#pragma unroll
      for (int w = 0; w < NW; ++w) s += red[pb][r][w];
      }
      const int64_t row = row_lo + first + rot(step) * group + r;
      const ACC bi = b != nullptr ? (ACC)bval[buf][r] : (ACC)0;
      if (row < row_hi) {
        s -= bi;
        rr += (double)s * (double)s;
      } else {
        s = (ACC)0;
      }
      res[r] = s;
      if constexpr (CB) {       // phase 1 of a column-blocked pass stores the residual; phase 2 (res_out == nullptr) only reads it
        if (ys.res_out != nullptr && tid == 0 && row < row_hi)
          ys.res_out[row] = (ys.res_accum ? ys.res_out[row] : 0.f) - (float)s;
      }
      if constexpr (DUAL) {
        ACC s2 = (ACC)0;
        if constexpr (NW == 1) s2 = part[R + r];
        else
#pragma unroll
        for (int w = 0; w < NW; ++w) s2 += red[pb][R + r][w];
        if (row < row_hi) { s2 -= bi; rr2 += (double)s2 * (double)s2; }
      }
    }
    if constexpr (WITH_G) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
#pragma unroll
        for (int c = 0; c < K; ++c) {
          float a[EPC];
          if constexpr (sizeof(ACC) == 8 && !KEEPCVT) {
            // fp64 form: hide the tile from value numbering, or the compiler keeps the v_cvt_f64_f32 results of the dot
            // phase alive across the barrier for reuse here (2 VGPRs per element of the tile: spills at 64 KiB rows).
            // KEEPCVT = true does keep them (bf16 storage: the second unpack + conversion is 16 of the 56 VALU
            // instructions per chunk of a pass that is VALU-bound, and the raw tile is dead once converted)
            u32x4 t = tile[buf][r][c];
            asm volatile("" : "+v"(t));
            Tr::unpack(t, a);
          } else {
            Tr::unpack(tile[buf][r][c], a);
          }
#pragma unroll
          for (int e = 0; e < EPC; ++e) gv[c][e] = fma_acc(a[e], res[r], gv[c][e]);
        }
      }
    }
  };

  // Software pipeline, written so that the steady-state loop is straight-line code: the compiler can then wait with
  // a COUNTED vmcnt (only for the tile it consumes) and leave the NBUF-1 younger tiles in flight.  Any branch around
  // an issue() makes it fall back to vmcnt(0).  Prefetches past the end re-read the last row (clamped): L2 hits.
  if (nsteps > 0) {
#pragma unroll
    for (int u = 0; u < NBUF - 1; ++u) issue(u, u);
    int64_t s = 0;
    for (; s + NBUF <= nsteps; s += NBUF) {
#pragma unroll
      for (int u = 0; u < NBUF; ++u) {
        issue((u + NBUF - 1) % NBUF, s + u + NBUF - 1);
        if constexpr (DRAIN) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifndef FOS_NO_SCHED_BARRIER
        __builtin_amdgcn_sched_barrier(0);   // keep the written order: without it the scheduler interleaves the
#endif
        consume(u, s + u);                   // steps of the unrolled group and the register tiles no longer fit
#ifndef FOS_NO_SCHED_BARRIER
        __builtin_amdgcn_sched_barrier(0);
#endif
      }
    }
    // tail: fewer than NBUF steps left, their tiles are already in flight (step s+u sits in buffer u)
#pragma unroll
    for (int u = 0; u < NBUF - 1; ++u)
      if (s + u < nsteps) consume(u, s + u);
  }

  // ---- epilogue: this workgroup's slab -------------------------------------------------------------
  ACC* slab = slabs + (int64_t)blockIdx.x * (CB && ys.slab_stride ? ys.slab_stride : (int64_t)n);
#pragma unroll
  for (int c = 0; c < K; ++c) {
    if (WITH_G && live[c]) {
      const int col = (c * THREADS + tid) * EPC;
#pragma unroll
      for (int q = 0; q < EPC / 4; ++q) {
        if constexpr (sizeof(ACC) == 4) {
          f32x4 o = {(float)gv[c][4 * q + 0], (float)gv[c][4 * q + 1], (float)gv[c][4 * q + 2], (float)gv[c][4 * q + 3]};
          *reinterpret_cast<f32x4*>(slab + col + 4 * q) = o;
        } else {
          f64x2 o0 = {(double)gv[c][4 * q + 0], (double)gv[c][4 * q + 1]}, o1 = {(double)gv[c][4 * q + 2], (double)gv[c][4 * q + 3]};
          *reinterpret_cast<f64x2*>(slab + col + 4 * q) = o0;
          *reinterpret_cast<f64x2*>(slab + col + 4 * q + 2) = o1;
        }
      }
    }
  }
  if (tid == 0) {
    rr_part[blockIdx.x] = rr;
    if constexpr (DUAL) rr2_part[blockIdx.x] = rr2;
  }
}

// ---------------------------------------------------------------------------------------------------------
// Shape-generic two-pass fallback (any n, any lda, any alignment): correctness path for ragged problems
// such as the 1000 x 5 Boston design (config 1).  Pass 1: one wave per row -> r.  Pass 2: thread per column.
// Not bandwidth-critical, so both passes accumulate in fp64 (products of two floats are exact in a double):
// the ill-conditioned config-1 data (cond(A^T A) ~ 1e9) then tracks the fp64 reference to < 1e-5.
// ---------------------------------------------------------------------------------------------------------
template <typename T> __device__ inline float elem_to_float(T v);
template <> __device__ inline float elem_to_float<float>(float v) { return v; }
template <> __device__ inline float elem_to_float<bf16_t>(bf16_t v) { return __uint_as_float((unsigned)v.bits << 16); }

template <typename T>
__global__ __launch_bounds__(256) void residual_rows_kernel(const T* __restrict__ A, int64_t lda,
                                                           const float* __restrict__ b, int64_t m, int n,
                                                           YSource ys, double* __restrict__ r_out,
                                                           double* __restrict__ rr_part) {
  if (ys.stopped != nullptr && *ys.stopped != 0) return;
  __shared__ double wsum[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const double beta = source_beta(ys);
  double rr = 0.0;
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < m; row += (int64_t)gridDim.x * 4) {
    const T* ar = A + row * lda;
    double acc = 0.0;
    for (int j = lane; j < n; j += 64) acc += (double)elem_to_float<T>(ar[j]) * source_y(ys, j, beta);   // y kept in fp64
    acc = wave_sum(acc);
    if (b != nullptr) acc -= (double)b[row];
    if (lane == 0) r_out[row] = acc;
    rr += acc * acc;
  }
  if (lane == 0) wsum[wave] = rr;
  __syncthreads();
  if (threadIdx.x == 0) rr_part[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// grid = (ceil(n/256), nchunks): block (cx, cy) sums rows of chunk cy for 256 columns -> slab[cy].
template <typename T, typename ST = float>
__global__ __launch_bounds__(256) void transpose_rows_kernel(const T* __restrict__ A, int64_t lda, int64_t m, int n,
                                                            const double* __restrict__ r, const int* stopped,
                                                            int64_t rows_per_chunk, ST* __restrict__ slabs) {
  if (stopped != nullptr && *stopped != 0) return;
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int64_t lo = (int64_t)blockIdx.y * rows_per_chunk;
  int64_t hi = lo + rows_per_chunk;
  if (hi > m) hi = m;
  if (j >= n) return;
  double acc = 0.0;
  for (int64_t row = lo; row < hi; ++row) acc += (double)elem_to_float<T>(A[row * lda + j]) * r[row];
  slabs[(int64_t)blockIdx.y * n + j] = (ST)acc;
}

}  // namespace fos
