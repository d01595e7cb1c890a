// L-BFGS device pieces: two-loop recursion in ONE launch + the small vector kernels of the line search.
//
// The reference has no L-BFGS arithmetic of its own: lbfgs.py:64 calls scipy.optimize.fmin_l_bfgs_b.
// Spec restated in SURVEY.md 8(c) / oracle/fos_oracle.py::two_loop_direction:
//   newest -> oldest:  rho_i = 1/(y_i.s_i), a_i = rho_i s_i.q, q -= a_i y_i
//   q *= (s_last.y_last)/(y_last.y_last)
//   oldest -> newest:  b = rho_i y_i.q, q += s_i (a_i - b);   d = -q
//
// n-vectors are tiny next to A (32-64 KiB): the kernel is launch-latency bound, so everything runs in one
// 1024-thread workgroup with q kept in global/L2, coalesced float4 traffic and wave-level (shuffle)
// reductions accumulated in fp64.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gemv_pair.hpp"

namespace fos {

constexpr int LB_THREADS = 1024;
constexpr int LB_MAXHIST = 64;

// Workgroup-wide sum of NV doubles, result in every thread.  Two barriers, no serial stage and almost no LDS traffic:
// lane l of every wave reads the partial of wave l & 15 and a 16-lane DPP butterfly (quad_perm, quad_perm,
// row_half_mirror, row_mirror) leaves the total in every lane - the same pairing in every lane, so all threads hold
// the same bits (deterministic).  (Every thread adding the 16 partials itself was 48 LDS reads per thread and step:
// 1.3 us of LDS pipe per step of the two-loop chain, most of its time.)
template <int NV>
__device__ inline void block_sum_bcast(double (&v)[NV], double (*lds)[LB_THREADS / 64]) {
  static_assert(LB_THREADS / 64 == 16, "one 16-lane DPP row per set of wave partials");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  wave_sum_n(v);                         // DPP path: the NV sums advance together, ~6 short steps instead of 12*NV
  __syncthreads();                       // previous readers of lds are done
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) lds[i][wave] = v[i];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = lds[i][lane & 15];
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] += dpp_fetch<0xB1, 0xF>(v[i]);      // quad_perm [1,0,3,2]
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] += dpp_fetch<0x4E, 0xF>(v[i]);      // quad_perm [2,3,0,1]
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] += dpp_fetch<0x141, 0xF>(v[i]);     // row_half_mirror
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] += dpp_fetch<0x140, 0xF>(v[i]);     // row_mirror
}

// Four consecutive elements of a float / double vector (16-byte aligned for float, 32-byte for double).
__device__ inline void load4(const float* p, double (&o)[4]) {
  const f32x4 t = *reinterpret_cast<const f32x4*>(p);
  o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w;
}
__device__ inline void load4(const double* p, double (&o)[4]) {
  const f64x2 a = *reinterpret_cast<const f64x2*>(p), b = *reinterpret_cast<const f64x2*>(p + 2);
  o[0] = a.x; o[1] = a.y; o[2] = b.x; o[3] = b.y;
}
__device__ inline void store4(float* p, const double (&v)[4]) {
  *reinterpret_cast<f32x4*>(p) = f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
__device__ inline void store4(double* p, const double (&v)[4]) {
  *reinterpret_cast<f64x2*>(p) = f64x2{v[0], v[1]};
  *reinterpret_cast<f64x2*>(p + 2) = f64x2{v[2], v[3]};
}

// Four consecutive elements in their storage type (the staged history vectors stay in VT until they are used).
__device__ inline void loadv(const float* p, float (&o)[4]) {
  const f32x4 t = *reinterpret_cast<const f32x4*>(p);
  o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w;
}
__device__ inline void loadv(const double* p, double (&o)[4]) {
  const f64x2 a = *reinterpret_cast<const f64x2*>(p), b = *reinterpret_cast<const f64x2*>(p + 2);
  o[0] = a.x; o[1] = a.y; o[2] = b.x; o[3] = b.y;
}

// VT: element type of g, S, Y and d.  float: the storage type of the round-1 driver (q is rounded to fp32 after every
// update, like a float32 NumPy vector would be).  double: SciPy's own precision (lbfgs.py:64) - q, the history and d are
// fp64 end to end; this is what LBFGSSolver.fit uses.
// NQ = 4-element chunks of q per thread kept in REGISTERS (n <= NQ * 4096); NQ = 0: q lives in global memory
// (any n; each thread only ever touches its own elements, so no cross-thread hazard on q).
// The 2*hist steps are a dependent chain (dot -> workgroup sum -> axpy), so the register form hides the memory latency
// of the history behind it: the pair of step k+1 is requested before the reduction of step k and lands while the
// workgroup sums (one load latency in total instead of one per step).
template <typename VT, int NQ>
__global__ __launch_bounds__(LB_THREADS) void lbfgs_two_loop_kernel(const VT* __restrict__ g,
                                                                    const VT* __restrict__ S,
                                                                    const VT* __restrict__ Y, int hist, int head,
                                                                    int cap, int64_t n, VT* __restrict__ qout,
                                                                    double* __restrict__ gd_out = nullptr) {
  __shared__ double lds[3][LB_THREADS / 64];
  __shared__ double coef[LB_MAXHIST];
  __shared__ double rho[LB_MAXHIST];
  const int tid = threadIdx.x;
  constexpr int NR = NQ > 0 ? NQ : 1;
  constexpr bool PF = NQ > 0 && NQ <= 2;      // prefetch the next pair (n <= 8192); beyond that the registers are q's
  double q[NR][4];                            // held in fp64; rounded to VT after every update (no-op for double)
  auto col_of = [&](int c) { return (int64_t)(c * LB_THREADS + tid) * 4; };
  auto rnd = [](double v) { return (double)(VT)v; };
  VT cs[NR][4], cy[NR][4];                    // the pair of the current step (register form only)
  auto fetch = [&](int h, auto& s_, auto& y_) {
    const int slot = (head + h) % cap;
    const VT* s = S + (int64_t)slot * n;
    const VT* y = Y + (int64_t)slot * n;
#pragma unroll
    for (int c = 0; c < NR; ++c) {
      const int64_t col = col_of(c);
      const bool in = col < n;                // lanes past the end read element 0 and keep zeros (no divergent loads)
      VT ts[4], ty[4];
      loadv(s + (in ? col : 0), ts);
      loadv(y + (in ? col : 0), ty);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s_[c][e] = in ? ts[e] : (VT)0;
        y_[c][e] = in ? ty[e] : (VT)0;
      }
    }
  };
  auto rotate = [&](const VT (&s_)[PF ? NR : 1][4], const VT (&y_)[PF ? NR : 1][4]) {
#pragma unroll
    for (int c = 0; c < NR; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e) { cs[c][e] = s_[c][e]; cy[c][e] = y_[c][e]; }
  };
  if constexpr (NQ > 0) {
    if (PF && hist > 0) fetch(hist - 1, cs, cy);
#pragma unroll
    for (int c = 0; c < NQ; ++c) {
      const int64_t col = col_of(c);
      if (col < n) load4(g + col, q[c]);
      else q[c][0] = q[c][1] = q[c][2] = q[c][3] = 0.0;
    }
  } else {
    for (int64_t i = tid; i < n; i += LB_THREADS) qout[i] = g[i];
  }
  double sy_last = 1.0, yy_last = 1.0;
  for (int h = hist - 1; h >= 0; --h) {
    double acc[3] = {0.0, 0.0, 0.0};          // s.q, y.s, y.y
    VT ns[PF ? NR : 1][4], ny[PF ? NR : 1][4];
    if constexpr (NQ > 0) {
      if constexpr (PF) fetch(h > 0 ? h - 1 : 0, ns, ny);       // next step's pair (h = 0: the second sweep starts with pair 0)
      const VT* s = S + (int64_t)((head + h) % cap) * n;
      const VT* y = Y + (int64_t)((head + h) % cap) * n;
#pragma unroll
      for (int c = 0; c < NQ; ++c) {
        VT st[4];
        if constexpr (!PF) {                  // s is only needed for the dots: one chunk at a time
          const int64_t col = col_of(c);
          st[0] = st[1] = st[2] = st[3] = (VT)0;
          cy[c][0] = cy[c][1] = cy[c][2] = cy[c][3] = (VT)0;
          if (col < n) { loadv(s + col, st); loadv(y + col, cy[c]); }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const double sv = (double)(PF ? cs[c][e] : st[e]), yv = (double)cy[c][e];
          acc[0] += sv * q[c][e];
          acc[1] += yv * sv;
          acc[2] += yv * yv;
        }
      }
    } else {
      const int slot = (head + h) % cap;
      const VT* s = S + (int64_t)slot * n;
      const VT* y = Y + (int64_t)slot * n;
      for (int64_t i = tid; i < n; i += LB_THREADS) {
        const double sv = s[i], yv = y[i];
        acc[0] += sv * (double)qout[i];
        acc[1] += yv * sv;
        acc[2] += yv * yv;
      }
    }
    block_sum_bcast<3>(acc, lds);
    if (h == hist - 1) { sy_last = acc[1]; yy_last = acc[2]; }
    const double r = 1.0 / acc[1];
    const double a = r * acc[0];
    if (tid == 0) { coef[h] = a; rho[h] = r; }
    if constexpr (NQ > 0) {
#pragma unroll
      for (int c = 0; c < NQ; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) q[c][e] = rnd(q[c][e] - a * (double)cy[c][e]);
      if constexpr (PF) rotate(ns, ny);
    } else {
      const VT* y = Y + (int64_t)((head + h) % cap) * n;
      for (int64_t i = tid; i < n; i += LB_THREADS) qout[i] = (VT)((double)qout[i] - a * (double)y[i]);
    }
  }
  __syncthreads();                            // coef / rho visible to everyone
  if (hist > 0) {
    const double gam = sy_last / yy_last;
    if constexpr (NQ > 0) {
#pragma unroll
      for (int c = 0; c < NQ; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) q[c][e] = rnd(q[c][e] * gam);
    } else {
      for (int64_t i = tid; i < n; i += LB_THREADS) qout[i] = (VT)((double)qout[i] * gam);
    }
  }
  for (int h = 0; h < hist; ++h) {
    double acc[1] = {0.0};
    VT ns[PF ? NR : 1][4], ny[PF ? NR : 1][4];
    if constexpr (NQ > 0) {
      if constexpr (PF) fetch(h + 1 < hist ? h + 1 : h, ns, ny);
      const VT* s = S + (int64_t)((head + h) % cap) * n;
      const VT* y = Y + (int64_t)((head + h) % cap) * n;
#pragma unroll
      for (int c = 0; c < NQ; ++c) {
        VT yt[4];
        if constexpr (!PF) {
          const int64_t col = col_of(c);
          yt[0] = yt[1] = yt[2] = yt[3] = (VT)0;
          cs[c][0] = cs[c][1] = cs[c][2] = cs[c][3] = (VT)0;
          if (col < n) { loadv(y + col, yt); loadv(s + col, cs[c]); }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[0] += (double)(PF ? cy[c][e] : yt[e]) * q[c][e];
      }
    } else {
      const VT* y = Y + (int64_t)((head + h) % cap) * n;
      for (int64_t i = tid; i < n; i += LB_THREADS) acc[0] += (double)y[i] * (double)qout[i];
    }
    block_sum_bcast<1>(acc, lds);
    const double w = coef[h] - rho[h] * acc[0];
    if constexpr (NQ > 0) {
#pragma unroll
      for (int c = 0; c < NQ; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) q[c][e] = rnd(q[c][e] + w * (double)cs[c][e]);
      if constexpr (PF) rotate(ns, ny);
    } else {
      const VT* s = S + (int64_t)((head + h) % cap) * n;
      for (int64_t i = tid; i < n; i += LB_THREADS) qout[i] = (VT)((double)qout[i] + w * (double)s[i]);
    }
  }
  double fin[2] = {0.0, 0.0};                 // g.d and d.d of the direction d = -q (the line search starts from them)
  if constexpr (NQ > 0) {
#pragma unroll
    for (int c = 0; c < NQ; ++c) {
      const int64_t col = col_of(c);
      const double neg[4] = {-q[c][0], -q[c][1], -q[c][2], -q[c][3]};
      if (col < n) {
        store4(qout + col, neg);
        if (gd_out != nullptr) {
          double gv[4];
          load4(g + col, gv);
#pragma unroll
          for (int e = 0; e < 4; ++e) { fin[0] += gv[e] * (double)(VT)neg[e]; fin[1] += (double)(VT)neg[e] * (double)(VT)neg[e]; }
        }
      }
    }
  } else {
    for (int64_t i = tid; i < n; i += LB_THREADS) {
      const VT dv = -qout[i];
      qout[i] = dv;
      fin[0] += (double)g[i] * (double)dv;
      fin[1] += (double)dv * (double)dv;
    }
  }
  if (gd_out != nullptr) {                    // uniform branch
    block_sum_bcast<2>(fin, lds);
    if (tid == 0) { gd_out[0] = fin[0]; gd_out[1] = fin[1]; }
  }
}

// ---------------------------------------------------------------------------------------------------------
// The same direction on MANY compute units (fos_lbfgs_direction_dd).  The two-loop chain above is 2*hist dependent
// dot -> axpy steps, which only one workgroup can run: its time is (4*hist + 2) n-vectors through ONE CU's L2 port
// (~90 GB/s: 35 us at n = 8192, 0.9 ms at n = 65536).  Every dot product of the chain is a bilinear form of the basis
// {s_0..s_{h-1}, y_0..y_{h-1}, g}: with q = sum_j delta_j b_j, s_i.q = sum_j delta_j (b_i.b_j).  So
//   1. lbfgs_gram_kernel     every workgroup takes 128 columns of the 2h+1 basis vectors and forms its share of the
//                            Gram matrix G = B B^T ((2h+1)(h+1) <= 231 dot products), one partial per workgroup;
//   2. lbfgs_combine_kernel  every workgroup sums the partials (fixed order), ONE wave runs the two-loop recursion on
//                            the coefficients delta (lane j holds delta_j; the chain is now 2h wave reductions over
//                            <= 21 numbers, ~1 us), and the workgroup writes d = -sum_j delta_j b_j for its columns.
// Two launches that each read the history once with the whole chip, and g.d = -sum_j delta_j G[g][j],
// d.d = delta^T G delta come out of the same coefficients.  Identical to the two-loop recursion in exact arithmetic
// (it is how L-BFGS-B itself represents the matrix); in fp64 the two agree to ~1e-13 relative (tests).
// ---------------------------------------------------------------------------------------------------------
constexpr int VL_MAXH = 10;                       // SciPy's m (lbfgs.py:64 leaves the default)
constexpr int VL_NB = 2 * VL_MAXH + 1;            // basis vectors
constexpr int VL_PAIRS = VL_NB * (VL_NB + 1) / 2; // 231
constexpr int VL_COLS = 128, VL_THREADS = 256, VL_PSTRIDE = 256, VL_MAXPARTS = 64;

__device__ inline const double* vl_row(const double* g, const double* S, const double* Y, int r, int hist, int head, int cap,
                                       int64_t n) {
  if (r < hist) return S + (int64_t)((head + r) % cap) * n;
  if (r < 2 * hist) return Y + (int64_t)((head + r - hist) % cap) * n;
  return g;
}
// pair index p of (i <= j) in row-major upper-triangular order over nb rows
__device__ inline void vl_pair(int p, int nb, int& i, int& j) {
  int row = 0, left = p;
  while (left >= nb - row) { left -= nb - row; ++row; }
  i = row; j = row + left;
}
__device__ inline int vl_index(int i, int j, int nb) {          // i <= j
  return i * nb - i * (i - 1) / 2 + (j - i);
}

static __global__ __launch_bounds__(VL_THREADS) void lbfgs_gram_kernel(const double* __restrict__ g, const double* __restrict__ S,
                                                                const double* __restrict__ Y, int hist, int head, int cap,
                                                                int64_t n, double* __restrict__ partial) {
  __shared__ double L[VL_NB][VL_COLS + 1];
  const int nb = 2 * hist + 1, tid = threadIdx.x;
  const int npairs = nb * (nb + 1) / 2;
  int i = 0, j = 0;
  if (tid < npairs) vl_pair(tid, nb, i, j);
  double acc = 0.0;
  const int64_t nchunks = (n + VL_COLS - 1) / VL_COLS;
  constexpr int RPT = (VL_NB + 1) / 2;               // rows per thread: 128 columns x 2 row phases = 256 threads
  const int c = tid % VL_COLS, r0 = tid / VL_COLS;
  for (int64_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {     // at most VL_MAXPARTS workgroups
    const int64_t col = chunk * VL_COLS + c;
    const bool in = col < n;
    double v[RPT];
#pragma unroll
    for (int k = 0; k < RPT; ++k) {                  // all loads of the tile in flight together (branch-free: clamped row)
      const int r = r0 + 2 * k;
      v[k] = vl_row(g, S, Y, r < nb ? r : nb - 1, hist, head, cap, n)[in ? col : 0];
    }
    __syncthreads();                                 // the previous chunk has been consumed
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      const int r = r0 + 2 * k;
      if (r < nb) L[r][c] = in ? v[k] : 0.0;
    }
    __syncthreads();
    if (tid < npairs) {
#pragma unroll 8
      for (int cc = 0; cc < VL_COLS; ++cc) acc += L[i][cc] * L[j][cc];
    }
  }
  if (tid < npairs) partial[(int64_t)blockIdx.x * VL_PSTRIDE + tid] = acc;
}

static __global__ __launch_bounds__(VL_THREADS) void lbfgs_combine_kernel(const double* __restrict__ g, const double* __restrict__ S,
                                                                   const double* __restrict__ Y, int hist, int head, int cap,
                                                                   int64_t n, const double* __restrict__ partial, int nparts,
                                                                   double* __restrict__ d_out, double* __restrict__ gd_out,
                                                                   double* __restrict__ x_step = nullptr,
                                                                   double* __restrict__ x_old = nullptr) {
  // x_step (fos_lbfgs_minimize, every iteration but the first): the first trial point of the line search rides along -
  // x_old = x, x = 1.0*d + x (L-BFGS-B's unit first step, products and sum rounded as lbfgs_first_trial_kernel does)
  __shared__ double G[VL_NB][VL_NB + 1];
  __shared__ double delta_s[VL_NB];
  const int nb = 2 * hist + 1, tid = threadIdx.x;
  const int npairs = nb * (nb + 1) / 2;
  // this thread's column: the 2h+1 basis values are requested first and land while the coefficients are worked out
  const int64_t col = (int64_t)blockIdx.x * VL_THREADS + tid;
  double bv[VL_NB];
#pragma unroll
  for (int r = 0; r < VL_NB; ++r)
    bv[r] = vl_row(g, S, Y, r < nb ? r : nb - 1, hist, head, cap, n)[col < n ? col : 0];
  if (tid < npairs) {
    double acc = 0.0;
#pragma unroll
    for (int w = 0; w < VL_MAXPARTS; ++w) {          // fixed order, all loads in flight (clamped, weight 0 past the end)
      const double v = partial[(int64_t)(w < nparts ? w : nparts - 1) * VL_PSTRIDE + tid];
      acc += w < nparts ? v : 0.0;
    }
    int i, j;
    vl_pair(tid, nb, i, j);
    G[i][j] = acc;
    G[j][i] = acc;
  }
  __syncthreads();
  if (tid < 64) {                                  // one wave: lane j <-> basis vector j
    const int j = tid;
    const bool on = j < nb;
    double delta = (j == 2 * hist) ? 1.0 : 0.0;   // q = g
    const double rho = j < hist ? 1.0 / G[j][hist + j] : 0.0;               // lane i: 1 / (s_i . y_i)
    auto lane_of = [](double v, int lane) {
      const long long bits = __double_as_longlong(v);
      const int lo = __builtin_amdgcn_readlane((int)bits, lane), hi = __builtin_amdgcn_readlane((int)(bits >> 32), lane);
      return __longlong_as_double(((long long)hi << 32) | (long long)(unsigned)lo);
    };
    double a_mine = 0.0;                           // lane i < hist keeps a_i for the second sweep
    for (int i = hist - 1; i >= 0; --i) {
      const double dot = wave_sum_dpp(on ? delta * G[i][j] : 0.0);          // s_i . q
      const double a = lane_of(rho, i) * dot;
      if (j == i) a_mine = a;
      if (j == hist + i) delta -= a;                                         // q -= a_i y_i
    }
    if (hist > 0) delta *= G[hist - 1][2 * hist - 1] / G[2 * hist - 1][2 * hist - 1];   // (s.y)/(y.y) of the newest pair
    for (int i = 0; i < hist; ++i) {
      const double dot = wave_sum_dpp(on ? delta * G[hist + i][j] : 0.0);   // y_i . q
      const double beta = lane_of(rho, i) * dot;
      if (j == i) delta += a_mine - beta;                                    // q += s_i (a_i - beta)
    }
    if (on) delta_s[j] = delta;
    if (blockIdx.x == 0 && gd_out != nullptr) {
      // d = -q:  g.d = -sum_j delta_j G[g][j],  d.d = sum_ij delta_i delta_j G[i][j]
      const double gq = wave_sum_dpp(on ? delta * G[2 * hist][j] : 0.0);
      double row = 0.0;
      for (int i = 0; i < nb; ++i) row += lane_of(delta, i) * G[i][on ? j : 0];
      const double qq = wave_sum_dpp(on ? row * delta : 0.0);
      if (j == 0) { gd_out[0] = -gq; gd_out[1] = qq; }
    }
  }
  __syncthreads();
  // this workgroup's columns of d = -sum_r delta_r b_r (fixed order r = 0 .. 2h)
  if (col < n) {
    double acc = 0.0;
#pragma unroll
    for (int r = 0; r < VL_NB; ++r) acc += (r < nb ? delta_s[r] : 0.0) * bv[r];
    d_out[col] = -acc;
    if (x_step != nullptr) {
      const double xo = x_step[col];
      x_old[col] = xo;
      x_step[col] = __dadd_rn(__dmul_rn(1.0, -acc), xo);
    }
  }
}

// out5 = { x.x, g.d, d.d, max|g|, ||x||_1 }; any pointer may be NULL (its entries are then 0).  XT: float or double
// iterate.  With `extra` the scalar *extra rides along as out5[5] (the ||r||^2 of the evaluation: one host read for all).
template <typename XT, typename GT = float>
__global__ __launch_bounds__(LB_THREADS) void vec_stats_kernel(const XT* __restrict__ x, const GT* __restrict__ g,
                                                               const GT* __restrict__ d, int64_t n,
                                                               double* __restrict__ out5,
                                                               const double* __restrict__ extra = nullptr,
                                                               unsigned long long* flag = nullptr,
                                                               unsigned long long seq = 0,
                                                               const unsigned long long* t_start = nullptr) {
  __shared__ double lds[5][16];
  // t_start (fos_lbfgs_minimize): the constant-rate wall clock read by stamp_kernel in front of the evaluation; this
  // kernel runs right behind it, so (now - *t_start) is the evaluation's device time - out5[9], no hipEvent on the stream
  const unsigned long long t_now = t_start != nullptr ? wall_clock64() : 0ull;
  double xx = 0.0, gd = 0.0, dd = 0.0, gm = 0.0, x1 = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += LB_THREADS) {
    const double xv = x ? (double)x[i] : 0.0, gv = g ? (double)g[i] : 0.0, dv = d ? (double)d[i] : 0.0;
    xx += xv * xv;
    x1 += fabs(xv);
    gd += gv * dv;
    dd += dv * dv;
    gm = fmax(gm, fabs(gv));
  }
  xx = wave_sum(xx); gd = wave_sum(gd); dd = wave_sum(dd); x1 = wave_sum(x1);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) gm = fmax(gm, __shfl_xor(gm, off, 64));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { lds[0][wave] = xx; lds[1][wave] = gd; lds[2][wave] = dd; lds[3][wave] = gm; lds[4][wave] = x1; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0.0, b = 0.0, c = 0.0, e = 0.0, f = 0.0;
    for (int i = 0; i < 16; ++i) { a += lds[0][i]; b += lds[1][i]; c += lds[2][i]; e = fmax(e, lds[3][i]); f += lds[4][i]; }
    out5[0] = a; out5[1] = b; out5[2] = c; out5[3] = e; out5[4] = f;
    if (extra != nullptr) out5[5] = *extra;
    if (t_start != nullptr) out5[9] = (double)(t_now - *t_start);
    if (flag != nullptr) {                  // out5 in pinned host memory: the host polls `flag` instead of draining the stream
      __threadfence_system();
      __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

static __global__ void stamp_kernel(unsigned long long* out) { *out = wall_clock64(); }

// out3 = { ||x||^2, ||x||_1, 0 }
static __global__ __launch_bounds__(LB_THREADS) void vec_norms_kernel(const float* __restrict__ x, int64_t n,
                                                               double* __restrict__ out2) {
  __shared__ double lds[2][16];
  double a = 0.0, b = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += LB_THREADS) { const double v = x[i]; a += v * v; b += fabs(v); }
  a = wave_sum(a); b = wave_sum(b);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { lds[0][wave] = a; lds[1][wave] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double sa = 0.0, sb = 0.0;
    for (int i = 0; i < 16; ++i) { sa += lds[0][i]; sb += lds[1][i]; }
    out2[0] = sa; out2[1] = sb;
  }
}

static __global__ __launch_bounds__(256) void vec_axpby_kernel(float a, const float* __restrict__ x, float b,
                                                        const float* __restrict__ y, float* __restrict__ out,
                                                        int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float v = a * x[i];
    if (y != nullptr) v = fmaf(b, y[i], v);
    out[i] = v;
  }
}

static __global__ __launch_bounds__(256) void cast_f64_f32_kernel(const double* __restrict__ in, float* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = (float)in[i];
}

// grad_out = gbuf + alpha2 * y   (L-BFGS fg: lbfgs.py:50-51).  YT: float or double iterate.
template <typename YT>
__global__ __launch_bounds__(256) void add_l2_kernel(const float* __restrict__ gbuf, double alpha2,
                                                     const YT* __restrict__ y, float* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    out[i] = alpha2 != 0.0 ? (float)((double)gbuf[i] + alpha2 * (double)y[i]) : gbuf[i];
}

// out (fp64) = a*x (fp64) + b*y (YT = fp32 or fp64): the L-BFGS trial point x_old + stp*d with the iterate kept in fp64.
// Each product and the sum round separately (no fma contraction): bit for bit what NumPy's `stp * d + x_old` gives.
template <typename YT>
__global__ __launch_bounds__(256) void vec_axpby_f64_kernel(double a, const double* __restrict__ x, double b,
                                                            const YT* __restrict__ y, double* __restrict__ out,
                                                            int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    double v = __dmul_rn(a, x[i]);
    if (y != nullptr) v = __dadd_rn(__dmul_rn(b, (double)y[i]), v);
    out[i] = v;
  }
}

// First trial point of an L-BFGS iteration: x_old = x, x = stp*d + x_old (rounded like vec_axpby_f64_kernel) in one launch.
static __global__ __launch_bounds__(256) void lbfgs_first_trial_kernel(double* __restrict__ x, const double* __restrict__ d,
                                                                double stp, double* __restrict__ x_old, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const double xo = x[i];
    x_old[i] = xo;
    x[i] = __dadd_rn(__dmul_rn(stp, d[i]), xo);
  }
}

// End of an L-BFGS iteration in one launch: the accepted iterate goes to the record (iter_out, may be NULL) and, when the
// curvature test kept it, the new correction pair is stored: s = stp * d, y = g - g_old.
static __global__ __launch_bounds__(256) void lbfgs_store_pair_kernel(double stp, const double* __restrict__ d,
                                                               const double* __restrict__ g,
                                                               const double* __restrict__ g_old, double* __restrict__ s_out,
                                                               double* __restrict__ y_out, const double* __restrict__ x,
                                                               double* __restrict__ iter_out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    if (iter_out != nullptr) iter_out[i] = x[i];
    if (s_out != nullptr) {
      s_out[i] = __dmul_rn(stp, d[i]);
      y_out[i] = __dadd_rn(g[i], -g_old[i]);
    }
  }
}

}  // namespace fos
