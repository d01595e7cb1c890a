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

// Workgroup-wide sum of NV doubles, result broadcast to every thread.  Two barriers, no serial stage: every
// thread adds the 16 wave partials itself (LDS broadcast reads), in the same fixed order -> deterministic.
template <int NV>
__device__ inline void block_sum_bcast(double (&v)[NV], double (*lds)[LB_THREADS / 64]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  wave_sum_n(v);                         // DPP path: the NV sums advance together, ~6 short steps instead of 12*NV
  __syncthreads();                       // previous readers of lds are done
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) lds[i][wave] = v[i];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < LB_THREADS / 64; ++w) s += lds[i][w];
    v[i] = s;
  }
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

// VT: element type of g, S, Y and d.  float: the storage type of the round-1 driver (q is rounded to fp32 after every
// update, like a float32 NumPy vector would be).  double: SciPy's own precision (lbfgs.py:64) - q, the history and d are
// fp64 end to end; this is what LBFGSSolver.fit uses.
// NQ = 4-element chunks of q per thread kept in REGISTERS (n <= NQ * 4096); NQ = 0: q lives in global memory
// (any n; each thread only ever touches its own elements, so no cross-thread hazard on q).
template <typename VT, int NQ>
__global__ __launch_bounds__(LB_THREADS) void lbfgs_two_loop_kernel(const VT* __restrict__ g,
                                                                    const VT* __restrict__ S,
                                                                    const VT* __restrict__ Y, int hist, int head,
                                                                    int cap, int64_t n, VT* __restrict__ qout) {
  __shared__ double lds[3][LB_THREADS / 64];
  __shared__ double coef[LB_MAXHIST];
  __shared__ double rho[LB_MAXHIST];
  const int tid = threadIdx.x;
  constexpr int NR = NQ > 0 ? NQ : 1;
  double q[NR][4];                            // held in fp64; rounded to VT after every update (no-op for double)
  auto col_of = [&](int c) { return (int64_t)(c * LB_THREADS + tid) * 4; };
  auto rnd = [](double v) { return (double)(VT)v; };
  if constexpr (NQ > 0) {
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
    const int slot = (head + h) % cap;
    const VT* s = S + (int64_t)slot * n;
    const VT* y = Y + (int64_t)slot * n;
    double acc[3] = {0.0, 0.0, 0.0};          // s.q, y.s, y.y
    double yk[NR][4];
    if constexpr (NQ > 0) {
#pragma unroll
      for (int c = 0; c < NQ; ++c) {
        const int64_t col = col_of(c);
        double sv[4] = {0.0, 0.0, 0.0, 0.0};
        yk[c][0] = yk[c][1] = yk[c][2] = yk[c][3] = 0.0;
        if (col < n) { load4(s + col, sv); load4(y + col, yk[c]); }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc[0] += sv[e] * q[c][e];
          acc[1] += yk[c][e] * sv[e];
          acc[2] += yk[c][e] * yk[c][e];
        }
      }
    } else {
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
        for (int e = 0; e < 4; ++e) q[c][e] = rnd(q[c][e] - a * yk[c][e]);
    } else {
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
    const int slot = (head + h) % cap;
    const VT* s = S + (int64_t)slot * n;
    const VT* y = Y + (int64_t)slot * n;
    double acc[1] = {0.0};
    double sk[NR][4];
    if constexpr (NQ > 0) {
#pragma unroll
      for (int c = 0; c < NQ; ++c) {
        const int64_t col = col_of(c);
        double yv[4] = {0.0, 0.0, 0.0, 0.0};
        sk[c][0] = sk[c][1] = sk[c][2] = sk[c][3] = 0.0;
        if (col < n) { load4(y + col, yv); load4(s + col, sk[c]); }
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[0] += yv[e] * q[c][e];
      }
    } else {
      for (int64_t i = tid; i < n; i += LB_THREADS) acc[0] += (double)y[i] * (double)qout[i];
    }
    block_sum_bcast<1>(acc, lds);
    const double w = coef[h] - rho[h] * acc[0];
    if constexpr (NQ > 0) {
#pragma unroll
      for (int c = 0; c < NQ; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) q[c][e] = rnd(q[c][e] + w * sk[c][e]);
    } else {
      for (int64_t i = tid; i < n; i += LB_THREADS) qout[i] = (VT)((double)qout[i] + w * (double)s[i]);
    }
  }
  if constexpr (NQ > 0) {
#pragma unroll
    for (int c = 0; c < NQ; ++c) {
      const int64_t col = col_of(c);
      const double neg[4] = {-q[c][0], -q[c][1], -q[c][2], -q[c][3]};
      if (col < n) store4(qout + col, neg);
    }
  } else {
    for (int64_t i = tid; i < n; i += LB_THREADS) qout[i] = -qout[i];
  }
}

// out5 = { x.x, g.d, d.d, max|g|, ||x||_1 }; any pointer may be NULL (its entries are then 0).  XT: float or double
// iterate.
template <typename XT, typename GT = float>
__global__ __launch_bounds__(LB_THREADS) void vec_stats_kernel(const XT* __restrict__ x, const GT* __restrict__ g,
                                                               const GT* __restrict__ d, int64_t n,
                                                               double* __restrict__ out5) {
  __shared__ double lds[5][16];
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
  }
}

// out3 = { ||x||^2, ||x||_1, 0 }
__global__ __launch_bounds__(LB_THREADS) void vec_norms_kernel(const float* __restrict__ x, int64_t n,
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

__global__ __launch_bounds__(256) void vec_axpby_kernel(float a, const float* __restrict__ x, float b,
                                                        const float* __restrict__ y, float* __restrict__ out,
                                                        int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float v = a * x[i];
    if (y != nullptr) v = fmaf(b, y[i], v);
    out[i] = v;
  }
}

__global__ __launch_bounds__(256) void cast_f64_f32_kernel(const double* __restrict__ in, float* __restrict__ out, int64_t n) {
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

}  // namespace fos
