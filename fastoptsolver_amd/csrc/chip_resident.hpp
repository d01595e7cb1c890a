// Tall-skinny problems beyond one CU's LDS, opt-in (fos_fista_run_chip): the plain FISTA / FISTA-delta / fused-ISTA loop with
// A resident in the LDS of UP TO ALL CUs and ONE grid-wide barrier per iteration (iterative_solvers.py:170-242, :289-342).
//
// The two-launch step of these shapes is bound by launch latency, not by HBM: 20000 x 5 (400 KB) takes 10.9 us per iteration,
// 100000 x 5 11.0 us (profiles/r03_midsize.txt), and the single-workgroup resident loop (resident.hpp, 2.4 us) ends at 40 KiB.
// With n <= 16 the gradient is 16 doubles: every workgroup can read ALL partial gradients behind one barrier and apply the
// identical fp64 update to its own copy of the iterate, so no second grid-wide seam is needed (the general persistent step
// needs two: fused_step.hpp).
//
//   launch   G = min(#CUs, ceil(m / 256)) workgroups of 256 threads; workgroup w copies its rows of A (zero-padded to NC
//            columns, row stride NC + 4 floats) and of b into LDS once - A is never read from HBM again
//   per iteration
//     pass    a thread walks its rows in LDS: r_i = a_i . y - b_i, g += r_i a_i, rr += r_i^2 - fp64 throughout, y = x_k +
//             beta_k (x_k - x_{k-1}) from the replicated fp64 state (every thread holds x_k, x_{k-1} in registers)
//     fold    wave sums (DPP ladder) and four waves through LDS; the workgroup's NC + 1 partials go to part[k & 1][w]
//     barrier grid-wide, two-level (fz_grid_barrier)
//     gather  wave 0 sums the G partials per column in workgroup order (fixed order: every workgroup computes the same bits)
//     update  +alpha2 y, prox, momentum for all n columns, by every thread, identically
//   exit     workgroup 0 writes x_k, x_{k-1}, the step sums of the last two iterations and ||A y - b||^2 of the last pass
// Requirements (host-checked): fp32 A, n <= 16, plain run, unsharded, rows per workgroup within the LDS budget.
#pragma once
#include "fused_step.hpp"

namespace fos {

constexpr int CR_THREADS = 256, CR_NW = CR_THREADS / 64, CR_LDS_BUDGET = 150 * 1024;

struct ChipArgs {
  const float* A; int64_t lda; const float* b; int64_t m; int n; int64_t rows_per_wg;
  double* part;            // [2][G][NC + 1]: partial gradient and ||r||^2 of workgroup w, by parity of the iteration
  double* x_cur; double* x_prev;      // n doubles each (state)
  const double* beta;      // [iters + 1]: beta[k] forms y_k from (x_k, x_{k-1}) (the momentum sequence of a plain run is data-free)
  double* stats;           // [2][4]: {sum d^2, sum gf^2, sum |x|, sum x^2} of the last ([0]) and the previous ([1]) iteration
  double* rr_out;          // ||A y - b||^2 of the last iteration's pass
  unsigned* bar;           // FZ_BAR_WORDS words, zero between launches
  int iters; int prox_kind;
  double tau, alpha1, alpha2;
  unsigned long long timeout_ticks;
  // CTRL = true (adaptive restart / step or ratio tolerances: iterative_solvers.py:209-221, :235-242): momentum and stops are
  // decided on the device every iteration - by EVERY workgroup alike, from its own copy of the iterate's step norms: no
  // exchange beyond the one barrier; `beta` is unused, the scalar state is read from and written back to `scal`
  FistaScalars* scal;
  FistaParams prm;
};

__host__ __device__ inline size_t cr_lds_bytes(int nc, int64_t rows) {
  return (size_t)rows * (nc + 4) * 4 + (size_t)((rows + 1) & ~(int64_t)1) * 4 + (size_t)CR_NW * (nc + 1) * 8 + (size_t)(nc + 1) * 8 + 16;
}
__host__ __device__ inline int64_t cr_rows_cap(int nc) { return (CR_LDS_BUDGET - 1024) / ((nc + 4) * 4 + 4); }

template <int NC, bool CTRL>
__global__ __launch_bounds__(CR_THREADS) void fista_chip_resident_kernel(ChipArgs a) {
  constexpr int S = NC + 4;                                    // row stride in floats (16-byte aligned, not a power of two)
  constexpr int NP = NC + 1;
  extern __shared__ __attribute__((aligned(16))) float cr_lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int w = blockIdx.x;
  const unsigned G = gridDim.x;
  const int n = a.n;
  const int64_t row_lo = (int64_t)w * a.rows_per_wg;
  int64_t row_hi = row_lo + a.rows_per_wg;
  if (row_hi > a.m) row_hi = a.m;
  const int rows = row_hi > row_lo ? (int)(row_hi - row_lo) : 0;
  float* a_s = cr_lds;                                                         // [rows_per_wg][S]
  float* b_s = a_s + a.rows_per_wg * S;                                        // [rows_per_wg]
  double* red = reinterpret_cast<double*>(b_s + ((a.rows_per_wg + 1) & ~(int64_t)1));   // [CR_NW][NP]
  double* gt = red + CR_NW * NP;                                               // [NP] totals of the iteration
  int* ok_lds = reinterpret_cast<int*>(gt + NP);

  // ---- A and b of this workgroup's rows: HBM -> LDS, once -------------------------------------------------------------
  for (int64_t i = tid; i < (int64_t)rows * S; i += CR_THREADS) {
    const int r = (int)(i / S), c = (int)(i % S);
    a_s[i] = c < n ? a.A[(row_lo + r) * a.lda + c] : 0.f;
  }
  for (int r = tid; r < rows; r += CR_THREADS) b_s[r] = a.b != nullptr ? a.b[row_lo + r] : 0.f;
  unsigned gen = 0;
  if (tid == 0) *ok_lds = (int)__hip_atomic_load(a.bar + (2 + FZ_NG + w % FZ_NG) * FZ_LINE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  gen = (unsigned)*ok_lds;
  __syncthreads();

  double xc[NC], xp[NC];                                       // the replicated iterate
#pragma unroll
  for (int c = 0; c < NC; ++c) { xc[c] = c < n ? a.x_cur[c] : 0.0; xp[c] = c < n ? a.x_prev[c] : 0.0; }
  double st_cur[4] = {0.0, 0.0, 0.0, 0.0}, st_prev[4] = {0.0, 0.0, 0.0, 0.0};
  // the scalar state machine of a controlled run, replicated like the iterate (fista_finalize_body, reduce_update.hpp)
  double c_t = 1.0, c_beta = 0.0, c_this = 0.0, c_prev = 0.0, c_ratio = INFINITY;
  long long c_k = 0;
  int c_restarts = 0, c_stop = STOP_NONE;
  if constexpr (CTRL) {
    c_t = a.scal->t_prev; c_beta = a.scal->beta; c_this = a.scal->this_step; c_prev = a.scal->prev_step; c_ratio = a.scal->ratio;
    c_k = a.scal->k; c_restarts = a.scal->restarts; c_stop = a.scal->stopped;
  }
  int done = 0;

  for (int it = 0; it < a.iters && c_stop == STOP_NONE; ++it) {
    const double beta = CTRL ? c_beta : a.beta[it];
    double y[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) y[c] = form_y(xc[c], xp[c], beta);
    double v[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) v[k] = 0.0;
    for (int r = tid; r < rows; r += CR_THREADS) {
      float av[NC];
#pragma unroll
      for (int q = 0; q < NC / 4; ++q) {
        const f32x4_t t = *reinterpret_cast<const f32x4_t*>(a_s + (size_t)r * S + 4 * q);
        av[4 * q] = t.x; av[4 * q + 1] = t.y; av[4 * q + 2] = t.z; av[4 * q + 3] = t.w;
      }
      double dot = 0.0;
#pragma unroll
      for (int c = 0; c < NC; ++c) dot = fma((double)av[c], y[c], dot);
      const double ri = dot - (double)b_s[r];
      v[NC] = fma(ri, ri, v[NC]);
#pragma unroll
      for (int c = 0; c < NC; ++c) v[c] = fma((double)av[c], ri, v[c]);
    }
    wave_sum_n(v);
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < NP; ++k) red[wave * NP + k] = v[k];
    }
    __syncthreads();
    double* mine = a.part + ((size_t)(it & 1) * G + w) * NP;
    if (tid < NP) mine[tid] = (red[tid] + red[NP + tid]) + (red[2 * NP + tid] + red[3 * NP + tid]);
    if (!fz_grid_barrier(a.bar, G, gen, a.timeout_ticks, ok_lds)) return;

    // ---- gather: the G partials per column, summed in workgroup order by wave 0 -------------------------------------------
    if (wave == 0) {
      const double* all = a.part + (size_t)(it & 1) * G * NP;
      double s[NP];
#pragma unroll
      for (int k = 0; k < NP; ++k) s[k] = 0.0;
      for (unsigned ww = lane; ww < G; ww += 64) {
#pragma unroll
        for (int k = 0; k < NP; ++k) s[k] += all[(size_t)ww * NP + k];
      }
      wave_sum_n(s);
      if (lane == 0) {
#pragma unroll
        for (int k = 0; k < NP; ++k) gt[k] = s[k];
      }
    }
    __syncthreads();
    // ---- update: every thread, identically (iterative_solvers.py:200-221) ----------------------------------------------------
#pragma unroll
    for (int k = 0; k < 4; ++k) { st_prev[k] = st_cur[k]; st_cur[k] = 0.0; }
    const double thr = a.tau * a.alpha1, shrink = 1.0 / (1.0 + a.tau * a.alpha2);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      if (c < n) {
        double gf = gt[c];
        if (a.prox_kind == PROX_L1 && a.alpha2 > 0.0) gf += a.alpha2 * y[c];
        const double vv = y[c] - a.tau * gf;
        double xn = a.alpha1 > 0.0 ? soft_threshold(vv, thr) : vv;
        if (a.prox_kind == PROX_ENET) xn *= shrink;
        const double d = xn - xc[c];
        st_cur[0] += d * d; st_cur[1] += gf * gf; st_cur[2] += fabs(xn); st_cur[3] += xn * xn;
        xp[c] = xc[c];
        xc[c] = xn;
      }
    }
    if constexpr (CTRL) {                                        // identical inputs in every thread of every workgroup
      const double step = sqrt(st_cur[0]);
      const double prev = c_this;
      const double ratio = prev > 0.0 ? step / prev : INFINITY;
      if (a.prm.mode == MODE_FISTA) {
        if (a.prm.adaptive_restart && ratio > a.prm.restart_threshold) {
          c_t = 1.0; c_beta = 0.0; c_restarts += 1;
        } else {
          const double t_new = 0.5 * (1.0 + sqrt(1.0 + 4.0 * c_t * c_t));
          c_beta = (c_t - 1.0) / t_new;
          c_t = t_new;
        }
      } else if (a.prm.mode == MODE_DELTA) {
        const double kk = (double)(c_k + 1);
        c_beta = kk / (kk + 1.0 + a.prm.delta);
      } else {
        c_beta = 0.0;
      }
      c_prev = prev; c_this = step; c_ratio = ratio; c_k += 1;
      if (a.prm.tol_step > 0.0 && step < a.prm.tol_step) c_stop = STOP_STEP;
      if (c_stop == STOP_NONE && a.prm.tol_ratio > 0.0 && ratio < a.prm.tol_ratio) c_stop = STOP_RATIO;
    }
    done += 1;
    __syncthreads();                                             // gt is rewritten by the next iteration's gather
  }
  if (w == 0 && tid == 0 && done > 0) {
    if constexpr (CTRL) {
      FistaScalars* sc = a.scal;
      sc->t_prev = c_t; sc->beta = c_beta; sc->this_step = c_this; sc->prev_step = c_prev; sc->ratio = c_ratio;
      sc->rr = gt[NC]; sc->gnorm2 = st_cur[1]; sc->xnorm1 = st_cur[2]; sc->xnorm2 = st_cur[3];
      sc->k = c_k; sc->restarts = c_restarts; sc->stopped = c_stop;
    }
#pragma unroll
    for (int c = 0; c < NC; ++c)
      if (c < n) { a.x_cur[c] = xc[c]; a.x_prev[c] = xp[c]; }
#pragma unroll
    for (int k = 0; k < 4; ++k) { a.stats[k] = st_cur[k]; a.stats[4 + k] = st_prev[k]; }
    *a.rr_out = gt[NC];
  }
}

}  // namespace fos
