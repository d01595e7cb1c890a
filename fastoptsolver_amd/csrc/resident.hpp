// LDS-resident solver loop for small problems (BASELINE config 1: the reference's own 1000 x 5 Boston case, and the
// 80-scenario grid behind its figures).
//
// When A fits in one CU's LDS the whole FISTA / FISTA-delta / ISTA loop runs inside ONE launch of ONE workgroup:
// A (fp32) and b stay in LDS, the iterate state (x_k, x_{k-1}, y_k) in LDS doubles, and an iteration is three phases
// separated by workgroup barriers - no kernel boundary, no grid barrier, no host round trip.  The multi-launch path
// costs 15-35 us per iteration on such problems (dependent-kernel turnaround, DESIGN.md "Small problems"); this loop
// costs a few microseconds.  All arithmetic is fp64 on the fp32-stored A (strictly more accurate than the fp32 pass of
// the streaming kernels), and the per-element / scalar formulas are the ones of fista_update_kernel and
// fista_finalize_kernel (reduce_update.hpp), so the device state machine can be entered and left at any iteration:
// grad / trial / update of the host-driven modes keep working on the same FistaScalars and x_cur / x_prev.
//
//   phase A  r_i = A_i . y - b_i            rows strided over the 512 threads
//   phase B  g_j = sum_i A_ij r_i           per-thread partials over its rows, 8 columns at a time, block-reduced
//   phase C  x_next = prox(y - tau g), step norms, momentum / restart / stop      wave 0, lane j owns coordinate j
#pragma once
#include "gemv_pair.hpp"
#include "reduce_update.hpp"

namespace fos {

constexpr int RS_THREADS = 512;
constexpr int RS_WAVES = RS_THREADS / 64;
constexpr int RS_MAX_N = 64;          // one wave owns all coordinates in phase C
constexpr int RS_MAX_M = 4096;        // b in LDS: 16 KiB
constexpr int RS_MAX_A = 10240;       // floats of A in LDS (40 KiB) incl. the odd row stride
constexpr int RS_CHUNK = 8;           // columns reduced per block reduction
constexpr int RS_SMALL_M = 2048;      // n <= RS_CHUNK and m <= this: rows of A live in registers (4 rows per thread)
constexpr int RS_MAX_SHRINKS = 4096;  // exit condition of the Armijo loop every wave reaches (NaN inputs would spin)

// Per-launch options of the resident loop beyond FistaParams (all host-driven features of the multi-launch path).
struct ResidentOpts {
  int backtracking;        // Armijo search per iteration (iterative_solvers.py:183-197, :298-312, :92-108)
  double eta;              // t *= eta on rejection
  double armijo_c;         // the reference's module global C (:11), read by the host at call time
  double grad_tol;         // > 0: stop BEFORE the update when ||grad|| < grad_tol (:179)
  int* ls_out;             // nullable, iters ints: shrink count of each iteration's search
  double* tau_out;         // nullable, iters doubles: step used by each iteration
  double* tau_final;       // nullable, 1 double: step after the last iteration (tau persists, :197)
  int* iters_done;         // nullable, 1 int: iterations completed by this launch
};

__host__ __device__ inline int rs_stride(int n) { return n | 1; }   // odd row stride: conflict-free column walks
inline bool resident_fits(int64_t m, int64_t n) {
  return n >= 1 && n <= RS_MAX_N && m >= 1 && m <= RS_MAX_M && m * rs_stride((int)n) <= RS_MAX_A;
}

// hist (nullable): iters x 4 doubles { ||A x - b||^2, ||x||_1, ||x||_2^2, ||x - x_before||^2 } per iterate;
// x_hist (nullable): iters x n doubles.  Iterations after a stop are not executed (state untouched, rows not written).
// SMALL (n <= RS_CHUNK): every thread keeps its rows of A and b in registers and the loop reads LDS only for the
// n-vectors - the shape of the reference's own problems (n = 5).
template <typename T, bool SMALL>
__global__ __launch_bounds__(RS_THREADS) void fista_resident_kernel(const T* __restrict__ A, int64_t lda,
                                                                   const float* __restrict__ b, int m, int n,
                                                                   double* __restrict__ x_cur, double* __restrict__ x_prev,
                                                                   FistaScalars* __restrict__ scal, FistaParams prm,
                                                                   int iters, double* __restrict__ x_hist,
                                                                   double* __restrict__ hist, ResidentOpts opt) {
  __shared__ float a_s[RS_MAX_A];
  __shared__ float b_s[RS_MAX_M];
  __shared__ double y_s[RS_MAX_N], xc_s[RS_MAX_N], xp_s[RS_MAX_N], g_s[RS_MAX_N], tmp_s[RS_MAX_N];
  __shared__ double sc_s[5];       // ||Ay-b||^2, ||grad||^2, ||y||^2, grad.dlt, ||x_tmp||^2
  __shared__ double red[RS_WAVES][RS_CHUNK + 1];
  __shared__ int stop_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ns = rs_stride(n);
  if (scal->stopped != 0) {
    if (tid == 0) {
      if (opt.iters_done) *opt.iters_done = 0;
      if (opt.tau_final) *opt.tau_final = prm.tau;
    }
    return;
  }

  for (int i = tid; i < m * n; i += RS_THREADS) {
    const int r = i / n, c = i - r * n;
    a_s[r * ns + c] = elem_to_float<T>(A[(int64_t)r * lda + c]);
  }
  for (int i = tid; i < m; i += RS_THREADS) b_s[i] = b ? b[i] : 0.f;
  if (tid < n) {
    const double xc = x_cur[tid], xp = x_prev[tid];
    xc_s[tid] = xc;
    xp_s[tid] = xp;
    y_s[tid] = form_y(xc, xp, scal->beta);
  }
  if (tid == 0) stop_s = 0;
  // scalar state lives in lane 0 of wave 0 for the whole run
  double t_prev = scal->t_prev, this_step = scal->this_step, prev_step = scal->prev_step, ratio = scal->ratio;
  double gnorm2 = scal->gnorm2, x1 = scal->xnorm1, x2 = scal->xnorm2, rr_last = scal->rr, beta = scal->beta;
  long long k = scal->k;
  int restarts = scal->restarts, stop = STOP_NONE, done = 0;
  double tau = prm.tau;
  // smooth l2 weight: in the gradient for the L1 prox (:174-175), inside the prox for PROX_ENET (ista's prox_h)
  const double sa2 = (prm.prox_kind == PROX_L1 && prm.alpha2 > 0.0) ? prm.alpha2 : 0.0;
  __syncthreads();

  // residual of the vector v_s over this thread's rows; r_loc keeps them for phase B
  constexpr int RPT = SMALL ? RS_SMALL_M / RS_THREADS : (RS_MAX_M + RS_THREADS - 1) / RS_THREADS;   // rows per thread: 4 / 8
  double r_loc[RPT];
  float a_reg[SMALL ? RPT : 1][RS_CHUNK], b_reg[SMALL ? RPT : 1];
  if constexpr (SMALL) {
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
      const int row = tid + q * RS_THREADS;
#pragma unroll
      for (int c = 0; c < RS_CHUNK; ++c) a_reg[q][c] = (row < m && c < n) ? a_s[row * ns + c] : 0.f;   // zero rows / columns
      b_reg[q] = row < m ? b_s[row] : 0.f;
    }
  }
  auto residual = [&](const double* v_s) {
    double rr = 0.0;
    if constexpr (SMALL) {
      double v[RS_CHUNK];
#pragma unroll
      for (int c = 0; c < RS_CHUNK; ++c) v[c] = c < n ? v_s[c] : 0.0;
#pragma unroll
      for (int q = 0; q < RPT; ++q) {
        double acc = -(double)b_reg[q];
#pragma unroll
        for (int c = 0; c < RS_CHUNK; ++c) acc += (double)a_reg[q][c] * v[c];
        r_loc[q] = acc;                          // rows beyond m: a = 0, b = 0 -> r = 0
        rr += acc * acc;
      }
      return rr;
    }
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
      const int row = tid + q * RS_THREADS;
      double acc = 0.0;
      if (row < m) {
        const float* ar = a_s + row * ns;
        for (int j = 0; j < n; ++j) acc += (double)ar[j] * v_s[j];
        acc -= (double)b_s[row];
      }
      r_loc[q] = acc;
      rr += acc * acc;
    }
    return rr;
  };
  // block sum of one double per thread -> every thread gets the total (uses red[][RS_CHUNK], two barriers)
  auto block_total = [&](double v) {
    v = wave_sum_dpp(v);
    if (lane == 0) red[wave][RS_CHUNK] = v;
    __syncthreads();
    double tot = 0.0;
#pragma unroll
    for (int w = 0; w < RS_WAVES; ++w) tot += red[w][RS_CHUNK];
    __syncthreads();
    return tot;
  };

  for (int it = 0; it < iters; ++it) {
    // ---- phase A: r = A y - b (per-thread rows; ||r||^2 rides along with the first column chunk's reduction) --------
    double rr_part = residual(y_s);
    // ---- phase B: g = A^T r, RS_CHUNK columns per block reduction ---------------------------------------------------
    for (int c0 = 0; c0 < n; c0 += RS_CHUNK) {
      double p[RS_CHUNK];
#pragma unroll
      for (int c = 0; c < RS_CHUNK; ++c) p[c] = 0.0;
      if constexpr (SMALL) {
#pragma unroll
        for (int q = 0; q < RPT; ++q)
#pragma unroll
          for (int c = 0; c < RS_CHUNK; ++c) p[c] += (double)a_reg[q][c] * r_loc[q];
      } else {
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
          const int row = tid + q * RS_THREADS;
          if (row < m) {
            const float* ar = a_s + row * ns + c0;
#pragma unroll
            for (int c = 0; c < RS_CHUNK; ++c)
              if (c0 + c < n) p[c] += (double)ar[c] * r_loc[q];
          }
        }
      }
      {
        double v[RS_CHUNK + 1];
#pragma unroll
        for (int c = 0; c < RS_CHUNK; ++c) v[c] = p[c];
        v[RS_CHUNK] = rr_part;                           // only chunk 0's value is used
        wave_sum_n(v);
        if (lane == 0) {
#pragma unroll
          for (int c = 0; c <= RS_CHUNK; ++c)
            if (c < RS_CHUNK || c0 == 0) red[wave][c] = v[c];
        }
      }
      __syncthreads();
      if (tid < RS_CHUNK && c0 + tid < n) {
        double tot = 0.0;
#pragma unroll
        for (int w = 0; w < RS_WAVES; ++w) tot += red[w][tid];
        g_s[c0 + tid] = tot;
      }
      if (c0 == 0 && tid == RS_CHUNK) {
        double tot = 0.0;
#pragma unroll
        for (int w = 0; w < RS_WAVES; ++w) tot += red[w][RS_CHUNK];
        sc_s[0] = tot;                                   // ||A y - b||^2
      }
      __syncthreads();
    }
    const double rr = sc_s[0];
    // ---- gradient-norm stop (:179, before the update) and Armijo search (:183-197) -------------------------------
    if (opt.grad_tol > 0.0 || opt.backtracking) {
      if (wave == 0) {
        const double y = lane < n ? y_s[lane] : 0.0;
        const double gfl = lane < n ? g_s[lane] + sa2 * y : 0.0;
        double gy[2] = {gfl * gfl, y * y};
        wave_sum_n(gy);
        if (lane == 0) { sc_s[1] = gy[0]; sc_s[2] = gy[1]; }
      }
      __syncthreads();
      if (opt.grad_tol > 0.0 && sqrt(sc_s[1]) < opt.grad_tol) {      // uniform: every thread reads the same LDS value
        stop = STOP_GRAD;
        break;
      }
      if (opt.backtracking) {
        // the reference's comparison itself, g(x_tmp) <= g(y) + C*grad.(x_tmp - y) with g evaluated twice (:187-191),
        // in fp64 on the LDS copy of A; tau persists across iterations (:197)
        const double g_y = 0.5 * rr + 0.5 * sa2 * sc_s[2];
        int shrinks = 0;
        while (true) {
          if (wave == 0) {
            double xt = 0.0, dl = 0.0, gfl = 0.0;
            if (lane < n) {
              const double y = y_s[lane];
              gfl = g_s[lane] + sa2 * y;
              const double v = y - tau * gfl;
              xt = prm.alpha1 > 0.0 ? soft_threshold(v, tau * prm.alpha1) : v;
              if (prm.prox_kind == PROX_ENET) xt *= 1.0 / (1.0 + tau * prm.alpha2);
              tmp_s[lane] = xt;
              dl = xt - y;
            }
            double gx[2] = {gfl * dl, xt * xt};
            wave_sum_n(gx);
            if (lane == 0) { sc_s[3] = gx[0]; sc_s[4] = gx[1]; }
          }
          __syncthreads();
          const double gd = sc_s[3], x2t = sc_s[4];                  // into registers before the barriers below
          const double rr_t = block_total(residual(tmp_s));
          const double lhs = 0.5 * rr_t + 0.5 * sa2 * x2t;
          if (lhs <= g_y + opt.armijo_c * gd || shrinks >= RS_MAX_SHRINKS) break;
          tau *= opt.eta;
          shrinks += 1;
        }
        if (tid == 0 && opt.ls_out != nullptr) opt.ls_out[it] = shrinks;
      }
    }
    // ---- phase C: prox step, norms, momentum (fista_update_kernel + fista_finalize_kernel) ------------------------
    if (wave == 0) {
      const bool own = lane < n;
      double xn = 0.0, xc = 0.0, d = 0.0, gf = 0.0;
      if (own) {
        const double y = y_s[lane];
        xc = xc_s[lane];
        gf = g_s[lane] + sa2 * y;
        const double v = y - tau * gf;
        xn = prm.alpha1 > 0.0 ? soft_threshold(v, tau * prm.alpha1) : v;
        if (prm.prox_kind == PROX_ENET) xn *= 1.0 / (1.0 + tau * prm.alpha2);
        d = xn - xc;
      }
      double sums[4] = {d * d, gf * gf, fabs(xn), xn * xn};
      wave_sum_n(sums);
      const double s0 = sums[0], s1 = sums[1], s2 = sums[2], s3 = sums[3];
      // every lane computes the same scalars (wave_sum broadcasts), so no further exchange is needed
      const double step = sqrt(s0);
      const double prev = this_step;
      ratio = prev > 0.0 ? step / prev : INFINITY;
      beta = 0.0;
      if (prm.mode == MODE_FISTA) {
        double t_new;
        if (prm.adaptive_restart && ratio > prm.restart_threshold) {
          t_new = 1.0;
          restarts += 1;
        } else {
          t_new = 0.5 * (1.0 + sqrt(1.0 + 4.0 * t_prev * t_prev));
          beta = (t_prev - 1.0) / t_new;
        }
        t_prev = t_new;
      } else if (prm.mode == MODE_DELTA) {
        const double kk = (double)(k + 1);
        beta = kk / (kk + 1.0 + prm.delta);
      }
      prev_step = prev;
      this_step = step;
      gnorm2 = s1;
      x1 = s2;
      x2 = s3;
      rr_last = rr;
      k += 1;
      if (prm.tol_step > 0.0 && step < prm.tol_step) stop = STOP_STEP;
      if (stop == STOP_NONE && prm.tol_ratio > 0.0 && ratio < prm.tol_ratio) stop = STOP_RATIO;
      if (own) {
        xp_s[lane] = xc;
        xc_s[lane] = xn;
        y_s[lane] = form_y(xn, xc, beta);
        if (x_hist != nullptr) x_hist[(int64_t)it * n + lane] = xn;
      }
      if (lane == 0) {
        stop_s = stop;
        if (opt.tau_out != nullptr) opt.tau_out[it] = tau;
        if (hist != nullptr) {
          hist[it * 4 + 1] = s2;
          hist[it * 4 + 2] = s3;
          hist[it * 4 + 3] = s0;
        }
      }
    }
    __syncthreads();
    if (hist != nullptr) {                       // ||A x_next - b||^2: one more sweep over the LDS copy of A
      const double rrx = block_total(residual(xc_s));
      if (tid == 0) hist[it * 4 + 0] = rrx;
    }
    done += 1;
    if (stop_s != STOP_NONE) break;
  }
  if (tid < n) {
    x_cur[tid] = xc_s[tid];
    x_prev[tid] = xp_s[tid];
  }
  if (tid == 0) {
    if (opt.iters_done) *opt.iters_done = done;
    if (opt.tau_final) *opt.tau_final = tau;
    scal->t_prev = t_prev;
    scal->beta = beta;
    scal->this_step = this_step;
    scal->prev_step = prev_step;
    scal->ratio = ratio;
    scal->rr = rr_last;
    scal->gnorm2 = gnorm2;
    scal->xnorm1 = x1;
    scal->xnorm2 = x2;
    scal->k = k;
    scal->stopped = stop;
    scal->restarts = restarts;
  }
}

// One gradient of a small problem in one launch, all in fp64 on the fp32-stored A (L-BFGS fg, lbfgs.py:43-54):
// grad = A^T (A y - b) + alpha2 y (GT = float: rounded to fp32 once, on output; GT = double: fos_gemv_pair_dd), *rr_out = ||A y - b||^2.  Three launches of the
// streaming path (fp32 pass, slab reduce, +alpha2 y) become one, and the line search of L-BFGS - which compares
// objective values that differ in the 7th digit near the solution - sees a float64-accurate f and grad.
template <typename T, typename GT = float>
__global__ __launch_bounds__(RS_THREADS) void gemv_pair_resident_kernel(const T* __restrict__ A, int64_t lda,
                                                                       const float* __restrict__ b, int m, int n,
                                                                       const double* __restrict__ y, double alpha2,
                                                                       GT* __restrict__ grad, double* __restrict__ rr_out) {
  __shared__ double y_s[RS_MAX_N];
  __shared__ double red[RS_WAVES][RS_CHUNK + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < n) y_s[tid] = y[tid];
  __syncthreads();
  constexpr int RPT = (RS_MAX_M + RS_THREADS - 1) / RS_THREADS;
  double r_loc[RPT], rr = 0.0;
#pragma unroll
  for (int q = 0; q < RPT; ++q) {
    const int row = tid + q * RS_THREADS;
    double acc = 0.0;
    if (row < m) {
      const T* ar = A + (int64_t)row * lda;
      for (int j = 0; j < n; ++j) acc += (double)elem_to_float<T>(ar[j]) * y_s[j];
      if (b != nullptr) acc -= (double)b[row];
    }
    r_loc[q] = acc;
    rr += acc * acc;
  }
  for (int c0 = 0; c0 < n; c0 += RS_CHUNK) {
    double v[RS_CHUNK + 1];
#pragma unroll
    for (int c = 0; c <= RS_CHUNK; ++c) v[c] = 0.0;
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
      const int row = tid + q * RS_THREADS;
      if (row < m) {
        const T* ar = A + (int64_t)row * lda + c0;                       // second touch: L1 / L2 hits
#pragma unroll
        for (int c = 0; c < RS_CHUNK; ++c)
          if (c0 + c < n) v[c] += (double)elem_to_float<T>(ar[c]) * r_loc[q];
      }
    }
    v[RS_CHUNK] = rr;
    wave_sum_n(v);
    if (lane == 0) {
#pragma unroll
      for (int c = 0; c <= RS_CHUNK; ++c) red[wave][c] = v[c];
    }
    __syncthreads();
    if (tid <= RS_CHUNK && (tid == RS_CHUNK ? c0 == 0 : c0 + tid < n)) {
      double tot = 0.0;
#pragma unroll
      for (int w = 0; w < RS_WAVES; ++w) tot += red[w][tid];
      if (tid == RS_CHUNK) { if (rr_out != nullptr) *rr_out = tot; }
      else grad[c0 + tid] = (GT)(tot + alpha2 * y_s[c0 + tid]);
    }
    __syncthreads();
  }
}

// Power iteration (iterative_solvers.py:45-60) in the same resident form: v normalised start vector in, L sequence out.
// Lout: n_iter doubles (L after each step); iters_used: index of the step at which |L - prev| < tol fired (+1), or n_iter.
template <typename T>
__global__ __launch_bounds__(RS_THREADS) void power_resident_kernel(const T* __restrict__ A, int64_t lda, int m, int n,
                                                                   float* __restrict__ v_inout, int n_iter, double tol,
                                                                   double* __restrict__ L_out, int* __restrict__ iters_used) {
  __shared__ float a_s[RS_MAX_A];
  __shared__ double v_s[RS_MAX_N], w_s[RS_MAX_N];
  __shared__ double red[RS_WAVES][RS_CHUNK + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ns = rs_stride(n);
  for (int i = tid; i < m * n; i += RS_THREADS) {
    const int r = i / n, c = i - r * n;
    a_s[r * ns + c] = elem_to_float<T>(A[(int64_t)r * lda + c]);
  }
  if (wave == 0) {                                              // v = v0 / ||v0||   (:51)
    const double v0 = lane < n ? (double)v_inout[lane] : 0.0;
    const double nrm = sqrt(wave_sum(v0 * v0));
    if (lane < n) v_s[lane] = v0 / nrm;
  }
  __syncthreads();
  constexpr int RPT = (RS_MAX_M + RS_THREADS - 1) / RS_THREADS;
  double r_loc[RPT];
  double prev = 0.0, L = 0.0;
  int used = n_iter;
  for (int it = 0; it < n_iter; ++it) {
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
      const int row = tid + q * RS_THREADS;
      double acc = 0.0;
      if (row < m) {
        const float* ar = a_s + row * ns;
        for (int j = 0; j < n; ++j) acc += (double)ar[j] * v_s[j];
      }
      r_loc[q] = acc;
    }
    for (int c0 = 0; c0 < n; c0 += RS_CHUNK) {
      double p[RS_CHUNK];
#pragma unroll
      for (int c = 0; c < RS_CHUNK; ++c) p[c] = 0.0;
#pragma unroll
      for (int q = 0; q < RPT; ++q) {
        const int row = tid + q * RS_THREADS;
        if (row < m) {
          const float* ar = a_s + row * ns + c0;
#pragma unroll
          for (int c = 0; c < RS_CHUNK; ++c)
            if (c0 + c < n) p[c] += (double)ar[c] * r_loc[q];
        }
      }
      wave_sum_n(p);
      if (lane == 0) {
#pragma unroll
        for (int c = 0; c < RS_CHUNK; ++c) red[wave][c] = p[c];
      }
      __syncthreads();
      if (tid < RS_CHUNK && c0 + tid < n) {
        double tot = 0.0;
#pragma unroll
        for (int w = 0; w < RS_WAVES; ++w) tot += red[w][tid];
        w_s[c0 + tid] = tot;
      }
      __syncthreads();
    }
    // L = ||w||, v = w / L, break rule (:55-58); every thread evaluates the same numbers from LDS
    double s = 0.0;
    for (int j = 0; j < n; ++j) s += w_s[j] * w_s[j];
    L = sqrt(s);
    __syncthreads();
    if (tid < n) v_s[tid] = w_s[tid] / L;
    if (tid == 0) L_out[it] = L;
    __syncthreads();
    if (fabs(L - prev) < tol) { used = it + 1; break; }
    prev = L;
  }
  if (tid < n) v_inout[tid] = (float)v_s[tid];
  if (tid == 0) *iters_used = used;
}

}  // namespace fos
