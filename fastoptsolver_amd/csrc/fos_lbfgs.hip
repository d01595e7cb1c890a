// libfos_hip.so, translation unit 4 of 4 - L-BFGS under the C ABI (include/fos.h): two-loop / whole-chip direction, the
// fp64 vector kernels, the More-Thuente line search entry points and the native optimiser loop (lbfgs.py:43-73).
#include "fos_internal.hpp"

using namespace fosapi;

extern "C" {

// ---- L-BFGS pieces ---------------------------------------------------------------------------------------
int fos_lbfgs_two_loop(const float* g, const float* S, const float* Y, int hist, int head, int cap, int64_t n,
                       float* d_out, void* stream) {
  if (!g || !d_out || n <= 0 || hist < 0 || hist > fos::LB_MAXHIST || cap < hist || (hist > 0 && (!S || !Y)) ||
      head < 0 || (cap > 0 && head >= cap))
    return fail(FOS_ERR_ARG, "fos_lbfgs_two_loop: bad argument");
  // q in registers when the vectors are float4-addressable and short enough; otherwise the generic form.
  const bool vec = (n % 4 == 0) && ((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(S) |
                                      reinterpret_cast<uintptr_t>(Y) | reinterpret_cast<uintptr_t>(d_out)) & 15u) == 0;
  hipStream_t st = (hipStream_t)stream;
  const int capk = std::max(cap, 1);
#define FOS_TL(NQ) hipLaunchKernelGGL((fos::lbfgs_two_loop_kernel<float, NQ>), dim3(1), dim3(fos::LB_THREADS), 0, st, g, S, Y, \
                                      hist, head, capk, n, d_out)
  if (vec && n <= 4096) FOS_TL(1);
  else if (vec && n <= 8192) FOS_TL(2);
  else if (vec && n <= 16384) FOS_TL(4);
  else FOS_TL(0);
#undef FOS_TL
  LAUNCH_CHECK();
  return FOS_OK;
}

int fos_lbfgs_two_loop_dd(const double* g, const double* S, const double* Y, int hist, int head, int cap, int64_t n,
                          double* d_out, void* stream) {
  if (!g || !d_out || n <= 0 || hist < 0 || hist > fos::LB_MAXHIST || cap < hist || (hist > 0 && (!S || !Y)) ||
      head < 0 || (cap > 0 && head >= cap))
    return fail(FOS_ERR_ARG, "fos_lbfgs_two_loop_dd: bad argument");
  const bool vec = (n % 4 == 0) && ((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(S) |
                                      reinterpret_cast<uintptr_t>(Y) | reinterpret_cast<uintptr_t>(d_out)) & 31u) == 0;
  hipStream_t st = (hipStream_t)stream;
  const int capk = std::max(cap, 1);
#define FOS_TL(NQ) hipLaunchKernelGGL((fos::lbfgs_two_loop_kernel<double, NQ>), dim3(1), dim3(fos::LB_THREADS), 0, st, g, S, \
                                      Y, hist, head, capk, n, d_out)
  if (vec && n <= 4096) FOS_TL(1);
  else if (vec && n <= 8192) FOS_TL(2);
  else if (vec && n <= 16384) FOS_TL(4);
  else FOS_TL(0);
#undef FOS_TL
  LAUNCH_CHECK();
  return FOS_OK;
}

namespace {
inline int vl_parts(int64_t n) { return (int)std::min<int64_t>((n + fos::VL_COLS - 1) / fos::VL_COLS, fos::VL_MAXPARTS); }
// d = -H g on the whole chip (lbfgs_kernels.hpp): Gram matrix of the basis, then coefficients + combination
int launch_direction(const double* g, const double* S, const double* Y, int hist, int head, int cap, int64_t n, double* d_out,
                     double* gd_out, double* work, hipStream_t st, double* x_step = nullptr, double* x_old = nullptr) {
  const int parts = vl_parts(n);
  hipLaunchKernelGGL(fos::lbfgs_gram_kernel, dim3(parts), dim3(fos::VL_THREADS), 0, st, g, S, Y, hist, head, std::max(cap, 1),
                     n, work);
  const int grid = (int)((n + fos::VL_THREADS - 1) / fos::VL_THREADS);      // one column per thread
  hipLaunchKernelGGL(fos::lbfgs_combine_kernel, dim3(grid), dim3(fos::VL_THREADS), 0, st, g, S, Y, hist, head,
                     std::max(cap, 1), n, (const double*)work, parts, d_out, gd_out, x_step, x_old);
  LAUNCH_CHECK();
  return FOS_OK;
}
}  // namespace

int64_t fos_lbfgs_direction_work(int64_t n) { return n > 0 ? (int64_t)vl_parts(n) * fos::VL_PSTRIDE : 0; }

int fos_lbfgs_direction_dd(const double* g, const double* S, const double* Y, int hist, int head, int cap, int64_t n,
                           double* d_out, double* gd_out, double* work, int64_t work_doubles, void* stream) {
  if (!g || !d_out || !work || n <= 0 || hist < 0 || cap < hist || (hist > 0 && (!S || !Y)) || head < 0 ||
      (cap > 0 && head >= cap) || work_doubles < fos_lbfgs_direction_work(n))
    return fail(FOS_ERR_ARG, "fos_lbfgs_direction_dd: bad argument");
  if (hist > fos::VL_MAXH) return fail(FOS_ERR_UNSUPPORTED, "fos_lbfgs_direction_dd: at most 10 pairs (fos_lbfgs_two_loop_dd takes 64)");
  return launch_direction(g, S, Y, hist, head, cap, n, d_out, gd_out, work, (hipStream_t)stream);
}

// Column-sharded problems (fos_problem_set_comm_cols): every basis vector is partitioned over the ranks, so the Gram matrix
// of the basis is a sum over the column blocks - the partial Gram matrices (a fixed-size block of VL_MAXPARTS slots, unused
// slots zero, so that ranks with blocks of different width agree on the count) are all-reduced between the two kernels;
// the coefficient recursion is then replicated and every rank combines its own block of d.  g.d and d.d come out global.
int fos_lbfgs_direction_cols(fos_problem* p, const double* g, const double* S, const double* Y, int hist, int head, int cap,
                             double* d_out, double* gd_out, double* work, int64_t work_doubles) {
  if (!p || !p->col_sharded || !g || !d_out || !work || hist < 0 || cap < hist || (hist > 0 && (!S || !Y)) || head < 0 ||
      (cap > 0 && head >= cap) || work_doubles < (int64_t)fos::VL_MAXPARTS * fos::VL_PSTRIDE)
    return fail(FOS_ERR_ARG, "fos_lbfgs_direction_cols: bad argument (needs a column-sharded problem and 64 x 256 doubles of work)");
  if (hist > fos::VL_MAXH) return fail(FOS_ERR_UNSUPPORTED, "fos_lbfgs_direction_cols: at most 10 pairs");
  const int64_t n = p->n;
  const int parts = vl_parts(n);
  HIP_TRY(hipMemsetAsync(work, 0, (size_t)fos::VL_MAXPARTS * fos::VL_PSTRIDE * sizeof(double), p->stream));
  hipLaunchKernelGGL(fos::lbfgs_gram_kernel, dim3(parts), dim3(fos::VL_THREADS), 0, p->stream, g, S, Y, hist, head,
                     std::max(cap, 1), n, work);
  LAUNCH_CHECK();
  int rc = reduce_across(p, work, (size_t)fos::VL_MAXPARTS * fos::VL_PSTRIDE, true);
  if (rc) return rc;
  const int grid = (int)((n + fos::VL_THREADS - 1) / fos::VL_THREADS);
  hipLaunchKernelGGL(fos::lbfgs_combine_kernel, dim3(grid), dim3(fos::VL_THREADS), 0, p->stream, g, S, Y, hist, head,
                     std::max(cap, 1), n, (const double*)work, (int)fos::VL_MAXPARTS, d_out, gd_out);
  LAUNCH_CHECK();
  return FOS_OK;
}

int fos_vec_stats_dd(const double* x, const double* g, const double* d, int64_t n, double* out5, void* stream) {
  if (!out5 || n <= 0) return fail(FOS_ERR_ARG, "fos_vec_stats_dd: bad argument");
  hipLaunchKernelGGL((fos::vec_stats_kernel<double, double>), dim3(1), dim3(fos::LB_THREADS), 0, (hipStream_t)stream, x, g,
                     d, n, out5);
  LAUNCH_CHECK();
  return FOS_OK;
}

int fos_vec_axpby_dd(double a, const double* x, double b, const double* y, double* out, int64_t n, void* stream) {
  if (!x || !out || n <= 0 || (b != 0.0 && !y)) return fail(FOS_ERR_ARG, "fos_vec_axpby_dd: bad argument");
  hipLaunchKernelGGL(fos::vec_axpby_f64_kernel<double>, dim3(grid_1d(n, 256, 1024)), dim3(256), 0, (hipStream_t)stream, a,
                     x, b, b != 0.0 ? y : nullptr, out, n);
  LAUNCH_CHECK();
  return FOS_OK;
}

double fos_linesearch_begin(fos_linesearch* ls, double stp, double f0, double d0) {
  return ls ? fos_ls_begin_impl(ls, stp, f0, d0) : stp;
}
double fos_linesearch_step(fos_linesearch* ls, double stp, double f, double d) {
  return ls ? fos_ls_step_impl(ls, stp, f, d) : stp;
}

int fos_lbfgs_minimize(fos_problem* p, double alpha2, int max_iter, double pgtol, double* x, double* hist,
                       double* iterates, float* fg_ms, int fg_cap, fos_lbfgs_result* res) {
  if (!p || !x || !res || max_iter < 0) return fail(FOS_ERR_ARG, "fos_lbfgs_minimize: bad argument");
  if (p->col_sharded)
    return fail(FOS_ERR_UNSUPPORTED, "fos_lbfgs_minimize: a column-sharded problem partitions the iterate - every scalar of "
                                     "the iteration is a sum over the ranks; use the driver above the ABI (LBFGSSolver.fit(cols=))");
  constexpr int M = 10, MAXLS = 20;
  constexpr double FACTR = 1e7, EPS = 2.220446049250313e-16;
  const int64_t n = p->n;
  const size_t nb = (size_t)n * sizeof(double);
  hipStream_t st = p->stream;
  if (p->lbfgs == nullptr || p->lbfgs->n != n) {
    delete p->lbfgs;
    p->lbfgs = new LbfgsWork();
    LbfgsWork& nw = *p->lbfgs;
    HIP_TRY(hipMalloc(&nw.g, nb + 8 * sizeof(double)));
    HIP_TRY(hipMalloc(&nw.g_old, nb + 8 * sizeof(double)));
    HIP_TRY(hipMalloc(&nw.d, nb));
    HIP_TRY(hipMalloc(&nw.x_old, nb));
    HIP_TRY(hipMalloc(&nw.S, nb * M));
    HIP_TRY(hipMalloc(&nw.Y, nb * M));
    HIP_TRY(hipMalloc(&nw.vl, (size_t)fos_lbfgs_direction_work(n) * sizeof(double)));
    HIP_TRY(hipHostMalloc(&nw.host, 16 * sizeof(double)));
    HIP_TRY(hipMalloc(&nw.t_start, sizeof(unsigned long long)));
    int dev = 0, khz = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) == hipSuccess && khz > 0) nw.ticks_per_ms = (double)khz;
    nw.n = n;                                  // complete: a partial allocation is rebuilt by the next call
  }
  LbfgsWork& w = *p->lbfgs;
  {
    bool stamps = false;
    int rcq = dd_pass_stamps(p, &stamps);
    if (rcq) return rcq;
    w.pass_stamps = stamps;
  }
  // The scalars of an evaluation cross to the host in pinned memory the kernels write themselves (no copy engine on the
  // round trip):  [0..4] x.x, g.d, d.d, max|g|, ||x||_1   [5] ||r||^2   [6] g.d and [7] d.d of the newest direction
  // [8] sequence number of the evaluation, stored last (system-scope release): the host polls it rather than waiting for
  // the stream to drain, and falls back to hipStreamSynchronize when it has not appeared after a few milliseconds.
  double* host_dev = nullptr;
  HIP_TRY(hipHostGetDevicePointer((void**)&host_dev, w.host, 0));
  unsigned long long* flag_host = reinterpret_cast<unsigned long long*>(w.host + 8);
  unsigned long long* flag_dev = reinterpret_cast<unsigned long long*>(host_dev + 8);
  *flag_host = 0;
  unsigned long long seq = 0;
  auto wait_fg = [&]() -> int {
    const auto t0 = std::chrono::steady_clock::now();
    for (int spin = 0;; ++spin) {
      if (__atomic_load_n(flag_host, __ATOMIC_ACQUIRE) == seq) return FOS_OK;
      if ((spin & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(4)) break;
    }
    HIP_TRY(hipStreamSynchronize(st));
    if (__atomic_load_n(flag_host, __ATOMIC_ACQUIRE) != seq) return fail(FOS_ERR_HIP, "fos_lbfgs_minimize: evaluation did not report");
    return FOS_OK;
  };
  const int ax_grid = grid_1d(n, 256, 1024);
  int nfev = 0;
  double xnorm1 = 0.0;

  // loss and gradient at xv (lbfgs.py:43-54) plus g.d for the line search: enqueue only ...
  auto enqueue_fg = [&](const double* xv, double* gv, const double* dv) -> int {
    // device time of the evaluation for the reference's grad_call_times metric: a wall-clock stamp in front of the pass,
    // read back by the statistics kernel behind it (hipEvents cost a 6.5 us bubble each on this stream: kernel trace)
    // (the streaming fp64 kernel stamps its own start; plans served by other kernels get a one-thread stamp kernel in front)
    const bool timed = fg_ms != nullptr;
    if (timed && !w.pass_stamps) {
      hipLaunchKernelGGL(fos::stamp_kernel, dim3(1), dim3(1), 0, st, w.t_start);
      LAUNCH_CHECK();
    }
    int rc = gemv_pair_dd_stamped(p, xv, alpha2, gv, (timed && w.pass_stamps) ? w.t_start : nullptr, nullptr);
    if (rc) return rc;
    seq += 1;
    hipLaunchKernelGGL((fos::vec_stats_kernel<double, double>), dim3(1), dim3(fos::LB_THREADS), 0, st, xv, (const double*)gv,
                       dv, n, host_dev, (const double*)(gv + n), flag_dev, seq, timed ? w.t_start : nullptr);
    LAUNCH_CHECK();
    return FOS_OK;
  };
  // ... and take its scalars once the stream has drained
  auto take_fg = [&](double* loss, double* gd, double* gmax) {
    if (fg_ms && nfev < fg_cap) fg_ms[nfev] = (float)(w.host[9] / w.ticks_per_ms);
    nfev += 1;
    *loss = 0.5 * w.host[5] + 0.5 * alpha2 * w.host[0];
    *gd = w.host[1];
    *gmax = w.host[3];
    xnorm1 = w.host[4];
  };
  auto axpby = [&](double a, const double* xv, double b, const double* yv, double* out) {
    hipLaunchKernelGGL(fos::vec_axpby_f64_kernel<double>, dim3(ax_grid), dim3(256), 0, st, a, xv, b, b != 0.0 ? yv : nullptr,
                       out, n);
  };
  auto finish = [&](double f, double gmax, int nit, int task) -> int {
    res->f = f; res->gmax = gmax; res->nit = nit; res->nfev = nfev; res->task = task; res->reserved = 0;
    return FOS_OK;
  };

  double *g = w.g, *g_old = w.g_old;
  int hist_n = 0, head = 0, nit = 0;
  double f = 0.0, gd = 0.0, gmax = 0.0;
  int rc = enqueue_fg(x, g, nullptr);
  if (rc) return rc;
  if ((rc = wait_fg())) return rc;
  take_fg(&f, &gd, &gmax);
  if (gmax <= pgtol) return finish(f, gmax, 0, 0);
  for (;;) {
    // direction d = -H g (two-loop recursion over the stored pairs); the kernel leaves g.d and d.d in host[6..7]
    const bool vec = (n % 4 == 0);
#define FOS_TL(NQ) hipLaunchKernelGGL((fos::lbfgs_two_loop_kernel<double, NQ>), dim3(1), dim3(fos::LB_THREADS), 0, st, \
                                      (const double*)g, (const double*)w.S, (const double*)w.Y, hist_n, head, M, n, w.d, \
                                      host_dev + 6)
    // (from the second iteration on the unit first step x_old = x, x += d is written by the combine kernel itself)
    const bool fused_first_trial = n >= 2048 && nit > 0;
    if (n >= 2048) {                            // whole-chip form: two launches, each one read of the history
      if ((rc = launch_direction(g, w.S, w.Y, hist_n, head, M, n, w.d, host_dev + 6, w.vl, st,
                                 fused_first_trial ? x : nullptr, fused_first_trial ? w.x_old : nullptr))) return rc;
    } else if (vec) FOS_TL(1);
    else FOS_TL(0);
#undef FOS_TL
    LAUNCH_CHECK();
    // The first trial step is known without looking at the direction (1 after the first iteration: L-BFGS-B's rule), so
    // the trial point and its evaluation are enqueued behind the two-loop kernel and ONE host round trip serves both.
    double stp = 1.0;
    if (nit == 0) {
      HIP_TRY(hipStreamSynchronize(st));
      if (w.host[6] >= 0.0) return finish(f, gmax, nit, 3);          // not a descent direction and no memory to drop
      stp = std::min(1.0 / std::sqrt(w.host[7]), 1e10);
    }
    if (!fused_first_trial) {
      hipLaunchKernelGGL(fos::lbfgs_first_trial_kernel, dim3(ax_grid), dim3(256), 0, st, x, (const double*)w.d, stp, w.x_old, n);
      LAUNCH_CHECK();
    }
    std::swap(g, g_old);                        // g_old holds the gradient at x_old; g receives the trial gradients
    if ((rc = enqueue_fg(x, g, w.d))) return rc;
    if ((rc = wait_fg())) return rc;
    const double gd0 = w.host[6];
    const double f_old = f, gmax_old = gmax;
    if (gd0 >= 0.0) {                           // not a descent direction: drop the memory (L-BFGS-B info = -4);
      HIP_TRY(hipMemcpyAsync(x, w.x_old, nb, hipMemcpyDeviceToDevice, st));   // the speculative evaluation never happened
      std::swap(g, g_old);
      if (hist_n == 0) return finish(f, gmax, nit, 3);
      hist_n = 0; head = 0;
      continue;
    }
    fos_linesearch ls{};
    stp = fos_ls_begin_impl(&ls, stp, f_old, gd0);
    int evals = 0;
    bool failed = false;
    double gd1 = gd0, stp_used = stp;
    for (;;) {
      if (evals >= MAXLS) { failed = true; break; }
      if (evals > 0) {
        axpby(1.0, w.x_old, stp, w.d, x);       // x = stp*d + x_old, products and sum rounded separately (NumPy's)
        LAUNCH_CHECK();
        if ((rc = enqueue_fg(x, g, w.d))) return rc;
        if ((rc = wait_fg())) return rc;
      }
      take_fg(&f, &gd1, &gmax);
      evals += 1;
      stp_used = stp;
      stp = fos_ls_step_impl(&ls, stp, f, gd1);
      if (ls.status != FOS_LS_FG) break;
    }
    if (failed || ls.status == FOS_LS_ERROR) {
      HIP_TRY(hipMemcpyAsync(x, w.x_old, nb, hipMemcpyDeviceToDevice, st));
      std::swap(g, g_old);
      f = f_old; gmax = gmax_old;
      if (hist_n == 0) return finish(f, gmax, nit, 3);
      hist_n = 0; head = 0;
      continue;
    }
    stp = stp_used;
    if (hist) { hist[2 * nit] = f; hist[2 * nit + 1] = xnorm1; }
    {                                           // record the iterate; keep the pair only if its curvature is positive
      const double sy = (gd1 - gd0) * stp;
      const bool keep_pair = sy > EPS * (-gd0 * stp);
      int slot = 0;
      if (keep_pair) {
        slot = (head + hist_n) % M;
        if (hist_n == M) head = (head + 1) % M;
        else hist_n += 1;
      }
      if (keep_pair || iterates) {
        hipLaunchKernelGGL(fos::lbfgs_store_pair_kernel, dim3(ax_grid), dim3(256), 0, st, stp, (const double*)w.d,
                           (const double*)g, (const double*)g_old, keep_pair ? w.S + (size_t)slot * n : nullptr,
                           keep_pair ? w.Y + (size_t)slot * n : nullptr, (const double*)x,
                           iterates ? iterates + (size_t)nit * n : nullptr, n);
        LAUNCH_CHECK();
      }
    }
    nit += 1;
    if (nit >= max_iter) return finish(f, gmax, nit, 2);
    if (gmax <= pgtol) return finish(f, gmax, nit, 0);
    if ((f_old - f) <= EPS * FACTR * std::max(std::max(std::fabs(f_old), std::fabs(f)), 1.0)) return finish(f, gmax, nit, 1);
  }
}

int fos_vec_stats(const float* x, const float* g, const float* d, int64_t n, double* out5, void* stream) {
  if (!out5 || n <= 0) return fail(FOS_ERR_ARG, "fos_vec_stats: bad argument");
  hipLaunchKernelGGL(fos::vec_stats_kernel<float>, dim3(1), dim3(fos::LB_THREADS), 0, (hipStream_t)stream, x, g, d, n,
                     out5);
  LAUNCH_CHECK();
  return FOS_OK;
}

int fos_vec_stats_f64(const double* x, const float* g, const float* d, int64_t n, double* out5, void* stream) {
  if (!out5 || n <= 0) return fail(FOS_ERR_ARG, "fos_vec_stats_f64: bad argument");
  hipLaunchKernelGGL(fos::vec_stats_kernel<double>, dim3(1), dim3(fos::LB_THREADS), 0, (hipStream_t)stream, x, g, d, n,
                     out5);
  LAUNCH_CHECK();
  return FOS_OK;
}

int fos_vec_axpby_f64(double a, const double* x, double b, const float* y, double* out, int64_t n, void* stream) {
  if (!x || !out || n <= 0 || (b != 0.0 && !y)) return fail(FOS_ERR_ARG, "fos_vec_axpby_f64: bad argument");
  hipLaunchKernelGGL(fos::vec_axpby_f64_kernel<float>, dim3(grid_1d(n, 256, 1024)), dim3(256), 0, (hipStream_t)stream, a, x, b,
                     b != 0.0 ? y : nullptr, out, n);
  LAUNCH_CHECK();
  return FOS_OK;
}

int fos_vec_axpby(double a, const float* x, double b, const float* y, float* out, int64_t n, void* stream) {
  if (!x || !out || n <= 0 || (b != 0.0 && !y)) return fail(FOS_ERR_ARG, "fos_vec_axpby: bad argument");
  hipLaunchKernelGGL(fos::vec_axpby_kernel, dim3(grid_1d(n, 256, 1024)), dim3(256), 0, (hipStream_t)stream, (float)a, x,
                     (float)b, b != 0.0 ? y : nullptr, out, n);
  LAUNCH_CHECK();
  return FOS_OK;
}

}  // extern "C"
