// libfos_hip.so, translation unit 3 of 4 - the FISTA / ISTA / FISTA-delta state machine of the C ABI (include/fos.h):
// fused, split, recorded, backtracking, resident and lockstep (multi-lambda) runs.  iterative_solvers.py:65-344.
#include "fos_internal.hpp"
#include "fused_step.hpp"
#include "chip_resident.hpp"

using namespace fosapi;

template <int NQ>
static int launch_fused(const fos::FusedArgs& a, int G, size_t lds, hipStream_t st) {
  auto kern = fos::fista_fused_kernel<NQ>;
  static std::atomic<uint64_t> done{0};
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  const uint64_t bit = 1ull << (dev & 63);
  if (!(done.load(std::memory_order_acquire) & bit)) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    done.fetch_or(bit, std::memory_order_release);
  }
  // a plain launch: G = #CUs workgroups of 512 threads and ~136 KiB of LDS are co-resident by grid size (one per CU), which
  // is all hipLaunchCooperativeKernel would check; every grid-wide wait in the kernel is bounded
  hipLaunchKernelGGL(kern, dim3(G), dim3(fos::FZ_THREADS), lds, st, a);
  LAUNCH_CHECK();
  return FOS_OK;
}

template <int NC, bool CTRL>
static int launch_chip(const fos::ChipArgs& a, int G, size_t lds, hipStream_t st) {
  auto kern = fos::fista_chip_resident_kernel<NC, CTRL>;
  static std::atomic<uint64_t> done{0};
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  const uint64_t bit = 1ull << (dev & 63);
  if (!(done.load(std::memory_order_acquire) & bit)) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, fos::CR_LDS_BUDGET));
    done.fetch_or(bit, std::memory_order_release);
  }
  // a plain launch of at most #CUs workgroups with up to 150 KiB of LDS each: co-resident by grid size; every wait is bounded
  hipLaunchKernelGGL(kern, dim3(G), dim3(fos::CR_THREADS), lds, st, a);
  LAUNCH_CHECK();
  return FOS_OK;
}

extern "C" {

// ---- FISTA ---------------------------------------------------------------------------------------------
int fos_fista_create(fos_problem* p, fos_fista** out) {
  if (!p || !out) return fail(FOS_ERR_ARG, "fos_fista_create: null");
  fos_fista* f = new fos_fista();
  f->p = p;
  const size_t nb = (size_t)p->n * sizeof(double);
  hipError_t he = hipMalloc(&f->x_cur, nb);
  if (he == hipSuccess) he = hipMalloc(&f->x_prev, nb);
  if (he == hipSuccess) he = hipMalloc(&f->dlt, (size_t)p->n * sizeof(float));
  if (he == hipSuccess) he = hipMalloc(&f->scal, sizeof(fos::FistaScalars));
  if (he == hipSuccess) he = hipMalloc(&f->out5, 8 * sizeof(double));
  f->nupd = (int)((p->n + fos::RCOLS - 1) / fos::RCOLS);
  if (he == hipSuccess) he = hipMalloc(&f->part2, (size_t)2 * f->nupd * 4 * sizeof(double));
  if (he == hipSuccess) he = hipMalloc(&f->ynext, (size_t)p->n * sizeof(float));
  if (he != hipSuccess) {
    fos_fista_destroy(f);
    return fail(FOS_ERR_HIP, std::string("fos_fista_create: ") + hipGetErrorString(he));
  }
  *out = f;
  return FOS_OK;
}

int fos_fista_destroy(fos_fista* f) {
  if (!f) return FOS_OK;
  void* bufs[] = {f->x_cur, f->x_prev, f->dlt, f->scal, f->out5, f->part2, f->ynext, f->gbuf64_owned ? f->gbuf64 : nullptr, f->folded};
  for (void* q : bufs)
    if (q) (void)hipFree(q);
  delete f;
  return FOS_OK;
}

static void to_dev_params(const fos_fista_params* s, fos::FistaParams* d) {
  d->alpha1 = s->alpha1;
  d->alpha2 = s->alpha2;
  d->tau = s->tau;
  d->mode = s->mode;
  d->prox_kind = s->prox_kind;
  d->delta = s->delta;
  d->adaptive_restart = s->adaptive_restart;
  d->restart_threshold = s->restart_threshold;
  d->tol_step = s->tol_step;
  d->tol_ratio = s->tol_ratio;
  d->tol_grad = s->tol_grad;
  d->tau_from_state = 0;
}

int fos_fista_reset(fos_fista* f, const fos_fista_params* prm, const double* x0) {
  if (!f || !prm) return fail(FOS_ERR_ARG, "fos_fista_reset: null");
  if (prm->mode < 0 || prm->mode > 2 || prm->prox_kind < 0 || prm->prox_kind > 1 || !(prm->tau > 0.0) ||
      prm->tol_grad < 0.0 || prm->tol_step < 0.0 || prm->tol_ratio < 0.0)
    return fail(FOS_ERR_ARG, "fos_fista_reset: bad mode/prox_kind/tau/tolerance");
  to_dev_params(prm, &f->prm);
  fos_problem* p = f->p;
  const size_t nb = (size_t)p->n * sizeof(double);
  if (x0) {
    HIP_TRY(hipMemcpyAsync(f->x_cur, x0, nb, hipMemcpyDeviceToDevice, p->stream));
    HIP_TRY(hipMemcpyAsync(f->x_prev, x0, nb, hipMemcpyDeviceToDevice, p->stream));
  } else {
    HIP_TRY(hipMemsetAsync(f->x_cur, 0, nb, p->stream));
    HIP_TRY(hipMemsetAsync(f->x_prev, 0, nb, p->stream));
  }
  hipLaunchKernelGGL(fos::fista_init_scalars_kernel, dim3(1), dim3(1), 0, p->stream, f->scal);
  LAUNCH_CHECK();
  f->host_valid = true;
  f->tau_on_device = false;
  f->y_valid = false;
  f->pending = false;
  f->plain_count = 0;
  f->h_t = 1.0;
  f->h_beta = 0.0;
  f->h_k = 0;
  return FOS_OK;
}

int fos_fista_set_tau(fos_fista* f, double tau) {
  if (!f || !(tau > 0.0)) return fail(FOS_ERR_ARG, "fos_fista_set_tau: bad argument");
  f->prm.tau = tau;
  f->tau_on_device = false;
  return FOS_OK;
}

static fos::GradSrc grad_src(const fos_fista* f) {
  return fos::GradSrc{f->p->gbuf, f->precise ? f->gbuf64 : nullptr};
}

int fos_fista_set_precise(fos_fista* f, int on) {
  if (!f) return fail(FOS_ERR_ARG, "fos_fista_set_precise: null");
  if (on && !f->gbuf64) {
    HIP_TRY(hipMalloc(&f->gbuf64, (size_t)(f->p->n + 4) * sizeof(double)));
    f->gbuf64_owned = true;
  }
  f->precise = on != 0;
  return FOS_OK;
}

int fos_fista_set_gbuf64(fos_fista* f, double* buf) {
  if (!f) return fail(FOS_ERR_ARG, "fos_fista_set_gbuf64: null");
  if (buf && (reinterpret_cast<uintptr_t>(buf) & 15u)) return fail(FOS_ERR_ARG, "fos_fista_set_gbuf64: misaligned");
  if (f->gbuf64 && f->gbuf64_owned) (void)hipFree(f->gbuf64);
  f->gbuf64 = buf;
  f->gbuf64_owned = false;
  if (!buf) f->precise = false;
  return FOS_OK;
}

__global__ void rr_from_gbuf64_kernel(const double* __restrict__ g64, int n, double* __restrict__ rr_out, const int* stopped) {
  if (stopped != nullptr && *stopped != 0) return;
  *rr_out = g64[n];
}

static YSource fista_source(fos_fista* f) {
  return YSource{nullptr, f->x_cur, f->x_prev, &f->scal->beta, &f->scal->stopped, 0.0};
}

__global__ __launch_bounds__(64) void fold4_kernel(const double* __restrict__ part, int nparts, double* __restrict__ out4,
                                                   const int* stopped) {
  if (stopped != nullptr && *stopped != 0) return;
  double s[4] = {0.0, 0.0, 0.0, 0.0};
  for (int i = threadIdx.x; i < nparts; i += 64)
    for (int j = 0; j < 4; ++j) s[j] += part[i * 4 + j];
  for (int j = 0; j < 4; ++j) s[j] = fos::wave_sum(s[j]);
  if (threadIdx.x == 0)
    for (int j = 0; j < 4; ++j) out4[j] = s[j];
}

// Column-sharded lockstep: the step partials of every weight folded to [weight][4] (re-derived every iteration - the
// in-place all-reduce behind it must never see its own result, stopped weights included).
__global__ __launch_bounds__(64) void fold4_multi_kernel(fos::MultiControl mc, int nparts, double* __restrict__ out) {
  const int v = blockIdx.x;
  const double* part = mc.part[v];
  double s[4] = {0.0, 0.0, 0.0, 0.0};
  for (int i = threadIdx.x; i < nparts; i += 64)
    for (int j = 0; j < 4; ++j) s[j] += part[i * 4 + j];
  for (int j = 0; j < 4; ++j) s[j] = fos::wave_sum(s[j]);
  if (threadIdx.x == 0)
    for (int j = 0; j < 4; ++j) out[v * 4 + j] = s[j];
}

static int launch_finalize(fos_fista* f, int n_rr, double* hist_row = nullptr) {
  fos_problem* p = f->p;
  const double* part = p->part;
  int nparts = f->nupd;
  if (p->col_sharded) {
    // x is partitioned over the ranks: step norms, ||grad||^2, ||x||_1, ||x||^2 are sums over ALL column blocks
    if (!f->folded) HIP_TRY(hipMalloc(&f->folded, 8 * sizeof(double)));
    hipLaunchKernelGGL(fold4_kernel, dim3(1), dim3(64), 0, p->stream, p->part, f->nupd, f->folded, (const int*)nullptr);   // re-derived after a stop: the in-place all-reduce below must never see its own result
    LAUNCH_CHECK();
    int rc = reduce_across(p, f->folded, 4, true);
    if (rc) return rc;
    part = f->folded;
    nparts = 1;
  }
  hipLaunchKernelGGL(fos::fista_finalize_kernel, dim3(1), dim3(64), 0, p->stream, part, nparts, p->rr_part, n_rr,
                     f->scal, f->prm, hist_row);
  LAUNCH_CHECK();
  return FOS_OK;
}

// Momentum of the iteration that follows iteration index k (0-based), given t_k: iterative_solvers.py:215-216, :330.
static void host_momentum(const fos::FistaParams& prm, long long k, double* t, double* beta) {
  if (prm.mode == fos::MODE_FISTA) {
    const double t_new = 0.5 * (1.0 + std::sqrt(1.0 + 4.0 * (*t) * (*t)));
    *beta = (*t - 1.0) / t_new;
    *t = t_new;
  } else if (prm.mode == fos::MODE_DELTA) {
    const double kk = (double)(k + 1);
    *beta = kk / (kk + 1.0 + prm.delta);
  } else {
    *beta = 0.0;
  }
}

static void launch_update_from_slabs(fos_fista* f, double* part, int host_beta, double beta_val,
                                     double* x_hist = nullptr, float* y_next = nullptr, double beta_next = 0.0,
                                     const float* slabs = nullptr, int64_t slab_stride = 0, int nslabs = 0,
                                     int y_mode = fos::YOUT_VECTOR, int y_slot = 0) {
  fos_problem* p = f->p;
  if (slabs == nullptr) slabs = p->slabs;
  if (slab_stride == 0) slab_stride = p->slab_stride;
  if (nslabs == 0) nslabs = p->nslabs;
  if (p->vec4)
    hipLaunchKernelGGL((fos::fista_update_kernel<true, true>), dim3(f->nupd), dim3(256), 0, p->stream, slabs,
                       nslabs, fos::GradSrc{nullptr, nullptr}, (int)p->n, f->x_cur, f->x_prev, f->scal, f->prm, part, host_beta,
                       beta_val, x_hist, y_next, beta_next, slab_stride, y_mode, y_slot);
  else
    hipLaunchKernelGGL((fos::fista_update_kernel<true, false>), dim3(f->nupd), dim3(256), 0, p->stream, slabs,
                       nslabs, fos::GradSrc{nullptr, nullptr}, (int)p->n, f->x_cur, f->x_prev, f->scal, f->prm, part, host_beta,
                       beta_val, x_hist, y_next, beta_next, slab_stride, y_mode, y_slot);
}

// y source of a plain-run iteration: the fp32 vector the previous update kernel wrote, or (first iteration after a
// reset / split-mode call) the fp64 state with the host's beta.  Both give bit-identical y.
static YSource plain_source(fos_fista* f) {
  if (f->y_valid) return YSource{f->ynext, nullptr, nullptr, nullptr, &f->scal->stopped, 0.0, nullptr};
  return YSource{nullptr, f->x_cur, f->x_prev, nullptr, &f->scal->stopped, f->h_beta, nullptr};
}

// Bring the device scalars up to date after plain split-mode updates (their bookkeeping is deferred so that a
// sharded run pays two launches + one collective per iteration).  n_rr = 0: rr was written by slab_reduce.
static int flush_pending(fos_fista* f) {
  if (!f->pending) return FOS_OK;
  fos_problem* p = f->p;
  const size_t psz = (size_t)f->nupd * 4;
  const long long last = f->h_k - 1;
  const double* cur = f->part2 + (size_t)(last & 1) * psz;
  const double* prev = f->plain_count >= 2 ? f->part2 + (size_t)((last - 1) & 1) * psz : nullptr;
  hipLaunchKernelGGL(fos::fista_finalize_plain_kernel, dim3(1), dim3(64), 0, p->stream, cur, prev, f->nupd, p->rr_part, 0,
                     f->scal, f->h_t, f->h_beta, f->h_k);
  LAUNCH_CHECK();
  f->pending = false;
  return FOS_OK;
}

static bool plain_run(const fos_fista* f) {
  return !(f->prm.mode == fos::MODE_FISTA && f->prm.adaptive_restart) && f->prm.tol_step == 0.0 &&
         f->prm.tol_ratio == 0.0 && f->prm.tol_grad == 0.0;
}

// The gradient-norm stop sits between the reduced gradient and the update (fos_fista_params.tol_grad).
static int launch_grad_norm_stop(fos_fista* f) {
  fos_problem* p = f->p;
  if (p->col_sharded) {                        // ||grad||^2 = sum over the column blocks of all ranks
    if (!f->folded) HIP_TRY(hipMalloc(&f->folded, 8 * sizeof(double)));
    hipLaunchKernelGGL(fos::grad_norm_stop_kernel, dim3(1), dim3(1024), 0, p->stream, grad_src(f), (int)p->n, f->x_cur,
                       f->x_prev, f->scal, f->prm, f->folded + 4);
    LAUNCH_CHECK();
    int rc = reduce_across(p, f->folded + 4, 1, true);
    if (rc) return rc;
    hipLaunchKernelGGL(fos::grad_norm_decide_kernel, dim3(1), dim3(1), 0, p->stream, f->folded + 4, f->scal, f->prm);
    LAUNCH_CHECK();
    return FOS_OK;
  }
  hipLaunchKernelGGL(fos::grad_norm_stop_kernel, dim3(1), dim3(1024), 0, p->stream, grad_src(f), (int)p->n, f->x_cur, f->x_prev,
                     f->scal, f->prm);
  LAUNCH_CHECK();
  return FOS_OK;
}

static int refresh_host_scalars(fos_fista* f, bool* stopped) {
  *stopped = false;
  if (f->host_valid) return FOS_OK;
  fos_fista_status st;
  int rc = fos_fista_status_get(f, &st);      // synchronises once after split-mode / device-driven calls
  if (rc) return rc;
  *stopped = st.stopped != FOS_STOP_NONE;
  f->h_t = st.t_prev; f->h_beta = st.beta; f->h_k = st.k;
  f->host_valid = true;
  return FOS_OK;
}

// Whole run in ONE launch of ONE workgroup (resident.hpp): A, b and the iterate state stay in LDS.
static int run_resident(fos_fista* f, int iters, double* x_hist, double* hist, fos::ResidentOpts opt = fos::ResidentOpts{}) {
  fos_problem* p = f->p;
  if (opt.grad_tol == 0.0) opt.grad_tol = f->prm.tol_grad;     // the handle's own gradient-norm stop (:179)
  int rc = flush_pending(f);                   // device scalars must be current: the kernel continues from them
  if (rc) return rc;
  const bool small = p->n <= fos::RS_CHUNK && p->m <= fos::RS_SMALL_M;    // rows of A in registers (resident.hpp)
#define FOS_RS_LAUNCH(T, SMALL)                                                                                          \
  hipLaunchKernelGGL((fos::fista_resident_kernel<T, SMALL>), dim3(1), dim3(fos::RS_THREADS), 0, p->stream,               \
                     (const T*)p->A, p->lda, p->b, (int)p->m, (int)p->n, f->x_cur, f->x_prev, f->scal, f->prm, iters,    \
                     x_hist, hist, opt)
  if (p->dtype == FOS_F32) { if (small) FOS_RS_LAUNCH(float, true); else FOS_RS_LAUNCH(float, false); }
  else { if (small) FOS_RS_LAUNCH(fos::bf16_t, true); else FOS_RS_LAUNCH(fos::bf16_t, false); }
#undef FOS_RS_LAUNCH
  LAUNCH_CHECK();
  f->host_valid = false;                       // t, beta, k now live on the device only
  f->y_valid = false;
  f->plain_count = 0;
  return FOS_OK;
}

int fos_fista_run_resident(fos_fista* f, int iters, int backtracking, double eta, double armijo_c, double grad_tol,
                           double* x_hist, double* hist, int32_t* ls_iters, double* tau_hist, int32_t* iters_done,
                           double* tau_out) {
  if (!f || iters < 0 || !iters_done || !tau_out || (backtracking && !(eta > 0.0 && eta < 1.0)))
    return fail(FOS_ERR_ARG, "fos_fista_run_resident: bad argument");
  fos_problem* p = f->p;
  if (!p->resident) return fail(FOS_ERR_UNSUPPORTED, "fos_fista_run_resident: problem does not fit the LDS-resident loop");
  *iters_done = 0;
  *tau_out = f->prm.tau;
  if (iters == 0) return FOS_OK;
  double* tau_dev = p->dscal + 241;
  int* done_dev = reinterpret_cast<int*>(p->dscal + 242);
  fos::ResidentOpts opt{backtracking ? 1 : 0, eta, armijo_c, grad_tol, ls_iters, tau_hist, tau_dev, done_dev};
  int rc = run_resident(f, iters, x_hist, hist, opt);
  if (rc) return rc;
  int done = 0;
  double tau = f->prm.tau;
  HIP_TRY(hipMemcpyAsync(&done, done_dev, sizeof(int), hipMemcpyDeviceToHost, p->stream));
  HIP_TRY(hipMemcpyAsync(&tau, tau_dev, sizeof(double), hipMemcpyDeviceToHost, p->stream));
  HIP_TRY(hipStreamSynchronize(p->stream));
  f->prm.tau = tau;                            // tau persists (iterative_solvers.py:197)
  *iters_done = done;
  *tau_out = tau;
  return FOS_OK;
}

int64_t fos_fista_history_workspace(fos_fista* f, int iters) {
  if (!f || iters < 0) return -1;
  return ((int64_t)(iters + 1) * f->p->nwg + (int64_t)iters * f->nupd * 4) * (int64_t)sizeof(double);
}

int fos_fista_run_history(fos_fista* f, int iters, double* x_hist, double* hist, void* work) {
  if (!f || iters < 0 || (iters > 0 && (!x_hist || !hist || !work)))
    return fail(FOS_ERR_ARG, "fos_fista_run_history: bad argument");
  fos_problem* p = f->p;
  if (plain_run(f) && p->resident) return iters == 0 ? FOS_OK : run_resident(f, iters, x_hist, hist);
  if (!plain_run(f) || p->path != 0 || p->colblock || p->entry->dual == nullptr || p->comm != nullptr)
    return fail(FOS_ERR_UNSUPPORTED, "fos_fista_run_history: needs a plain run on the fused path with a DUAL kernel");
  if (iters == 0) return FOS_OK;
  bool stopped = false;
  int rc = refresh_host_scalars(f, &stopped);
  if (rc) return rc;
  if (stopped) return fail(FOS_ERR_STATE, "fos_fista_run_history: solver already stopped");
  const int nwg = p->nwg;
  double* rr2_slots = reinterpret_cast<double*>(work);                 // (iters + 1) x nwg
  double* part_slots = rr2_slots + (size_t)(iters + 1) * nwg;          // iters x nupd x 4
  double* saved_rr2 = p->rr2_part;
  int n_rr = 0;
  if ((rc = flush_pending(f))) return rc;
  f->y_valid = false;      // the DUAL pass needs x_k itself, so it always rebuilds y from the fp64 state
  f->plain_count = 0;
  for (int it = 0; it < iters; ++it) {
    YSource ys{nullptr, f->x_cur, f->x_prev, nullptr, &f->scal->stopped, f->h_beta, nullptr};
    p->rr2_part = rr2_slots + (size_t)it * nwg;                        // slot it = residual of the iterate BEFORE it
    rc = launch_pass(p, ys, p->b, true, &n_rr, true);
    p->rr2_part = saved_rr2;
    if (rc) return rc;
    launch_update_from_slabs(f, part_slots + (size_t)it * f->nupd * 4, 1, f->h_beta, x_hist + (size_t)it * p->n);
    LAUNCH_CHECK();
    host_momentum(f->prm, f->h_k, &f->h_t, &f->h_beta);
    f->h_k += 1;
  }
  // closing residual pass: ||A x_last - b||^2 -> slot iters (written by the residual-only kernel into rr_part)
  hipLaunchKernelGGL(fos::cast_f64_f32_kernel, dim3(grid_1d(p->n, 256, 1024)), dim3(256), 0, p->stream, f->x_cur, p->ybuf,
                     p->n);
  LAUNCH_CHECK();
  {
    YSource ys{p->ybuf, nullptr, nullptr, nullptr, nullptr};
    double* saved_rr = p->rr_part;
    p->rr_part = rr2_slots + (size_t)iters * nwg;
    rc = launch_pass(p, ys, p->b, false, &n_rr);
    p->rr_part = saved_rr;
    if (rc) return rc;
  }
  hipLaunchKernelGGL(fos::history_fold_kernel, dim3(iters), dim3(64), 0, p->stream, rr2_slots, nwg, part_slots, f->nupd,
                     hist);
  LAUNCH_CHECK();
  // device scalars: step norms of the last two iterations, momentum from the host
  const double* cur = part_slots + (size_t)(iters - 1) * f->nupd * 4;
  const double* prev = iters >= 2 ? part_slots + (size_t)(iters - 2) * f->nupd * 4 : nullptr;
  hipLaunchKernelGGL(fos::fista_finalize_plain_kernel, dim3(1), dim3(64), 0, p->stream, cur, prev, f->nupd, p->rr_part,
                     nwg, f->scal, f->h_t, f->h_beta, f->h_k);
  LAUNCH_CHECK();
  return FOS_OK;
}

int fos_fista_run(fos_fista* f, int iters) {
  if (!f || iters < 0) return fail(FOS_ERR_ARG, "fos_fista_run: bad argument");
  fos_problem* p = f->p;
  if (iters == 0) return FOS_OK;
  if (p->resident) return run_resident(f, iters, nullptr, nullptr);
  // Tall-skinny runs without backtracking / history: A in the LDS of up to all CUs, one grid barrier per iteration
  // (chip_resident.hpp); adaptive restart and the step / ratio stops are decided on the device by every workgroup alike.  The
  // planner's own region is where the plain loop measured at least 1.4x ahead of the two launches below (tools/bench_chip.py:
  // 9000 ... 100000 x 5 5.6-7.7 us per iteration against 10.8-11.2; runs with restart save a third launch on top);
  // FOS_PLAN_CHIP_RESIDENT / FOS_PLAN_NO_CHIP_RESIDENT widen it to every served shape / switch it off.
  {
    // (9..16 columns - the 16-column instantiation carries twice the registers: 20000 x 16 1.2x plain, 1.6x with restart, even at
    //  100000 x 16)
    const bool chip_region = p->dtype == FOS_F32 && p->m >= 512 && iters >= 8 &&
                             (p->n <= 8 ? p->m <= 131072 : (p->n <= 16 && p->m <= 32768));
    if ((p->chip_mode == 1 || (p->chip_mode == 0 && chip_region)) && !p->comm && !f->precise && !f->prm.tau_from_state &&
        f->prm.tol_grad == 0.0) {
      const int rcc = fos_fista_run_chip(f, iters);
      // not served, or its grid could not become co-resident within the bound (state untouched): the loops below
      if (rcc != FOS_ERR_UNSUPPORTED && rcc != FOS_ERR_STATE) return rcc;
    }
  }
  if (f->prm.tol_grad > 0.0 && !p->comm) {
    // gradient-norm stop: K2 -> slab reduce (gbuf) -> norm check -> update from gbuf -> finalize, all enqueued
    for (int it = 0; it < iters; ++it) {
      int rc;
      if ((rc = fos_fista_grad(f))) return rc;
      if ((rc = launch_grad_norm_stop(f))) return rc;
      if ((rc = fos_fista_update(f))) return rc;
    }
    return FOS_OK;
  }
  if (p->comm) {
    // Row-sharded problem: K2 on this rank's rows -> slab reduction -> all-reduce of [gradient ; ||r||^2] (n + 1 floats)
    // -> prox + momentum from the reduced gradient, all enqueued on one stream; every rank applies the identical fp64
    // update to identical numbers, so the replicated iterates stay bit-identical.
    for (int it = 0; it < iters; ++it) {
      int rc;
      if ((rc = fos_fista_grad(f))) return rc;
      if (f->prm.tol_grad > 0.0 && (rc = launch_grad_norm_stop(f))) return rc;
      if ((rc = fos_fista_update(f))) return rc;
    }
    return flush_pending(f);
  }
  // Plain run: no data-dependent control (adaptive restart / stopping tolerances).  t_k and beta_k are then a fixed
  // sequence: the host passes beta_k to both kernels by value, and the scalar bookkeeping kernel runs once per call
  // instead of once per iteration (two launches per iteration instead of three).
  if (plain_run(f) && p->fused_on && !f->precise && !f->prm.tau_from_state) {      // opt-in: the one-launch persistent step
    const int rcf = fos_fista_run_fused(f, iters);
    if (rcf != FOS_ERR_UNSUPPORTED) return rcf;
  }
  if (plain_run(f)) {
    bool stopped = false;
    int rc0 = refresh_host_scalars(f, &stopped);
    if (rc0) return rc0;
    if (stopped) return FOS_OK;
    const size_t psz = (size_t)f->nupd * 4;
    int n_rr = 0;
    for (int it = 0; it < iters; ++it) {
      int rc;
      if ((rc = launch_pass(p, plain_source(f), p->b, true, &n_rr))) return rc;
      const double beta_k = f->h_beta;
      host_momentum(f->prm, f->h_k, &f->h_t, &f->h_beta);          // beta_{k+1}
      launch_update_from_slabs(f, f->part2 + (size_t)(f->h_k & 1) * psz, 1, beta_k, nullptr, f->ynext, f->h_beta);
      LAUNCH_CHECK();
      f->y_valid = true;
      f->h_k += 1;
      f->plain_count += 1;
    }
    f->pending = false;
    const long long last = f->h_k - 1;
    const double* cur = f->part2 + (size_t)(last & 1) * psz;
    const double* prev = f->plain_count >= 2 ? f->part2 + (size_t)((last - 1) & 1) * psz : nullptr;
    hipLaunchKernelGGL(fos::fista_finalize_plain_kernel, dim3(1), dim3(64), 0, p->stream, cur, prev, f->nupd, p->rr_part,
                       n_rr, f->scal, f->h_t, f->h_beta, f->h_k);
    LAUNCH_CHECK();
    return FOS_OK;
  }
  f->host_valid = false;
  f->y_valid = false;
  f->plain_count = 0;
  for (int it = 0; it < iters; ++it) {
    int n_rr = 0, rc;
    if ((rc = launch_pass(p, fista_source(f), p->b, true, &n_rr))) return rc;
    launch_update_from_slabs(f, p->part, 0, 0.0);
    LAUNCH_CHECK();
    if ((rc = launch_finalize(f, n_rr))) return rc;
  }
  return FOS_OK;
}

// The step in ONE persistent launch with the row dots on the matrix cores and A staged through LDS (fused_step.hpp):
int fos_problem_set_fused_stamps(fos_problem* p, unsigned long long* stamps) {
  if (!p) return fail(FOS_ERR_ARG, "fos_problem_set_fused_stamps: null");
  p->fz_stamps = stamps;
  return FOS_OK;
}

// BASELINE north_star's literal design, opt-in (the two-launch VALU step measures faster).  Plain runs only.
int fos_fista_run_fused(fos_fista* f, int iters) {
  if (!f || iters < 0) return fail(FOS_ERR_ARG, "fos_fista_run_fused: bad argument");
  fos_problem* p = f->p;
  const int G = p->ncu;
  if (p->dtype != FOS_F32 || p->path != 0 || p->tall || p->colblock || p->resident || p->comm || p->n % 2048 != 0 ||
      p->n > 8192 || p->lda % 4 != 0 || (reinterpret_cast<uintptr_t>(p->A) & 15u) || p->m < 8 * (int64_t)G ||
      (p->n + G - 1) / G > fos::FZ_OWN_MAX)
    return fail(FOS_ERR_UNSUPPORTED, "fos_fista_run_fused: fp32 A, n in {2048, 4096, 6144, 8192}, aligned rows, m >= 8 x CUs, unsharded");
  if (!plain_run(f) || f->precise || f->prm.tau_from_state)
    return fail(FOS_ERR_UNSUPPORTED, "fos_fista_run_fused: plain runs only (no adaptive restart, tolerances, device-held step)");
  if (iters == 0) return FOS_OK;
  int rc = flush_pending(f);
  if (rc) return rc;
  bool stopped = false;
  if ((rc = refresh_host_scalars(f, &stopped))) return rc;
  if (stopped) return FOS_OK;
  // workspace: G slabs (the planner's are reused when it planned G workgroups), barrier words, beta sequence, partials
  if (G > p->slab_cap) {
    if (p->slabs) (void)hipFree(p->slabs);
    p->slabs = nullptr; p->slab_cap = 0;
    HIP_TRY(hipMalloc(&p->slabs, (size_t)G * p->n * sizeof(float)));
    p->slab_cap = G;
  }
  if (G > p->rr_cap) {
    if (p->rr_part) (void)hipFree(p->rr_part);
    if (p->rr2_part) (void)hipFree(p->rr2_part);
    p->rr_part = p->rr2_part = nullptr; p->rr_cap = 0;
    HIP_TRY(hipMalloc(&p->rr_part, (size_t)G * sizeof(double)));
    HIP_TRY(hipMalloc(&p->rr2_part, (size_t)G * sizeof(double)));
    p->rr_cap = G;
  }
  if (!p->fz_bar) {
    HIP_TRY(hipMalloc(&p->fz_bar, fos::FZ_BAR_WORDS * sizeof(unsigned)));
    HIP_TRY(hipMemsetAsync(p->fz_bar, 0, fos::FZ_BAR_WORDS * sizeof(unsigned), p->stream));
    HIP_TRY(hipMalloc(&p->fz_part, (size_t)2 * G * 4 * sizeof(double)));
  }
  if (iters + 1 > p->fz_beta_cap) {
    if (p->fz_beta) (void)hipFree(p->fz_beta);
    p->fz_beta = nullptr; p->fz_beta_cap = 0;
    HIP_TRY(hipMalloc(&p->fz_beta, (size_t)(iters + 1) * sizeof(double)));
    p->fz_beta_cap = iters + 1;
  }
  if (!f->y_valid) {
    hipLaunchKernelGGL(fos::form_y_kernel, dim3(grid_1d(p->n, 256, 256)), dim3(256), 0, p->stream, f->x_cur, f->x_prev, f->h_beta,
                       f->ynext, p->n);
    LAUNCH_CHECK();
    f->y_valid = true;
  }
  // the momentum sequence of a plain run does not depend on the data: beta[k] for y_k, beta[k + 1] ... beta[iters]
  std::vector<double> betas((size_t)iters + 1);
  betas[0] = f->h_beta;
  const long long k0 = f->h_k;
  for (int k = 0; k < iters; ++k) {
    host_momentum(f->prm, f->h_k, &f->h_t, &f->h_beta);
    betas[(size_t)k + 1] = f->h_beta;
    f->h_k += 1;
  }
  HIP_TRY(hipMemcpyAsync(p->fz_beta, betas.data(), betas.size() * sizeof(double), hipMemcpyHostToDevice, p->stream));
  HIP_TRY(hipStreamSynchronize(p->stream));          // (the host vector must outlive the copy)
  fos::FusedArgs a{};
  a.A = (const float*)p->A; a.lda = p->lda; a.b = p->b; a.m = p->m; a.n = (int)p->n;
  a.rows_per_wg = ((p->m + G - 1) / G + fos::FZ_ROWS - 1) / fos::FZ_ROWS * fos::FZ_ROWS;
  a.slabs = p->slabs; a.y = f->ynext; a.x_cur = f->x_cur; a.x_prev = f->x_prev; a.beta = p->fz_beta; a.part = p->fz_part;
  a.rr_part = p->rr_part; a.bar = p->fz_bar; a.iters = iters; a.prox_kind = f->prm.prox_kind; a.k0 = k0;
  a.tau = f->prm.tau; a.alpha1 = f->prm.alpha1; a.alpha2 = f->prm.alpha2;
  a.timeout_ticks = 100000000ull * 2ull;           // 2 s of the 100 MHz wall clock per wait
  a.stamps = p->fz_stamps;
  if ((rc = prof_mark(p, true))) return rc;
  const size_t lds = fos::fz_lds_bytes((int)p->n);
  switch ((int)(p->n / 2048)) {
    case 1: rc = launch_fused<1>(a, G, lds, p->stream); break;
    case 2: rc = launch_fused<2>(a, G, lds, p->stream); break;
    case 3: rc = launch_fused<3>(a, G, lds, p->stream); break;
    default: rc = launch_fused<4>(a, G, lds, p->stream); break;
  }
  if (rc) return rc;
  if ((rc = prof_mark(p, false))) return rc;
  f->pending = false;
  f->plain_count += iters;
  const long long last = f->h_k - 1;
  const double* cur = p->fz_part + (size_t)(last & 1) * G * 4;
  const double* prev = iters >= 2 ? p->fz_part + (size_t)((last - 1) & 1) * G * 4 : nullptr;
  hipLaunchKernelGGL(fos::fista_finalize_plain_kernel, dim3(1), dim3(64), 0, p->stream, cur, prev, G, p->rr_part, G, f->scal,
                     f->h_t, f->h_beta, f->h_k);
  LAUNCH_CHECK();
  f->plain_count = 0;                              // part2 of the two-launch path starts afresh
  // a grid-wide wait that ran out leaves the state invalid: report it (synchronises)
  unsigned bad = 0;
  HIP_TRY(hipMemcpyAsync(&bad, p->fz_bar + fos::FZ_LINE, sizeof(unsigned), hipMemcpyDeviceToHost, p->stream));
  HIP_TRY(hipStreamSynchronize(p->stream));
  if (bad) return fail(FOS_ERR_STATE, "fos_fista_run_fused: a grid-wide wait timed out (workgroups not co-resident?); state invalid");
  return FOS_OK;
}

// Tall-skinny plain runs with A resident in the LDS of up to all CUs, one grid barrier per iteration (chip_resident.hpp). Opt-in.
int fos_fista_run_chip(fos_fista* f, int iters) {
  if (!f || iters < 0) return fail(FOS_ERR_ARG, "fos_fista_run_chip: bad argument");
  fos_problem* p = f->p;
  const int nc = p->n <= 8 ? 8 : 16;
  const int64_t cap = fos::cr_rows_cap(nc);
  if (p->dtype != FOS_F32 || p->n > 16 || p->comm || p->m < 512 || p->m > cap * (int64_t)p->ncu)
    return fail(FOS_ERR_UNSUPPORTED, "fos_fista_run_chip: fp32 A, n <= 16, 512 <= m <= rows that fit the LDS of all CUs, unsharded");
  // plain runs take the momentum sequence from the host; adaptive restart and the step / ratio stops are decided on the device
  // by every workgroup alike (CTRL); the gradient-norm rule, the fp64 split gradient and a device-held step are not served
  if (f->precise || f->prm.tau_from_state || f->prm.tol_grad > 0.0)
    return fail(FOS_ERR_UNSUPPORTED, "fos_fista_run_chip: no gradient-norm rule, fp64 split gradient or device-held step (backtracking)");
  const bool ctrl = !plain_run(f);
  if (iters == 0) return FOS_OK;
  int rc = flush_pending(f);
  if (rc) return rc;
  bool stopped = false;
  if (!ctrl) {
    if ((rc = refresh_host_scalars(f, &stopped))) return rc;
    if (stopped) return FOS_OK;
  }
  // about 1024 rows (four per thread) per workgroup, at most what its LDS holds: the barrier, not the pass, is the cost, and it
  // grows with the number of workgroups (tools/bench_chip.py: 100000 x 5 took 11.9 us per iteration on 256 workgroups)
  int64_t G = std::max<int64_t>(1, std::min<int64_t>(p->ncu, (p->m + 1023) / 1024));
  G = std::max<int64_t>(G, std::min<int64_t>(p->ncu, (p->m + cap - 1) / cap));
  int64_t rpw = (p->m + G - 1) / G;
  if (rpw > cap) return fail(FOS_ERR_UNSUPPORTED, "fos_fista_run_chip: rows per workgroup exceed the LDS budget");
  G = (p->m + rpw - 1) / rpw;
  if (!p->cr_part) {
    HIP_TRY(hipMalloc(&p->cr_part, ((size_t)2 * p->ncu * 17 + 16) * sizeof(double)));
    HIP_TRY(hipMalloc(&p->cr_bar, fos::FZ_BAR_WORDS * sizeof(unsigned)));
    HIP_TRY(hipMemsetAsync(p->cr_bar, 0, fos::FZ_BAR_WORDS * sizeof(unsigned), p->stream));
  }
  if (iters + 1 > p->fz_beta_cap) {
    if (p->fz_beta) (void)hipFree(p->fz_beta);
    p->fz_beta = nullptr; p->fz_beta_cap = 0;
    HIP_TRY(hipMalloc(&p->fz_beta, (size_t)(iters + 1) * sizeof(double)));
    p->fz_beta_cap = iters + 1;
  }
  std::vector<double> betas((size_t)iters + 1);
  betas[0] = f->h_beta;
  const long long k_before = f->h_k;
  const double t_before = f->h_t, beta_before = f->h_beta;
  for (int k = 0; k < iters; ++k) {
    host_momentum(f->prm, f->h_k, &f->h_t, &f->h_beta);
    betas[(size_t)k + 1] = f->h_beta;
    f->h_k += 1;
  }
  HIP_TRY(hipMemcpyAsync(p->fz_beta, betas.data(), betas.size() * sizeof(double), hipMemcpyHostToDevice, p->stream));
  HIP_TRY(hipStreamSynchronize(p->stream));          // (the host vector must outlive the copy)
  double* stats = p->cr_part + (size_t)2 * p->ncu * 17;
  fos::ChipArgs a{};
  a.A = (const float*)p->A; a.lda = p->lda; a.b = p->b; a.m = p->m; a.n = (int)p->n; a.rows_per_wg = rpw;
  a.part = p->cr_part; a.x_cur = f->x_cur; a.x_prev = f->x_prev; a.beta = p->fz_beta; a.stats = stats; a.rr_out = stats + 8;
  a.bar = p->cr_bar; a.iters = iters; a.prox_kind = f->prm.prox_kind;
  a.tau = f->prm.tau; a.alpha1 = f->prm.alpha1; a.alpha2 = f->prm.alpha2;
  a.timeout_ticks = 100000000ull * 2ull;
  if ((rc = prof_mark(p, true))) return rc;
  const size_t lds = fos::cr_lds_bytes(nc, rpw);
  a.scal = f->scal; a.prm = f->prm;
  if (ctrl) rc = nc == 8 ? launch_chip<8, true>(a, (int)G, lds, p->stream) : launch_chip<16, true>(a, (int)G, lds, p->stream);
  else rc = nc == 8 ? launch_chip<8, false>(a, (int)G, lds, p->stream) : launch_chip<16, false>(a, (int)G, lds, p->stream);
  if (rc) return rc;
  if ((rc = prof_mark(p, false))) return rc;
  // A grid-wide wait that ran out (workgroups not co-resident: another kernel held CUs for longer than the bound) ends the
  // launch BEFORE anything is written back: the iterate on the device is the one the call started from.  The handle's
  // momentum counters are put back, the barrier words cleared, and the caller is told - it can run the two-launch loop.
  unsigned bad = 0;
  HIP_TRY(hipMemcpyAsync(&bad, p->cr_bar + fos::FZ_LINE, sizeof(unsigned), hipMemcpyDeviceToHost, p->stream));
  HIP_TRY(hipStreamSynchronize(p->stream));
  if (bad) {
    f->h_k = k_before; f->h_t = t_before; f->h_beta = beta_before;
    HIP_TRY(hipMemsetAsync(p->cr_bar, 0, fos::FZ_BAR_WORDS * sizeof(unsigned), p->stream));
    return fail(FOS_ERR_STATE, "fos_fista_run_chip: a grid-wide wait timed out (workgroups not co-resident); the state is "
                               "the one before the call");
  }
  f->pending = false;
  f->y_valid = false;                              // the fp32 y vector of the two-launch path is not maintained here
  f->plain_count = 0;
  if (ctrl) {                                      // the kernel advanced FistaScalars itself; the host's mirrors are stale
    f->host_valid = false;
    f->h_k = k_before; f->h_t = t_before; f->h_beta = beta_before;
    return FOS_OK;
  }
  hipLaunchKernelGGL(fos::fista_finalize_plain_kernel, dim3(1), dim3(64), 0, p->stream, stats, iters >= 2 ? stats + 4 : (const double*)nullptr,
                     1, stats + 8, 1, f->scal, f->h_t, f->h_beta, f->h_k);
  LAUNCH_CHECK();
  return FOS_OK;
}

// Up to 16 state machines in lockstep on the matrix cores (gram_batch.hpp): per iteration and per row panel, product 1
// (R = A_panel Y - b, from HBM) and product 2 (G += R^T A_panel, the panel again from the Infinity Cache), then one
// update kernel per state machine, which leaves its y_{k+1} in the candidate block of the next product 1.
static int run_multi_mfma(fos_fista* const* fs, int nv, int iters, bool controlled = false) {
  fos_problem* p = fs[0]->p;
  int rc = ensure_batch_workspace(p);
  if (rc) return rc;
  const bool is_bf16 = p->dtype == FOS_BF16;
  const int64_t esz = is_bf16 ? 2 : 4;
  // One-read form (cluster_pass.hpp): fp32, 2049..16384 columns in strips of 1024 -> 4, 8 or 16 members per cluster, all
  // CUs busy, at least 8 panels per cluster; FOS_PLAN_CLUSTER / FOS_PLAN_NO_CLUSTER force it on (where served) / off.
  // Planner default (profiles/r03_cluster_crossover.md, 16 weights, us per iteration, two products -> one read): the
  // one-read form wins where every member of a cluster has a full 1024-column strip and the matrix is large -
  // 131072 x 4096 826 -> 640, 524288 x 4096 3306 -> 2407, 65536 x 8192 670 -> 643, 262144 x 8192 2668 -> 2430 - ties at
  // 32768 x 8192 and loses with idle members (6144 columns +10 %, 3072 +5 %) and at 16 members (131072 x 16384 +5 %).
  const int cs_need = (int)((p->n + fos::CP_W - 1) / fos::CP_W);
  const bool cluster_wins = (p->n == 4096 || p->n == 8192) && p->m * p->n >= (1ll << 29) && !p->comm;
  const bool want_cluster = p->cp_mode == 1 || (p->cp_mode == 0 && cluster_wins);
  if (!p->rbuf16 && want_cluster && !is_bf16 && p->ncu % 8 == 0) {
    const int cs = cs_need <= 2 ? 0 : cs_need <= 4 ? 4 : cs_need <= 8 ? 8 : cs_need <= 16 ? 16 : 0;
    if (cs && (p->ncu / 8) % cs == 0 && p->m >= (int64_t)(p->ncu / cs) * fos::CP_ROWS * 8) {
      p->cp_cs = cs;
      p->cp_clusters = p->ncu / cs;
      p->cp_rows_per_cluster = ((p->m + p->cp_clusters - 1) / p->cp_clusters + fos::CP_ROWS - 1) / fos::CP_ROWS * fos::CP_ROWS;
      HIP_TRY(hipMalloc(&p->cp_xchg, (size_t)p->ncu * fos::CP_SLOTS * 256 * sizeof(float)));
      HIP_TRY(hipMalloc(&p->cp_flags, (size_t)p->ncu * fos::CP_FLAG_STRIDE * sizeof(unsigned)));
      HIP_TRY(hipMemsetAsync(p->cp_flags, 0, (size_t)p->ncu * fos::CP_FLAG_STRIDE * sizeof(unsigned), p->stream));
      HIP_TRY(hipMalloc(&p->cp_error, sizeof(int)));
      HIP_TRY(hipMemsetAsync(p->cp_error, 0, sizeof(int), p->stream));
    }
  }
  const bool cols = p->col_sharded;
  if (cols && !controlled) return fail(FOS_ERR_STATE, "run_multi_mfma: a column-sharded lockstep is device-controlled");
  if (cols && !p->mfold) HIP_TRY(hipMalloc(&p->mfold, (size_t)fos::BT_NV * 4 * sizeof(double)));
  if (!p->rbuf16) {
    // Panel: product 1 gives a workgroup 64-128 whole rows, so it needs >= 128 * CUs * 2 rows to fill the chip; row
    // splits of product 2: enough (strip, split) workgroups for two per CU.  (A panel that fits the Infinity Cache
    // - ~3000 rows at n = 8192 - would need a split-K product 1; see DESIGN.md "Multi-lambda".)
    const int64_t rows = 256 * (int64_t)p->ncu;
    p->panel_rows = std::min<int64_t>(rows, (p->m + 255) / 256 * 256);
    if (cols && p->comm->kind != 0) {          // mesh transport: a panel's 16 residual columns are one message
      const int64_t fit = (int64_t)(p->comm->cap_bytes / (fos::BT_NV * sizeof(float))) / 256 * 256;
      if (fit < 256) return fail(FOS_ERR_ARG, "fos_fista_run_multi: the communicator's inbox rows hold less than one 256-row "
                                              "panel of 16 residual columns (16 KiB)");
      p->panel_rows = std::min<int64_t>(p->panel_rows, fit);
    }
    const int64_t strips = (p->n + (is_bf16 ? fos::GQ_COLS : fos::GB_COLS) - 1) / (is_bf16 ? fos::GQ_COLS : fos::GB_COLS);
    int64_t splits = std::max<int64_t>(1, (2 * (int64_t)p->ncu + strips - 1) / strips);
    splits = std::min<int64_t>(splits, std::max<int64_t>(1, p->panel_rows / 256));
    p->gram_rows_per_split = ((p->panel_rows + splits - 1) / splits + fos::GB_ROWS - 1) / fos::GB_ROWS * fos::GB_ROWS;
    p->gram_splits = (int)((p->panel_rows + p->gram_rows_per_split - 1) / p->gram_rows_per_split);
    if (p->cp_cs) p->gram_splits = p->cp_clusters;      // one slab set per cluster
    HIP_TRY(hipMalloc(&p->rbuf16, (size_t)p->panel_rows * fos::BT_NV * sizeof(float)));
    HIP_TRY(hipMalloc(&p->slabs16, (size_t)p->gram_splits * fos::BT_NV * p->n * sizeof(float)));
  }
  // candidate block: zero everywhere (padding columns, unused slots), then y_k of every state machine
  const size_t per_entry = is_bf16 ? 3 * sizeof(unsigned short) : sizeof(float);
  HIP_TRY(hipMemsetAsync(p->xp, 0, (size_t)p->n_pad * fos::BT_NV * per_entry, p->stream));
  // Controlled run (adaptive restart / step or ratio tolerance on any weight): momentum and stops are decided on the
  // device per state machine, every iteration; a stopped weight is a masked column of the block.
  fos::MultiControl mc{};
  if (controlled) {
    for (int v = 0; v < nv; ++v) {
      fos_fista* f = fs[v];
      if ((rc = flush_pending(f))) return rc;
      f->host_valid = false; f->y_valid = false; f->plain_count = 0;
      mc.scal[v] = f->scal; mc.part[v] = f->part2; mc.x_cur[v] = f->x_cur; mc.x_prev[v] = f->x_prev;
      mc.adaptive_restart[v] = f->prm.adaptive_restart; mc.restart_threshold[v] = f->prm.restart_threshold;
      mc.tol_step[v] = f->prm.tol_step; mc.tol_ratio[v] = f->prm.tol_ratio;
    }
    hipLaunchKernelGGL(fos::form_y_multi_kernel, dim3(grid_1d(p->n, 256, 64), nv), dim3(256), 0, p->stream, mc, (int)p->n, p->xp,
                       is_bf16 ? fos::YOUT_XQ : fos::YOUT_XP, 1);
    LAUNCH_CHECK();
  }
  for (int v = 0; v < nv && !controlled; ++v) {
    fos_fista* f = fs[v];
    if ((rc = flush_pending(f))) return rc;
    bool stopped = false;
    if ((rc = refresh_host_scalars(f, &stopped))) return rc;
    if (stopped) return fail(FOS_ERR_STATE, "fos_fista_run_multi: a handle has already stopped");
    hipLaunchKernelGGL(fos::form_y_block_kernel, dim3(grid_1d(p->n, 256, 256)), dim3(256), 0, p->stream, f->x_cur, f->x_prev,
                       f->h_beta, (int)p->n, v, is_bf16 ? (float*)nullptr : p->xp,
                       is_bf16 ? (unsigned short*)p->xp : (unsigned short*)nullptr);
    LAUNCH_CHECK();
    f->y_valid = false;              // the fp32 y vector of the single-vector path is not maintained here
    f->plain_count = 0;
  }
  const size_t psz = (size_t)fs[0]->nupd * 4;
  const int64_t strips = (p->n + (is_bf16 ? fos::GQ_COLS : fos::GB_COLS) - 1) / (is_bf16 ? fos::GQ_COLS : fos::GB_COLS);
  // one update launch for all state machines when they differ in weights and steps only (a regularisation path does)
  bool same_family = true;
  for (int v = 1; v < nv; ++v) {
    const fos::FistaParams &a = fs[0]->prm, &c = fs[v]->prm;
    same_family = same_family && a.mode == c.mode && a.prox_kind == c.prox_kind && a.delta == c.delta;
  }
  for (int it = 0; it < iters; ++it) {
    if ((rc = prof_mark(p, true))) return rc;
    if (p->cp_cs) {
      if ((rc = launch_cluster_pass(p))) {
        if (p->cp_mode == 1 || it > 0) return rc;
        (void)hipGetLastError();                 // planner's own choice refused (cooperative launch): two products instead
        p->cp_cs = 0;
        p->cp_mode = 2;
        HIP_TRY(hipStreamSynchronize(p->stream));
        (void)hipFree(p->rbuf16); (void)hipFree(p->slabs16);       // sized for the cluster form: re-planned by the re-entry
        p->rbuf16 = p->slabs16 = nullptr;
        return run_multi_mfma(fs, nv, iters, controlled);
      }
    } else
    for (int64_t row0 = 0, panel = 0; row0 < p->m; row0 += p->panel_rows, ++panel) {
      const int64_t rows = std::min<int64_t>(p->panel_rows, p->m - row0);
      const char* Ap = reinterpret_cast<const char*>(p->A) + (size_t)row0 * p->lda * esz;
      int nwg1 = 0;
      // column-sharded: b enters the sum over the ranks once (rank 0); R = sum_p A_p Y_p - b is the ONE exchange per panel
      const float* bp = (p->b && !(cols && p->comm->rank != 0)) ? p->b + row0 : nullptr;
      if ((rc = launch_batch_product(p, Ap, bp, rows, 1, p->rbuf16, &nwg1))) return rc;
      if (cols && (rc = reduce_across(p, p->rbuf16, (size_t)rows * fos::BT_NV, false))) return rc;
      const dim3 grid((unsigned)strips, (unsigned)p->gram_splits);
#define FOS_GRAM(T, ACC)                                                                                                  \
  hipLaunchKernelGGL((fos::gram_batch_mfma_kernel<T, ACC>), grid, dim3(fos::GB_THREADS), 0, p->stream, (const T*)Ap, p->lda, \
                     rows, (int)p->n, p->rbuf16, p->gram_rows_per_split, p->slabs16, p->n)
      if (is_bf16) {
        if (panel)
          hipLaunchKernelGGL(fos::gram_batch_mfma_bf16_kernel<true>, grid, dim3(fos::GB_THREADS), 0, p->stream,
                             (const fos::bf16_t*)Ap, p->lda, rows, (int)p->n, p->rbuf16, p->gram_rows_per_split, p->slabs16, p->n);
        else
          hipLaunchKernelGGL(fos::gram_batch_mfma_bf16_kernel<false>, grid, dim3(fos::GB_THREADS), 0, p->stream,
                             (const fos::bf16_t*)Ap, p->lda, rows, (int)p->n, p->rbuf16, p->gram_rows_per_split, p->slabs16, p->n);
      } else { if (panel) FOS_GRAM(float, true); else FOS_GRAM(float, false); }
#undef FOS_GRAM
      LAUNCH_CHECK();
    }
    if ((rc = prof_mark(p, false))) return rc;
    // row-sharded problem: the 16 partial gradients (all row splits) are summed over the ranks before the updates
    // (column-sharded: the gradient block is local)
    if (!cols && (rc = reduce_across(p, p->slabs16, (size_t)p->gram_splits * fos::BT_NV * p->n, false))) return rc;
    if (controlled) {                            // update (device beta) -> bookkeeping of all weights -> their y_{k+1}
      fos::MultiUpdate mu{};
      for (int v = 0; v < nv; ++v) {
        fos_fista* f = fs[v];
        mu.x_cur[v] = f->x_cur; mu.x_prev[v] = f->x_prev; mu.scal[v] = f->scal; mu.part[v] = f->part2;
        mu.alpha1[v] = f->prm.alpha1; mu.alpha2[v] = f->prm.alpha2; mu.tau[v] = f->prm.tau;
      }
      hipLaunchKernelGGL(fos::fista_update_multi_kernel, dim3(fs[0]->nupd, nv), dim3(256), 0, p->stream, p->slabs16,
                         p->gram_splits, (int)p->n, mu, fs[0]->prm, p->xp, is_bf16 ? fos::YOUT_XQ : fos::YOUT_XP, 0);
      LAUNCH_CHECK();
      if (cols) {                                // step norms, ||x||^2, ||x||_1 are sums over the column blocks of all ranks
        hipLaunchKernelGGL(fold4_multi_kernel, dim3(nv), dim3(64), 0, p->stream, mc, fs[0]->nupd, p->mfold);
        LAUNCH_CHECK();
        if ((rc = reduce_across(p, p->mfold, (size_t)nv * 4, true))) return rc;
        fos::MultiControl mf = mc;
        for (int v = 0; v < nv; ++v) mf.part[v] = p->mfold + (size_t)v * 4;
        hipLaunchKernelGGL(fos::fista_finalize_multi_kernel, dim3(nv), dim3(64), 0, p->stream, mf, 1, fs[0]->prm);
      } else
        hipLaunchKernelGGL(fos::fista_finalize_multi_kernel, dim3(nv), dim3(64), 0, p->stream, mc, fs[0]->nupd, fs[0]->prm);
      LAUNCH_CHECK();
      hipLaunchKernelGGL(fos::form_y_multi_kernel, dim3(grid_1d(p->n, 256, 64), nv), dim3(256), 0, p->stream, mc, (int)p->n, p->xp,
                         is_bf16 ? fos::YOUT_XQ : fos::YOUT_XP, 0);
      LAUNCH_CHECK();
    } else if (same_family) {                    // one launch updates all state machines
      fos::MultiUpdate mu{};
      for (int v = 0; v < nv; ++v) {
        fos_fista* f = fs[v];
        mu.x_cur[v] = f->x_cur; mu.x_prev[v] = f->x_prev; mu.scal[v] = f->scal;
        mu.part[v] = f->part2 + (size_t)(f->h_k & 1) * psz;
        mu.beta[v] = f->h_beta;
        host_momentum(f->prm, f->h_k, &f->h_t, &f->h_beta);
        mu.beta_next[v] = f->h_beta;
        mu.alpha1[v] = f->prm.alpha1; mu.alpha2[v] = f->prm.alpha2; mu.tau[v] = f->prm.tau;
        f->h_k += 1;
        f->plain_count += 1;
      }
      hipLaunchKernelGGL(fos::fista_update_multi_kernel, dim3(fs[0]->nupd, nv), dim3(256), 0, p->stream, p->slabs16,
                         p->gram_splits, (int)p->n, mu, fs[0]->prm, p->xp, is_bf16 ? fos::YOUT_XQ : fos::YOUT_XP);
      LAUNCH_CHECK();
    } else {
      for (int v = 0; v < nv; ++v) {
        fos_fista* f = fs[v];
        const double beta_k = f->h_beta;
        host_momentum(f->prm, f->h_k, &f->h_t, &f->h_beta);
        launch_update_from_slabs(f, f->part2 + (size_t)(f->h_k & 1) * psz, 1, beta_k, nullptr, p->xp, f->h_beta,
                                 p->slabs16 + (size_t)v * p->n, (int64_t)fos::BT_NV * p->n, p->gram_splits,
                                 is_bf16 ? fos::YOUT_XQ : fos::YOUT_XP, v);
        LAUNCH_CHECK();
        f->h_k += 1;
        f->plain_count += 1;
      }
    }
  }
  if (p->cp_cs) {        // a cluster member that waited out its bound parked itself and raised the flag: the sums are invalid
    int bad = 0;
    HIP_TRY(hipMemcpyAsync(&bad, p->cp_error, sizeof(int), hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    if (bad) return fail(FOS_ERR_STATE, "fos_fista_run_multi: the one-read cluster pass timed out waiting for a member "
                                        "(results invalid); rerun with FOS_PLAN_NO_CLUSTER");
  }
  for (int v = 0; v < nv && !controlled; ++v) {
    fos_fista* f = fs[v];
    f->pending = false;
    const long long last = f->h_k - 1;
    const double* cur = f->part2 + (size_t)(last & 1) * psz;
    const double* prev = f->plain_count >= 2 ? f->part2 + (size_t)((last - 1) & 1) * psz : nullptr;
    hipLaunchKernelGGL(fos::fista_finalize_plain_kernel, dim3(1), dim3(64), 0, p->stream, cur, prev, f->nupd, p->rr_part, 0,
                       f->scal, f->h_t, f->h_beta, f->h_k);
    LAUNCH_CHECK();
    f->plain_count = 0;              // part2 of the next single-vector run starts afresh
  }
  return FOS_OK;
}

int fos_fista_run_multi(fos_fista* const* fs, int nv, int iters) {
  if (!fs || nv < 1 || nv > fos::BT_NV || iters < 0) return fail(FOS_ERR_ARG, "fos_fista_run_multi: bad argument");
  for (int v = 0; v < nv; ++v)
    if (!fs[v] || fs[v]->p != fs[0]->p) return fail(FOS_ERR_ARG, "fos_fista_run_multi: handles must share one problem");
  if (nv == 1) return fos_fista_run(fs[0], iters);
  fos_problem* p = fs[0]->p;
  bool all_plain = true, controllable = true, same_family = true;
  for (int v = 0; v < nv; ++v) {
    all_plain = all_plain && plain_run(fs[v]);
    // what the lockstep bookkeeping decides on the device: adaptive restart, step and ratio tolerances (the gradient-norm
    // rule sits BEFORE the update and backtracking needs its own candidates per weight: those run one by one)
    controllable = controllable && fs[v]->prm.tol_grad == 0.0 && !fs[v]->precise && !fs[v]->prm.tau_from_state;
    const fos::FistaParams &a = fs[0]->prm, &c = fs[v]->prm;
    same_family = same_family && a.mode == c.mode && a.prox_kind == c.prox_kind && a.delta == c.delta;
  }
  const bool shape_ok = batch_supported(p) && !p->colblock && !p->resident && !p->col_sharded;
  // Column-sharded (very wide A): the two products per panel with ONE exchange of the panel's 16 residual columns between
  // them; always device-controlled (step norms are sums over the ranks).  The matrix-core kernels tile any width.
  if (p->col_sharded) {
    if (!(controllable && same_family))
      return fail(FOS_ERR_UNSUPPORTED, "fos_fista_run_multi: a column-sharded lockstep serves one family without the "
                                       "gradient-norm rule / fp64 gradient / persisted step");
    if (iters == 0) return FOS_OK;
    return run_multi_mfma(fs, nv, iters, true);
  }
  if (!all_plain && controllable && same_family && shape_ok && p->entry != wide_entry(p->dtype) && (nv >= 3 || p->comm)) {
    if (iters == 0) return FOS_OK;
    return run_multi_mfma(fs, nv, iters, true);
  }
  const bool streaming = shape_ok && all_plain;
  // (a sharded problem takes the matrix-core pass for any number of weights: its 16 gradients are one 16 x n all-reduce)
  MultiLaunch fn = (streaming && !p->tall && !p->comm && p->dtype == FOS_F32 && p->entry != wide_entry(p->dtype)) ? find_multi(p->n, nv) : nullptr;
  // the two-product pass costs about two single-vector passes per iteration whatever the number of weights: it pays
  // from three weights on (profiles/r02_multilambda.md); two weights without a VALU multi-vector kernel run one by one
  if (!fn && streaming && p->entry != wide_entry(p->dtype) && (nv >= 3 || p->comm)) {
    if (iters == 0) return FOS_OK;
    return run_multi_mfma(fs, nv, iters);          // 5..16 weights, n up to 16384, fp32 and bf16
  }
  if (!fn) return fail(FOS_ERR_UNSUPPORTED, "fos_fista_run_multi: no multi-vector kernel for this shape / configuration");
  if (iters == 0) return FOS_OK;
  // workspace: nv interleaved slab sets and rr partials per workgroup
  const int nwg = p->nwg;
  if (nwg * nv > p->slab_cap) {
    if (p->slabs) (void)hipFree(p->slabs);
    p->slabs = nullptr;
    p->slab_cap = 0;
    HIP_TRY(hipMalloc(&p->slabs, (size_t)nwg * nv * p->n * sizeof(float)));
    p->slab_cap = nwg * nv;
  }
  if (nwg * nv > p->rr_cap) {
    if (p->rr_part) (void)hipFree(p->rr_part);
    if (p->rr2_part) (void)hipFree(p->rr2_part);
    p->rr_part = p->rr2_part = nullptr;
    p->rr_cap = 0;
    HIP_TRY(hipMalloc(&p->rr_part, (size_t)nwg * nv * sizeof(double)));
    HIP_TRY(hipMalloc(&p->rr2_part, (size_t)nwg * nv * sizeof(double)));
    p->rr_cap = nwg * nv;
  }
  fos::MultiY ys{};
  ys.stopped = nullptr;
  for (int v = 0; v < nv; ++v) {
    fos_fista* f = fs[v];
    int rc = flush_pending(f);
    if (rc) return rc;
    bool stopped = false;
    if ((rc = refresh_host_scalars(f, &stopped))) return rc;
    if (stopped) return fail(FOS_ERR_STATE, "fos_fista_run_multi: a handle has already stopped");
    if (!f->y_valid) {
      hipLaunchKernelGGL(fos::form_y_kernel, dim3(grid_1d(p->n, 256, 256)), dim3(256), 0, p->stream, f->x_cur, f->x_prev,
                         f->h_beta, f->ynext, p->n);
      LAUNCH_CHECK();
      f->y_valid = true;
    }
    ys.y[v] = f->ynext;
  }
  for (int v = nv; v < 4; ++v) ys.y[v] = ys.y[0];
  const size_t psz = (size_t)fs[0]->nupd * 4;
  for (int it = 0; it < iters; ++it) {
    int rc = prof_mark(p, true);
    if (rc) return rc;
    fn((const float*)p->A, p->lda, p->b, p->m, (int)p->n, ys, p->rows_per_wg, p->slabs, p->rr_part, nwg, p->stream);
    LAUNCH_CHECK();
    if ((rc = prof_mark(p, false))) return rc;
    for (int v = 0; v < nv; ++v) {
      fos_fista* f = fs[v];
      const double beta_k = f->h_beta;
      host_momentum(f->prm, f->h_k, &f->h_t, &f->h_beta);
      launch_update_from_slabs(f, f->part2 + (size_t)(f->h_k & 1) * psz, 1, beta_k, nullptr, f->ynext, f->h_beta,
                               p->slabs + (size_t)v * p->n, (int64_t)nv * p->n);
      LAUNCH_CHECK();
      f->h_k += 1;
      f->plain_count += 1;
    }
  }
  for (int v = 0; v < nv; ++v) {
    fos_fista* f = fs[v];
    f->pending = false;
    const long long last = f->h_k - 1;
    const double* cur = f->part2 + (size_t)(last & 1) * psz;
    const double* prev = f->plain_count >= 2 ? f->part2 + (size_t)((last - 1) & 1) * psz : nullptr;
    hipLaunchKernelGGL(fos::fista_finalize_plain_kernel, dim3(1), dim3(64), 0, p->stream, cur, prev, f->nupd, p->rr_part, 0,
                       f->scal, f->h_t, f->h_beta, f->h_k);
    LAUNCH_CHECK();
  }
  return FOS_OK;
}

int fos_fista_grad(fos_fista* f) {
  if (!f) return fail(FOS_ERR_ARG, "fos_fista_grad: null");
  fos_problem* p = f->p;
  int n_rr = 0, rc;
  if (f->precise && !p->resident) {
    // fp64-accumulating pass at the unrounded y_k = x_k + beta (x_k - x_{k-1}); alpha2*y is added by the consumers
    YSource ys = (plain_run(f) && f->host_valid)
                     ? YSource{nullptr, f->x_cur, f->x_prev, nullptr, &f->scal->stopped, f->h_beta, nullptr}
                     : fista_source(f);
    if ((rc = launch_pass_dd(p, ys, 0.0, nullptr, f->gbuf64))) return rc;
    hipLaunchKernelGGL(rr_from_gbuf64_kernel, dim3(1), dim3(1), 0, p->stream, f->gbuf64, (int)p->n, &f->scal->rr,
                       &f->scal->stopped);
    LAUNCH_CHECK();
    return FOS_OK;
  }
  const YSource ys = (plain_run(f) && f->host_valid && !p->col_sharded) ? plain_source(f) : fista_source(f);
  if ((rc = launch_pass(p, ys, p->b, true, &n_rr))) return rc;
  return launch_slab_reduce(p, n_rr, p->gbuf, &f->scal->rr, &f->scal->stopped);
}

int fos_fista_grad_dual(fos_fista* f) {
  if (!f) return fail(FOS_ERR_ARG, "fos_fista_grad_dual: null");
  fos_problem* p = f->p;
  int n_rr = 0, rc;
  if ((rc = flush_pending(f))) return rc;
  if (p->path == 0 && !p->colblock && p->entry->dual != nullptr && !(f->precise && !p->resident)) {
    if ((rc = launch_pass(p, fista_source(f), p->b, true, &n_rr, true))) return rc;
    if ((rc = launch_slab_reduce(p, n_rr, p->gbuf, &f->scal->rr, &f->scal->stopped))) return rc;
    hipLaunchKernelGGL(fos::fold_partials_kernel, dim3(1), dim3(fos::FOLD_THREADS), 0, p->stream, p->rr2_part, n_rr, 1,
                       &f->scal->rr_x);
    LAUNCH_CHECK();
    return p->col_sharded ? FOS_OK : reduce_across(p, &f->scal->rr_x, 1, true);
  }
  // no DUAL instantiation (fallback path / wide geometries): a separate residual pass on x_k, then the gradient
  hipLaunchKernelGGL(fos::cast_f64_f32_kernel, dim3(grid_1d(p->n, 256, 1024)), dim3(256), 0, p->stream, f->x_cur, p->ybuf,
                     p->n);
  LAUNCH_CHECK();
  YSource ys{p->ybuf, nullptr, nullptr, nullptr, &f->scal->stopped};
  if ((rc = launch_pass(p, ys, p->b, false, &n_rr))) return rc;
  hipLaunchKernelGGL(fos::fold_partials_kernel, dim3(1), dim3(fos::FOLD_THREADS), 0, p->stream, p->rr_part, n_rr, 1,
                     &f->scal->rr_x);
  LAUNCH_CHECK();
  if (!p->col_sharded && (rc = reduce_across(p, &f->scal->rr_x, 1, true))) return rc;
  return fos_fista_grad(f);
}

int fos_fista_update(fos_fista* f) {
  if (!f) return fail(FOS_ERR_ARG, "fos_fista_update: null");
  fos_problem* p = f->p;
  const bool plain = plain_run(f) && f->host_valid && !p->col_sharded;
  double* part = p->part;
  int host_beta = 0;
  double beta_k = 0.0, beta_next = 0.0;
  float* y_next = nullptr;
  if (plain) {
    // host-driven momentum (see fos_fista_run): no per-iteration bookkeeping launch, y handed on as one fp32 vector
    beta_k = f->h_beta;
    host_momentum(f->prm, f->h_k, &f->h_t, &f->h_beta);
    beta_next = f->h_beta;
    part = f->part2 + (size_t)(f->h_k & 1) * (size_t)f->nupd * 4;
    host_beta = 1;
    y_next = f->ynext;
  }
  if (p->vec4)
    hipLaunchKernelGGL((fos::fista_update_kernel<false, true>), dim3(f->nupd), dim3(256), 0, p->stream,
                       (const float*)nullptr, 0, grad_src(f), (int)p->n, f->x_cur, f->x_prev, f->scal, f->prm, part, host_beta,
                       beta_k, (double*)nullptr, y_next, beta_next);
  else
    hipLaunchKernelGGL((fos::fista_update_kernel<false, false>), dim3(f->nupd), dim3(256), 0, p->stream,
                       (const float*)nullptr, 0, grad_src(f), (int)p->n, f->x_cur, f->x_prev, f->scal, f->prm, part, host_beta,
                       beta_k, (double*)nullptr, y_next, beta_next);
  LAUNCH_CHECK();
  if (plain) {
    f->h_k += 1;
    f->plain_count += 1;
    f->y_valid = true;
    f->pending = true;
    return FOS_OK;
  }
  f->host_valid = false;
  f->y_valid = false;
  f->plain_count = 0;
  return launch_finalize(f, 0);
}

int fos_fista_trial(fos_fista* f, double t, int with_residual, double out8[8]) {
  if (!f || !out8 || !(t > 0.0)) return fail(FOS_ERR_ARG, "fos_fista_trial: bad argument");
  fos_problem* p = f->p;
  if (p->col_sharded) return fail(FOS_ERR_UNSUPPORTED, "fos_fista_trial: no column-sharded form (||A dlt||^2 needs an m-vector exchange per candidate)");
  { int rcf = flush_pending(f); if (rcf) return rcf; }
  const int grid = grid_1d(p->n, 256, 256);
  HIP_TRY(hipMemsetAsync(f->out5, 0, 8 * sizeof(double), p->stream));
  hipLaunchKernelGGL(fos::fista_trial_kernel, dim3(grid), dim3(256), 0, p->stream, grad_src(f), (int)p->n, f->x_cur,
                     f->x_prev, f->scal, f->prm, t, f->dlt, p->part);
  hipLaunchKernelGGL(fos::fold_partials_kernel, dim3(1), dim3(fos::FOLD_THREADS), 0, p->stream, p->part, grid, fos::TRIAL_W, f->out5);
  LAUNCH_CHECK();
  // rr(y_k) was produced by fos_fista_grad; copy it before the trial pass reuses the partial buffer
  HIP_TRY(hipMemcpyAsync(f->out5 + 6, &f->scal->rr, sizeof(double), hipMemcpyDeviceToDevice, p->stream));
  if (with_residual) {
    YSource ys{f->dlt, nullptr, nullptr, nullptr, nullptr};
    int n_rr = 0, rc;
    if ((rc = launch_pass(p, ys, nullptr, false, &n_rr))) return rc;        // ||A dlt||^2  (b = 0)
    hipLaunchKernelGGL(fos::fold_partials_kernel, dim3(1), dim3(fos::FOLD_THREADS), 0, p->stream, p->rr_part, n_rr, 1, f->out5 + 5);
    LAUNCH_CHECK();
    if ((rc = reduce_across(p, f->out5 + 5, 1, true))) return rc;
  }
  HIP_TRY(hipMemcpyAsync(out8, f->out5, 8 * sizeof(double), hipMemcpyDeviceToHost, p->stream));
  HIP_TRY(hipStreamSynchronize(p->stream));
  return FOS_OK;
}

// Enqueue one batch of Armijo candidates t, t*eta, ...: candidate kernel, fold of its sums -> bt_out[0..50), the matrix-
// core pass ||A dlt_j||^2 -> bt_out[64..80).  t_from_state: t is FistaScalars::tau on the device (no host value).
static int enqueue_trial_batch(fos_fista* f, double t, double eta, int nv, int t_from_state) {
  fos_problem* p = f->p;
  const int grid = grid_1d(p->n_pad, 256, 64);
  const int* stopped = t_from_state ? &f->scal->stopped : nullptr;
  if (p->dtype == FOS_BF16)
    hipLaunchKernelGGL(fos::fista_trial_batch_bf16_kernel, dim3(grid), dim3(256), 0, p->stream, grad_src(f), (int)p->n,
                       (int)p->n_pad, f->x_cur, f->x_prev, f->scal, f->prm, t, eta, nv, (unsigned short*)p->xp, p->part,
                       t_from_state);
  else
    hipLaunchKernelGGL(fos::fista_trial_batch_kernel, dim3(grid), dim3(256), 0, p->stream, grad_src(f), (int)p->n,
                       (int)p->n_pad, f->x_cur, f->x_prev, f->scal, f->prm, t, eta, nv, p->xp, p->part, t_from_state);
  hipLaunchKernelGGL(fos::fold_partials_kernel, dim3(1), dim3(fos::FOLD_THREADS), 0, p->stream, p->part, grid, fos::BT_W, p->bt_out);
  LAUNCH_CHECK();
  if (p->col_sharded) {              // grad.dlt_j, ||dlt_j||^2, the counts, ||grad||^2, ||y||^2 are sums over the column blocks
    int rc = reduce_across(p, p->bt_out, fos::BT_W, true);
    if (rc) return rc;
  }
  return launch_residual_batch(p, 0, p->bt_out + 64, stopped);
}

// Device-driven iterations with data-dependent control - Armijo search, adaptive restart, the stopping rules - and,
// optionally, the history recorded on the device.  One body behind fos_fista_run_backtracking and fos_fista_run_recorded.
static int run_device_driven(fos_fista* f, int iters, bool backtracking, double eta, double armijo_c, double grad_eps,
                             int32_t* ls_iters, double* tau_hist, double* x_hist, double* hist, double* rr_seen) {
  fos_problem* p = f->p;
  int rc = flush_pending(f);
  if (rc) return rc;
  if (backtracking) {
    if ((rc = ensure_batch_workspace(p))) return rc;
    // the step lives on the device from here on (tau persists, :197)
    if (!f->tau_on_device) {
      hipLaunchKernelGGL(fos::set_state_tau_kernel, dim3(1), dim3(1), 0, p->stream, f->scal, f->prm.tau);
      LAUNCH_CHECK();
      f->tau_on_device = true;
    }
  }
  f->host_valid = false;                       // t_k, beta_k depend on nothing the host knows any more
  f->y_valid = false;
  f->plain_count = 0;
  fos::FistaParams prm_dev = f->prm;
  prm_dev.tau_from_state = backtracking ? 1 : 0;
  const bool record = hist != nullptr;
  for (int it = 0; it < iters; ++it) {
    // gradient (:173-175; the fp64 pass in precise mode); recording: the same pass (or a residual pass of its own where
    // there is no DUAL kernel) also yields ||A x_k - b||^2 of the iterate this iteration starts from
    if (record && rr_seen != nullptr) {
      if ((rc = fos_fista_grad_dual(f))) return rc;
      hipLaunchKernelGGL(fos::record_rr_x_kernel, dim3(1), dim3(1), 0, p->stream, f->scal, rr_seen + it);
      LAUNCH_CHECK();
    } else if ((rc = fos_fista_grad(f))) {
      return rc;
    }
    if (f->prm.tol_grad > 0.0 && (rc = launch_grad_norm_stop(f))) return rc;   // :179
    if (backtracking) {
      if ((rc = enqueue_trial_batch(f, 0.0, eta, fos::BT_NV, 1))) return rc;   // :187-191 for 16 candidates
      hipLaunchKernelGGL(fos::armijo_decide_kernel, dim3(1), dim3(1), 0, p->stream, p->bt_out, f->scal, f->prm, eta,
                         armijo_c, grad_eps, fos::BT_NV, ls_iters, tau_hist, (long long)it);
      LAUNCH_CHECK();
    }
    // update (with the step the decision left in FistaScalars::tau), then the scalar bookkeeping / history row
    double* xrow = x_hist ? x_hist + (size_t)it * p->n : nullptr;
    if (p->vec4)
      hipLaunchKernelGGL((fos::fista_update_kernel<false, true>), dim3(f->nupd), dim3(256), 0, p->stream,
                         (const float*)nullptr, 0, grad_src(f), (int)p->n, f->x_cur, f->x_prev, f->scal, prm_dev, p->part, 0,
                         0.0, xrow, (float*)nullptr, 0.0);
    else
      hipLaunchKernelGGL((fos::fista_update_kernel<false, false>), dim3(f->nupd), dim3(256), 0, p->stream,
                         (const float*)nullptr, 0, grad_src(f), (int)p->n, f->x_cur, f->x_prev, f->scal, prm_dev, p->part, 0,
                         0.0, xrow, (float*)nullptr, 0.0);
    LAUNCH_CHECK();
    if ((rc = launch_finalize(f, 0, record ? hist + (size_t)it * 4 : nullptr))) return rc;
  }
  return FOS_OK;
}

int fos_fista_run_backtracking(fos_fista* f, int iters, double eta, double armijo_c, double grad_eps, int32_t* ls_iters,
                               double* tau_hist) {
  if (!f || iters < 0 || !(eta > 0.0 && eta < 1.0) || !(grad_eps >= 0.0))
    return fail(FOS_ERR_ARG, "fos_fista_run_backtracking: bad argument");
  fos_problem* p = f->p;
  if (!batch_supported(p) || p->resident)
    return fail(FOS_ERR_UNSUPPORTED, "fos_fista_run_backtracking: needs the matrix-core candidate pass (streaming plans)");
  if (iters == 0) return FOS_OK;
  return run_device_driven(f, iters, true, eta, armijo_c, grad_eps, ls_iters, tau_hist, nullptr, nullptr, nullptr);
}

int fos_fista_run_recorded(fos_fista* f, int iters, int backtracking, double eta, double armijo_c, double grad_eps,
                           double* x_hist, double* hist, double* rr_seen, int32_t* ls_iters, double* tau_hist) {
  if (!f || iters < 0 || (iters > 0 && (!x_hist || !hist)) ||
      (backtracking && (!(eta > 0.0 && eta < 1.0) || !(grad_eps >= 0.0))))
    return fail(FOS_ERR_ARG, "fos_fista_run_recorded: bad argument");
  fos_problem* p = f->p;
  if (p->resident || (backtracking && !batch_supported(p)))
    return fail(FOS_ERR_UNSUPPORTED, "fos_fista_run_recorded: resident problems record inside their one launch; "
                                     "backtracking needs the matrix-core candidate pass");
  if (iters == 0) return FOS_OK;
  return run_device_driven(f, iters, backtracking != 0, eta, armijo_c, grad_eps, ls_iters, tau_hist, x_hist, hist, rr_seen);
}

int fos_fista_resume_after_stall(fos_fista* f, double* tau_out) {
  if (!f || !tau_out) return fail(FOS_ERR_ARG, "fos_fista_resume_after_stall: null");
  fos_problem* p = f->p;
  fos::FistaScalars h;
  HIP_TRY(hipMemcpyAsync(&h, f->scal, sizeof(h), hipMemcpyDeviceToHost, p->stream));
  HIP_TRY(hipStreamSynchronize(p->stream));
  *tau_out = h.tau;
  f->prm.tau = h.tau;                          // tau persists (:197), also across the hand-over to the host
  hipLaunchKernelGGL(fos::clear_stall_kernel, dim3(1), dim3(1), 0, p->stream, f->scal);
  LAUNCH_CHECK();
  return FOS_OK;
}

int fos_fista_trial_batch(fos_fista* f, double t, double eta, int nv, double* out) {
  if (!f || !out || !(t > 0.0) || !(eta > 0.0) || nv < 1 || nv > fos::BT_NV)
    return fail(FOS_ERR_ARG, "fos_fista_trial_batch: bad argument");
  fos_problem* p = f->p;
  { int rcf = flush_pending(f); if (rcf) return rcf; }
  if (!batch_supported(p)) return fail(FOS_ERR_UNSUPPORTED, "fos_fista_trial_batch: needs the fused path");
  int rc = ensure_batch_workspace(p);
  if (rc) return rc;
  if ((rc = enqueue_trial_batch(f, t, eta, nv, 0))) return rc;
  HIP_TRY(hipMemcpyAsync(p->bt_out + 100, &f->scal->rr, sizeof(double), hipMemcpyDeviceToDevice, p->stream));
  double h[128];
  HIP_TRY(hipMemcpyAsync(h, p->bt_out, 128 * sizeof(double), hipMemcpyDeviceToHost, p->stream));
  HIP_TRY(hipStreamSynchronize(p->stream));
  for (int j = 0; j < nv; ++j) {
    double* o = out + 8 * j;
    o[0] = h[j];                        // grad . dlt_j
    o[1] = h[fos::BT_NV + j];           // ||dlt_j||^2
    o[2] = h[2 * fos::BT_NV + j];       // #(dlt_j != 0)
    o[3] = h[3 * fos::BT_NV];           // ||grad||^2
    o[4] = h[3 * fos::BT_NV + 1];       // ||y||^2
    o[5] = h[64 + j];                   // ||A dlt_j||^2
    o[6] = h[100];                      // ||A y - b||^2
    o[7] = 0.0;
  }
  return FOS_OK;
}

int fos_fista_status_get(fos_fista* f, fos_fista_status* out) {
  if (!f || !out) return fail(FOS_ERR_ARG, "fos_fista_status_get: null");
  { int rcf = flush_pending(f); if (rcf) return rcf; }
  fos::FistaScalars h;
  HIP_TRY(hipMemcpyAsync(&h, f->scal, sizeof(h), hipMemcpyDeviceToHost, f->p->stream));
  HIP_TRY(hipStreamSynchronize(f->p->stream));
  out->t_prev = h.t_prev; out->beta = h.beta; out->this_step = h.this_step; out->prev_step = h.prev_step;
  out->ratio = h.ratio; out->rr = h.rr; out->gnorm2 = h.gnorm2; out->xnorm1 = h.xnorm1; out->xnorm2 = h.xnorm2;
  out->rr_x = h.rr_x;
  out->tau = h.tau;
  out->k = h.k; out->stopped = h.stopped; out->restarts = h.restarts;
  return FOS_OK;
}

int fos_fista_get_x(fos_fista* f, double* dst) {
  if (!f || !dst) return fail(FOS_ERR_ARG, "fos_fista_get_x: null");
  HIP_TRY(hipMemcpyAsync(dst, f->x_cur, (size_t)f->p->n * sizeof(double), hipMemcpyDeviceToDevice, f->p->stream));
  return FOS_OK;
}
double* fos_fista_x(fos_fista* f) { return f ? f->x_cur : nullptr; }
float* fos_fista_gbuf(fos_fista* f) { return f ? f->p->gbuf : nullptr; }

}  // extern "C"
