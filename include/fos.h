/* fos.h — C ABI of the MI355X-native FISTA / L-BFGS inner loop (libfos_hip.so).
 *
 * The reference (ElBaldo1/FastOptSolver) is pure Python/NumPy and has NO FFI layer; the "boundary" it
 * exposes is a set of Python callables.  Each entry point below names the reference expression it
 * replaces (file:line in the reference checkout); INTEGRATION.md shows the ctypes binding a maintainer
 * would add to the reference to route its loops through this library.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error (FOS_ERR_*); text via fos_last_error().
 *   - no C++ exception crosses the boundary; no torch types appear in any signature.
 *   - all data pointers are DEVICE pointers borrowed from the caller (e.g. tensor.data_ptr()); they must
 *     outlive the handle that stores them.  `stream` is a hipStream_t passed as void*.
 *   - calls only ENQUEUE work on the handle's stream unless documented "synchronises".
 *   - a handle is single-threaded; distinct handles may be used from distinct threads.
 *   - A is row-major (C order), m rows, n columns, leading dimension lda (elements).
 */
#ifndef FOS_H_
#define FOS_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FOS_ABI_VERSION 3

enum { FOS_OK = 0, FOS_ERR_ARG = -1, FOS_ERR_HIP = -2, FOS_ERR_STATE = -3, FOS_ERR_UNSUPPORTED = -4 };
enum { FOS_F32 = 0, FOS_BF16 = 1 };                    /* element type of A */
enum { FOS_MODE_FISTA = 0, FOS_MODE_DELTA = 1, FOS_MODE_ISTA = 2 };
enum { FOS_PROX_L1 = 0, FOS_PROX_ENET = 1 };
enum { FOS_STOP_NONE = 0, FOS_STOP_STEP = 1, FOS_STOP_RATIO = 2, FOS_STOP_GRAD = 3, FOS_STOP_LS_STALL = 4 };

enum { FOS_PLAN_NO_RESIDENT = 1, FOS_PLAN_NO_TALL = 2, FOS_PLAN_NO_WIDE = 4, FOS_PLAN_NO_COLBLOCK = 8,
       FOS_PLAN_CLUSTER = 16,     /* fos_problem_replan; FOS_PLAN_CLUSTER OPTS IN to the one-read cluster form of the
                                     multi-weight matrix-core pass wherever it is served (csrc/cluster_pass.hpp:
                                     cooperative launch, fp32, 2049..16384 columns) */
       FOS_PLAN_INTERLEAVE = 32,  /* rows of the streaming pass dealt round-robin to the workgroups (all CUs read one
                                     contiguous window) instead of one contiguous block per workgroup ...           */
       FOS_PLAN_NO_INTERLEAVE = 64, /* ... or never; neither bit: the planner's default for the shape */
       FOS_PLAN_NO_CLUSTER = 128,   /* never the one-read cluster form (neither cluster bit: the planner takes it where it
                                       measured ahead - fp32, exactly 4096 or 8192 columns, >= 2 GiB, unsharded) */
       FOS_PLAN_FUSED_MFMA = 256,   /* OPT IN: plain fos_fista_run calls take the one-launch persistent step of
                                       fos_fista_run_fused (LDS-staged panels, row dots on the matrix cores, resident
                                       iterate) wherever the shape is served */
       FOS_PLAN_CHIP_RESIDENT = 512, /* fos_fista_run calls without backtracking on tall-skinny problems (n <= 16) take fos_fista_run_chip
                                       (A resident in the LDS of up to all CUs, one grid barrier per iteration) WHEREVER
                                       it is served ...                                                              */
       FOS_PLAN_NO_CHIP_RESIDENT = 1024 /* ... or never; neither bit: the planner takes it where it measured at least 1.4x
                                       ahead - fp32, 512 <= m <= 131072 rows at n <= 8 (32768 rows at n <= 16), calls of 8 iterations or more.
                                       Problems that fit ONE CU's LDS keep the single-workgroup resident loop. */ };

typedef struct fos_problem fos_problem;   /* A, b, launch plan, workspace            */
typedef struct fos_comm fos_comm;         /* communicator of a row-sharded problem   */
typedef struct fos_fista fos_fista;       /* iterate + momentum state of one solve   */

/* Solver parameters: the keyword arguments of fista() iterative_solvers.py:132-147,
 * fista_delta() :251-265 and the prox closure of ista() :65-77. */
typedef struct fos_fista_params {
  double tau;               /* step t_init_factor / L                     :158, :276, :81        */
  double alpha1;            /* l1 weight                                   :201                   */
  double alpha2;            /* l2 weight                                   :174-175 or prox :15   */
  double delta;             /* FISTA-delta parameter                       :330                   */
  double restart_threshold; /*                                             :210                   */
  double tol_step;          /* stop when ||x_next - x_k|| < tol_step       :238, :337  (0 = off)  */
  double tol_ratio;         /* stop when ratio < tol_ratio                 :242, :341  (0 = off)  */
  double tol_grad;          /* stop BEFORE the update when ||grad|| < tol_grad   :179   (0 = off)  */
  int32_t mode;             /* FOS_MODE_*                                                         */
  int32_t prox_kind;        /* FOS_PROX_L1: l2 in the gradient; FOS_PROX_ENET: l2 in the prox     */
  int32_t adaptive_restart; /*                                             :209                   */
  int32_t reserved;
} fos_fista_params;

/* Host copy of the device-resident loop state (fos_fista_status synchronises). */
typedef struct fos_fista_status {
  double t_prev, beta, this_step, prev_step, ratio;
  double rr;        /* ||A y - b||^2 of the last gradient   */
  double gnorm2;    /* ||grad_smooth(y)||^2 of the last update */
  double xnorm1;    /* ||x_k||_1   */
  double xnorm2;    /* ||x_k||_2^2 */
  double rr_x;      /* ||A x_k - b||^2 at the iterate the last fos_fista_grad_dual started from */
  double tau;       /* device-driven backtracking: the step after the last search (fos_fista_run_backtracking) */
  int64_t k;        /* completed iterations */
  int32_t stopped;  /* FOS_STOP_* */
  int32_t restarts;
} fos_fista_status;

const char* fos_last_error(void);
int fos_abi_version(void);

/* ---- problem ------------------------------------------------------------------------------------ */
/* Bind A (m x n, a_dtype) and b (m floats, may be NULL = zero vector, as in estimate_lipschitz
 * iterative_solvers.py:54).  Allocates the partial-gradient slabs.  Synchronises (allocation). */
int fos_problem_create(fos_problem** out, const void* A, int64_t m, int64_t n, int64_t lda, int a_dtype,
                       const float* b, void* stream);
int fos_problem_destroy(fos_problem* p);
/* Move the handle (and the fos_fista handles on it) to another stream; the new stream first waits for whatever the old
 * one still has enqueued on the handle's workspace.  The Python layer calls this whenever torch's current stream differs
 * from the one the handle last ran on, so a prepared problem can be used inside `with torch.cuda.stream(s)`. */
int fos_problem_set_stream(fos_problem* p, void* stream);
/* plan[0..7] = {path (0 fused single pass, 1 two-pass fallback), threads, chunks/thread, rows/step,
 *               workgroups, slabs, flags (bit 0: non-temporal loads; bit 1: small enough for the single-launch
 *               LDS-resident loop that fos_fista_run / _run_history / fos_power_iter then use; bit 2: n <= 64 (or aligned rows of up to 128 columns), the
 *               single pass is the row-per-thread kernel, which has no alignment requirements; bit 3: rows wider than
 *               any single-pass kernel - column blocks through the streaming kernel in two phases, A read twice;
 *               bit 4: set once fos_fista_run_multi has planned the one-read cluster form of the matrix-core pass;
 *               bit 5: the streaming pass deals its rows round-robin to the workgroups; bit 6: FOS_PLAN_FUSED_MFMA; bit 7: the chip-resident loop is forced on), CUs} */
int fos_problem_plan(const fos_problem* p, int32_t plan[8]);
/* Re-run the planner with kernel families switched off (FOS_PLAN_* bits): NO_RESIDENT keeps small problems off the
 * one-launch LDS-resident loop, NO_TALL keeps n <= 64 off the row-per-thread pass, NO_WIDE keeps 16384 < n <= 32768 off
 * the y-in-LDS pass, NO_COLBLOCK keeps wider rows off the column-blocked streaming passes; each then takes the next
 * family that fits (streaming single pass, column blocks, two-pass).  For tests and
 * A/B measurements; call before creating fos_fista handles on the problem.  Synchronises (reallocates workspace). */
int fos_problem_replan(fos_problem* p, unsigned flags);
/* Benchmark/tuning override of the fused-kernel geometry; returns FOS_ERR_UNSUPPORTED if not instantiated. */
int fos_problem_tune(fos_problem* p, int threads, int chunks, int rows, int workgroups);
/* The same for the streaming fp64-accumulating pass (fos_gemv_pair_dd): its workgroup count (0 = back to the planner's);
 * takes effect with the next fp64 pass.  Synchronises.  Not for the tall / resident plans (they share the fp32 pass's grid). */
int fos_problem_tune_dd(fos_problem* p, int workgroups);
/* Use a caller-owned gradient buffer (n + 4 floats, 16-byte aligned) instead of the internal one, e.g. a
 * torch tensor that torch.distributed all-reduces between fos_fista_grad and fos_fista_update. */
int fos_problem_set_gbuf(fos_problem* p, float* gbuf);

/* ---- row-sharded problems (the reference is single-process) --------------------------------------------
 * One process per GPU; each rank binds ITS rows of A and b to a fos_problem and attaches a communicator.  From then on
 * every result of a pass over A that is a sum over rows - fos_gemv_pair* (gradient and ||r||^2), fos_residual_objective
 * and fos_residual_batch (||A x - b||^2), the Armijo trials, the history residual of fos_fista_grad_dual, the power
 * iteration - is summed over the ranks on the handle's stream before anything consumes it: ONE all-reduce of n + 1
 * floats per FISTA iteration (n + 1 doubles per L-BFGS fg), alpha2*y added after the reduction.  x_k, x_{k-1} and the
 * momentum scalars are replicated; every rank computes the identical update from identical numbers.  fos_fista_run
 * stays enqueue-only.  All ranks must issue the same calls in the same order.  After a device-side stop or a parked
 * search the remaining enqueued iterations are no-ops on the iterate state; their all-reduces still run (a collective
 * cannot be skipped by one rank), on partials that are RE-DERIVED from the unchanged workspace each time, so the
 * gradient buffer keeps the last active iteration's sums (it never compounds).
 * Transport: RCCL over xGMI, resolved with dlopen at first use (fos_comm_transport() says which library), or the
 * one-shot full-mesh kernel below.
 *   fos_comm_unique_id   rank 0 creates the 128-byte id; the caller broadcasts it out of band (e.g. torch.distributed)
 *   fos_comm_create      collective over all ranks (ncclCommInitRank) on the CURRENT device
 *   fos_problem_set_comm attach (NULL detaches); the problem then never uses the one-workgroup resident loop. */
int fos_comm_unique_id(char id[128]);
int fos_comm_create(fos_comm** out, const char id[128], int nranks, int rank);
/* Second transport: a one-shot all-reduce kernel over the full xGMI mesh - every rank writes its
 * vector into an inbox on every peer (IPC-mapped device memory) and sums the P inbox rows in rank order: one launch, one
 * link latency, results bit-identical on all ranks.  For the latency-bound n + 1 element exchange; messages up to
 * cap_bytes.  fos_comm_mesh_create allocates this rank's inbox and returns 128 bytes of IPC handles; the caller gathers
 * the handles of all ranks (rank order, 128 bytes each) and passes them to fos_comm_mesh_connect.  A peer that does not
 * deliver within 20 s raises a device flag instead of hanging the kernel: fos_comm_check (synchronises) reports it. */
int fos_comm_mesh_create(fos_comm** out, int nranks, int rank, int64_t cap_bytes, char handles[128]);
int fos_comm_mesh_connect(fos_comm* c, const char* all_handles);
int fos_comm_check(fos_comm* c, void* stream);
int fos_comm_destroy(fos_comm* c);
int fos_comm_info(const fos_comm* c, int* nranks, int* rank);
/* Mesh communicators: whether inbox / flags are fine-grained (inter-device coherent) allocations, and the inbox row size. */
int fos_comm_mesh_info(const fos_comm* c, int* fine_grained, int64_t* cap_bytes);
const char* fos_comm_transport(void);
/* In-place sum over the ranks of `count` floats (is_f64 = 0) or doubles (1) on `stream`; enqueues only. */
int fos_comm_allreduce(fos_comm* c, void* buf, int64_t count, int is_f64, void* stream);
int fos_problem_set_comm(fos_problem* p, fos_comm* c);
/* COLUMN sharding (very wide A: SURVEY.md 8f rank 4): every rank binds A[:, its columns] (all m rows, n_p columns) and
 * the whole b; the iterate is PARTITIONED - a rank owns and updates only its block of x, nothing is replicated.  The
 * pass becomes two phases around ONE exchange of an m-vector: r = sum_p A_p y_p - b (all-reduce of m floats), then
 * g_p = A_p^T r locally; prox and momentum are separable, so the update is local; the four scalar sums of an iteration
 * (step norms, ||grad||^2, ||x||_1, ||x||^2) are all-reduced (4 doubles) so that restart and the stopping rules decide on
 * global numbers.  A_p is read twice per iteration (before and after the exchange).  fos_fista_run / _grad / _update,
 * fos_gemv_pair, fos_residual_objective, the batched Armijo trials (fos_fista_run_backtracking), the fp64 pass
 * (fos_gemv_pair_dd), fos_lbfgs_direction_cols and fos_fista_run_multi work on such a problem; the single-candidate trial
 * (fos_fista_trial), fos_lbfgs_minimize, fos_problem_replan and the one-launch power iteration do not
 * (FOS_ERR_UNSUPPORTED).  Needs the streaming layout (n_p > 64, 16-byte rows). */
int fos_problem_set_comm_cols(fos_problem* p, fos_comm* c);

/* Kernel timing for roofline reports: while enabled (enable = N > 0), every N-th launch of the single-pass kernel
 * (or of the two fallback kernels, or of the batched MFMA pass) is bracketed by a pair of hipEvents recorded on the
 * handle's stream; N > 1 keeps the markers from perturbing the run being measured.
 * fos_problem_profile_read synchronises, returns the accumulated device milliseconds and the number of
 * bracketed launches since the last read, and resets both. */
int fos_problem_profile(fos_problem* p, int enable);
int fos_problem_profile_read(fos_problem* p, double* ms_total, int64_t* launches);

/* Measurement aid (bench.py, SURVEY.md 8d "the box's measured stream-read bandwidth"): `launches` read-only passes over
 * `bytes` of device memory (16-byte aligned, a multiple of 16) by a kernel that only loads and sums; returns GB/s (1e9) and
 * microseconds per pass (us_out may be NULL).  Synchronises.  Not part of the solver path. */
int fos_stream_read_probe(const void* buf, size_t bytes, int launches, void* stream, double* gbps_out, double* us_out);

/* K2: grad = A^T (A y - b) + alpha2*y ; *rr_out (device double, may be NULL) = ||A y - b||^2.
 * Replaces iterative_solvers.py:54, :173-175, :292-294 and lbfgs.py:46-51.  A is read ONCE. */
int fos_gemv_pair(fos_problem* p, const float* y, float alpha2, float* grad, double* rr_out);

/* LEGACY (round-1 generation; no longer called by the Python layer, kept so that the ABI only grows): the same with the
 * iterate y given in fp64 (n doubles), rounded once to fp32 for the pass over A on the fused path.  The L-BFGS `fg` is
 * fos_gemv_pair_dd (fp64 accumulation, y never rounded). */
int fos_gemv_pair_f64(fos_problem* p, const double* y, double alpha2, float* grad, double* rr_out);

/* The L-BFGS `fg` (lbfgs.py:43-54) at the precision SciPy's optimiser runs it in (lbfgs.py:64): x is n doubles and is
 * never rounded, every product and every sum of the pass is fp64 (A and b are what the handle stores: fp32 / bf16),
 * grad_rr[0..n) = A^T (A x - b) + alpha2*x, grad_rr[n] = ||A x - b||^2.  Still ONE read of A (the fp64-accumulating
 * instantiation of the single-pass kernel; y in LDS for 64 KiB rows); ragged / misaligned layouts and rows wider than
 * 16384 columns run fp64 two-pass kernels.  A sharded run all-reduces the n+1 doubles. */
int fos_gemv_pair_dd(fos_problem* p, const double* x, double alpha2, double* grad_rr);

/* K5: out3 (device doubles) = { ||A x - b||^2, ||x||_2^2, ||x||_1 } — one pass over A.
 * Replaces g_smooth iterative_solvers.py:163-168 and compute_objective objective_functions.py:13-24. */
int fos_residual_objective(fos_problem* p, const float* x, double* out3);

/* Batched K5 on the matrix cores: out16[j] (device doubles) = ||A X_j - use_b*b||^2 for the nv <= 16 vectors stored as
 * the columns of X (n x 16 floats, row-major: X[k*16 + j]); one pass over A.  fp32 A: v_mfma_f32_16x16x4_f32;
 * bf16 A: v_mfma_f32_16x16x32_bf16 with each vector split into three bf16 terms (24 mantissa bits), so both keep
 * fp32 accuracy.  FOS_ERR_UNSUPPORTED on fallback-path problems. */
int fos_residual_batch(fos_problem* p, const float* X, int nv, int use_b, double* out16);

/* Power iteration, iterative_solvers.py:45-60.  v_inout: start vector (n floats, need not be normalised),
 * overwritten with the last iterate.  Any n_iter >= 1.  Synchronises; *L_out and *iters_out are host values. */
int fos_power_iter(fos_problem* p, float* v_inout, int n_iter, double tol, double* L_out, int* iters_out);

/* ---- stand-alone prox (K3), prox_operators.py:3-8 and :10-16 --------------------------------------- */
int fos_prox_l1(const float* v, float thr, float* out, int64_t n, void* stream);
int fos_prox_l1_vec(const float* v, const float* thr, float* out, int64_t n, void* stream);   /* per-element threshold */
int fos_prox_elastic_net(const float* v, float tau, float alpha1, float alpha2, float* out, int64_t n, void* stream);
int fos_prox_elastic_net_vec(const float* v, const float* tau, float alpha1, float alpha2, float* out, int64_t n,
                             void* stream);                                                  /* per-element tau */

/* ---- FISTA / FISTA-delta / ISTA state machine ------------------------------------------------------- */
int fos_fista_create(fos_problem* p, fos_fista** out);
int fos_fista_destroy(fos_fista* f);
/* x_0 = x0 (device, n DOUBLES) or zeros when NULL; t = 1; beta = 0; k = 0.  :149-161, :269-280, :79-81.  Enqueues only.
 * The iterate state x_k, x_{k-1} is fp64 on the device (n-vectors are negligible traffic; an fp32 state alone
 * costs 1e-4 of parity on ill-conditioned data).  y_k is rounded once to fp32 for the pass over A. */
int fos_fista_reset(fos_fista* f, const fos_fista_params* prm, const double* x0);
int fos_fista_set_tau(fos_fista* f, double tau);
/* Precise mode for the split form (fos_fista_grad / _trial / _trial_batch / _update): the gradient comes from the
 * fp64-accumulating pass at the UNROUNDED y_k (the kernel of fos_gemv_pair_dd), so the terms of the Armijo comparison
 * :191 - above all the cancelling sum grad.(x_tmp - y) - are fp64-accurate and the search takes the reference's decisions
 * wherever those are decidable at all.  The pass costs 5-25 % more than the fp32 one; fista(backtracking=True) turns it
 * on.  Plain fos_fista_run is unaffected. */
int fos_fista_set_precise(fos_fista* f, int on);
/* Precise mode with a CALLER-OWNED gradient buffer (n + 4 doubles, 16-byte aligned; NULL returns to fp32): split-form
 * sharding sums [gradient ; ||r||^2] (n + 1 doubles) over the ranks itself, between fos_fista_grad and the consumers. */
int fos_fista_set_gbuf64(fos_fista* f, double* buf);
/* Enqueue `iters` full iterations (gradient, prox, momentum, restart and stop logic all on the device;
 * no host round trip).  Iterations after a device-side stop are no-ops.  :170-242, :289-342 */
int fos_fista_run(fos_fista* f, int iters);
/* Plain run that also records the history on the device (return_history=True of fista :224-232 / fista_delta
 * :319-322 without a host round trip per iteration and without the reference's extra pass over A per iteration):
 *   x_hist   iters x n doubles: row i = iterate after iteration i
 *   hist     iters x 4 doubles: { ||A x - b||^2, ||x||_1, ||x||_2^2, ||x - x_before||^2 } of that iterate
 *   work     scratch of fos_fista_history_workspace(f, iters) bytes (caller-owned device memory)
 * Only for runs without data-dependent control (no adaptive restart, no stopping tolerance) on the fused path with a
 * DUAL kernel; otherwise FOS_ERR_UNSUPPORTED and the caller drives the split form below.  Enqueues only. */
int64_t fos_fista_history_workspace(fos_fista* f, int iters);
/* Small problems (fos_problem_plan flags bit 1: A fits one CU's LDS, n <= 64): up to `iters` iterations with EVERY
 * option of the reference's loops inside ONE launch of ONE workgroup - backtracking (:183-197, the reference's own
 * comparison g(x_tmp) <= g(y) + C*grad.(x_tmp - y) in fp64; `armijo_c` = its module global C), adaptive restart and
 * the step / ratio stops of fos_fista_params, the gradient-norm stop (grad_tol > 0, :179), history.  A, b and the
 * iterate state stay in LDS for the whole run.  Optional device outputs (NULL to skip): x_hist iters x n doubles,
 * hist iters x 4 doubles (layout of fos_fista_run_history), ls_iters iters int32 (shrinks per search), tau_hist iters
 * doubles (step used).  Synchronises; *iters_done = iterations completed (a stop ends the run early), *tau_out = the
 * step after the last search, which the handle keeps.  FOS_ERR_UNSUPPORTED when the problem does not fit. */
int fos_fista_run_resident(fos_fista* f, int iters, int backtracking, double eta, double armijo_c, double grad_tol,
                           double* x_hist, double* hist, int32_t* ls_iters, double* tau_hist, int32_t* iters_done,
                           double* tau_out);
int fos_fista_run_history(fos_fista* f, int iters, double* x_hist, double* hist, void* work);
/* Regularisation path: nv <= 16 state machines bound to the SAME fos_problem (different alpha1 / alpha2 / tau) advance
 * `iters` plain iterations in lockstep.  nv <= 4, fp32, n <= 8192: the multi-vector form of the single-pass kernel, A read
 * from HBM once per iteration for all of them.  Otherwise (up to 16 weights, fp32 and bf16 storage, any streaming shape):
 * two GEMM-shaped products per iteration on the matrix cores for all weights together, R = A Y - b (v_mfma_f32_16x16x4_f32
 * / v_mfma_f32_16x16x32_bf16) and G = A^T R (csrc/gram_batch.hpp).  Results equal running each handle with
 * fos_fista_run (1e-6).  A row-sharded problem (fos_problem_set_comm) takes the matrix-core pass for any number of
 * weights: the 16 partial gradients are summed over the ranks in ONE all-reduce per iteration.
 * Handles with data-dependent control (adaptive restart, step / ratio tolerances: iterative_solvers.py:209-221, :235-242)
 * keep the lockstep from three weights on: momentum and stops are decided on the device per state machine every
 * iteration (update -> bookkeeping of all weights in one launch -> their y_{k+1}), a stopped weight becomes a masked
 * column of the block.  FOS_ERR_UNSUPPORTED when the shape has no such kernel (two-pass path, n <= 64, n > 16384 fp32) or
 * a handle uses the gradient-norm rule (tol_grad: it sits before the update) or a device-held step (backtracking): the
 * caller then runs the handles one by one.
 * A COLUMN-sharded problem (fos_problem_set_comm_cols) takes the matrix-core pass for any number of weights and any block
 * width, always device-controlled: per row panel ONE all-reduce of the panel's 16 residual columns between the two
 * products (rows x 16 floats; on the mesh transport the panel is sized to the communicator's inbox rows, >= 16 KiB), the
 * gradient blocks stay local, 4 doubles per weight of step norms cross per iteration.
 * SURVEY.md 8(f) rank 3. */
int fos_fista_run_multi(fos_fista* const* fs, int nv, int iters);
/* BASELINE north_star's literal step, opt-in: `iters` plain iterations in ONE persistent launch - A staged through LDS
 * in 4-row panels, the row dots A y on v_mfma_f32_4x4x1_16B_f32, A^T r on the VALU from the staged tile, a grid-wide
 * barrier, then prox + momentum by the workgroup that OWNS the columns, whose slice of x_k, x_{k-1} stays in its LDS for
 * the whole run (csrc/fused_step.hpp; iterative_solvers.py:170-242, :289-342).  Same iterates as fos_fista_run to 1e-6.
 * fp32 A with 2048 / 4096 / 6144 / 8192 columns, aligned rows, m >= 8 x CUs, unsharded, plain runs (FOS_ERR_UNSUPPORTED
 * otherwise).  Synchronises at the end (reports a timed-out grid-wide wait as FOS_ERR_STATE).  The default two-launch step
 * measures faster: this entry point exists so that the comparison is a measurement. */
int fos_fista_run_fused(fos_fista* f, int iters);
/* Tall-skinny problems beyond one CU's LDS (csrc/chip_resident.hpp): `iters` iterations in ONE launch with A resident in the
 * LDS of up to all CUs and one grid-wide barrier per iteration - every workgroup reads all partial gradients (n <= 16) and
 * applies the identical fp64 update to its copy of the iterate; adaptive restart and the step / ratio stops are decided the
 * same way, by every workgroup alike.  Not served (FOS_ERR_UNSUPPORTED): the gradient-norm rule, backtracking, the fp64 split
 * gradient, sharded problems.  fp32 A, n <= 16, 512 <= m <= (150 KiB / row bytes) x #CUs.  Same state hand-over as
 * fos_fista_run, which takes this loop by itself in the planner's region (FOS_PLAN_CHIP_RESIDENT above).  Every grid-wide
 * wait is bounded (2 s); one that runs out ends the launch before anything is written back: FOS_ERR_STATE, the handle is as
 * before the call.  Synchronises at the end. */
int fos_fista_run_chip(fos_fista* f, int iters);
/* Measurement hook of the persistent step: stamps = device buffer of #CUs x 8 uint64 (NULL = off); every workgroup then
 * records the 100 MHz wall clock of the LAST iteration of a fos_fista_run_fused call - [0] phase A starts, [1] phase A
 * ends, [2] behind the first grid barrier, [3] phase B ends, [4] behind the second grid barrier.  The buffer is borrowed. */
int fos_problem_set_fused_stamps(fos_problem* p, unsigned long long* stamps);
/* Split form for host-driven control (grad-norm stop :179, backtracking :183-197, sharded runs):
 *   fos_fista_grad    gbuf[0..n) = A^T (A y_k - b) (WITHOUT alpha2*y), gbuf[n] = ||A y_k - b||^2 (float)
 *   fos_fista_update  prox + momentum from gbuf (after an optional all-reduce of gbuf[0..n]) */
int fos_fista_grad(fos_fista* f);
/* fos_fista_grad that ALSO returns ||A x_k - b||^2 (status.rr_x) from the same pass over A: the history objective
 * f(x_k) (:225-230, :321) without the extra pass per iteration the reference pays. */
int fos_fista_grad_dual(fos_fista* f);
int fos_fista_update(fos_fista* f);
/* Armijo trial at step t (:187-191) in cancellation-free form.  With x_tmp = prox(y_k - t*grad) and
 * dlt = x_tmp - y_k the reference's test g(x_tmp) <= g(y_k) + C*grad.dlt is, exactly (g is quadratic),
 *     (1 - C)*grad.dlt + 0.5*||A dlt||^2 + 0.5*alpha2*||dlt||^2 <= 0.
 * Synchronises.  out8 (host) = { grad.dlt, ||dlt||^2, #(dlt != 0), ||grad||^2, ||y_k||^2, ||A dlt||^2,
 * ||A y_k - b||^2, 0 }, grad including alpha2*y.  with_residual = 0 skips the pass over A (out8[5] = 0): the
 * cheap way to read ||grad|| for the gradient-norm stop (:179). */
int fos_fista_trial(fos_fista* f, double t, int with_residual, double out8[8]);
/* The same test for nv <= 16 candidate steps t, t*eta, t*eta^2, ... decided by ONE pass over A on the matrix
 * cores (v_mfma_f32_16x16x4_f32, or v_mfma_f32_16x16x32_bf16 on three-term bf16 candidates when A is bf16;
 * LDS-staged A tiles; the candidates are the N dimension).  out (host) = nv rows of
 * 8 doubles, each laid out like out8 of fos_fista_trial.  Synchronises.  FOS_ERR_UNSUPPORTED when the problem runs
 * the two-pass fallback (ragged shapes): callers then loop over fos_fista_trial.  SURVEY.md 8(f) rank 1. */
int fos_fista_trial_batch(fos_fista* f, double t, double eta, int nv, double* out);
/* Backtracking WITHOUT a host round trip per iteration (:183-197 on the device): per iteration the gradient, one matrix-
 * core batch of 16 candidate steps tau, tau*eta, ..., a one-thread kernel that takes the reference's decision for them in
 * order (the same clauses as the host's test: (1-C) grad.dlt + 0.5||A dlt||^2 + 0.5 alpha2 ||dlt||^2 <= noise, grad_eps =
 * relative resolution of the gradient pass), the update with the accepted step and the scalar bookkeeping - all enqueued.
 * The step lives in the device state and persists across iterations and calls (:197).  ls_iters / tau_hist (device,
 * `iters` entries, nullable): shrinks and accepted step of every completed search of this call.  If no candidate of a
 * batch is accepted (the reference's step-underflow regime) the pipeline parks itself: status.stopped =
 * FOS_STOP_LS_STALL, status.k = iterations completed; fos_fista_resume_after_stall hands the current step to the host,
 * which finishes that search with fos_fista_trial_batch / fos_fista_update and may call this function again.
 * FOS_ERR_UNSUPPORTED on plans without the candidate pass (two-pass, n <= 64) and on resident problems (their whole
 * loop, backtracking included, is one launch already).  Enqueues only. */
int fos_fista_run_backtracking(fos_fista* f, int iters, double eta, double armijo_c, double grad_eps, int32_t* ls_iters,
                               double* tau_hist);
/* The same device-driven loop with the history (:224-232, :319-322) recorded on the device, for every configuration with
 * data-dependent control (backtracking = 1: Armijo search as above; adaptive restart and the stopping rules of
 * fos_fista_params in either case) - the plain configurations have fos_fista_run_history.  Per iteration i of this call:
 *   x_hist   row i  = the iterate after the iteration (iters x n doubles)
 *   hist     row i  = { unused, ||x||_1, ||x||_2^2, ||x - x_before||^2 } of that iterate (iters x 4 doubles)
 *   rr_seen  [i]    = ||A x - b||^2 of the iterate the iteration STARTED from (it comes out of the gradient pass - DUAL
 *                     kernel - or a residual pass of its own in precise mode), i.e. the objective ingredient of the
 *                     previous row; the caller closes the last row with one fos_residual_objective.  NULL: not wanted
 *                     (ista's log needs x, t and ||dx|| only) - the gradient pass is then the plain one.
 * Rows of iterations that did not complete (stop, parked search) are not written; status.k says how many did.
 * Enqueues only. */
int fos_fista_run_recorded(fos_fista* f, int iters, int backtracking, double eta, double armijo_c, double grad_eps,
                           double* x_hist, double* hist, double* rr_seen, int32_t* ls_iters, double* tau_hist);
int fos_fista_resume_after_stall(fos_fista* f, double* tau_out);      /* synchronises */
int fos_fista_status_get(fos_fista* f, fos_fista_status* out);   /* synchronises */
int fos_fista_get_x(fos_fista* f, double* dst);  /* enqueue copy of x_k (n doubles) to dst (device) */
double* fos_fista_x(fos_fista* f);       /* device pointer to x_k (n doubles), borrowed     */
float* fos_fista_gbuf(fos_fista* f);     /* device pointer to gbuf (n+1 floats), borrowed   */

/* ---- L-BFGS device pieces (the arithmetic behind lbfgs.py:64; spec in SURVEY.md 8c) -----------------
 * Three generations sit side by side because the ABI only grows: fp32 vectors (fos_lbfgs_two_loop, fos_vec_stats,
 * fos_vec_axpby: round 1; fos_vec_stats / fos_vec_axpby still serve the FISTA front-ends), fp64 iterate with fp32
 * gradients (*_f64: LEGACY, unused by the product), and fp64 throughout (*_dd, fos_lbfgs_direction_dd,
 * fos_lbfgs_minimize: what LBFGSSolver runs). */
/* K4: d = -H g by the two-loop recursion over the `hist` newest pairs.  S, Y: [cap][n] ring buffers,
 * slot (head + i) % cap holds the i-th oldest pair.  One launch. */
int fos_lbfgs_two_loop(const float* g, const float* S, const float* Y, int hist, int head, int cap, int64_t n,
                       float* d_out, void* stream);
/* out5 (device doubles) = { x.x, g.d, d.d, max|g|, ||x||_1 } in one launch; x, g, d may each be NULL. */
int fos_vec_stats(const float* x, const float* g, const float* d, int64_t n, double* out5, void* stream);
/* out = a*x + b*y (y may be NULL when b == 0). */
int fos_vec_axpby(double a, const float* x, double b, const float* y, float* out, int64_t n, void* stream);
/* fp64-iterate forms: x and out in fp64, g / d / y in fp32. */
int fos_vec_stats_f64(const double* x, const float* g, const float* d, int64_t n, double* out5, void* stream);
int fos_vec_axpby_f64(double a, const double* x, double b, const float* y, double* out, int64_t n, void* stream);

/* All-fp64 forms (g, d, S, Y, x doubles): what LBFGSSolver.fit runs, so that the direction, the curvature pairs and the
 * line-search scalars carry SciPy's precision.  Vectors 32-byte aligned for the register-resident two-loop kernel
 * (any alignment works, through the generic form). */
int fos_lbfgs_two_loop_dd(const double* g, const double* S, const double* Y, int hist, int head, int cap, int64_t n,
                          double* d_out, void* stream);
/* The same direction for a COLUMN-SHARDED problem (fos_problem_set_comm_cols): g, S, Y, d are this rank's blocks; the
 * partial Gram matrices of all ranks are summed by one all-reduce (64 x 256 doubles) between the two kernels, the
 * coefficient recursion is replicated, gd_out = {g.d, d.d} are global.  work: >= 64 * 256 doubles.  Enqueue-only. */
int fos_lbfgs_direction_cols(fos_problem* p, const double* g, const double* S, const double* Y, int hist, int head, int cap,
                             double* d_out, double* gd_out, double* work, int64_t work_doubles);
/* The same direction d = -H g computed by the whole chip in two launches (the two-loop chain above is bound to ONE
 * workgroup: 35 us at n = 8192 and 10 pairs, 0.9 ms at n = 65536): the Gram matrix of {s_i, y_i, g} by many workgroups,
 * then the recursion on 2*hist+1 coefficients and d as their combination (L-BFGS-B's own compact representation; equal to
 * the two-loop result in exact arithmetic, ~1e-13 relative in fp64).  hist <= 10 (FOS_ERR_UNSUPPORTED beyond).
 * gd_out (device or pinned host memory, may be NULL) receives { g.d, d.d }.  work: fos_lbfgs_direction_work(n) doubles of
 * device scratch.  This is what fos_lbfgs_minimize runs for n >= 2048. */
int64_t fos_lbfgs_direction_work(int64_t n);
int fos_lbfgs_direction_dd(const double* g, const double* S, const double* Y, int hist, int head, int cap, int64_t n,
                           double* d_out, double* gd_out, double* work, int64_t work_doubles, void* stream);
int fos_vec_stats_dd(const double* x, const double* g, const double* d, int64_t n, double* out5, void* stream);
/* out = a*x + b*y with the two products and the sum rounded separately (no fma contraction): bit for bit NumPy's
 * `stp * d + x_old`, `g - g_old`, `stp * d` (oracle lbfgs_minimize). */
int fos_vec_axpby_dd(double a, const double* x, double b, const double* y, double* out, int64_t n, void* stream);

/* ---- the optimiser itself: what scipy.optimize.fmin_l_bfgs_b is to lbfgs.py:64 --------------------------------------
 * Moré–Thuente line search (MINPACK-2 dcsrch with L-BFGS-B's constants ftol 1e-3, gtol 0.9, xtol 0.1, stpmin 0,
 * stpmax 1e10) in reverse communication: host scalars only, usable without a GPU.
 *   stp = fos_linesearch_begin(&ls, stp0, f(0), f'(0));  loop: stp = fos_linesearch_step(&ls, stp, f(stp), f'(stp))
 * until ls.status != FOS_LS_FG. */
enum { FOS_LS_FG = 0, FOS_LS_CONVERGENCE = 1, FOS_LS_WARNING = 2, FOS_LS_ERROR = 3 };
typedef struct fos_linesearch {
  double f0, d0, dtest, width, width1;
  double lo, flo, dlo, hi, fhi, dhi, smin, smax;
  int32_t stage1, brackt, status, reserved;
} fos_linesearch;
double fos_linesearch_begin(fos_linesearch* ls, double stp, double f0, double d0);
double fos_linesearch_step(fos_linesearch* ls, double stp, double f, double d);

/* The whole unbounded L-BFGS-B iteration of lbfgs.py:63-70 (SciPy defaults m = 10, factr = 1e7, maxls = 20; maxiter and
 * pgtol from the caller) as ONE call: fg = fos_gemv_pair_dd, direction = fos_lbfgs_direction_dd (n >= 2048) or the fp64
 * two-loop kernel, the line search above on the host, all vectors fp64 on the device.  ONE host round trip per fg
 * evaluation: the first trial point of an iteration (step 1 after the first iteration, L-BFGS-B's rule) and its
 * evaluation are enqueued together with the direction, the 8 scalars involved (loss terms, g.d, max|g|, ||x||_1 and the
 * direction's g.d, d.d) arrive in pinned host memory written by the kernels, behind a sequence flag the host polls; a
 * direction that turns out not to descend discards that evaluation (it is not counted in nfev).
 *   x         device, n doubles: start point in, solution out                                   lbfgs.py:63, :71
 *   hist      host, 2*max_iter doubles (nullable): after iteration k, hist[2k] = loss of fg at x_k, hist[2k+1] = ||x_k||_1
 *             (the callback's compute_objective(x_k) = loss + alpha1*||x_k||_1 without an extra pass)   lbfgs.py:56-61
 *   iterates  device, max_iter*n doubles (nullable): x_k after every iteration
 *   fg_ms     host, fg_cap floats (nullable): device milliseconds of every fg evaluation (get_metrics)
 * task: 0 |proj g| <= pgtol, 1 relative reduction of f <= factr*epsmch, 2 iteration limit, 3 abnormal termination in
 * the line search.  Works on sharded problems (fos_problem_set_comm).  Synchronises. */
typedef struct fos_lbfgs_result {
  double f, gmax;
  int32_t nit, nfev, task, reserved;
} fos_lbfgs_result;
int fos_lbfgs_minimize(fos_problem* p, double alpha2, int max_iter, double pgtol, double* x, double* hist,
                       double* iterates, float* fg_ms, int fg_cap, fos_lbfgs_result* res);

#ifdef __cplusplus
}
#endif
#endif /* FOS_H_ */
