// Internal header of libfos_hip.so: the handle types behind include/fos.h and the launch helpers shared by the translation
// units of the C ABI (fos_plan.hip: planner, kernel menus, problem-level entry points; fos_comm.hip: communicators;
// fos_fista.hip: the FISTA state machine; fos_lbfgs.hip: L-BFGS).  Nothing here is exported.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <chrono>
#include <vector>

// the library is built with -fvisibility=hidden: only what include/fos.h declares is exported
#pragma GCC visibility push(default)
#include "../../include/fos.h"
#pragma GCC visibility pop
#include "batch_trial.hpp"
#include "cluster_pass.hpp"
#include "comm.hpp"
#include "gemv_multi.hpp"
#include "gemv_pair.hpp"
#include "gemv_tall.hpp"
#include "gemv_wide.hpp"
#include "gram_batch.hpp"
#include "lbfgs_driver.hpp"
#include "lbfgs_kernels.hpp"
#include "reduce_update.hpp"
#include "resident.hpp"


namespace fosapi {

extern thread_local std::string g_err;            // fos_last_error(): per-thread text of the last failure

inline int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
#define HIP_TRY(expr)                                                                                   \
  do {                                                                                                  \
    hipError_t e_ = (expr);                                                                             \
    if (e_ != hipSuccess)                                                                               \
      return fail(FOS_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));                      \
  } while (0)
#define LAUNCH_CHECK()                                                                                  \
  do {                                                                                                  \
    hipError_t e_ = hipGetLastError();                                                                  \
    if (e_ != hipSuccess) return fail(FOS_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e_)); \
  } while (0)


using fos::YSource;

typedef void (*FusedLaunch)(const void* A, int64_t lda, const float* b, int64_t m, int n, YSource ys, int64_t rpw,
                            float* slabs, double* rr_part, double* rr2_part, int nwg, hipStream_t st);

typedef void (*FusedLaunchDD)(const void* A, int64_t lda, const float* b, int64_t m, int n, YSource ys, int64_t rpw,
                              double* slabs, double* rr_part, int nwg, hipStream_t st);
struct MenuEntry {
  int dtype, threads, k, r;
  FusedLaunch with_g, resid_only, dual;   // dual may be null (geometry without a DUAL instantiation)
  FusedLaunchDD dd;                       // tall entries only: the same kernel writing fp64 slabs
  // column-block instantiations (CB = true: negated-residual store + slab stride), only on the two geometries a column
  // block can land on (block widths lie in (8192, 16384]); null elsewhere
  FusedLaunch with_g_cb = nullptr, resid_only_cb = nullptr;
  // interleaved-rows instantiations (IL = true: rows dealt round-robin, all CUs stream ONE contiguous window), on the
  // geometries of rows >= 16 KiB; null elsewhere
  FusedLaunch with_g_il = nullptr, resid_only_il = nullptr, dual_il = nullptr;
};
// Streaming geometries of the fp64-accumulating pass, ordered by capacity.  y and the gradient slice cost two VGPRs
// per column here, so the wide rows take 512 threads x 8 chunks (2 waves per SIMD, 256 VGPRs) instead of 1024 x 4.
struct DdEntry { int dtype, threads, k, r; FusedLaunchDD fn; FusedLaunchDD fn_il = nullptr; };
typedef void (*MultiLaunch)(const float* A, int64_t lda, const float* b, int64_t m, int n, fos::MultiY ys, int64_t rpw,
                            float* slabs, double* rr_part, int nwg, hipStream_t st);
}  // namespace fosapi

using fosapi::MenuEntry;
using fosapi::DdEntry;

// Device + pinned workspace of fos_lbfgs_minimize, cached on the problem handle: round 2 allocated it per fit (7 hipMalloc +
// hipHostMalloc + 2 events per fg, and the hipFree's at the end drain the device) inside an 8 ms fit.
struct LbfgsWork {
  int64_t n = 0;
  double *g = nullptr, *g_old = nullptr, *d = nullptr, *x_old = nullptr, *S = nullptr, *Y = nullptr, *vl = nullptr;
  double* host = nullptr;              // pinned: 16 doubles
  unsigned long long* t_start = nullptr;   // device: wall-clock stamp taken in front of an evaluation
  double ticks_per_ms = 1e5;           // hipDeviceAttributeWallClockRate (kHz)
  bool pass_stamps = false;           // the pass kernel of this plan stamps its own start (dd_pass_stamps)
  ~LbfgsWork() {
    void* bufs[] = {g, g_old, d, x_old, S, Y, vl, t_start};
    for (void* q : bufs)
      if (q) (void)hipFree(q);
    if (host) (void)hipHostFree(host);
  }
};

struct fos_problem {
  const void* A = nullptr;
  const float* b = nullptr;
  int64_t m = 0, n = 0, lda = 0;
  int dtype = FOS_F32;
  hipStream_t stream = nullptr;
  int ncu = 256;
  fos_comm* comm = nullptr;          // row-sharded problem: sums of partial results go through it (comm.hpp)
  LbfgsWork* lbfgs = nullptr;        // fos_lbfgs_minimize workspace, allocated by the first fit
  bool col_sharded = false;          // comm splits the COLUMNS instead: this rank holds A[:, its columns], x is partitioned
  unsigned plan_flags = 0;           // FOS_PLAN_* given to fos_problem_replan
  bool allow_resident = true;
  bool il = false;                   // rows dealt round-robin to the workgroups (FOS_PLAN_INTERLEAVE / planner default for big rows)
  // plan
  int path = 0;                      // 0 fused, 1 two-pass fallback
  bool resident = false;             // small enough for the single-launch LDS-resident loop (resident.hpp)
  bool tall = false;                 // n <= 64: row-per-thread single pass (gemv_tall.hpp); no alignment requirements
  // Rows wider than one workgroup's registers / LDS (fp32 > 32768 columns, bf16 > 16384): column blocks of cb_width
  // columns through the streaming kernel in two phases, r = A y - b block by block, then A^T r block by block
  bool colblock = false;
  int64_t cb_width = 0;
  float* rneg = nullptr;             // m floats: the negated residual between the two phases
  float* zeros = nullptr;            // cb_width floats of zeros (phase 2 runs the same kernel with y = 0, b = -r)
  int64_t slab_stride = 0;           // floats between slab rows (0 = n); the tall pass pads rows to a multiple of 4
  const MenuEntry* entry = nullptr;
  int nwg = 0;                       // workgroups of the fused kernel
  int nslabs = 0;
  int64_t rows_per_wg = 0;
  int resid_grid = 0;                // fallback pass-1 grid
  bool vec4 = false;                 // n % 4 == 0: float4 epilogues
  // workspace
  int slab_cap = 0, rr_cap = 0;
  float* slabs = nullptr;
  double* rr_part = nullptr;
  double* rr2_part = nullptr;        // DUAL pass: partials of ||A x_k - b||^2
  double* rvec = nullptr;            // fallback: residual (m doubles)
  float* gbuf = nullptr;             // n + 4 floats (internal, or caller-owned after fos_problem_set_gbuf)
  float* gbuf_own = nullptr;
  // fp64-accumulating pass (fos_gemv_pair_dd): own geometry and fp64 slabs, allocated on first use
  const DdEntry* dd_entry = nullptr;
  int dd_nwg = 0;
  int dd_nwg_hint = 0;               // fos_problem_tune_dd: workgroups of the streaming fp64 pass (0 = planner)
  int64_t dd_rows_per_wg = 0;
  double* slabs_dd = nullptr;
  double* rr_dd = nullptr;
  int dd_two_pass_chunks = 0;        // > 0: this shape runs the fp64 two-pass kernels for the dd pass
  float* ybuf = nullptr;             // n floats: aligned copy of a caller vector when needed
  double* dscal = nullptr;           // 256 device doubles (scalars)
  double* lhist = nullptr;           // power iteration: L after every step (n_iter + 1 doubles, grown on demand)
  int lhist_cap = 0;
  double* part = nullptr;            // partial sums of the small kernels
  int part_cap = 0;
  // batched (MFMA) residual: permuted candidate block, per-workgroup partials, folded results
  float* xp = nullptr;
  double* q_part = nullptr;
  double* bt_out = nullptr;          // 128 doubles
  int64_t n_pad = 0;
  // multi-lambda pass on the matrix cores (gram_batch.hpp): residual panel and the 16 gradient slab sets
  float* rbuf16 = nullptr;           // panel_rows x 16 floats
  float* rcols16 = nullptr;          // column-sharded candidate pass: m x 16 partial residuals (summed over the ranks)
  double* cr_part = nullptr;         // chip-resident loop (chip_resident.hpp): [2][G][17] partials + [8] step sums + [1] rr
  unsigned* cr_bar = nullptr;
  unsigned long long* fz_stamps = nullptr;   // caller-owned (fos_problem_set_fused_stamps), [ncu][8]
  double* mfold = nullptr;           // column-sharded lockstep: 16 x 4 folded step partials (summed over the ranks)
  float* slabs16 = nullptr;          // splits x 16 x n floats
  int64_t panel_rows = 0;
  int gram_splits = 0;
  int64_t gram_rows_per_split = 0;
  // one-read form of the same pass (cluster_pass.hpp): hand-off ring, flags, launch epoch
  int cp_cs = 0, cp_clusters = 0;    // members per cluster (0: shape not served), clusters
  int64_t cp_rows_per_cluster = 0;
  float* cp_xchg = nullptr;
  unsigned* cp_flags = nullptr;
  int* cp_error = nullptr;
  unsigned cp_epoch = 1;
  int cp_mode = 0;                   // 0: planner's choice, 1: FOS_PLAN_CLUSTER, 2: FOS_PLAN_NO_CLUSTER
  // fused persistent step (fused_step.hpp, fos_fista_run_fused): barrier words, per-workgroup partials, beta sequence
  int chip_mode = 0;                 // 0: planner (fos_fista_run_chip where it measured ahead), 1: FOS_PLAN_CHIP_RESIDENT, 2: never
  bool fused_on = false;             // FOS_PLAN_FUSED_MFMA: plain fos_fista_run calls take fos_fista_run_fused where served
  unsigned* fz_bar = nullptr;
  double* fz_part = nullptr;
  double* fz_beta = nullptr;
  int fz_beta_cap = 0;
  // optional kernel timing (fos_problem_profile)
  int profiling = 0;                 // 0 off, N: bracket every N-th launch of the A pass
  int64_t prof_seq = 0;
  bool prof_open = false;
  std::vector<hipEvent_t> ev_pool;   // pairs: [2i] start, [2i+1] stop
  size_t ev_used = 0;
  double prof_ms = 0.0;
  int64_t prof_launches = 0;
};

struct fos_fista {
  fos_problem* p = nullptr;
  fos::FistaParams prm{};
  // host mirror of the momentum scalars, valid while only plain fos_fista_run calls advance the state
  bool host_valid = false;
  double h_t = 1.0, h_beta = 0.0;
  long long h_k = 0;
  double* part2 = nullptr;           // ping-pong partials for plain runs: 2 * nupd * 4 doubles
  float* ynext = nullptr;            // plain runs: y_{k+1} in fp32 written by the update kernel
  bool y_valid = false;              // ynext holds y for iteration h_k
  bool pending = false;              // plain split-mode updates whose scalar bookkeeping has not run yet
  long long plain_count = 0;         // consecutive plain iterations whose partials sit in part2
  double *x_cur = nullptr, *x_prev = nullptr;   // fp64 iterate state
  float* dlt = nullptr;                         // trial difference vector x_tmp - y_k (fp32)
  fos::FistaScalars* scal = nullptr;
  // precise mode (fos_fista_set_precise): the split-form gradient comes from the fp64-accumulating pass at the unrounded
  // fp64 y_k, so that the Armijo comparison g(x_tmp) <= g(y) + C grad.dlt is decided on fp64-accurate terms
  bool precise = false;
  double* folded = nullptr;          // column-sharded: the 4 update sums of an iteration, folded and summed over the ranks
  bool tau_on_device = false;        // FistaScalars::tau is authoritative (device-driven backtracking ran since the last set_tau / reset)
  double* gbuf64 = nullptr;          // n + 4 doubles: [gradient ; ||r||^2]
  bool gbuf64_owned = false;         // allocated by fos_fista_set_precise (false: the caller's, fos_fista_set_gbuf64)
  double* out5 = nullptr;            // device
  int nupd = 0;                      // workgroups of the update kernel
};

namespace fosapi {

// ---- fos_plan.hip ---------------------------------------------------------------------------------------------------
const MenuEntry* wide_entry(int dtype);           // the y-in-LDS pass (gemv_wide.hpp): 16385..32768 fp32, 24577..32768 bf16 columns
const MenuEntry* find_entry(int dtype, int threads, int k, int r);
const MenuEntry* default_entry(int dtype, int64_t n);
int epc_of(int dtype);
int grid_1d(int64_t n, int per_block, int cap);
void plan_fused(fos_problem* p, const MenuEntry* e, int nwg_hint);
void apply_plan(fos_problem* p, unsigned flags);
int ensure_workspace(fos_problem* p);
int ensure_batch_workspace(fos_problem* p);
int ensure_dd(fos_problem* p);
int prof_drain(fos_problem* p);
int prof_mark(fos_problem* p, bool start);
int aligned_vec(fos_problem* p, const float* v, const float** out);
bool batch_supported(const fos_problem* p);
MultiLaunch find_multi(int64_t n, int nv);
// Enqueue the A pass for `ys`.  with_g: also produce the slabs (A^T r).  *n_rr: number of rr partials written.
int launch_pass(fos_problem* p, const YSource& ys, const float* b, bool with_g, int* n_rr, bool dual = false);
// slabs -> gbuf[0..n], summed over the ranks when the problem is row-sharded; rr_out (nullable) = the global ||r||^2
int launch_slab_reduce(fos_problem* p, int n_rr, float* gbuf, double* rr_out, const int* stopped);
// Product 1 on `rows` rows starting at A / b: q_part[wg][16] partial squared norms, rout (nullable): the residuals
int launch_batch_product(fos_problem* p, const void* A, const float* b, int64_t rows_total, int use_b, float* rout, int* nwg_out,
                         const int* stopped = nullptr);
// q[j] = ||A Xp_j - use_b*b||^2 -> out16 (device); Xp already in p->xp
int launch_residual_batch(fos_problem* p, int use_b, double* out16, const int* stopped = nullptr);
int launch_cluster_pass(fos_problem* p);
// the fp64-accumulating pass for any y source: out[0..n) = A^T (A y - b) + alpha2*l2vec, out[n] = ||A y - b||^2
int launch_pass_dd(fos_problem* p, const YSource& ys, double alpha2, const double* l2vec, double* out);
int gemv_pair_dd_stamped(fos_problem* p, const double* x, double alpha2, double* grad_rr, unsigned long long* t_stamp, bool* stamped);
int dd_pass_stamps(fos_problem* p, bool* yes);
// ---- fos_comm.hip ---------------------------------------------------------------------------------------------------
// in-place sum over the ranks of a communicator on `st`: RCCL, or the one-shot full-mesh kernel (comm.hpp)
int comm_allreduce(fos_comm* c, void* buf, size_t count, bool f64, hipStream_t st);
// sum `count` floats / doubles over the ranks of a sharded problem, in place, on the handle's stream (no-op otherwise)
int reduce_across(fos_problem* p, void* buf, size_t count, bool f64);

}  // namespace fosapi
