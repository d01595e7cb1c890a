// libfos_hip.so — C ABI (include/fos.h) over the gfx950 kernels.  Host code only decides launch geometry and
// enqueues kernels; there is no CPU compute fallback anywhere in this file.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <chrono>
#include <vector>

#include "../../include/fos.h"
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

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
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

// ---- fused-kernel menu -------------------------------------------------------------------------------
typedef void (*FusedLaunch)(const void* A, int64_t lda, const float* b, int64_t m, int n, YSource ys, int64_t rpw,
                            float* slabs, double* rr_part, double* rr2_part, int nwg, hipStream_t st);

template <typename T, int THREADS, int K, int R, int MINW, bool WITH_G, int NBUF, bool DUAL, bool DRAIN = false, bool CB = false,
          bool IL = false>
void fused_launch(const void* A, int64_t lda, const float* b, int64_t m, int n, YSource ys, int64_t rpw, float* slabs,
                  double* rr_part, double* rr2_part, int nwg, hipStream_t st) {
  hipLaunchKernelGGL((fos::gemv_pair_kernel<T, THREADS, K, R, true, MINW, WITH_G, NBUF, IL, DUAL, DRAIN, float, false, CB>), dim3(nwg),
                     dim3(THREADS), 0, st, reinterpret_cast<const T*>(A), lda, b, m, n, ys, rpw, slabs, rr_part,
                     rr2_part);
}

// fp64-accumulating form (fos_gemv_pair_dd): y and the slabs are doubles
typedef void (*FusedLaunchDD)(const void* A, int64_t lda, const float* b, int64_t m, int n, YSource ys, int64_t rpw,
                              double* slabs, double* rr_part, int nwg, hipStream_t st);
// Dynamic shared memory above the 64 KiB default has to be granted per kernel AND per device (the attribute lives in
// the device's code object); a bitmask of devices already served, updated atomically, keeps this thread-safe.
template <typename K>
int raise_dynamic_lds(K kernel, size_t bytes, std::atomic<uint64_t>& done) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return FOS_ERR_HIP;
  const uint64_t bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return FOS_OK;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) !=
      hipSuccess)
    return FOS_ERR_HIP;
  done.fetch_or(bit, std::memory_order_release);
  return FOS_OK;
}

template <typename T, int THREADS, int K, int R, int MINW, bool YLDS, int NB = 2, bool IL = false, bool KEEPCVT = false>
void fused_launch_dd(const void* A, int64_t lda, const float* b, int64_t m, int n, YSource ys, int64_t rpw, double* slabs,
                     double* rr_part, int nwg, hipStream_t st) {
  auto kern = fos::gemv_pair_kernel<T, THREADS, K, R, true, MINW, true, NB, IL, false, false, double, YLDS, false, false, KEEPCVT>;
  constexpr size_t lds = YLDS ? (size_t)THREADS * K * fos::ElemTraits<T>::EPC * sizeof(double) : 0;
  if constexpr (lds > 65536) {
    static std::atomic<uint64_t> done{0};
    (void)raise_dynamic_lds(kern, lds, done);       // on failure the launch below fails and LAUNCH_CHECK reports it
  }
  hipLaunchKernelGGL(kern, dim3(nwg), dim3(THREADS), lds, st, reinterpret_cast<const T*>(A), lda, b, m, n, ys, rpw, slabs,
                     rr_part, (double*)nullptr);
}

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
// three register tiles in flight where that measured faster (same-run ratio to the fp32 kernel, tools/bench_dd.py:
// (512,4) fp32 0.96 -> 0.99, bf16 (256,4) 0.77 -> 0.83, bf16 (512,4) 0.77 -> 0.80; the 256-thread fp32 geometries lost
// 2-3 % and the (512,8) fp32 geometry would spill: those keep two)
#define DD_ENTRY(DT, T, TH, K, R, YL) { DT, TH, K, R, fused_launch_dd<T, TH, K, R, 2, YL, 2> }
#define DD_ENTRY3(DT, T, TH, K, R, YL) { DT, TH, K, R, fused_launch_dd<T, TH, K, R, 2, YL, 3> }
#define DD_ENTRY_IL(DT, T, TH, K, R, YL) { DT, TH, K, R, fused_launch_dd<T, TH, K, R, 2, YL, 2>, fused_launch_dd<T, TH, K, R, 2, YL, 2, true> }
#define DD_ENTRY3_IL(DT, T, TH, K, R, YL) { DT, TH, K, R, fused_launch_dd<T, TH, K, R, 2, YL, 3>, fused_launch_dd<T, TH, K, R, 2, YL, 3, true> }
const DdEntry kDdMenu[] = {
    DD_ENTRY(FOS_F32, float, 64, 1, 4, false), DD_ENTRY(FOS_F32, float, 64, 2, 2, false),
    DD_ENTRY(FOS_F32, float, 256, 1, 2, false), DD_ENTRY(FOS_F32, float, 256, 2, 2, false),
    DD_ENTRY_IL(FOS_F32, float, 256, 4, 1, false), DD_ENTRY3_IL(FOS_F32, float, 512, 4, 1, false),
    DD_ENTRY_IL(FOS_F32, float, 512, 8, 1, true),
    DD_ENTRY(FOS_BF16, fos::bf16_t, 64, 1, 2, false), DD_ENTRY(FOS_BF16, fos::bf16_t, 256, 1, 2, false),
    DD_ENTRY(FOS_BF16, fos::bf16_t, 256, 2, 1, false),
    // bf16 rows of 4 chunks per thread: the pass is VALU-bound (56 VALU instructions per 16-byte chunk: unpack + convert
    // for the dot, AGAIN for the gradient update, 16 v_fma_f64), so these keep the converted tile across the barrier
    // (KEEPCVT: 40 instructions) and pay with registers - two tiles in flight instead of three, per-chunk scheduling
    // barriers, the cross-wave sum through the DPP ladder.  tools/dd_bench: 262144 x 8192 69.5 -> 78.2 % of 8 TB/s,
    // 131072 x 16384 66.6 -> 74.5 %.
    { FOS_BF16, 256, 4, 1, fused_launch_dd<fos::bf16_t, 256, 4, 1, 2, true, 2, false, true>,
      fused_launch_dd<fos::bf16_t, 256, 4, 1, 2, true, 2, true, true> },
    { FOS_BF16, 512, 4, 1, fused_launch_dd<fos::bf16_t, 512, 4, 1, 2, true, 2, false, true>,
      fused_launch_dd<fos::bf16_t, 512, 4, 1, 2, true, 2, true, true> },
};
#undef DD_ENTRY
#undef DD_ENTRY3
#undef DD_ENTRY_IL
#undef DD_ENTRY3_IL
// with-gradient / residual-only pair of a geometry: NB register tiles, drained or not, column-block (CB) or interleaved (IL)
#define PAIR(T, TH, K, R, W, NB, DRAIN, CB, IL) \
  fused_launch<T, TH, K, R, W, true, NB, false, DRAIN, CB, IL>, fused_launch<T, TH, K, R, W, false, NB, false, DRAIN, CB, IL>
// Every entry carries the column-block pair: a COLUMN-SHARDED problem (fos_problem_set_comm_cols) runs the two-phase
// plan at whatever width a rank's block has; the unsharded planner only ever lands on the two widest geometries.
#define ENTRY(DT, T, TH, K, R, W) \
  { DT, TH, K, R, PAIR(T, TH, K, R, W, 2, false, false, false), nullptr, nullptr, PAIR(T, TH, K, R, W, 2, false, true, false) }
// D: with DUAL
#define ENTRY_D(DT, T, TH, K, R, W) \
  { DT, TH, K, R, PAIR(T, TH, K, R, W, 2, false, false, false), fused_launch<T, TH, K, R, W, true, 2, true>, nullptr, \
    PAIR(T, TH, K, R, W, 2, false, true, false) }
// NB register tiles in flight (profiles/r01_kbench_exp_*.log: 3 tiles are worth 1.5 % at n = 8192) + DUAL + the
// interleaved-rows forms (rows >= 16 KiB)
#define ENTRY_NB_IL(DT, T, TH, K, R, W, NB) \
  { DT, TH, K, R, PAIR(T, TH, K, R, W, NB, false, false, false), fused_launch<T, TH, K, R, W, true, 2, true>, nullptr, \
    PAIR(T, TH, K, R, W, NB, false, true, false), PAIR(T, TH, K, R, W, NB, false, false, true), \
    fused_launch<T, TH, K, R, W, true, 2, true, false, false, true> }
// drained pipeline (profiles/r01_kbench_exp2_*: best form for 64 KiB rows), with DUAL
// The DUAL pass of such an entry runs another geometry of the same row step R (the 1024-thread form has no registers
// left for the second vector): TH2 x K2 must cover the same n.
#define ENTRY_DRAIN(DT, T, TH, K, R, W, TH2, K2, W2) \
  { DT, TH, K, R, PAIR(T, TH, K, R, W, 2, true, false, false), fused_launch<T, TH2, K2, R, W2, true, 2, true, false>, nullptr, \
    PAIR(T, TH, K, R, W, 2, true, true, false), PAIR(T, TH, K, R, W, 2, true, false, true), \
    fused_launch<T, TH2, K2, R, W2, true, 2, true, false, false, true> }
// Ordered by capacity (threads*k*EPC columns); first entry that fits n is the default.
// The 64-thread entries give narrow rows (65..512 columns) one WAVE per row instead of a 256-thread workgroup whose
// lanes would mostly idle (200000 x 256 ran at 18 % of the roofline on the 256-thread geometry); they are launched
// with proportionally more workgroups (plan_fused).
const MenuEntry kMenu[] = {
    ENTRY_D(FOS_F32, float, 64, 1, 4, 2), ENTRY_D(FOS_F32, float, 64, 2, 4, 2),
    ENTRY_D(FOS_F32, float, 256, 1, 4, 2), ENTRY_D(FOS_F32, float, 256, 2, 4, 2), ENTRY_D(FOS_F32, float, 256, 4, 2, 2),
    ENTRY_NB_IL(FOS_F32, float, 512, 4, 1, 2, 3),   ENTRY_DRAIN(FOS_F32, float, 1024, 4, 1, 4, 512, 8, 2),
    ENTRY_D(FOS_F32, float, 512, 8, 1, 2),  ENTRY(FOS_F32, float, 1024, 2, 2, 4),
    ENTRY(FOS_BF16, fos::bf16_t, 64, 1, 4, 2),                      // one wave per row: up to 512 bf16 columns
    ENTRY(FOS_BF16, fos::bf16_t, 256, 1, 4, 2), ENTRY(FOS_BF16, fos::bf16_t, 256, 2, 2, 2),
    ENTRY_NB_IL(FOS_BF16, fos::bf16_t, 256, 4, 1, 2, 3), ENTRY_NB_IL(FOS_BF16, fos::bf16_t, 512, 4, 1, 2, 3),
};

const MenuEntry* find_entry(int dtype, int threads, int k, int r) {
  for (const auto& e : kMenu)
    if (e.dtype == dtype && e.threads == threads && e.k == k && e.r == r) return &e;
  return nullptr;
}
int epc_of(int dtype) { return dtype == FOS_F32 ? 4 : 8; }
const MenuEntry* default_entry(int dtype, int64_t n) {
  for (const auto& e : kMenu)
    if (e.dtype == dtype && (int64_t)e.threads * e.k * epc_of(dtype) >= n) return &e;
  return nullptr;
}

// ---- wide rows (gemv_wide.hpp): 16384 < n <= 32768 fp32, y in LDS (dynamic shared memory above the 64 KiB default) ---
template <bool WITH_G>
void wide_launch(const void* A, int64_t lda, const float* b, int64_t m, int n, YSource ys, int64_t rpw, float* slabs,
                 double* rr_part, double* /*rr2_part*/, int nwg, hipStream_t st) {
  static std::atomic<uint64_t> done{0};
  (void)raise_dynamic_lds(&fos::gemv_wide_kernel<WITH_G>, fos::WD_MAX_N * sizeof(float), done);   // failure: see LAUNCH_CHECK
  hipLaunchKernelGGL((fos::gemv_wide_kernel<WITH_G>), dim3(nwg), dim3(fos::WD_THREADS), (size_t)n * sizeof(float), st,
                     reinterpret_cast<const float*>(A), lda, b, m, n, ys, rpw, slabs, rr_part);
}
const MenuEntry kWideF32 = {FOS_F32, fos::WD_THREADS, fos::WD_K, 1, wide_launch<true>, wide_launch<false>, nullptr};

// ---- tall-skinny entries (gemv_tall.hpp): n <= 64, any m / lda; one entry per column capacity and load form --------
template <typename T, int NC, int LOAD, bool WITH_G, bool DUAL>
void tall_launch(const void* A, int64_t lda, const float* b, int64_t m, int n, YSource ys, int64_t rpw, float* slabs,
                 double* rr_part, double* rr2_part, int nwg, hipStream_t st) {
  // the staged float4 copy streams one contiguous span: non-temporal (8000000 x 5: 61 -> 70 % of 8 TB/s, tools/tall_bench)
  hipLaunchKernelGGL((fos::gemv_tall_kernel<T, NC, LOAD, WITH_G, DUAL, float, LOAD == fos::TL_STAGE4>), dim3(nwg),
                     dim3(fos::TL_THREADS), 0, st, reinterpret_cast<const T*>(A), lda, b, m, n, ys, rpw, slabs, rr_part, rr2_part);
}
template <typename T, int NC, int LOAD>
void tall_launch_dd(const void* A, int64_t lda, const float* b, int64_t m, int n, YSource ys, int64_t rpw, double* slabs,
                    double* rr_part, int nwg, hipStream_t st) {
  hipLaunchKernelGGL((fos::gemv_tall_kernel<T, NC, LOAD, true, false, double, LOAD == fos::TL_STAGE4>), dim3(nwg),
                     dim3(fos::TL_THREADS), 0, st, reinterpret_cast<const T*>(A), lda, b, m, n, ys, rpw, slabs, rr_part, (double*)nullptr);
}
template <typename T, bool VEC>
void tallq_launch_dd(const void* A, int64_t lda, const float* b, int64_t m, int n, YSource ys, int64_t rpw, double* slabs,
                     double* rr_part, int nwg, hipStream_t st) {
  hipLaunchKernelGGL((fos::gemv_tall_quad_kernel<T, VEC, true, false, double>), dim3(nwg), dim3(fos::TL_THREADS), 0, st,
                     reinterpret_cast<const T*>(A), lda, b, m, n, ys, rpw, slabs, rr_part, (double*)nullptr);
}
template <typename T, bool VEC, bool WITH_G, bool DUAL>
void tallq_launch(const void* A, int64_t lda, const float* b, int64_t m, int n, YSource ys, int64_t rpw, float* slabs,
                  double* rr_part, double* rr2_part, int nwg, hipStream_t st) {
  hipLaunchKernelGGL((fos::gemv_tall_quad_kernel<T, VEC, WITH_G, DUAL>), dim3(nwg), dim3(fos::TL_THREADS), 0, st,
                     reinterpret_cast<const T*>(A), lda, b, m, n, ys, rpw, slabs, rr_part, rr2_part);
}
template <typename T, int LPR, bool WITH_G, bool DUAL>
void tallr_launch(const void* A, int64_t lda, const float* b, int64_t m, int n, YSource ys, int64_t rpw, float* slabs,
                  double* rr_part, double* rr2_part, int nwg, hipStream_t st) {
  hipLaunchKernelGGL((fos::gemv_tall_rows_kernel<T, LPR, WITH_G, DUAL>), dim3(nwg), dim3(fos::TL_THREADS), 0, st,
                     reinterpret_cast<const T*>(A), lda, b, m, n, ys, rpw, slabs, rr_part, rr2_part);
}
template <typename T, int LPR>
void tallr_launch_dd(const void* A, int64_t lda, const float* b, int64_t m, int n, YSource ys, int64_t rpw, double* slabs,
                     double* rr_part, int nwg, hipStream_t st) {
  hipLaunchKernelGGL((fos::gemv_tall_rows_kernel<T, LPR, true, false, double>), dim3(nwg), dim3(fos::TL_THREADS), 0, st,
                     reinterpret_cast<const T*>(A), lda, b, m, n, ys, rpw, slabs, rr_part, (double*)nullptr);
}
// aligned rows: a row per LPR lanes, one 16-byte chunk per lane (gemv_tall_rows_kernel); k = LPR marks the entry
#define TALLR(DT, T, LPR) \
  { DT, fos::TL_THREADS, LPR, 0, tallr_launch<T, LPR, true, false>, tallr_launch<T, LPR, false, false>, \
    tallr_launch<T, LPR, true, true>, tallr_launch_dd<T, LPR> }
// (a row per 4 lanes - up to 4 chunks - measured slower than the row-per-thread form and is not instantiated)
const MenuEntry kTallRowsF32[2] = {TALLR(FOS_F32, float, 8), TALLR(FOS_F32, float, 16)};
const MenuEntry kTallRowsBf16[1] = {TALLR(FOS_BF16, fos::bf16_t, 8)};
#undef TALLR
// 33..64 columns: a row per quad of lanes (gemv_tall_quad_kernel)
#define TALLQ(DT, T, VEC) \
  { DT, fos::TL_THREADS, 0, 0, tallq_launch<T, VEC, true, false>, tallq_launch<T, VEC, false, false>, \
    tallq_launch<T, VEC, true, true>, tallq_launch_dd<T, VEC> }
#define TALL(DT, T, NC, LD) \
  { DT, fos::TL_THREADS, 0, 0, tall_launch<T, NC, LD, true, false>, tall_launch<T, NC, LD, false, false>, \
    tall_launch<T, NC, LD, true, true>, tall_launch_dd<T, NC, LD> }
#define TALL_ROW(DT, T, NC) \
  { TALL(DT, T, NC, fos::TL_DIRECT), TALL(DT, T, NC, fos::TL_VEC), TALL(DT, T, NC, fos::TL_STAGE), TALL(DT, T, NC, fos::TL_STAGE4) }
const MenuEntry kTallF32[4][4] = {
    TALL_ROW(FOS_F32, float, 8), TALL_ROW(FOS_F32, float, 16), TALL_ROW(FOS_F32, float, 32),
    {TALLQ(FOS_F32, float, false), TALLQ(FOS_F32, float, true), TALLQ(FOS_F32, float, false), TALLQ(FOS_F32, float, false)}};
const MenuEntry kTallBf16[4][2] = {
    {TALL(FOS_BF16, fos::bf16_t, 8, fos::TL_DIRECT), TALL(FOS_BF16, fos::bf16_t, 8, fos::TL_STAGE)},
    {TALL(FOS_BF16, fos::bf16_t, 16, fos::TL_DIRECT), TALL(FOS_BF16, fos::bf16_t, 16, fos::TL_STAGE)},
    {TALL(FOS_BF16, fos::bf16_t, 32, fos::TL_DIRECT), TALL(FOS_BF16, fos::bf16_t, 32, fos::TL_STAGE)},
    {TALLQ(FOS_BF16, fos::bf16_t, false), TALLQ(FOS_BF16, fos::bf16_t, false)}};
#undef TALL
#undef TALL_ROW
#undef TALLQ
// load form: 16-byte row loads when the layout allows, LDS staging for contiguous ragged matrices, scalar loads otherwise
const MenuEntry* tall_entry(int dtype, int64_t n, int64_t lda, const void* A) {
  const int epc = dtype == FOS_F32 ? 4 : 8;
  // Rows of 5..16 chunks of 16 bytes (fp32: 17..64 columns, bf16: 33..64): a row per 8 / 16 lanes, one chunk per lane
  // (4000000 x 32: 52 % -> 74 % of the roofline, 2000000 x 64: 53 % -> 73 %).  Up to 4 chunks the row-per-thread form
  // with 16-byte loads is the faster one (4000000 x 16: 72-76 % against 67-69 %, profiles/r02_sweep_wgs.log).
  if (n % epc == 0 && lda % epc == 0 && (reinterpret_cast<uintptr_t>(A) & 15u) == 0 && n / epc > 4) {
    const int chunks = (int)(n / epc);
    if (dtype == FOS_F32) return &kTallRowsF32[chunks <= 8 ? 0 : 1];
    return &kTallRowsBf16[0];
  }
  const int idx = n <= 8 ? 0 : n <= 16 ? 1 : n <= 32 ? 2 : 3;
  const bool contiguous = (lda == n);
  if (dtype == FOS_F32) {
    const bool vec = n % 4 == 0 && lda % 4 == 0 && (reinterpret_cast<uintptr_t>(A) & 15u) == 0;
    // contiguous ragged rows: staged through LDS; float4 copy when the matrix is 16-byte aligned (plan_tall keeps the
    // rows per workgroup a multiple of 4, so every block then starts on a 16-byte boundary)
    const bool a16 = (reinterpret_cast<uintptr_t>(A) & 15u) == 0;
    return &kTallF32[idx][vec ? fos::TL_VEC : (contiguous ? (a16 && idx < 3 ? fos::TL_STAGE4 : fos::TL_STAGE) : fos::TL_DIRECT)];
  }
  return &kTallBf16[idx][contiguous ? 1 : 0];
}

int grid_1d(int64_t n, int per_block, int cap) {
  int64_t g = (n + per_block - 1) / per_block;
  return (int)std::max<int64_t>(1, std::min<int64_t>(g, cap));
}

}  // namespace

// Device + pinned workspace of fos_lbfgs_minimize, cached on the problem handle: round 2 allocated it per fit (7 hipMalloc +
// hipHostMalloc + 2 events per fg, and the hipFree's at the end drain the device) inside an 8 ms fit.
struct LbfgsWork {
  int64_t n = 0;
  double *g = nullptr, *g_old = nullptr, *d = nullptr, *x_old = nullptr, *S = nullptr, *Y = nullptr, *vl = nullptr;
  double* host = nullptr;              // pinned: 16 doubles
  unsigned long long* t_start = nullptr;   // device: wall-clock stamp taken in front of an evaluation
  double ticks_per_ms = 1e5;           // hipDeviceAttributeWallClockRate (kHz)
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
  bool cp_on = false;                // FOS_PLAN_CLUSTER (opt-in)
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
  double* out5 = nullptr;            // device
  int nupd = 0;                      // workgroups of the update kernel
};

namespace {

void plan_fused(fos_problem* p, const MenuEntry* e, int nwg_hint) {
  p->entry = e;
  p->path = 0;
  int64_t m = p->m;
  // Workgroups per CU (profiles/r02_sweep_wgs.log): one for the wide geometries, whose register tiles already hold
  // 64-96 KiB of rows in flight per CU; two for the 256-thread geometries with 1-2 chunks per thread (1048576 x 1024:
  // 67 % -> 88 % of the roofline - one such workgroup has 32 KiB in flight, below HBM latency x bandwidth per CU);
  // 16 single-wave workgroups for the one-wave-per-row geometries.
  const int per_cu = e->threads >= 512 ? 1 : e->threads == 256 ? (e->k <= 2 ? 2 : 1) : 1024 / e->threads;
  int nwg = nwg_hint > 0 ? nwg_hint : p->ncu * per_cu;
  // at least 2 row steps and 32 KiB of rows per workgroup: below that the slab (one row of n floats per workgroup) and the
  // slab sums rival the rows they cover (20000 x 256: 26 -> 17.5 us per iteration at 32 rows, 4096 x 512 best at 16 rows,
  // 100000 x 128 at 49-98 rows: profiles/r02_sweep_mid.txt)
  const int64_t row_bytes = p->n * (p->dtype == FOS_F32 ? 4 : 2);
  const int64_t min_rows = std::max<int64_t>(2 * (int64_t)e->r, (32768 + row_bytes - 1) / row_bytes);
  if (m < (int64_t)nwg * min_rows) nwg = (int)std::max<int64_t>(1, m / min_rows);
  p->rows_per_wg = (m + nwg - 1) / nwg;
  p->nwg = (int)((m + p->rows_per_wg - 1) / p->rows_per_wg);
  p->nslabs = p->nwg;
}

// Row-per-thread pass: every thread gets at least 4 rows when m allows, at most 4 workgroups per CU.
void plan_tall(fos_problem* p, const MenuEntry* e) {
  p->entry = e;
  p->path = 0;
  p->tall = true;
  p->slab_stride = fos::tall_slab_stride((int)p->n);
  p->vec4 = true;                    // padded slab rows: the float4 epilogues serve ragged n as well
  // 4 workgroups per CU; 8 for the chunk-per-lane form (17-40 VGPRs: 8 workgroups are resident, each with 4 KiB per
  // wave in flight) - profiles/r02_sweep_wgs.log
  const int per_cu = e->k > 0 ? 8 : 4;
  int64_t nwg = std::max<int64_t>(1, std::min<int64_t>(per_cu * (int64_t)p->ncu, p->m / (4 * fos::TL_THREADS)));
  p->rows_per_wg = ((p->m + nwg - 1) / nwg + 3) / 4 * 4;     // a multiple of 4 rows: 16-byte aligned block starts (staged copy)
  p->nwg = (int)((p->m + p->rows_per_wg - 1) / p->rows_per_wg);
  p->nslabs = p->nwg;
}

void plan_fallback(fos_problem* p) {
  p->entry = nullptr;
  p->path = 1;
  int chunks = (int)std::max<int64_t>(1, std::min<int64_t>(64, p->m / 64));
  p->rows_per_wg = (p->m + chunks - 1) / chunks;
  p->nslabs = (int)((p->m + p->rows_per_wg - 1) / p->rows_per_wg);
  p->nwg = p->nslabs;
  p->resid_grid = (int)std::max<int64_t>(1, std::min<int64_t>(2048, (p->m + 3) / 4));
}

int ensure_workspace(fos_problem* p) {
  const int need_slabs = p->nslabs;
  if (need_slabs > p->slab_cap) {
    if (p->slabs) (void)hipFree(p->slabs);
    p->slabs = nullptr;
    p->slab_cap = 0;
    HIP_TRY(hipMalloc(&p->slabs, (size_t)need_slabs * (p->slab_stride ? p->slab_stride : p->n) * sizeof(float)));
    p->slab_cap = need_slabs;
  }
  const int need_rr = std::max(std::max(p->nwg, p->colblock ? 256 : 1), std::max(p->resid_grid, 1));
  if (need_rr > p->rr_cap) {
    if (p->rr_part) (void)hipFree(p->rr_part);
    if (p->rr2_part) (void)hipFree(p->rr2_part);
    p->rr_part = p->rr2_part = nullptr;
    p->rr_cap = 0;
    HIP_TRY(hipMalloc(&p->rr_part, (size_t)need_rr * sizeof(double)));
    HIP_TRY(hipMalloc(&p->rr2_part, (size_t)need_rr * sizeof(double)));
    p->rr_cap = need_rr;
  }
  if (p->path == 1 && p->rvec == nullptr) HIP_TRY(hipMalloc(&p->rvec, (size_t)p->m * sizeof(double)));
  if (p->colblock && p->rneg == nullptr) {
    HIP_TRY(hipMalloc(&p->rneg, (size_t)p->m * sizeof(float)));
    HIP_TRY(hipMalloc(&p->zeros, (size_t)p->cb_width * sizeof(float)));
    HIP_TRY(hipMemset(p->zeros, 0, (size_t)p->cb_width * sizeof(float)));
  }
  return FOS_OK;
}

int prof_drain(fos_problem* p) {
  if (p->ev_used == 0) return FOS_OK;
  HIP_TRY(hipEventSynchronize(p->ev_pool[p->ev_used - 1]));
  for (size_t i = 0; i + 1 < p->ev_used; i += 2) {
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, p->ev_pool[i], p->ev_pool[i + 1]));
    p->prof_ms += ms;
    p->prof_launches += 1;
  }
  p->ev_used = 0;
  return FOS_OK;
}

int prof_mark(fos_problem* p, bool start) {
  if (!p->profiling) return FOS_OK;
  if (start) {
    p->prof_open = (p->prof_seq++ % p->profiling) == 0;
    if (!p->prof_open) return FOS_OK;
  } else if (!p->prof_open) {
    return FOS_OK;
  }
  if (start && p->ev_used + 2 > p->ev_pool.size()) {
    if (p->ev_pool.size() >= 8192) {          // bounded pool: fold what we have (synchronises)
      int rc = prof_drain(p);
      if (rc) return rc;
    } else {
      for (int i = 0; i < 2; ++i) {
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        p->ev_pool.push_back(e);
      }
    }
  }
  HIP_TRY(hipEventRecord(p->ev_pool[p->ev_used++], p->stream));
  return FOS_OK;
}

// Enqueue the A pass for `ys`.  with_g: also produce the slabs (A^T r).  Returns number of rr partials.
int launch_pass_inner(fos_problem* p, const YSource& ys, const float* b, bool with_g, int* n_rr, bool dual);
int launch_pass(fos_problem* p, const YSource& ys, const float* b, bool with_g, int* n_rr, bool dual = false) {
  int rc = prof_mark(p, true);
  if (rc) return rc;
  if ((rc = launch_pass_inner(p, ys, b, with_g, n_rr, dual))) return rc;
  return prof_mark(p, false);
}

int reduce_across(fos_problem* p, void* buf, size_t count, bool f64);

__global__ __launch_bounds__(256) void sumsq_partials_kernel(const float* __restrict__ v, int64_t m, double* __restrict__ part,
                                                            const int* stopped) {
  if (stopped != nullptr && *stopped != 0) return;
  __shared__ double ws[4];
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < m; i += (int64_t)gridDim.x * 256) acc += (double)v[i] * (double)v[i];
  acc = fos::wave_sum(acc);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (ws[0] + ws[1]) + (ws[2] + ws[3]);
}

// v *= scale, only once the solver has stopped (column-sharded runs: keeps the in-place all-reduce of a no-op iteration
// from compounding; the values are not consumed any more, this only keeps them finite)
__global__ __launch_bounds__(256) void unsum_if_stopped_kernel(float* __restrict__ v, int64_t m, float scale, const int* stopped) {
  if (*stopped == 0) return;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < m; i += (int64_t)gridDim.x * 256) v[i] *= scale;
}

// Column blocks (rows wider than any single-pass kernel): phase 1 accumulates the negated residual block by block with the
// residual-only form of the streaming kernel, phase 2 is the SAME with-gradient kernel per block with y = 0 and b = -r
// (its row "dot" is then exactly r_i), writing its columns of full-width slabs.  A is read twice, at streaming speed.
int launch_pass_colblock(fos_problem* p, const YSource& ys, const float* b, bool with_g, int* n_rr) {
  const int64_t W = p->cb_width, esz = p->dtype == FOS_F32 ? 4 : 2;
  const int nblk = (int)((p->n + W - 1) / W);
  const char* Ab = reinterpret_cast<const char*>(p->A);
  for (int cb = 0; cb < nblk; ++cb) {
    const int64_t c0 = cb * W;
    const int nb = (int)std::min<int64_t>(W, p->n - c0);
    YSource yb = ys;
    if (ys.y) yb.y = ys.y + c0;
    if (ys.x_cur) { yb.x_cur = ys.x_cur + c0; yb.x_prev = ys.x_prev + c0; }
    if (ys.yd) yb.yd = ys.yd + c0;
    yb.res_out = p->rneg;
    yb.res_accum = cb > 0;
    // column-sharded: b enters the sum over the ranks once (rank 0)
    const float* b_here = (cb == 0 && !(p->col_sharded && p->comm->rank != 0)) ? b : nullptr;
    p->entry->resid_only_cb(Ab + c0 * esz, p->lda, b_here, p->m, nb, yb, p->rows_per_wg, p->slabs, p->rr2_part,
                            p->rr2_part, p->nwg, p->stream);
    LAUNCH_CHECK();
  }
  if (p->col_sharded) {              // r = sum over the column blocks of ALL ranks: the one m-vector exchange
    if (ys.stopped != nullptr) {     // after a stop the passes above were no-ops and rneg still holds the last SUM: divide
      hipLaunchKernelGGL(unsum_if_stopped_kernel, dim3(grid_1d(p->m, 256, 1024)), dim3(256), 0, p->stream, p->rneg, p->m,   // it back
                         1.0f / (float)p->comm->nranks, ys.stopped);
      LAUNCH_CHECK();
    }
    int rc = reduce_across(p, p->rneg, (size_t)p->m, false);
    if (rc) return rc;
  }
  if (!with_g) {
    hipLaunchKernelGGL(sumsq_partials_kernel, dim3(256), dim3(256), 0, p->stream, p->rneg, p->m, p->rr_part, ys.stopped);
    LAUNCH_CHECK();
    *n_rr = 256;
    return FOS_OK;
  }
  for (int cb = 0; cb < nblk; ++cb) {
    const int64_t c0 = cb * W;
    const int nb = (int)std::min<int64_t>(W, p->n - c0);
    YSource yz{p->zeros, nullptr, nullptr, nullptr, ys.stopped};
    yz.slab_stride = p->n;
    p->entry->with_g_cb(Ab + c0 * esz, p->lda, p->rneg, p->m, nb, yz, p->rows_per_wg, p->slabs + c0, cb == 0 ? p->rr_part : p->rr2_part,
                        p->rr2_part, p->nwg, p->stream);
    LAUNCH_CHECK();
  }
  *n_rr = p->nwg;
  return FOS_OK;
}

int launch_pass_inner(fos_problem* p, const YSource& ys, const float* b, bool with_g, int* n_rr, bool dual) {
  if (p->path == 0 && p->colblock) return launch_pass_colblock(p, ys, b, with_g, n_rr);
  if (p->path == 0) {
    FusedLaunch fn = dual ? p->entry->dual : (with_g ? p->entry->with_g : p->entry->resid_only);
    if (p->il) {
      FusedLaunch fi = dual ? p->entry->dual_il : (with_g ? p->entry->with_g_il : p->entry->resid_only_il);
      if (fi) fn = fi;
    }
    fn(p->A, p->lda, b, p->m, (int)p->n, ys, p->rows_per_wg, p->slabs, p->rr_part, p->rr2_part, p->nwg, p->stream);
    LAUNCH_CHECK();
    *n_rr = p->nwg;
    return FOS_OK;
  }
  if (p->dtype == FOS_F32)
    hipLaunchKernelGGL(fos::residual_rows_kernel<float>, dim3(p->resid_grid), dim3(256), 0, p->stream,
                       (const float*)p->A, p->lda, b, p->m, (int)p->n, ys, p->rvec, p->rr_part);
  else
    hipLaunchKernelGGL(fos::residual_rows_kernel<fos::bf16_t>, dim3(p->resid_grid), dim3(256), 0, p->stream,
                       (const fos::bf16_t*)p->A, p->lda, b, p->m, (int)p->n, ys, p->rvec, p->rr_part);
  LAUNCH_CHECK();
  *n_rr = p->resid_grid;
  if (with_g) {
    dim3 grid((unsigned)((p->n + 255) / 256), (unsigned)p->nslabs);
    if (p->dtype == FOS_F32)
      hipLaunchKernelGGL(fos::transpose_rows_kernel<float>, grid, dim3(256), 0, p->stream, (const float*)p->A, p->lda,
                         p->m, (int)p->n, p->rvec, ys.stopped, p->rows_per_wg, p->slabs);
    else
      hipLaunchKernelGGL(fos::transpose_rows_kernel<fos::bf16_t>, grid, dim3(256), 0, p->stream,
                         (const fos::bf16_t*)p->A, p->lda, p->m, (int)p->n, p->rvec, ys.stopped, p->rows_per_wg,
                         p->slabs);
    LAUNCH_CHECK();
  }
  return FOS_OK;
}

// In-place sum over the ranks of a communicator on `st`: RCCL, or the one-shot full-mesh kernel (comm.hpp).
int comm_allreduce(fos_comm* c, void* buf, size_t count, bool f64, hipStream_t st) {
  if (c->kind == 0) {
    const ncclResult_t r = c->api->AllReduce(buf, buf, count, f64 ? ncclDouble : ncclFloat, ncclSum, c->nccl, st);
    if (r != ncclSuccess) return fail(FOS_ERR_HIP, std::string("ncclAllReduce: ") + c->api->GetErrorString(r));
    return FOS_OK;
  }
  const size_t esz = f64 ? 8 : 4;
  if (count * esz > c->cap_bytes)
    return fail(FOS_ERR_ARG, "mesh all-reduce: message larger than the inbox rows the communicator was created with");
  if (count == 0) return FOS_OK;
  c->seq += 1;
  const int nwg = (int)std::max<size_t>(1, std::min<size_t>(fos::MESH_MAXWG, (count * esz + 4095) / 4096));
  const unsigned long long timeout = 100000000ull * 20ull;      // 20 s of the 100 MHz wall clock
  if (f64)
    hipLaunchKernelGGL(fos::mesh_allreduce_kernel<double>, dim3(nwg), dim3(fos::MESH_THREADS), 0, st, c->peers, c->nranks,
                       c->rank, (double*)buf, (long long)count, c->seq, (long long)(c->cap_bytes / 8), timeout, c->err);
  else
    hipLaunchKernelGGL(fos::mesh_allreduce_kernel<float>, dim3(nwg), dim3(fos::MESH_THREADS), 0, st, c->peers, c->nranks,
                       c->rank, (float*)buf, (long long)count, c->seq, (long long)(c->cap_bytes / 4), timeout, c->err);
  LAUNCH_CHECK();
  return FOS_OK;
}

// Sum `count` floats / doubles over the ranks of a sharded problem, in place, on the handle's stream (no-op otherwise).
int reduce_across(fos_problem* p, void* buf, size_t count, bool f64) {
  if (!p->comm) return FOS_OK;
  return comm_allreduce(p->comm, buf, count, f64, p->stream);
}

__global__ void rr_from_gbuf_kernel(const float* __restrict__ gbuf, int n, double* __restrict__ rr_out, const int* stopped) {
  if (stopped != nullptr && *stopped != 0) return;
  *rr_out = (double)gbuf[n];
}

int launch_slab_reduce_local(fos_problem* p, int n_rr, float* gbuf, double* rr_out, const int* stopped);
// slabs -> gbuf[0..n], summed over the ranks when the problem is sharded; rr_out (nullable) = the global ||r||^2
int launch_slab_reduce(fos_problem* p, int n_rr, float* gbuf, double* rr_out, const int* stopped) {
  if (!p->comm || p->col_sharded)                       // column-sharded: the gradient block is local, ||r||^2 already global
    return launch_slab_reduce_local(p, n_rr, gbuf, rr_out, stopped);
  // Row-sharded: gbuf is all-reduced IN PLACE, and the collective cannot be skipped on the device.  After a device-side
  // stop the A pass is a no-op and the slabs keep the last active iteration's partials, so the local sum is re-derived
  // from them UNCONDITIONALLY: every further enqueued iteration then all-reduces the same partials to the same sums (a
  // guarded no-op would leave the previous SUM in gbuf and the next all-reduce would multiply it by the number of ranks:
  // fp32 overflow after ~43 no-op iterations at 8 ranks).  rr_out is written after the exchange, guarded.
  int rc = launch_slab_reduce_local(p, n_rr, gbuf, nullptr, nullptr);
  if (rc) return rc;
  if ((rc = reduce_across(p, gbuf, (size_t)p->n + 1, false))) return rc;
  if (rr_out != nullptr) {
    hipLaunchKernelGGL(rr_from_gbuf_kernel, dim3(1), dim3(1), 0, p->stream, gbuf, (int)p->n, rr_out, stopped);
    LAUNCH_CHECK();
  }
  return FOS_OK;
}

int launch_slab_reduce_local(fos_problem* p, int n_rr, float* gbuf, double* rr_out, const int* stopped) {
  const int grid = (int)((p->n + fos::RCOLS - 1) / fos::RCOLS);
  if (p->vec4)
    hipLaunchKernelGGL(fos::slab_reduce_kernel<true>, dim3(grid), dim3(256), 0, p->stream, p->slabs, p->nslabs,
                       (int)p->n, p->rr_part, n_rr, gbuf, rr_out, stopped, p->slab_stride);
  else
    hipLaunchKernelGGL(fos::slab_reduce_kernel<false>, dim3(grid), dim3(256), 0, p->stream, p->slabs, p->nslabs,
                       (int)p->n, p->rr_part, n_rr, gbuf, rr_out, stopped);
  LAUNCH_CHECK();
  return FOS_OK;
}

__global__ void xp_pack_kernel(const float* __restrict__ X, int n, int n_pad, int nv, float* __restrict__ xp) {
  for (int col = blockIdx.x * 256 + threadIdx.x; col < n_pad; col += gridDim.x * 256)
    for (int j = 0; j < fos::BT_NV; ++j) xp[fos::xp_index(col, j)] = (col < n && j < nv) ? X[(int64_t)col * fos::BT_NV + j] : 0.f;
}

int ensure_batch_workspace(fos_problem* p) {
  if (p->xp && p->q_part && p->bt_out) return FOS_OK;     // all three or nothing: a failed attempt is retried cleanly
  const int64_t tile_cols = p->dtype == FOS_BF16 ? fos::BQ_COLS : fos::BT_COLS;
  p->n_pad = (p->n + tile_cols - 1) / tile_cols * tile_cols;
  // fp32: Xp, one float per (column, candidate); bf16: Xq, three bf16 terms per (column, candidate)
  const size_t per_entry = p->dtype == FOS_BF16 ? 3 * sizeof(unsigned short) : sizeof(float);
  if (!p->xp) HIP_TRY(hipMalloc(&p->xp, (size_t)p->n_pad * fos::BT_NV * per_entry));
  if (!p->q_part) HIP_TRY(hipMalloc(&p->q_part, (size_t)(3 * p->ncu + 8) * fos::BT_NV * sizeof(double)));
  if (!p->bt_out) HIP_TRY(hipMalloc(&p->bt_out, 128 * sizeof(double)));
  return FOS_OK;
}

// bf16 variants of the batched kernel: (row blocks per wave, tile columns), rows per workgroup, workgroups per CU.
// Measured at 65536 x 8192 (tools/bench_bq.py): <1,128> 208.6 us, <2,64> 213.5 us, <2,128> 166.6 us (80.6 % of HBM),
// <4,64> 170.4 us.  The 128-row tile halves the LDS re-reads of the candidate fragments per byte of A; the 64-row
// tile is kept for short problems, where it gives twice as many workgroups.
typedef void (*Bf16Batch)(const fos::bf16_t*, int64_t, const float*, int, int64_t, int, const unsigned short*, int64_t, double*,
                          float*, const int*);
struct Bf16BatchVariant { Bf16Batch fn, fn_store; int rows; int wg_per_cu; };     // fn_store: also keeps R (gram_batch.hpp)
const Bf16BatchVariant kBf16Batch[] = {
    {fos::residual_batch_mfma_bf16_kernel<1, 128>, fos::residual_batch_mfma_bf16_kernel<1, 128, true>, 64, 2},
    {fos::residual_batch_mfma_bf16_kernel<2, 128>, fos::residual_batch_mfma_bf16_kernel<2, 128, true>, 128, 1},
};

typedef void (*F32Batch)(const float*, int64_t, const float*, int, int64_t, int, const float*, int64_t, double*, float*,
                         const int*);
struct F32BatchVariant { F32Batch fn, fn_store; int rows; int wg_per_cu; };
// fp32, measured at 65536 x 8192: <1> 64-row tile 368-395 us, <2> 128-row tile 335.7 us (80 % of HBM), <4> 336.8 us.
const F32BatchVariant kF32Batch[] = {
    {fos::residual_batch_mfma_kernel<1>, fos::residual_batch_mfma_kernel<1, true>, 64, 3},
    {fos::residual_batch_mfma_kernel<2>, fos::residual_batch_mfma_kernel<2, true>, 128, 2},
};

// Product 1 on `rows` rows starting at A / b: q_part[wg][16] partial squared norms, rout (nullable): the residuals.
// Returns the number of workgroups (rows of q_part).
int launch_batch_product(fos_problem* p, const void* A, const float* b, int64_t rows_total, int use_b, float* rout, int* nwg_out,
                         const int* stopped = nullptr) {
  const bool is_bf16 = p->dtype == FOS_BF16;
  const int variant = rows_total >= 128 * (int64_t)p->ncu ? 1 : 0;
  const int rows = is_bf16 ? kBf16Batch[variant].rows : kF32Batch[variant].rows;
  const int per_cu = is_bf16 ? kBf16Batch[variant].wg_per_cu : kF32Batch[variant].wg_per_cu;
  const int64_t ngroups = (rows_total + rows - 1) / rows;
  int64_t nwg = std::min<int64_t>(ngroups, per_cu * (int64_t)p->ncu);
  const int64_t gpw = (ngroups + nwg - 1) / nwg;
  nwg = (ngroups + gpw - 1) / gpw;
  if (is_bf16)
    hipLaunchKernelGGL(rout ? kBf16Batch[variant].fn_store : kBf16Batch[variant].fn, dim3((unsigned)nwg),
                       dim3(fos::BT_THREADS), 0, p->stream, (const fos::bf16_t*)A, p->lda, b, (use_b && b) ? 1 : 0,
                       rows_total, (int)p->n, (const unsigned short*)p->xp, gpw, p->q_part, rout, stopped);
  else
    hipLaunchKernelGGL(rout ? kF32Batch[variant].fn_store : kF32Batch[variant].fn, dim3((unsigned)nwg),
                       dim3(fos::BT_THREADS), 0, p->stream, (const float*)A, p->lda, b, (use_b && b) ? 1 : 0, rows_total,
                       (int)p->n, p->xp, gpw, p->q_part, rout, stopped);
  LAUNCH_CHECK();
  *nwg_out = (int)nwg;
  return FOS_OK;
}

// q[j] = ||A Xp_j - use_b*b||^2 -> out16 (device); Xp already in p->xp.
// q[j] = sum_i R[i][j]^2 of an m x 16 residual block (column-sharded candidate pass, after the sum over the ranks)
__global__ __launch_bounds__(256) void colnorms16_partials_kernel(const float* __restrict__ R, int64_t m, double* __restrict__ part,
                                                                 const int* stopped) {
  if (stopped != nullptr && *stopped != 0) return;
  __shared__ double ws[16][17];
  const int j = threadIdx.x & 15, sub = threadIdx.x >> 4;          // 16 rows per trip, a 64-byte row of R per 16 lanes
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 16 + sub; i < m; i += (int64_t)gridDim.x * 16) {
    const double v = (double)R[i * fos::BT_NV + j];
    acc += v * v;
  }
  ws[sub][j] = acc;
  __syncthreads();
  if (threadIdx.x < 16) {
    double t = 0.0;
    for (int s = 0; s < 16; ++s) t += ws[s][threadIdx.x];
    part[(int64_t)blockIdx.x * fos::BT_NV + threadIdx.x] = t;
  }
}
__global__ __launch_bounds__(256) void unsum16_if_stopped_kernel(float* __restrict__ v, int64_t count, float scale, const int* stopped) {
  if (*stopped == 0) return;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) v[i] *= scale;
}

int launch_residual_batch(fos_problem* p, int use_b, double* out16, const int* stopped = nullptr) {
  int rc = prof_mark(p, true);
  if (rc) return rc;
  int nwg = 0;
  if (p->col_sharded) {
    // Column-sharded: ||A dlt_j||^2 = ||sum_p A_p dlt_{j,p}||^2.  Every rank's product 1 keeps its partial residuals
    // (m x 16 floats), ONE all-reduce sums the blocks of all 16 candidates (4 MiB at m = 65536), the column norms follow.
    if (!p->rcols16) HIP_TRY(hipMalloc(&p->rcols16, (size_t)p->m * fos::BT_NV * sizeof(float)));
    const float* b_here = p->comm->rank == 0 ? p->b : nullptr;
    if ((rc = launch_batch_product(p, p->A, b_here, p->m, use_b, p->rcols16, &nwg, stopped))) return rc;
    if ((rc = prof_mark(p, false))) return rc;
    if (stopped != nullptr) {        // a no-op product leaves the last SUM in place: divide it back before the in-place all-reduce
      hipLaunchKernelGGL(unsum16_if_stopped_kernel, dim3(grid_1d(p->m * fos::BT_NV, 256, 1024)), dim3(256), 0, p->stream,
                         p->rcols16, p->m * fos::BT_NV, 1.0f / (float)p->comm->nranks, stopped);
      LAUNCH_CHECK();
    }
    if ((rc = reduce_across(p, p->rcols16, (size_t)p->m * fos::BT_NV, false))) return rc;
    const int g = grid_1d(p->m, 16 * 16, 3 * p->ncu);
    hipLaunchKernelGGL(colnorms16_partials_kernel, dim3(g), dim3(256), 0, p->stream, p->rcols16, p->m, p->q_part, stopped);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(fos::fold_partials_kernel, dim3(1), dim3(fos::FOLD_THREADS), 0, p->stream, p->q_part, g, fos::BT_NV, out16);
    LAUNCH_CHECK();
    return FOS_OK;
  }
  if ((rc = launch_batch_product(p, p->A, p->b, p->m, use_b, nullptr, &nwg, stopped))) return rc;
  if ((rc = prof_mark(p, false))) return rc;
  hipLaunchKernelGGL(fos::fold_partials_kernel, dim3(1), dim3(fos::FOLD_THREADS), 0, p->stream, p->q_part, (int)nwg, fos::BT_NV, out16);
  LAUNCH_CHECK();
  return reduce_across(p, out16, fos::BT_NV, true);      // sharded: ||A dlt_j||^2 = sum over the row blocks
}

// ---- multi-lambda lockstep run ----------------------------------------------------------------------------------
typedef void (*MultiLaunch)(const float* A, int64_t lda, const float* b, int64_t m, int n, fos::MultiY ys, int64_t rpw,
                            float* slabs, double* rr_part, int nwg, hipStream_t st);
template <int THREADS, int K, int NVEC>
void multi_launch(const float* A, int64_t lda, const float* b, int64_t m, int n, fos::MultiY ys, int64_t rpw,
                         float* slabs, double* rr_part, int nwg, hipStream_t st) {
  hipLaunchKernelGGL((fos::gemv_multi_kernel<float, THREADS, K, 2, NVEC, 2>), dim3(nwg), dim3(THREADS), 0, st, A, lda, b,
                     m, n, ys, rpw, slabs, rr_part);
}
MultiLaunch find_multi(int64_t n, int nv) {
  static const MultiLaunch small[3] = {multi_launch<256, 4, 2>, multi_launch<256, 4, 3>, multi_launch<256, 4, 4>};
  static const MultiLaunch big[3] = {multi_launch<512, 4, 2>, multi_launch<512, 4, 3>, multi_launch<512, 4, 4>};
  if (nv < 2 || nv > 4) return nullptr;
  if (n <= 4096) return small[nv - 2];
  if (n <= 8192) return big[nv - 2];
  return nullptr;
}

bool batch_supported(const fos_problem* p) { return p->path == 0 && !p->tall; }   // fp32: f32 MFMA; bf16: 3-term bf16 MFMA

// A caller vector the fused prologue can read with 16-byte loads.
int aligned_vec(fos_problem* p, const float* v, const float** out) {
  if ((reinterpret_cast<uintptr_t>(v) & 15u) == 0) {
    *out = v;
    return FOS_OK;
  }
  HIP_TRY(hipMemcpyAsync(p->ybuf, v, (size_t)p->n * sizeof(float), hipMemcpyDeviceToDevice, p->stream));
  *out = p->ybuf;
  return FOS_OK;
}

// Row order of the streaming pass.  Contiguous blocks per workgroup keep 256 streams 8-256 MiB apart; dealing the rows
// round-robin (IL) makes all CUs read ONE contiguous window (256 rows = 16 MiB at 64 KiB rows) that sweeps the matrix.
// profiles/r03_row_order.md (tools/skew_bench, tools/order_probe.py, in-bench A/B): on a quiet device the two are within
// 2 % of each other either way (8 GiB: blocks 1186 us / IL 1208 us; 16 GiB: 2425-2458 / 2394-2417; 32 GiB: 4800-4840 /
// 4777-4823; 64 GiB: 9546 / 9525), but for seconds after ANY large free on the device (this process's or the previous
// one's - the driver releases and clears VRAM in the background) the block form loses 3-6 % while the interleaved form
// does not move: cfg4 in bench.py, three fresh processes each: blocks 9989-9996 us, interleaved 9550-9571 us.  From 12 GiB
// on the interleaved form is never behind, so it is the default there; below, blocks keep their quiet-state edge.
bool il_default(const fos_problem* p) {
  const int64_t bytes = p->m * p->n * (p->dtype == FOS_F32 ? 4 : 2);
  return bytes >= (12ll << 30);
}

// Choose the kernel family for this problem; `flags` (FOS_PLAN_*) switch individual families off (fos_problem_replan).
void apply_plan(fos_problem* p, unsigned flags) {
  const int64_t m = p->m, n = p->n;
  p->plan_flags = flags;
  p->tall = false;
  p->colblock = false;
  p->slab_stride = 0;
  p->vec4 = (n % 4 == 0);
  p->allow_resident = !(flags & FOS_PLAN_NO_RESIDENT);
  p->il = (flags & FOS_PLAN_INTERLEAVE) ? true : (flags & FOS_PLAN_NO_INTERLEAVE) ? false : il_default(p);
  p->resident = fos::resident_fits(m, n) && p->allow_resident && p->comm == nullptr;
  const int epc = epc_of(p->dtype);
  const bool vec_ok = (n % epc == 0) && (p->lda % epc == 0) && ((reinterpret_cast<uintptr_t>(p->A) & 15u) == 0);
  const MenuEntry* e = vec_ok ? default_entry(p->dtype, n) : nullptr;
  if (n <= fos::TL_MAX_N && !(flags & FOS_PLAN_NO_TALL))
    plan_tall(p, tall_entry(p->dtype, n, p->lda, p->A));
  else if (e) plan_fused(p, e, 0);
  else if (vec_ok && p->dtype == FOS_F32 && n <= fos::WD_MAX_N && !(flags & FOS_PLAN_NO_WIDE)) plan_fused(p, &kWideF32, 0);
  else if (vec_ok && !(flags & FOS_PLAN_NO_COLBLOCK)) {
    // column blocks of equal width (a multiple of 64 columns, at most the widest streaming geometry)
    const int64_t cap = 16384;
    const int64_t blocks = (n + cap - 1) / cap;
    p->cb_width = ((n + blocks - 1) / blocks + 63) / 64 * 64;
    const MenuEntry* ce = default_entry(p->dtype, p->cb_width);
    if (ce && ce->with_g_cb && ce->resid_only_cb) {
      plan_fused(p, ce, 0);
      p->colblock = true;
    } else plan_fallback(p);
  } else plan_fallback(p);
}

// ---- fp64-accumulating pass (L-BFGS fg) -----------------------------------------------------------------------------
// Geometry and workspace of fos_gemv_pair_dd, decided on first use: the tall kernels and the resident kernel serve it as
// they are (they accumulate in fp64 anyway), streaming shapes get the ACC = double instantiation of gemv_pair_kernel,
// everything else (ragged / misaligned layouts, rows wider than the dd menu) the fp64 two-pass kernels.
int ensure_dd(fos_problem* p) {
  if (p->slabs_dd || p->resident) return FOS_OK;
  int nslabs = 0;
  int64_t stride = p->n;
  int n_rr = 0;
  if (p->tall) {
    nslabs = p->nwg; stride = p->slab_stride; n_rr = p->nwg;
  } else {
    const DdEntry* e = nullptr;
    if (p->path == 0 && !p->col_sharded)      // column-sharded: the two-pass form, r all-reduced between the passes
      for (const auto& c : kDdMenu)
        if (c.dtype == p->dtype && (int64_t)c.threads * c.k * epc_of(p->dtype) >= p->n) { e = &c; break; }
    if (e) {
      p->dd_entry = e;
      // fp64 form: two workgroups per CU for the 256-thread geometries (they hold 2 waves per SIMD at most 256 VGPRs
      // each); four for the one-chunk geometry (76 VGPRs; 1048576 x 1024: 723 -> 660 us = 81 % of 8 TB/s, tools/dd_bench;
      // the two-chunk geometry is best at two: 524288 x 2048 87.5 % against 82 %)
      int nwg = p->ncu * (e->threads >= 512 ? 1 : e->threads == 256 ? (e->k == 1 ? 4 : 2) : 1024 / e->threads);
      const int64_t row_bytes = p->n * (p->dtype == FOS_F32 ? 4 : 2);
      const int64_t min_rows = std::max<int64_t>(2 * (int64_t)e->r, (65536 + row_bytes - 1) / row_bytes);   // fp64 slabs
      if (p->m < (int64_t)nwg * min_rows) nwg = (int)std::max<int64_t>(1, p->m / min_rows);
      p->dd_rows_per_wg = (p->m + nwg - 1) / nwg;
      p->dd_nwg = (int)((p->m + p->dd_rows_per_wg - 1) / p->dd_rows_per_wg);
      nslabs = p->dd_nwg; n_rr = p->dd_nwg;
    } else {
      const int chunks = (int)std::max<int64_t>(1, std::min<int64_t>(64, p->m / 64));
      p->dd_rows_per_wg = (p->m + chunks - 1) / chunks;
      p->dd_two_pass_chunks = (int)((p->m + p->dd_rows_per_wg - 1) / p->dd_rows_per_wg);
      p->dd_nwg = (int)std::max<int64_t>(1, std::min<int64_t>(2048, (p->m + 3) / 4));      // pass-1 grid
      nslabs = p->dd_two_pass_chunks; n_rr = p->dd_nwg;
      if (p->rvec == nullptr) HIP_TRY(hipMalloc(&p->rvec, (size_t)p->m * sizeof(double)));
    }
  }
  HIP_TRY(hipMalloc(&p->rr_dd, (size_t)std::max(n_rr, 1) * sizeof(double)));
  HIP_TRY(hipMalloc(&p->slabs_dd, (size_t)nslabs * stride * sizeof(double)));
  return FOS_OK;
}

}  // namespace

namespace {
// One launch of the one-read multi-lambda pass (cluster_pass.hpp); cooperative: all members of a cluster must be resident.
template <int CS>
int launch_cluster_pass_cs(fos_problem* p) {
  auto kern = fos::cluster_pass_kernel<CS>;
  static std::atomic<uint64_t> done{0};
  if (raise_dynamic_lds(kern, fos::CP_LDS_BYTES, done)) return fail(FOS_ERR_HIP, "cluster pass: dynamic LDS size refused");
  const float* A = (const float*)p->A;
  int64_t lda = p->lda, m = p->m, rpc = p->cp_rows_per_cluster, n_stride = p->n;
  int n = (int)p->n, n_pad = (int)p->n_pad, xcd_aware = 1;
  const float* b = p->b;
  const float* xp = p->xp;
  unsigned epoch = p->cp_epoch;
  void* args[] = {&A, &lda, &b, &m, &n, &n_pad, &xp, &rpc, &xcd_aware, &p->cp_xchg, &p->cp_flags, &epoch, &p->slabs16, &n_stride,
                  &p->cp_error};
  HIP_TRY(hipLaunchCooperativeKernel(reinterpret_cast<const void*>(kern), dim3((unsigned)(p->cp_clusters * CS)),
                                     dim3(fos::CP_THREADS), args, (unsigned)fos::CP_LDS_BYTES, p->stream));
  p->cp_epoch += (unsigned)(rpc / fos::CP_ROWS) + 16u;
  return FOS_OK;
}
int launch_cluster_pass(fos_problem* p) {
  switch (p->cp_cs) {
    case 4: return launch_cluster_pass_cs<4>(p);
    case 8: return launch_cluster_pass_cs<8>(p);
    case 16: return launch_cluster_pass_cs<16>(p);
  }
  return fail(FOS_ERR_STATE, "cluster pass: no plan");
}

}  // namespace

extern "C" {

const char* fos_last_error(void) { return g_err.c_str(); }
int fos_abi_version(void) { return FOS_ABI_VERSION; }

int fos_problem_create(fos_problem** out, const void* A, int64_t m, int64_t n, int64_t lda, int a_dtype,
                       const float* b, void* stream) {
  if (!out || !A || m <= 0 || n <= 0 || lda < n) return fail(FOS_ERR_ARG, "fos_problem_create: bad shape/pointer");
  if (a_dtype != FOS_F32 && a_dtype != FOS_BF16) return fail(FOS_ERR_ARG, "fos_problem_create: bad a_dtype");
  if (n > (int64_t)1 << 30) return fail(FOS_ERR_ARG, "fos_problem_create: n too large");
  fos_problem* p = new fos_problem();
  p->A = A; p->b = b; p->m = m; p->n = n; p->lda = lda; p->dtype = a_dtype;
  p->stream = reinterpret_cast<hipStream_t>(stream);
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
    delete p;
    return fail(FOS_ERR_HIP, "fos_problem_create: no HIP device");
  }
  p->ncu = prop.multiProcessorCount;
  apply_plan(p, 0);
  int rc = ensure_workspace(p);
  if (rc == FOS_OK) {
    hipError_t he = hipMalloc(&p->gbuf_own, (size_t)(n + 4) * sizeof(float));
    p->gbuf = p->gbuf_own;
    if (he == hipSuccess) he = hipMalloc(&p->ybuf, (size_t)n * sizeof(float));
    if (he == hipSuccess) he = hipMalloc(&p->dscal, 256 * sizeof(double));
    p->part_cap = std::max(1024, (int)((n + fos::RCOLS - 1) / fos::RCOLS)) * 4;
    if (he == hipSuccess) he = hipMalloc(&p->part, (size_t)p->part_cap * sizeof(double));
    if (he != hipSuccess) rc = fail(FOS_ERR_HIP, std::string("fos_problem_create: ") + hipGetErrorString(he));
  }
  if (rc != FOS_OK) {
    fos_problem_destroy(p);
    return rc;
  }
  *out = p;
  return FOS_OK;
}

// ---- communicator (comm.hpp) ------------------------------------------------------------------------------------
int fos_comm_unique_id(char id[128]) {
  if (!id) return fail(FOS_ERR_ARG, "fos_comm_unique_id: null");
  std::string err;
  const fos::RcclApi* api = fos::rccl_api(&err);
  if (!api) return fail(FOS_ERR_UNSUPPORTED, err);
  ncclUniqueId uid;
  const ncclResult_t r = api->GetUniqueId(&uid);
  if (r != ncclSuccess) return fail(FOS_ERR_HIP, std::string("ncclGetUniqueId: ") + api->GetErrorString(r));
  static_assert(sizeof(uid) == 128, "ncclUniqueId is 128 bytes");
  std::memcpy(id, &uid, sizeof(uid));
  return FOS_OK;
}

int fos_comm_create(fos_comm** out, const char id[128], int nranks, int rank) {
  if (!out || !id || nranks < 1 || rank < 0 || rank >= nranks) return fail(FOS_ERR_ARG, "fos_comm_create: bad argument");
  std::string err;
  const fos::RcclApi* api = fos::rccl_api(&err);
  if (!api) return fail(FOS_ERR_UNSUPPORTED, err);
  ncclUniqueId uid;
  std::memcpy(&uid, id, sizeof(uid));
  fos_comm* c = new fos_comm();
  c->nranks = nranks; c->rank = rank; c->api = api;
  const ncclResult_t r = api->CommInitRank(&c->nccl, nranks, uid, rank);
  if (r != ncclSuccess) {
    delete c;
    return fail(FOS_ERR_HIP, std::string("ncclCommInitRank: ") + api->GetErrorString(r));
  }
  *out = c;
  return FOS_OK;
}

int fos_comm_destroy(fos_comm* c) {
  if (!c) return FOS_OK;
  if (c->nccl) (void)c->api->CommDestroy(c->nccl);
  for (int i = 0; i < c->n_opened; ++i) (void)hipIpcCloseMemHandle(c->opened[i]);
  void* bufs[] = {c->inbox, c->flags, c->err};
  for (void* q : bufs)
    if (q) (void)hipFree(q);
  delete c;
  return FOS_OK;
}

// ---- full-mesh transport: local allocation + IPC handles, then the peers' handles -----------------------------------
int fos_comm_mesh_create(fos_comm** out, int nranks, int rank, int64_t cap_bytes, char handles[128]) {
  if (!out || !handles || nranks < 1 || nranks > fos::MESH_MAXRANKS || rank < 0 || rank >= nranks || cap_bytes < 16)
    return fail(FOS_ERR_ARG, "fos_comm_mesh_create: bad argument (1..8 ranks)");
  fos_comm* c = new fos_comm();
  c->kind = 1; c->nranks = nranks; c->rank = rank;
  c->cap_bytes = ((size_t)cap_bytes + 63) & ~(size_t)63;
  const size_t inbox_bytes = 2 * (size_t)nranks * c->cap_bytes;
  const size_t flag_bytes = 2 * (size_t)nranks * fos::MESH_MAXWG * sizeof(unsigned long long);
  // fine-grained (inter-device coherent) memory for everything a PEER reads or writes while kernels run; a runtime that
  // refuses the flag for IPC-shared memory falls back to plain device memory (reported by fos_comm_mesh_info)
  hipError_t e = hipExtMallocWithFlags((void**)&c->inbox, inbox_bytes, hipDeviceMallocFinegrained);
  if (e == hipSuccess) e = hipExtMallocWithFlags((void**)&c->flags, flag_bytes, hipDeviceMallocFinegrained);
  c->fine_grained = (e == hipSuccess);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    if (c->inbox) (void)hipFree(c->inbox);
    c->inbox = nullptr; c->flags = nullptr;
    e = hipMalloc(&c->inbox, inbox_bytes);
    if (e == hipSuccess) e = hipMalloc(&c->flags, flag_bytes);
  }
  if (e == hipSuccess) e = hipMalloc(&c->err, sizeof(int));
  if (e == hipSuccess) e = hipMemset(c->inbox, 0, inbox_bytes);
  if (e == hipSuccess) e = hipMemset(c->flags, 0, flag_bytes);
  if (e == hipSuccess) e = hipMemset(c->err, 0, sizeof(int));
  hipIpcMemHandle_t h0, h1;
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "two IPC handles travel in 128 bytes");
  if (e == hipSuccess) e = hipIpcGetMemHandle(&h0, c->inbox);
  if (e == hipSuccess) e = hipIpcGetMemHandle(&h1, c->flags);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) {
    fos_comm_destroy(c);
    return fail(FOS_ERR_HIP, std::string("fos_comm_mesh_create: ") + hipGetErrorString(e));
  }
  std::memcpy(handles, &h0, 64);
  std::memcpy(handles + 64, &h1, 64);
  c->peers.inbox[rank] = c->inbox;
  c->peers.flags[rank] = c->flags;
  *out = c;
  return FOS_OK;
}

int fos_comm_mesh_connect(fos_comm* c, const char* all_handles) {
  if (!c || c->kind != 1 || !all_handles) return fail(FOS_ERR_ARG, "fos_comm_mesh_connect: bad argument");
  for (int p = 0; p < c->nranks; ++p) {
    if (p == c->rank) continue;
    hipIpcMemHandle_t h0, h1;
    std::memcpy(&h0, all_handles + (size_t)p * 128, 64);
    std::memcpy(&h1, all_handles + (size_t)p * 128 + 64, 64);
    void *a = nullptr, *b = nullptr;
    HIP_TRY(hipIpcOpenMemHandle(&a, h0, hipIpcMemLazyEnablePeerAccess));
    c->opened[c->n_opened++] = a;
    HIP_TRY(hipIpcOpenMemHandle(&b, h1, hipIpcMemLazyEnablePeerAccess));
    c->opened[c->n_opened++] = b;
    c->peers.inbox[p] = (char*)a;
    c->peers.flags[p] = (unsigned long long*)b;
  }
  return FOS_OK;
}

int fos_comm_check(fos_comm* c, void* stream) {
  if (!c) return fail(FOS_ERR_ARG, "fos_comm_check: null");
  if (c->kind != 1) return FOS_OK;
  int bad = 0;
  HIP_TRY(hipMemcpyAsync(&bad, c->err, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  if (bad) return fail(FOS_ERR_STATE, "mesh all-reduce: a peer did not deliver within the time limit; results are invalid");
  return FOS_OK;
}

int fos_comm_mesh_info(const fos_comm* c, int* fine_grained, int64_t* cap_bytes) {
  if (!c || c->kind != 1) return fail(FOS_ERR_ARG, "fos_comm_mesh_info: not a mesh communicator");
  if (fine_grained) *fine_grained = c->fine_grained ? 1 : 0;
  if (cap_bytes) *cap_bytes = (int64_t)c->cap_bytes;
  return FOS_OK;
}

int fos_comm_info(const fos_comm* c, int* nranks, int* rank) {
  if (!c) return fail(FOS_ERR_ARG, "fos_comm_info: null");
  if (nranks) *nranks = c->nranks;
  if (rank) *rank = c->rank;
  return FOS_OK;
}

const char* fos_comm_transport(void) {
  static std::string text;
  std::string err;
  const fos::RcclApi* api = fos::rccl_api(&err);
  text = api ? ("rccl: " + api->origin) : ("none: " + err);
  return text.c_str();
}

int fos_comm_allreduce(fos_comm* c, void* buf, int64_t count, int is_f64, void* stream) {
  if (!c || !buf || count < 0) return fail(FOS_ERR_ARG, "fos_comm_allreduce: bad argument");
  return comm_allreduce(c, buf, (size_t)count, is_f64 != 0, (hipStream_t)stream);
}

int fos_problem_set_comm(fos_problem* p, fos_comm* c) {
  if (!p) return fail(FOS_ERR_ARG, "fos_problem_set_comm: null");
  p->comm = c;
  if (c) p->resident = false;        // the one-workgroup resident loop has no exchange step
  else p->resident = fos::resident_fits(p->m, p->n) && p->allow_resident;
  return FOS_OK;
}

int fos_problem_set_comm_cols(fos_problem* p, fos_comm* c) {
  if (!p || !c) return fail(FOS_ERR_ARG, "fos_problem_set_comm_cols: null");
  const int epc = epc_of(p->dtype);
  const bool vec_ok = (p->n % epc == 0) && (p->lda % epc == 0) && ((reinterpret_cast<uintptr_t>(p->A) & 15u) == 0);
  if (!vec_ok || p->n <= fos::TL_MAX_N)
    return fail(FOS_ERR_UNSUPPORTED, "fos_problem_set_comm_cols: needs the streaming layout (aligned, n > 64 per rank)");
  // the two-phase column-block plan, whatever the width: r = sum_p A_p y_p - b is exchanged between the phases
  void* drop[] = {p->slabs, p->rr_part, p->rr2_part};
  for (void* q : drop)
    if (q) (void)hipFree(q);
  p->slabs = nullptr; p->rr_part = p->rr2_part = nullptr;
  p->slab_cap = p->rr_cap = 0;
  p->tall = false; p->slab_stride = 0; p->vec4 = true; p->resident = false;
  const int64_t blocks = (p->n + 16383) / 16384;
  p->cb_width = ((p->n + blocks - 1) / blocks + 63) / 64 * 64;
  const MenuEntry* ce = default_entry(p->dtype, p->cb_width);
  if (!ce || !ce->with_g_cb || !ce->resid_only_cb)
    return fail(FOS_ERR_UNSUPPORTED, "fos_problem_set_comm_cols: no column-block kernel for this block width");
  plan_fused(p, ce, 0);
  p->colblock = true;
  p->comm = c;
  p->col_sharded = true;
  return ensure_workspace(p);
}

int fos_problem_set_stream(fos_problem* p, void* stream) {
  if (!p) return fail(FOS_ERR_ARG, "fos_problem_set_stream: null");
  hipStream_t ns = reinterpret_cast<hipStream_t>(stream);
  if (ns == p->stream) return FOS_OK;
  // work already enqueued on the old stream touches the handle's workspace: the new stream waits for it
  hipEvent_t ev;
  HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  hipError_t e = hipEventRecord(ev, p->stream);
  if (e == hipSuccess) e = hipStreamWaitEvent(ns, ev, 0);
  (void)hipEventDestroy(ev);
  if (e != hipSuccess) return fail(FOS_ERR_HIP, std::string("fos_problem_set_stream: ") + hipGetErrorString(e));
  p->stream = ns;
  return FOS_OK;
}

int fos_problem_replan(fos_problem* p, unsigned flags) {
  if (!p) return fail(FOS_ERR_ARG, "fos_problem_replan: null");
  if (p->col_sharded) return fail(FOS_ERR_UNSUPPORTED, "fos_problem_replan: a column-sharded problem keeps its two-phase plan");
  if (flags & ~(unsigned)(FOS_PLAN_NO_RESIDENT | FOS_PLAN_NO_TALL | FOS_PLAN_NO_WIDE | FOS_PLAN_NO_COLBLOCK | FOS_PLAN_CLUSTER |
                           FOS_PLAN_INTERLEAVE | FOS_PLAN_NO_INTERLEAVE))
    return fail(FOS_ERR_ARG, "fos_problem_replan: unknown flag");
  // the multi-lambda workspace follows its own plan (one-read cluster form or two products): rebuilt on first use
  {
    void* multi[] = {p->rbuf16, p->slabs16, p->cp_xchg, p->cp_flags, p->cp_error};
    for (void* q : multi)
      if (q) (void)hipFree(q);
    p->rbuf16 = p->slabs16 = p->cp_xchg = nullptr; p->cp_flags = nullptr; p->cp_error = nullptr;
    p->cp_cs = p->cp_clusters = 0;
    p->cp_on = (flags & FOS_PLAN_CLUSTER) != 0;
    flags &= ~(unsigned)FOS_PLAN_CLUSTER;
  }
  // workspace sized for the old plan (slab stride, fp64 slabs) is dropped and rebuilt
  void* drop[] = {p->slabs, p->rr_part, p->rr2_part, p->slabs_dd, p->rr_dd};
  for (void* q : drop)
    if (q) (void)hipFree(q);
  p->slabs = nullptr; p->rr_part = p->rr2_part = nullptr; p->slabs_dd = nullptr; p->rr_dd = nullptr;
  p->slab_cap = p->rr_cap = 0;
  p->dd_entry = nullptr; p->dd_two_pass_chunks = 0;
  apply_plan(p, flags);
  return ensure_workspace(p);
}

int fos_problem_profile(fos_problem* p, int enable) {
  if (!p) return fail(FOS_ERR_ARG, "fos_problem_profile: null");
  p->profiling = enable > 0 ? enable : 0;
  p->prof_seq = 0;
  p->prof_open = false;
  return FOS_OK;
}

int fos_problem_profile_read(fos_problem* p, double* ms_total, int64_t* launches) {
  if (!p || !ms_total || !launches) return fail(FOS_ERR_ARG, "fos_problem_profile_read: null");
  int rc = prof_drain(p);
  if (rc) return rc;
  *ms_total = p->prof_ms;
  *launches = p->prof_launches;
  p->prof_ms = 0.0;
  p->prof_launches = 0;
  return FOS_OK;
}

int fos_problem_destroy(fos_problem* p) {
  if (!p) return FOS_OK;
  for (hipEvent_t e : p->ev_pool) (void)hipEventDestroy(e);
  void* bufs[] = {p->slabs, p->rr_part, p->rr2_part, p->rvec, p->gbuf_own, p->ybuf, p->dscal, p->part, p->xp, p->q_part, p->bt_out,
                  p->slabs_dd, p->rr_dd, p->lhist, p->rbuf16, p->rcols16, p->slabs16, p->rneg, p->zeros, p->cp_xchg, p->cp_flags,
                  p->cp_error};
  for (void* q : bufs)
    if (q) (void)hipFree(q);
  delete p->lbfgs;
  delete p;
  return FOS_OK;
}

int fos_problem_plan(const fos_problem* p, int32_t plan[8]) {
  if (!p || !plan) return fail(FOS_ERR_ARG, "fos_problem_plan: null");
  plan[0] = p->path;
  plan[1] = p->entry ? p->entry->threads : 256;
  plan[2] = p->entry ? p->entry->k : 0;
  plan[3] = p->entry ? p->entry->r : 0;
  plan[4] = p->nwg;
  plan[5] = p->nslabs;
  plan[6] = (p->path == 0 ? 1 : 0) | (p->resident ? 2 : 0) | (p->tall ? 4 : 0) | (p->colblock ? 8 : 0) | (p->cp_cs ? 16 : 0) |
            ((p->il && p->entry && p->entry->with_g_il && p->path == 0 && !p->colblock && !p->tall) ? 32 : 0);
  plan[7] = p->ncu;
  return FOS_OK;
}

int fos_problem_tune(fos_problem* p, int threads, int chunks, int rows, int workgroups) {
  if (!p) return fail(FOS_ERR_ARG, "fos_problem_tune: null");
  if (p->path == 0 && p->tall && workgroups > 0) {        // row-per-thread pass: only the workgroup count is tunable
    p->rows_per_wg = ((p->m + workgroups - 1) / workgroups + 3) / 4 * 4;
    p->nwg = (int)((p->m + p->rows_per_wg - 1) / p->rows_per_wg);
    p->nslabs = p->nwg;
    return ensure_workspace(p);
  }
  if (p->path != 0 || p->tall || p->colblock)
    return fail(FOS_ERR_UNSUPPORTED, "fos_problem_tune: only the streaming single-pass kernel has a geometry menu");
  const MenuEntry* e = find_entry(p->dtype, threads, chunks, rows);
  if (!e || (int64_t)e->threads * e->k * epc_of(p->dtype) < p->n)
    return fail(FOS_ERR_UNSUPPORTED, "fos_problem_tune: geometry not instantiated or too narrow for n");
  plan_fused(p, e, workgroups);
  return ensure_workspace(p);
}

int fos_problem_set_gbuf(fos_problem* p, float* gbuf) {
  if (!p) return fail(FOS_ERR_ARG, "fos_problem_set_gbuf: null");
  if (gbuf && (reinterpret_cast<uintptr_t>(gbuf) & 15u)) return fail(FOS_ERR_ARG, "fos_problem_set_gbuf: misaligned");
  p->gbuf = gbuf ? gbuf : p->gbuf_own;
  return FOS_OK;
}

int fos_gemv_pair(fos_problem* p, const float* y, float alpha2, float* grad, double* rr_out) {
  if (!p || !y || !grad) return fail(FOS_ERR_ARG, "fos_gemv_pair: null");
  const float* ya = nullptr;
  int rc = aligned_vec(p, y, &ya);
  if (rc) return rc;
  YSource ys{ya, nullptr, nullptr, nullptr, nullptr};
  int n_rr = 0;
  if ((rc = launch_pass(p, ys, p->b, true, &n_rr))) return rc;
  if ((rc = launch_slab_reduce(p, n_rr, p->gbuf, rr_out, nullptr))) return rc;
  hipLaunchKernelGGL(fos::add_l2_kernel<float>, dim3(grid_1d(p->n, 256, 1024)), dim3(256), 0, p->stream, p->gbuf,
                     (double)alpha2, ya, grad, p->n);
  LAUNCH_CHECK();
  return FOS_OK;
}

int fos_gemv_pair_f64(fos_problem* p, const double* y, double alpha2, float* grad, double* rr_out) {
  if (!p || !y || !grad) return fail(FOS_ERR_ARG, "fos_gemv_pair_f64: null");
  if (p->resident) {                           // small problem: one launch, fp64 throughout (resident.hpp)
    if (p->dtype == FOS_F32)
      hipLaunchKernelGGL(fos::gemv_pair_resident_kernel<float>, dim3(1), dim3(fos::RS_THREADS), 0, p->stream,
                         (const float*)p->A, p->lda, p->b, (int)p->m, (int)p->n, y, alpha2, grad, rr_out);
    else
      hipLaunchKernelGGL(fos::gemv_pair_resident_kernel<fos::bf16_t>, dim3(1), dim3(fos::RS_THREADS), 0, p->stream,
                         (const fos::bf16_t*)p->A, p->lda, p->b, (int)p->m, (int)p->n, y, alpha2, grad, rr_out);
    LAUNCH_CHECK();
    return FOS_OK;
  }
  YSource ys{nullptr, nullptr, nullptr, nullptr, nullptr, 0.0, y};
  int n_rr = 0, rc;
  if ((rc = launch_pass(p, ys, p->b, true, &n_rr))) return rc;
  if ((rc = launch_slab_reduce(p, n_rr, p->gbuf, rr_out, nullptr))) return rc;
  hipLaunchKernelGGL(fos::add_l2_kernel<double>, dim3(grid_1d(p->n, 256, 1024)), dim3(256), 0, p->stream, p->gbuf, alpha2,
                     y, grad, p->n);
  LAUNCH_CHECK();
  return FOS_OK;
}

// The fp64-accumulating pass for any y source: out[0..n) = A^T (A y - b) + alpha2*l2vec (l2vec may be NULL when alpha2 = 0),
// out[n] = ||A y - b||^2, summed over the ranks of a sharded problem.  Not for resident-planned problems (callers check).
__global__ __launch_bounds__(256) void sumsq_f64_partials_kernel(const double* __restrict__ v, int64_t m, double* __restrict__ part,
                                                                const int* stopped) {
  if (stopped != nullptr && *stopped != 0) return;
  __shared__ double ws[4];
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < m; i += (int64_t)gridDim.x * 256) acc += v[i] * v[i];
  acc = fos::wave_sum(acc);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (ws[0] + ws[1]) + (ws[2] + ws[3]);
}
__global__ __launch_bounds__(256) void unsum_f64_if_stopped_kernel(double* __restrict__ v, int64_t m, double scale, const int* stopped) {
  if (*stopped == 0) return;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < m; i += (int64_t)gridDim.x * 256) v[i] *= scale;
}

// Column-sharded form (fos_problem_set_comm_cols): this rank holds A[:, its columns] and its block of y.  Pass 1: the
// partial residual A_p y_p (rank 0 also subtracts b) in fp64, ONE all-reduce of the m doubles, pass 2: g_p = A_p^T r for
// the local block; alpha2*y_p is local.  out[0..n) is this rank's block of the gradient, out[n] the global ||r||^2.
static int launch_pass_dd_cols(fos_problem* p, const YSource& ys, double alpha2, const double* l2vec, double* out) {
  const float* b_here = p->comm->rank == 0 ? p->b : nullptr;
  const dim3 grid((unsigned)((p->n + 255) / 256), (unsigned)p->dd_two_pass_chunks);
  if (p->dtype == FOS_F32)
    hipLaunchKernelGGL(fos::residual_rows_kernel<float>, dim3(p->dd_nwg), dim3(256), 0, p->stream, (const float*)p->A, p->lda,
                       b_here, p->m, (int)p->n, ys, p->rvec, p->rr_dd);
  else
    hipLaunchKernelGGL(fos::residual_rows_kernel<fos::bf16_t>, dim3(p->dd_nwg), dim3(256), 0, p->stream,
                       (const fos::bf16_t*)p->A, p->lda, b_here, p->m, (int)p->n, ys, p->rvec, p->rr_dd);
  LAUNCH_CHECK();
  if (ys.stopped != nullptr) {       // after a stop pass 1 was a no-op and rvec still holds the last SUM: divide it back
    hipLaunchKernelGGL(unsum_f64_if_stopped_kernel, dim3(grid_1d(p->m, 256, 1024)), dim3(256), 0, p->stream, p->rvec, p->m,
                       1.0 / (double)p->comm->nranks, ys.stopped);
    LAUNCH_CHECK();
  }
  int rc = reduce_across(p, p->rvec, (size_t)p->m, true);
  if (rc) return rc;
  const int n_rr = std::min(p->dd_nwg, 256);
  hipLaunchKernelGGL(sumsq_f64_partials_kernel, dim3(n_rr), dim3(256), 0, p->stream, p->rvec, p->m, p->rr_dd, ys.stopped);
  LAUNCH_CHECK();
  if (p->dtype == FOS_F32)
    hipLaunchKernelGGL((fos::transpose_rows_kernel<float, double>), grid, dim3(256), 0, p->stream, (const float*)p->A, p->lda,
                       p->m, (int)p->n, p->rvec, ys.stopped, p->dd_rows_per_wg, p->slabs_dd);
  else
    hipLaunchKernelGGL((fos::transpose_rows_kernel<fos::bf16_t, double>), grid, dim3(256), 0, p->stream,
                       (const fos::bf16_t*)p->A, p->lda, p->m, (int)p->n, p->rvec, ys.stopped, p->dd_rows_per_wg, p->slabs_dd);
  LAUNCH_CHECK();
  hipLaunchKernelGGL(fos::slab_reduce_dd_kernel, dim3((unsigned)((p->n + fos::SRD_COLS - 1) / fos::SRD_COLS)),
                     dim3(fos::SRD_THREADS), 0, p->stream, p->slabs_dd, p->dd_two_pass_chunks, (int)p->n, (int64_t)p->n, p->rr_dd,
                     n_rr, alpha2, l2vec, out, ys.stopped);
  LAUNCH_CHECK();
  return FOS_OK;
}

static int launch_pass_dd(fos_problem* p, const YSource& ys, double alpha2, const double* l2vec, double* out) {
  int rc = ensure_dd(p);
  if (rc) return rc;
  if (p->col_sharded) {
    if ((rc = prof_mark(p, true))) return rc;
    if ((rc = launch_pass_dd_cols(p, ys, alpha2, l2vec, out))) return rc;
    return prof_mark(p, false);
  }
  if (rc) return rc;
  int nslabs = 0, n_rr = 0;
  int64_t stride = p->n;
  if ((rc = prof_mark(p, true))) return rc;
  if (p->tall) {
    p->entry->dd(p->A, p->lda, p->b, p->m, (int)p->n, ys, p->rows_per_wg, p->slabs_dd, p->rr_dd, p->nwg, p->stream);
    nslabs = n_rr = p->nwg;
    stride = p->slab_stride;
  } else if (p->dd_entry) {
    (p->il && p->dd_entry->fn_il ? p->dd_entry->fn_il : p->dd_entry->fn)(p->A, p->lda, p->b, p->m, (int)p->n, ys, p->dd_rows_per_wg, p->slabs_dd, p->rr_dd, p->dd_nwg, p->stream);
    nslabs = n_rr = p->dd_nwg;
  } else {
    dim3 grid((unsigned)((p->n + 255) / 256), (unsigned)p->dd_two_pass_chunks);
    if (p->dtype == FOS_F32) {
      hipLaunchKernelGGL(fos::residual_rows_kernel<float>, dim3(p->dd_nwg), dim3(256), 0, p->stream, (const float*)p->A,
                         p->lda, p->b, p->m, (int)p->n, ys, p->rvec, p->rr_dd);
      hipLaunchKernelGGL((fos::transpose_rows_kernel<float, double>), grid, dim3(256), 0, p->stream, (const float*)p->A,
                         p->lda, p->m, (int)p->n, p->rvec, ys.stopped, p->dd_rows_per_wg, p->slabs_dd);
    } else {
      hipLaunchKernelGGL(fos::residual_rows_kernel<fos::bf16_t>, dim3(p->dd_nwg), dim3(256), 0, p->stream,
                         (const fos::bf16_t*)p->A, p->lda, p->b, p->m, (int)p->n, ys, p->rvec, p->rr_dd);
      hipLaunchKernelGGL((fos::transpose_rows_kernel<fos::bf16_t, double>), grid, dim3(256), 0, p->stream,
                         (const fos::bf16_t*)p->A, p->lda, p->m, (int)p->n, p->rvec, ys.stopped, p->dd_rows_per_wg,
                         p->slabs_dd);
    }
    nslabs = p->dd_two_pass_chunks;
    n_rr = p->dd_nwg;
  }
  LAUNCH_CHECK();
  if ((rc = prof_mark(p, false))) return rc;
  // sharded: alpha2*x enters the sum over the ranks exactly once (rank 0 adds it to its partial)
  const double a2_here = (p->comm && p->comm->rank != 0) ? 0.0 : alpha2;
  hipLaunchKernelGGL(fos::slab_reduce_dd_kernel, dim3((unsigned)((p->n + fos::SRD_COLS - 1) / fos::SRD_COLS)),
                     dim3(fos::SRD_THREADS), 0, p->stream, p->slabs_dd,
                     nslabs, (int)p->n, stride, p->rr_dd, n_rr, a2_here, l2vec, out, p->comm ? nullptr : ys.stopped);
  LAUNCH_CHECK();
  return reduce_across(p, out, (size_t)p->n + 1, true);      // (sharded: re-derived after a stop, see launch_slab_reduce)
}

int fos_gemv_pair_dd(fos_problem* p, const double* x, double alpha2, double* grad_rr) {
  if (!p || !x || !grad_rr) return fail(FOS_ERR_ARG, "fos_gemv_pair_dd: null");
  if (p->resident) {                           // small problem: one launch, fp64 throughout (resident.hpp)
    if (p->dtype == FOS_F32)
      hipLaunchKernelGGL((fos::gemv_pair_resident_kernel<float, double>), dim3(1), dim3(fos::RS_THREADS), 0, p->stream,
                         (const float*)p->A, p->lda, p->b, (int)p->m, (int)p->n, x, alpha2, grad_rr, grad_rr + p->n);
    else
      hipLaunchKernelGGL((fos::gemv_pair_resident_kernel<fos::bf16_t, double>), dim3(1), dim3(fos::RS_THREADS), 0,
                         p->stream, (const fos::bf16_t*)p->A, p->lda, p->b, (int)p->m, (int)p->n, x, alpha2, grad_rr,
                         grad_rr + p->n);
    LAUNCH_CHECK();
    return FOS_OK;
  }
  YSource ys{nullptr, nullptr, nullptr, nullptr, nullptr, 0.0, x};
  return launch_pass_dd(p, ys, alpha2, x, grad_rr);
}

int fos_residual_objective(fos_problem* p, const float* x, double* out3) {
  if (!p || !x || !out3) return fail(FOS_ERR_ARG, "fos_residual_objective: null");
  const float* xa = nullptr;
  int rc = aligned_vec(p, x, &xa);
  if (rc) return rc;
  YSource ys{xa, nullptr, nullptr, nullptr, nullptr};
  int n_rr = 0;
  if ((rc = launch_pass(p, ys, p->b, false, &n_rr))) return rc;
  hipLaunchKernelGGL(fos::fold_partials_kernel, dim3(1), dim3(fos::FOLD_THREADS), 0, p->stream, p->rr_part, n_rr, 1, out3);
  LAUNCH_CHECK();
  if (!p->col_sharded && (rc = reduce_across(p, out3, 1, true))) return rc;
  hipLaunchKernelGGL(fos::vec_norms_kernel, dim3(1), dim3(fos::LB_THREADS), 0, p->stream, xa, p->n, out3 + 1);
  LAUNCH_CHECK();
  if (p->col_sharded) return reduce_across(p, out3 + 1, 2, true);     // ||x||^2, ||x||_1 over the column blocks
  return FOS_OK;
}

int fos_residual_batch(fos_problem* p, const float* X, int nv, int use_b, double* out16) {
  if (!p || !X || !out16 || nv < 1 || nv > fos::BT_NV) return fail(FOS_ERR_ARG, "fos_residual_batch: bad argument");
  if (!batch_supported(p)) return fail(FOS_ERR_UNSUPPORTED, "fos_residual_batch: needs the fused path");
  int rc = ensure_batch_workspace(p);
  if (rc) return rc;
  if (p->dtype == FOS_BF16)
    hipLaunchKernelGGL(fos::xq_pack_kernel, dim3(grid_1d(p->n_pad, 256, 256)), dim3(256), 0, p->stream, X, (int)p->n,
                       (int)p->n_pad, nv, (unsigned short*)p->xp);
  else
    hipLaunchKernelGGL(xp_pack_kernel, dim3(grid_1d(p->n_pad, 256, 256)), dim3(256), 0, p->stream, X, (int)p->n,
                       (int)p->n_pad, nv, p->xp);
  LAUNCH_CHECK();
  return launch_residual_batch(p, use_b, out16);
}

int fos_power_iter(fos_problem* p, float* v_inout, int n_iter, double tol, double* L_out, int* iters_out) {
  if (!p || !v_inout || !L_out || n_iter <= 0) return fail(FOS_ERR_ARG, "fos_power_iter: bad argument");
  if (p->col_sharded)
    return fail(FOS_ERR_UNSUPPORTED, "fos_power_iter: column-sharded problems normalise over the ranks (see _lipschitz_cols)");
  if (n_iter + 1 > p->lhist_cap) {             // L after every step (+ the norm of v0): sized by the caller's n_iter
    if (p->lhist) (void)hipFree(p->lhist);
    p->lhist = nullptr;
    p->lhist_cap = 0;
    HIP_TRY(hipMalloc(&p->lhist, (size_t)(n_iter + 1) * sizeof(double)));
    p->lhist_cap = n_iter + 1;
  }
  double* Lh = p->lhist;        // n_iter + 1 slots
  if (p->resident) {
    // all iterations in one launch; the break rule (:57) is evaluated on the device
    int* used_dev = reinterpret_cast<int*>(p->dscal + 240);
    if (p->dtype == FOS_F32)
      hipLaunchKernelGGL(fos::power_resident_kernel<float>, dim3(1), dim3(fos::RS_THREADS), 0, p->stream,
                         (const float*)p->A, p->lda, (int)p->m, (int)p->n, v_inout, n_iter, tol, Lh, used_dev);
    else
      hipLaunchKernelGGL(fos::power_resident_kernel<fos::bf16_t>, dim3(1), dim3(fos::RS_THREADS), 0, p->stream,
                         (const fos::bf16_t*)p->A, p->lda, (int)p->m, (int)p->n, v_inout, n_iter, tol, Lh, used_dev);
    LAUNCH_CHECK();
    std::vector<double> hL(n_iter);
    int used = 0;
    HIP_TRY(hipMemcpyAsync(&used, used_dev, sizeof(int), hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipMemcpyAsync(hL.data(), Lh, (size_t)n_iter * sizeof(double), hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    if (used < 1 || used > n_iter) return fail(FOS_ERR_HIP, "fos_power_iter: resident kernel returned no result");
    *L_out = hL[used - 1];
    if (iters_out) *iters_out = used;
    return FOS_OK;
  }
  float* v = p->ybuf;
  // v = v0 / ||v0||   (iterative_solvers.py:51)
  hipLaunchKernelGGL(fos::power_normalize_kernel, dim3(1), dim3(1024), 0, p->stream, v_inout, (int)p->n, v, Lh + n_iter);
  LAUNCH_CHECK();
  // The reference breaks as soon as |L - prev| < tol (:57).  The iterations are enqueued in chunks; after each chunk
  // the L values are read back and the break rule is replayed on the host, so a matrix with a dominant eigenvalue
  // stops after a chunk instead of running all n_iter passes (the answer is the same either way).
  std::vector<double> hL(n_iter);
  const int chunk = 16;
  double prev = 0.0;
  int used = n_iter, done = 0;
  bool hit = false;
  while (done < n_iter && !hit) {
    const int todo = std::min(chunk, n_iter - done);
    for (int it = done; it < done + todo; ++it) {
      YSource ys{v, nullptr, nullptr, nullptr, nullptr};
      int n_rr = 0, rc;
      if ((rc = launch_pass(p, ys, nullptr, true, &n_rr))) return rc;                       // w = A^T (A v)   :54
      if ((rc = launch_slab_reduce(p, n_rr, p->gbuf, nullptr, nullptr))) return rc;
      hipLaunchKernelGGL(fos::power_normalize_kernel, dim3(1), dim3(1024), 0, p->stream, p->gbuf, (int)p->n, v,
                         Lh + it);                                                         // L = ||w||, v = w/L :55-56
      LAUNCH_CHECK();
    }
    HIP_TRY(hipMemcpyAsync(hL.data() + done, Lh + done, (size_t)todo * sizeof(double), hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    for (int it = done; it < done + todo; ++it) {      // |L - prev| < tol -> break   :57
      if (std::fabs(hL[it] - prev) < tol) { used = it + 1; hit = true; break; }
      prev = hL[it];
    }
    done += todo;
  }
  HIP_TRY(hipMemcpyAsync(v_inout, v, (size_t)p->n * sizeof(float), hipMemcpyDeviceToDevice, p->stream));
  HIP_TRY(hipStreamSynchronize(p->stream));
  *L_out = hL[used - 1];
  if (iters_out) *iters_out = used;
  return FOS_OK;
}

int fos_prox_l1(const float* v, float thr, float* out, int64_t n, void* stream) {
  if (n == 0) return FOS_OK;
  if (!v || !out || n < 0) return fail(FOS_ERR_ARG, "fos_prox_l1: bad argument");
  hipLaunchKernelGGL(fos::prox_l1_kernel, dim3(grid_1d(n, 256, 2048)), dim3(256), 0, (hipStream_t)stream, v, thr, out, n);
  LAUNCH_CHECK();
  return FOS_OK;
}

int fos_prox_l1_vec(const float* v, const float* thr, float* out, int64_t n, void* stream) {
  if (n == 0) return FOS_OK;
  if (!v || !thr || !out || n < 0) return fail(FOS_ERR_ARG, "fos_prox_l1_vec: bad argument");
  hipLaunchKernelGGL(fos::prox_l1_vec_kernel, dim3(grid_1d(n, 256, 2048)), dim3(256), 0, (hipStream_t)stream, v, thr, out, n);
  LAUNCH_CHECK();
  return FOS_OK;
}

int fos_prox_elastic_net(const float* v, float tau, float alpha1, float alpha2, float* out, int64_t n, void* stream) {
  if (n == 0) return FOS_OK;
  if (!v || !out || n < 0) return fail(FOS_ERR_ARG, "fos_prox_elastic_net: bad argument");
  hipLaunchKernelGGL(fos::prox_enet_kernel, dim3(grid_1d(n, 256, 2048)), dim3(256), 0, (hipStream_t)stream, v, tau,
                     alpha1, alpha2, out, n);
  LAUNCH_CHECK();
  return FOS_OK;
}

int fos_prox_elastic_net_vec(const float* v, const float* tau, float alpha1, float alpha2, float* out, int64_t n,
                             void* stream) {
  if (n == 0) return FOS_OK;
  if (!v || !tau || !out || n < 0) return fail(FOS_ERR_ARG, "fos_prox_elastic_net_vec: bad argument");
  hipLaunchKernelGGL(fos::prox_enet_vec_kernel, dim3(grid_1d(n, 256, 2048)), dim3(256), 0, (hipStream_t)stream, v, tau,
                     alpha1, alpha2, out, n);
  LAUNCH_CHECK();
  return FOS_OK;
}

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
  void* bufs[] = {f->x_cur, f->x_prev, f->dlt, f->scal, f->out5, f->part2, f->ynext, f->gbuf64, f->folded};
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
  if (on && !f->gbuf64) HIP_TRY(hipMalloc(&f->gbuf64, (size_t)(f->p->n + 4) * sizeof(double)));
  f->precise = on != 0;
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

// Up to 16 state machines in lockstep on the matrix cores (gram_batch.hpp): per iteration and per row panel, product 1
// (R = A_panel Y - b, from HBM) and product 2 (G += R^T A_panel, the panel again from the Infinity Cache), then one
// update kernel per state machine, which leaves its y_{k+1} in the candidate block of the next product 1.
static int run_multi_mfma(fos_fista* const* fs, int nv, int iters, bool controlled = false) {
  fos_problem* p = fs[0]->p;
  int rc = ensure_batch_workspace(p);
  if (rc) return rc;
  const bool is_bf16 = p->dtype == FOS_BF16;
  const int64_t esz = is_bf16 ? 2 : 4;
  // One-read form (cluster_pass.hpp), on request (FOS_PLAN_CLUSTER): fp32, 2049..16384 columns in strips of 1024 -> 4, 8
  // or 16 members per cluster, all CUs busy, at least 8 panels per cluster.  Anything else takes the two-product form
  // below, which is also the default: the hand-off chain of the cluster form is exposed under HBM load.
  if (!p->rbuf16 && p->cp_on && !is_bf16 && p->ncu % 8 == 0) {
    const int cs_need = (int)((p->n + fos::CP_W - 1) / fos::CP_W);
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
  if (!p->rbuf16) {
    // Panel: product 1 gives a workgroup 64-128 whole rows, so it needs >= 128 * CUs * 2 rows to fill the chip; row
    // splits of product 2: enough (strip, split) workgroups for two per CU.  (A panel that fits the Infinity Cache
    // - ~3000 rows at n = 8192 - would need a split-K product 1; see DESIGN.md "Multi-lambda".)
    const int64_t rows = 256 * (int64_t)p->ncu;
    p->panel_rows = std::min<int64_t>(rows, (p->m + 255) / 256 * 256);
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
      if ((rc = launch_cluster_pass(p))) return rc;
    } else
    for (int64_t row0 = 0, panel = 0; row0 < p->m; row0 += p->panel_rows, ++panel) {
      const int64_t rows = std::min<int64_t>(p->panel_rows, p->m - row0);
      const char* Ap = reinterpret_cast<const char*>(p->A) + (size_t)row0 * p->lda * esz;
      int nwg1 = 0;
      if ((rc = launch_batch_product(p, Ap, p->b ? p->b + row0 : nullptr, rows, 1, p->rbuf16, &nwg1))) return rc;
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
    if ((rc = reduce_across(p, p->slabs16, (size_t)p->gram_splits * fos::BT_NV * p->n, false))) return rc;
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
  const bool shape_ok = p->path == 0 && !p->tall && !p->colblock && !p->resident && !p->col_sharded;
  if (!all_plain && controllable && same_family && shape_ok && p->entry != &kWideF32 && (nv >= 3 || p->comm)) {
    if (iters == 0) return FOS_OK;
    return run_multi_mfma(fs, nv, iters, true);
  }
  const bool streaming = shape_ok && all_plain;
  // (a sharded problem takes the matrix-core pass for any number of weights: its 16 gradients are one 16 x n all-reduce)
  MultiLaunch fn = (streaming && !p->comm && p->dtype == FOS_F32 && p->entry != &kWideF32) ? find_multi(p->n, nv) : nullptr;
  // the two-product pass costs about two single-vector passes per iteration whatever the number of weights: it pays
  // from three weights on (profiles/r02_multilambda.md); two weights without a VALU multi-vector kernel run one by one
  if (!fn && streaming && p->entry != &kWideF32 && (nv >= 3 || p->comm)) {
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
                     double* gd_out, double* work, hipStream_t st) {
  const int parts = vl_parts(n);
  hipLaunchKernelGGL(fos::lbfgs_gram_kernel, dim3(parts), dim3(fos::VL_THREADS), 0, st, g, S, Y, hist, head, std::max(cap, 1),
                     n, work);
  const int grid = (int)((n + fos::VL_THREADS - 1) / fos::VL_THREADS);      // one column per thread
  hipLaunchKernelGGL(fos::lbfgs_combine_kernel, dim3(grid), dim3(fos::VL_THREADS), 0, st, g, S, Y, hist, head,
                     std::max(cap, 1), n, (const double*)work, parts, d_out, gd_out);
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
    const bool timed = fg_ms != nullptr;
    if (timed) {
      hipLaunchKernelGGL(fos::stamp_kernel, dim3(1), dim3(1), 0, st, w.t_start);
      LAUNCH_CHECK();
    }
    int rc = fos_gemv_pair_dd(p, xv, alpha2, gv);
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
    if (n >= 2048) {                            // whole-chip form: two launches, each one read of the history
      if ((rc = launch_direction(g, w.S, w.Y, hist_n, head, M, n, w.d, host_dev + 6, w.vl, st))) return rc;
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
    hipLaunchKernelGGL(fos::lbfgs_first_trial_kernel, dim3(ax_grid), dim3(256), 0, st, x, (const double*)w.d, stp, w.x_old, n);
    LAUNCH_CHECK();
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
