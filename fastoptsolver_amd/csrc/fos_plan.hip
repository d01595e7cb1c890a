// libfos_hip.so, translation unit 1 of 4 - planner, kernel menus and the problem-level entry points of the C ABI
// (include/fos.h).  Host code only decides launch geometry and enqueues kernels; there is no CPU compute fallback.
#include "fos_internal.hpp"

namespace fosapi {

thread_local std::string g_err;


// ---- fused-kernel menu -------------------------------------------------------------------------------

template <typename T, int THREADS, int K, int R, int MINW, bool WITH_G, int NBUF, bool DUAL, bool DRAIN = false, bool CB = false,
          bool IL = false>
void fused_launch(const void* A, int64_t lda, const float* b, int64_t m, int n, YSource ys, int64_t rpw, float* slabs,
                  double* rr_part, double* rr2_part, int nwg, hipStream_t st) {
  hipLaunchKernelGGL((fos::gemv_pair_kernel<T, THREADS, K, R, true, MINW, WITH_G, NBUF, IL, DUAL, DRAIN, float, false, CB>), dim3(nwg),
                     dim3(THREADS), 0, st, reinterpret_cast<const T*>(A), lda, b, m, n, ys, rpw, slabs, rr_part,
                     rr2_part);
}

// fp64-accumulating form (fos_gemv_pair_dd): y and the slabs are doubles

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



// three register tiles in flight where that measured faster (same-run ratio to the fp32 kernel, tools/bench_dd.py:
// (512,4) fp32 0.96 -> 0.99, bf16 (256,4) 0.77 -> 0.83, bf16 (512,4) 0.77 -> 0.80; the 256-thread fp32 geometries lost
// 2-3 % and the (512,8) fp32 geometry would spill: those keep two)
#define DD_ENTRY(DT, T, TH, K, R, YL) { DT, TH, K, R, fused_launch_dd<T, TH, K, R, 2, YL, 2> }
#define DD_ENTRY3(DT, T, TH, K, R, YL) { DT, TH, K, R, fused_launch_dd<T, TH, K, R, 2, YL, 3> }
#define DD_ENTRY_IL(DT, T, TH, K, R, YL) { DT, TH, K, R, fused_launch_dd<T, TH, K, R, 2, YL, 2>, fused_launch_dd<T, TH, K, R, 2, YL, 2, true> }
#define DD_ENTRY3_IL(DT, T, TH, K, R, YL) { DT, TH, K, R, fused_launch_dd<T, TH, K, R, 2, YL, 3>, fused_launch_dd<T, TH, K, R, 2, YL, 3, true> }
const DdEntry kDdMenu[] = {
    DD_ENTRY(FOS_F32, float, 64, 1, 4, false), DD_ENTRY(FOS_F32, float, 64, 2, 2, false),
    DD_ENTRY(FOS_F32, float, 64, 3, 2, false),             // 513..768 columns (on 256 x 1 at 62-75 % filled: 58-66 % of 8 TB/s)
    DD_ENTRY(FOS_F32, float, 256, 1, 2, false), DD_ENTRY(FOS_F32, float, 256, 2, 2, false),
    DD_ENTRY(FOS_F32, float, 256, 3, 1, false),            // (3, 5 chunks: widths between the powers of two, as in kMenu: +3-5 %)
    DD_ENTRY_IL(FOS_F32, float, 256, 4, 1, false), DD_ENTRY_IL(FOS_F32, float, 512, 3, 1, false),
    DD_ENTRY3_IL(FOS_F32, float, 512, 4, 1, false), DD_ENTRY_IL(FOS_F32, float, 512, 5, 1, false),
    DD_ENTRY_IL(FOS_F32, float, 512, 8, 1, true),          // (512 x 6 with y in LDS measured 3 % behind this one at 12288 columns)
    DD_ENTRY(FOS_BF16, fos::bf16_t, 64, 1, 2, false), DD_ENTRY(FOS_BF16, fos::bf16_t, 64, 2, 2, false),
    DD_ENTRY(FOS_BF16, fos::bf16_t, 256, 1, 2, false),
    DD_ENTRY(FOS_BF16, fos::bf16_t, 256, 2, 1, false), DD_ENTRY_IL(FOS_BF16, fos::bf16_t, 256, 3, 1, false),
    // bf16 rows of 4 chunks per thread: the pass is VALU-bound (56 VALU instructions per 16-byte chunk: unpack + convert
    // for the dot, AGAIN for the gradient update, 16 v_fma_f64), so these keep the converted tile across the barrier
    // (KEEPCVT: 40 instructions) and pay with registers - two tiles in flight instead of three, per-chunk scheduling
    // barriers, the cross-wave sum through the DPP ladder.  tools/dd_bench: 262144 x 8192 69.5 -> 78.2 % of 8 TB/s,
    // 131072 x 16384 66.6 -> 74.5 %.
    { FOS_BF16, 256, 4, 1, fused_launch_dd<fos::bf16_t, 256, 4, 1, 2, true, 2, false, true>,
      fused_launch_dd<fos::bf16_t, 256, 4, 1, 2, true, 2, true, true> },
    DD_ENTRY_IL(FOS_BF16, fos::bf16_t, 512, 3, 1, false),
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
// Three-chunk geometries (round 3): a chunk beyond n is not idle - it re-reads the thread's chunk 0 to keep the loads
// branch-free - so a width of 3 * 2^k columns on the next power-of-two geometry paid for a quarter more load instructions
// (fp32 12288 columns 75 %, 10240 70 % of 8 TB/s against 84 % at 16384: tools/bench_widths.py).  Capacities are now
// 1, 2, 3, 4, 5, 6, 8, 10, 12, 14, 16 x 1024 columns (fp32), so from 2048 columns on at least 4/5 of the issued loads are
// live at any width (n = 5000 and 10000 land on the five-chunk geometries at 98 %).
// bf16 rows of 16385 ... 24576 columns get 512 threads x 6 chunks (1024 threads at 128 VGPRs spill; 8 chunks - y and the
// gradient slice 64 VGPRs each - spill at 256), 24577 ... 32768 the y-in-LDS kernel (gemv_wide.hpp), instead of two passes in
// column blocks (36 % of the roofline -> a single read).
#define ENTRY_IL_ND(DT, T, TH, K, R, W) \
  { DT, TH, K, R, PAIR(T, TH, K, R, W, 2, false, false, false), nullptr, nullptr, \
    PAIR(T, TH, K, R, W, 2, false, true, false), PAIR(T, TH, K, R, W, 2, false, false, true), nullptr }
const MenuEntry kMenu[] = {
    ENTRY_D(FOS_F32, float, 64, 1, 4, 2), ENTRY_D(FOS_F32, float, 64, 2, 4, 2),
    ENTRY_D(FOS_F32, float, 256, 1, 4, 2), ENTRY_D(FOS_F32, float, 256, 2, 4, 2), ENTRY_D(FOS_F32, float, 256, 3, 2, 2),
    ENTRY_D(FOS_F32, float, 256, 4, 2, 2), ENTRY_NB_IL(FOS_F32, float, 256, 5, 2, 2, 2),
    ENTRY_NB_IL(FOS_F32, float, 512, 3, 1, 2, 3),   ENTRY_NB_IL(FOS_F32, float, 512, 4, 1, 2, 3),
    ENTRY_NB_IL(FOS_F32, float, 512, 5, 1, 2, 3),   ENTRY_DRAIN(FOS_F32, float, 1024, 3, 1, 4, 512, 6, 2),
    ENTRY_NB_IL(FOS_F32, float, 512, 7, 1, 2, 2),   ENTRY_DRAIN(FOS_F32, float, 1024, 4, 1, 4, 512, 8, 2),
    ENTRY_D(FOS_F32, float, 512, 8, 1, 2),  ENTRY(FOS_F32, float, 1024, 2, 2, 4),
    ENTRY(FOS_BF16, fos::bf16_t, 64, 1, 4, 2), ENTRY(FOS_BF16, fos::bf16_t, 64, 2, 4, 2),   // one wave per row: up to 512 / 1024 bf16 columns
    ENTRY(FOS_BF16, fos::bf16_t, 256, 1, 4, 2), ENTRY(FOS_BF16, fos::bf16_t, 256, 2, 2, 2),
    ENTRY_NB_IL(FOS_BF16, fos::bf16_t, 256, 3, 1, 2, 3), ENTRY_NB_IL(FOS_BF16, fos::bf16_t, 256, 4, 1, 2, 3),
    ENTRY_IL_ND(FOS_BF16, fos::bf16_t, 256, 5, 1, 2),
    ENTRY_NB_IL(FOS_BF16, fos::bf16_t, 512, 3, 1, 2, 3), ENTRY_NB_IL(FOS_BF16, fos::bf16_t, 512, 4, 1, 2, 3),
    ENTRY_IL_ND(FOS_BF16, fos::bf16_t, 512, 5, 1, 2), ENTRY_IL_ND(FOS_BF16, fos::bf16_t, 512, 6, 1, 2),
};

const MenuEntry* find_entry(int dtype, int threads, int k, int r) {
  for (const auto& e : kMenu)
    if (e.dtype == dtype && e.threads == threads && e.k == k && e.r == r) return &e;
  return nullptr;
}
int epc_of(int dtype) { return dtype == FOS_F32 ? 4 : 8; }
// chunk-per-lane rows: up to 32 lanes per row - 128 fp32 / 256 bf16 columns
int64_t tlr_max_n(int dtype) { return 32 * (int64_t)epc_of(dtype); }
const MenuEntry* default_entry(int dtype, int64_t n) {
  for (const auto& e : kMenu)
    if (e.dtype == dtype && (int64_t)e.threads * e.k * epc_of(dtype) >= n) return &e;
  return nullptr;
}

// ---- wide rows (gemv_wide.hpp): y in LDS (dynamic shared memory above the 64 KiB default): 16384 < n <= 32768 fp32,
// 24576 < n <= 32768 bf16 ---
template <bool WITH_G, typename T>
void wide_launch(const void* A, int64_t lda, const float* b, int64_t m, int n, YSource ys, int64_t rpw, float* slabs,
                 double* rr_part, double* /*rr2_part*/, int nwg, hipStream_t st) {
  static std::atomic<uint64_t> done{0};
  (void)raise_dynamic_lds(&fos::gemv_wide_kernel<WITH_G, T>, fos::WD_MAX_N * sizeof(float), done);   // failure: see LAUNCH_CHECK
  hipLaunchKernelGGL((fos::gemv_wide_kernel<WITH_G, T>), dim3(nwg), dim3(fos::WD_THREADS), (size_t)n * sizeof(float), st,
                     reinterpret_cast<const T*>(A), lda, b, m, n, ys, rpw, slabs, rr_part);
}
const MenuEntry kWideF32 = {FOS_F32, fos::WD_THREADS, fos::WD_K, 1, wide_launch<true, float>, wide_launch<false, float>, nullptr};
const MenuEntry kWideBf16 = {FOS_BF16, fos::WD_THREADS, fos::WD_K / 2, 1, wide_launch<true, fos::bf16_t>,
                             wide_launch<false, fos::bf16_t>, nullptr};
const MenuEntry* wide_entry(int dtype) { return dtype == FOS_F32 ? &kWideF32 : &kWideBf16; }

// ---- tall-skinny entries (gemv_tall.hpp): n <= 64, any m / lda; one entry per column capacity and load form --------
template <typename T, int NC, int LOAD, bool WITH_G, bool DUAL>
void tall_launch(const void* A, int64_t lda, const float* b, int64_t m, int n, YSource ys, int64_t rpw, float* slabs,
                 double* rr_part, double* rr2_part, int nwg, hipStream_t st) {
  // The staged float4 copy streams one contiguous span: non-temporal once the matrix outgrows the 256 MiB Infinity Cache
  // (32000000 x 5: 71.8 -> 82.2 % of 8 TB/s); below that the cached form keeps the re-reads of a solver loop in the
  // cache (8000000 x 5 = 160 MB: 32.9 us cached, 34.7 us non-temporal).  tools/tall_bench.
  if (LOAD == fos::TL_STAGE4 && m * (int64_t)n * (int64_t)sizeof(T) > (192ll << 20))
    hipLaunchKernelGGL((fos::gemv_tall_kernel<T, NC, LOAD, WITH_G, DUAL, float, LOAD == fos::TL_STAGE4>), dim3(nwg),
                       dim3(fos::TL_THREADS), 0, st, reinterpret_cast<const T*>(A), lda, b, m, n, ys, rpw, slabs, rr_part, rr2_part);
  else
    hipLaunchKernelGGL((fos::gemv_tall_kernel<T, NC, LOAD, WITH_G, DUAL, float, false>), dim3(nwg), dim3(fos::TL_THREADS), 0, st,
                       reinterpret_cast<const T*>(A), lda, b, m, n, ys, rpw, slabs, rr_part, rr2_part);
}
template <typename T, int NC, int LOAD>
void tall_launch_dd(const void* A, int64_t lda, const float* b, int64_t m, int n, YSource ys, int64_t rpw, double* slabs,
                    double* rr_part, int nwg, hipStream_t st) {
  if (LOAD == fos::TL_STAGE4 && m * (int64_t)n * (int64_t)sizeof(T) > (192ll << 20))
    hipLaunchKernelGGL((fos::gemv_tall_kernel<T, NC, LOAD, true, false, double, LOAD == fos::TL_STAGE4>), dim3(nwg),
                       dim3(fos::TL_THREADS), 0, st, reinterpret_cast<const T*>(A), lda, b, m, n, ys, rpw, slabs, rr_part, (double*)nullptr);
  else
    hipLaunchKernelGGL((fos::gemv_tall_kernel<T, NC, LOAD, true, false, double, false>), dim3(nwg), dim3(fos::TL_THREADS), 0, st,
                       reinterpret_cast<const T*>(A), lda, b, m, n, ys, rpw, slabs, rr_part, (double*)nullptr);
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
  // fp32 pass: bf16 storage accumulates in fp32 like its streaming siblings, fp32 storage in fp64 (gemv_tall.hpp "AT")
  // (fp32 rows of 65..128 columns - 32 lanes per row - ran on the fp32-accumulating streaming kernel until round 3: fp32 too)
  using AT = typename std::conditional<std::is_same<T, float>::value && (LPR < 32), double, float>::type;
  hipLaunchKernelGGL((fos::gemv_tall_rows_kernel<T, LPR, WITH_G, DUAL, float, AT>), dim3(nwg), dim3(fos::TL_THREADS), 0, st,
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
const MenuEntry kTallRowsF32[3] = {TALLR(FOS_F32, float, 8), TALLR(FOS_F32, float, 16), TALLR(FOS_F32, float, 32)};
const MenuEntry kTallRowsBf16[3] = {TALLR(FOS_BF16, fos::bf16_t, 8), TALLR(FOS_BF16, fos::bf16_t, 16), TALLR(FOS_BF16, fos::bf16_t, 32)};
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
    // (round 3) 65..128 columns: a row per 32 lanes (fp32) / 16 lanes (bf16) - on the one-wave-per-row streaming geometry
    // a 72-column row left 46 of 64 lanes re-reading chunk 0 (53 % of 8 TB/s at 7456512 x 72; tools/bench_widths.py narrow)
    const int chunks = (int)(n / epc);
    if (dtype == FOS_F32) return &kTallRowsF32[chunks <= 8 ? 0 : chunks <= 16 ? 1 : 2];
    return &kTallRowsBf16[chunks <= 8 ? 0 : chunks <= 16 ? 1 : 2];       // bf16: up to 256 columns
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


void plan_fused(fos_problem* p, const MenuEntry* e, int nwg_hint) {
  p->entry = e;
  p->path = 0;
  int64_t m = p->m;
  // Workgroups per CU (profiles/r02_sweep_wgs.log): one for the wide geometries, whose register tiles already hold
  // 64-96 KiB of rows in flight per CU; two for the 256-thread geometries with 1-2 chunks per thread (1048576 x 1024:
  // 67 % -> 88 % of the roofline - one such workgroup has 32 KiB in flight, below HBM latency x bandwidth per CU);
  // 16 single-wave workgroups for the one-wave-per-row geometries.
  // One-wave-per-row geometries (64 threads), round 3: FOUR workgroups per CU - a wave per SIMD - not sixteen.  Whole step at
  // ~2 GiB (tools/wg_sweep.py, profiles/r03_wg_sweep.txt): 2097152 x 256 77.5 -> 86.1 % of 8 TB/s, 1048576 x 512 75.6 -> 84.3 %,
  // 1398016 x 320 71.2 -> 82.3 %; the A pass itself is 5-8 % faster with fewer, longer streams, and the update kernel sums a
  // quarter of the slabs (200000 x 256: its share fell from 26 to 8 us of a 63 -> 44 us iteration).  Rows that leave more than
  // a fifth of the wave's lanes without a chunk (n <= 204 on the one-chunk geometry) need eight: 2796032 x 160 74.3 -> 79.9 %.
  const bool sparse_wave = e->threads == 64 && e->k == 1 && p->n * 5 <= 4 * 64 * (int64_t)epc_of(p->dtype);
  // 256 threads: two per CU up to two chunks per thread, one above; a row that leaves lanes without a chunk wants one more -
  // 838656 x 640 on 256 x 1 (62 % filled): 77.5 -> 82.4 % with three, 209664 x 2560 on 256 x 3 (83 %): 82.3 -> 86.1 % with two
  // (full rows on 256 x 3: 88 % with one, 82 % with two)
  const int64_t cap256 = 256 * (int64_t)e->k * epc_of(p->dtype);
  const bool sparse_256 = e->threads == 256 && ((e->k == 1 && p->n * 10 <= 7 * cap256) || (e->k == 3 && p->n * 10 <= 9 * cap256));
  const int per_cu = e->threads >= 512 ? 1 : e->threads == 256 ? (e->k <= 2 ? 2 : 1) + (sparse_256 ? 1 : 0) : (sparse_wave ? 8 : 4);
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
  // 4 workgroups per CU for the row-per-thread forms (profiles/r02_sweep_wgs.log)
  // (round 3, whole-step sweep tools/wg_sweep.py: the chunk-per-lane form wants 2 workgroups per CU when its rows fill the
  //  lanes - 2000000 x 64: 102.7 -> 89.6 us, 4000000 x 32: 93.3 -> 89.5 us - and 4 when a fifth or more of them idle -
  //  2000000 x 96: 155.6 -> 148.3 us; eight only made the update kernel sum more slabs.  It also keeps its workgroup count
  //  up on shorter matrices - 384 rows per workgroup are enough: 300000 x 64 31.6 -> 25.7 us, 200000 x 100 43.7 -> 30.0 us.)
  const int64_t lane_cols = (int64_t)e->k * epc_of(p->dtype);            // k = lanes per row of the chunk-per-lane entries
  // (a row per 32 lanes, 8 rows per workgroup step: three - 4194304 x 128: 76.1 % at two, 80.0 % at three or four)
  const int per_cu = e->k > 0 ? (p->n * 5 <= 4 * lane_cols ? 4 : (e->k >= 32 ? 3 : 2)) : 4;
  const int64_t rows_min = e->k > 0 ? 384 : 4 * fos::TL_THREADS;
  // Short matrices are latency-bound: up to one workgroup per CU from 64 rows each, whatever rows_min says (20000 x 5: 13.0 ->
  // 10.9 us per iteration, 10000 x 100: 14.6 -> 11.6 us, 30000 x 100: 16.8 -> 13.1 us; profiles/r03_wg_sweep.txt)
  const int64_t by_rows = std::max<int64_t>(p->m / rows_min, std::min<int64_t>(p->ncu, p->m / 64));
  int64_t nwg = std::max<int64_t>(1, std::min<int64_t>(per_cu * (int64_t)p->ncu, by_rows));
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
int launch_pass(fos_problem* p, const YSource& ys, const float* b, bool with_g, int* n_rr, bool dual) {
  int rc = prof_mark(p, true);
  if (rc) return rc;
  if ((rc = launch_pass_inner(p, ys, b, with_g, n_rr, dual))) return rc;
  return prof_mark(p, false);
}


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
                         const int* stopped) {
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

int launch_residual_batch(fos_problem* p, int use_b, double* out16, const int* stopped) {
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

// The matrix-core passes (16 candidates / 16 weights) need the aligned layout only - not a streaming plan: rows of 65..128
// columns run the chunk-per-lane pass (p->tall) and keep them; n <= 64 may be ragged / misaligned and does not.
bool batch_supported(const fos_problem* p) { return p->path == 0 && (!p->tall || p->n > fos::TL_MAX_N); }   // fp32: f32 MFMA; bf16: 3-term bf16 MFMA

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
  if ((n <= fos::TL_MAX_N || (n <= tlr_max_n(p->dtype) && vec_ok)) && !(flags & FOS_PLAN_NO_TALL))
    plan_tall(p, tall_entry(p->dtype, n, p->lda, p->A));
  else if (e) plan_fused(p, e, 0);
  else if (vec_ok && n <= fos::WD_MAX_N && !(flags & FOS_PLAN_NO_WIDE)) plan_fused(p, wide_entry(p->dtype), 0);
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
      // Swept on the whole fg (pass + fp64 slab sum, tools/wg_sweep_dd.py, profiles/r03_wg_sweep.txt): the single-wave
      // geometries want eight per CU (2097152 x 256: 378 -> 348 us, 1677568 x 320: 420 -> 357 us) except full two-chunk rows
      // (1048576 x 512: four, 308 us against 336); two-chunk 256-thread rows that leave 30 % of the lanes without a chunk four
      // instead of two (419328 x 1280: 403 -> 371 us).
      const int64_t cap = (int64_t)e->threads * e->k * epc_of(p->dtype);
      const bool full = p->n * 10 > 9 * cap, sparse = p->n * 10 <= 7 * cap;
      int nwg = p->ncu * (e->threads >= 512 ? 1
                          : e->threads == 256 ? (e->k == 1 ? 4 : (e->k == 2 && sparse ? 4 : 2))
                                              : (e->k >= 3 || (e->k == 2 && full) ? 4 : 8));      // (64 x 3, 838656 x 640: 313 us at four, 331 at eight)
      const int64_t row_bytes = p->n * (p->dtype == FOS_F32 ? 4 : 2);
      // (below half a GiB the slab sum outweighs it: 200000 x 256 51 us with four per CU, 57 us with eight)
      if (e->threads == 64 && p->m * row_bytes < (512ll << 20)) nwg = std::min(nwg, 4 * p->ncu);
      const int64_t min_rows = std::max<int64_t>(2 * (int64_t)e->r, (65536 + row_bytes - 1) / row_bytes);   // fp64 slabs
      if (p->dd_nwg_hint > 0) nwg = p->dd_nwg_hint;
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

// One launch of the one-read multi-lambda pass (cluster_pass.hpp): all members of a cluster must be resident (they are:
// one workgroup per CU).
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
  // A PLAIN launch: #CUs workgroups with > 80 KiB of LDS each are co-resident by grid size - all a cooperative launch
  // would check (MI355X_MICROARCH.md "coop-launch") - every wait in the kernel is bounded, and a process that used
  // hipLaunchCooperativeKernel dies in rocprofv3's exit handler (tools/exit_probe.py, round 3).
  hipLaunchKernelGGL(kern, dim3((unsigned)(p->cp_clusters * CS)), dim3(fos::CP_THREADS), (unsigned)fos::CP_LDS_BYTES, p->stream, A,
                     lda, b, m, n, n_pad, xp, rpc, xcd_aware, p->cp_xchg, p->cp_flags, epoch, p->slabs16, n_stride, p->cp_error);
  LAUNCH_CHECK();
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


// Read-only streaming probe: what this box's HBM delivers to a kernel that does nothing but load (the context figure SURVEY.md
// 8d asks for beside the nominal 8 TB/s).  The fastest of 18 x 4 loads-only variants (tools/read_probe_sweep.hip,
// profiles/r03_read_probe_sweep.txt: 7.19 TB/s; grid-stride order 7.1, temporal loads 6.2, 4 workgroups per CU 6.1): one
// 512-thread workgroup per CU streams its own contiguous range with 8 independent 16-byte non-temporal loads in flight per
// thread - the access pattern of the single-pass kernel without its arithmetic.
// ORDER 0: every workgroup streams its own contiguous range; 1: all workgroups sweep one window, 8 KiB per workgroup and trip;
// 2: the same with 64 KiB per workgroup and trip - the interleaved row order of the product kernel at 64 KiB rows
template <int ORDER>
__global__ __launch_bounds__(512) void stream_read_kernel(const fos::f32x4* __restrict__ src, size_t n16, float* __restrict__ sink) {
  constexpr int UNR = 8, THREADS = 512;
  constexpr bool BLOCKS = ORDER == 0;
  fos::f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if constexpr (ORDER == 2) {
    constexpr size_t CH = (size_t)UNR * THREADS;       // 16-byte units per workgroup and trip = 64 KiB
    size_t base = (size_t)blockIdx.x * CH;
    for (; base + CH <= n16; base += (size_t)gridDim.x * CH) {
      fos::f32x4 v[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) v[u] = __builtin_nontemporal_load(src + base + threadIdx.x + (size_t)u * THREADS);
#pragma unroll
      for (int u = 0; u < UNR; ++u) acc += v[u];
    }
    for (size_t i = base + threadIdx.x; i < n16 && i < base + CH; i += THREADS) acc += __builtin_nontemporal_load(src + i);
  } else if constexpr (BLOCKS) {                        // every workgroup streams its own contiguous range
    const size_t per = ((n16 + gridDim.x - 1) / gridDim.x + THREADS * UNR - 1) / (THREADS * UNR) * (THREADS * UNR);
    const size_t lo = per * blockIdx.x, hi = lo + per < n16 ? lo + per : n16;
    size_t i = lo + threadIdx.x;
    for (; i + (size_t)(UNR - 1) * THREADS < hi; i += (size_t)UNR * THREADS) {
      fos::f32x4 v[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) v[u] = __builtin_nontemporal_load(src + i + (size_t)u * THREADS);
#pragma unroll
      for (int u = 0; u < UNR; ++u) acc += v[u];
    }
    for (; i < hi; i += THREADS) acc += __builtin_nontemporal_load(src + i);
  } else {                                              // all workgroups sweep one window (the interleaved row order's pattern)
    const size_t stride = (size_t)gridDim.x * THREADS;
    size_t i = (size_t)blockIdx.x * THREADS + threadIdx.x;
    for (; i + (UNR - 1) * stride < n16; i += UNR * stride) {
      fos::f32x4 v[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) v[u] = __builtin_nontemporal_load(src + i + u * stride);
#pragma unroll
      for (int u = 0; u < UNR; ++u) acc += v[u];
    }
    for (; i < n16; i += stride) acc += __builtin_nontemporal_load(src + i);
  }
  const float t = fos::wave_sum((acc.x + acc.y) + (acc.z + acc.w));
  if ((threadIdx.x & 63) == 0) sink[blockIdx.x * 8 + (threadIdx.x >> 6)] = t;
}

}  // namespace fosapi

using namespace fosapi;

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
  if (!vec_ok || p->n <= tlr_max_n(p->dtype))
    return fail(FOS_ERR_UNSUPPORTED, "fos_problem_set_comm_cols: needs the streaming layout (aligned, more than 32 chunks of 16 bytes per row and rank)");
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
                           FOS_PLAN_INTERLEAVE | FOS_PLAN_NO_INTERLEAVE | FOS_PLAN_NO_CLUSTER | FOS_PLAN_FUSED_MFMA |
                           FOS_PLAN_CHIP_RESIDENT | FOS_PLAN_NO_CHIP_RESIDENT))
    return fail(FOS_ERR_ARG, "fos_problem_replan: unknown flag");
  // the multi-lambda workspace follows its own plan (one-read cluster form or two products): rebuilt on first use
  {
    void* multi[] = {p->rbuf16, p->slabs16, p->cp_xchg, p->cp_flags, p->cp_error};
    for (void* q : multi)
      if (q) (void)hipFree(q);
    p->rbuf16 = p->slabs16 = p->cp_xchg = nullptr; p->cp_flags = nullptr; p->cp_error = nullptr;
    p->cp_cs = p->cp_clusters = 0;
    p->cp_mode = (flags & FOS_PLAN_CLUSTER) ? 1 : (flags & FOS_PLAN_NO_CLUSTER) ? 2 : 0;
    p->fused_on = (flags & FOS_PLAN_FUSED_MFMA) != 0;
    p->chip_mode = (flags & FOS_PLAN_CHIP_RESIDENT) ? 1 : (flags & FOS_PLAN_NO_CHIP_RESIDENT) ? 2 : 0;
    flags &= ~(unsigned)(FOS_PLAN_CLUSTER | FOS_PLAN_NO_CLUSTER | FOS_PLAN_FUSED_MFMA | FOS_PLAN_CHIP_RESIDENT |
                         FOS_PLAN_NO_CHIP_RESIDENT);
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

int fos_stream_read_probe(const void* buf, size_t bytes, int launches, void* stream, double* gbps_out, double* us_out) {
  if (!buf || bytes < 16 || (bytes & 15) || (reinterpret_cast<uintptr_t>(buf) & 15) || launches < 1 || !gbps_out)
    return fail(FOS_ERR_ARG, "fos_stream_read_probe: needs a 16-byte aligned buffer of a multiple of 16 bytes and launches >= 1");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  int dev = 0, ncu = 0;
  HIP_TRY(hipGetDevice(&dev));
  HIP_TRY(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
  const size_t n16 = bytes / 16;
  const int grid = (int)std::max<size_t>(1, std::min<size_t>((size_t)ncu, (n16 + 4095) / 4096));
  float* sink = nullptr;
  HIP_TRY(hipMalloc(&sink, (size_t)grid * 8 * sizeof(float)));
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipError_t e = hipEventCreate(&e0);
  if (e == hipSuccess) e = hipEventCreate(&e1);
  // three orders - contiguous ranges per workgroup, one window swept by all in 8 KiB or in 64 KiB pieces - and the fastest one
  // counts: which of them leads depends on the size and on the state of the device (profiles/r03_row_order.md), as for the
  // product kernel
  float ms = 0.f;
  for (int order = 0; order < 3 && e == hipSuccess; ++order) {
    void (*kern)(const fos::f32x4*, size_t, float*) =
        order == 0 ? stream_read_kernel<0> : order == 1 ? stream_read_kernel<1> : stream_read_kernel<2>;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 0, st, (const fos::f32x4*)buf, n16, sink);      // warm-up
    e = hipEventRecord(e0, st);
    for (int i = 0; i < launches && e == hipSuccess; ++i) {
      hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 0, st, (const fos::f32x4*)buf, n16, sink);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipEventRecord(e1, st);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    float t = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&t, e0, e1);
    if (e == hipSuccess && (order == 0 || t < ms)) ms = t;
  }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  (void)hipFree(sink);
  if (e != hipSuccess) return fail(FOS_ERR_HIP, std::string("fos_stream_read_probe: ") + hipGetErrorString(e));
  const double us = (double)ms * 1e3 / launches;
  *gbps_out = (double)bytes / (us * 1e-6) / 1e9;
  if (us_out) *us_out = us;
  return FOS_OK;
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
                  p->slabs_dd, p->rr_dd, p->lhist, p->rbuf16, p->rcols16, p->mfold, p->cr_part, p->cr_bar, p->slabs16, p->rneg, p->zeros, p->cp_xchg, p->cp_flags, p->fz_bar, p->fz_part, p->fz_beta,
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
            ((p->il && p->entry && p->entry->with_g_il && p->path == 0 && !p->colblock && !p->tall) ? 32 : 0) |
            (p->fused_on ? 64 : 0) | (p->chip_mode == 1 ? 128 : 0);
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

int fos_problem_tune_dd(fos_problem* p, int workgroups) {
  if (!p || workgroups < 0) return fail(FOS_ERR_ARG, "fos_problem_tune_dd: bad argument");
  if (p->tall || p->resident) return fail(FOS_ERR_UNSUPPORTED, "fos_problem_tune_dd: the tall / resident plans share the fp32 pass's grid");
  HIP_TRY(hipStreamSynchronize(p->stream));
  if (p->slabs_dd) (void)hipFree(p->slabs_dd);
  if (p->rr_dd) (void)hipFree(p->rr_dd);
  p->slabs_dd = nullptr; p->rr_dd = nullptr;
  p->dd_entry = nullptr; p->dd_two_pass_chunks = 0;
  p->dd_nwg_hint = workgroups;
  return FOS_OK;                                   // re-planned on the next fp64 pass (ensure_dd)
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

}  // extern "C"
int fosapi::launch_pass_dd(fos_problem* p, const YSource& ys, double alpha2, const double* l2vec, double* out) {
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
extern "C" {

int fos_gemv_pair_dd(fos_problem* p, const double* x, double alpha2, double* grad_rr) {
  return fosapi::gemv_pair_dd_stamped(p, x, alpha2, grad_rr, nullptr, nullptr);
}

}  // extern "C"
// fos_gemv_pair_dd; t_stamp (nullable): where the pass leaves the wall clock of its start, *stamped says whether the
// kernel that serves this plan does (the streaming fp64 kernel; the others leave it to the caller's stamp kernel)
int fosapi::gemv_pair_dd_stamped(fos_problem* p, const double* x, double alpha2, double* grad_rr, unsigned long long* t_stamp,
                                 bool* stamped) {
  if (stamped) *stamped = false;
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
  if (t_stamp != nullptr) {
    bool ok = false;
    int rc = dd_pass_stamps(p, &ok);
    if (rc) return rc;
    if (ok) ys.t_stamp = t_stamp;
    if (stamped) *stamped = ok;
  }
  return launch_pass_dd(p, ys, alpha2, x, grad_rr);
}
// whether the kernel behind fos_gemv_pair_dd on this plan writes YSource::t_stamp (the streaming fp64 kernel does)
int fosapi::dd_pass_stamps(fos_problem* p, bool* yes) {
  *yes = false;
  if (p->resident) return FOS_OK;
  int rc = ensure_dd(p);
  if (rc) return rc;
  *yes = p->dd_entry != nullptr && !p->tall && !p->col_sharded;
  return FOS_OK;
}
extern "C" {

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

}  // extern "C"
