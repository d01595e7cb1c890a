// The FISTA step as BASELINE's north_star words it, literally: the GEMV pair on LDS-staged tiles of A with a CDNA4 matrix
// instruction, fused in ONE launch with the soft-threshold prox and the momentum update, the iterate resident on the chip
// for the whole run (iterative_solvers.py:170-242, :289-342; plain runs: no backtracking / restart / tolerances).
//
// It is an OPT-IN plan (FOS_PLAN_FUSED_MFMA): the default step - two launches, VALU on register tiles (gemv_pair.hpp +
// reduce_update.hpp) - measures faster, for the reasons DESIGN.md "MFMA analysis" gives; this kernel exists so that the
// comparison is a measurement, not an argument.
//
// One persistent launch of G = #CUs workgroups (512 threads, one per CU: co-resident by grid size), `iters` iterations:
//   phase A  workgroup w streams its rows in PANELS of 4.  Wave v owns the column strip [v n/8, (v+1) n/8):
//            - the panel arrives with fully coalesced 16-byte loads (1 KiB per wave instruction, non-temporal) in registers,
//              one panel ahead, and is STAGED in LDS (row-major, the wave's own region of the tile);
//            - product 1, r = A_panel y - b:  v_mfma_f32_4x4x1_16B_f32 (16 independent 4x4 outer products): block k of the
//              instruction takes column c0 + 4k + t of the 4 rows (A operand: lane 4k + i <- row i, read back from LDS as
//              one ds_read_b128 per 4 instructions) against y[c0 + 4k + t] in the B operand's column 0 - a quarter of the
//              instruction's N dimension is used, the most any matrix instruction offers one right-hand side;
//            - the 16 blocks' partial dots are summed over the lanes (DPP ladder) and over the 8 waves (LDS, the ONE
//              workgroup barrier of a panel);
//            - product 2, g += A_panel^T r, on the VALU from the staged tile (a matrix instruction cannot contract over
//              the rows: they sit on its OUTPUT index): lane l owns 16 columns of the strip, 4 rows x 16 FMAs;
//            the workgroup's partial gradient goes to its slab (as in the two-launch design: fixed-order sums).
//   barrier  grid-wide (arrival counter + generation, agent-scope release / acquire, every spin bounded)
//   phase B  workgroup w OWNS n/G columns of the iterate: it sums the G slabs for them in slab order, applies
//            +alpha2 y, the prox and the momentum in fp64 on x_k, x_{k-1} that live in its LDS for the whole run, and
//            publishes its slice of y_{k+1} (fp32, as the two-launch design rounds it)
//   barrier  grid-wide; every workgroup re-reads y_{k+1} into its matrix-instruction operands.
// Requirements (host-checked): fp32 A, n % 2048 == 0, n <= 8192, 16-byte aligned rows, m >= 8 G.
#pragma once
#include "gemv_pair.hpp"
#include "reduce_update.hpp"

namespace fos {

constexpr int FZ_THREADS = 512, FZ_NW = FZ_THREADS / 64, FZ_ROWS = 4, FZ_PAD = 16, FZ_OWN_MAX = 64;

typedef float f32x4_t __attribute__((ext_vector_type(4)));

struct FusedArgs {
  const float* A; int64_t lda; const float* b; int64_t m; int n; int64_t rows_per_wg;
  float* slabs;            // [G][n]
  float* y;                // n floats: y_k on entry, y_{k+iters} on exit
  double* x_cur;           // n doubles (state)
  double* x_prev;
  const double* beta;      // [iters + 1]: beta[k] forms y_k from (x_k, x_{k-1}); beta[k + 1] the next one
  double* part;            // [2][G][4]: {sum d^2, sum gf^2, sum |x|, sum x^2} of the last two iterations (parity of k)
  double* rr_part;         // [G]: ||A_w y - b_w||^2 of the LAST iteration's pass
  unsigned* bar;           // FZ_BAR_WORDS words, zeroed once: see fz_grid_barrier
  int iters; int prox_kind; long long k0;
  double tau, alpha1, alpha2;
  unsigned long long timeout_ticks;
  unsigned long long* stamps;   // nullable, [G][8]: 100 MHz wall clock of the LAST iteration - phase A start / end, behind
};                              // barrier 1, phase B end, behind barrier 2 (measurement hook: fos_problem_set_fused_stamps)

// Grid-wide barrier for a grid that is co-resident by construction, in two levels (round 3: with one arrival counter and the
// generation word beside it, 256 workgroups arriving together took 9-20 us to get through - 256 read-modify-writes and 255
// pollers on ONE line, tools/fused_phases.py).  Workgroup w arrives on the counter of group w % 16 (its own 128-byte line);
// the last of a group arrives on the root; the last of the root bumps the 16 group generations (a line each), and a
// workgroup polls its group's only.  Layout in 32-word lines: [0] root counter, [1] error flag, [2 + g] group counters,
// [18 + g] group generations.  Returns false (and raises the error flag) when a wait ran out - the caller leaves the kernel;
// every other workgroup then runs out too.
constexpr int FZ_NG = 16, FZ_LINE = 32, FZ_BAR_WORDS = (2 + 2 * FZ_NG) * FZ_LINE;
__device__ inline bool fz_grid_barrier(unsigned* bar, unsigned nwg, unsigned& gen, unsigned long long timeout_ticks, int* ok_lds) {
  __syncthreads();                                   // this workgroup's stores are issued ...
  if (threadIdx.x == 0) {
    __threadfence();                                 // ... and released at agent scope
    int ok = 1;
    const unsigned w = blockIdx.x, grp = w % FZ_NG;
    const unsigned ngroups = nwg < (unsigned)FZ_NG ? nwg : (unsigned)FZ_NG;
    const unsigned gsize = (nwg - grp + FZ_NG - 1) / FZ_NG;              // workgroups w' < nwg with w' % 16 == grp
    unsigned* root = bar;
    unsigned* err = bar + FZ_LINE;
    unsigned* gcnt = bar + (2 + grp) * FZ_LINE;
    unsigned* ggen = bar + (2 + FZ_NG + grp) * FZ_LINE;
    bool releaser = false;
    if (__hip_atomic_fetch_add(gcnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gsize - 1) {
      __hip_atomic_store(gcnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // (release: the reset above is ordered before whatever lets a workgroup arrive on this group's counter again)
      if (__hip_atomic_fetch_add(root, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT) == ngroups - 1) {
        __hip_atomic_store(root, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        releaser = true;
      }
    }
    if (releaser) {
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");         // one fence, then 16 stores in flight together (a release
      for (unsigned g = 0; g < ngroups; ++g)                     // per store staggered the groups by 0.5 us each)
        __hip_atomic_store(bar + (2 + FZ_NG + g) * FZ_LINE, gen + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      const unsigned long long t0 = wall_clock64();
      while (__hip_atomic_load(ggen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen) {
        if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u || wall_clock64() - t0 > timeout_ticks) {
          __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ok = 0;
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    *ok_lds = ok;
  }
  gen += 1;
  __syncthreads();
  return *ok_lds != 0;
}

// NQ = (n / 8) / 256: 16-byte chunks per lane and row in the wave's strip (1..4); NP = 4 NQ patches of 64 columns.
template <int NQ>
__global__ __launch_bounds__(FZ_THREADS, 2) void fista_fused_kernel(FusedArgs a) {
  constexpr int NP = 4 * NQ;
  extern __shared__ __attribute__((aligned(16))) float fz_lds[];
  const int n = a.n;
  const int ldt = n + FZ_PAD;                                  // tile row stride in floats
  float* tile = fz_lds;                                        // [4][n + PAD]
  float* red = tile + FZ_ROWS * ldt;                           // [2][FZ_NW][4]
  double* xs_cur = reinterpret_cast<double*>(red + 2 * FZ_NW * 4);   // [FZ_OWN_MAX]
  double* xs_prev = xs_cur + FZ_OWN_MAX;
  float* gsum = reinterpret_cast<float*>(xs_prev + FZ_OWN_MAX);       // [16][FZ_OWN_MAX] phase-B partial sums
  double* accs = reinterpret_cast<double*>(gsum + 16 * FZ_OWN_MAX);   // [FZ_OWN_MAX][4]
  int* ok_lds = reinterpret_cast<int*>(accs + FZ_OWN_MAX * 4);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned G = gridDim.x;
  const int w = blockIdx.x;
  const int strip = n / FZ_NW;
  const int sbase = wave * strip;                              // first column of this wave's strip
  const int own = n / (int)G > 0 ? (n + (int)G - 1) / (int)G : 1;      // columns this workgroup owns in phase B
  const int own_lo = w * own;
  const int own_n = own_lo >= n ? 0 : (own_lo + own > n ? n - own_lo : own);
  const int64_t row_lo = (int64_t)w * a.rows_per_wg;
  int64_t row_hi = row_lo + a.rows_per_wg;
  if (row_hi > a.m) row_hi = a.m;
  const int64_t npanels = row_hi > row_lo ? (row_hi - row_lo + FZ_ROWS - 1) / FZ_ROWS : 0;
  const char* Ab = reinterpret_cast<const char*>(a.A);
  const int64_t row_bytes = a.lda * 4;

  unsigned gen = 0;
  if (tid == 0)                                                 // the generation this launch starts from (all groups agree)
    *ok_lds = (int)__hip_atomic_load(a.bar + (2 + FZ_NG + w % FZ_NG) * FZ_LINE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  gen = (unsigned)*ok_lds;
  __syncthreads();

  // the iterate's slice this workgroup owns: resident in LDS for the whole run
  if (tid < own_n) { xs_cur[tid] = a.x_cur[own_lo + tid]; xs_prev[tid] = a.x_prev[own_lo + tid]; }
  __syncthreads();

  const int blk = lane >> 2, sub = lane & 3;                   // matrix-instruction block / row (A) or column (B) of this lane

  for (int it = 0; it < a.iters; ++it) {
    // ---- phase A ----------------------------------------------------------------------------------------------------
    // B operands: y of the strip in the block layout (column 0 of each 1 x 4 block row; the other three columns are zero)
    float ymf[NP][4];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const f32x4_t yv = *reinterpret_cast<const f32x4_t*>(a.y + sbase + 64 * p + 4 * blk);
      ymf[p][0] = sub == 0 ? yv.x : 0.f; ymf[p][1] = sub == 0 ? yv.y : 0.f;
      ymf[p][2] = sub == 0 ? yv.z : 0.f; ymf[p][3] = sub == 0 ? yv.w : 0.f;
    }
    float gacc[NQ][4];
#pragma unroll
    for (int q = 0; q < NQ; ++q) gacc[q][0] = gacc[q][1] = gacc[q][2] = gacc[q][3] = 0.f;
    double rr = 0.0;
    const bool stamp = a.stamps != nullptr && it == a.iters - 1 && tid == 0;
    if (stamp) a.stamps[w * 8 + 0] = wall_clock64();

    u32x4 pre[FZ_ROWS][NQ];
    float bpre[FZ_ROWS];
    auto issue = [&](int64_t panel) {
#pragma unroll
      for (int r = 0; r < FZ_ROWS; ++r) {
        int64_t row = row_lo + panel * FZ_ROWS + r;
        if (row >= row_hi) row = row_hi - 1;                  // clamp: weighted by zero below
        bpre[r] = a.b != nullptr ? a.b[row] : 0.f;
        const char* rp = Ab + row * row_bytes + (size_t)sbase * 4 + (size_t)lane * 16;
#pragma unroll
        for (int q = 0; q < NQ; ++q) pre[r][q] = load16<true>(rp + (size_t)q * 1024);
      }
    };
    if (it == 0 && npanels > 0) issue(0);                       // later iterations: issued before the grid barrier below
    for (int64_t panel = 0; panel < npanels; ++panel) {
      // stage the panel: this wave's region of the tile (private to the wave: no workgroup barrier around it)
      float bcur[FZ_ROWS];
#pragma unroll
      for (int r = 0; r < FZ_ROWS; ++r) {
        bcur[r] = bpre[r];
#pragma unroll
        for (int q = 0; q < NQ; ++q)
          *reinterpret_cast<u32x4*>(tile + r * ldt + sbase + 256 * q + 4 * lane) = pre[r][q];
      }
      issue(panel + 1 < npanels ? panel + 1 : panel);           // next panel in flight (last: a re-read, L2 hit)
      // product 1 on the matrix cores: D[i][0] of block k accumulates sum_t A[i][c0 + 4k + t] y[c0 + 4k + t]
      f32x4_t D = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const f32x4_t av = *reinterpret_cast<const f32x4_t*>(tile + sub * ldt + sbase + 64 * p + 4 * blk);
        D = __builtin_amdgcn_mfma_f32_4x4x1f32(av.x, ymf[p][0], D, 0, 0, 0);
        D = __builtin_amdgcn_mfma_f32_4x4x1f32(av.y, ymf[p][1], D, 0, 0, 0);
        D = __builtin_amdgcn_mfma_f32_4x4x1f32(av.z, ymf[p][2], D, 0, 0, 0);
        D = __builtin_amdgcn_mfma_f32_4x4x1f32(av.w, ymf[p][3], D, 0, 0, 0);
        if ((p & 3) == 3) __builtin_amdgcn_sched_barrier(0);      // at most four operand reads ahead (registers)
      }
      // register v of lane 4k holds row v's partial of block k (lanes 4k + 1..3: products with the zero columns):
      // sum over the wave, then over the waves
      float part[FZ_ROWS] = {wave_sum(D.x), wave_sum(D.y), wave_sum(D.z), wave_sum(D.w)};
      const int pb = (int)(panel & 1);
      if (lane == 0) {
#pragma unroll
        for (int r = 0; r < FZ_ROWS; ++r) red[(pb * FZ_NW + wave) * 4 + r] = part[r];
      }
      __syncthreads();
      // the 8 x 4 wave partials: one per lane (lane 4 v + r), two DPP shifts fold the waves, eight readlanes make the four
      // residuals wave-uniform (the straightforward 32 LDS reads per lane cost 32 VGPRs this kernel does not have)
      float fold = lane < FZ_NW * 4 ? red[pb * FZ_NW * 4 + lane] : 0.f;
      fold += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(fold), 0x114, 0xF, 0xF, false));   // row_shr:4
      fold += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(fold), 0x118, 0xF, 0xF, false));   // row_shr:8
      float res[FZ_ROWS];
#pragma unroll
      for (int r = 0; r < FZ_ROWS; ++r) {
        float s = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(fold), 12 + r)) +
                  __int_as_float(__builtin_amdgcn_readlane(__float_as_int(fold), 28 + r));
        const int64_t row = row_lo + panel * FZ_ROWS + r;
        if (row < row_hi) { s -= bcur[r]; rr += (double)s * (double)s; } else s = 0.f;
        res[r] = s;
      }
      // product 2 on the VALU, from the staged tile in the plain layout (lane l owns columns 256 q + 4 l .. + 3)
#pragma unroll
      for (int r = 0; r < FZ_ROWS; ++r) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const f32x4_t av = *reinterpret_cast<const f32x4_t*>(tile + r * ldt + sbase + 256 * q + 4 * lane);
          gacc[q][0] = fmaf(av.x, res[r], gacc[q][0]); gacc[q][1] = fmaf(av.y, res[r], gacc[q][1]);
          gacc[q][2] = fmaf(av.z, res[r], gacc[q][2]); gacc[q][3] = fmaf(av.w, res[r], gacc[q][3]);
        }
      }
    }
    {
      float* slab = a.slabs + (int64_t)w * n + sbase;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const f32x4_t o = {gacc[q][0], gacc[q][1], gacc[q][2], gacc[q][3]};
        *reinterpret_cast<f32x4_t*>(slab + 256 * q + 4 * lane) = o;
      }
      if (tid == 0 && it == a.iters - 1) a.rr_part[w] = rr;     // (every thread holds the same rr)
    }
    if (stamp) a.stamps[w * 8 + 1] = wall_clock64();
    // A does not change: the first panel of the NEXT iteration is requested now and arrives behind the two grid barriers
    // and phase B (its registers are idle there) - the pass restarts without an HBM round trip
    if (it + 1 < a.iters && npanels > 0) issue(0);
    if (!fz_grid_barrier(a.bar, G, gen, a.timeout_ticks, ok_lds)) return;
    if (stamp) a.stamps[w * 8 + 2] = wall_clock64();

    // ---- phase B: the owned columns ------------------------------------------------------------------------------------
    if (own_n > 0) {
      const int c = tid % own, grp = tid / own;                  // 512 / own groups
      const int ngrp = FZ_THREADS / own > 16 ? 16 : FZ_THREADS / own;
      if ((own & 3) == 0 && own_n == own) {
        // The G slab rows of the owned columns, ALL requested at once: a thread takes 4 columns (16 bytes) of every
        // rows_par-th slab, four independent loads per trip (G = 256, own = 32: 64 rows in parallel, one trip) - the slabs
        // were written by other XCDs, every load is a trip to the memory side and a dependent chain of them was the
        // phase's whole cost.  Fixed order: rows -> 16 groups -> column.  The tile is idle here and serves as scratch.
        const int cg = own >> 2, rows_par = FZ_THREADS / cg;
        const int c4 = tid % cg, sr = tid / cg;
        float* scratch = tile;                                   // [rows_par][own] <= 2048 floats
        if (sr < rows_par) {
          const float* base = a.slabs + own_lo + 4 * c4;
          f32x4_t s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
          unsigned sl = (unsigned)sr;
          const unsigned rp = (unsigned)rows_par;
          for (; sl + 3 * rp < G; sl += 4 * rp) {
            const f32x4_t v0 = *reinterpret_cast<const f32x4_t*>(base + (int64_t)sl * n);
            const f32x4_t v1 = *reinterpret_cast<const f32x4_t*>(base + (int64_t)(sl + rp) * n);
            const f32x4_t v2 = *reinterpret_cast<const f32x4_t*>(base + (int64_t)(sl + 2 * rp) * n);
            const f32x4_t v3 = *reinterpret_cast<const f32x4_t*>(base + (int64_t)(sl + 3 * rp) * n);
            s0 += v0; s1 += v1; s2 += v2; s3 += v3;
          }
          for (; sl < G; sl += rp) s0 += *reinterpret_cast<const f32x4_t*>(base + (int64_t)sl * n);
          *reinterpret_cast<f32x4_t*>(scratch + sr * own + 4 * c4) = (s0 + s1) + (s2 + s3);
        }
        __syncthreads();
        if (grp < ngrp) {
          float s = 0.f;
          for (int r = grp; r < rows_par; r += ngrp) s += scratch[r * own + c];
          gsum[grp * FZ_OWN_MAX + c] = s;
        }
      } else if (grp < ngrp && c < own_n) {
        float s = 0.f;
        for (unsigned sl = (unsigned)grp; sl < G; sl += (unsigned)ngrp) s += a.slabs[(int64_t)sl * n + own_lo + c];
        gsum[grp * FZ_OWN_MAX + c] = s;
      }
      __syncthreads();
      if (tid < own_n) {
        double g = 0.0;
        for (int q = 0; q < ngrp; ++q) g += (double)gsum[q * FZ_OWN_MAX + tid];
        const double beta = a.beta[it], beta_next = a.beta[it + 1];
        const double xc = xs_cur[tid], xp = xs_prev[tid];
        const double yk = form_y(xc, xp, beta);
        double gf = g;
        if (a.prox_kind == PROX_L1 && a.alpha2 > 0.0) gf += a.alpha2 * yk;
        const double v = yk - a.tau * gf;
        double xn = a.alpha1 > 0.0 ? soft_threshold(v, a.tau * a.alpha1) : v;
        if (a.prox_kind == PROX_ENET) xn *= 1.0 / (1.0 + a.tau * a.alpha2);
        const double d = xn - xc;
        xs_prev[tid] = xc;
        xs_cur[tid] = xn;
        a.y[own_lo + tid] = (float)form_y(xn, xc, beta_next);
        accs[tid * 4 + 0] = d * d; accs[tid * 4 + 1] = gf * gf; accs[tid * 4 + 2] = fabs(xn); accs[tid * 4 + 3] = xn * xn;
      }
      __syncthreads();
      if (tid < 4) {
        double s = 0.0;
        for (int c2 = 0; c2 < own_n; ++c2) s += accs[c2 * 4 + tid];
        a.part[(((a.k0 + it) & 1) * (long long)G + w) * 4 + tid] = s;
      }
    } else if (tid < 4) {
      a.part[(((a.k0 + it) & 1) * (long long)G + w) * 4 + tid] = 0.0;
    }
    if (stamp) a.stamps[w * 8 + 3] = wall_clock64();
    if (!fz_grid_barrier(a.bar, G, gen, a.timeout_ticks, ok_lds)) return;
    if (stamp) a.stamps[w * 8 + 4] = wall_clock64();
  }
  if (tid < own_n) { a.x_cur[own_lo + tid] = xs_cur[tid]; a.x_prev[own_lo + tid] = xs_prev[tid]; }
}

// dynamic LDS of the kernel for n columns
__host__ __device__ inline size_t fz_lds_bytes(int n) {
  return (size_t)FZ_ROWS * (n + FZ_PAD) * 4 + 2 * FZ_NW * 4 * 4 + 2 * FZ_OWN_MAX * 8 + 16 * FZ_OWN_MAX * 4 + FZ_OWN_MAX * 4 * 8 + 16;
}

}  // namespace fos
