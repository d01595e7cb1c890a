// Slab reduction + prox / momentum epilogue + scalar finalisation for the FISTA family.
//
// Spec (reference is NumPy; these are new kernels):
//   grad += alpha2 * y                     iterative_solvers.py:174-175, :293-294
//   v = y - tau*grad ; x_next = prox_l1    iterative_solvers.py:200-201, :315-316 ; prox_operators.py:8, :15-16
//   this_step / prev_step / ratio          iterative_solvers.py:204-206, :325-327
//   momentum, adaptive restart             iterative_solvers.py:209-221 ; theta_k = k/(k+1+delta) :330-331
//   stopping rules                         iterative_solvers.py:238, :242, :337, :341
//
// The slabs are summed in a fixed order (slab 0,1,2,... inside each group, groups 0..G-1): the result is
// bit-reproducible run to run and identical on every rank of a sharded run after the all-reduce.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include "gemv_pair.hpp"

namespace fos {

enum : int { MODE_FISTA = 0, MODE_DELTA = 1, MODE_ISTA = 2 };
enum : int { PROX_L1 = 0, PROX_ENET = 1 };   // PROX_ENET: l2 term inside the prox (prox_operators.py:10-16)
enum : int { STOP_NONE = 0, STOP_STEP = 1, STOP_RATIO = 2, STOP_GRAD = 3, STOP_LS_STALL = 4 };

// Loop-carried scalars that live on the device so that a run of iterations needs no host round trip.
struct FistaScalars {
  double t_prev;       // FISTA t_{k}
  double beta;         // momentum used to rebuild y_k = x_k + beta (x_k - x_prev)
  double this_step;    // ||x_k - x_{k-1}|| of the last completed iteration
  double prev_step;    // the one before
  double ratio;        // this/prev (inf when prev == 0)
  double rr;           // ||A y - b||^2 of the last gradient
  double gnorm2;       // ||grad||^2 of the last gradient (smooth part, incl. alpha2*y)
  double xnorm1;       // ||x_k||_1   (filled by the update kernel: free by-products for the objective)
  double xnorm2;       // ||x_k||_2^2
  double rr_x;         // ||A x_k - b||^2 of the iterate the last DUAL gradient pass started from
  long long k;         // completed iterations
  int stopped;         // STOP_*
  int restarts;
  double tau;          // device-driven backtracking (fos_fista_run_backtracking): the step, carried across iterations
};

struct FistaParams {
  double alpha1;
  double alpha2;       // smooth l2 weight (added to the gradient) when prox_kind == PROX_L1
  double tau;          // step
  double delta;        // FISTA-delta parameter
  double restart_threshold;
  double tol_step;     // stop when this_step < tol_step   (0 = off)
  double tol_ratio;    // stop when ratio < tol_ratio       (0 = off)
  double tol_grad;     // stop BEFORE the update when ||grad_smooth(y_k)|| < tol_grad   (0 = off)   iterative_solvers.py:179
  int mode;            // MODE_*
  int prox_kind;       // PROX_*
  int adaptive_restart;
  int tau_from_state;  // 1: the update takes the step from FistaScalars::tau (device-driven backtracking), not from `tau`
};

// ---- candidate / multi-lambda block layouts of the matrix-core kernels (batch_trial.hpp, gram_batch.hpp) -----------
constexpr int BT_NV = 16;            // right-hand sides per pass (MFMA N)
// X block layout ("Xp"): for column k and candidate j,
//   Xp[ ((k/16)*4 + (k%16)/4) * 64 + j*4 + (k%4) ]
// i.e. per 16-column subtile a [q = 4][j = 16][c = 4] cube: the float4 a lane needs for one subtile is contiguous and
// the 64 lanes of a wave read 1 KiB contiguous.
__device__ __host__ inline int64_t xp_index(int64_t k, int j) {
  return ((k / 16) * 4 + (k % 16) / 4) * 64 + (int64_t)j * 4 + (k % 4);
}
// bf16 A: candidate block "Xq", three bf16 terms per value (see batch_trial.hpp):
//   Xq[ (((k/32)*3 + p)*4 + (k%32)/8) * 128 + j*8 + (k%8) ]
__device__ __host__ inline int64_t xq_index(int64_t k, int j, int part) {
  return (((k / 32) * 3 + part) * 4 + (k % 32) / 8) * 128 + (int64_t)j * 8 + (k % 8);
}
__device__ inline unsigned short f32_to_bf16_rn(float f) {
  const unsigned u = __float_as_uint(f);
  return (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);      // finite inputs only (differences of iterates)
}
__device__ inline float bf16_to_f32(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
// Where a kernel takes the reduced gradient from: the fp32 buffer of the fp32 pass, or (precise mode: backtracking runs,
// fos_fista_set_precise) the n doubles the fp64-accumulating pass left.
struct GradSrc {
  const float* f32;
  const double* f64;
};
__device__ inline double grad_at(const GradSrc& g, int64_t i) { return g.f64 != nullptr ? g.f64[i] : (double)g.f32[i]; }

// where the update kernel leaves y_{k+1} for the next pass over A
enum : int { YOUT_VECTOR = 0, YOUT_XP = 1, YOUT_XQ = 2 };

__device__ inline double soft_threshold(double v, double thr) {
  double mag = fabs(v) - thr;
  mag = (mag < 0.0) ? 0.0 : mag;
  return v == 0.0 ? 0.0 : copysign(mag, v);
}

__device__ inline float soft_threshold(float v, float thr) {
  // sign(v) * max(|v| - thr, 0)  (prox_operators.py:8): NaN propagates, shrunk negatives give -0.0,
  // np.sign(+-0) = +0 gives +0 whatever thr is.
  float mag = fabsf(v) - thr;
  mag = (mag < 0.0f) ? 0.0f : mag;
  return v == 0.0f ? 0.0f : copysignf(mag, v);
}

// Block-wide sum of NV doubles per thread (256 threads); result valid in thread 0.
template <int NV>
__device__ inline void block_sum_256(double (&v)[NV], double* lds /* NV*4 doubles */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = wave_sum(v[i]);
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) lds[i * 4 + wave] = v[i];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = (lds[i * 4 + 0] + lds[i * 4 + 1]) + (lds[i * 4 + 2] + lds[i * 4 + 3]);
  }
}

// ---------------------------------------------------------------------------------------------------------
// Column-block reduction of the slabs.  Workgroup = 256 threads = QUADS float4 columns x GROUPS slab groups.
// Returns (in the threads with grp == 0) the float4 sum for quad `q`; other threads get garbage.
// ---------------------------------------------------------------------------------------------------------
constexpr int RQ = 16;             // float4 columns per workgroup  -> 64 columns
constexpr int RG = 16;             // slab groups per workgroup
constexpr int RCOLS = RQ * 4;

__device__ inline f32x4 reduce_slab_block(const float* __restrict__ slabs, int nslabs, int n, int col0,
                                          f32x4 (*lds)[RQ], int64_t stride = 0) {
  if (stride == 0) stride = n;       // distance between consecutive slabs (multi-lambda runs interleave NVEC sets)
  // Quads in use by this workgroup: all RQ = 16 for a full 64-column block; for narrow problems (n <= 64 runs in ONE
  // workgroup and may have ~1000 slabs to fold) the idle quads' threads become extra slab groups: nq quads x 256/nq
  // groups.  nq is a power of two, so the lds rows of RQ float4 are simply re-indexed flat.
  int quads = (n - col0 + 3) / 4;
  int nq = RQ;
  while (nq > 1 && (nq >> 1) >= quads) nq >>= 1;
  const int ng = (RQ * RG) / nq;
  const int q = threadIdx.x % nq, grp = threadIdx.x / nq;
  const int col = col0 + 4 * q;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (col < n) {
    const float* p = slabs + col;
    int s = grp;
    // eight loads in flight first (round 3; sixteen measured the same): every trip is a round trip to the memory side (the slabs were written by other
    // XCDs), and a narrow plan's 1024 slabs were 16 dependent trips of four - same order of additions, same sums
    for (; s + 7 * ng < nslabs; s += 8 * ng) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(p + (int64_t)(s + u * ng) * stride);
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; s + 3 * ng < nslabs; s += 4 * ng) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(p + (int64_t)s * stride);
      const f32x4 b = *reinterpret_cast<const f32x4*>(p + (int64_t)(s + ng) * stride);
      const f32x4 c = *reinterpret_cast<const f32x4*>(p + (int64_t)(s + 2 * ng) * stride);
      const f32x4 d = *reinterpret_cast<const f32x4*>(p + (int64_t)(s + 3 * ng) * stride);
      acc += a; acc += b; acc += c; acc += d;
    }
    for (; s < nslabs; s += ng) acc += *reinterpret_cast<const f32x4*>(p + (int64_t)s * stride);
  }
  f32x4* flat = &lds[0][0];
  flat[threadIdx.x] = acc;           // index grp * nq + q
  __syncthreads();
  f32x4 tot = {0.f, 0.f, 0.f, 0.f};
  if (grp == 0) {
    for (int g = 0; g < ng; ++g) tot += flat[g * nq + q];
  }
  return tot;                        // valid in the threads with threadIdx.x < nq (grp == 0), quad q = threadIdx.x
}

// Scalar-tail variant for n % 4 != 0 (fallback path): one column per thread-quad element.
__device__ inline float reduce_slab_scalar(const float* __restrict__ slabs, int nslabs, int n, int col) {
  float acc = 0.f;
  if (col < n)
    for (int s = 0; s < nslabs; ++s) acc += slabs[(int64_t)s * n + col];
  return acc;
}

// slabs -> gbuf[0..n) (+ sum of rr partials -> scal->rr and gbuf[n]).  grid = ceil(n / RCOLS).
template <bool VEC>
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slabs, int nslabs, int n,
                                                         const double* __restrict__ rr_part, int n_rr,
                                                         float* __restrict__ gbuf, double* __restrict__ rr_out,
                                                         const int* stopped, int64_t slab_stride = 0) {
  if (stopped != nullptr && *stopped != 0) return;
  __shared__ f32x4 lds[RG][RQ];
  const int col0 = blockIdx.x * RCOLS;
  if constexpr (VEC) {
    const f32x4 tot = reduce_slab_block(slabs, nslabs, n, col0, lds, slab_stride);
    const int q = threadIdx.x % RQ, grp = threadIdx.x / RQ;
    const int col = col0 + 4 * q;
    if (grp == 0 && col + 3 < n) *reinterpret_cast<f32x4*>(gbuf + col) = tot;
    else if (grp == 0 && col < n) {                  // ragged n: gbuf[n] belongs to the rr fold below
      gbuf[col] = tot.x;
      if (col + 1 < n) gbuf[col + 1] = tot.y;
      if (col + 2 < n) gbuf[col + 2] = tot.z;
    }
  } else {
    if (threadIdx.x < RCOLS) {
      const int col = col0 + threadIdx.x;
      const float v = reduce_slab_scalar(slabs, nslabs, n, col);
      if (col < n) gbuf[col] = v;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < 64) {      // one wave folds the rr partials (fixed order)
    double s = 0.0;
    for (int i = threadIdx.x; i < n_rr; i += 64) s += rr_part[i];
    s = wave_sum(s);
    if (threadIdx.x == 0) {
      if (rr_out != nullptr) *rr_out = s;
      gbuf[n] = (float)s;     // rides along with the gradient in the multi-GPU all-reduce
    }
  }
}

// fp64 slabs (fos_gemv_pair_dd, the L-BFGS fg): out[j] = sum_s slabs[s][j] + alpha2*y[j] in fp64, fixed order;
// out[n] = sum of the rr partials.  The slabs are nslabs x stride doubles (stride >= n) - 512 x 8192 x 8 B = 32 MiB at
// config 3 - so the sum is spread wide: a workgroup owns 64 columns, 512 threads = 32 column pairs (16-byte loads) x 16
// slab groups, each group sums every 16th slab and the groups are added in index order.  grid = ceil(n / 64).
constexpr int SRD_GROUPS = 16, SRD_COLS = 64, SRD_THREADS = SRD_GROUPS * SRD_COLS / 2;
static __global__ __launch_bounds__(SRD_THREADS) void slab_reduce_dd_kernel(const double* __restrict__ slabs, int nslabs, int n,
                                                                    int64_t stride, const double* __restrict__ rr_part,
                                                                    int n_rr, double alpha2, const double* __restrict__ y,
                                                                    double* __restrict__ out, const int* stopped = nullptr) {
  if (stopped != nullptr && *stopped != 0) return;
  __shared__ double lds[SRD_GROUPS][SRD_COLS];
  const int pair = threadIdx.x & (SRD_COLS / 2 - 1), grp = threadIdx.x / (SRD_COLS / 2);
  const int col = blockIdx.x * SRD_COLS + 2 * pair;
  double a0 = 0.0, a1 = 0.0;
  if (col + 1 < n && (stride & 1) == 0) {
#pragma unroll 4
    for (int s = grp; s < nslabs; s += SRD_GROUPS) {
      const f64x2 v = *reinterpret_cast<const f64x2*>(slabs + (int64_t)s * stride + col);
      a0 += v.x; a1 += v.y;
    }
  } else if (col < n) {
    for (int s = grp; s < nslabs; s += SRD_GROUPS) {
      a0 += slabs[(int64_t)s * stride + col];
      if (col + 1 < n) a1 += slabs[(int64_t)s * stride + col + 1];
    }
  }
  lds[grp][2 * pair] = a0;
  lds[grp][2 * pair + 1] = a1;
  __syncthreads();
  if (threadIdx.x < SRD_COLS) {
    const int c = blockIdx.x * SRD_COLS + threadIdx.x;
    if (c < n) {
      double tot = lds[0][threadIdx.x];
#pragma unroll
      for (int g = 1; g < SRD_GROUPS; ++g) tot += lds[g][threadIdx.x];
      if (alpha2 != 0.0) tot += alpha2 * y[c];
      out[c] = tot;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x >= SRD_THREADS - 64) {      // the last wave folds the rr partials (fixed order)
    const int l = threadIdx.x - (SRD_THREADS - 64);
    double s = 0.0;
    for (int i = l; i < n_rr; i += 64) s += rr_part[i];
    s = wave_sum(s);
    if (l == 0) out[n] = s;
  }
}

// ---------------------------------------------------------------------------------------------------------
// FISTA update for RCOLS columns per workgroup.  FROM_SLABS: sum the slabs first (single-GPU fused path);
// otherwise take the (all-reduced) gradient from gbuf.  Writes x_prev <- x_k, x_k <- x_next in place and
// per-workgroup partial sums  part[wg] = {sum d^2, sum g_full^2, sum |x_next|, sum x_next^2}.
// ---------------------------------------------------------------------------------------------------------
template <bool FROM_SLABS, bool VEC>
__device__ inline void fista_update_body(const float* __restrict__ slabs, int nslabs,
                                                          GradSrc gsrc, int n,
                                                          double* __restrict__ x_cur, double* __restrict__ x_prev,
                                                          const FistaScalars* __restrict__ scal, FistaParams prm,
                                                          double* __restrict__ part, int host_beta, double beta_val,
                                                          double* __restrict__ x_hist,
                                                          float* __restrict__ y_next, double beta_next,
                                                          int64_t slab_stride, int y_mode, int y_slot) {
  if (scal->stopped != 0) return;
  __shared__ f32x4 lds[RG][RQ];
  __shared__ double dl[4 * 4];
  const int col0 = blockIdx.x * RCOLS;
  const double beta = host_beta ? beta_val : scal->beta;
  double g[4] = {0.0, 0.0, 0.0, 0.0};
  int col, cnt = 0;
  bool owner;
  if constexpr (VEC) {
    const int q = threadIdx.x % RQ, grp = threadIdx.x / RQ;
    col = col0 + 4 * q;
    owner = (grp == 0) && (col < n);
    if constexpr (FROM_SLABS) {
      const f32x4 tot = reduce_slab_block(slabs, nslabs, n, col0, lds, slab_stride);
      g[0] = tot.x; g[1] = tot.y; g[2] = tot.z; g[3] = tot.w;
    } else if (owner) {
#pragma unroll
      for (int e = 0; e < 4; ++e) g[e] = col + e < n ? grad_at(gsrc, col + e) : 0.0;
    }
    cnt = owner ? (n - col < 4 ? n - col : 4) : 0;      // ragged n (padded slab rows): the last quad is partial
  } else {
    col = col0 + threadIdx.x;
    owner = (threadIdx.x < RCOLS) && (col < n);
    if (owner) g[0] = FROM_SLABS ? (double)reduce_slab_scalar(slabs, nslabs, n, col) : grad_at(gsrc, col);
    cnt = owner ? 1 : 0;
  }
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  const double tau = prm.tau_from_state ? scal->tau : prm.tau;
  const double thr = tau * prm.alpha1;
  const double shrink = 1.0 / (1.0 + tau * prm.alpha2);
  for (int e = 0; e < cnt; ++e) {
    const double xc = x_cur[col + e], xp = x_prev[col + e];
    const double y = form_y(xc, xp, beta);
    double gf = g[e];
    if (prm.prox_kind == PROX_L1 && prm.alpha2 > 0.0) gf += prm.alpha2 * y;
    const double v = y - tau * gf;
    double xn = prm.alpha1 > 0.0 ? soft_threshold(v, thr) : v;
    if (prm.prox_kind == PROX_ENET) xn *= shrink;
    const double d = xn - xc;
    acc[0] += d * d;
    acc[1] += gf * gf;
    acc[2] += fabs(xn);
    acc[3] += xn * xn;
    x_prev[col + e] = xc;
    x_cur[col + e] = xn;
    if (x_hist != nullptr) x_hist[col + e] = xn;      // device-resident history (fos_fista_run_history)
    // plain runs: beta_{k+1} is known, hand the next A pass its y as ONE fp32 vector (same form_y, same rounding)
    if (y_next != nullptr) {
      const float yn = (float)form_y(xn, xc, beta_next);
      if (y_mode == YOUT_VECTOR) {
        y_next[col + e] = yn;
      } else if (y_mode == YOUT_XP) {                  // slot of the multi-lambda block (gram_batch.hpp)
        y_next[xp_index(col + e, y_slot)] = yn;
      } else {                                         // bf16 A: three bf16 terms
        unsigned short* xq = reinterpret_cast<unsigned short*>(y_next);
        const unsigned short hi = f32_to_bf16_rn(yn);
        const float r1 = yn - bf16_to_f32(hi);
        const unsigned short mid = f32_to_bf16_rn(r1);
        xq[xq_index(col + e, y_slot, 0)] = hi;
        xq[xq_index(col + e, y_slot, 1)] = mid;
        xq[xq_index(col + e, y_slot, 2)] = f32_to_bf16_rn(r1 - bf16_to_f32(mid));
      }
    }
  }
  block_sum_256<4>(acc, dl);
  if (threadIdx.x == 0) {
    part[blockIdx.x * 4 + 0] = acc[0]; part[blockIdx.x * 4 + 1] = acc[1];
    part[blockIdx.x * 4 + 2] = acc[2]; part[blockIdx.x * 4 + 3] = acc[3];
  }
}


template <bool FROM_SLABS, bool VEC>
__global__ __launch_bounds__(256) void fista_update_kernel(const float* __restrict__ slabs, int nslabs,
                                                          GradSrc gsrc, int n,
                                                          double* __restrict__ x_cur, double* __restrict__ x_prev,
                                                          const FistaScalars* __restrict__ scal, FistaParams prm,
                                                          double* __restrict__ part, int host_beta, double beta_val,
                                                          double* __restrict__ x_hist = nullptr,
                                                          float* __restrict__ y_next = nullptr, double beta_next = 0.0,
                                                          int64_t slab_stride = 0, int y_mode = YOUT_VECTOR,
                                                          int y_slot = 0) {
  fista_update_body<FROM_SLABS, VEC>(slabs, nslabs, gsrc, n, x_cur, x_prev, scal, prm, part, host_beta, beta_val, x_hist,
                                     y_next, beta_next, slab_stride, y_mode, y_slot);
}

// The updates of up to 16 lockstep state machines in ONE launch (multi-lambda pass, gram_batch.hpp): blockIdx.y selects
// the state machine; its slab set is slabs + y*n with slab stride 16*n, its y_{k+1} goes to slot y of the block.
struct MultiUpdate {
  double* x_cur[BT_NV];
  double* x_prev[BT_NV];
  const FistaScalars* scal[BT_NV];
  double* part[BT_NV];
  double beta[BT_NV], beta_next[BT_NV];
  double alpha1[BT_NV], alpha2[BT_NV], tau[BT_NV];
};
// host_beta = 0 (controlled runs): beta comes from each state machine's device scalars and no y is written (beta_{k+1} is
// decided by fista_finalize_multi_kernel behind this launch; form_y_multi_kernel then fills the block).
static __global__ __launch_bounds__(256) void fista_update_multi_kernel(const float* __restrict__ slabs, int nslabs, int n,
                                                                MultiUpdate mu, FistaParams prm0, float* __restrict__ y_block,
                                                                int y_mode, int host_beta = 1) {
  const int v = blockIdx.y;
  FistaParams prm = prm0;                       // mode / prox kind / delta are common to the path; weights and steps are not
  prm.alpha1 = mu.alpha1[v]; prm.alpha2 = mu.alpha2[v]; prm.tau = mu.tau[v];
  fista_update_body<true, true>(slabs + (int64_t)v * n, nslabs, GradSrc{nullptr, nullptr}, n, mu.x_cur[v], mu.x_prev[v], mu.scal[v], prm,
                                mu.part[v], host_beta, mu.beta[v], nullptr, host_beta ? y_block : nullptr, mu.beta_next[v],
                                (int64_t)BT_NV * n, y_mode, v);
}

// One wave: fold the partials and advance the scalar state.  iterative_solvers.py:204-221, :235-242, :325-342.
__device__ inline void fista_finalize_body(const double* __restrict__ part, int nparts,
                                           const double* __restrict__ rr_part, int n_rr,
                                           FistaScalars* __restrict__ scal, const FistaParams& prm,
                                           double* __restrict__ hist_row) {
  if (scal->stopped != 0) return;
  // issue every load before the first use: the partials were written by other CUs (L2 / MALL latency each)
  double s[4] = {0.0, 0.0, 0.0, 0.0};
  double rr = 0.0;
  constexpr int UNR = 4;
  for (int base = 0; base < nparts; base += 64 * UNR) {
    double v[UNR][4];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int i = base + u * 64 + threadIdx.x;
#pragma unroll
      for (int j = 0; j < 4; ++j) v[u][j] = i < nparts ? part[i * 4 + j] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j) s[j] += v[u][j];
  }
  for (int base = 0; base < n_rr; base += 64 * UNR) {
    double v[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int i = base + u * 64 + threadIdx.x;
      v[u] = i < n_rr ? rr_part[i] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) rr += v[u];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) s[i] = wave_sum(s[i]);
  rr = wave_sum(rr);
  if (threadIdx.x != 0) return;
  const double step = sqrt(s[0]);
  const double prev = scal->this_step;
  const double ratio = prev > 0.0 ? step / prev : INFINITY;
  double beta = 0.0;
  if (prm.mode == MODE_FISTA) {
    const double t = scal->t_prev;
    double t_new;
    if (prm.adaptive_restart && ratio > prm.restart_threshold) {
      t_new = 1.0;
      beta = 0.0;
      scal->restarts += 1;
    } else {
      t_new = 0.5 * (1.0 + sqrt(1.0 + 4.0 * t * t));
      beta = (t - 1.0) / t_new;
    }
    scal->t_prev = t_new;
  } else if (prm.mode == MODE_DELTA) {
    const double kk = (double)(scal->k + 1);
    beta = kk / (kk + 1.0 + prm.delta);
  }
  scal->beta = beta;
  scal->prev_step = prev;
  scal->this_step = step;
  scal->ratio = ratio;
  scal->gnorm2 = s[1];
  scal->xnorm1 = s[2];
  scal->xnorm2 = s[3];
  if (n_rr > 0) scal->rr = rr;
  scal->k += 1;
  if (hist_row != nullptr) {        // device-recorded history (fos_fista_run_recorded): { -, ||x||_1, ||x||_2^2, ||dx||^2 }
    hist_row[1] = s[2]; hist_row[2] = s[3]; hist_row[3] = s[0];
  }
  int stop = STOP_NONE;
  if (prm.tol_step > 0.0 && step < prm.tol_step) stop = STOP_STEP;
  if (stop == STOP_NONE && prm.tol_ratio > 0.0 && ratio < prm.tol_ratio) stop = STOP_RATIO;
  scal->stopped = stop;
}
static __global__ __launch_bounds__(64) void fista_finalize_kernel(const double* __restrict__ part, int nparts,
                                                           const double* __restrict__ rr_part, int n_rr,
                                                           FistaScalars* __restrict__ scal, FistaParams prm,
                                                           double* __restrict__ hist_row = nullptr) {
  fista_finalize_body(part, nparts, rr_part, n_rr, scal, prm, hist_row);
}

// Lockstep weights with data-dependent control (adaptive restart, step / ratio tolerances per weight): the scalar
// bookkeeping of ALL state machines in one launch (blockIdx.x = state machine), then their y_{k+1} into the Y block of the
// next matrix-core product.  A stopped state machine is a masked column: its update, bookkeeping and y are no-ops and its
// column of the block keeps the last y (the products still carry it along; nothing reads its gradient).
struct MultiControl {
  FistaScalars* scal[BT_NV];
  const double* part[BT_NV];
  const double* x_cur[BT_NV];
  const double* x_prev[BT_NV];
  int adaptive_restart[BT_NV];
  double restart_threshold[BT_NV], tol_step[BT_NV], tol_ratio[BT_NV];
};
static __global__ __launch_bounds__(64) void fista_finalize_multi_kernel(MultiControl mc, int nparts, FistaParams prm0) {
  const int v = blockIdx.x;
  FistaParams prm = prm0;
  prm.adaptive_restart = mc.adaptive_restart[v];
  prm.restart_threshold = mc.restart_threshold[v];
  prm.tol_step = mc.tol_step[v];
  prm.tol_ratio = mc.tol_ratio[v];
  fista_finalize_body(mc.part[v], nparts, nullptr, 0, mc.scal[v], prm, nullptr);
}
// force: also write the columns of stopped state machines (the block's first fill)
static __global__ __launch_bounds__(256) void form_y_multi_kernel(MultiControl mc, int n, float* __restrict__ y_block, int y_mode,
                                                          int force) {
  const int v = blockIdx.y;
  if (!force && mc.scal[v]->stopped != 0) return;
  const double beta = mc.scal[v]->beta;
  for (int col = blockIdx.x * 256 + threadIdx.x; col < n; col += gridDim.x * 256) {
    const float yn = (float)form_y(mc.x_cur[v][col], mc.x_prev[v][col], beta);
    if (y_mode == YOUT_XP) {
      y_block[xp_index(col, v)] = yn;
    } else {
      unsigned short* xq = reinterpret_cast<unsigned short*>(y_block);
      const unsigned short hi = f32_to_bf16_rn(yn);
      const float r1 = yn - bf16_to_f32(hi);
      const unsigned short mid = f32_to_bf16_rn(r1);
      xq[xq_index(col, v, 0)] = hi;
      xq[xq_index(col, v, 1)] = mid;
      xq[xq_index(col, v, 2)] = f32_to_bf16_rn(r1 - bf16_to_f32(mid));
    }
  }
}

// Fresh loop state (fos_fista_reset): t = 1, ratio = inf, everything else 0 - written by the device so that the reset
// only enqueues.
static __global__ void fista_init_scalars_kernel(FistaScalars* __restrict__ scal) {
  FistaScalars z{};
  z.t_prev = 1.0;
  z.ratio = INFINITY;
  *scal = z;
}

// Gradient-norm stop (iterative_solvers.py:179: `if tol > 0 and ||grad|| < tol: break`, checked BEFORE the update, grad of
// the smooth part at y_k incl. alpha2*y) on the device, so that fista(tol > 0) stays enqueue-only: one workgroup reads the
// reduced gradient in gbuf and raises the stop flag; the update and finalize kernels behind it are then no-ops.
// partial_out != nullptr (column-sharded problems): only this rank's sum of squares is written; after the sum over the
// ranks grad_norm_decide_kernel takes the decision.
static __global__ __launch_bounds__(1024) void grad_norm_stop_kernel(GradSrc gsrc, int n,
                                                             const double* __restrict__ x_cur,
                                                             const double* __restrict__ x_prev,
                                                             FistaScalars* __restrict__ scal, FistaParams prm,
                                                             double* __restrict__ partial_out = nullptr) {
  // (with partial_out the partial is re-derived even after a stop: it is all-reduced in place behind this kernel)
  if (scal->stopped != 0 && partial_out == nullptr) return;
  __shared__ double ws[16];
  const double beta = scal->beta;
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += 1024) {
    double gf = grad_at(gsrc, i);
    if (prm.prox_kind == PROX_L1 && prm.alpha2 > 0.0) gf += prm.alpha2 * form_y(x_cur[i], x_prev[i], beta);
    acc += gf * gf;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int i = 0; i < 16; ++i) s += ws[i];
    if (partial_out != nullptr) { *partial_out = s; return; }
    scal->gnorm2 = s;
    if (sqrt(s) < prm.tol_grad) scal->stopped = STOP_GRAD;
  }
}
static __global__ void grad_norm_decide_kernel(const double* __restrict__ total, FistaScalars* __restrict__ scal, FistaParams prm) {
  if (scal->stopped != 0) return;
  scal->gnorm2 = *total;
  if (sqrt(*total) < prm.tol_grad) scal->stopped = STOP_GRAD;
}

// Device-side Armijo decision (iterative_solvers.py:187-195 for the candidates t, t*eta, ... of one matrix-core batch):
// the first candidate that passes  (1-C) grad.dlt + 0.5||A dlt||^2 + 0.5 a2 ||dlt||^2 <= noise  (the host's
// _armijo_accepts, same clauses) becomes the step; FistaScalars::tau carries it to the update kernel and to the next
// iteration (tau persists, :197).  No candidate of the batch accepted: the flag STOP_LS_STALL parks the pipeline - every
// later kernel is a no-op - until the host finishes this search (rare: the reference's own step-underflow regime).
// bt: folded sums of the candidate kernel { gd_j (16), dd_j (16), nnz_j (16), ||grad||^2, ||y||^2 }, q = bt + 64.
static __global__ void armijo_decide_kernel(const double* __restrict__ bt, FistaScalars* __restrict__ scal, FistaParams prm,
                                     double eta, double armijo_c, double grad_eps, int nv, int* __restrict__ ls_out,
                                     double* __restrict__ tau_hist, long long slot) {
  if (scal->stopped != 0) return;
  const double a2s = (prm.prox_kind == PROX_L1 && prm.alpha2 > 0.0) ? prm.alpha2 : 0.0;
  const double gn2 = bt[3 * BT_NV], y2 = bt[3 * BT_NV + 1];
  const double g_y = 0.5 * scal->rr + 0.5 * a2s * y2;
  double t = scal->tau;
  int j = 0;
  bool accepted = false;
  for (; j < nv; ++j) {
    const double gd = bt[j], dd = bt[BT_NV + j], nnz = bt[2 * BT_NV + j], q = bt[64 + j];
    const double excess = (1.0 - armijo_c) * gd + 0.5 * q + 0.5 * a2s * dd;
    const double noise = fmax(2.220446049250313e-16 * g_y, grad_eps * sqrt(gn2 * dd));
    const bool ball = dd <= (grad_eps * t) * (grad_eps * t) * gn2;
    if (nnz == 0.0 || excess <= noise || ball) { accepted = true; break; }
    t *= eta;
  }
  scal->tau = t;
  if (accepted) {
    if (ls_out != nullptr) ls_out[slot] = j;
    if (tau_hist != nullptr) tau_hist[slot] = t;
  } else {
    scal->stopped = STOP_LS_STALL;
  }
}

static __global__ void set_state_tau_kernel(FistaScalars* __restrict__ scal, double tau) { scal->tau = tau; }
// ||A x_k - b||^2 seen by this iteration's gradient pass (the iterate BEFORE the update) -> its slot of the record
static __global__ void record_rr_x_kernel(const FistaScalars* __restrict__ scal, double* __restrict__ slot) {
  if (scal->stopped != 0) return;
  *slot = scal->rr_x;
}
static __global__ void clear_stall_kernel(FistaScalars* __restrict__ scal) {
  if (scal->stopped == STOP_LS_STALL) scal->stopped = STOP_NONE;
}

// y = (float)(x_cur + beta (x_cur - x_prev)) as one fp32 vector (entry of a lockstep multi-lambda run).
static __global__ __launch_bounds__(256) void form_y_kernel(const double* __restrict__ x_cur, const double* __restrict__ x_prev,
                                                     double beta, float* __restrict__ y, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    y[i] = (float)form_y(x_cur[i], x_prev[i], beta);
}

// Plain runs (no adaptive restart, no stopping tolerance): t_k and beta_k do not depend on the data, the host hands
// beta_k to the kernels by value and this kernel runs ONCE per fos_fista_run call to bring the device scalars up to
// date: step norms from the partials of the last two iterations, momentum scalars from the host.
static __global__ __launch_bounds__(64) void fista_finalize_plain_kernel(const double* __restrict__ part_cur,
                                                                 const double* __restrict__ part_prev, int nparts,
                                                                 const double* __restrict__ rr_part, int n_rr,
                                                                 FistaScalars* __restrict__ scal, double t_prev,
                                                                 double beta_next, long long k_total) {
  if (scal->stopped != 0) return;
  double s[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
  for (int i = threadIdx.x; i < nparts; i += 64) {
#pragma unroll
    for (int j = 0; j < 4; ++j) s[j] += part_cur[i * 4 + j];
    if (part_prev != nullptr) s[4] += part_prev[i * 4];
  }
  double rr = 0.0;
  for (int i = threadIdx.x; i < n_rr; i += 64) rr += rr_part[i];
#pragma unroll
  for (int i = 0; i < 5; ++i) s[i] = wave_sum(s[i]);
  rr = wave_sum(rr);
  if (threadIdx.x != 0) return;
  const double step = sqrt(s[0]);
  const double prev = part_prev != nullptr ? sqrt(s[4]) : scal->this_step;
  scal->prev_step = prev;
  scal->this_step = step;
  scal->ratio = prev > 0.0 ? step / prev : INFINITY;
  scal->gnorm2 = s[1];
  scal->xnorm1 = s[2];
  scal->xnorm2 = s[3];
  if (n_rr > 0) scal->rr = rr;
  scal->t_prev = t_prev;
  scal->beta = beta_next;
  scal->k = k_total;
}

// History fold (fos_fista_run_history): block i turns the raw partials of iteration i into
// hist[i] = { ||A x_{i+1} - b||^2, ||x_{i+1}||_1, ||x_{i+1}||_2^2, ||x_{i+1} - x_i||^2 }.
// rr2 partials of slot i+1 belong to the iterate produced by iteration i (slot `iters` = the closing residual pass).
static __global__ __launch_bounds__(64) void history_fold_kernel(const double* __restrict__ rr2_slots, int n_rr,
                                                         const double* __restrict__ part_slots, int nparts,
                                                         double* __restrict__ hist) {
  const int i = blockIdx.x;
  const double* rr2 = rr2_slots + (int64_t)(i + 1) * n_rr;
  const double* part = part_slots + (int64_t)i * nparts * 4;
  double rr = 0.0, s[4] = {0.0, 0.0, 0.0, 0.0};
  for (int j = threadIdx.x; j < n_rr; j += 64) rr += rr2[j];
  for (int j = threadIdx.x; j < nparts; j += 64) {
#pragma unroll
    for (int c = 0; c < 4; ++c) s[c] += part[j * 4 + c];
  }
  rr = wave_sum(rr);
#pragma unroll
  for (int c = 0; c < 4; ++c) s[c] = wave_sum(s[c]);
  if (threadIdx.x == 0) {
    hist[i * 4 + 0] = rr;
    hist[i * 4 + 1] = s[2];
    hist[i * 4 + 2] = s[3];
    hist[i * 4 + 3] = s[0];
  }
}

// ---------------------------------------------------------------------------------------------------------
// Backtracking trial (iterative_solvers.py:187-191) in cancellation-free form.
//
// The reference tests  g(x_tmp) <= g(y) + C * grad.(x_tmp - y)  by evaluating g twice and subtracting two nearly
// equal numbers.  g is quadratic, so with  dlt = x_tmp - y  the same test is, exactly,
//        (1 - C) * grad.dlt + 0.5 * ||A dlt||^2 + 0.5 * alpha2 * ||dlt||^2  <=  0 .
// This kernel forms x_tmp = prox(y - t*grad) in fp64 and writes only dlt (as fp32: dlt keeps full RELATIVE
// precision however small it is, which x_tmp = y + dlt would lose); ||A dlt||^2 is one residual pass (K5, b = 0).
// out per workgroup: { grad.dlt, ||dlt||^2, #(dlt != 0), ||grad||^2, ||y||^2 }, grad including alpha2*y.
// ---------------------------------------------------------------------------------------------------------
constexpr int TRIAL_W = 5;
static __global__ __launch_bounds__(256) void fista_trial_kernel(GradSrc gsrc, int n,
                                                         const double* __restrict__ x_cur,
                                                         const double* __restrict__ x_prev,
                                                         const FistaScalars* __restrict__ scal, FistaParams prm,
                                                         double t_trial, float* __restrict__ dlt_out,
                                                         double* __restrict__ part) {
  __shared__ double dl[TRIAL_W * 4];
  const double beta = scal->beta;
  const double thr = t_trial * prm.alpha1;
  const double shrink = 1.0 / (1.0 + t_trial * prm.alpha2);
  double acc[TRIAL_W] = {0.0, 0.0, 0.0, 0.0, 0.0};
  for (int col = blockIdx.x * 256 + threadIdx.x; col < n; col += gridDim.x * 256) {
    const double y = form_y(x_cur[col], x_prev[col], beta);
    double gf = grad_at(gsrc, col);
    if (prm.prox_kind == PROX_L1 && prm.alpha2 > 0.0) gf += prm.alpha2 * y;
    const double v = y - t_trial * gf;
    double xt = prm.alpha1 > 0.0 ? soft_threshold(v, thr) : v;
    if (prm.prox_kind == PROX_ENET) xt *= shrink;
    const double d = xt - y;
    dlt_out[col] = (float)d;
    acc[0] += gf * d;
    acc[1] += d * d;
    acc[2] += (d != 0.0) ? 1.0 : 0.0;
    acc[3] += gf * gf;
    acc[4] += y * y;
  }
  block_sum_256<TRIAL_W>(acc, dl);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < TRIAL_W; ++i) part[blockIdx.x * TRIAL_W + i] = acc[i];
  }
}

// Fold `nparts` rows of `width` doubles (fixed order) into out[width].  One workgroup of 16 waves: wave v folds
// the columns v, v+16, ... (lanes stride over the rows, butterfly at the end) - deterministic.
constexpr int FOLD_THREADS = 1024;
static __global__ __launch_bounds__(FOLD_THREADS) void fold_partials_kernel(const double* __restrict__ part, int nparts,
                                                                    int width, double* __restrict__ out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int w = wave; w < width; w += FOLD_THREADS / 64) {
    double s = 0.0;
    for (int i = lane; i < nparts; i += 64) s += part[(int64_t)i * width + w];
    s = wave_sum(s);
    if (lane == 0) out[w] = s;
  }
}

// ---------------------------------------------------------------------------------------------------------
// Stand-alone prox kernels (prox_operators.py:3-8, :10-16) for the ista() callable path.
// ---------------------------------------------------------------------------------------------------------
static __global__ __launch_bounds__(256) void prox_l1_kernel(const float* __restrict__ v, float thr, float* __restrict__ out,
                                                     int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    out[i] = soft_threshold(v[i], thr);
}
// per-element threshold (the reference's prox_l1 broadcasts an array-valued tau: prox_operators.py:8)
static __global__ __launch_bounds__(256) void prox_l1_vec_kernel(const float* __restrict__ v, const float* __restrict__ thr,
                                                         float* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    out[i] = soft_threshold(v[i], thr[i]);
}
static __global__ __launch_bounds__(256) void prox_enet_kernel(const float* __restrict__ v, float tau, float a1, float a2,
                                                       float* __restrict__ out, int64_t n) {
  const float thr = tau * a1, inv = 1.0f + tau * a2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    out[i] = soft_threshold(v[i], thr) / inv;
}

// per-element tau (the reference's expression broadcasts an array-valued tau: prox_operators.py:15-16)
static __global__ __launch_bounds__(256) void prox_enet_vec_kernel(const float* __restrict__ v, const float* __restrict__ tau,
                                                           float a1, float a2, float* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    out[i] = soft_threshold(v[i], tau[i] * a1) / (1.0f + tau[i] * a2);
}

// ---------------------------------------------------------------------------------------------------------
// Power iteration tail (iterative_solvers.py:55-56): L = ||w||, v = w / L.  One workgroup.
// ---------------------------------------------------------------------------------------------------------
static __global__ __launch_bounds__(1024) void power_normalize_kernel(const float* __restrict__ w, int n,
                                                              float* __restrict__ v, double* __restrict__ L_out) {
  __shared__ double ws[16];
  __shared__ double Ls;
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += 1024) acc += (double)w[i] * (double)w[i];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int i = 0; i < 16; ++i) s += ws[i];
    Ls = sqrt(s);
    *L_out = Ls;
  }
  __syncthreads();
  const double L = Ls;
  for (int i = threadIdx.x; i < n; i += 1024) v[i] = (float)((double)w[i] / L);
}

}  // namespace fos
