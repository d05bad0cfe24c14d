// Batched small dense factorizations for the chi-truncation step (gfx950).
//
//  * chol_upper_kernel : R^T R = G (f64 Gram of the carried block, cols x cols), semi-definite
//    safe; replaces the R factor of qlten::QR at bmps_impl.h:821 (only R is ever needed, see
//    DESIGN.md "Q-less absorption").
//  * jacobi_rows_kernel: one-sided (Hestenes) Jacobi that orthogonalises the ROWS of
//    M = R_i T_i in place; the rotated rows are sigma_k v_k^T, so the right singular vectors the
//    reference takes from qlten::SVD (bmps_impl.h:235-238) come out without accumulating any
//    rotation matrix.
//  * select_rows_kernel: norms, rank by counting, keep the chi largest, normalise -> Vt, S.
//  * normalize_kernel  : x /= |x|, logscale += log|x| (the reference never normalises and
//    relies on fp64 range, monte_carlo_engine.h:206-240; fp32 needs the log-scale).
#pragma once
#include "common.h"
#include "cplx.h"

namespace pepsgpu {

template <typename T> struct Eps;
template <> struct Eps<float> { static constexpr float v = 5.9604645e-8f; };
template <> struct Eps<double> { static constexpr double v = 1.1102230246251565e-16; };
template <typename R> struct Eps<cplx<R>> { static constexpr R v = Eps<R>::v; };
// Error-budget experiments (scripts/error_budget.py): the float64 engine can run with the noise floors of the float32 one
// (PEPSGPU_F64_EPS=<eps> at context creation sets this device global; default = the float64 epsilon, i.e. no change).
// PEPSGPU_F32_EPS: the same instrument the other way round -- the factor kernels of the float32 engine with lower (or higher) floors.
__device__ double g_eps64_rt = 1.1102230246251565e-16;
__device__ double g_eps32_rt = 5.9604644775390625e-8;
template <typename T> __device__ __forceinline__ double eps_rt() { return (double)Eps<T>::v; }
template <> __device__ __forceinline__ double eps_rt<float>() { return g_eps32_rt; }
template <> __device__ __forceinline__ double eps_rt<double>() { return g_eps64_rt; }

// 1 / x and 1 / sqrt(x) in float64 from the hardware estimates + three Newton steps: a few dependent instructions where
// the IEEE division / square root sequences are ~50 each -- on the critical path of every pivot step and every Jacobi pair (round 6)
// (three steps: with two, a rank-20 Gram matrix kept a 21st pivot above the 2.3e-13 threshold -- the estimates of this part are
// coarser than the 2^-26 the blocked Cholesky of round 3 assumed; e -> 1.5 e^2 per step reaches 1e-16 from 2^-10 in three)
__device__ __forceinline__ double jr_rcp64(const double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = r * (2.0 - x * r);
  r = r * (2.0 - x * r);
  return r * (2.0 - x * r);
}
__device__ __forceinline__ double jr_rsq64(const double x) {
  double r = __builtin_amdgcn_rsq(x);
  r = r * (1.5 - 0.5 * x * r * r);
  r = r * (1.5 - 0.5 * x * r * r);
  return r * (1.5 - 0.5 * x * r * r);
}

// 64-lane all-reductions on the DPP / permlane-swap path (a few cycles of latency per step, no LDS crossbar): the step
// loops of the factor kernels are chains of dependent reductions, where the ds_bpermute behind __shfl_xor costs most
template <int CTRL>
__device__ __forceinline__ int lw_dpp(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }   // (old unused: see jr_dpp_add)
__device__ __forceinline__ int wave_min_dpp(int v) {
  v = min(v, lw_dpp<0xB1>(v));     // quad_perm [1,0,3,2]
  v = min(v, lw_dpp<0x4E>(v));     // quad_perm [2,3,0,1]
  v = min(v, lw_dpp<0x141>(v));    // row_half_mirror
  v = min(v, lw_dpp<0x140>(v));    // row_mirror
  {
    const auto p = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
    v = min((int)p[0], (int)p[1]);
  }
  {
    const auto p = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
    v = min((int)p[0], (int)p[1]);
  }
  return v;
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v += __builtin_bit_cast(float, lw_dpp<0xB1>(__builtin_bit_cast(int, v)));
  v += __builtin_bit_cast(float, lw_dpp<0x4E>(__builtin_bit_cast(int, v)));
  v += __builtin_bit_cast(float, lw_dpp<0x141>(__builtin_bit_cast(int, v)));
  v += __builtin_bit_cast(float, lw_dpp<0x140>(__builtin_bit_cast(int, v)));
  {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto p = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    v = __builtin_bit_cast(float, (unsigned)p[0]) + __builtin_bit_cast(float, (unsigned)p[1]);
  }
  {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto p = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    v = __builtin_bit_cast(float, (unsigned)p[0]) + __builtin_bit_cast(float, (unsigned)p[1]);
  }
  return v;
}

// the same for f64: both halves of the value travel through the same DPP / swap step
template <int CTRL>
__device__ __forceinline__ double lw_dpp_f64(double v) {
  const int lo = lw_dpp<CTRL>(__double2loint(v)), hi = lw_dpp<CTRL>(__double2hiint(v));
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_dpp(double v) {
  v += lw_dpp_f64<0xB1>(v);
  v += lw_dpp_f64<0x4E>(v);
  v += lw_dpp_f64<0x141>(v);
  v += lw_dpp_f64<0x140>(v);
  {
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const auto pl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto ph = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    v = __hiloint2double((int)ph[0], (int)pl[0]) + __hiloint2double((int)ph[1], (int)pl[1]);
  }
  {
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const auto pl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto ph = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    v = __hiloint2double((int)ph[0], (int)pl[0]) + __hiloint2double((int)ph[1], (int)pl[1]);
  }
  return v;
}

// sum over the 64 lanes of a wave, result in every lane
template <typename T>
__device__ __forceinline__ T wave_sum(T v) { return wave_sum_dpp(v); }

// ---------------------------------------------------------------------------------------------
// In-place upper Cholesky of the symmetric PSD Gram matrix G (n x n, row-major, f64) and rank
// compaction of the factor.  R^T R = G; a pivot below max(n*eps64, (NOISE_C*eps_T)^2) * max(diag)
// zeroes its row: that direction carries less weight than the rounding of the T-typed data the
// Gram was built from.  The left-looking update only visits live rows, so the cost follows the
// numerical rank.  On exit the live rows (norm above NOISE_C*eps_T*|R|_F), in their original
// order, are written as type T to the first mlive rows of Rout (n x n row-major, scaled by
// 1/sqrt(max diag)); the remaining rows are zero and mlive_out[b] = mlive.  Only R^T R matters
// downstream, so dropping and reordering rows is free.  One 256-thread block per batch entry.
constexpr int CH_NB = 16;
constexpr double NOISE_C = 8.0;

inline size_t chol_smem_bytes(int n) {
  return sizeof(double) * ((size_t)CH_NB * n + 64 * CH_NB + n) + sizeof(short) * 2 * (size_t)n + 64;
}

template <typename T>
__global__ __launch_bounds__(256) void chol_upper_kernel(double *__restrict__ Gg, long wG, int n,
                                                         T *__restrict__ Rg, long wR, int *__restrict__ mlive_out,
                                                         int only_flagged = 0, int ld = 0,
                                                         const int *__restrict__ ndyn = nullptr, int ndyn_mul = 1,
                                                         const int *__restrict__ run_flag = nullptr) {
  // only_flagged: chol_lowrank_kernel ran first and left mlive_out[b] = -1 where the rank exceeded its cap
  if (only_flagged && mlive_out[blockIdx.x] >= 0) return;
  // run_flag: batch entries with run_flag[b] >= 0 are not on this route.  ld: row stride of G and of the output
  // (default n); ndyn: per-entry order of the matrix (<= n), the rest of the ld x ld buffer is never touched.
  if (run_flag && run_flag[blockIdx.x] >= 0) return;
  const int ldg = ld ? ld : n;
  if (ndyn) n = max(0, min(n, ndyn[blockIdx.x] * ndyn_mul));
  extern __shared__ double ch_smem[];
  double *sP = ch_smem;                          // [CH_NB][n]   current block row of R
  double *sK = sP + CH_NB * n;                   // [64][CH_NB]  staged R[list[k]][jb..jb+nb)
  double *sN = sK + 64 * CH_NB;                  // [n]          row norms^2 of the finished factor
  short *sList = reinterpret_cast<short *>(sN + n);   // [n] live rows so far
  short *sPos = sList + n;                             // [n] output position of a row, -1 = dropped
  __shared__ double s_maxd, s_fro;
  __shared__ double s_red[4];
  __shared__ int s_nlive;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double *G = Gg + (long)blockIdx.x * wG;
  T *Rout = Rg + (long)blockIdx.x * wR;

  double md = 0.0;
  for (int i = tid; i < n; i += 256) md = fmax(md, G[(long)i * ldg + i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) md = fmax(md, __shfl_xor(md, o, 64));
  if (lane == 0) s_red[wave] = md;
  if (tid == 0) s_nlive = 0;
  __syncthreads();
  if (tid == 0) s_maxd = fmax(fmax(s_red[0], s_red[1]), fmax(s_red[2], s_red[3]));
  __syncthreads();
  const double maxd = s_maxd;
  const double eT = NOISE_C * eps_rt<T>();
  const double thresh = fmax((double)n * 2.220446049250313e-16, eT * eT) * maxd;

  for (int jb = 0; jb < n; jb += CH_NB) {
    const int nb = min(CH_NB, n - jb);
    for (int e = tid; e < nb * n; e += 256) {
      int c = e / n, r = e % n;
      sP[c * n + r] = (r >= jb) ? G[(long)(jb + c) * ldg + r] : 0.0;
    }
    __syncthreads();
    // left-looking update with the finished LIVE rows (held in G's upper triangle)
    const int nprev = s_nlive;
    for (int k0 = 0; k0 < nprev; k0 += 64) {
      const int kc = min(64, nprev - k0);
      const int kc8 = (kc + 7) & ~7;          // zero padded to the unroll width below
      for (int e = tid; e < kc8 * CH_NB; e += 256) {
        int k = e / CH_NB, c = e % CH_NB;
        sK[k * CH_NB + c] = (k < kc && c < nb) ? G[(long)sList[k0 + k] * ldg + jb + c] : 0.0;
      }
      __syncthreads();
      for (int r = jb + tid; r < n; r += 256) {
        double acc[CH_NB];
#pragma unroll
        for (int c = 0; c < CH_NB; ++c) acc[c] = 0.0;
        for (int k = 0; k < kc8; k += 8) {
          double rv[8];                          // eight independent loads in flight (L2 latency)
#pragma unroll
          for (int q = 0; q < 8; ++q) rv[q] = (k + q < kc) ? G[(long)sList[k0 + k + q] * ldg + r] : 0.0;
#pragma unroll
          for (int q = 0; q < 8; ++q)
#pragma unroll
            for (int c = 0; c < CH_NB; ++c) acc[c] += sK[(k + q) * CH_NB + c] * rv[q];
        }
#pragma unroll
        for (int c = 0; c < CH_NB; ++c)
          if (c < nb) sP[c * n + r] -= acc[c];
      }
      __syncthreads();
    }
    // factor the panel row by row.  Dead pivots (numerically zero directions; most of the panel
    // when the carry has low rank) cost no barrier: every wave finds the next live column itself
    // from the current diagonal and the columns in between are zeroed in one pass.
    int c = 0;
    while (c < nb) {
      const bool live_l = (lane < nb - c) && sP[(c + lane) * n + jb + c + lane] > thresh;
      const unsigned long long lm = __ballot(live_l);
      const int f = lm ? c + __ffsll((long long)lm) - 1 : nb;     // first live column at or after c
      for (int e = tid; e < (f - c) * (n - jb); e += 256) sP[(c + e / (n - jb)) * n + jb + e % (n - jb)] = 0.0;
      if (f >= nb) break;
      const double inv = jr_rsq64(sP[f * n + jb + f]);
      __syncthreads();                       // every wave has read the diagonal before row f is scaled
      for (int r = jb + f + tid; r < n; r += 256) sP[f * n + r] *= inv;
      if (tid == 0) { sList[s_nlive] = (short)(jb + f); s_nlive = s_nlive + 1; }
      __syncthreads();
      for (int e = tid; e < (nb - f - 1) * (n - jb); e += 256) {
        int c2 = f + 1 + e / (n - jb), r = jb + e % (n - jb);
        if (r >= jb + c2) sP[c2 * n + r] -= sP[f * n + jb + c2] * sP[f * n + r];
      }
      __syncthreads();
      c = f + 1;
    }
    __syncthreads();
    // publish the finished rows (needed by later panels) -- upper part only
    for (int e = tid; e < nb * n; e += 256) {
      int c = e / n, r = e % n;
      if (r >= jb + c) G[(long)(jb + c) * ldg + r] = sP[c * n + r];
    }
    __syncthreads();
  }
  // ---- rank compaction: rows with norm below NOISE_C*eps_T*|R|_F are dropped ----
  // (only the rows that passed the pivot test can be non-zero: sList[0..nfac))
  const int nfac = s_nlive;
  for (int q = wave; q < nfac; q += 4) {
    const int k = sList[q];
    double a = 0.0;
    for (int r = k + lane; r < n; r += 64) { double x = G[(long)k * ldg + r]; a += x * x; }
    a = wave_sum(a);
    if (lane == 0) sN[q] = a;
  }
  __syncthreads();
  {
    double f = 0.0;
    for (int q = tid; q < nfac; q += 256) f += sN[q];
    f = wave_sum(f);
    if (lane == 0) s_red[wave] = f;
    __syncthreads();
    if (tid == 0) s_fro = s_red[0] + s_red[1] + s_red[2] + s_red[3];
    __syncthreads();
  }
  const double nfloor = eT * eT * s_fro;
  if (wave == 0) {
    int cnt = 0;
    for (int base = 0; base < nfac; base += 64) {
      const int q = base + lane;
      const bool f = q < nfac && sN[q] > nfloor;
      const unsigned long long mask = __ballot(f);
      if (q < nfac) sPos[q] = f ? (short)(cnt + __popcll(mask & ((1ull << lane) - 1ull))) : (short)-1;
      cnt += __popcll(mask);
    }
    if (lane == 0) { s_nlive = cnt; if (mlive_out) mlive_out[blockIdx.x] = cnt; }
  }
  __syncthreads();
  const int mlive = s_nlive;
  const double sc = maxd > 0.0 ? 1.0 / sqrt(maxd) : 1.0;
  for (int q = wave; q < nfac; q += 4) {
    const int pos = sPos[q];
    if (pos < 0) continue;
    const int k = sList[q];
    // with a per-entry order (ndyn) the consumer reads whole rows of the ld-wide buffer: columns n..ld are written as zeros
    for (int r = lane; r < (ndyn ? ldg : n); r += 64) Rout[(long)pos * ldg + r] = (r >= k && r < n) ? T(G[(long)k * ldg + r] * sc) : T(0);
  }
  // rows beyond mlive are never read when the caller takes the live count (dynamic extents)
  if (mlive_out) return;
  for (int e = tid + mlive * n; e < n * n; e += 256) Rout[(long)(e / n) * ldg + (e % n)] = T(0);
}

// ---------------------------------------------------------------------------------------------
// The same factorisation (same pivot rule, same output contract, drop-in signature) restructured for orders of 100-256 at high
// rank (round 3: the forward factor and the first compression of the truncation route of a dense walker batch, where the
// kernel above spent 1.66 ms per launch of 2048 walkers on 5.6 Mflop per walker: a chain of ~50 barriers per 16-column panel):
//   * left-looking update of a panel = a small GEMM  P[16 x (n - jb)] -= R[live, panel]^T R[live, jb..n)  on
//     v_mfma_f64_16x16x4_f64: a wave owns up to four 16-column tiles, both operands are 128-byte segments of the finished factor
//     rows (read from G in L2), two k-steps in flight;
//   * the 16 x 16 diagonal block is factored by ONE wave in registers (lane c' holds column c', the pivot row travels by
//     v_readlane: no barrier inside the block), dropping the rows whose pivot is below the noise of the T-typed data;
//   * the rest of the panel row is a forward substitution per column, one thread per column, the factor of the block read as
//     LDS broadcasts.
// Three barriers per panel instead of three per column.
typedef double chb_f64x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double chb_readlane(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}
constexpr int CH_NS = 256;                      // LDS row stride of the panel (orders up to 256)
inline size_t chol_blocked_smem_bytes(int n) {
  return sizeof(double) * ((size_t)CH_NB * CH_NS + n) + sizeof(short) * 2 * (size_t)n + 64;
}

template <typename T, int MINB = 3, int PFD = 3>      // PFD: k-steps of the update in flight (loads from L2 ahead of the MFMAs)
__global__ __launch_bounds__(256, MINB) void chol_blocked_kernel(double *__restrict__ Gg, long wG, int n,
                                                           T *__restrict__ Rg, long wR, int *__restrict__ mlive_out,
                                                           int only_flagged = 0, int ld = 0,
                                                           const int *__restrict__ ndyn = nullptr, int ndyn_mul = 1,
                                                           const int *__restrict__ run_flag = nullptr, double thresh_scale = 1.0) {
  // thresh_scale: multiplies the pivot threshold (the dense f64 route redoes the few walkers whose factor kept more than 128 rows
  // with a higher one)
  if (only_flagged && mlive_out[blockIdx.x] >= 0) return;
  if (run_flag && run_flag[blockIdx.x] >= 0) return;
  const int ldg = ld ? ld : n;
  if (ndyn) n = max(0, min(n, ndyn[blockIdx.x] * ndyn_mul));
  extern __shared__ double chb_smem[];
  double *sP = chb_smem;                          // [CH_NB][CH_NS]   current block row
  double *sN = sP + CH_NB * CH_NS;                // [n]          row norms^2 of the finished factor
  short *sList = reinterpret_cast<short *>(sN + n);   // [n] live rows so far
  short *sPos = sList + n;                             // [n] output position of a row, -1 = dropped
  __shared__ double sD[CH_NB][CH_NB + 1], sDinv[CH_NB];
  __shared__ double s_maxd, s_fro;
  __shared__ double s_red[4];
  __shared__ int s_nlive;
  __shared__ unsigned s_livemask;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double *G = Gg + (long)blockIdx.x * wG;
  T *Rout = Rg + (long)blockIdx.x * wR;

  double md = 0.0;
  for (int i = tid; i < n; i += 256) md = fmax(md, G[(long)i * ldg + i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) md = fmax(md, __shfl_xor(md, o, 64));
  if (lane == 0) s_red[wave] = md;
  if (tid == 0) s_nlive = 0;
  __syncthreads();
  if (tid == 0) s_maxd = fmax(fmax(s_red[0], s_red[1]), fmax(s_red[2], s_red[3]));
  __syncthreads();
  const double maxd = s_maxd;
  const double eT = NOISE_C * eps_rt<T>();
  const double thresh = fmax((double)n * 2.220446049250313e-16, eT * eT) * maxd * thresh_scale;
  const double sc_out = maxd > 0.0 ? 1.0 / sqrt(maxd) : 1.0;
  const int i16 = lane & 15, k4 = lane >> 4;

  // Round 6 (phase counters of an instrumented build, scripts/chb_phase_patch.py: a walker alone took 475 us -- update 266, publish 68,
  // diagonal block 48, substitution 47, staging 17): thread = column everywhere (no e / n index arithmetic: ~3 us per panel), the
  // panel's rows of G travel to LDS by LDS-DMA while the update loop runs (no registers: PFD k-steps of the update in flight instead
  // of two), two rows of the substitution's factor reads in flight.  Measured and not kept: panels in PAIRS (one pass over the finished
  // rows for two panels, the second panel's sums waiting in registers: update 266 -> 160 us, but 32 more live registers through the
  // diagonal block / substitution / publish phases spill at three blocks per CU -- 560 us per walker at full occupancy against 440).
  constexpr int NS = CH_NS;                     // LDS row stride of the panel (the DMA writes whole 16-byte chunks: a fixed, even stride)
  const bool dma = (ldg & 1) == 0 && n >= 2;    // 16-byte aligned row starts (block-uniform); odd leading dimensions take the register path
  auto request_panel = [&](int jb, int nb) {
    // wave w brings rows w, w + 4, ...: lane l the doubles 2 l, 2 l + 1 (+ 128 for the second half) of the row, clamped inside the row
    for (int c = wave; c < nb; c += 4) {
      const double *row = G + (long)(jb + c) * ldg;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (128 * h < n) {
          const int e = min(128 * h + 2 * lane, ((n + 1) & ~1) - 2);      // (odd n: column n exists, the leading dimension is even)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(row + e),
                                           (__attribute__((address_space(3))) void *)(sP + c * NS + 128 * h), 16, 0, 0);
        }
      }
    }
  };
  auto load_panel = [&](int jb, int nb, double (&pv)[CH_NB]) {
#pragma unroll
    for (int c = 0; c < CH_NB; ++c) pv[c] = (c < nb && tid >= jb && tid < n) ? G[(long)(jb + c) * ldg + tid] : 0.0;
  };

  for (int jb = 0; jb < n; jb += CH_NB) {
    const int nb = min(CH_NB, n - jb);
    const int nprev = s_nlive;
    // the panel's rows of G: requested now, in LDS when the update loop is through
    double pv[CH_NB];
    if (dma) request_panel(jb, nb); else load_panel(jb, nb, pv);
    chb_f64x4 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[q][r] = 0.0;
    const int ntile = (n - jb + 15) >> 4;
    // ---- left-looking update on the matrix cores ----
    {
      int col[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int t = wave + 4 * q;
        col[q] = jb + 16 * t + i16;
        if (!(t < ntile && col[q] < n)) col[q] = jb;      // clamped address: the product lands in an entry that is not written back
      }
      const bool aok = jb + i16 < n;
      const int acol = aok ? jb + i16 : jb;
      const int nq = __builtin_amdgcn_readfirstlane((ntile - wave + 3) >> 2);    // tiles of this wave (wave-uniform)
      // The loop body is straight-line code for a compile-time tile count NQC (the wave's tiles of the block row; a switch outside the
      // loop picks it): every load and every MFMA unconditional -- rows beyond the live list are clamped copies whose A operand is
      // zeroed.  (Rounds 3-5 and the first form of this round had `if (q < nq)` / `if (cur + PF - 1 < nks)` inside: the compiler turned
      // them into exec-mask branches and waited for vmcnt(0) before the first MFMA of a k-step -- ONE k-step in flight whatever PF, which
      // is why 2 / 3 / 5 / 8 k-steps in flight all measured the same.)
      auto run = [&](auto NQC_) {
        constexpr int NQC = decltype(NQC_)::value;
        constexpr int PF = PFD;
        double av[PF], bv[PF][NQC];
        const int nks = (nprev + 3) >> 2;
        auto load = [&](int k0, double &a, double (&b)[NQC]) {
          const int kk = min(k0 + k4, nprev - 1);
          const unsigned row = (unsigned)(kk * ldg);   // finished rows sit compacted in the first rows of G (see the publish step);
          a = G[row + (unsigned)acol];                 // 32-bit offsets from the walker's (scalar) base: no 64-bit address per load
#pragma unroll
          for (int q = 0; q < NQC; ++q) b[q] = G[row + (unsigned)col[q]];
        };
#pragma unroll
        for (int p = 0; p < PF - 1; ++p) load(4 * p, av[p], bv[p]);
        for (int ks = 0; ks < nks; ks += PF) {
#pragma unroll
          for (int p = 0; p < PF; ++p) {
            const int cur = ks + p;
            load(4 * (cur + PF - 1), av[(p + PF - 1) % PF], bv[(p + PF - 1) % PF]);
            __builtin_amdgcn_sched_barrier(0);      // keep the rotation: without the fences the scheduler gathers the loads of PF steps
            const double am = (4 * cur + k4 < nprev && aok) ? av[p] : 0.0;      // in front of their MFMAs and waits for all of them
#pragma unroll
            for (int q = 0; q < NQC; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(am, bv[p][q], acc[q], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      };
      if (nprev > 0) {
        switch (nq) {
          case 4: run(std::integral_constant<int, 4>()); break;
          case 3: run(std::integral_constant<int, 3>()); break;
          case 2: run(std::integral_constant<int, 2>()); break;
          case 1: run(std::integral_constant<int, 1>()); break;
          default: break;
        }
      }
    }
    if (!dma && tid < n) {
#pragma unroll
      for (int c = 0; c < CH_NB; ++c) sP[c * NS + tid] = pv[c];
    }
    __syncthreads();                           // the panel is staged (the barrier's fence waits for the LDS-DMA: vmcnt(0))
    if (nprev > 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int t = wave + 4 * q;
        const int cc = jb + 16 * t + i16;
        if (t < ntile && cc < n) {
#pragma unroll
          for (int r = 0; r < 4; ++r) sP[(k4 + 4 * r) * NS + cc] -= acc[q][r];      // acc[r] = C[k4 + 4 r][i16]
        }
      }
    }
    __syncthreads();
    // ---- the 16 x 16 diagonal block, one wave, registers ----
    if (wave == 0) {
      double d[CH_NB];
#pragma unroll
      for (int c = 0; c < CH_NB; ++c) d[c] = (lane < nb && c <= lane) ? sP[c * NS + jb + lane] : 0.0;
      unsigned livemask = 0;
#pragma unroll
      for (int c = 0; c < CH_NB; ++c) {
        const double piv = chb_readlane(d[c], c);
        const bool live = c < nb && piv > thresh;             // wave-uniform
        // branch-free (round 4): a dropped pivot zeroes its row and the updates below become no-ops -- under `if (live)` the
        // compiler copied the whole d[] (16 v_mov_b64) at the join of every pivot step, 256 moves per panel on the one wave
        // every other wave of the block waits for
        const double pvt = live ? piv : 1.0;
        const double sc = jr_rsq64(pvt);
        d[c] = live ? d[c] * sc : 0.0;
#pragma unroll
        for (int c2 = c + 1; c2 < CH_NB; ++c2) {
          const double f = chb_readlane(d[c], c2);
          d[c2] -= f * d[c];
        }
        livemask |= live ? 1u << c : 0u;
      }
#pragma unroll
      for (int c = 0; c < CH_NB; ++c)
        if (lane < CH_NB) sD[c][lane] = lane >= c ? d[c] : 0.0;
      if (lane < CH_NB) {
        double dg = 0.0;
#pragma unroll
        for (int c = 0; c < CH_NB; ++c) dg = lane == c ? d[c] : dg;
        const double iv = jr_rcp64(dg);
        sDinv[lane] = ((livemask >> lane) & 1u) ? iv : 0.0;
      }
      if (lane == 0) s_livemask = livemask;
    }
    __syncthreads();
    const unsigned livemask = s_livemask;
    // ---- the rest of the block row: forward substitution per column; the block's own columns take the factor ----
    {
      const int r = jb + tid;
      if (r < n) {
        if (tid < CH_NB) {
#pragma unroll
          for (int c = 0; c < CH_NB; ++c) sP[c * NS + r] = sD[c][tid];
        } else {
          double x[CH_NB];
#pragma unroll
          for (int c = 0; c < CH_NB; ++c) {
            double v = sP[c * NS + r];
#pragma unroll
            for (int c1 = 0; c1 < c; ++c1) v -= sD[c1][c] * x[c1];
            x[c] = v * sDinv[c];                  // dropped row: sDinv = 0
            sP[c * NS + r] = x[c];
            if (c & 1) __asm__ volatile("" ::: "memory");    // keep the factor reads of the later rows from being hoisted (136 doubles): two rows' worth in flight
          }
        }
      }
    }
    __syncthreads();
    // publish the finished rows (upper part) for the later panels and -- scaled, as type T -- straight into the output at
    // their provisional position (= their rank in the live list: the noise-floor compaction at the end rarely drops one);
    // their squared norms while they are in LDS
    // (G row `pos` <= jb + c: a row of this or an earlier panel, already consumed -- the factor rows are COMPACTED into the
    // first rows of G, so the update loop addresses them without the live list; their columns below jb + c keep stale data
    // that no later panel reads)
    const int wout = ndyn ? ldg : n;
    if (tid < n) {
#pragma unroll
      for (int c = 0; c < CH_NB; ++c)
        if (c < nb && tid >= jb + c && ((livemask >> c) & 1u))
          G[(long)(nprev + __popc(livemask & ((1u << c) - 1u))) * ldg + tid] = sP[c * NS + tid];
    }
    for (int c = wave; c < nb; c += 4) {
      if (!((livemask >> c) & 1u)) continue;
      const int pos = nprev + __popc(livemask & ((1u << c) - 1u));
      double a = 0.0;
      for (int r = lane; r < wout; r += 64) {
        const double x = (r >= jb + c && r < n) ? sP[c * NS + r] : 0.0;
        a += x * x;
        Rout[(long)pos * ldg + r] = T(x * sc_out);
      }
      a = wave_sum(a);
      if (lane == 0) sN[pos] = a;
    }
    if (tid == 0) {
      int nl = s_nlive;
      for (int c = 0; c < nb; ++c)
        if ((livemask >> c) & 1u) sList[nl++] = (short)(jb + c);
      s_nlive = nl;
    }
    __threadfence_block();
    __syncthreads();
  }
  // ---- rank compaction: rows with norm below NOISE_C*eps_T*|R|_F are dropped (as chol_upper_kernel) ----
  const int nfac = s_nlive;
  {
    double f = 0.0;
    for (int q = tid; q < nfac; q += 256) f += sN[q];
    f = wave_sum(f);
    if (lane == 0) s_red[wave] = f;
    __syncthreads();
    if (tid == 0) s_fro = s_red[0] + s_red[1] + s_red[2] + s_red[3];
    __syncthreads();
  }
  const double nfloor = eT * eT * s_fro;
  if (wave == 0) {
    int cnt = 0;
    for (int base = 0; base < nfac; base += 64) {
      const int q = base + lane;
      const bool f = q < nfac && sN[q] > nfloor;
      const unsigned long long mask = __ballot(f);
      if (q < nfac) sPos[q] = f ? (short)(cnt + __popcll(mask & ((1ull << lane) - 1ull))) : (short)-1;
      cnt += __popcll(mask);
    }
    if (lane == 0) { s_nlive = cnt; if (mlive_out) mlive_out[blockIdx.x] = cnt; }
  }
  __syncthreads();
  const int mlive = s_nlive;
  if (mlive != nfac) {
    // a row fell below the floor (rare): the rows after it move up, in order (the target of a row is never below it)
    const int wrow = ndyn ? ldg : n;
    for (int q = 0; q < nfac; ++q) {
      const int pos = sPos[q];
      if (pos >= 0 && pos != q)
        for (int r = tid; r < wrow; r += 256) Rout[(long)pos * ldg + r] = Rout[(long)q * ldg + r];
      __syncthreads();
    }
  }
  if (mlive_out) return;
  for (int e = tid + mlive * n; e < n * n; e += 256) Rout[(long)(e / n) * ldg + (e % n)] = T(0);
}

// launch of the blocked factorisation: the MFMA form above for orders >= 48, chol_upper_kernel below that.
// (Measured and removed in round 6, numbers in HISTORY.md: the register-resident kernel of round 5 -- 181 us per block of order 256
// against 452, but one block per CU against three: 4.57 ms against 3.67 ms per launch of 8192 walkers --, four / two blocks per CU, and
// three / five / eight k-steps of the panel update in flight.)
template <typename T>
inline void launch_chol_upper(hipStream_t s, int nbatch, double *G, long wG, int n, T *R, long wR, int *mlive_out, int only_flagged = 0,
                              int ld = 0, const int *ndyn = nullptr, int ndyn_mul = 1, const int *run_flag = nullptr, double thresh_scale = 1.0) {
  PG_REQUIRE(thresh_scale == 1.0 || (n >= 48 && n <= 256), 1, "pivot threshold scaling needs the blocked Cholesky (orders 48 .. 256)");
  // (round 6: the blocked kernel's four waves own four 16-column tiles each = 256 columns; rounds 3-5 launched it for ANY order >= 48 and
  // a carry of D chi > 256 columns got a factor whose columns beyond 256 were never updated -- wrong amplitudes without a flag on full-rank
  // states with D chi > 256, found with scripts/bigbond_probe.py; larger orders take the general kernel)
  if (n >= 48 && n <= 256) {
    const size_t smem = chol_blocked_smem_bytes(n);
    allow_dynamic_lds(reinterpret_cast<const void *>(&chol_blocked_kernel<T, 3, 4>), smem);
    hipLaunchKernelGGL((chol_blocked_kernel<T, 3, 4>), dim3(nbatch), dim3(256), smem, s, G, wG, n, R, wR, mlive_out, only_flagged, ld, ndyn,
                       ndyn_mul, run_flag, thresh_scale);
  } else {
    const size_t smem = chol_smem_bytes(n);
    allow_dynamic_lds(reinterpret_cast<const void *>(&chol_upper_kernel<T>), smem);
    hipLaunchKernelGGL(chol_upper_kernel<T>, dim3(nbatch), dim3(256), smem, s, G, wG, n, R, wR, mlive_out, only_flagged, ld, ndyn, ndyn_mul,
                       run_flag);
  }
  PG_CHECK_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
// Low-rank variant of the same factorisation (same column order, same pivot rule, same output):
// right-looking, one step per LIVE row instead of one pass per 16-column panel, the factor held in
// LDS.  The cost of the blocked kernel above is ~n/16 dependent global-memory round trips whatever
// the rank; here it is one row read per live row.  When the rank exceeds CH_LR_CAP the kernel
// leaves mlive_out[b] = -1 (G is untouched) and chol_upper_kernel(only_flagged = 1) redoes that
// walker.  n <= 256 * CH_LR_Q.  Reads the upper triangle of G only.
constexpr int CH_LR_CAP = 32, CH_LR_Q = 4;
inline size_t chol_lowrank_smem_bytes(int n) { return sizeof(double) * (size_t)CH_LR_CAP * n; }

template <typename T>
__global__ __launch_bounds__(256) void chol_lowrank_kernel(const double *__restrict__ Gg, long wG, int n,
                                                           T *__restrict__ Rg, long wR, int *__restrict__ mlive_out,
                                                           int only_flagged = 0) {
  if (only_flagged && mlive_out[blockIdx.x] >= 0) return;   // gram_chol_lowrank_kernel finished this walker
  extern __shared__ double lr_R[];                 // [CH_LR_CAP][n]
  __shared__ double s_red[4], s_nrm[CH_LR_CAP];
  __shared__ int s_first[2][4];
  __shared__ short s_pos[CH_LR_CAP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const double *G = Gg + (long)blockIdx.x * wG;
  T *Rout = Rg + (long)blockIdx.x * wR;
  double d[CH_LR_Q];
  double md = 0.0;
#pragma unroll
  for (int q = 0; q < CH_LR_Q; ++q) {
    const int r = tid + 256 * q;
    d[q] = r < n ? G[(long)r * n + r] : 0.0;
    md = fmax(md, d[q]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) md = fmax(md, __shfl_xor(md, o, 64));
  if (lane == 0) s_red[wave] = md;
  __syncthreads();
  const double maxd = fmax(fmax(s_red[0], s_red[1]), fmax(s_red[2], s_red[3]));
  const double eT = NOISE_C * eps_rt<T>();
  const double thresh = fmax((double)n * 2.220446049250313e-16, eT * eT) * maxd;
  int nl = 0, f = -1;
  for (int step = 0;; ++step) {
    int cand = 0x7fffffff;
#pragma unroll
    for (int q = CH_LR_Q - 1; q >= 0; --q) {
      const int r = tid + 256 * q;
      if (r < n && r > f && d[q] > thresh) cand = r;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cand = min(cand, __shfl_xor(cand, o, 64));
    if (lane == 0) s_first[step & 1][wave] = cand;
    __syncthreads();
    f = min(min(s_first[step & 1][0], s_first[step & 1][1]), min(s_first[step & 1][2], s_first[step & 1][3]));
    if (f == 0x7fffffff) break;
    if (nl == CH_LR_CAP) {                 // rank above the cap: hand the walker to the blocked kernel
      if (tid == 0) mlive_out[blockIdx.x] = -1;
      return;
    }
    double piv = G[(long)f * n + f];
    for (int j = 0; j < nl; ++j) { const double x = lr_R[j * n + f]; piv -= x * x; }
    const double inv = jr_rsq64(piv);
#pragma unroll
    for (int q = 0; q < CH_LR_Q; ++q) {
      const int r = tid + 256 * q;
      if (r < n) {
        double v = 0.0;
        if (r >= f) {
          v = G[(long)f * n + r];
          for (int j = 0; j < nl; ++j) v -= lr_R[j * n + f] * lr_R[j * n + r];
          v *= inv;
        }
        lr_R[nl * n + r] = v;
        if (r > f) d[q] -= v * v;
      }
    }
    ++nl;
    __syncthreads();
  }
  // ---- rank compaction, as in chol_upper_kernel ----
  for (int j = wave; j < nl; j += 4) {
    double a = 0.0;
    for (int r = lane; r < n; r += 64) { const double x = lr_R[j * n + r]; a += x * x; }
    a = wave_sum(a);
    if (lane == 0) s_nrm[j] = a;
  }
  __syncthreads();
  double fro = 0.0;
  for (int j = 0; j < nl; ++j) fro += s_nrm[j];
  const double nfloor = eT * eT * fro;
  if (tid == 0) {
    int cnt = 0;
    for (int j = 0; j < nl; ++j) s_pos[j] = s_nrm[j] > nfloor ? (short)cnt++ : (short)-1;
    mlive_out[blockIdx.x] = cnt;
  }
  __syncthreads();
  const double sc = maxd > 0.0 ? 1.0 / sqrt(maxd) : 1.0;
  for (int j = 0; j < nl; ++j) {
    const int pos = s_pos[j];
    if (pos < 0) continue;
    for (int r = tid; r < n; r += 256) Rout[(long)pos * n + r] = T(lr_R[j * n + r] * sc);
  }
}

// ---------------------------------------------------------------------------------------------
// Gram-free variant for the low-rank case: R^T R = P^T P straight from the live rows of P (K x n,
// K = live carry rows, n <= 256 columns) without forming the n x n Gram matrix in memory.  Thread r
// OWNS column r: its column of P (K values) and its column of the factor (<= CH_LR_CAP values) stay
// in registers; a step needs one row of the Gram, G[f, r] = sum_k P[k, f] P[k, r], for which thread
// f broadcasts its two columns through LDS.  Same column order, pivot rule, compaction and output
// as chol_upper_kernel.  mlive_out[b] = -1 (nothing written) when K > KCAP or the rank exceeds
// CH_LR_CAP: the Gram GEMM and chol_upper_kernel then run for that walker only (batch_flag /
// only_flagged).  One 256-thread block per walker; f64 accumulation throughout.
// NT threads per walker: with a live inner extent the data columns are packed onto the first threads,
// so the 128-thread variant (twice as many walkers resident) takes every walker with <= 128 data
// columns and leaves mlive_out[b] = -2 for the 256-thread variant (retry_only = 1) otherwise.
template <typename T, int KCAP, int NT, int RCAP>
__device__ __forceinline__ void gram_chol_lowrank_body(const int walker, const T *__restrict__ Pg, long wP, int n,
                                                                const int *__restrict__ kdyn, int kdyn_mul, int kmax,
                                                                T *__restrict__ Rg, long wR, int *__restrict__ mlive_out,
                                                                int inner, const int *__restrict__ inner_live,
                                                                int retry_only, int max_pass, int small_first) {
  // More live rows than a thread holds (moderate rank): the rows of P are folded in over several passes INSIDE the
  // kernel -- pass 0 factors the first KCAP rows, every later pass factors [running factor (<= RCAP rows, still
  // in registers) ; next KCAP - RCAP rows of P]; up to max_pass passes, beyond that the walker is declined.
  constexpr int NWV = NT / 64;
  // retry_only: 0 every walker; 1 those a narrower launch left at -2 (more data columns than its threads); 2 those the
  // short launch (small_first: fewer rows and a smaller rank cap per thread, more walkers resident) handed on at -4
  if (retry_only == 1 && mlive_out[walker] != -2) return;
  if (retry_only == 2 && mlive_out[walker] != -4) return;
  // columns are (outer, inner) with `inner` fastest; inner_live[b] (optional) = live extent of the
  // inner index (live bond of the boundary MPS): columns beyond hold no data and are never read
  __shared__ __attribute__((aligned(16))) T s_pf[KCAP];   // column f of P
  __shared__ double s_rf[RCAP];     // column f of the factor
  __shared__ double s_piv;               // remaining diagonal of the pivot column = G[f,f] - sum_j R[j,f]^2
  __shared__ double s_red[2][NWV], s_nrm[RCAP], s_part[2 * NWV];
  __shared__ int s_first[2][NWV];
  __shared__ short s_pos[RCAP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Ktot = kdyn ? max(0, min(kmax, kdyn[walker] * kdyn_mul)) : kmax;
  if (Ktot > KCAP + (max_pass - 1) * (KCAP - RCAP)) {     // too many rows: decline
    if (tid == 0) mlive_out[walker] = small_first ? -4 : -1;
    return;
  }
  const T *P = Pg + (long)walker * wP;
  T *Rout = Rg + (long)walker * wR;
  // thread -> column: data columns (inner index below its live extent) packed in increasing order
  const int ilive = inner_live ? min(inner, inner_live[walker]) : inner;
  const int ncols = (n / inner) * ilive;
  if (ncols > NT) {
    if (tid == 0) mlive_out[walker] = -2;
    return;
  }
  const bool col_ok = tid < ncols;
  const int r = col_ok ? (tid / ilive) * inner + (tid % ilive) : n;
  const double eT = NOISE_C * eps_rt<T>();
  T pc[KCAP];
  double rc[RCAP];                   // own column of the factor (f64: pivots near the threshold amplify its rounding)
#pragma unroll
  for (int j = 0; j < RCAP; ++j) rc[j] = 0.0;
  int nl = 0, step = 0, k0 = 0, pass = 0;
  double maxd = 0.0;
#pragma unroll 1
  for (;; ++pass) {
    const int nfr = nl;                                                    // rows of the running factor
    const int npr = min(Ktot - k0, KCAP - (pass > 0 ? RCAP : 0));    // rows of P taken in this pass
    const int K = nfr + npr;
    double d = 0.0;
    // (rows requested unconditionally at clamped addresses, the predicates applied to the values: see gram_chol_wave_kernel)
    const int klast = max(K - 1, nfr);
    const unsigned rcl = col_ok ? (unsigned)r : 0u;
#pragma unroll
    for (int k = 0; k < KCAP; ++k) pc[k] = P[(unsigned)((k0 + min(max(k, nfr), klast) - nfr) * n) + rcl];
#pragma unroll
    for (int k = 0; k < KCAP; ++k) {
      T x = (k < K && col_ok) ? pc[k] : T(0);
      if (k < RCAP && k < nfr) x = T(rc[k < RCAP ? k : 0]);
      pc[k] = x;
      d += (double)pc[k] * (double)pc[k];
    }
#pragma unroll
    for (int j = 0; j < RCAP; ++j) rc[j] = 0.0;
    double md = d;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) md = fmax(md, __shfl_xor(md, o, 64));
    if (lane == 0) s_red[pass & 1][wave] = md;
    __syncthreads();
    maxd = s_red[pass & 1][0];
#pragma unroll
    for (int q = 1; q < NWV; ++q) maxd = fmax(maxd, s_red[pass & 1][q]);
    const double thresh = fmax((double)n * 2.220446049250313e-16, eT * eT) * maxd;
    int f = -1;
    nl = 0;
#pragma unroll 1
    for (;; ++step) {
      int cand = (r < n && r > f && d > thresh) ? r : 0x7fffffff;
      cand = wave_min_dpp(cand);
      if (lane == 0) s_first[step & 1][wave] = cand;
      __syncthreads();
      // squared norm of the row finished in the previous step (partials written before the barrier)
      if (tid == 0 && nl > 0) {
        double a = 0.0;
#pragma unroll
        for (int q = 0; q < NWV; ++q) a += s_part[NWV * ((step + 1) & 1) + q];
        s_nrm[nl - 1] = a;
      }
      f = s_first[step & 1][0];
#pragma unroll
      for (int q = 1; q < NWV; ++q) f = min(f, s_first[step & 1][q]);
      if (f == 0x7fffffff || nl >= K) { ++step; break; }   // the rank cannot exceed the K rows: later pivots are rounding noise
      if (nl == RCAP) {                 // rank above the cap: the blocked path (or the full-size launch) redoes this walker
        if (tid == 0) mlive_out[walker] = small_first ? -4 : -1;
        return;
      }
      if (r == f) {                          // the owner of the pivot column publishes it
#pragma unroll
        for (int k = 0; k < KCAP; ++k) s_pf[k] = pc[k];
#pragma unroll
        for (int j = 0; j < RCAP; ++j) s_rf[j] = rc[j];
        s_piv = d;                           // the owner's running diagonal IS the pivot: no thread recomputes it
      }
      __syncthreads();
      double g = 0.0;                        // g = G[f, r] - sum_j R[j,f] R[j,r]
      const double piv = s_piv;
#pragma unroll
      for (int kb = 0; kb < KCAP; kb += 16) {
        if (kb >= K) break;
        asm volatile("" ::: "memory");         // keep the LDS reads of later chunks from being hoisted (register pressure)
        T pfv[16];                             // one batch of LDS reads, then the arithmetic
#pragma unroll
        for (int k = 0; k < 16; ++k) pfv[k] = s_pf[kb + k];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          const double pf = (double)pfv[k];
          T pk = pc[kb + k];
          asm volatile("" : "+v"(pk));         // opaque copy: keeps the T -> f64 conversion of the (loop-invariant)
                                               // column inside the step loop instead of 2*KCAP live registers
          g = fma(pf, (double)pk, g);
        }
      }
#pragma unroll
      for (int jb = 0; jb < RCAP; jb += 8) {   // nl is block-uniform: whole chunks beyond it are skipped
        if (jb < nl) {
#pragma unroll
          for (int j = jb; j < jb + 8; ++j) {
            if (j < nl) {
              const double rf = s_rf[j];
              g = fma(-rf, rc[j], g);
            }
          }
        }
      }
      const double v = (r >= f && r < n) ? g * jr_rsq64(piv) : 0.0;
#pragma unroll
      for (int jb = 0; jb < RCAP; jb += 8) {
        if ((nl & ~7) == jb) {
#pragma unroll
          for (int j = jb; j < jb + 8; ++j) rc[j] = (j == nl) ? v : rc[j];
        }
      }
      const double v2 = v * v;
      if (r > f) d -= v2;
      const double a = (double)wave_sum_dpp((float)v2);   // row norm^2 for the compaction floor: f32 is ample
      if (lane == 0) s_part[NWV * (step & 1) + wave] = a;
      ++nl;
    }
    k0 += npr;
    if (k0 >= Ktot) break;
  }
  __syncthreads();
  double fro = 0.0;
  for (int j = 0; j < nl; ++j) fro += s_nrm[j];
  const double nfloor = eT * eT * fro;
  const double sc = maxd > 0.0 ? 1.0 / sqrt(maxd) : 1.0;
  if (tid == 0) {
    int cnt = 0;
    for (int j = 0; j < nl; ++j) s_pos[j] = s_nrm[j] > nfloor ? (short)cnt++ : (short)-1;
    mlive_out[walker] = cnt;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < RCAP; ++j) {
    if (j < nl) {
      const int pos = s_pos[j];
      if (pos >= 0 && r < n) Rout[(long)pos * n + r] = T(rc[j] * sc);
    }
  }
}

template <typename T, int KCAP, int NT, int MINW = 2, int RCAP = CH_LR_CAP>
__global__ __launch_bounds__(NT, MINW) void gram_chol_lowrank_kernel(const T *__restrict__ Pg, long wP, int n,
                                                                const int *__restrict__ kdyn, int kdyn_mul, int kmax,
                                                                T *__restrict__ Rg, long wR, int *__restrict__ mlive_out,
                                                                int inner = 1, const int *__restrict__ inner_live = nullptr,
                                                                int retry_only = 0, int max_pass = 1, int small_first = 0) {
  gram_chol_lowrank_body<T, KCAP, NT, RCAP>(blockIdx.x, Pg, wP, n, kdyn, kdyn_mul, kmax, Rg, wR, mlive_out, inner, inner_live, retry_only,
                                            max_pass, small_first);
}

// The same kernel over a device-built LIST of walkers (list[0 .. *count)): a small grid whose blocks walk the list.  Round 3: the
// walkers gram_chol_wave_kernel hands on (-4) are a few per launch on the headline workload; the full-grid fallback launch cost
// 110 us per site with every block returning at once (4 % of the step).
template <typename T, int KCAP, int NT, int MINW = 2, int RCAP = CH_LR_CAP>
__global__ __launch_bounds__(NT, MINW) void gram_chol_lowrank_list_kernel(const T *__restrict__ Pg, long wP, int n,
                                                                     const int *__restrict__ kdyn, int kdyn_mul, int kmax,
                                                                     T *__restrict__ Rg, long wR, int *__restrict__ mlive_out, int inner,
                                                                     const int *__restrict__ inner_live, int retry_only, int max_pass,
                                                                     const int *__restrict__ list, const int *__restrict__ count) {
  const int nl = *count;
  for (int q = blockIdx.x; q < nl; q += gridDim.x) {
    gram_chol_lowrank_body<T, KCAP, NT, RCAP>(list[q], Pg, wP, n, kdyn, kdyn_mul, kmax, Rg, wR, mlive_out, inner, inner_live, retry_only, max_pass, 0);
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// The Gram-free factor with ONE WAVE PER WALKER (f32 data; the first kernel every walker of the rank-adaptive absorption
// meets).  A lane owns two packed data columns (lane, lane + 64: at most 128 columns): its rows of P (64 per pass) and its
// entries of the factor (rank <= 16, f64) stay in registers; the pivot column is handed to the other lanes with
// v_readlane (a scalar operand of the FMAs) instead of an LDS broadcast, and nothing in the step loop crosses a workgroup
// barrier -- the 128-thread kernel above spends two barriers and an LDS round trip per pivot with three waves per SIMD to
// hide them.  Same arithmetic (f64 accumulation of exact f32 products, pivot rule, multi-pass folding of more than 64 rows,
// row compaction, scaling) and the same output contract; a walker it cannot take (more rows than the passes cover, more
// than 128 data columns, rank above 16) gets mlive_out[b] = -4 and goes to gram_chol_lowrank_kernel.
__device__ __forceinline__ double gw_readlane_f64(const double v, const int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}

__global__ __launch_bounds__(256, 2) void gram_chol_wave_kernel(const float *__restrict__ Pg, long wP, int n,
                                                              const int *__restrict__ kdyn, int kdyn_mul, int kmax,
                                                              float *__restrict__ Rg, long wR, int *__restrict__ mlive_out,
                                                              int inner, const int *__restrict__ inner_live, int max_pass,
                                                              int nbatch, int *__restrict__ decl_list = nullptr) {
  // decl_list (optional, nbatch + 1 ints, the count behind the list, zeroed by the caller): the walkers handed on at -4
  constexpr int KC = 64, RC = 16;
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= nbatch) return;
  const int Ktot = kdyn ? max(0, min(kmax, kdyn[b] * kdyn_mul)) : kmax;
  const int ilive = inner_live ? min(inner, inner_live[b]) : inner;
  const int ncols = (n / inner) * ilive;
  if (Ktot > KC + (max_pass - 1) * (KC - RC) || ncols > 128) {
    if (lane == 0) { mlive_out[b] = -4; if (decl_list) decl_list[atomicAdd(decl_list + nbatch, 1)] = b; }
    return;
  }
  const float *P = Pg + (long)b * wP;
  float *Rout = Rg + (long)b * wR;
  const int c0 = lane, c1 = lane + 64;                  // packed columns of this lane
  const bool ok0 = c0 < ncols, ok1 = c1 < ncols;
  const int r0 = ok0 ? (c0 / ilive) * inner + (c0 % ilive) : 0, r1 = ok1 ? (c1 / ilive) * inner + (c1 % ilive) : 0;
  const double eT = NOISE_C * (double)Eps<float>::v;
  float p0[KC], p1[KC];
  double q0[RC], q1[RC];                                // own entries of the factor rows
#pragma unroll
  for (int j = 0; j < RC; ++j) { q0[j] = 0.0; q1[j] = 0.0; }
  int nl = 0, k0 = 0;
  double maxd = 0.0;
  float nrm_mine = 0.f;                                 // lane j: squared norm of factor row j
#pragma unroll 1
  for (int pass = 0;; ++pass) {
    const int nfr = nl;                                 // rows of the running factor, folded in as the first rows
    const int npr = min(Ktot - k0, KC - (pass > 0 ? RC : 0));
    const int K = nfr + npr;
    double d0 = 0.0, d1 = 0.0;
    // Every row is requested unconditionally at a clamped address (a row that does not exist reads the last one that does, a
    // column that does not exist reads column 0) and the predicates are applied to the VALUES: under `if (ok0) x0 = P[...]` each
    // load sat in its own exec-masked block with an s_waitcnt vmcnt(0) at the join -- 18 full memory round trips per pass where
    // the 128 requests now go out back to back (round 6, found in the ISA: 304 exec branches in this loop).
    const int klast = max(K - 1, nfr);
    float y0[KC], y1[KC];
#pragma unroll
    for (int k = 0; k < KC; ++k) {
      const unsigned ro = (unsigned)((k0 + min(max(k, nfr), klast) - nfr) * n);
      y0[k] = P[ro + (unsigned)r0];
      y1[k] = P[ro + (unsigned)r1];
    }
#pragma unroll
    for (int k = 0; k < KC; ++k) {
      float x0 = (k < K && ok0) ? y0[k] : 0.f, x1 = (k < K && ok1) ? y1[k] : 0.f;
      if (k < RC && k < nfr) { x0 = (float)q0[k < RC ? k : 0]; x1 = (float)q1[k < RC ? k : 0]; }
      p0[k] = x0; p1[k] = x1;
      d0 += (double)x0 * (double)x0;
      d1 += (double)x1 * (double)x1;
    }
#pragma unroll
    for (int j = 0; j < RC; ++j) { q0[j] = 0.0; q1[j] = 0.0; }
    {
      double md = fmax(d0, d1);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) md = fmax(md, __shfl_xor(md, o, 64));
      maxd = md;
    }
    const double thresh = fmax((double)n * 2.220446049250313e-16, eT * eT) * maxd;
    int f = -1;
    nl = 0;
    nrm_mine = 0.f;
#pragma unroll 1
    for (;;) {
      int cand = (ok0 && c0 > f && d0 > thresh) ? c0 : ((ok1 && c1 > f && d1 > thresh) ? c1 : 0x7fffffff);
      cand = wave_min_dpp(cand);
      if (cand == 0x7fffffff || nl >= K) break;         // the rank cannot exceed the K rows
      if (nl == RC) {                                   // rank above the cap: the 128-thread kernels redo this walker
        if (lane == 0) { mlive_out[b] = -4; if (decl_list) decl_list[atomicAdd(decl_list + nbatch, 1)] = b; }
        return;
      }
      f = __builtin_amdgcn_readfirstlane(cand);
      const int lf = f & 63;
      double g0 = 0.0, g1 = 0.0, h0 = 0.0, h1 = 0.0, piv;   // (two partial sums per column: four independent FMA chains)
      auto step = [&](const float (&pp)[KC], const double (&qq)[RC], const double dd) {
        piv = gw_readlane_f64(dd, lf);
#pragma unroll
        for (int kb = 0; kb < KC; kb += 16) {
          if (kb >= K) break;
#pragma unroll
          for (int k = kb; k < kb + 16; ++k) {
            const double pf = (double)__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pp[k]), lf));
            // the f32 -> f64 conversions of the (loop-invariant) columns stay inside the step loop instead of 256 more live
            // registers: the conversion itself is the opaque statement (rounds 2-4 converted an opaque COPY of the column: two
            // v_mov per k, a quarter of the instructions of this loop)
            double x0, x1;
            asm volatile("v_cvt_f64_f32_e32 %0, %1" : "=v"(x0) : "v"(p0[k]));
            asm volatile("v_cvt_f64_f32_e32 %0, %1" : "=v"(x1) : "v"(p1[k]));
            if (k & 1) { h0 = fma(pf, x0, h0); h1 = fma(pf, x1, h1); }
            else { g0 = fma(pf, x0, g0); g1 = fma(pf, x1, g1); }
          }
        }
#pragma unroll
        for (int jb = 0; jb < RC; jb += 8) {
          if (jb < nl) {
#pragma unroll
            for (int j = jb; j < jb + 8; ++j) {
              if (j < nl) {
                const double rf = gw_readlane_f64(qq[j], lf);
                g0 = fma(-rf, q0[j], g0);
                g1 = fma(-rf, q1[j], g1);
              }
            }
          }
        }
      };
      if (f < 64) step(p0, q0, d0); else step(p1, q1, d1);
      g0 += h0; g1 += h1;
      const double inv = jr_rsq64(piv);
      const double v0 = (ok0 && c0 >= f) ? g0 * inv : 0.0, v1 = (ok1 && c1 >= f) ? g1 * inv : 0.0;
#pragma unroll
      for (int jb = 0; jb < RC; jb += 8) {
        if ((nl & ~7) == jb) {
#pragma unroll
          for (int j = jb; j < jb + 8; ++j) { q0[j] = (j == nl) ? v0 : q0[j]; q1[j] = (j == nl) ? v1 : q1[j]; }
        }
      }
      if (c0 > f) d0 -= v0 * v0;
      if (c1 > f) d1 -= v1 * v1;
      const float a = wave_sum_dpp((float)(v0 * v0 + v1 * v1));   // row norm^2 for the compaction floor: f32 is ample
      if (lane == nl) nrm_mine = a;
      ++nl;
    }
    k0 += npr;
    if (k0 >= Ktot) break;
  }
  const float fro = wave_sum_dpp(lane < nl ? nrm_mine : 0.f);
  const bool keep = lane < nl && (double)nrm_mine > eT * eT * (double)fro;
  const unsigned long long km = __ballot(keep);
  const int pos = keep ? __popcll(km & ((1ull << lane) - 1ull)) : -1;
  if (lane == 0) mlive_out[b] = __popcll(km);
  const double sc = maxd > 0.0 ? 1.0 / sqrt(maxd) : 1.0;
#pragma unroll
  for (int j = 0; j < RC; ++j) {
    if (j < nl) {
      const int pj = __builtin_amdgcn_readlane(pos, j);
      if (pj >= 0) {
        if (ok0) Rout[(long)pj * n + r0] = (float)(q0[j] * sc);
        if (ok1) Rout[(long)pj * n + r1] = (float)(q1[j] * sc);
      }
    }
  }
}

// (defined in trunc_mid.h) MFMA Gram of the live columns in registers + low-rank Cholesky, one kernel per walker
template <typename T>
inline void launch_colgram_chol(hipStream_t s, int nbatch, const T *P, long wP, int n, const int *kdyn, int kdyn_mul, int kmax,
                                T *R, long wR, int *mlive, int inner, const int *inner_live, int decline_code, bool hint_dense);

// Both variants in sequence: 128 threads per walker where the data columns fit, 256 otherwise.
template <typename T, int KCAP>
inline void launch_gram_chol_lowrank(hipStream_t s, int nbatch, const T *P, long wP, int n, const int *kdyn, int kdyn_mul,
                                     int kmax, T *R, long wR, int *mlive, int inner, const int *inner_live,
                                     int max_pass = 1, bool hint_dense = false, int *scratch_list = nullptr) {
  // scratch_list (optional, nbatch + 1 ints of device memory): the walkers the one-wave kernel hands on go through a list kernel
  constexpr bool no_narrow = false;
  const bool narrow = !no_narrow && (inner_live != nullptr || n <= 128);
  // (a Householder form of the factor -- qr_lowrank_kernel, rounds 2-5 behind a switch -- was slower at the same noise floor: removed, HISTORY.md)
  // Short columns and a small rank cap first (fewer registers: three waves per SIMD instead of two), then the walkers
  // it handed on (-4) with the full KCAP / CH_LR_CAP, then the ones with more data columns than 128 threads (-2).
  constexpr int KCAP_S = sizeof(T) == 4 ? 64 : 32;
  constexpr bool short_on = true;   // measured: cholesky category 272 -> 215 ms per two steps
  // (without per-walker row counts -- the second site of an absorption, whose 64 static rows fit one pass -- the one-wave kernel
  // takes the launch as well: that site ran on the 128-thread kernel, 1.49 ms against ~1.0 ms per launch of 49 152 walkers)
  constexpr bool no_wave_static = false;
  const bool wave_static = sizeof(T) == 4 && !no_wave_static && kdyn == nullptr && kmax <= 64 && true;
  const bool short_first = narrow && short_on && (kdyn != nullptr || wave_static);
  bool first_done = false;
  if constexpr (sizeof(T) == 4) {
    // dense states (hint): MFMA Gram in registers + low-rank Cholesky (trunc_mid.h) takes every walker with <= 128 live
    // columns and rank <= 96, whatever the number of rows; what it declines (-4) goes down the kernels below
    constexpr bool no_cg = false;
    if (!no_cg && hint_dense && inner > 0 && n % inner == 0) {
      launch_colgram_chol<T>(s, nbatch, P, wP, n, kdyn, kdyn_mul, kmax, R, wR, mlive, inner, inner_live, -4, hint_dense);
      first_done = true;
    }
  }
  if (short_first && !first_done) {
    constexpr bool no_wave = false;
    bool done = false;
    if constexpr (sizeof(T) == 4) {
      if (!no_wave) {   // one wave per walker, no LDS, no barrier (gram_chol_wave_kernel)
        constexpr bool no_list = false;
        int *dl = no_list ? nullptr : scratch_list;
        if (dl) PG_CHECK_HIP(hipMemsetAsync(dl + nbatch, 0, sizeof(int), s));
        hipLaunchKernelGGL(gram_chol_wave_kernel, dim3((nbatch + 3) / 4), dim3(256), 0, s, (const float *)P, wP, n, kdyn, kdyn_mul, kmax,
                           (float *)R, wR, mlive, inner, inner_live, std::max(max_pass, 2), nbatch, dl);
        done = true;
        if (dl && narrow && n <= 128) {     // the handed-on walkers: list kernel, then nothing else to launch
          hipLaunchKernelGGL((gram_chol_lowrank_list_kernel<T, KCAP, 128>), dim3(std::min(nbatch, 512)), dim3(128), 0, s, P, wP, n, kdyn,
                             kdyn_mul, kmax, R, wR, mlive, inner, inner_live, 2, max_pass, (const int *)dl, (const int *)(dl + nbatch));
          PG_CHECK_HIP(hipGetLastError());
          return;
        }
      }
    }
    if (!done)
      hipLaunchKernelGGL((gram_chol_lowrank_kernel<T, KCAP_S, 128, 3, 16>), dim3(nbatch), dim3(128), 0, s, P, wP, n, kdyn, kdyn_mul, kmax,
                         R, wR, mlive, inner, inner_live, 0, std::max(max_pass, 2), 1);
  }
  const bool handed = short_first || first_done;
  if (narrow)
    hipLaunchKernelGGL((gram_chol_lowrank_kernel<T, KCAP, 128>), dim3(nbatch), dim3(128), 0, s, P, wP, n, kdyn, kdyn_mul, kmax,
                       R, wR, mlive, inner, inner_live, handed ? 2 : 0, max_pass, 0);
  if (!narrow || n > 128)
    hipLaunchKernelGGL((gram_chol_lowrank_kernel<T, KCAP, 256>), dim3(nbatch), dim3(256), 0, s, P, wP, n, kdyn, kdyn_mul, kmax,
                       R, wR, mlive, inner, inner_live, narrow ? 1 : (first_done ? 2 : 0), max_pass, 0);
  PG_CHECK_HIP(hipGetLastError());
}

// Walkers that gram_chol_lowrank_kernel declined while the block still has fewer rows than columns
// (any R with R^T R = P^T P serves, P itself included): adopt the K live rows of P, normalised, as
// the carry and report K.  One block per walker; walkers with mlive >= 0 are left alone.
template <typename T>
__global__ __launch_bounds__(256) void adopt_rows_flagged_kernel(const T *__restrict__ Pg, long wP, int cols,
                                                                 const int *__restrict__ kdyn, int kdyn_mul, int kmax,
                                                                 T *__restrict__ Rg, long wR, int *__restrict__ mlive,
                                                                 int inner = 1, const int *__restrict__ inner_live = nullptr) {
  if (mlive[blockIdx.x] >= 0) return;
  const int ilive = inner_live ? inner_live[blockIdx.x] : inner;
  __shared__ double s_red[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = kdyn ? max(0, min(kmax, kdyn[blockIdx.x] * kdyn_mul)) : kmax;
  const T *P = Pg + (long)blockIdx.x * wP;
  T *R = Rg + (long)blockIdx.x * wR;
  const long cnt = (long)K * cols;
  double a = 0.0;
  for (long e = tid; e < cnt; e += 256) { const double x = (e % inner) < ilive ? (double)P[e] : 0.0; a += x * x; }
  a = wave_sum(a);
  if (lane == 0) s_red[wave] = a;
  __syncthreads();
  const double nrm2 = s_red[0] + s_red[1] + s_red[2] + s_red[3];
  const double sc = nrm2 > 0.0 ? 1.0 / sqrt(nrm2) : 1.0;
  for (long e = tid; e < cnt; e += 256) R[e] = (e % inner) < ilive ? T((double)P[e] * sc) : T(0);
  if (tid == 0) mlive[blockIdx.x] = K;
}

// Columns (outer, inner) of the K live rows of P whose inner index is beyond the walker's live
// extent were never written: zero them before a kernel that reads whole rows (Gram GEMM, normalise).
// flag (optional): only walkers with flag[b] < 0.
template <typename T>
__global__ __launch_bounds__(256) void zero_dead_cols_kernel(T *__restrict__ Pg, long wP, int cols,
                                                             const int *__restrict__ kdyn, int kdyn_mul, int kmax,
                                                             int inner, const int *__restrict__ inner_live,
                                                             const int *__restrict__ flag) {
  if (flag && flag[blockIdx.x] >= 0) return;
  const int ilive = inner_live[blockIdx.x];
  if (ilive >= inner) return;
  const int K = kdyn ? max(0, min(kmax, kdyn[blockIdx.x] * kdyn_mul)) : kmax;
  T *P = Pg + (long)blockIdx.x * wP;
  const long cnt = (long)K * cols;
  for (long e = threadIdx.x; e < cnt; e += 256)
    if ((e % inner) >= ilive) P[e] = T(0);
}

// ---------------------------------------------------------------------------------------------
// One-sided Jacobi on the rows of M (m x len, row stride ld), in place.  Round-robin tournament
// ordering, one wave per row pair.  Rows whose norm is below NOISE_C*eps*|M|_F (an invariant of
// the rotations) are numerically zero and take no part: with more rows than the rank (m > len,
// or a rank-deficient carry) the surplus rows can never become "relatively" orthogonal.  The
// working copy lives in LDS when it fits (use_lds), else in global memory (L2-resident).
// mdyn (optional): per-walker number of existing rows, mdyn[b]*mdyn_mul <= m (rank-adaptive carry).
// Rotation of a row pair (squared norms alpha, beta, inner product gamma), round 6: reciprocal and reciprocal square root from the
// hardware estimates (~2^-26) + two Newton steps each instead of four IEEE square roots and three divisions in float64 (~350
// dependent instructions on the critical path of every pair); the test |gamma| > tol sqrt(alpha beta) on the squares.
template <typename T>
__device__ __forceinline__ bool jr_rotation(const T alpha, const T beta, const T gamma, const T tol, const T floor2, T &cs, T &sn) {
  const double a = (double)alpha, b = (double)beta, g = (double)gamma;
  if (!(g * g > (double)tol * (double)tol * (a * b) && alpha > floor2 && beta > floor2)) return false;
  const double zeta = (b - a) * jr_rcp64(2.0 * g);
  const double az = fabs(zeta), w = fma(az, az, 1.0);
  const double td = copysign(jr_rcp64(az + w * jr_rsq64(w)), zeta);
  const double cd = jr_rsq64(fma(td, td, 1.0));
  cs = T(cd); sn = T(cd * td);
  return true;
}

// LDS-resident rows of 129 .. 256 elements (the Z = U M blocks of the dense f64 route: 40 % of its step), round 6: the sweeps on the
// LDS array itself -- ds_read instead of the FLAT loads behind the generic pointer `use_lds ? sM : Mglob`, which were waited for one at
// a time (the dot-product and rotation loops of a pair were 4 + 4 serial round trips) -- with both rows of a pair in registers between
// the inner products and the rotation (one read and one write per row and pair instead of two reads and a write).  The same sums in
// the same order: bit-identical results.  Two instantiations (rows of 129-192 and of 193-256 elements; C5's 144-element rows: Jacobi
// 115 -> 94 ms).  Measured and not kept: instantiations for rows of <= 128 elements too (C5 f64 6.99 k -> 6.36 k amp/s, the real state's
// 256-element rows 733 -> 775 ms: the kernel grows by two more copies of the sweep), ONE restructured body for every storage and
// length (169 against 154 ms on C5).  Shorter / longer rows and rows in global memory keep the loops below.
template <typename T, int NQ = 4>      // NQ: 64-element pieces of a row, len in (64 (NQ - 1), 64 NQ]
__device__ __forceinline__ int jacobi_rows_lds256(T *sM, const int m, const int len, const int lds_ld, const int max_sweeps, int &s_rot,
                                                  double *s_fro, short *s_idx, unsigned char *s_flag, int &s_nl) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  const T tol = T(2) * sqrt(T(len)) * T(eps_rt<T>());
  T *M = sM;
  constexpr bool REG = true;
  {
    {
      double f = 0.0;
      for (int e = tid; e < m * len; e += blockDim.x) { double x = (double)M[(long)(e / len) * lds_ld + (e % len)]; f += x * x; }
      f = wave_sum(f);
      if (lane == 0) s_fro[wave] = f;
      __syncthreads();
      if (tid == 0) { double t = 0.0; for (int w = 0; w < nw; ++w) t += s_fro[w]; s_fro[0] = t; }
      __syncthreads();
    }
    const T floor2 = T(NOISE_C * NOISE_C * eps_rt<T>() * eps_rt<T>() * s_fro[0]);
    int sweep = 0;
    for (; sweep < max_sweeps; ++sweep) {
      if (tid == 0) s_rot = 0;
      for (int r = wave; r < m; r += nw) {
        const auto *pr = M + (long)r * lds_ld;
        T n2 = 0;
        for (int c = lane; c < len; c += 64) { T x = pr[c]; n2 += x * x; }
        n2 = wave_sum(n2);
        if (lane == 0) s_flag[r] = n2 > floor2;
      }
      __syncthreads();
      if (wave == 0) {   // deterministic compaction of the live row list
        int cnt = 0;
        for (int base = 0; base < m; base += 64) {
          const int r = base + lane;
          const bool f = r < m && s_flag[r];
          const unsigned long long mask = __ballot(f);
          if (f) s_idx[cnt + __popcll(mask & ((1ull << lane) - 1ull))] = (short)r;
          cnt += __popcll(mask);
        }
        if (lane == 0) s_nl = cnt;
      }
      __syncthreads();
      const int nl = s_nl;
      const int lp = nl + (nl & 1);
      // (round 5, measured and removed: a wave taking four pairs of a round at a time -- eight rows requested together, rotated from
      // registers -- moved the f64 mode on the dense real state 25.2 -> 28.6 amp/s and C5 f64 2 268 -> 1 897: with the rows in global
      // memory the kernel is bound by the TRAFFIC of a sweep, m - 1 passes over the whole matrix (261 MB per sweep of a 256 x 256
      // float64 block, 512 blocks in flight = 256 MB of working set), not by the latency of a pair.  The dense f64 sites take the
      // two-level preconditioned route of engine_impl.h instead, whose Jacobi problems fit LDS.)
      for (int r = 0; r < lp - 1; ++r) {
        auto pair_of = [&](const int p, int &a, int &b) -> bool {      // rows of pair p of round r; false: a bye
          if (p >= lp / 2) return false;
          if (p == 0) { a = lp - 1; b = r; }
          else { a = r + p; a -= a >= lp - 1 ? lp - 1 : 0; b = r - p; b += b < 0 ? lp - 1 : 0; }      // (0 <= r < lp - 1, 0 < p < lp / 2: no division)
          if (a > b) { int t = a; a = b; b = t; }
          if (b >= nl) return false;
          a = s_idx[a]; b = s_idx[b];
          return true;
        };
        // (measured and not kept, round 6: two pairs of the round per pass of a wave with their reads, reductions and rotation
        // parameters interleaved -- f64 real state: Jacobi 1 057 -> 1 107 ms per step, C5 f64 169 -> 206)
        for (int p = wave; p < lp / 2; p += nw) {
          int a, b;
          if (!pair_of(p, a, b)) continue;
          auto *pa = M + (long)a * lds_ld, *pb = M + (long)b * lds_ld;
          T alpha = 0, beta = 0, gamma = 0;
          T xr[REG ? NQ : 1], yr[REG ? NQ : 1];
          if constexpr (REG) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {      // unconditional reads at clamped positions, the predicate on the value
              const int c = lane + 64 * q, cc = min(c, len - 1);
              const T x = pa[cc], y = pb[cc];
              xr[q] = c < len ? x : T(0); yr[q] = c < len ? y : T(0);
            }
#pragma unroll
            for (int q = 0; q < NQ; ++q) { alpha += xr[q] * xr[q]; beta += yr[q] * yr[q]; gamma += xr[q] * yr[q]; }
          } else {
            for (int c = lane; c < len; c += 64) {
              T x = pa[c], y = pb[c];
              alpha += x * x; beta += y * y; gamma += x * y;
            }
          }
          alpha = wave_sum(alpha); beta = wave_sum(beta); gamma = wave_sum(gamma);
          // no de Rijk row swapping: exchanging rows inside a round-robin tournament breaks the
          // pair coverage of the sweep (measured: 2x the sweeps); select_rows_kernel sorts afterwards
          T cs, sn;      // rotation parameters in f64 (one scalar per pair): keeps c^2 + s^2 = 1 to f32 rounding
          if (jr_rotation(alpha, beta, gamma, tol, floor2, cs, sn)) {
            if constexpr (REG) {
#pragma unroll
              for (int q = 0; q < NQ; ++q) {
                const int c = lane + 64 * q;
                if (c < len) { pa[c] = cs * xr[q] - sn * yr[q]; pb[c] = sn * xr[q] + cs * yr[q]; }
              }
            } else {
              for (int c = lane; c < len; c += 64) {
                T x = pa[c], y = pb[c];
                T xn = cs * x - sn * y, yn = sn * x + cs * y;
                pa[c] = xn;
                pb[c] = yn;
              }
            }
            if (lane == 0) atomicAdd(&s_rot, 1);
          }
        }
        __threadfence_block();
        __syncthreads();
      }
      const int rot = s_rot;
      __syncthreads();
      if (rot == 0) { ++sweep; break; }
    }
    return sweep;
  }
}

template <typename T>
__global__ __launch_bounds__(1024) void jacobi_rows_kernel(T *__restrict__ Mg, long wM, int m, int len,
                                                           int ld, int max_sweeps, int use_lds,
                                                           int *__restrict__ sweeps_out,
                                                           const int *__restrict__ mdyn, int mdyn_mul,
                                                           int skip_small = 0, int skip_le = 0, int lds_cap_bytes = 0) {
  extern __shared__ unsigned char jc_smem_raw[];
  T *sM = reinterpret_cast<T *>(jc_smem_raw);
  if (mdyn) m = max(0, min(m, mdyn[blockIdx.x] * mdyn_mul));
  if (skip_small && m <= max(skip_small, 32)) return;   // the one-wave kernels (jacobi_reg.h) / the mid route took this walker
  if (skip_le && m <= skip_le) return;                  // (f64: the short-row kernel took it)
  // use_lds == 2 (round 5): the static block does not fit LDS, but the launch carries lds_cap_bytes of dynamic LDS and a walker
  // whose LIVE rows fit takes them -- a sweep over rows in global memory is m - 1 passes over the matrix through L2 / HBM
  // (C5 f64: 88 % of the step on blocks of a few tens of live rows of 144 doubles)
  if (use_lds == 2) use_lds = ((size_t)m * (size_t)(len | 1) * sizeof(T) <= (size_t)lds_cap_bytes) ? 1 : 0;
  __shared__ int s_rot;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  T *Mglob = Mg + (long)blockIdx.x * wM;
  T *M = Mglob;
  int lds_ld = ld;
  if (use_lds) {
    lds_ld = len | 1;  // odd pitch: rows of a pair start on different banks
    for (int e = tid; e < m * len; e += blockDim.x) sM[(e / len) * lds_ld + (e % len)] = Mglob[(long)(e / len) * ld + (e % len)];
    M = sM;
    __syncthreads();
  }
  const int mp = m + (m & 1);
  const T tol = T(2) * sqrt(T(len)) * T(eps_rt<T>());
  __shared__ double s_fro[16];
  __shared__ short s_idx[1024];
  __shared__ unsigned char s_flag[1024];
  __shared__ int s_nl;
  if (use_lds && len > 128 && len <= 256) {      // block-uniform
    const int sw = len > 192 ? jacobi_rows_lds256<T, 4>(sM, m, len, lds_ld, max_sweeps, s_rot, s_fro, s_idx, s_flag, s_nl)
                             : jacobi_rows_lds256<T, 3>(sM, m, len, lds_ld, max_sweeps, s_rot, s_fro, s_idx, s_flag, s_nl);
    for (int e = tid; e < m * len; e += blockDim.x) Mglob[(long)(e / len) * ld + (e % len)] = sM[(e / len) * lds_ld + (e % len)];
    if (tid == 0 && sweeps_out) sweeps_out[blockIdx.x] = sw;
    return;
  }
  {
    double f = 0.0;
    for (int e = tid; e < m * len; e += blockDim.x) { double x = (double)M[(long)(e / len) * lds_ld + (e % len)]; f += x * x; }
    f = wave_sum(f);
    if (lane == 0) s_fro[wave] = f;
    __syncthreads();
    if (tid == 0) { double t = 0.0; for (int w = 0; w < nw; ++w) t += s_fro[w]; s_fro[0] = t; }
    __syncthreads();
  }
  const T floor2 = T(NOISE_C * NOISE_C * eps_rt<T>() * eps_rt<T>() * s_fro[0]);
  // The tournament runs over the LIVE rows only (norm above the noise floor), re-listed at the
  // start of every sweep: a row below the floor takes part in no rotation anyway, and with a
  // fast-decaying boundary spectrum most rows of M are dead (cost ~ rank^2, not m^2).
  (void)mp;
  int sweep = 0;
  for (; sweep < max_sweeps; ++sweep) {
    if (tid == 0) s_rot = 0;
    for (int r = wave; r < m; r += nw) {
      const T *pr = M + (long)r * lds_ld;
      T n2 = 0;
      for (int c = lane; c < len; c += 64) { T x = pr[c]; n2 += x * x; }
      n2 = wave_sum(n2);
      if (lane == 0) s_flag[r] = n2 > floor2;
    }
    __syncthreads();
    if (wave == 0) {   // deterministic compaction of the live row list
      int cnt = 0;
      for (int base = 0; base < m; base += 64) {
        const int r = base + lane;
        const bool f = r < m && s_flag[r];
        const unsigned long long mask = __ballot(f);
        if (f) s_idx[cnt + __popcll(mask & ((1ull << lane) - 1ull))] = (short)r;
        cnt += __popcll(mask);
      }
      if (lane == 0) s_nl = cnt;
    }
    __syncthreads();
    const int nl = s_nl;
    const int lp = nl + (nl & 1);
    // (round 5, measured and removed: a wave taking four pairs of a round at a time -- eight rows requested together, rotated from
    // registers -- moved the f64 mode on the dense real state 25.2 -> 28.6 amp/s and C5 f64 2 268 -> 1 897: with the rows in global
    // memory the kernel is bound by the TRAFFIC of a sweep, m - 1 passes over the whole matrix (261 MB per sweep of a 256 x 256
    // float64 block, 512 blocks in flight = 256 MB of working set), not by the latency of a pair.  The dense f64 sites take the
    // two-level preconditioned route of engine_impl.h instead, whose Jacobi problems fit LDS.)
    for (int r = 0; r < lp - 1; ++r) {
      for (int p = wave; p < lp / 2; p += nw) {
        int a, b;
        if (p == 0) { a = lp - 1; b = r; }
        else { a = r + p; a -= a >= lp - 1 ? lp - 1 : 0; b = r - p; b += b < 0 ? lp - 1 : 0; }      // (0 <= r < lp - 1, 0 < p < lp / 2: no division)
        if (a > b) { int t = a; a = b; b = t; }
        if (b >= nl) continue;
        a = s_idx[a]; b = s_idx[b];
        T *pa = M + (long)a * lds_ld, *pb = M + (long)b * lds_ld;
        T alpha = 0, beta = 0, gamma = 0;
        for (int c = lane; c < len; c += 64) {
          T x = pa[c], y = pb[c];
          alpha += x * x; beta += y * y; gamma += x * y;
        }
        alpha = wave_sum(alpha); beta = wave_sum(beta); gamma = wave_sum(gamma);
        // no de Rijk row swapping: exchanging rows inside a round-robin tournament breaks the
        // pair coverage of the sweep (measured: 2x the sweeps); select_rows_kernel sorts afterwards
        T cs, sn;      // rotation parameters in f64 (one scalar per pair): keeps c^2 + s^2 = 1 to f32 rounding
        if (jr_rotation(alpha, beta, gamma, tol, floor2, cs, sn)) {
          for (int c = lane; c < len; c += 64) {
            T x = pa[c], y = pb[c];
            T xn = cs * x - sn * y, yn = sn * x + cs * y;
            pa[c] = xn;
            pb[c] = yn;
          }
          if (lane == 0) atomicAdd(&s_rot, 1);
        }
      }
      __threadfence_block();
      __syncthreads();
    }
    const int rot = s_rot;
    __syncthreads();
    if (rot == 0) { ++sweep; break; }
  }
  if (use_lds) {
    for (int e = tid; e < m * len; e += blockDim.x) Mglob[(long)(e / len) * ld + (e % len)] = sM[(e / len) * lds_ld + (e % len)];
  }
  if (tid == 0 && sweeps_out) sweeps_out[blockIdx.x] = sweep;
}

// Optional epilogue of the four-walkers-per-wave kernel: what select_rows_kernel does with the rotated rows (rank by norm,
// truncation rule, normalise into Vt, live count), straight from the registers -- the rows are then NOT written back to M,
// and select_rows_kernel skips the walkers taken here (1 <= live rows <= JR_BR).
struct JrSelect {
  float *V = nullptr;      // [walker][k][len], nullptr: write the rows back to M as before
  long wV = 0;
  int k = 0;
  int *klive_out = nullptr;
  double trunc_err = 0.0;
  int dmin = 0;
};


// ---------------------------------------------------------------------------------------------
// Rows of M are mutually orthogonal: sigma_i = |row_i|.  Keep the k largest (ties by index),
// write Vt[rank][:] = row/sigma (zero row if sigma == 0) and S[rank] = sigma.
// Block = 256 threads, one batch entry per block; m <= 1024.
template <typename T>
__global__ __launch_bounds__(256) void select_rows_kernel(const T *__restrict__ Mg, long wM, int m, int len,
                                                          int ld, int k, T *__restrict__ Vg, long wV,
                                                          T *__restrict__ Sg, long wS,
                                                          const int *__restrict__ mdyn, int mdyn_mul,
                                                          int *__restrict__ klive_out = nullptr,
                                                          double trunc_err = 0.0, int dmin = 0,
                                                          double *__restrict__ err_out = nullptr,
                                                          const int *__restrict__ run_flag = nullptr, int run_if_neg = 1,
                                                          int skip_rows_le = 0) {
  // run_flag: with run_if_neg = 1 only the entries with run_flag[b] < 0 run, with 0 only those with run_flag[b] >= 0
  if (run_flag && ((run_flag[blockIdx.x] < 0) != (run_if_neg != 0))) return;
  // skip_rows_le: entries with 1 .. skip_rows_le live rows were selected by the Jacobi kernel itself (JrSelect, jacobi_reg.h)
  if (skip_rows_le > 0) {
    const int ml = mdyn ? max(0, min(m, mdyn[blockIdx.x] * mdyn_mul)) : m;
    if (ml >= 1 && ml <= skip_rows_le) return;
  }
  __shared__ double s_norm[1024];
  __shared__ int s_rank[1024];
  __shared__ int s_klive, s_kcut;
  if (threadIdx.x == 0) { s_klive = 0; s_kcut = k; }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const T *M = Mg + (long)blockIdx.x * wM;
  T *V = Vg + (long)blockIdx.x * wV;
  if (mdyn) m = max(0, min(m, mdyn[blockIdx.x] * mdyn_mul));
  // fewer existing rows than kept bonds: the surplus rows of Vt are zero
  for (int e = tid + min(m, k) * len; e < k * len; e += 256) V[e] = T(0);
  if (Sg) for (int r = min(m, k) + tid; r < k; r += 256) Sg[(long)blockIdx.x * wS + r] = T(0);
  for (int r = wave; r < m; r += 4) {
    double a = 0.0;
    for (int c = lane; c < len; c += 64) a += abs2_of(M[(long)r * ld + c]);
    a = wave_sum(a);
    if (lane == 0) s_norm[r] = sqrt(a);
  }
  __syncthreads();
  for (int r = tid; r < m; r += 256) {
    double v = s_norm[r];
    int rk = 0;
    for (int q = 0; q < m; ++q) {
      double u = s_norm[q];
      rk += (u > v) || (u == v && q < r);
    }
    s_rank[r] = rk;
    if (rk < k && Sg) Sg[(long)blockIdx.x * wS + rk] = T(v);
  }
  __syncthreads();
  double fro2 = 0.0;
  for (int q = 0; q < m; ++q) fro2 += s_norm[q] * s_norm[q];
  // Liveness floor = TWICE the floor below which the Jacobi kernels freeze a row (NOISE_C eps |M|_F): a frozen row was
  // never orthogonalised against the others, and a row whose norm sits at the common threshold could be frozen there and
  // still counted as live here (the two kernels sum the norm differently).  Normalised, such a row -- mostly rounding
  // residue of the dominant direction -- enters Vt with an O(1) overlap with it and corrupts the projector (measured:
  // one walker in 8192 off by 1.4e-3 on one contraction route).  With the factor of two every live row was rotated.
  const double nfloor = 2.0 * NOISE_C * eps_rt<T>() * sqrt(fro2);
  if (trunc_err > 0.0 || err_out) {
    // qlten::SVD(trunc_err, Dmin, Dmax) as bmps_impl.h:235-238 calls it: singular values go from the
    // smallest while more than Dmax are kept, or more than Dmin and the discarded weight / total weight
    // stays below trunc_err (oracle/tensor.py truncation_rank; the rule itself lives in TensorToolkit).
    __shared__ double s_sorted[1024];
    for (int r = tid; r < m; r += 256) s_sorted[s_rank[r]] = s_norm[r];
    __syncthreads();
    if (tid == 0) {
      int kept = m;
      double err = 0.0;
      while (kept > 0) {
        if (kept <= dmin && kept <= k) break;
        const double w = fro2 > 0.0 ? s_sorted[kept - 1] * s_sorted[kept - 1] / fro2 : 0.0;
        if (kept > k || (kept > dmin && err + w < trunc_err)) { err += w; --kept; }
        else break;
      }
      s_kcut = max(kept, 1);
      if (err_out) err_out[blockIdx.x] = err;
    }
    __syncthreads();
    // rows of Vt between the kept count and k are zero
    const int kc = s_kcut;
    for (int e = tid + kc * len; e < min(m, k) * len; e += 256) V[e] = T(0);
    if (Sg) for (int r = kc + tid; r < min(m, k); r += 256) Sg[(long)blockIdx.x * wS + r] = T(0);
  }
  const int kcut = s_kcut;
  for (int r = wave; r < m; r += 4) {
    int rk = s_rank[r];
    if (rk >= kcut) continue;
    double nv = s_norm[r];
    const double inv = nv > nfloor ? 1.0 / nv : 0.0;   // numerically zero direction -> zero row of Vt
    for (int c = lane; c < len; c += 64) V[(long)rk * len + c] = scaled(M[(long)r * ld + c], inv);
    if (lane == 0 && nv > nfloor) atomicAdd(&s_klive, 1);
  }
  // rows are ranked by norm, so the non-zero rows of Vt are its first klive rows
  if (klive_out) {
    __syncthreads();
    if (tid == 0) klive_out[blockIdx.x] = s_klive;
  }
}

// Mid-rank route of the chi-truncation (Engine::absorb_impl): which walkers take it, and the order of their Gram matrix.
//   ml = live rows of M (mdyn[b] * mul, or the static m);  lo < ml <= hi  ->  flag[b] = -1, nmid[b] = ml;  else 0, 0
__global__ void mid_route_flag_kernel(const int *__restrict__ mdyn, int mdyn_mul, int m, int lo, int hi, int nbatch,
                                      int *__restrict__ flag, int *__restrict__ nmid) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nbatch) return;
  const int ml = mdyn ? max(0, min(m, mdyn[b] * mdyn_mul)) : m;
  const bool mid = ml > lo && ml <= hi;
  flag[b] = mid ? -1 : 0;
  nmid[b] = mid ? ml : 0;
}

// Bookkeeping of the dense float64 truncation route (Engine::absorb_impl, round 5): flag[b] = -1 on the route, 0 off it
__global__ void f64_route_init_kernel(const int *__restrict__ mdyn, int mdyn_mul, int m, int nbatch, int *__restrict__ rows,
                                      int *__restrict__ flag) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nbatch) return;
  rows[b] = mdyn ? max(0, min(m, mdyn[b] * mdyn_mul)) : m;
  flag[b] = -1;
}
// walkers whose first factor kept more than hi rows get a second factorisation with a higher pivot threshold: redo[b] = -1 (run), lvl[b] = 1
// (level 1 and level 2: the second and the third factorisation; lvl[b] keeps the highest level a walker was redone at)
__global__ void f64_route_redo_kernel(const int *__restrict__ rows, int hi, int nbatch, int *__restrict__ redo, int *__restrict__ lvl,
                                      int level = 1) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nbatch) return;
  const bool again = rows[b] > hi;
  redo[b] = again ? -1 : 0;
  if (level == 1) lvl[b] = again ? 1 : 0;
  else if (again) lvl[b] = level;
}
// a walker whose factor kept fewer than lo or more than hi rows leaves the route; the row count of every walker off the route reads 0
__global__ void f64_route_check_kernel(int *__restrict__ flag, int *__restrict__ rows, int lo, int hi, int nbatch) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nbatch) return;
  if (flag[b] < 0 && (rows[b] < lo || rows[b] > hi)) flag[b] = 0;
  if (flag[b] >= 0) rows[b] = 0;
}
// Guard of the route: from the norms of the rotated rows of Z (the singular values inside the oversampled subspace, kz live rows)
// the mixing the two Gram compressions can have left between the kept direction k and what lies beyond the subspace is bounded by
//     eps_chol (s_1 / s_k)^2 (s_last / s_1),   s_last = the weakest direction of the subspace (>= the strongest one outside),
// eps_chol ~ 3e-15 (sqrt(n) eps of the float64 Cholesky of an order-256 Gram).  A walker whose bound exceeds `tol` -- a spectrum
// that falls to the resolution of a Gram (2.4e-7 s_1) within the subspace -- leaves the route; the general Jacobi redoes it.
// kq: the subspace dimension aimed at; a walker that kept fewer directions (its factors dropped the rest below the resolution of a
// Gram, 2.4e-7 s_1) is priced with THAT as the strongest direction outside.
template <typename TZ>
__global__ __launch_bounds__(256) void f64_route_guard_kernel(const TZ *__restrict__ Zg, long wZ, int len, const int *__restrict__ kz,
                                                              int k, double tol, int *__restrict__ flag, int kq,
                                                              const int *__restrict__ lvl = nullptr, double lvl_floor2 = 0.0,
                                                              double lvl2_floor2 = 0.0, double base_floor2 = 5.7e-14, int base_always = 0) {
  // base_floor2: (relative, squared) what the FIRST factorisation drops -- (2.4e-7)^2 at the default pivot threshold; base_always: it
  // was raised for every walker (tscale of the route), so it bounds the strongest direction outside whatever was kept
  // lvl[b] != 0: the walker's first factor was taken with the raised pivot threshold -- what it dropped is up to sqrt(lvl_floor2) s_1
  const int b = blockIdx.x;
  if (flag[b] >= 0) return;
  __shared__ double s_n[64];
  const int rows = min(kz[b], 64), lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const TZ *Z = Zg + (long)b * wZ;
  for (int r = wave; r < rows; r += 4) {
    double a = 0.0;
    for (int c = lane; c < len; c += 64) a += abs2_of(Z[(long)r * len + c]);
    a = wave_sum(a);
    if (lane == 0) s_n[r] = a;           // squared norms
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double s1 = 0.0, slast = 1e300;
    for (int r = 0; r < rows; ++r) { s1 = fmax(s1, s_n[r]); slast = fmin(slast, s_n[r]); }
    // k-th largest squared norm (rows <= 64: selection by counting)
    double sk = 0.0;
    const int kk = min(k, rows);
    for (int r = 0; r < rows; ++r) {
      int larger = 0;
      for (int q = 0; q < rows; ++q) larger += (s_n[q] > s_n[r]) || (s_n[q] == s_n[r] && q < r);
      if (larger == kk - 1) sk = s_n[r];
    }
    if (rows < kq || base_always) slast = fmax(slast, base_floor2 * s1);        // (squared norms)
    if (lvl && lvl[b]) slast = fmax(slast, (lvl[b] >= 2 ? lvl2_floor2 : lvl_floor2) * s1);
    const bool ok = rows >= k && s1 > 0.0 && sk > 0.0 && 3e-15 * (s1 / sk) * sqrt(slast / s1) <= tol;
    if (!ok) flag[b] = 0;
  }
}

// Guard of the pivoted form of the route (round 6): resid[b] = the largest pivot the cap cut off, relative to max diag(M M^T) ~ sigma_1^2
// (0: the factor took every direction above its threshold).  After one step of subspace iteration what is left of it inside the kept
// directions is bounded by resid (sigma_1 / sigma_k)^2; above `tol`, or with fewer than k directions in a capped subspace, the walker
// leaves the route.
template <typename TZ>
__global__ __launch_bounds__(256) void f64_pivot_guard_kernel(const TZ *__restrict__ Zg, long wZ, int len, const int *__restrict__ kz, int k,
                                                              const double *__restrict__ resid, double tol, int *__restrict__ flag) {
  const int b = blockIdx.x;
  if (flag[b] >= 0) return;
  __shared__ double s_n[64];
  const int rows = min(kz[b], 64), lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const TZ *Z = Zg + (long)b * wZ;
  for (int r = wave; r < rows; r += 4) {
    double a = 0.0;
    for (int c = lane; c < len; c += 64) a += abs2_of(Z[(long)r * len + c]);
    a = wave_sum(a);
    if (lane == 0) s_n[r] = a;           // squared norms
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double s1 = 0.0;
    for (int r = 0; r < rows; ++r) s1 = fmax(s1, s_n[r]);
    double sk = 0.0;
    const int kk = min(k, rows);
    for (int r = 0; r < rows; ++r) {
      int larger = 0;
      for (int q = 0; q < rows; ++q) larger += (s_n[q] > s_n[r]) || (s_n[q] == s_n[r] && q < r);
      if (larger == kk - 1) sk = s_n[r];
    }
    const double rs = resid[b];
    const bool ok = rows >= 1 && s1 > 0.0 && (rs == 0.0 || (rows >= k && sk > 0.0 && rs * (s1 / sk) <= tol));
    if (!ok) flag[b] = 0;
  }
}

// live rows of M for the walkers off the route (the general kernels take them), 0 for the walkers on it.  `early` (optional): the
// walkers that left at the first check are being handled on the side stream already (early[b] >= 0): they count 0 rows here and
// `late_flag` (optional) marks what is left for the main stream (0: left the route later, -1: nothing to do)
__global__ void f64_route_fallback_kernel(const int *__restrict__ flag, const int *__restrict__ rows_m, int nbatch, int *__restrict__ fb,
                                          const int *__restrict__ early = nullptr, int *__restrict__ late_flag = nullptr) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nbatch) return;
  const bool off = flag[b] >= 0 && !(early && early[b] >= 0);
  fb[b] = off ? rows_m[b] : 0;
  if (late_flag) late_flag[b] = off ? 0 : -1;
}

// Two-level form of the route: of the walkers with more than 128 live rows of M (hiflag) those whose factor B kept at most
// `cap` rows (mB) go through a second compression (flag2 / rows2); all other walkers of the route run their Jacobi on B itself
// (flagA / rowsA).
__global__ void mid_split_kernel(const int *__restrict__ midflag, const int *__restrict__ hiflag, const int *__restrict__ mB, int cap,
                                 int nbatch, int *__restrict__ flagA, int *__restrict__ rowsA, int *__restrict__ flag2,
                                 int *__restrict__ rows2, int *__restrict__ big_list = nullptr, int *__restrict__ big_count = nullptr) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nbatch) return;
  const bool mid = midflag[b] < 0, hi = hiflag[b] < 0;
  const int r = mB[b];
  const bool two = mid && hi && r > 0 && r <= cap;
  flag2[b] = two ? -1 : 0;
  rows2[b] = two ? r : 0;
  const bool a = mid && !two;
  flagA[b] = a ? -1 : 0;
  rowsA[b] = a ? r : 0;
  // walkers whose factor kept more than `cap` rows: the list of the 256-row Jacobi (big_count zeroed by the caller)
  if (big_list && a && r > cap) big_list[atomicAdd(big_count, 1)] = b;
}

// x[b][0..n) /= |x|;  logscale[b] += log|x|;  zero / non-finite norm sets flag[b] = 1.
template <typename T>
__global__ __launch_bounds__(256) void normalize_kernel(T *__restrict__ Xg, long wX, int n,
                                                        double *__restrict__ logscale, int *__restrict__ flag,
                                                        const int *__restrict__ ndyn, int ndyn_mul) {
  __shared__ double s_red[4];
  __shared__ double s_nrm;
  const int tid = threadIdx.x;
  T *X = Xg + (long)blockIdx.x * wX;
  if (ndyn) n = max(0, min(n, ndyn[blockIdx.x] * ndyn_mul));
  double a = 0.0;
  for (int i = tid; i < n; i += 256) a += abs2_of(X[i]);
  a = wave_sum(a);
  if ((tid & 63) == 0) s_red[tid >> 6] = a;
  __syncthreads();
  if (tid == 0) s_nrm = sqrt(s_red[0] + s_red[1] + s_red[2] + s_red[3]);
  __syncthreads();
  const double nrm = s_nrm;
  if (!(nrm > 0.0) || !isfinite(nrm)) {
    if (tid == 0 && flag) flag[blockIdx.x] = 1;
    return;
  }
  const double inv = 1.0 / nrm;
  for (int i = tid; i < n; i += 256) X[i] = scaled(X[i], inv);
  if (tid == 0 && logscale) logscale[blockIdx.x] += log(nrm);
}

// The same with ONE WAVE per batch entry (four entries per block) for short tensors: no barrier, four times the entries in
// flight -- a block of 256 threads per entry spends its time in the chain load -> reduce -> barrier -> scale for a few
// hundred elements.
template <typename T>
__global__ __launch_bounds__(256) void normalize_wave_kernel(T *__restrict__ Xg, long wX, int n, double *__restrict__ logscale,
                                                             int *__restrict__ flag, const int *__restrict__ ndyn, int ndyn_mul,
                                                             int nbatch) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= nbatch) return;
  T *X = Xg + (long)b * wX;
  if (ndyn) n = max(0, min(n, ndyn[b] * ndyn_mul));
  double a = 0.0;
  for (int i = lane; i < n; i += 64) a += abs2_of(X[i]);
  a = wave_sum(a);
  const double nrm = sqrt(a);
  if (!(nrm > 0.0) || !isfinite(nrm)) {
    if (lane == 0 && flag) flag[b] = 1;
    return;
  }
  const double inv = 1.0 / nrm;
  for (int i = lane; i < n; i += 64) X[i] = scaled(X[i], inv);
  if (lane == 0 && logscale) logscale[b] += log(nrm);
}

// out[b] = max_w v[b][w]  (v[b] == nullptr: -1); one block per entry of the pointer table
// (tables from index neg_from on are read as "is any entry negative": 1 / 0 -- a flag some kernel left on a walker)
__global__ __launch_bounds__(256) void max_over_walkers_kernel(const int *const *__restrict__ tab, int nw, int *__restrict__ out,
                                                               int neg_from = 1 << 30) {
  __shared__ int s_red[4];
  const int *v = tab[blockIdx.x];
  const bool neg = (int)blockIdx.x >= neg_from;
  int m = -1;
  if (v)
    for (int w = threadIdx.x; w < nw; w += 256) m = max(m, neg ? (v[w] < 0 ? 1 : 0) : v[w]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = max(max(s_red[0], s_red[1]), max(s_red[2], s_red[3]));
}

// Rows of Vt made orthonormal to float64 accuracy before they are used (round 4, f32 error budget of DESIGN 3e): one
// Newton-Schulz step  V <- (3/2 I - 1/2 V V^T) V  with the k x k Gram matrix and the product accumulated in float64 from the
// float32 rows (k <= 64 live rows of <= 1024 elements, staged in LDS).  The one-sided Jacobi leaves pairs of rows with
// |cos| up to its threshold 2 sqrt(len) eps32 ~ 2e-6 and norms rounded in float32; V^T V then misses a projector by that
// much, which enters <S|Psi> at FIRST order and -- on a state whose tensors repeat from site to site -- with the same sign
// at every site.  After the step the deviation is its square (1e-12), what remains is the storage rounding of V itself.
__global__ __launch_bounds__(256) void ortho_rows_kernel(float *__restrict__ Vg, long wV, int k, int len,
                                                         const int *__restrict__ klive, int ld, const int *__restrict__ skip_flag = nullptr) {
  extern __shared__ double or_smem[];
  const int b = blockIdx.x, tid = threadIdx.x;
  if (skip_flag && skip_flag[b] < 0) return;       // made orthonormal by rows_qr_kernel already
  const int kl = klive ? max(0, min(k, klive[b])) : k;
  if (kl <= 0) return;
  double *sC = or_smem;                                   // [kl][kl]
  float *sV = reinterpret_cast<float *>(sC + k * k);      // [kl][ld], ld = len + 1 (odd: conflict-free column walks)
  float *V = Vg + (long)b * wV;
  for (int e = tid; e < kl * len; e += 256) { const int a = e / len, j = e - a * len; sV[a * ld + j] = V[e]; }
  __syncthreads();
  for (int e = tid; e < kl * kl; e += 256) {
    const int a = e / kl, c = e - a * kl;
    const float *x = sV + a * ld, *y = sV + c * ld;
    double s0 = 0.0, s1 = 0.0;
    int j = 0;
    for (; j + 1 < len; j += 2) { s0 = fma((double)x[j], (double)y[j], s0); s1 = fma((double)x[j + 1], (double)y[j + 1], s1); }
    if (j < len) s0 = fma((double)x[j], (double)y[j], s0);
    sC[e] = (a == c ? 1.5 : 0.0) - 0.5 * (s0 + s1);
  }
  __syncthreads();
  for (int e = tid; e < kl * len; e += 256) {
    const int a = e / len, j = e - a * len;
    const double *cr = sC + a * kl;
    double s = 0.0;
    for (int c = 0; c < kl; ++c) s = fma(cr[c], (double)sV[c * ld + j], s);
    V[e] = (float)s;
  }
}
inline size_t ortho_rows_smem(int k, int len) { return sizeof(double) * (size_t)k * k + sizeof(float) * (size_t)k * (len + 1); }

// The closing contraction of every trace: res[b] = sum_{a,q,c} t2[b / div2][a][q][c] * t5[b / div5][c][q][a]  (no conjugation:
// a plain tensor contraction, as trace.h:131-156).  One block per batch entry; t5 is staged through LDS in its own order (coalesced)
// and read back transposed, t2 streams in order; float64 (complex float64) accumulation.  Replaces a 1 x 1 output on the LDS-tiled
// tensor GEMM (64 x 64 tiles for one number: 1.36 ms per launch of 8192 walkers, 42 % of a Monte-Carlo sweep's kernel time).
template <typename T, typename AccT>
__global__ __launch_bounds__(256) void trace_dot_kernel(const T *__restrict__ t2g, long w2, int div2, const T *__restrict__ t5g, long w5,
                                                        int div5, int d0, int d1, int d2, AccT *__restrict__ res, int use_lds) {
  extern __shared__ unsigned char td_smem[];
  T *s5 = reinterpret_cast<T *>(td_smem);
  __shared__ AccT s_red[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const T *t2 = t2g + (long)(b / div2) * w2, *t5 = t5g + (long)(b / div5) * w5;
  const int n = d0 * d1 * d2;
  if (use_lds) {
    for (int e = tid; e < n; e += 256) s5[e] = t5[e];
    __syncthreads();
  }
  const T *src5 = use_lds ? s5 : t5;
  AccT acc = AccT(0);
  const int d12 = d1 * d2;
  for (int e = tid; e < n; e += 256) {
    const int a = e / d12, r = e - a * d12, q = r / d2, c = r - q * d2;      // t2[a][q][c]
    acc += AccT(t2[e]) * AccT(src5[(c * d1 + q) * d0 + a]);                  // t5[c][q][a]
  }
  if constexpr (is_cplx<AccT>::value) { acc.re = wave_sum(acc.re); acc.im = wave_sum(acc.im); }
  else acc = wave_sum(acc);
  if ((tid & 63) == 0) s_red[tid >> 6] = acc;
  __syncthreads();
  if (tid == 0) res[b] = s_red[0] + s_red[1] + s_red[2] + s_red[3];
}

// error-budget experiments: a float64 buffer rounded to float32 values in place (Engine::inject)
__global__ void round_f32_kernel(double *p, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = (double)(float)p[i];
}

template <typename T>
__global__ void fill_kernel(T *p, long n, T v) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// hole of one site for every walker -> the resident hole store (grid.y = walker)
template <typename T>
__global__ void store_hole_kernel(const T *__restrict__ src, long n, T *__restrict__ dst, long wdst,
                                  const double *__restrict__ ls, double *__restrict__ ls_dst, int ls_stride) {
  const int w = blockIdx.y;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[(long)w * wdst + i] = src[(long)w * n + i];
  if (blockIdx.x == 0 && threadIdx.x == 0) ls_dst[(long)w * ls_stride] = ls[w];
}

// S_O[site][config_w(site)] += f_w hole_w(site),  S_EO += e_w f_w hole_w(site), f_w = sgn_w exp(ls - logabs_w)
// (mc_energy_grad_evaluator.h:257-278 / exact_summation_energy_evaluator.h:218-239, fused over walkers;
// deterministic: one thread owns one element of one site and loops over the walkers).
template <typename T>
__global__ __launch_bounds__(256) void grad_accumulate_kernel(const T *__restrict__ holes, const double *__restrict__ holes_ls,
                                                              const int *__restrict__ cfg, const double *__restrict__ logf,
                                                              const double *__restrict__ sgn, const double *__restrict__ ew,
                                                              double *__restrict__ so, double *__restrict__ seo, int nw,
                                                              int sites, long slot, int dp) {
  const int site = blockIdx.y;
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= slot) return;
  for (int s = 0; s < dp; ++s) {
    double a = 0.0, b = 0.0;
    for (int w = 0; w < nw; ++w) {
      if (cfg[(long)w * sites + site] != s) continue;
      const double f = sgn[w] * exp(holes_ls[(long)w * sites + site] + logf[w]);
      const double v = f * (double)holes[((long)w * sites + site) * slot + e];
      a += v;
      b += ew[w] * v;
    }
    so[((long)site * dp + s) * slot + e] += a;
    seo[((long)site * dp + s) * slot + e] += b;
  }
}

// the same for the complex element type: v = conj(hole) * phase(psi) * |psi|^(+-1) (exp(logf)), S_O += v, S_EO += conj(E_loc) v;
// accumulators = interleaved (re, im) doubles
template <typename T>
__global__ __launch_bounds__(256) void grad_accumulate_cplx_kernel(const T *__restrict__ holes, const double *__restrict__ holes_ls,
                                                                   const int *__restrict__ cfg, const double *__restrict__ logf,
                                                                   const double *__restrict__ ph_re, const double *__restrict__ ph_im,
                                                                   const double *__restrict__ ec_re, const double *__restrict__ ec_im,
                                                                   double *__restrict__ so, double *__restrict__ seo, int nw,
                                                                   int sites, long slot, int dp) {
  const int site = blockIdx.y;
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= slot) return;
  for (int s = 0; s < dp; ++s) {
    double ar = 0.0, ai = 0.0, br = 0.0, bi = 0.0;
    for (int w = 0; w < nw; ++w) {
      if (cfg[(long)w * sites + site] != s) continue;
      const double f = exp(holes_ls[(long)w * sites + site] + logf[w]);
      const T hv = holes[((long)w * sites + site) * slot + e];
      const double hr = (double)hv.re, hi = -(double)hv.im;            // Dag(hole)
      const double vr = f * (hr * ph_re[w] - hi * ph_im[w]), vi = f * (hr * ph_im[w] + hi * ph_re[w]);
      ar += vr; ai += vi;
      br += ec_re[w] * vr - ec_im[w] * vi;
      bi += ec_re[w] * vi + ec_im[w] * vr;
    }
    const long q = 2 * (((long)site * dp + s) * slot + e);
    so[q] += ar; so[q + 1] += ai;
    seo[q] += br; seo[q + 1] += bi;
  }
}

}  // namespace pepsgpu
