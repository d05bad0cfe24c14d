// Preconditioner of the mid-rank truncation route (Engine::absorb_impl) in ONE kernel per walker:
//
//     G = M M^T   (n x n float64, n = live rows of M <= 128; v_mfma_f64_16x16x4_f64 from LDS-staged chunks of M)
//     B^T B = G   (upper Cholesky in LDS, right-looking, semi-definite safe: a pivot below the noise of the T-typed data
//                  drops its row -- the rule of chol_upper_kernel)
//     B -> global as type T, live rows compacted, scaled by 1 / sqrt(max diag), row stride ld; mB[b] = live rows
//
// The one-sided Jacobi then runs on the rows of B (jacobi_rows_regx_kernel) instead of on the rows of M: the
// preconditioned Jacobi SVD of Drmac / Veselic, restated for "M is wide and only R^T R = M M^T is needed".
// Before: a batched LDS-tiled GEMM for G (64 x 64 tiles on a 70 x 70 result: 1.5 TF) + the blocked Cholesky through
// global memory, two launches and a round trip of G; here G never leaves the CU.
#pragma once
#include "common.h"
#include "linalg.h"

namespace pepsgpu {

typedef double tm_f64x4 __attribute__((ext_vector_type(4)));
constexpr int TM_KC = 32;    // columns of M staged per chunk
constexpr int TM_LDM = TM_KC + 2;   // row pitch of the chunk: 16 rows x 2 k-lanes of an operand read hit 32 different banks

inline size_t mid_gram_chol_smem_bytes(int cap) {
  return sizeof(double) * ((size_t)cap * (cap + 1) + 2 * (size_t)cap) + sizeof(float) * (size_t)cap * TM_LDM + sizeof(short) * 2 * (size_t)cap + 64;
}

// run_flag[b] < 0: the entry is on the route; n = nrows[b] in (lo, cap] is taken by this launch (another launch with a
// different cap takes the rest).
template <typename T>
__global__ __launch_bounds__(256) void mid_gram_chol_kernel(const T *__restrict__ Mg, long wM, int uk, const int *__restrict__ nrows,
                                                            const int *__restrict__ run_flag, int lo, int cap,
                                                            T *__restrict__ Bg, long wB, int ld, int *__restrict__ mB) {
  static_assert(sizeof(T) == 4, "the mid route is f32 only (the f64 mode keeps the direct Jacobi)");
  const int b = blockIdx.x;
  if (run_flag && run_flag[b] >= 0) return;
  const int n = nrows[b];
  if (n <= lo || n > cap) return;
  extern __shared__ double tm_smem[];
  const int ldG = cap + 1;
  double *sG = tm_smem;                                 // [cap][ldG] upper triangle of G, then of the factor
  double *sPiv = sG + (size_t)cap * ldG;                // [cap] pivot of a live row (0 = dropped)
  double *sNrm = sPiv + cap;                            // [cap] squared norm of a factor row
  float *sM = reinterpret_cast<float *>(sNrm + cap);    // [cap][TM_KC + 1] chunk of M
  short *sList = reinterpret_cast<short *>(sM + (size_t)cap * TM_LDM);   // [cap] live rows in order
  short *sPos = sList + cap;                            // [cap] output position, -1 = dropped
  __shared__ double s_red[4], s_maxd, s_fro;
  __shared__ int s_nl, s_cnt;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const T *M = Mg + (long)b * wM;
  T *B = Bg + (long)b * wB;

  // ---- G = M M^T: 16 x 16 tiles on or above the diagonal, dealt round-robin to the four waves ----
  const int nt = (n + 15) >> 4, ntiles = nt * (nt + 1) / 2;
  constexpr int TPW = 9;                                // 36 tiles (n = 128) / 4 waves
  int ti[TPW], tj[TPW];
#pragma unroll
  for (int q = 0; q < TPW; ++q) {
    int t = wave + 4 * q, i = 0;
    if (t < ntiles) { while (t >= nt - i) { t -= nt - i; ++i; } ti[q] = i; tj[q] = i + t; }
    else { ti[q] = -1; tj[q] = -1; }
  }
  tm_f64x4 acc[TPW];
#pragma unroll
  for (int q = 0; q < TPW; ++q)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[q][r] = 0.0;
  const int r16 = lane & 15, k4 = lane >> 4;
  for (int kc = 0; kc < uk; kc += TM_KC) {
    const int kw = min(TM_KC, uk - kc);
    __syncthreads();
    for (int e = tid; e < n * TM_KC; e += 256) {
      const int r = e / TM_KC, c = e % TM_KC;
      sM[r * TM_LDM + c] = c < kw ? (float)M[(long)r * uk + kc + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < TPW; ++q) {
      if (ti[q] < 0) continue;                           // wave-uniform
      const int ra = 16 * ti[q] + r16, rb = 16 * tj[q] + r16;
#pragma unroll
      for (int s = 0; s < TM_KC / 4; ++s) {
        const double a = ra < n ? (double)sM[ra * TM_LDM + 4 * s + k4] : 0.0;
        const double bb = rb < n ? (double)sM[rb * TM_LDM + 4 * s + k4] : 0.0;
        acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, acc[q], 0, 0, 0);
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < TPW; ++q) {
    if (ti[q] < 0) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {                        // acc[r] = C[(lane >> 4) + 4 r][lane & 15]
      const int i = 16 * ti[q] + k4 + 4 * r, j = 16 * tj[q] + r16;
      if (i < n && j < n) sG[i * ldG + j] = acc[q][r];
    }
  }
  __syncthreads();

  // ---- upper Cholesky, right-looking, one barrier per live pivot ----
  double md = 0.0;
  for (int i = tid; i < n; i += 256) md = fmax(md, sG[i * ldG + i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) md = fmax(md, __shfl_xor(md, o, 64));
  if (lane == 0) s_red[wave] = md;
  if (tid == 0) s_nl = 0;
  __syncthreads();
  if (tid == 0) s_maxd = fmax(fmax(s_red[0], s_red[1]), fmax(s_red[2], s_red[3]));
  __syncthreads();
  const double maxd = s_maxd;
  const double eT = NOISE_C * (double)Eps<T>::v;
  const double thresh = fmax((double)n * 2.220446049250313e-16, eT * eT) * maxd;
  for (int j = 0; j < n; ++j) {
    const double piv = sG[j * ldG + j];                  // every thread reads the same, settled value
    if (!(piv > thresh)) {                               // dead direction: its row takes no part (block-uniform branch)
      if (tid == 0) sPiv[j] = 0.0;
      continue;
    }
    if (tid == 0) { sPiv[j] = piv; sList[s_nl] = (short)j; s_nl = s_nl + 1; }
    const double invp = 1.0 / piv;
    // trailing update with the UNSCALED row j: G[i][r] -= G[j][i] G[j][r] / piv, i > j, r >= i
    for (int i = j + 1 + wave; i < n; i += 4) {          // a wave per trailing row, lanes along the row
      const double f = sG[j * ldG + i] * invp;
      for (int r = i + lane; r < n; r += 64) sG[i * ldG + r] -= f * sG[j * ldG + r];
    }
    __syncthreads();
  }
  __syncthreads();
  const int nl = s_nl;
  // ---- rank compaction (rows with norm below NOISE_C eps_T |B|_F are dropped), as chol_upper_kernel ----
  for (int q = wave; q < nl; q += 4) {
    const int j = sList[q];
    double a = 0.0;
    for (int r = j + lane; r < n; r += 64) { const double x = sG[j * ldG + r]; a += x * x; }
    a = wave_sum(a);
    if (lane == 0) sNrm[q] = a / sPiv[j];
  }
  __syncthreads();
  if (tid == 0) {
    double f = 0.0;
    for (int q = 0; q < nl; ++q) f += sNrm[q];
    const double nfloor = eT * eT * f;
    int cnt = 0;
    for (int q = 0; q < nl; ++q) sPos[q] = sNrm[q] > nfloor ? (short)cnt++ : (short)-1;
    s_cnt = cnt;
    mB[b] = cnt;
  }
  __syncthreads();
  const double sc = maxd > 0.0 ? 1.0 / sqrt(maxd) : 1.0;
  for (int q = wave; q < nl; q += 4) {
    const int pos = sPos[q];
    if (pos < 0) continue;
    const int j = sList[q];
    const double f = sc / sqrt(sPiv[j]);
    // the Jacobi reads whole rows of the ld-wide buffer: columns outside [j, n) are written as zeros
    for (int r = lane; r < ld; r += 64) B[(long)pos * ld + r] = (r >= j && r < n) ? T(sG[j * ldG + r] * f) : T(0);
  }
}

template <typename T>
inline void launch_mid_gram_chol(hipStream_t s, int nbatch, const T *M, long wM, int uk, const int *nrows, const int *run_flag,
                                 int GS, T *B, long wB, int *mB) {
  // two size classes: most walkers of a dense state have 50-80 live rows (52 KB of LDS: three per CU)
  const int caps[2] = {80, 128};
  int lo = 0;
  for (int c = 0; c < 2; ++c) {
    const int cap = std::min(caps[c], GS);
    if (cap <= lo) break;
    const size_t smem = mid_gram_chol_smem_bytes(cap);
    allow_dynamic_lds(reinterpret_cast<const void *>(&mid_gram_chol_kernel<T>), smem);
    hipLaunchKernelGGL(mid_gram_chol_kernel<T>, dim3(nbatch), dim3(256), smem, s, M, wM, uk, nrows, run_flag, lo, cap, B, wB, GS, mB);
    PG_CHECK_HIP(hipGetLastError());
    lo = cap;
  }
}

}  // namespace pepsgpu
