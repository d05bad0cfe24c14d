// Orthonormal rows of the projected truncation input in one launch (round 6): V = rows of  L^-1 D V',  L L^T = D (V' V'^T) D.
//
// Where.  On the preconditioned routes of the truncation (engine_impl.h, `mid`) the k <= chi rows V' = U^T M come out as
// sigma_q v_q^T up to the rounding of u_q -- nearly orthogonal, each contaminated by the dominant directions at 1e-7 sigma_1.  Rounds 3-5
// handed them to the one-sided Jacobi once more (jacobi_rows_grp_kernel<1,16>: 0.8 ms per launch of 8192 walkers), normalised them with
// select_rows_kernel (0.1 ms) and, on precise sites, made them orthonormal to float64 accuracy with a Newton-Schulz step
// (ortho_rows_kernel: 0.58 ms).  Only the SPAN of the rows leaves the site (Y = T V^T and V carry the same bond gauge), so any
// orthonormalisation serves; this one is a Cholesky-QR in float64 on the rows scaled to unit norm (the scaled Gram is I + E with
// |E| ~ 1e-7 sigma_1 / sigma_q: perfectly conditioned, so the factor is accurate to float64 rounding): Gram-Schmidt in the order of the
// rows, i.e. by decreasing sigma -- a small row is cleaned of the dominant directions, never the other way round.
// A row whose remaining part (after the rows before it) is below the liveness floor of select_rows_kernel (2 NOISE_C eps |V'|_F) is dropped;
// the live rows come out first, klive_out = their count, the other rows of V are zero.
//
// One 256-thread block per walker: V' (k <= 32 rows, len <= 256) staged in LDS once, the k x k Gram on v_mfma_f64_16x16x4 (one 16 x 16
// tile per wave), the 32-step factorisation with its 528 entries dealt over the block, the forward substitution one thread per column with the solution in registers.
#pragma once
#include "linalg.h"
#include "cplx.h"

namespace pepsgpu {

constexpr int RQ_K = 32, RQ_LEN = 256, RQ_LD = RQ_LEN + 4;
inline bool rows_qr_ok(int k, int len) { return k >= 1 && k <= RQ_K && len >= 4 && len <= RQ_LEN && len % 4 == 0; }

__global__ __launch_bounds__(256) void rows_qr_kernel(const float *__restrict__ Xg, long wX, int k, int len, const int *__restrict__ kdyn,
                                                      float *__restrict__ Vg, long wV, int *__restrict__ klive_out,
                                                      const int *__restrict__ run_flag) {
  const int b = blockIdx.x;
  if (run_flag && run_flag[b] >= 0) return;
  __shared__ float sV[RQ_K * RQ_LD];
  __shared__ double sS[RQ_K][RQ_K + 1];
  __shared__ double sD[RQ_K];          // 1 / |row|
  __shared__ double sInv[RQ_K];        // 1 / L[a][a] of the factor (0: dead row)
  __shared__ double sN[RQ_K];          // |row|^2
  __shared__ short sPos[RQ_K];
  __shared__ int s_cnt;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int kl = kdyn ? max(0, min(k, kdyn[b])) : k;
  const float *X = Xg + (long)b * wX;
  float *V = Vg + (long)b * wV;
  if (kl <= 0) {
    for (int e = t; e < k * len; e += 256) V[e] = 0.f;
    if (t == 0 && klive_out) klive_out[b] = 0;
    return;
  }
  // ---- stage the rows (rows beyond kl: zeros) ----
  for (int e = t; e < RQ_K * (len >> 2); e += 256) {
    const int a = e / (len >> 2), c4 = e - a * (len >> 2);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a < kl) v = *reinterpret_cast<const float4 *>(X + (long)a * len + 4 * c4);
    *reinterpret_cast<float4 *>(sV + a * RQ_LD + 4 * c4) = v;
  }
  __syncthreads();
  // ---- Gram on the f64 matrix cores: wave w owns tile (w >> 1, w & 1) ----
  {
    const int i16 = lane & 15, k4 = lane >> 4;
    const int ti = wave >> 1, tj = wave & 1;
    chb_f64x4 acc;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = 0.0;
    if (16 * ti < kl && 16 * tj < kl) {                       // (wave-uniform)
      const float *pa = sV + (16 * ti + i16) * RQ_LD + k4, *pb = sV + (16 * tj + i16) * RQ_LD + k4;
      for (int s = 0; s < len; s += 4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64((double)pa[s], (double)pb[s], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) sS[16 * ti + k4 + 4 * r][16 * tj + i16] = acc[r];      // acc[r] = C[k4 + 4 r][i16] (f64 16 x 16 x 4)
  }
  __syncthreads();
  if (t < RQ_K) {
    const double n2 = sS[t][t];
    sN[t] = n2;
    sD[t] = (t < kl && n2 > 0.0) ? 1.0 / sqrt(n2) : 0.0;
  }
  __syncthreads();
  // ---- scaled Gram (lower triangle), factored in place, right-looking: the 528 entries dealt over the block, two barriers per step ----
  int ea[3], ec[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int e = t + 256 * q;
    int ra = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5);
    while ((ra + 1) * (ra + 2) / 2 <= e) ++ra;
    while (ra * (ra + 1) / 2 > e) --ra;
    ea[q] = e < RQ_K * (RQ_K + 1) / 2 ? ra : -1;
    ec[q] = e - ra * (ra + 1) / 2;
  }
#pragma unroll
  for (int q = 0; q < 3; ++q)
    if (ea[q] >= 0) sS[ea[q]][ec[q]] *= sD[ea[q]] * sD[ec[q]];
  double fro2 = 0.0;
  for (int a = 0; a < kl; ++a) fro2 += sN[a];
  const double nfloor = 2.0 * NOISE_C * (double)Eps<float>::v * sqrt(fro2);
  __syncthreads();
  for (int j = 0; j < kl; ++j) {
    const double piv = sS[j][j];
    // remaining part of row j: sqrt(piv) in units of its own norm, sqrt(piv |row|^2) in absolute terms
    const bool live = sD[j] > 0.0 && piv > 0.0 && piv * sN[j] > nfloor * nfloor;       // (block-uniform)
    const double inv = live ? jr_rsq64(piv) : 0.0;
    __syncthreads();
    if (t < RQ_K && t >= j) sS[t][j] = (t == j) ? piv * inv : sS[t][j] * inv;      // (sqrt(piv) = piv / sqrt(piv); dead row: 0)
    if (t == 0) sInv[j] = inv;
    __syncthreads();
    if (live) {
#pragma unroll
      for (int q = 0; q < 3; ++q)
        if (ea[q] >= 0 && ec[q] > j) sS[ea[q]][ec[q]] = fma(-sS[ea[q]][j], sS[ec[q]][j], sS[ea[q]][ec[q]]);
    }
    __syncthreads();
  }
  if (t == 0) {
    int cnt = 0;
    for (int j = 0; j < kl; ++j) sPos[j] = sS[j][j] > 0.0 ? (short)cnt++ : (short)-1;
    for (int j = kl; j < RQ_K; ++j) sPos[j] = -1;
    s_cnt = cnt;
    if (klive_out) klive_out[b] = cnt;
  }
  __syncthreads();
  // ---- forward substitution, one thread per column: x_a = (d_a X[a][c] - sum_{b < a} L[a][b] x_b) / L[a][a]; dead rows: x = 0 ----
  const int cnt = s_cnt;
  if (t < len) {
    double x[RQ_K];
#pragma unroll
    for (int a = 0; a < RQ_K; ++a) {
      double s = 0.0;
      if (a < kl) {                                           // (block-uniform)
        const double laa = sS[a][a];
        s = sD[a] * (double)sV[a * RQ_LD + t];
#pragma unroll
        for (int q = 0; q < a; ++q) s = fma(-sS[a][q], x[q], s);
        s = laa > 0.0 ? s * sInv[a] : 0.0;        // (1 / L[a][a] kept from the factorisation: an IEEE division per row was most of this loop)
      }
      x[a] = s;
    }
#pragma unroll
    for (int a = 0; a < RQ_K; ++a)
      if (a < kl) {
        const int pos = sPos[a];
        if (pos >= 0) V[(long)pos * len + t] = (float)x[a];
      }
  }
  for (int e = t + cnt * len; e < k * len; e += 256) V[e] = 0.f;
}

// X <- L^-1 X per walker, L L^T = S (round 6, the float64 dense route): one pass of a Cholesky-QR on r <= 64 rows that are already
// in global memory together with their Gram matrix S (upper triangle, r x r, leading dimension lds) -- the rows of the pivoted factor
// B (S = B B^T), then the rows of the result once more (Cholesky-QR2: the first pass leaves eps cond(B)^2 ~ 1e-5 of
// non-orthogonality at cond(B) ~ 1e5, the second removes it).  r = rows[b] (dynamic), len <= 256 columns (one thread per column, the
// solution in registers), L in LDS.  A pivot that is not positive (rows dependent to rounding) zeroes its row.
constexpr int CS_K = 64;
__global__ __launch_bounds__(256) void chol_solve_rows_kernel(const double *__restrict__ Sg, long wS, int lds, double *__restrict__ Xg, long wX,
                                                              int len, const int *__restrict__ rows, const int *__restrict__ run_flag) {
  const int b = blockIdx.x;
  if (run_flag && run_flag[b] >= 0) return;
  __shared__ double sL[CS_K][CS_K + 1];
  __shared__ double sInv[CS_K];        // 1 / L[a][a] (0: dead row)
  const int t = threadIdx.x;
  const int r = max(0, min(CS_K, rows[b]));
  if (r <= 0) return;
  const double *S = Sg + (long)b * wS;
  double *X = Xg + (long)b * wX;
  for (int e = t; e < r * r; e += 256) {
    const int i = e / r, j = e - i * r;
    if (j >= i) sL[j][i] = S[(long)i * lds + j];         // lower triangle of the symmetric matrix from its upper storage
  }
  __syncthreads();
  for (int j = 0; j < r; ++j) {
    const double piv = sL[j][j];
    const bool live = piv > 0.0;                          // (block-uniform)
    const double inv = live ? jr_rsq64(piv) : 0.0;
    __syncthreads();
    if (t < r && t >= j) sL[t][j] = (t == j) ? piv * inv : sL[t][j] * inv;
    if (t == 0) sInv[j] = inv;
    __syncthreads();
    if (live) {
      // trailing block (a, c), j < c <= a < r, on a 16 x 16 thread grid (round 6: `e / nrem`, `e % nrem` per element with a run-time
      // nrem were two integer divisions per FMA)
      for (int a = j + 1 + (t >> 4); a < r; a += 16)
        for (int c = j + 1 + (t & 15); c <= a; c += 16) sL[a][c] = fma(-sL[a][j], sL[c][j], sL[a][c]);
    }
    __syncthreads();
  }
  // forward substitution, one thread per column, the solution in registers.  Rows in chunks of CS_CH behind block-uniform branches: a
  // run-time loop over the chunks with static register indices inside (the fully unrolled triangle -- 2016 FMAs, as many LDS reads --
  // came out of the compiler with 256 VGPRs and 464 spilled dwords, as chol_pivot_kernel's did)
  constexpr int CS_CH = 8, CS_NCH = CS_K / CS_CH;
  for (int c0 = 0; c0 < len; c0 += 256) {
    const int c = min(c0 + t, len - 1);
    double x[CS_K];
#pragma unroll
    for (int a = 0; a < CS_K; ++a) x[a] = 0.0;
    const int nch = (r + CS_CH - 1) / CS_CH;
    for (int ca = 0; ca < nch; ++ca) {
      double v[CS_CH];
#pragma unroll
      for (int a = 0; a < CS_CH; ++a) v[a] = (ca * CS_CH + a < r) ? X[(long)(ca * CS_CH + a) * len + c] : 0.0;
#pragma unroll
      for (int cq = 0; cq < CS_NCH; ++cq)
        if (cq < ca) {
#pragma unroll
          for (int k = 0; k < CS_CH; ++k) {
            const double xq = x[cq * CS_CH + k];
#pragma unroll
            for (int a = 0; a < CS_CH; ++a) v[a] = fma(-sL[ca * CS_CH + a][cq * CS_CH + k], xq, v[a]);
          }
        }
#pragma unroll
      for (int a = 0; a < CS_CH; ++a) {
        const int ra = ca * CS_CH + a;
        double sv = v[a];
#pragma unroll
        for (int q = 0; q < a; ++q) sv = fma(-sL[ra][ca * CS_CH + q], v[q], sv);
        v[a] = ra < r ? sv * sInv[ra] : 0.0;       // (1 / L[a][a] from the factorisation; 0 for a dead row)
      }
#pragma unroll
      for (int cq = 0; cq < CS_NCH; ++cq)
        if (cq == ca) {
#pragma unroll
          for (int a = 0; a < CS_CH; ++a) x[cq * CS_CH + a] = v[a];
        }
      if (c0 + t < len) {
#pragma unroll
        for (int a = 0; a < CS_CH; ++a)
          if (ca * CS_CH + a < r) X[(long)(ca * CS_CH + a) * len + c] = v[a];
      }
    }
  }
}

// The same pass for COMPLEX rows (round 6, the complex dense route): S = X X^H Hermitian (upper triangle stored), L L^H = S,
// X <- L^-1 X in place.  The solution of 64 complex rows does not fit the registers of one thread: the rows go in chunks of
// CSC_CH = 16 (the chunk in registers, the rows solved before re-read from global memory -- written by the same thread).
constexpr int CSC_CH = 16;
__global__ __launch_bounds__(256, 2) void chol_solve_rows_cplx_kernel(const c128 *__restrict__ Sg, long wS, int lds, c128 *__restrict__ Xg,
                                                                      long wX, int len, int r, const int *__restrict__ run_flag) {
  const int b = blockIdx.x;
  if (run_flag && run_flag[b] >= 0) return;
  __shared__ c128 sL[CS_K][CS_K + 1];
  __shared__ double sInv[CS_K];
  const int t = threadIdx.x;
  r = max(0, min(CS_K, r));
  if (r <= 0) return;
  const c128 *S = Sg + (long)b * wS;
  c128 *X = Xg + (long)b * wX;
  for (int e = t; e < r * r; e += 256) {
    const int i = e / r, j = e - i * r;
    if (j >= i) sL[j][i] = conj_of(S[(long)i * lds + j]);   // lower triangle of the Hermitian matrix from its upper storage
  }
  __syncthreads();
  for (int j = 0; j < r; ++j) {
    const double piv = sL[j][j].re;
    const bool live = piv > 0.0;                          // (block-uniform)
    const double inv = live ? jr_rsq64(piv) : 0.0;
    __syncthreads();
    if (t < r && t >= j) sL[t][j] = (t == j) ? c128(piv * inv, 0.0) : sL[t][j] * inv;
    if (t == 0) sInv[j] = inv;
    __syncthreads();
    if (live) {
      for (int a = j + 1 + (t >> 4); a < r; a += 16)
        for (int c = j + 1 + (t & 15); c <= a; c += 16) sL[a][c] -= sL[a][j] * conj_of(sL[c][j]);
    }
    __syncthreads();
  }
  for (int c0 = 0; c0 < len; c0 += 256) {
    const int c = c0 + t;
    if (c >= len) continue;
    for (int a0 = 0; a0 < r; a0 += CSC_CH) {
      c128 v[CSC_CH];
#pragma unroll
      for (int a = 0; a < CSC_CH; ++a) v[a] = (a0 + a < r) ? X[(long)(a0 + a) * len + c] : c128(0.0, 0.0);
      for (int q = 0; q < a0; ++q) {
        const c128 xq = X[(long)q * len + c];
#pragma unroll
        for (int a = 0; a < CSC_CH; ++a) v[a] -= sL[min(a0 + a, CS_K - 1)][q] * xq;
      }
#pragma unroll
      for (int a = 0; a < CSC_CH; ++a) {
        const int ra = min(a0 + a, CS_K - 1);
        c128 sv = v[a];
#pragma unroll
        for (int q = 0; q < a; ++q) sv -= sL[ra][a0 + q] * v[q];
        v[a] = a0 + a < r ? sv * sInv[ra] : c128(0.0, 0.0);
      }
#pragma unroll
      for (int a = 0; a < CSC_CH; ++a)
        if (a0 + a < r) X[(long)(a0 + a) * len + c] = v[a];
    }
  }
}

// A fixed table of signs (+1 / -1, the same for every walker): the start of the randomised range finder of the complex route
template <typename T>
__global__ void sign_table_kernel(T *__restrict__ out, int rows, int cols) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= rows * cols) return;
  unsigned h = (unsigned)e * 2654435761u;
  h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
  out[e] = T((h & 1u) ? 1.0 : -1.0);
}

// Q[b][j][:] = M[b][piv[b][j]][:] for the j < rows[b] pivot rows a selection run of chol_pivot_kernel listed (pivot order); len elements per row
template <typename T>
__global__ __launch_bounds__(256) void gather_rows_kernel(const T *__restrict__ Mg, long wM, int len, const int *__restrict__ piv, int slots,
                                                          const int *__restrict__ rows, T *__restrict__ Qg, long wQ,
                                                          const int *__restrict__ run_flag) {
  const int b = blockIdx.y, j = blockIdx.x;
  if (run_flag && run_flag[b] >= 0) return;
  if (j >= rows[b]) return;
  const int p = piv[(long)b * slots + j];
  if (p < 0) return;
  const T *src = Mg + (long)b * wM + (long)p * len;
  T *dst = Qg + (long)b * wQ + (long)j * len;
  for (int c = threadIdx.x; c < len; c += 256) dst[c] = src[c];
}

inline void launch_rows_qr(hipStream_t s, int nbatch, const float *X, long wX, int k, int len, const int *kdyn, float *V, long wV,
                           int *klive_out, const int *run_flag) {
  PG_REQUIRE(rows_qr_ok(k, len), 1, "rows_qr: k <= 32 rows of <= 256 elements (a multiple of 4)");
  hipLaunchKernelGGL(rows_qr_kernel, dim3(nbatch), dim3(256), 0, s, X, wX, k, len, kdyn, V, wV, klive_out, run_flag);
  PG_CHECK_HIP(hipGetLastError());
}

}  // namespace pepsgpu
