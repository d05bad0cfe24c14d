// Factorisation kernels of the chi-truncation for COMPLEX element types (TenElemT = QLTEN_Complex in the reference).
// Parity-grade, not tuned: straightforward one-block-per-walker kernels with the same mathematics, thresholds and output
// contract as their real counterparts in linalg.h (chol_upper_kernel, jacobi_rows_kernel); the tuned rank-adaptive
// kernels (MFMA, register-resident Jacobi, Gram-free factor) are real-only.
//   R^H R = G = P^H P (Hermitian, float64 complex)       replaces the R of qlten::QR, bmps_impl.h:821
//   rows of M = R T  ->  sigma_k v_k^H by complex Hestenes rotations   replaces qlten::SVD, bmps_impl.h:235-238
#pragma once
#include "linalg.h"

namespace pepsgpu {

// In-place upper Cholesky of the Hermitian PSD matrix G (n x n, row-major, upper triangle read), right-looking in global
// memory; a pivot below max(n eps64, (NOISE_C eps_T)^2) max(diag) drops its row.  Rout (n x n, type T): the live rows,
// in order, scaled by 1 / sqrt(max diag), then zero rows; mlive_out[b] (optional) = live count.
template <typename T>
__global__ __launch_bounds__(1024) void chol_upper_cplx_kernel(c128 *__restrict__ Gg, long wG, int n, T *__restrict__ Rg, long wR,
                                                              int *__restrict__ mlive_out, const int *__restrict__ run_flag = nullptr,
                                                              double thresh_scale = 1.0) {
  // run_flag (optional): only the entries with run_flag[b] < 0; thresh_scale: multiplies the pivot threshold (dense route, round 5)
  if (run_flag && run_flag[blockIdx.x] >= 0) return;
  // (any block size up to 1024: round 5 launches 1024 threads -- a pivot step is one barrier whatever the width, and the trailing
  // update of an order-256 matrix has 255 rows to hand out)
  __shared__ double s_red[16], s_maxd;
  __shared__ double s_piv[1024], s_nrm[1024];
  __shared__ short s_list[1024], s_pos[1024];
  __shared__ int s_nl, s_cnt;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwv = blockDim.x >> 6, nth = blockDim.x;
  c128 *G = Gg + (long)blockIdx.x * wG;
  T *R = Rg + (long)blockIdx.x * wR;
  double md = 0.0;
  for (int i = tid; i < n; i += nth) md = fmax(md, G[(long)i * n + i].re);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) md = fmax(md, __shfl_xor(md, o, 64));
  if (lane == 0) s_red[wave] = md;
  if (tid == 0) s_nl = 0;
  __syncthreads();
  if (tid == 0) { double t = 0.0; for (int w = 0; w < nwv; ++w) t = fmax(t, s_red[w]); s_maxd = t; }
  __syncthreads();
  const double maxd = s_maxd;
  const double eT = NOISE_C * (double)Eps<T>::v;
  const double thresh = fmax((double)n * 2.220446049250313e-16, eT * eT) * maxd * thresh_scale;
  for (int j = 0; j < n; ++j) {
    const double piv = G[(long)j * n + j].re;
    if (!(piv > thresh)) continue;                      // block-uniform
    if (tid == 0) { s_piv[s_nl] = piv; s_list[s_nl] = (short)j; s_nl = s_nl + 1; }
    const double invp = jr_rcp64(piv);
    for (int i = j + 1 + wave; i < n; i += nwv) {         // G[i][r] -= conj(G[j][i]) G[j][r] / piv,  r >= i
      const c128 f = conj_of(G[(long)j * n + i]) * invp;
      for (int r = i + lane; r < n; r += 64) G[(long)i * n + r] -= f * G[(long)j * n + r];
    }
    __syncthreads();
  }
  __syncthreads();
  const int nl = s_nl;
  for (int q = wave; q < nl; q += nwv) {
    const int j = s_list[q];
    double a = 0.0;
    for (int r = j + lane; r < n; r += 64) a += abs2_of(G[(long)j * n + r]);
    a = wave_sum(a);
    if (lane == 0) s_nrm[q] = a / s_piv[q];
  }
  __syncthreads();
  if (tid == 0) {
    double f = 0.0;
    for (int q = 0; q < nl; ++q) f += s_nrm[q];
    const double nfloor = eT * eT * f;
    int cnt = 0;
    for (int q = 0; q < nl; ++q) s_pos[q] = s_nrm[q] > nfloor ? (short)cnt++ : (short)-1;
    s_cnt = cnt;
    if (mlive_out) mlive_out[blockIdx.x] = cnt;
  }
  __syncthreads();
  const double sc = maxd > 0.0 ? 1.0 / sqrt(maxd) : 1.0;
  for (int q = wave; q < nl; q += nwv) {
    const int pos = s_pos[q];
    if (pos < 0) continue;
    const int j = s_list[q];
    const double f = sc / sqrt(s_piv[q]);
    for (int r = lane; r < n; r += 64) R[(long)pos * n + r] = r >= j ? T(scaled(G[(long)j * n + r], f)) : T(0);
  }
  for (int e = tid + s_cnt * n; e < n * n; e += nth) R[e] = T(0);
}

// One-sided Jacobi on the rows of a complex M (m x len, row stride ld) in global memory: round-robin tournament, one wave per
// row pair.  A pair (x, y) with gamma = x^H y: y is first turned by the phase conj(gamma) / |gamma| (rows of Vt are defined up
// to a phase), which makes the inner product real, then the real Hestenes rotation applies.  Threshold, noise floor and
// termination as jacobi_rows_kernel.
// complex rotation of a row pair: phase of the inner product and (c, s) from the hardware reciprocal / rsqrt estimates + Newton steps
// (jr_rcp64 / jr_rsq64 of linalg.h) instead of five IEEE square roots and five divisions per pair
__device__ __forceinline__ bool jr_rotation_cplx(const double alpha, const double beta, const double gre, const double gim, const double tol,
                                                 const double floor2, double &phre, double &phim, double &cd, double &sd) {
  const double g2 = gre * gre + gim * gim;
  if (!(g2 > tol * tol * (alpha * beta) && alpha > floor2 && beta > floor2)) return false;
  const double ig = jr_rsq64(g2);                 // 1 / |gamma|
  phre = gre * ig; phim = -gim * ig;
  const double zeta = 0.5 * (beta - alpha) * ig;
  const double az = fabs(zeta), w = fma(az, az, 1.0);
  const double td = copysign(jr_rcp64(az + w * jr_rsq64(w)), zeta);
  cd = jr_rsq64(fma(td, td, 1.0));
  sd = cd * td;
  return true;
}

template <typename T>
__global__ __launch_bounds__(1024) void jacobi_rows_cplx_kernel(T *__restrict__ Mg, long wM, int m, int len, int ld, int max_sweeps,
                                                                int *__restrict__ sweeps_out, const int *__restrict__ run_flag = nullptr,
                                                                int run_if_neg = 1) {
  // run_flag (optional): with run_if_neg = 1 only the entries with run_flag[b] < 0 run, with 0 only those with run_flag[b] >= 0
  if (run_flag && ((run_flag[blockIdx.x] < 0) != (run_if_neg != 0))) return;
  typedef typename real_of<T>::type R;
  __shared__ int s_rot;
  __shared__ double s_fro[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  T *M = Mg + (long)blockIdx.x * wM;
  {
    double f = 0.0;
    for (int e = tid; e < m * len; e += blockDim.x) f += abs2_of(M[(long)(e / len) * ld + (e % len)]);
    f = wave_sum(f);
    if (lane == 0) s_fro[wave] = f;
    __syncthreads();
    if (tid == 0) { double t = 0.0; for (int w = 0; w < nw; ++w) t += s_fro[w]; s_fro[0] = t; }
    __syncthreads();
  }
  const double floor2 = NOISE_C * NOISE_C * (double)Eps<T>::v * (double)Eps<T>::v * s_fro[0];
  const double tol = 2.0 * sqrt((double)len) * (double)Eps<T>::v;
  // Round 3: the tournament runs over the LIVE rows only (squared norm above the floor below which a row is never rotated
  // anyway).  The complex path keeps static shapes: after the factor's rank compaction most of the m rows of M = R T are zero
  // on a state of low numerical rank, and every pair of them still cost two row reads (C4 shapes: 12 amp/s).
  __shared__ short s_live[1024];
  __shared__ float s_n2[1024];
  __shared__ int s_ml;
  for (int r = wave; r < m; r += nw) {
    double a = 0.0;
    for (int c = lane; c < len; c += 64) a += abs2_of(M[(long)r * ld + c]);
    a = wave_sum(a);
    if (lane == 0) s_n2[r] = (float)(a > floor2 ? 1.f : 0.f);
  }
  __syncthreads();
  if (tid == 0) {
    int cnt = 0;
    for (int r = 0; r < m; ++r)
      if (s_n2[r] > 0.f) s_live[cnt++] = (short)r;
    s_ml = cnt;
  }
  __syncthreads();
  const int ml = s_ml;
  const int lp = ml + (ml & 1);
  int sweep = 0;
  for (; sweep < max_sweeps && ml > 1; ++sweep) {
    if (tid == 0) s_rot = 0;
    __syncthreads();
    for (int r = 0; r < lp - 1; ++r) {
      auto pair_of = [&](int p, int &a, int &b) -> bool {      // pair p of tournament step r; false: a bye or a dead row
        if (p >= lp / 2) return false;
        if (p == 0) { a = lp - 1; b = r; }
        else { a = r + p; a -= a >= lp - 1 ? lp - 1 : 0; b = r - p; b += b < 0 ? lp - 1 : 0; }      // (0 <= r < lp - 1, 0 < p < lp / 2: no division)
        if (a > b) { const int t = a; a = b; b = t; }
        return b < ml;
      };
      if (len <= 256) {
        // rows of up to 256 elements (the factors and Z of the dense route, round 5): both rows of a pair stay in registers between the
        // inner products and the rotation (one read and one write of a row instead of two reads and a write), and a wave works on TWO
        // pairs at a time so that the loads of the second travel under the reductions of the first
        for (int p0 = wave; p0 < lp / 2; p0 += 2 * nw) {
          int a[2], b[2];
          bool ok[2];
          T xr[2][4], yr[2][4];
          double alpha[2], beta[2], gre[2], gim[2];
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            ok[q] = pair_of(p0 + q * nw, a[q], b[q]);
            alpha[q] = beta[q] = gre[q] = gim[q] = 0.0;
            if (!ok[q]) continue;
            const T *pa = M + (long)s_live[a[q]] * ld, *pb = M + (long)s_live[b[q]] * ld;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int c = lane + 64 * j;
              xr[q][j] = c < len ? pa[c] : T(0);
              yr[q][j] = c < len ? pb[c] : T(0);
            }
          }
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            if (!ok[q]) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              alpha[q] += abs2_of(xr[q][j]); beta[q] += abs2_of(yr[q][j]);
              const T g = conj_of(xr[q][j]) * yr[q][j];
              gre[q] += (double)g.re; gim[q] += (double)g.im;
            }
            alpha[q] = wave_sum(alpha[q]); beta[q] = wave_sum(beta[q]); gre[q] = wave_sum(gre[q]); gim[q] = wave_sum(gim[q]);
          }
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            if (!ok[q]) continue;
            double phre, phim, cd, sd;
            if (jr_rotation_cplx(alpha[q], beta[q], gre[q], gim[q], tol, floor2, phre, phim, cd, sd)) {
              const T ph = T(R(phre), R(phim));
              T *pa = M + (long)s_live[a[q]] * ld, *pb = M + (long)s_live[b[q]] * ld;
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                const int c = lane + 64 * j;
                if (c < len) {
                  const T x = xr[q][j], y = yr[q][j] * ph;
                  pa[c] = scaled(x, cd) - scaled(y, sd);
                  pb[c] = scaled(x, sd) + scaled(y, cd);
                }
              }
              if (lane == 0) atomicAdd(&s_rot, 1);
            }
          }
        }
      } else {
      for (int p = wave; p < lp / 2; p += nw) {
        int a, b;
        if (!pair_of(p, a, b)) continue;
        T *pa = M + (long)s_live[a] * ld, *pb = M + (long)s_live[b] * ld;
        double alpha = 0.0, beta = 0.0, gre = 0.0, gim = 0.0;
        for (int c = lane; c < len; c += 64) {
          const T x = pa[c], y = pb[c];
          alpha += abs2_of(x); beta += abs2_of(y);
          const T g = conj_of(x) * y;
          gre += (double)g.re; gim += (double)g.im;
        }
        alpha = wave_sum(alpha); beta = wave_sum(beta); gre = wave_sum(gre); gim = wave_sum(gim);
        double phre, phim, cd, sd;
        if (jr_rotation_cplx(alpha, beta, gre, gim, tol, floor2, phre, phim, cd, sd)) {
          const T ph = T(R(phre), R(phim));
          for (int c = lane; c < len; c += 64) {
            const T x = pa[c], y = pb[c] * ph;
            pa[c] = scaled(x, cd) - scaled(y, sd);
            pb[c] = scaled(x, sd) + scaled(y, cd);
          }
          if (lane == 0) atomicAdd(&s_rot, 1);
        }
      }
      }
      __threadfence_block();
      __syncthreads();
    }
    const int rot = s_rot;
    __syncthreads();
    if (rot == 0) { ++sweep; break; }
  }
  if (tid == 0 && sweeps_out) sweeps_out[blockIdx.x] = sweep;
}

}  // namespace pepsgpu
