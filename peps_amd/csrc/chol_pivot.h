// Rank-revealing factor of the truncation Gram (round 6): diagonally pivoted Cholesky, stopped after KCAP rows.
//
// Where.  On the dense truncation route (engine_impl.h, walkers with 128 < live rows of M <= 256) the first compression
// B^T B = G = M M^T ran on chol_blocked_kernel: the full factorisation of an order-220 matrix (3.65 ms per launch of 8192 walkers, the
// single largest kernel of the real_rank leg) in the natural row order, keeping every direction above the f32 floor (r = 47..105 rows,
// the stragglers above 128 on a side stream) -- for a truncation that keeps chi = 32 of them.  The singular values of M fall by five
// decades over the first 32 and by another decade until 64, so a DIAGONALLY PIVOTED factorisation stopped after KCAP = 64 steps holds
// everything the truncation can see: measured on the truncation inputs of the tiled real state at C4 (scripts/proto_subspace.py, the
// part of the kept sigma_k u_k the subspace loses, relative to sigma_1): unpivoted + threshold (round 3-5) median 2.1e-7 / max 3.1e-6,
// pivoted and capped at 64 rows median 3.7e-8 / max 2.0e-7 -- the pivot order grades the rows of B, so their f32 rounding is relative
// to the size of the direction they carry.  What follows (second compression, Jacobi) never sees more than 64 rows.
//
// Shape of the work.  One 256-thread block per walker; thread t OWNS column t of G: its remaining diagonal and its column of the
// factor (KCAP doubles) stay in registers, so the right-looking update of a step is j fused multiply-adds per thread against the pivot
// column's factor entries, broadcast through LDS.  A step needs row p of
// G, p data dependent: pivots are taken NB = 4 at a time (the four largest remaining diagonals), their rows requested together -- one
// global round trip per round instead of one per pivot -- and processed in that order; inside a round the later candidates are
// corrected for the earlier ones (thread p' corrects its own pivot locally: the correction is the square of its own new entry).  A
// candidate whose remaining pivot fell below the threshold inside its round leaves a zero row (its slot is burnt; with the true top-NB
// selection that is rare: prototype, r = 57 median of 64 slots, 15 rounds).  G must hold BOTH triangles (the Gram kernel of the route
// mirrors its tiles, gram_i8.h `sym`): a row is one coalesced 2 KB read.
// Same pivot threshold, noise floor, output scaling and compaction as chol_blocked_kernel; rows come out in pivot order.
#pragma once
#include "linalg.h"

namespace pepsgpu {

template <typename T, int KCAP, int NB, int MINW>
__global__ __launch_bounds__(256, MINW) void chol_pivot_kernel(const double *__restrict__ Gg, long wG, int n, T *__restrict__ Rg, long wR,
                                                               int *__restrict__ mlive_out, int ld, const int *__restrict__ ndyn,
                                                               int ndyn_mul, const int *__restrict__ run_flag,
                                                               double *__restrict__ resid_out = nullptr, double thresh_scale = 1.0,
                                                               int *__restrict__ piv_out = nullptr) {
  // thresh_scale: multiplies the pivot threshold (0: pivots are taken down to the rounding noise of G -- the caller wants KCAP rows
  // SELECTED, not a factor: piv_out [b][KCAP] then lists the pivot columns in the order taken, -1 beyond the count, and the rows are
  // not compacted by their norm)
  // resid_out (optional): the largest diagonal of the Schur complement that is left when the kernel stops, relative to max diag(G) --
  // what the cap cut off (0 when every pivot above the threshold was taken); the float64 route prices its subspace with it
  static_assert(KCAP % NB == 0 && NB <= 4 && KCAP <= 64, "slots");
  const int b = blockIdx.x;
  if (run_flag && run_flag[b] >= 0) return;
  const int ldg = ld ? ld : n;
  if (ndyn) n = max(0, min(n, ndyn[b] * ndyn_mul));
  __shared__ __attribute__((aligned(32))) double s_rf[KCAP][NB];   // factor columns of the round's candidates (entry i of all NB side by side)
  __shared__ double s_wv[2][4];
  __shared__ int s_wi[2][4];
  __shared__ double s_piv[NB];
  __shared__ double s_x[NB][NB];
  __shared__ double s_part[4][KCAP];
  __shared__ short s_pos[KCAP];
  __shared__ short s_pcol[KCAP];
  __shared__ int s_cnt;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const double *G = Gg + (long)b * wG;
  T *Rout = Rg + (long)b * wR;

  double d = t < n ? G[(long)t * ldg + t] : -1.0;     // remaining diagonal; < 0: no such column / already a pivot
  {
    double md = fmax(d, 0.0);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) md = fmax(md, __shfl_xor(md, o, 64));
    if (lane == 0) s_wv[0][wave] = md;
    __syncthreads();
  }
  const double maxd = fmax(fmax(s_wv[0][0], s_wv[0][1]), fmax(s_wv[0][2], s_wv[0][3]));
  __syncthreads();
  const double eT = NOISE_C * eps_rt<T>();
  const double thresh = fmax((double)n * 2.220446049250313e-16, eT * eT) * maxd * thresh_scale;
  const double sc_out = maxd > 0.0 ? 1.0 / sqrt(maxd) : 1.0;

  double Lc[KCAP];
#pragma unroll
  for (int j = 0; j < KCAP; ++j) Lc[j] = 0.0;
  int nslots = 0;

  // The rounds are a RUN-TIME loop; inside, everything that touches Lc is unrolled over all KCAP entries in chunks of NB behind
  // block-uniform scalar branches (chunk c < rd: the factor so far; chunk c == rd: the rows of this round), so the register indices are
  // static and the body is one round long.  (The fully unrolled form -- every round its own code, 2000 FMAs + 1600 LDS reads in a
  // row -- came out of the compiler with 256 VGPRs and 300..1600 spilled dwords whatever KCAP.)  Thread-level decisions are selects;
  // what the thread of a candidate column has to publish is taken out of its lane with v_readlane by its whole wave.
  for (int rd = 0; rd < KCAP / NB; ++rd) {
    // ---- the NB largest remaining diagonals (exact: NB block-wide arg-max reductions; ties go to the lower column) ----
    int cand[NB];
    double dm = d;
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      double v = dm;
      int idx = t;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const double ov = __shfl_xor(v, o, 64);
        const int oi = __shfl_xor(idx, o, 64);
        const bool take = ov > v || (ov == v && oi < idx);
        v = take ? ov : v;
        idx = take ? oi : idx;
      }
      if (lane == 0) { s_wv[q & 1][wave] = v; s_wi[q & 1][wave] = idx; }
      __syncthreads();
      double bv = s_wv[q & 1][0];
      int bi = s_wi[q & 1][0];
#pragma unroll
      for (int w = 1; w < 4; ++w) {
        const double ov = s_wv[q & 1][w];
        const int oi = s_wi[q & 1][w];
        const bool take = ov > bv || (ov == bv && oi < bi);
        bv = take ? ov : bv;
        bi = take ? oi : bi;
      }
      bi = __builtin_amdgcn_readfirstlane(bi);
      cand[q] = __builtin_amdgcn_readfirstlane((bv > 0.0 && bv >= thresh) ? bi : -1);
      dm = (t == bi) ? -1.0 : dm;
    }
    if (cand[0] < 0) break;                      // nothing left above the threshold (block-uniform)
    nslots = (rd + 1) * NB;
    // ---- rows of G for the candidates: NB loads in flight, unconditional at clamped addresses ----
    double g[NB];
    const int tc = min(t, max(n - 1, 0));
#pragma unroll
    for (int q = 0; q < NB; ++q) g[q] = G[(long)max(cand[q], 0) * ldg + tc];
    // ---- the candidates' factor columns to LDS ----
#pragma unroll
    for (int q = 0; q < NB; ++q)
      if (cand[q] >= 0 && (cand[q] >> 6) == wave) {       // (wave-uniform)
        const int lc = cand[q] & 63;
#pragma unroll
        for (int c = 0; c < KCAP / NB; ++c)
          if (c < rd) {
#pragma unroll
            for (int k = 0; k < NB; ++k) s_rf[c * NB + k][q] = chb_readlane(Lc[c * NB + k], lc);
          }
      }
    __syncthreads();
    double v[NB];
#pragma unroll
    for (int q = 0; q < NB; ++q) v[q] = t < n ? g[q] : 0.0;
#pragma unroll
    for (int c = 0; c < KCAP / NB; ++c)
      if (c < rd) {
#pragma unroll
        for (int k = 0; k < NB; ++k) {
          const double lc = Lc[c * NB + k];
#pragma unroll
          for (int q = 0; q < NB; ++q) v[q] = fma(-s_rf[c * NB + k][q], lc, v[q]);
        }
      }
    if (cand[0] >= 0 && (cand[0] >> 6) == wave) s_piv[0] = chb_readlane(v[0], cand[0] & 63);
    __syncthreads();
    // ---- the steps of the round ----
    double xr[NB];
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      const double piv = cand[q] >= 0 ? s_piv[q] : 0.0;
      const bool live = piv > 0.0 && piv >= thresh;       // (block-uniform)
      const double pv = live ? piv : 1.0;
      const double sc = jr_rsq64(pv);
      // the pivot column itself gets sqrt(piv) = piv / sqrt(piv); columns that are pivots already (or do not exist) get zero
      const double num = (t == cand[q]) ? piv : ((d >= 0.0) ? v[q] : 0.0);
      const double x = live ? num * sc : 0.0;
      xr[q] = x;
      if (t == 0) s_pcol[rd * NB + q] = live ? (short)cand[q] : (short)-1;
      d = !live ? d : ((t == cand[q]) ? -1.0 : (d >= 0.0 ? fma(-x, x, d) : d));
      if (q + 1 < NB) {
        // later candidates: their entry of this row (for everybody's correction) and, for the next one, its corrected pivot
#pragma unroll
        for (int q2 = q + 1; q2 < NB; ++q2)
          if (cand[q2] >= 0 && (cand[q2] >> 6) == wave) {       // (wave-uniform)
            const double xs = chb_readlane(x, cand[q2] & 63);
            s_x[q][q2] = xs;
            if (q2 == q + 1) s_piv[q2] = fma(-xs, xs, chb_readlane(v[q2], cand[q2] & 63));
          }
        __syncthreads();
#pragma unroll
        for (int q2 = q + 1; q2 < NB; ++q2) v[q2] = cand[q2] >= 0 ? fma(-s_x[q][q2], x, v[q2]) : v[q2];
      }
    }
#pragma unroll
    for (int c = 0; c < KCAP / NB; ++c)
      if (c == rd) {
#pragma unroll
        for (int k = 0; k < NB; ++k) Lc[c * NB + k] = xr[k];
      }
    __syncthreads();      // s_rf / s_piv / s_x are rewritten by the next round
  }

  if (resid_out) {
    double md = fmax(d, 0.0);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) md = fmax(md, __shfl_xor(md, o, 64));
    if (lane == 0) s_wv[0][wave] = md;
    __syncthreads();
    if (t == 0) {
      const double r = fmax(fmax(s_wv[0][0], s_wv[0][1]), fmax(s_wv[0][2], s_wv[0][3]));
      resid_out[b] = (maxd > 0.0 && r >= thresh) ? r / maxd : 0.0;
    }
    __syncthreads();
  }
  // ---- rank compaction: rows with norm below NOISE_C * eps_T * |R|_F are dropped (as chol_blocked_kernel) ----
#pragma unroll
  for (int j = 0; j < KCAP; ++j) {
    if (j < nslots) {
      const double a = wave_sum(Lc[j] * Lc[j]);
      if (lane == 0) s_part[wave][j] = a;
    }
  }
  __syncthreads();
  if (t == 0) {
    double fro = 0.0;
    for (int j = 0; j < nslots; ++j) fro += s_part[0][j] + s_part[1][j] + s_part[2][j] + s_part[3][j];
    const double nfloor = eT * eT * fro;
    int cnt = 0;
    for (int j = 0; j < nslots; ++j) {
      const double nj = s_part[0][j] + s_part[1][j] + s_part[2][j] + s_part[3][j];
      const bool keep = piv_out ? (s_pcol[j] >= 0) : (nj > nfloor);
      s_pos[j] = keep ? (short)cnt++ : (short)-1;
      if (piv_out && keep) piv_out[(long)b * KCAP + cnt - 1] = s_pcol[j];
    }
    if (piv_out) for (int j = cnt; j < KCAP; ++j) piv_out[(long)b * KCAP + j] = -1;
    s_cnt = cnt;
    if (mlive_out) mlive_out[b] = cnt;
  }
  __syncthreads();
  // the consumers read whole rows of the ld-wide buffer: columns n .. ld come out as zeros (their Lc is zero)
  if (t < ldg) {
#pragma unroll
    for (int j = 0; j < KCAP; ++j) {
      if (j < nslots) {
        const int pos = s_pos[j];
        if (pos >= 0) Rout[(long)pos * ldg + t] = T(Lc[j] * sc_out);
      }
    }
  }
}

// KCAP rows at most; G holds both triangles.  Orders up to 256 (one thread per column).
inline int chol_pivot_slots(int kcap) { return kcap <= 48 ? 48 : kcap <= 56 ? 56 : 64; }

template <typename T>
inline void launch_chol_pivot(hipStream_t s, int nbatch, const double *G, long wG, int n, T *R, long wR, int *mlive_out, int ld,
                              const int *ndyn, int ndyn_mul, const int *run_flag, int kcap, double *resid_out = nullptr,
                              double thresh_scale = 1.0, int *piv_out = nullptr) {
  // (piv_out: [nbatch][KCAP of the instantiation taken = 48 / 56 / 64 for kcap <= 48 / <= 56 / else])
  PG_REQUIRE(n <= 256 && (ld == 0 || ld <= 256), 1, "pivoted Cholesky: order above 256");
  if (kcap <= 48)
    hipLaunchKernelGGL((chol_pivot_kernel<T, 48, 4, 3>), dim3(nbatch), dim3(256), 0, s, G, wG, n, R, wR, mlive_out, ld, ndyn, ndyn_mul, run_flag, resid_out, thresh_scale, piv_out);
  else if (kcap <= 56)
    hipLaunchKernelGGL((chol_pivot_kernel<T, 56, 4, 3>), dim3(nbatch), dim3(256), 0, s, G, wG, n, R, wR, mlive_out, ld, ndyn, ndyn_mul, run_flag, resid_out, thresh_scale, piv_out);
  else
    hipLaunchKernelGGL((chol_pivot_kernel<T, 64, 4, 2>), dim3(nbatch), dim3(256), 0, s, G, wG, n, R, wR, mlive_out, ld, ndyn, ndyn_mul, run_flag, resid_out, thresh_scale, piv_out);
  PG_CHECK_HIP(hipGetLastError());
}

}  // namespace pepsgpu
