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
#include "tgemm.h"   // tg_flop_counter / tg_byte_counter of the current profiling bracket

namespace pepsgpu {

typedef double tm_f64x4 __attribute__((ext_vector_type(4)));
constexpr int TM_KC = 32;    // columns of M staged per chunk
constexpr int TM_LDM = TM_KC + 2;   // row pitch of the chunk: 16 rows x 2 k-lanes of an operand read hit 32 different banks

// One staged chunk (TM_KC columns of the k index) of G += X X^T for the NQ tiles of this wave: per k-step of four, the 2 NQ
// operand reads are issued together, then the NQ MFMAs -- branch free (rows of the chunk beyond the matrix are zero filled
// once), so the LDS latency is paid once per k-step and not once per MFMA.  offa / offb = LDS offset of the tile's row.
constexpr int TM_PF = 1;                                 // chunks of the Gram operand in flight (registers; 3 measured: no gain, spills in the rows form)
constexpr int TM_TPW = 9;                                // 36 tiles (order 128) / 4 waves
template <int NQ>
__device__ __forceinline__ void tm_gram_chunk(tm_f64x4 (&acc)[TM_TPW], const float *__restrict__ sX, const int (&offa)[TM_TPW],
                                              const int (&offb)[TM_TPW], const int k4) {
#pragma unroll 2
  for (int s = 0; s < TM_KC / 4; ++s) {
    double a[NQ], b[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      a[q] = (double)sX[offa[q] + 4 * s + k4];
      b[q] = (double)sX[offb[q] + 4 * s + k4];
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], b[q], acc[q], 0, 0, 0);
  }
}
__device__ __forceinline__ void tm_gram_chunk_n(const int nq, tm_f64x4 (&acc)[TM_TPW], const float *__restrict__ sX,
                                                const int (&offa)[TM_TPW], const int (&offb)[TM_TPW], const int k4) {
  switch (nq) {                                          // wave-uniform
    case 1: tm_gram_chunk<1>(acc, sX, offa, offb, k4); break;
    case 2: tm_gram_chunk<2>(acc, sX, offa, offb, k4); break;
    case 3: tm_gram_chunk<3>(acc, sX, offa, offb, k4); break;
    case 4: tm_gram_chunk<4>(acc, sX, offa, offb, k4); break;
    case 5: tm_gram_chunk<5>(acc, sX, offa, offb, k4); break;
    case 6: tm_gram_chunk<6>(acc, sX, offa, offb, k4); break;
    case 7: tm_gram_chunk<7>(acc, sX, offa, offb, k4); break;
    case 8: tm_gram_chunk<8>(acc, sX, offa, offb, k4); break;
    case 9: tm_gram_chunk<9>(acc, sX, offa, offb, k4); break;
    default: break;
  }
}


// ---------------------------------------------------------------------------------------------
// Upper Cholesky of a symmetric n x n matrix whose upper triangle lives in LDS (element (i, c >= i) at sG[at(i, c)]), in place,
// blocked in panels of 16 rows (round 3; before: one barrier, and one LDS read-modify-write round trip of the trailing rows,
// per PIVOT -- ~1 us per pivot with two blocks on a CU, 128 us of a 128-column factor that holds 1.3 Mflop):
//   A. the 16 x 16 diagonal block is factored by wave 0 in registers: lane c holds column c, the pivot row travels by
//      v_readlane (no barrier inside the block); a pivot at or below `thresh`, or beyond the `kcap` live rows the data can
//      have, drops its row (zeros);
//   B. the rest of the block row is a forward substitution, one thread per column, the block's factor read as LDS broadcasts;
//   C. the trailing triangle takes the whole panel at once, G[i][r] -= sum_p U[p][i] U[p][r], as 16 x 16 tiles of
//      v_mfma_f64_16x16x4_f64 (four per tile) dealt round-robin to the four waves, operands straight from the LDS-resident rows.
// Three barriers per panel.  Rows come out SCALED (row j of the factor itself, R^T R = G); sList[0 .. return value) = the live
// rows in order.  Ends with a barrier.  256 threads.
struct LdsAtPitch {
  int ld;
  __device__ __forceinline__ int operator()(int i, int c) const { return i * ld + c; }
};
template <typename At>
__device__ __forceinline__ int lds_chol_blocked(double *__restrict__ sG, const At at, const int n, const double thresh, const int kcap,
                                                short *__restrict__ sList, unsigned long long *__restrict__ stats = nullptr,
                                                const int serial_wave = 0) {
  __shared__ double sD[16][17], sDinv[16];
  unsigned long long tA = 0, tB = 0, tC = 0, t0 = 0;
  __shared__ unsigned s_livemask;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, k4 = lane >> 4;
  int nl = 0;
  // the serial part runs in ONE wave: not the same wave (= SIMD) in the blocks that share a CU
  const int aw = serial_wave;
  for (int jb = 0; jb < n; jb += 16) {
    const int nb = min(16, n - jb);
    if (stats) t0 = wall_clock64();
    if (wave == aw) {
      double d[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) d[c] = (lane < nb && c <= lane) ? sG[at(jb + c, jb + lane)] : 0.0;
      unsigned livemask = 0;
      int cnt = nl;
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const double piv = chb_readlane(d[c], c);
        const bool live = c < nb && piv > thresh && cnt < kcap;   // wave-uniform
        // branch-free (round 4, as chol_blocked_kernel): no copy of d[] at the join of every pivot step
        const double pv = live ? piv : 1.0;
        const double sc = jr_rsq64(pv);
        d[c] = live ? d[c] * sc : 0.0;
#pragma unroll
        for (int c2 = c + 1; c2 < 16; ++c2) {
          const double f = chb_readlane(d[c], c2);
          d[c2] -= f * d[c];
        }
        livemask |= live ? 1u << c : 0u;
        cnt += live ? 1 : 0;
      }
      if (lane < 16) {
        double dg = 0.0;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
          sD[c][lane] = lane >= c ? d[c] : 0.0;
          if (lane < nb && c <= lane) sG[at(jb + c, jb + lane)] = d[c];
          dg = lane == c ? d[c] : dg;
        }
        const double iv = jr_rcp64(dg);
        const bool mine = (livemask >> lane) & 1u;
        sDinv[lane] = mine ? iv : 0.0;
        if (mine) sList[nl + __popc(livemask & ((1u << lane) - 1u))] = (short)(jb + lane);
      }
      if (lane == 0) s_livemask = livemask;
    }
    __syncthreads();
    if (stats) { const unsigned long long t1 = wall_clock64(); tA += t1 - t0; t0 = t1; }
    nl += __popc(s_livemask);
    if (jb + 16 >= n) break;                             // (block-uniform) the last panel has no columns to its right
    {
      // columns to the right of the block, a quarter to each wave (a latency chain per thread: spread over the SIMDs)
      const int Wd = n - jb - 16, chunk = (Wd + 3) >> 2;
      const int r = jb + 16 + wave * chunk + lane;
      if (lane < chunk && r < n) {
        double x[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) {
          double v = sG[at(jb + c, r)];
#pragma unroll
          for (int c1 = 0; c1 < c; ++c1) v -= sD[c1][c] * x[c1];
          x[c] = v * sDinv[c];                           // dropped row: sDinv = 0
          sG[at(jb + c, r)] = x[c];
          __asm__ volatile("" ::: "memory");             // keep the factor reads of the later rows from being hoisted (136 doubles)
        }
      }
    }
    __syncthreads();
    if (stats) { const unsigned long long t1 = wall_clock64(); tB += t1 - t0; t0 = t1; }
    {
      const int c0 = jb + 16;
      const int nt = (n - c0 + 15) >> 4, ntl = nt * (nt + 1) / 2;
      for (int t = wave; t < ntl; t += 4) {
        int tt = t, ti = 0;
        while (tt >= nt - ti) { tt -= nt - ti; ++ti; }
        const int tj = ti + tt;
        const int ca = min(c0 + 16 * ti + i16, n - 1), cb = min(c0 + 16 * tj + i16, n - 1);   // clamped: lands in entries not written
        chb_f64x4 acc;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = 0.0;
        double a[4], bq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { a[q] = sG[at(jb + 4 * q + k4, ca)]; bq[q] = sG[at(jb + 4 * q + k4, cb)]; }
#pragma unroll
        for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], bq[q], acc, 0, 0, 0);
        const int j = c0 + 16 * tj + i16;
#pragma unroll
        for (int r = 0; r < 4; ++r) {                     // acc[r] = C[k4 + 4 r][i16] of the tile
          const int i = c0 + 16 * ti + k4 + 4 * r;
          if (j >= i && j < n) sG[at(i, j)] -= acc[r];
        }
      }
    }
    __syncthreads();
    if (stats) { const unsigned long long t1 = wall_clock64(); tC += t1 - t0; t0 = t1; }
  }
  __syncthreads();
  if (stats && tid == 0) { atomicAdd(stats + 7, tA); atomicAdd(stats + 8, tB); atomicAdd(stats + 9, tC); }
  return nl;
}


// Tail of the LDS-resident factorisations (scaled rows, lds_chol_blocked): squared norms of the nl live rows, the noise floor
// NOISE_C eps_T |R|_F, output positions of the rows above it (sPos[q], -1 = dropped) by ballot in wave 0.  Returns the count
// (every thread); ends with a barrier.
template <typename At>
__device__ __forceinline__ int lds_chol_compact(const double *__restrict__ sG, const At at, const int n, const int nl,
                                                const short *__restrict__ sList, double *__restrict__ sNrm, short *__restrict__ sPos,
                                                const double eT) {
  __shared__ int s_cnt_out;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int q = wave; q < nl; q += 4) {
    const int j = sList[q];
    double a = 0.0;
    for (int r = j + lane; r < n; r += 64) { const double x = sG[at(j, r)]; a += x * x; }
    a = wave_sum(a);
    if (lane == 0) sNrm[q] = a;
  }
  __syncthreads();
  if (wave == 0) {
    double f = 0.0;
    for (int q = lane; q < nl; q += 64) f += sNrm[q];
    f = wave_sum(f);
    const double nfloor = eT * eT * f;
    int cnt = 0;
    for (int base = 0; base < nl; base += 64) {
      const int q = base + lane;
      const bool keep = q < nl && sNrm[q] > nfloor;
      const unsigned long long mask = __ballot(keep);
      if (q < nl) sPos[q] = keep ? (short)(cnt + __popcll(mask & ((1ull << lane) - 1ull))) : (short)-1;
      cnt += __popcll(mask);
    }
    if (lane == 0) s_cnt_out = cnt;
  }
  __syncthreads();
  return s_cnt_out;
}

// the same chunk step for an accumulator array of TPW tiles (the kernels below come in size classes: fewer tiles per wave,
// fewer registers, more blocks per CU)
template <int NQ, int TPW>
__device__ __forceinline__ void tm_gram_chunk_t(tm_f64x4 (&acc)[TPW], const float *__restrict__ sX, const int (&offa)[TPW],
                                                const int (&offb)[TPW], const int k4) {
#pragma unroll 2
  for (int s = 0; s < TM_KC / 4; ++s) {
    double a[NQ], b[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      a[q] = (double)sX[offa[q] + 4 * s + k4];
      b[q] = (double)sX[offb[q] + 4 * s + k4];
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], b[q], acc[q], 0, 0, 0);
  }
}
template <int TPW>
__device__ __forceinline__ void tm_gram_chunk_tn(const int nq, tm_f64x4 (&acc)[TPW], const float *__restrict__ sX,
                                                 const int (&offa)[TPW], const int (&offb)[TPW], const int k4) {
#define TM_CASE(N) if constexpr (TPW >= N) { if (nq == N) { tm_gram_chunk_t<N, TPW>(acc, sX, offa, offb, k4); return; } }
  TM_CASE(1) TM_CASE(2) TM_CASE(3) TM_CASE(4) TM_CASE(5) TM_CASE(6) TM_CASE(7) TM_CASE(8) TM_CASE(9)
#undef TM_CASE
}

// packed upper triangle of order nc in LDS: element (i, c >= i) at i (nc - 1) - i (i - 1) / 2 + c  (nc (nc + 1) / 2 doubles)
struct LdsAtPacked {
  int nc;
  __device__ __forceinline__ int operator()(int i, int c) const { return i * (nc - 1) - ((i * (i - 1)) >> 1) + c; }
};


// PEPSGPU_CG_STATS=1 (diagnostics): per-phase time of colgram_dense_kernel, summed over its blocks in a device buffer of 16 counters
// ([0] blocks, [1] sum K, [2] sum live columns, [3..5] 10-ns ticks of the Gram / Cholesky / output phase, [6] sum live rows out)
inline unsigned long long *cg_stats_dev() {
  static unsigned long long *p = []() -> unsigned long long * {
    if (!getenv("PEPSGPU_CG_STATS")) return nullptr;
    unsigned long long *q = nullptr;
    if (hipMalloc(&q, 16 * sizeof(unsigned long long)) != hipSuccess) return nullptr;
    (void)hipMemset(q, 0, 16 * sizeof(unsigned long long));
    return q;
  }();
  return p;
}

inline size_t mid_gram_chol_smem_bytes(int cap) {
  const size_t capr = ((size_t)cap + 15) & ~(size_t)15;      // the staging buffer covers whole 16-row tiles
  const size_t tri = sizeof(double) * (size_t)cap * (cap + 1) / 2, stg = (sizeof(float) * capr * TM_LDM + 7) & ~(size_t)7;
  return std::max(tri, stg) + sizeof(double) * (size_t)cap + sizeof(short) * 2 * (size_t)cap + 64;
}

// run_flag[b] < 0: the entry is on the route; n = nrows[b] in (lo, cap] is taken by this launch (another launch with a
// different cap takes the rest).  TPW = 16 x 16 tiles of G per wave this size class needs, MINB = blocks per CU it is built for.
// LDS (round 3): G / the factor as a PACKED triangle laid over the dead staging buffer -- 27 KB for cap = 80 (four blocks per
// CU; the square form took 64 KB: two), 67 KB for cap = 128 (two; was 151 KB: ONE).
template <typename T, int TPW, int MINB>
__global__ __launch_bounds__(256, MINB) void mid_gram_chol_kernel(const T *__restrict__ Mg, long wM, int uk, const int *__restrict__ nrows,
                                                               const int *__restrict__ run_flag, int lo, int cap,
                                                               T *__restrict__ Bg, long wB, int ld, int *__restrict__ mB,
                                                               unsigned long long *__restrict__ flopc,
                                                               unsigned long long *__restrict__ bytec, int flop_stride) {
  static_assert(sizeof(T) == 4, "the mid route is f32 only (the f64 mode keeps the direct Jacobi)");
  const int b = blockIdx.x;
  if (run_flag && run_flag[b] >= 0) return;
  const int n = nrows[b];
  if (n <= lo || n > cap) return;
  if (flopc && threadIdx.x == 0 && b % flop_stride == 0) {   // MFMA flops issued: 16 x 16 tiles on or above the diagonal
    const unsigned long long nt_ = (unsigned long long)((n + 15) >> 4);
    atomicAdd(flopc, (unsigned long long)flop_stride * (nt_ * (nt_ + 1) / 2) * 512ull * (unsigned long long)uk);
    if (bytec) atomicAdd(bytec, (unsigned long long)flop_stride * 4ull * ((unsigned long long)n * uk + (unsigned long long)n * n / 2));
  }
  extern __shared__ double tm_smem[];
  const int capr = (cap + 15) & ~15;
  const size_t tri = (size_t)cap * (cap + 1) / 2, stg = ((size_t)capr * TM_LDM + 1) / 2;
  double *sG = tm_smem;                                 // packed upper triangle of G, then of the factor (after the Gram phase)
  float *sM = reinterpret_cast<float *>(tm_smem);       // [capr][TM_LDM] chunk of M (Gram phase only)
  double *sNrm = tm_smem + (tri > stg ? tri : stg);     // [cap] squared norm of a factor row
  short *sList = reinterpret_cast<short *>(sNrm + cap); // [cap] live rows in order
  short *sPos = sList + cap;                            // [cap] output position, -1 = dropped
  const LdsAtPacked at{cap};
  __shared__ double s_red[4], s_maxd;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const T *M = Mg + (long)b * wM;
  T *B = Bg + (long)b * wB;

  // ---- G = M M^T: 16 x 16 tiles on or above the diagonal, dealt round-robin to the four waves ----
  const int nt = (n + 15) >> 4, ntiles = nt * (nt + 1) / 2;
  const int r16 = lane & 15, k4 = lane >> 4;
  int ti[TPW], tj[TPW], offa[TPW], offb[TPW];
  int nq = 0;
#pragma unroll
  for (int q = 0; q < TPW; ++q) {
    int t = wave + 4 * q, i = 0;
    if (t < ntiles) { while (t >= nt - i) { t -= nt - i; ++i; } ti[q] = i; tj[q] = i + t; nq = q + 1; }
    else { ti[q] = 0; tj[q] = 0; }
    offa[q] = (16 * ti[q] + r16) * TM_LDM;
    offb[q] = (16 * tj[q] + r16) * TM_LDM;
  }
  tm_f64x4 acc[TPW];
#pragma unroll
  for (int q = 0; q < TPW; ++q)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[q][r] = 0.0;
  // rows of the staging buffer beyond the matrix (tile padding) are read by the MFMA operands: zero, once
  for (int e = tid + n * TM_LDM; e < 16 * nt * TM_LDM; e += 256) sM[e] = 0.f;
  // 32 columns x up to 128 rows = 16 elements per thread: unconditional loads (clamped address); the chunk after the one being
  // multiplied is in flight during its MFMAs (registers v), stored once the waves have left the staging buffer
  float v[16];
  auto issue = [&](const int kc) {
    const int kw = min(TM_KC, uk - kc);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int e = tid + 256 * i, r = min(e >> 5, n - 1), c = min(e & 31, kw - 1);
      v[i] = (float)M[(long)r * uk + kc + c];
    }
  };
  if (uk > 0) issue(0);
  for (int kc = 0; kc < uk; kc += TM_KC) {
    const int kw = min(TM_KC, uk - kc);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int e = tid + 256 * i, r = e >> 5, c = e & 31;
      if (r < n) sM[r * TM_LDM + c] = c < kw ? v[i] : 0.f;
    }
    __syncthreads();
    if (kc + TM_KC < uk) issue(kc + TM_KC);
    tm_gram_chunk_tn<TPW>(nq, acc, sM, offa, offb, k4);
  }
  __syncthreads();                                       // the staging buffer is dead: G takes its place
#pragma unroll
  for (int q = 0; q < TPW; ++q) {
    if (q >= nq) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {                        // acc[r] = C[(lane >> 4) + 4 r][lane & 15]
      const int i = 16 * ti[q] + k4 + 4 * r, j = 16 * tj[q] + r16;
      if (i <= j && j < n) sG[at(i, j)] = acc[q][r];
    }
  }
  __syncthreads();

  // ---- upper Cholesky in LDS, panels of 16 rows (lds_chol_blocked), semi-definite safe ----
  double md = 0.0;
  for (int i = tid; i < n; i += 256) md = fmax(md, sG[at(i, i)]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) md = fmax(md, __shfl_xor(md, o, 64));
  if (lane == 0) s_red[wave] = md;
  __syncthreads();
  if (tid == 0) s_maxd = fmax(fmax(s_red[0], s_red[1]), fmax(s_red[2], s_red[3]));
  __syncthreads();
  const double maxd = s_maxd;
  const double eT = NOISE_C * (double)Eps<T>::v;
  const double thresh = fmax((double)n * 2.220446049250313e-16, eT * eT) * maxd;
  const int nl = lds_chol_blocked(sG, at, n, thresh, n, sList, nullptr, (int)(((unsigned)b >> 3) + ((unsigned)b >> 8)) & 3);
  // ---- rank compaction (rows with norm below NOISE_C eps_T |B|_F are dropped), as chol_upper_kernel ----
  const int cnt = lds_chol_compact(sG, at, n, nl, sList, sNrm, sPos, eT);
  if (tid == 0) mB[b] = cnt;
  const double sc = maxd > 0.0 ? 1.0 / sqrt(maxd) : 1.0;
  for (int q = wave; q < nl; q += 4) {
    const int pos = sPos[q];
    if (pos < 0) continue;
    const int j = sList[q];
    // the Jacobi reads whole rows of the ld-wide buffer: columns outside [j, n) are written as zeros
    for (int r = lane; r < ld; r += 64) B[(long)pos * ld + r] = (r >= j && r < n) ? T(sG[at(j, min(r, n - 1))] * sc) : T(0);
  }
}

template <typename T>
inline void launch_mid_gram_chol(hipStream_t s, int nbatch, const T *M, long wM, int uk, const int *nrows, const int *run_flag,
                                 int GS, T *B, long wB, int *mB) {
  // two size classes: most walkers of a dense state have 50-80 live rows
  const int caps[2] = {80, 128};
  int lo = 0;
  for (int c = 0; c < 2; ++c) {
    const int cap = std::min(caps[c], GS);
    if (cap <= lo) break;
    const size_t smem = mid_gram_chol_smem_bytes(cap);
    constexpr int minb = 4;
    if (cap <= 80 && minb >= 4) {      // <= 5 tile rows: 15 tiles, four per wave
      allow_dynamic_lds(reinterpret_cast<const void *>(&mid_gram_chol_kernel<T, 4, 4>), smem);
      hipLaunchKernelGGL((mid_gram_chol_kernel<T, 4, 4>), dim3(nbatch), dim3(256), smem, s, M, wM, uk, nrows, run_flag, lo, cap, B, wB, GS,
                         mB, tg_flop_counter, tg_byte_counter, nbatch >= 256 ? 64 : 1);
    } else if (cap <= 80) {
      allow_dynamic_lds(reinterpret_cast<const void *>(&mid_gram_chol_kernel<T, 4, 3>), smem);
      hipLaunchKernelGGL((mid_gram_chol_kernel<T, 4, 3>), dim3(nbatch), dim3(256), smem, s, M, wM, uk, nrows, run_flag, lo, cap, B, wB, GS,
                         mB, tg_flop_counter, tg_byte_counter, nbatch >= 256 ? 64 : 1);
    } else {
      allow_dynamic_lds(reinterpret_cast<const void *>(&mid_gram_chol_kernel<T, 9, 2>), smem);
      hipLaunchKernelGGL((mid_gram_chol_kernel<T, 9, 2>), dim3(nbatch), dim3(256), smem, s, M, wM, uk, nrows, run_flag, lo, cap, B, wB, GS,
                         mB, tg_flop_counter, tg_byte_counter, nbatch >= 256 ? 64 : 1);
    }
    PG_CHECK_HIP(hipGetLastError());
    lo = cap;
  }
}

}  // namespace pepsgpu

namespace pepsgpu {

// ---------------------------------------------------------------------------------------------
// Forward factor of the absorption in one kernel per walker: R^T R = P^T P for the LIVE columns of P (K live rows x n
// columns; columns = (outer, inner), inner index live below inner_live[b]; at most 128 live columns).
//
//   Gram:      G = P^T P by v_mfma_f64_16x16x4_f64 from row chunks of P staged (transposed) in LDS; the 16 x 16 tiles on or
//              above the diagonal stay in the ACCUMULATOR REGISTERS of the four waves -- G is never written anywhere.
//   Cholesky:  right-looking, low-rank (the rule of chol_lowrank_kernel): the next pivot is the first live column whose
//              remaining diagonal exceeds the noise of the T-typed data; row f of G is lifted out of the accumulators into
//              an LDS row buffer, every thread (= column) subtracts the finished factor rows (kept in LDS, f64) and adds
//              its entry of the new row.  Work and LDS follow the numerical RANK (rcap x n doubles), not n x n.
//
// Same output contract as gram_chol_lowrank_kernel (linalg.h): live rows compacted and scaled by 1 / sqrt(max diag),
// written as type T at the original column positions, mlive_out[b] = live rows; an entry it cannot take (more than 128
// live columns, rank above rcap) gets mlive_out[b] = decline_code and is left to the kernels launched after it.
// It replaces, for the walkers it takes, the thread-per-column Gram-free kernel (f64 VALU dot products of K rows per
// step: 26 % of the headline step) on the low-rank side and the streaming Gram + blocked Cholesky pair (global-memory
// round trip of the 128 x 128 Gram) on the dense side.
constexpr int CG_NC = 128;           // live columns at most
// The finished factor rows are kept in LDS as a packed upper triangle: the pivots are taken in increasing (packed) column
// order, so row j of the factor is zero before column f_j >= j and only its columns c >= j are stored, at cg_row(j) + c.
// 88 rows x 128 columns: 59 KB instead of the 91 KB of the rectangle -- two blocks per CU instead of one (the kernel is
// a chain of barriers and dependent LDS round trips per pivot: a second block is what fills the CU).
__host__ __device__ inline int cg_row(int j) { return j * (CG_NC - 1) - (j * (j - 1)) / 2; }
struct CgAtPacked {
  __device__ __forceinline__ int operator()(int i, int c) const { return cg_row(i) + c; }
};

inline size_t colgram_chol_smem_bytes(int rcap) {
  return sizeof(double) * ((size_t)rcap * CG_NC - (size_t)rcap * (rcap - 1) / 2 + CG_NC) + sizeof(float) * (size_t)CG_NC * TM_LDM + 64;
}

// The same factor for DENSE walkers (rank of the order of the column count): after the Gram phase the accumulators are laid
// down in LDS as the packed upper triangle of G (over the staging buffer, which is dead by then) and factored there in panels of
// 16 rows (lds_chol_blocked: diagonal block in one wave's registers, forward substitution, MFMA trailing update).  No rank cap.
// History: the low-rank pivot loop of colgram_chol_kernel was 73 % of the kernel on the dense sites of a full-rank state; round 2
// factored the triangle right-looking with a barrier per pivot, then in VALU panels; round 3: lds_chol_blocked, and two SIZE
// CLASSES -- NC = 96 live columns (21 tiles, six per wave, 38 KB of LDS: three blocks per CU) and NC = 128 (two blocks per CU):
// measured per block on the full-rank state (445 rows x 78 columns): Gram 41 us (at the f64 MFMA rate of two resident blocks),
// Cholesky 40 us (a latency chain in one wave), output 10 us -- the chain is what more blocks per CU overlap.
// A launch takes the walkers with lo < live columns <= NC; the last class declines (decline_code) what is wider than NC.
template <typename T, int NC, int TPW, int MINB>
__global__ __launch_bounds__(256, MINB) void colgram_dense_kernel(const T *__restrict__ Pg, long wP, int n, const int *__restrict__ kdyn,
                                                               int kdyn_mul, int kmax, T *__restrict__ Rg, long wR,
                                                               int *__restrict__ mlive_out, int inner,
                                                               const int *__restrict__ inner_live, int lo, int last, int decline_code,
                                                               unsigned long long *__restrict__ flopc,
                                                               unsigned long long *__restrict__ bytec, int flop_stride,
                                                               unsigned long long *__restrict__ stats) {
  static_assert(sizeof(T) == 4, "f32 element type");
  static_assert(NC % 16 == 0 && NC <= 128 && 4 * TPW >= (NC / 16) * (NC / 16 + 1) / 2, "tiles of the class over four waves");
  const int b = blockIdx.x;
  const unsigned long long t_0 = stats ? wall_clock64() : 0ull;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = kdyn ? max(0, min(kmax, kdyn[b] * kdyn_mul)) : kmax;
  const int ilive = inner_live ? min(inner, inner_live[b]) : inner;
  const int ncols = (n / inner) * ilive;
  if (ncols <= lo) return;                                 // an earlier launch took it
  if (ncols > NC) {
    if (last && tid == 0) mlive_out[b] = decline_code;
    return;
  }
  extern __shared__ double cg_smem[];
  constexpr size_t TRI = (size_t)NC * (NC + 1) / 2, STG = ((size_t)NC * TM_LDM + 1) / 2;
  float *sP = reinterpret_cast<float *>(cg_smem);          // [NC][TM_LDM] chunk of P, transposed (Gram phase only)
  double *sG = cg_smem;                                    // packed upper triangle of G / of the factor
  double *sNrm = sG + (TRI > STG ? TRI : STG);             // [NC] squared norm of a factor row
  short *sList = reinterpret_cast<short *>(sNrm + NC);     // [NC] live rows in order
  short *sPos = sList + NC;                                // [NC] output position, -1 = dropped
  const LdsAtPacked at{NC};
  __shared__ double s_red[4], s_maxd;
  const T *P = Pg + (long)b * wP;
  T *Rout = Rg + (long)b * wR;

  // ---- G = P^T P over the packed columns, upper tiles in registers ----
  const int nt = (ncols + 15) >> 4, ntiles = nt * (nt + 1) / 2;
  if (flopc && tid == 0 && b % flop_stride == 0) {     // MFMA flops issued (16 x 16 x 2 per tile and row of P) and compulsory bytes
    atomicAdd(flopc, (unsigned long long)flop_stride * (unsigned long long)ntiles * 512ull * (unsigned long long)K);
    if (bytec) atomicAdd(bytec, (unsigned long long)flop_stride * 4ull * (unsigned long long)K * ncols);
  }
  const int r16 = lane & 15, k4 = lane >> 4;
  int ti[TPW], tj[TPW], offa[TPW], offb[TPW];
  int nq = 0;
#pragma unroll
  for (int q = 0; q < TPW; ++q) {
    int t = wave + 4 * q, i = 0;
    if (t < ntiles) { while (t >= nt - i) { t -= nt - i; ++i; } ti[q] = i; tj[q] = i + t; nq = q + 1; }
    else { ti[q] = -1; tj[q] = -1; }
    offa[q] = (16 * max(ti[q], 0) + r16) * TM_LDM;
    offb[q] = (16 * max(tj[q], 0) + r16) * TM_LDM;
  }
  tm_f64x4 acc[TPW];
#pragma unroll
  for (int q = 0; q < TPW; ++q)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[q][r] = 0.0;
  const int st_c = tid & 127;                                            // staging: packed column of this thread
  const int st_r = st_c < ncols ? (st_c / ilive) * inner + (st_c % ilive) : -1;
  // columns of the staging buffer beyond the live ones (tile padding) are read by the MFMA operands: zero, once
  for (int e = tid + ncols * TM_LDM; e < 16 * nt * TM_LDM; e += 256) sP[e] = 0.f;
  // unconditional loads (clamped row), 16 per thread and chunk; the chunk after the one being multiplied is in flight during
  // its MFMAs (three chunks in flight measured: 40.9 us against 39.1 us -- the loads are not what the phase waits for: the MFMA
  // loop runs at the f64 matrix rate two resident blocks share, 27 of the 41 us; staging and barriers are the rest)
  float v[TM_KC / 2];
  auto issue = [&](const int k0) {
    const int kw = min(TM_KC, K - k0);
#pragma unroll
    for (int i = 0; i < TM_KC / 2; ++i) {
      const int k = min((tid >> 7) + 2 * i, kw - 1);
      v[i] = (float)P[(long)(k0 + k) * n + st_r];
    }
  };
  if (K > 0 && st_r >= 0) issue(0);
  for (int k0 = 0; k0 < K; k0 += TM_KC) {
    const int kw = min(TM_KC, K - k0);
    __syncthreads();
    if (st_r >= 0) {
#pragma unroll
      for (int i = 0; i < TM_KC / 2; ++i) {
        const int k = (tid >> 7) + 2 * i;
        sP[st_c * TM_LDM + k] = k < kw ? v[i] : 0.f;
      }
    }
    __syncthreads();
    if (k0 + TM_KC < K && st_r >= 0) issue(k0 + TM_KC);
    tm_gram_chunk_tn<TPW>(nq, acc, sP, offa, offb, k4);
  }
  __syncthreads();                                         // the staging buffer is dead: G takes its place
  const unsigned long long t_1 = stats ? wall_clock64() : 0ull;
#pragma unroll
  for (int q = 0; q < TPW; ++q) {
    if (ti[q] < 0) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {                          // acc[r] = G[(lane >> 4) + 4 r][lane & 15] of the tile
      const int i = 16 * ti[q] + k4 + 4 * r, j = 16 * tj[q] + r16;
      if (i <= j && j < ncols) sG[at(i, j)] = acc[q][r];
    }
  }
  __syncthreads();
  double md = 0.0;
  for (int i = tid; i < ncols; i += 256) md = fmax(md, sG[at(i, i)]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) md = fmax(md, __shfl_xor(md, o, 64));
  if (lane == 0) s_red[wave] = md;
  __syncthreads();
  if (tid == 0) s_maxd = fmax(fmax(s_red[0], s_red[1]), fmax(s_red[2], s_red[3]));
  __syncthreads();
  const double maxd = s_maxd;
  const double eT = NOISE_C * (double)Eps<T>::v;
  const double thresh = fmax((double)n * 2.220446049250313e-16, eT * eT) * maxd;
  // ---- upper Cholesky in LDS, panels of 16 rows; at most K live rows (the data has K rows: what is beyond is rounding noise) ----
  const int nl = lds_chol_blocked(sG, at, ncols, thresh, K, sList, stats, (int)(((unsigned)b >> 3) + ((unsigned)b >> 8)) & 3);
  const unsigned long long t_2 = stats ? wall_clock64() : 0ull;
  // ---- rank compaction (rows with norm below NOISE_C eps_T |R|_F are dropped) and output ----
  const int cnt = lds_chol_compact(sG, at, ncols, nl, sList, sNrm, sPos, eT);
  if (tid == 0) mlive_out[b] = cnt;
  const double sc = maxd > 0.0 ? 1.0 / sqrt(maxd) : 1.0;
  const int c0 = lane, c1 = lane + 64;                     // packed columns of this lane -> columns of P (once)
  const int rc0 = (c0 / ilive) * inner + (c0 % ilive), rc1 = (c1 / ilive) * inner + (c1 % ilive);
  for (int q = wave; q < nl; q += 4) {
    const int pos = sPos[q];
    if (pos < 0) continue;
    const int j = sList[q];
    T *row = Rout + (long)pos * n;
    if (c0 < ncols) row[rc0] = c0 >= j ? T(sG[at(j, c0)] * sc) : T(0);
    if (c1 < ncols) row[rc1] = c1 >= j ? T(sG[at(j, c1)] * sc) : T(0);
  }
  if (stats && tid == 0) {
    const unsigned long long t_3 = wall_clock64();
    atomicAdd(stats + 0, 1ull); atomicAdd(stats + 1, (unsigned long long)K); atomicAdd(stats + 2, (unsigned long long)ncols);
    atomicAdd(stats + 3, t_1 - t_0); atomicAdd(stats + 4, t_2 - t_1); atomicAdd(stats + 5, t_3 - t_2);
    atomicAdd(stats + 6, (unsigned long long)nl);
    atomicAdd(stats + 12 + (ncols <= 64 ? 0 : ncols <= 80 ? 1 : ncols <= 96 ? 2 : 3), 1ull);
  }
}

template <typename T>
__global__ __launch_bounds__(256, 2) void colgram_chol_kernel(const T *__restrict__ Pg, long wP, int n, const int *__restrict__ kdyn,
                                                           int kdyn_mul, int kmax, T *__restrict__ Rg, long wR,
                                                           int *__restrict__ mlive_out, int inner,
                                                           const int *__restrict__ inner_live, int rcap, int decline_code,
                                                           int only_code, unsigned long long *__restrict__ flopc,
                                                           unsigned long long *__restrict__ bytec, int flop_stride) {
  static_assert(sizeof(T) == 4, "f32 element type");
  const int b = blockIdx.x;
  if (only_code != 0 && mlive_out[b] != only_code) return;      // second launch (larger rcap): the entries the first declined
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = kdyn ? max(0, min(kmax, kdyn[b] * kdyn_mul)) : kmax;
  const int ilive = inner_live ? min(inner, inner_live[b]) : inner;
  const int ncols = (n / inner) * ilive;
  if (ncols > CG_NC) {
    if (tid == 0) mlive_out[b] = decline_code;
    return;
  }
  extern __shared__ double cg_smem[];
  double *sR = cg_smem;                                   // finished factor rows, packed triangle: (j, c >= j) at cg_row(j) + c
  double *sRow = sR + ((size_t)rcap * CG_NC - (size_t)rcap * (rcap - 1) / 2);   // [CG_NC] row f of G
  float *sP = reinterpret_cast<float *>(sRow + CG_NC);    // [CG_NC][TM_LDM] chunk of P, transposed: [column][row]
  __shared__ double s_red[4], s_nrm[128], s_piv;
  __shared__ int s_first[2][4];
  __shared__ short s_pos[128];
  const T *P = Pg + (long)b * wP;
  T *Rout = Rg + (long)b * wR;
  // packed column c -> column of P
  const int my_r = tid < ncols ? (tid / ilive) * inner + (tid % ilive) : -1;

  // ---- G = P^T P over the packed columns, upper tiles in registers ----
  const int nt = (ncols + 15) >> 4, ntiles = nt * (nt + 1) / 2;
  if (flopc && tid == 0 && b % flop_stride == 0) {     // MFMA flops issued (16 x 16 x 2 per tile and row of P) and compulsory bytes
    atomicAdd(flopc, (unsigned long long)flop_stride * (unsigned long long)ntiles * 512ull * (unsigned long long)K);
    if (bytec) atomicAdd(bytec, (unsigned long long)flop_stride * 4ull * (unsigned long long)K * ncols);
  }
  constexpr int TPW = TM_TPW;
  const int r16 = lane & 15, k4 = lane >> 4;
  int ti[TPW], tj[TPW], offa[TPW], offb[TPW];
  int nq = 0;
#pragma unroll
  for (int q = 0; q < TPW; ++q) {
    int t = wave + 4 * q, i = 0;
    if (t < ntiles) { while (t >= nt - i) { t -= nt - i; ++i; } ti[q] = i; tj[q] = i + t; nq = q + 1; }
    else { ti[q] = -1; tj[q] = -1; }
    offa[q] = (16 * max(ti[q], 0) + r16) * TM_LDM;
    offb[q] = (16 * max(tj[q], 0) + r16) * TM_LDM;
  }
  tm_f64x4 acc[TPW];
#pragma unroll
  for (int q = 0; q < TPW; ++q)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[q][r] = 0.0;
  const int st_c = tid & (CG_NC - 1);                                    // staging: packed column of this thread
  const int st_r = st_c < ncols ? (st_c / ilive) * inner + (st_c % ilive) : -1;
  // columns of the staging buffer beyond the live ones (tile padding) are read by the MFMA operands: zero, once
  for (int e = tid + ncols * TM_LDM; e < CG_NC * TM_LDM; e += 256) sP[e] = 0.f;
  // unconditional loads (clamped row), 16 per thread and chunk; the chunk after the one being multiplied is in flight during
  // its MFMAs (two blocks per CU: little else hides the latency of the loads)
  float v[TM_KC / 2];
  auto issue = [&](const int k0) {
    const int kw = min(TM_KC, K - k0);
#pragma unroll
    for (int i = 0; i < TM_KC / 2; ++i) {
      const int k = min((tid >> 7) + 2 * i, kw - 1);
      v[i] = (float)P[(long)(k0 + k) * n + st_r];
    }
  };
  if (K > 0 && st_r >= 0) issue(0);
  for (int k0 = 0; k0 < K; k0 += TM_KC) {
    const int kw = min(TM_KC, K - k0);
    __syncthreads();
    if (st_r >= 0) {
#pragma unroll
      for (int i = 0; i < TM_KC / 2; ++i) {
        const int k = (tid >> 7) + 2 * i;
        sP[st_c * TM_LDM + k] = k < kw ? v[i] : 0.f;
      }
    }
    __syncthreads();
    if (k0 + TM_KC < K && st_r >= 0) issue(k0 + TM_KC);
    tm_gram_chunk_n(nq, acc, sP, offa, offb, k4);
  }
  __syncthreads();
  // diagonal -> sRow (used once, as the initial running diagonal)
#pragma unroll
  for (int q = 0; q < TPW; ++q) {
    if (ti[q] < 0 || ti[q] != tj[q]) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (r16 == k4 + 4 * r) sRow[16 * ti[q] + r16] = acc[q][r];
  }
  __syncthreads();
  double d = tid < ncols ? sRow[tid] : 0.0;             // running diagonal of the own column
  double md = d;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) md = fmax(md, __shfl_xor(md, o, 64));
  if (lane == 0) s_red[wave] = md;
  __syncthreads();
  const double maxd = fmax(fmax(s_red[0], s_red[1]), fmax(s_red[2], s_red[3]));
  const double eT = NOISE_C * (double)Eps<T>::v;
  const double thresh = fmax((double)n * 2.220446049250313e-16, eT * eT) * maxd;

  // ---- low-rank Cholesky ----
  int nl = 0, f = -1;
  for (int step = 0;; ++step) {
    int cand = (tid < ncols && tid > f && d > thresh) ? tid : 0x7fffffff;
    cand = wave_min_dpp(cand);
    if (lane == 0) s_first[step & 1][wave] = cand;
    __syncthreads();                                    // also: sRow / sR of the previous step are settled
    f = min(min(s_first[step & 1][0], s_first[step & 1][1]), min(s_first[step & 1][2], s_first[step & 1][3]));
    if (f == 0x7fffffff || nl >= K) break;              // the rank cannot exceed the K rows: later pivots are rounding noise
    if (nl == rcap) {                                   // rank above the cap of this launch
      if (tid == 0) mlive_out[b] = decline_code;
      return;
    }
    // row f of G out of the accumulators: tiles (f / 16, tj >= f / 16), row f % 16 = lanes with k4 == (f % 16) % 4, register (f % 16) / 4
    const int tr = f >> 4, il = f & 15, rsel = il >> 2;
#pragma unroll
    for (int q = 0; q < TPW; ++q) {
      if (ti[q] != tr) continue;
      const double v = rsel == 0 ? acc[q][0] : rsel == 1 ? acc[q][1] : rsel == 2 ? acc[q][2] : acc[q][3];
      if (k4 == (il & 3)) sRow[16 * tj[q] + r16] = v;
    }
    if (tid == f) s_piv = d;                            // the owner's running diagonal IS the pivot: no thread recomputes it
    __syncthreads();
    if (tid >= f && tid < ncols) {                      // (columns before the pivot: their entry of the row is zero, not stored)
      // four independent partial sums: the step is a chain of dependent f64 FMAs otherwise (nl of them, twice)
      double g0 = sRow[tid], g1 = 0.0, g2 = 0.0, g3 = 0.0;
      int j = 0;
      for (; j + 4 <= nl; j += 4) {
        const int r0 = cg_row(j), r1 = cg_row(j + 1), r2 = cg_row(j + 2), r3 = cg_row(j + 3);
        g0 = fma(-sR[r0 + f], sR[r0 + tid], g0);
        g1 = fma(-sR[r1 + f], sR[r1 + tid], g1);
        g2 = fma(-sR[r2 + f], sR[r2 + tid], g2);
        g3 = fma(-sR[r3 + f], sR[r3 + tid], g3);
      }
      for (; j < nl; ++j) g0 = fma(-sR[cg_row(j) + f], sR[cg_row(j) + tid], g0);
      const double g = (g0 + g1) + (g2 + g3);
      const double v = g * jr_rsq64(s_piv);
      sR[cg_row(nl) + tid] = v;
      if (tid > f) d -= v * v;
    } else if (tid >= nl && tid < f) {
      sR[cg_row(nl) + tid] = 0.0;                       // stored columns of the row before its pivot
    }
    ++nl;
  }
  __syncthreads();
  // ---- rank compaction (rows below NOISE_C eps_T |R|_F are dropped) and output ----
  for (int j = wave; j < nl; j += 4) {
    double a = 0.0;
    for (int c = j + lane; c < ncols; c += 64) { const double x = sR[cg_row(j) + c]; a += x * x; }
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
    mlive_out[b] = cnt;
  }
  __syncthreads();
  const double sc = maxd > 0.0 ? 1.0 / sqrt(maxd) : 1.0;
  if (my_r >= 0) {
    for (int j = 0; j < nl; ++j) {
      const int pos = s_pos[j];
      if (pos >= 0) Rout[(long)pos * n + my_r] = tid >= j ? T(sR[cg_row(j) + tid] * sc) : T(0);
    }
  }
}

// Used for DENSE states only (hint of the row absorbed before: carry rank above the caps of the thread-per-column
// kernels), with the 88-row cap.  Measured on one MI355X, C4, two steps:
//   dense state (noise 1.0, 4096 walkers): Gram + Cholesky 660 ms (streaming Gram + blocked Cholesky) -> 608 ms here;
//   headline state (32768 walkers, rank ~10, cap 16): 200 ms against 194 ms of gram_chol_lowrank_kernel -- no gain: both
//   are bound by the same chain of dependent HBM round trips (stage P, barriers), not by the arithmetic (2 us of MFMA per
//   walker), so the headline keeps the thread-per-column kernel.  A launch whose blocks ask for the 118 KB of the 96-row cap
//   runs one block per CU even when every block returns at once (+400 us per site on the headline batch): one launch per
//   site, never a cascade of caps.
template <typename T>
inline void launch_colgram_chol(hipStream_t s, int nbatch, const T *P, long wP, int n, const int *kdyn, int kdyn_mul, int kmax,
                                T *R, long wR, int *mlive, int inner, const int *inner_live, int decline_code, bool hint_dense) {
  (void)hint_dense;
  constexpr bool lowrank_form = false;
  if (!lowrank_form) {   // dense walkers: blocked Cholesky of the packed triangle in LDS (colgram_dense_kernel), two size classes
    auto smem_of = [](int nc) {
      const size_t tri = sizeof(double) * (size_t)nc * (nc + 1) / 2, stg = (sizeof(float) * (size_t)nc * TM_LDM + 7) & ~(size_t)7;
      return std::max(tri, stg) + sizeof(double) * (size_t)nc + sizeof(short) * 2 * (size_t)nc + 64;
    };
    unsigned long long *st = nbatch >= 1024 ? cg_stats_dev() : nullptr;
    constexpr bool one_class = false;
    int lo = 0;
    // (a third class of 80 columns at four blocks per CU -- four of five walkers of a full-rank state -- was measured: cholesky
    // category 86 -> 91 ms per step: at 128 registers the kernel spills and four Gram phases share one MFMA pipe)
    if (!one_class) {
      const size_t sm = smem_of(96);
      allow_dynamic_lds(reinterpret_cast<const void *>(&colgram_dense_kernel<T, 96, 6, 3>), sm);
      hipLaunchKernelGGL((colgram_dense_kernel<T, 96, 6, 3>), dim3(nbatch), dim3(256), sm, s, P, wP, n, kdyn, kdyn_mul, kmax, R, wR, mlive,
                         inner, inner_live, lo, 0, decline_code, tg_flop_counter, tg_byte_counter, nbatch >= 256 ? 64 : 1, st);
      lo = 96;
    }
    const size_t sm = smem_of(128);
    allow_dynamic_lds(reinterpret_cast<const void *>(&colgram_dense_kernel<T, 128, 9, 2>), sm);
    hipLaunchKernelGGL((colgram_dense_kernel<T, 128, 9, 2>), dim3(nbatch), dim3(256), sm, s, P, wP, n, kdyn, kdyn_mul, kmax, R, wR, mlive,
                       inner, inner_live, lo, 1, decline_code, tg_flop_counter, tg_byte_counter, nbatch >= 256 ? 64 : 1, st);
    PG_CHECK_HIP(hipGetLastError());
    return;
  }
  constexpr int rcap_env = 88;   // (diagnostics: 0 = Gram phase only)
  const int rcap = rcap_env;
  const size_t sm = colgram_chol_smem_bytes(rcap);
  allow_dynamic_lds(reinterpret_cast<const void *>(&colgram_chol_kernel<T>), sm);
  hipLaunchKernelGGL(colgram_chol_kernel<T>, dim3(nbatch), dim3(256), sm, s, P, wP, n, kdyn, kdyn_mul, kmax, R, wR, mlive, inner,
                     inner_live, rcap, decline_code, 0, tg_flop_counter, tg_byte_counter, nbatch >= 256 ? 64 : 1);
  PG_CHECK_HIP(hipGetLastError());
}

}  // namespace pepsgpu
