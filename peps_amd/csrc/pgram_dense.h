// Forward Gram of a DENSE walker batch at the bulk shapes of C4 without the tensor P in HBM (round 3):
//
//     X[m,l,p,a2] = sum_a R[m,l,a] A[a,p,a2]                  (bmps_impl.h:806)
//     P[m,u,l2,a2] = sum_{l,p} X[m,l,p,a2] W[l,p,l2,u]        (bmps_impl.h:807, :815-817)
//     G = P^T P   (rows (m,u), columns (l2,a2): 256 x 256 float64, upper 16 x 16 tiles)
//
// On a real PEPS the carry R has 190-240 live rows: P is 1 900 x 256 floats = 2 MB per walker and site.  The chained
// contraction wrote it (16 GB per launch of 8192 walkers: 4.9 ms, bound by the write), the LDS Gram kernel read it back (12.4 ms,
// bound by the f64 matrix rate) -- and nothing else ever reads P: the forward pass keeps only the factor of G.  Here one workgroup
// (8 waves) per walker walks the carry in chunks of PG_MC = 4 rows:
//   S1  X chunk (32 x 64) = R chunk (32 x 32, from global, 16-byte loads) . A (32 x 64, in registers for the whole kernel):
//       eight 16 x 16 tiles of v_mfma_f32_16x16x4_f32, one per wave -> LDS (8 KB), laid out as the B operand of S2
//   S2  P chunk (32 rows (mc,u) x 256 columns (l2,a2)) = W (64 x 16, in registers for the whole kernel) . X chunk (16 x 128):
//       eight 32 x 32 tiles of v_mfma_f32_32x32x2_f32, one per wave -> LDS chunk buffer (double buffered), dead a2 / rows zeroed
//   S3  G += chunk^T chunk: exactly the accumulation of gram_cols_lds_kernel (17 tiles per wave, v_mfma_f64_16x16x4_f64)
// Two barriers per chunk; the R rows of the next chunk are requested before S3.  Output contract of gram_cols_lds_kernel (upper
// tiles of G, n = 256).  Any rank is handled by the Cholesky that follows: the dense-batch hint only selects the route.
#pragma once
#include <type_traits>
#include "common.h"
#include "gram.h"

namespace pepsgpu {

constexpr int PG_L = 8, PG_P = 2, PG_A = 32, PG_A2 = 32, PG_L2 = 8, PG_U = 8;
constexpr int PG_MC = 4;                          // carry rows per chunk: PG_MC * PG_U = 32 rows of P = one chunk of the LDS Gram
constexpr int PG_N = PG_L2 * PG_A2;               // 256 columns of P
constexpr int PG_XPITCH = PG_MC * PG_A2;          // X chunk as [k = (l,p)][j = (mc,a2)]: 16 x 128
static_assert(PG_MC * PG_U == GL_KC && PG_N <= GL_PITCH, "chunk of the LDS Gram");

typedef float pg_f32x4 __attribute__((ext_vector_type(4)));
typedef float pg_f32x16 __attribute__((ext_vector_type(16)));

inline size_t pgram_dense_smem_bytes() { return sizeof(float) * (2 * GL_KC * GL_PITCH + PG_L * PG_P * PG_XPITCH); }

inline bool pgram_dense_ok(int l, int p, int a, int a2, int l2, int u, long wR, long wA, const void *R, const void *A) {
  return l == PG_L && p == PG_P && a == PG_A && a2 == PG_A2 && l2 == PG_L2 && u == PG_U && wR % 4 == 0 && wA % 4 == 0 &&
         (((uintptr_t)R) & 15) == 0 && (((uintptr_t)A) & 15) == 0;
}

// Rg: carry [m][l][a] per walker (stride wR), mdyn (nullable): live rows = min(m, mdyn[b] * mdyn_mul)
// Ag: boundary tensor [a][p][a2] per walker (stride wA), alive / a2live (nullable): live extents of its two bonds
// Wg + wsel[b * wsel_inc] * wsel_mul: the site tensor of the walker's configuration, leg strides sl, sp, sl2, su
__global__ __launch_bounds__(512, 2) void pgram_dense_kernel(const float *__restrict__ Rg, long wR, int m, const int *__restrict__ mdyn,
                                                           int mdyn_mul, const float *__restrict__ Ag, long wA,
                                                           const int *__restrict__ alive_p, const int *__restrict__ a2live_p,
                                                           const float *__restrict__ Wg, const int *__restrict__ wsel, int wsel_inc,
                                                           long wsel_mul, int sl, int sp, int sl2, int su,
                                                           double *__restrict__ Gg, long wG,
                                                           unsigned long long *__restrict__ flopc,
                                                           unsigned long long *__restrict__ bytec, int flop_stride) {
  extern __shared__ float pg_smem[];
  float *sPbuf = pg_smem;                                   // [2][GL_KC][GL_PITCH]
  float *sX = pg_smem + 2 * GL_KC * GL_PITCH;               // [16][PG_XPITCH]
  const int b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int mlive = mdyn ? min(m, mdyn[b] * mdyn_mul) : m;
  const int alive = alive_p ? min(PG_A, alive_p[b]) : PG_A;
  const int a2live = a2live_p ? min(PG_A2, a2live_p[b]) : PG_A2;
  const int K = mlive * PG_U;
  if (flopc && tid == 0 && b % flop_stride == 0) {
    atomicAdd(flopc, (unsigned long long)flop_stride * PG_N * PG_N * K);
    if (bytec) atomicAdd(bytec, (unsigned long long)flop_stride * ((unsigned long long)mlive * PG_L * PG_A * 4 + (unsigned long long)PG_N * PG_N * 4));
  }
  const float *R = Rg + (long)b * wR;
  const float *A = Ag + (long)b * wA;
  const float *W = Wg + (wsel ? (long)wsel[(long)b * wsel_inc] * wsel_mul : 0L);
  double *G = Gg + (long)b * wG;
  const int i16 = lane & 15, k4 = lane >> 4, l31 = lane & 31, half = lane >> 5;
  const int nch = (mlive + PG_MC - 1) / PG_MC;

  // ---- operands that stay in registers for the whole kernel ----
  // S1, B operand: A[a][(p,a2)] for the wave's 16 columns jt; the k index lane k4 supplies to MFMA (s, t) is a = 16 s + 4 k4 + t
  // (any bijection of the 32 a serves as long as the A operand -- the 16-byte loads of R below -- uses the same one)
  const int it1 = wave >> 2, jt1 = wave & 3;
  float bA[2][4];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int a = 16 * s + 4 * k4 + t;
      bA[s][t] = a < alive ? A[a * (PG_P * PG_A2) + 16 * jt1 + i16] : 0.f;
    }
  // S2, A operand: W as [i = (u,l2)][k = (l,p)], the wave's 32 rows ih; lane (l31, half) supplies k = 2 s + half: l = s, p = half
  const int ih2 = wave >> 2, mc2 = wave & 3;
  float aW[PG_L];
  {
    const int i = 32 * ih2 + l31, uu = i >> 3, ll2 = i & 7;
#pragma unroll
    for (int s = 0; s < PG_L; ++s) aW[s] = W[s * sl + half * sp + ll2 * sl2 + uu * su];
  }
  // S1, A operand of the chunk being prepared: rows (mc,l) = 16 it1 + i16 of the chunk, two 16-byte loads along a
  pg_f32x4 rv[2];
  auto issue_r = [&](int ch) {
    const int i = 16 * it1 + i16, mrow = ch * PG_MC + (i >> 3);
    const bool ok = mrow < mlive;
    const float *src = R + ((long)(ch * PG_MC) * PG_L + i) * PG_A + 4 * k4;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      pg_f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (ok) v = *reinterpret_cast<const pg_f32x4 *>(src + 16 * s);
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (16 * s + 4 * k4 + t >= alive) v[t] = 0.f;       // columns of the carry beyond the live bond were never written
      rv[s] = v;
    }
  };
  auto stage12 = [&](int ch, int buf) {
    // S1: X chunk tile (16 rows (mc,l) x 16 columns (p,a2)) -> sX[(l,p)][(mc,a2)]
    {
      pg_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(rv[s][t], bA[s][t], acc, 0, 0, 0);
      const int col = 16 * jt1 + i16, pp = col >> 5, aa2 = col & 31;
#pragma unroll
      for (int r = 0; r < 4; ++r) {                         // acc[r] = C[4 k4 + r][i16] of the tile (f32 form: NOT the f64 map)
        const int i = 16 * it1 + 4 * k4 + r, mc = i >> 3, ll = i & 7;
        sX[(ll * PG_P + pp) * PG_XPITCH + mc * PG_A2 + aa2] = acc[r];
      }
    }
    __syncthreads();
    // S2: P chunk tile (32 rows (u,l2) x 32 columns a2 of carry row mc2) -> chunk buffer [(mc,u)][(l2,a2)]
    {
      pg_f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int s = 0; s < PG_L; ++s) {
        const float bx = sX[(2 * s + half) * PG_XPITCH + mc2 * PG_A2 + l31];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aW[s], bx, acc, 0, 0, 0);
      }
      float *dst = sPbuf + buf * GL_KC * GL_PITCH;
      const bool live = (ch * PG_MC + mc2 < mlive) && l31 < a2live;
#pragma unroll
      for (int r = 0; r < 16; ++r) {                        // acc[r]: row 8 (r / 4) + 4 half + (r % 4) of the tile, column l31
        const int i = 32 * ih2 + 8 * (r >> 2) + 4 * half + (r & 3), uu = i >> 3, ll2 = i & 7;
        dst[(mc2 * PG_U + uu) * GL_PITCH + ll2 * PG_A2 + l31] = live ? acc[r] : 0.f;
      }
    }
  };
  // S3 per wave: the 17 tiles of gram_cols_lds_kernel's deal (GlTile<W>)
  auto run = [&](auto wc) {
    constexpr int Wv = decltype(wc)::value;
    gr_f64x4 acc[17];
#pragma unroll
    for (int t = 0; t < 17; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[t][r] = 0.0;
    if (nch > 0) { issue_r(0); stage12(0, 0); }
    __syncthreads();
    for (int ch = 0; ch < nch; ++ch) {
      if (ch + 1 < nch) issue_r(ch + 1);
      const float *src = sPbuf + (ch & 1) * GL_KC * GL_PITCH;
#pragma unroll 2
      for (int s2 = 0; s2 < GL_KC / 4; ++s2) {
        const float *row = src + (4 * s2 + k4) * GL_PITCH + i16;
        double seg[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) seg[q] = (double)row[16 * q];
#pragma unroll
        for (int t = 0; t < 17; ++t)
          acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(seg[GlTile<Wv>::x(t)], seg[GlTile<Wv>::c(t)], acc[t], 0, 0, 0);
      }
      if (ch + 1 < nch) stage12(ch + 1, (ch + 1) & 1);     // (its barrier separates S1 from S2; sX is free: S2 of chunk ch is done)
      __syncthreads();
    }
#pragma unroll
    for (int t = 0; t < 17; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * GlTile<Wv>::x(t) + k4 + 4 * r, j = 16 * GlTile<Wv>::c(t) + i16;
        G[(long)i * PG_N + j] = acc[t][r];
      }
  };
  switch (wave) {      // (every branch executes the same number of barriers)
    case 0: run(std::integral_constant<int, 0>{}); break;
    case 1: run(std::integral_constant<int, 1>{}); break;
    case 2: run(std::integral_constant<int, 2>{}); break;
    case 3: run(std::integral_constant<int, 3>{}); break;
    case 4: run(std::integral_constant<int, 4>{}); break;
    case 5: run(std::integral_constant<int, 5>{}); break;
    case 6: run(std::integral_constant<int, 6>{}); break;
    default: run(std::integral_constant<int, 7>{}); break;
  }
}

inline void launch_pgram_dense(hipStream_t s, int nbatch, const float *R, long wR, int m, const int *mdyn, int mdyn_mul, const float *A,
                               long wA, const int *alive, const int *a2live, const float *W, const int *wsel, int wsel_inc,
                               long wsel_mul, int sl, int sp, int sl2, int su, double *G, unsigned long long *flopc,
                               unsigned long long *bytec) {
  const size_t smem = pgram_dense_smem_bytes();
  allow_dynamic_lds(reinterpret_cast<const void *>(&pgram_dense_kernel), smem);
  hipLaunchKernelGGL(pgram_dense_kernel, dim3(nbatch), dim3(512), smem, s, R, wR, m, mdyn, mdyn_mul, A, wA, alive, a2live, W, wsel, wsel_inc,
                     wsel_mul, sl, sp, sl2, su, G, (long)PG_N * PG_N, flopc, bytec, nbatch >= 256 ? 64 : 1);
  PG_CHECK_HIP(hipGetLastError());
}

}  // namespace pepsgpu
