// M = R Tt of the truncation pass for DENSE walkers (round 3): one workgroup per walker, operands through LDS once.
//
//     M[i][(u, k2)] = sum_{(l, a)} R[i][(l, a)] Tt[(l, a)][(u, k2)]       i < live carry rows (<= 256), (l, a) <= 256, (u, k2) <= 256
//
// (bmps_impl.h:254: the product Q_i . (u s) the reference hands to its SVD, in the Q-less form of engine_impl.h).  The generic
// LDS-tiled tensor GEMM runs this shape at 17 % of the f32 MFMA peak (64 x 64 tiles, a barrier pair per 16 k, offsets through
// tables); here the eight waves of a 512-thread workgroup each own one 32-column block of M and walk the live 32-row blocks
// (one B operand read feeds up to eight v_mfma_f32_32x32x2_f32), R and Tt pass through LDS in chunks of 16 k (R transposed on the
// way in so that the A operand is one conflict-free ds_read_b32 per lane; Tt permuted into the column order of M, its rows of
// dead a and columns of dead k2 zeroed), double buffered, the next chunk in flight during the MFMAs.
#pragma once
#include "common.h"
#include "tgemm.h"

namespace pepsgpu {

constexpr int MG_BK = 16, MG_PITCH = 264;
inline size_t mgemm_dense_smem_bytes() { return sizeof(float) * 2 * 2 * MG_BK * MG_PITCH; }

// u_dim, k2_dim: static extents of the column sub-indices of M (uk = u_dim * k2_dim <= 256, multiple of 32); tt_u_inner: Tt is
// stored [la][k2][u] (u innermost) instead of [la][u][k2]; a_dim: static extent of the inner contracted sub-index a
// (la = l_dim * a_dim <= 256, multiple of 16); a_live / k2_live / m_live: per-walker live extents (nullptr: static)
__global__ __launch_bounds__(512, 2) void mgemm_dense_kernel(const float *__restrict__ Rg, long wR, const float *__restrict__ Tg, long wT,
                                                             float *__restrict__ Mg, long wM, int m, int la, int a_dim, int u_dim,
                                                             int k2_dim, int tt_u_inner, const int *__restrict__ m_live, int m_mul,
                                                             const int *__restrict__ a_live, const int *__restrict__ k2_live,
                                                             unsigned long long *__restrict__ flopc, unsigned long long *__restrict__ bytec,
                                                             int flop_stride, int tri) {
  // tri (round 6): R is a row-compacted upper-triangular factor (row i is zero before column i): the 32-row blocks that start at or
  // beyond the end of a k-chunk hold zeros there and are not multiplied -- 40 % of the MFMAs at 220 live rows
  extern __shared__ float mg_smem[];
  const int b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ml = m_live ? max(0, min(m, m_live[b] * m_mul)) : m;
  const int al = a_live ? max(0, min(a_dim, a_live[b])) : a_dim;
  const int kl = k2_live ? max(0, min(k2_dim, k2_live[b])) : k2_dim;
  const int uk = u_dim * k2_dim;
  if (ml <= 0) return;
  if (flopc && tid == 0 && b % flop_stride == 0) {
    // (executed: with the triangular form the row blocks below a k-chunk are skipped -- counted as the trapezoid they cover)
    unsigned long long rows_k = (unsigned long long)ml * (la / a_dim * al);
    if (tri) {
      rows_k = 0;
      for (int ch = 0; ch < la / MG_BK; ++ch) rows_k += (unsigned long long)min(ml, ((ch * MG_BK + MG_BK + 31) >> 5) << 5) * MG_BK * al / a_dim;
    }
    atomicAdd(flopc, 2ull * flop_stride * rows_k * (unsigned long long)(u_dim * kl));
    if (bytec) atomicAdd(bytec, 4ull * flop_stride * ((unsigned long long)ml * la + (unsigned long long)la * uk + (unsigned long long)ml * uk));
  }
  const float *R = Rg + (long)b * wR;
  const float *Tt = Tg + (long)b * wT;
  float *M = Mg + (long)b * wM;
  float *sA = mg_smem;                                   // [2][MG_BK][MG_PITCH]   sA[k][i] = R[i][k0 + k]
  float *sB = mg_smem + 2 * MG_BK * MG_PITCH;            // [2][MG_BK][MG_PITCH]   sB[k][j] = Tt[k0 + k][j], j in the column order of M
  const int nI = (ml + 31) >> 5;                         // live 32-row blocks (wave-uniform, block-uniform)
  const int nch = la / MG_BK;
  tg_f32x16 acc[8];
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  // chunk loaders: A = 256 rows x 16 k (one float4 per (row, quarter): 1024 float4, two per thread), B = 16 rows x 256 (1024 float4)
  float4 av[2], bv[2];
  auto issue = [&](int ch) {
    const int k0 = ch * MG_BK;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int e = tid + 512 * q;
      const int row = e >> 2, qu = e & 3;
      av[q] = row < ml ? *reinterpret_cast<const float4 *>(R + (long)row * la + k0 + 4 * qu) : make_float4(0.f, 0.f, 0.f, 0.f);
      const int kr = e >> 6, c4 = e & 63;                 // B: row k0 + kr, float4 number c4 of the row
      const bool ok = 4 * c4 < uk && ((k0 + kr) % a_dim) < al;
      bv[q] = ok ? *reinterpret_cast<const float4 *>(Tt + (long)(k0 + kr) * uk + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto lay = [&](int buf) {
    float *dA = sA + buf * MG_BK * MG_PITCH, *dB = sB + buf * MG_BK * MG_PITCH;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int e = tid + 512 * q;
      const int row = e >> 2, qu = e & 3;
      const float a4[4] = {av[q].x, av[q].y, av[q].z, av[q].w};
#pragma unroll
      for (int z = 0; z < 4; ++z) dA[(4 * qu + z) * MG_PITCH + row] = a4[z];
      const int kr = e >> 6, c4 = e & 63;
      const float b4[4] = {bv[q].x, bv[q].y, bv[q].z, bv[q].w};
#pragma unroll
      for (int z = 0; z < 4; ++z) {
        const int s = 4 * c4 + z;                          // position inside the row of Tt
        int uu, kk;
        if (tt_u_inner) { kk = s / u_dim; uu = s - kk * u_dim; } else { uu = s / k2_dim; kk = s - uu * k2_dim; }
        if (s < uk) dB[kr * MG_PITCH + uu * k2_dim + kk] = kk < kl ? b4[z] : 0.f;
      }
    }
  };
  const int half = lane >> 5, l31 = lane & 31;
  auto mma = [&](int buf, int tmax) {
    const float *pA = sA + buf * MG_BK * MG_PITCH, *pB = sB + buf * MG_BK * MG_PITCH;
#pragma unroll
    for (int kk = 0; kk < MG_BK; kk += 2) {
      const float bb = pB[(kk + half) * MG_PITCH + 32 * wave + l31];
#pragma unroll
      for (int t = 0; t < 8; ++t)
        if (t < tmax) {
          const float a = pA[(kk + half) * MG_PITCH + 32 * t + l31];
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb, acc[t], 0, 0, 0);
        }
    }
  };
  const bool wave_on = 32 * wave < uk;
  if (nch > 0) { issue(0); lay(0); }
  __syncthreads();
  for (int ch = 0; ch < nch; ++ch) {
    if (ch + 1 < nch) issue(ch + 1);
    if (wave_on) mma(ch & 1, tri ? min(nI, (ch * MG_BK + MG_BK + 31) >> 5) : nI);
    if (ch + 1 < nch) lay((ch + 1) & 1);
    __syncthreads();
  }
  // accumulator r of a lane = row 8 (r / 4) + 4 half + (r % 4), column l31 of the 32 x 32 tile
  if (wave_on) {
    const int j = 32 * wave + l31;
#pragma unroll
    for (int t = 0; t < 8; ++t)
      if (t < nI) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int i = 32 * t + 8 * (r >> 2) + 4 * half + (r & 3);
          if (i < ml && j < uk) M[(long)i * uk + j] = acc[t][r];
        }
      }
  }
}

inline bool mgemm_dense_ok(int m, int la, int a_dim, int u_dim, int k2_dim, long wR, long wT, const void *R, const void *Tt) {
  const int uk = u_dim * k2_dim;
  return m <= 256 && la <= 256 && la % MG_BK == 0 && la % a_dim == 0 && uk <= 256 && uk % 32 == 0 && wR % 4 == 0 && wT % 4 == 0 &&
         (((uintptr_t)R) & 15) == 0 && (((uintptr_t)Tt) & 15) == 0;
}

inline void launch_mgemm_dense(hipStream_t s, int nbatch, const float *R, long wR, const float *Tt, long wT, float *M, long wM, int m, int la,
                               int a_dim, int u_dim, int k2_dim, int tt_u_inner, const int *m_live, int m_mul, const int *a_live,
                               const int *k2_live, unsigned long long *flopc, unsigned long long *bytec, int tri = 0) {
  const size_t smem = mgemm_dense_smem_bytes();
  allow_dynamic_lds(reinterpret_cast<const void *>(&mgemm_dense_kernel), smem);
  hipLaunchKernelGGL(mgemm_dense_kernel, dim3(nbatch), dim3(512), smem, s, R, wR, Tt, wT, M, wM, m, la, a_dim, u_dim, k2_dim, tt_u_inner, m_live,
                     m_mul, a_live, k2_live, flopc, bytec, nbatch >= 256 ? 64 : 1, tri);
  PG_CHECK_HIP(hipGetLastError());
}

}  // namespace pepsgpu
