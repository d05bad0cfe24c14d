// Batched strided tensor-contraction GEMM for gfx950 (MI355X).
//
//   C[b][(i0,i1,i2),(j0,j1,j2)] (+)= sum_{(k0,k1,k2)} A[b][(i..),(k..)] * B[b][(k..),(j..)]
//
// Every index group (I, J, K) is a product of up to three sub-indices with arbitrary element
// strides, so the leg permutations the reference performs with qlten::Transpose before each
// qlten::Contract (e.g. include/qlpeps/one_dim_tn/boundary_mps/bmps_impl.h:806-817) are folded
// into the address computation of the global->LDS staging; operands are never permuted in HBM.
// `b` runs over (walker x candidate); an optional per-batch selector adds sel[b]*mul to an
// operand base, which is how a walker's configuration picks its projected site tensor out of
// the shared SITPS buffer (reference: tensor_network_2d_basic_impl.h:24-74 copies it instead).
//
// Math: f32 -> v_mfma_f32_32x32x2_f32, f64 (and f32 inputs accumulated in f64 for the Gram
// matrices) -> v_mfma_f64_16x16x4_f64; 64x64x16 LDS tiles, 4 waves, register-prefetched
// global loads.  A VALU path (PEPSGPU_NO_MFMA=1) exists only to cross-check the MFMA lane maps.
#pragma once
#include "common.h"
#include "cplx.h"

namespace pepsgpu {

// Per-walker dynamic extent of ONE sub-index: min(static dim, p[b] * mul).  Default (mask = 0):
// the index space is compacted -- the flattened index runs over the reduced dims with the static
// strides, elements beyond are neither read nor written.  mask = 1 (I and J only): the static
// tiling is kept, the operand is read as zero beyond the extent and C is written (zeros) there.
struct TgDyn {
  const int *p = nullptr;
  int mul = 1;
  int mask = 0;
  int div = 1;      // the extent of batch entry b is p[b / div] (candidates of one walker share its live bonds)
};

struct TGemmDesc {
  int I[3] = {1, 1, 1}, J[3] = {1, 1, 1}, K[3] = {1, 1, 1};  // sub-dims, innermost last
  int sAi[3] = {0, 0, 0}, sAk[3] = {0, 0, 0};
  int sBk[3] = {0, 0, 0}, sBj[3] = {0, 0, 0};
  int sCi[3] = {0, 0, 0}, sCj[3] = {0, 0, 0};
  long wA = 0, wB = 0, wC = 0;   // batch strides (elements)
  const int *selA = nullptr, *selB = nullptr;
  int selA_inc = 0, selB_inc = 0;    // selector read at sel[b * inc]
  long selA_mul = 0, selB_mul = 0;   // base += sel * mul
  int bdivA = 1, bdivB = 1, bdivC = 1;  // operand batch index = b / bdiv (candidates share env.)
  int seldivA = 1, seldivB = 1;         // selector index = (b / seldiv) * inc
  // Per-walker dynamic extents (rank-adaptive carry): only the first dynI[b]*dynI_mul values of the
  // flattened I index (resp. K index) exist; tiles beyond exit at once, C rows beyond are not written.
  const int *dynI = nullptr, *dynK = nullptr;
  int dynI_mul = 1, dynK_mul = 1;
  // Per-walker dynamic extents of single sub-indices (live bond dimensions of the boundary MPS)
  TgDyn dI[3], dJ[3], dK[3];
  int Imask[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, Jmask[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff};  // set by the kernel
  int nbatch = 1;
  int accumulate = 0;
  int upper_only = 0;   // symmetric result (Gram): tiles strictly below the diagonal are not computed
  int conjA = 0, conjB = 0;   // complex element types: the operand enters conjugated (P^H P, T V^H); no-op for real types
  const int *batch_flag = nullptr;   // when set, batch entry b runs only if batch_flag[b] < 0
  unsigned long long *flopc = nullptr;   // profiling: += 2*I*J*K of the extents actually contracted (per batch entry)
  unsigned long long *bytec = nullptr;   // profiling: += bytes of the live operand and result elements (compulsory traffic)
  int flop_stride = 1;                    // ... sampled: every flop_stride-th batch entry adds flop_stride times its count
  double alpha = 1.0;
  // Normalisation of a result WITHOUT a pass over it (wave-per-tile kernels, one block per batch entry): the launch that
  // writes C[b] sums the squares of what it stores (registers) and leaves scale_out[b] = 1 / |C[b]|, norm_log[b] += log |C[b]|
  // (a zero / non-finite norm: scale 1, norm_flag[b] = 1); the launch that reads C[b] next multiplies its own result by
  // scale_in[b].  Replaces normalize_kernel on the carried tensor of the truncation pass (one launch and a read + write of
  // the tensor per site).
  float *scale_out = nullptr;
  double *norm_log = nullptr;
  int *norm_flag = nullptr;
  const float *scale_in = nullptr;
  // host-side routing hint: the live extents of this launch are large (dense carry: hundreds of rows) -> the LDS-tiled 64 x 64
  // kernel instead of the wave-per-tile kernel that is built for extents of a few tens (M = R Tt of a dense walker batch:
  // 233 -> 138 ms per step of 2048 walkers)
  int prefer_tiled = 0;
  // wave-per-tile kernel: float64 accumulation on the f64 matrix cores from the f32 operands (tg_direct_body_f64; round 4 drained
  // f32 chains of 8 products into float64 registers instead, which removed a third of the error only).  For the contractions of the
  // truncation pass whose f32 accumulation shows in the amplitude of a dense state.
  int acc64 = 0;

  __host__ __device__ int Itot() const { return I[0] * I[1] * I[2]; }
  __host__ __device__ int Jtot() const { return J[0] * J[1] * J[2]; }
  __host__ __device__ int Ktot() const { return K[0] * K[1] * K[2]; }
};

constexpr int TG_BM = 64, TG_BN = 64, TG_BK = 16, TG_KTAB = 2048;

template <typename T> struct TgPitch { static constexpr int v = 64; };
template <> struct TgPitch<double> { static constexpr int v = 80; };  // 640 B: halves land 128 B apart
template <> struct TgPitch<c128> { static constexpr int v = 65; };

// offset of flattened index idx; -1 when a sub-index is at or beyond its mask limit
__device__ __forceinline__ int tg_off3m(int idx, const int *dims, const int *strides, const int *lim) {
  int i2 = idx % dims[2];
  int r = idx / dims[2];
  int i1 = r % dims[1];
  int i0 = r / dims[1];
  if (i2 >= lim[2] || i1 >= lim[1] || i0 >= lim[0]) return -1;
  return i0 * strides[0] + i1 * strides[1] + i2 * strides[2];
}

__device__ __forceinline__ int tg_off3(int idx, const int *dims, const int *strides) {
  int i2 = idx % dims[2];
  int r = idx / dims[2];
  int i1 = r % dims[1];
  int i0 = r / dims[1];
  return i0 * strides[0] + i1 * strides[1] + i2 * strides[2];
}

typedef float tg_f32x16 __attribute__((ext_vector_type(16)));
typedef double tg_f64x4 __attribute__((ext_vector_type(4)));

template <typename TA, typename TB, typename TC, typename TAcc, bool USE_MFMA>
__device__ __forceinline__ void tgemm_tile(const TGemmDesc &d, const TA *__restrict__ Ag, const TB *__restrict__ Bg,
                                           TC *__restrict__ Cg, const int i0, const int Itot, const int Ktot) {
  constexpr int PITCH = TgPitch<TAcc>::v;
  __shared__ TAcc As[TG_BK][PITCH];
  __shared__ TAcc Bs[TG_BK][PITCH];
  __shared__ int offAi[TG_BM], offBj[TG_BN], offCi[TG_BM], offCj[TG_BN];
  __shared__ int offAk[TG_KTAB], offBk[TG_KTAB];

  const int tid = threadIdx.x;
  const int b = blockIdx.z;
  const int j0 = blockIdx.y * TG_BN;
  const int Jtot = d.Jtot();

  long baseA = (long)(b / d.bdivA) * d.wA, baseB = (long)(b / d.bdivB) * d.wB;
  if (d.selA) baseA += (long)d.selA[(long)(b / d.seldivA) * d.selA_inc] * d.selA_mul;
  if (d.selB) baseB += (long)d.selB[(long)(b / d.seldivB) * d.selB_inc] * d.selB_mul;
  const TA *A = Ag + baseA;
  const TB *B = Bg + baseB;
  TC *C = Cg + (long)(b / d.bdivC) * d.wC;

  if (tid < TG_BM) {
    int i = i0 + tid;
    offAi[tid] = (i < Itot) ? tg_off3m(i, d.I, d.sAi, d.Imask) : -1;
    offCi[tid] = (i < Itot) ? tg_off3(i, d.I, d.sCi) : -1;
  } else if (tid < TG_BM + TG_BN) {
    int j = j0 + tid - TG_BM;
    offBj[tid - TG_BM] = (j < Jtot) ? tg_off3m(j, d.J, d.sBj, d.Jmask) : -1;
    offCj[tid - TG_BM] = (j < Jtot) ? tg_off3(j, d.J, d.sCj) : -1;
  }

  __syncthreads();   // offsets visible even when the K loop is empty (dynamic K extent 0)
  // staging order: run consecutive threads along whichever index is closer to unit stride
  const bool a_ifast = d.sAi[2] <= d.sAk[2];
  const bool b_jfast = d.sBj[2] <= d.sBk[2];

  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  tg_f32x16 acc32;
  tg_f64x4 acc64[2][2];
  tg_f64x4 accI[2][2];     // complex: imaginary parts (acc64 holds the real parts)
  TAcc accv[4][4];
  constexpr bool CPLX = is_cplx<TAcc>::value;
  if constexpr (USE_MFMA && CPLX) {
    for (int a = 0; a < 2; ++a)
      for (int c = 0; c < 2; ++c)
        for (int r = 0; r < 4; ++r) { acc64[a][c][r] = 0.0; accI[a][c][r] = 0.0; }
  } else if constexpr (USE_MFMA) {
    if constexpr (sizeof(TAcc) == 4) {
      for (int r = 0; r < 16; ++r) acc32[r] = 0.f;
    } else {
      for (int a = 0; a < 2; ++a)
        for (int c = 0; c < 2; ++c)
          for (int r = 0; r < 4; ++r) acc64[a][c][r] = 0.0;
    }
  } else {
    for (int a = 0; a < 4; ++a)
      for (int c = 0; c < 4; ++c) accv[a][c] = TAcc(0);
  }

  TA va[4];
  TB vb[4];
  unsigned okm = 0u;      // bit 2 r: element r of A exists, bit 2 r + 1: of B
  for (int kc = 0; kc < Ktot; kc += TG_KTAB) {
    const int kchunk = min(TG_KTAB, Ktot - kc);
    __syncthreads();
    for (int k = tid; k < kchunk; k += 256) {
      offAk[k] = tg_off3(kc + k, d.K, d.sAk);
      offBk[k] = tg_off3(kc + k, d.K, d.sBk);
    }
    __syncthreads();
    const int nkt = (kchunk + TG_BK - 1) / TG_BK;

    auto load_regs = [&](int kt) {
      okm = 0u;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int e = tid + 256 * r;
        int ia, ka, jb, kb;
        if (a_ifast) { ia = e & 63; ka = e >> 6; } else { ka = e & 15; ia = e >> 4; }
        if (b_jfast) { jb = e & 63; kb = e >> 6; } else { kb = e & 15; jb = e >> 4; }
        int kka = kt * TG_BK + ka, kkb = kt * TG_BK + kb;
        int oa = offAi[ia], ob = offBj[jb];
        // unconditional loads at clamped (always valid) addresses; the predicate is applied to the VALUE when it is stored to LDS,
        // behind the MFMAs of the tile before: a load under `?:` became an exec-masked branch with its own s_waitcnt vmcnt(0) at
        // the join -- eight memory latencies in a row per k-tile (rounds 1-5, found in the ISA in round 6)
        va[r] = A[max(oa, 0) + offAk[min(kka, kchunk - 1)]];
        vb[r] = B[max(ob, 0) + offBk[min(kkb, kchunk - 1)]];
        okm |= ((oa >= 0 && kka < kchunk) ? 1u : 0u) << (2 * r) | ((ob >= 0 && kkb < kchunk) ? 2u : 0u) << (2 * r);
      }
    };
    auto store_regs = [&]() {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int e = tid + 256 * r;
        int ia, ka, jb, kb;
        if (a_ifast) { ia = e & 63; ka = e >> 6; } else { ka = e & 15; ia = e >> 4; }
        if (b_jfast) { jb = e & 63; kb = e >> 6; } else { kb = e & 15; jb = e >> 4; }
        TAcc ra = ((okm >> (2 * r)) & 1u) ? TAcc(va[r]) : TAcc(0);
        TAcc rb = ((okm >> (2 * r)) & 2u) ? TAcc(vb[r]) : TAcc(0);
        if constexpr (is_cplx<TAcc>::value) {
          if (d.conjA) ra = conj_of(ra);
          if (d.conjB) rb = conj_of(rb);
        }
        As[ka][ia] = ra;
        Bs[kb][jb] = rb;
      }
    };

    // (a second register set -- the loads of k-tile kt + 2 issued before the MFMAs of kt -- was measured in round 3: the dense
    // M = R Tt went 149 -> 171 ms per step, the f64-accumulating apply 70 -> 94: not adopted)
    load_regs(0);
    store_regs();
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
      if (kt + 1 < nkt) load_regs(kt + 1);
      if constexpr (USE_MFMA && CPLX) {
        // complex128 on the f64 matrix cores: the operands sit interleaved in LDS (one 16-byte read per element), the four real
        // products of (ar + i ai)(br + i bi) are four v_mfma_f64_16x16x4_f64 per tile: Re += ar br - ai bi, Im += ar bi + ai br
#pragma unroll
        for (int kk = 0; kk < TG_BK; kk += 4) {
          const TAcc a0 = As[kk + (lane >> 4)][wm * 32 + (lane & 15)];
          const TAcc a1 = As[kk + (lane >> 4)][wm * 32 + 16 + (lane & 15)];
          const TAcc b0 = Bs[kk + (lane >> 4)][wn * 32 + (lane & 15)];
          const TAcc b1 = Bs[kk + (lane >> 4)][wn * 32 + 16 + (lane & 15)];
          const double ar[2] = {(double)a0.re, (double)a1.re}, ai[2] = {(double)a0.im, (double)a1.im};
          const double br[2] = {(double)b0.re, (double)b1.re}, bi[2] = {(double)b0.im, (double)b1.im};
#pragma unroll
          for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
              acc64[a][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar[a], br[c], acc64[a][c], 0, 0, 0);
              acc64[a][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(-ai[a], bi[c], acc64[a][c], 0, 0, 0);
              accI[a][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar[a], bi[c], accI[a][c], 0, 0, 0);
              accI[a][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(ai[a], br[c], accI[a][c], 0, 0, 0);
            }
        }
      } else if constexpr (USE_MFMA) {
        if constexpr (sizeof(TAcc) == 4) {
#pragma unroll
          for (int kk = 0; kk < TG_BK; kk += 2) {
            float a = As[kk + (lane >> 5)][wm * 32 + (lane & 31)];
            float bb = Bs[kk + (lane >> 5)][wn * 32 + (lane & 31)];
            acc32 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb, acc32, 0, 0, 0);
          }
        } else {
#pragma unroll
          for (int kk = 0; kk < TG_BK; kk += 4) {
            double a0 = As[kk + (lane >> 4)][wm * 32 + (lane & 15)];
            double a1 = As[kk + (lane >> 4)][wm * 32 + 16 + (lane & 15)];
            double b0 = Bs[kk + (lane >> 4)][wn * 32 + (lane & 15)];
            double b1 = Bs[kk + (lane >> 4)][wn * 32 + 16 + (lane & 15)];
            acc64[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc64[0][0], 0, 0, 0);
            acc64[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc64[0][1], 0, 0, 0);
            acc64[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc64[1][0], 0, 0, 0);
            acc64[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc64[1][1], 0, 0, 0);
          }
        }
      } else {
        const int ti = (tid >> 4) * 4, tj = (tid & 15) * 4;
#pragma unroll
        for (int kk = 0; kk < TG_BK; ++kk) {
          TAcc av[4], bv[4];
          for (int a = 0; a < 4; ++a) av[a] = As[kk][ti + a];
          for (int c = 0; c < 4; ++c) bv[c] = Bs[kk][tj + c];
          for (int a = 0; a < 4; ++a)
            for (int c = 0; c < 4; ++c) accv[a][c] += av[a] * bv[c];
        }
      }
      __syncthreads();
      if (kt + 1 < nkt) {
        store_regs();
        __syncthreads();
      }
    }
  }

  const TAcc alpha = TAcc(d.alpha);
  auto put = [&](int li, int lj, TAcc v) {
    int oi = offCi[li], oj = offCj[lj];
    if (oi >= 0 && oj >= 0) {
      TC *p = C + oi + oj;
      v *= alpha;
      if (d.accumulate) v += TAcc(*p);
      *p = TC(v);
    }
  };
  if constexpr (USE_MFMA && CPLX) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int row = (lane >> 4) + 4 * r;
          put(wm * 32 + a * 16 + row, wn * 32 + c * 16 + (lane & 15), TAcc(acc64[a][c][r], accI[a][c][r]));
        }
  } else if constexpr (USE_MFMA) {
    if constexpr (sizeof(TAcc) == 4) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        put(wm * 32 + row, wn * 32 + (lane & 31), acc32[r]);
      }
    } else {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            int row = (lane >> 4) + 4 * r;
            put(wm * 32 + a * 16 + row, wn * 32 + c * 16 + (lane & 15), acc64[a][c][r]);
          }
    }
  } else {
    const int ti = (tid >> 4) * 4, tj = (tid & 15) * 4;
    for (int a = 0; a < 4; ++a)
      for (int c = 0; c < 4; ++c) put(ti + a, tj + c, accv[a][c]);
  }
}

// Skinny products with float64 accumulation (round 4): C = A B where one side is at most 32 wide -- Y = Tt V^T (256 x 32),
// V' = U^T M (32 x 256), the products of the two-level truncation route -- on the f64 matrix cores.  The 64 x 64 tile of
// tgemm_tile gives half of its waves columns (rows) that do not exist; here the block tile is BM x BN = 128 x 32 (four waves
// stacked along I) or 32 x 128 (along J), every wave owns a live 32 x 32 quadrant.  Same descriptor semantics (strides, masks,
// live extents, selectors), real element types, f32 or f64 operands.
template <typename TA, typename TB, typename TC, int BM, int BN>
__device__ __forceinline__ void tgemm_tile_skinny(const TGemmDesc &d, const TA *__restrict__ Ag, const TB *__restrict__ Bg,
                                                  TC *__restrict__ Cg, const int i0, const int Itot, const int Ktot) {
  constexpr int PA = BM + 16, PB = BN + 16;
  __shared__ double As[TG_BK][PA];
  __shared__ double Bs[TG_BK][PB];
  __shared__ int offAi[BM], offBj[BN], offCi[BM], offCj[BN];
  __shared__ int offAk[TG_KTAB], offBk[TG_KTAB];
  const int tid = threadIdx.x, b = blockIdx.z, j0 = blockIdx.y * BN, Jtot = d.Jtot();
  long baseA = (long)(b / d.bdivA) * d.wA, baseB = (long)(b / d.bdivB) * d.wB;
  if (d.selA) baseA += (long)d.selA[(long)(b / d.seldivA) * d.selA_inc] * d.selA_mul;
  if (d.selB) baseB += (long)d.selB[(long)(b / d.seldivB) * d.selB_inc] * d.selB_mul;
  const TA *A = Ag + baseA;
  const TB *B = Bg + baseB;
  TC *C = Cg + (long)(b / d.bdivC) * d.wC;
  for (int t = tid; t < BM + BN; t += 256) {
    if (t < BM) {
      const int i = i0 + t;
      offAi[t] = (i < Itot) ? tg_off3m(i, d.I, d.sAi, d.Imask) : -1;
      offCi[t] = (i < Itot) ? tg_off3(i, d.I, d.sCi) : -1;
    } else {
      const int j = j0 + t - BM;
      offBj[t - BM] = (j < Jtot) ? tg_off3m(j, d.J, d.sBj, d.Jmask) : -1;
      offCj[t - BM] = (j < Jtot) ? tg_off3(j, d.J, d.sCj) : -1;
    }
  }
  __syncthreads();
  const bool a_ifast = d.sAi[2] <= d.sAk[2], b_jfast = d.sBj[2] <= d.sBk[2];
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = BM == 128 ? wave : 0, wn = BM == 128 ? 0 : wave;
  tg_f64x4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[a][c][r] = 0.0;
  constexpr int NA = BM * TG_BK / 256, NB = BN * TG_BK / 256;
  TA va[NA];
  TB vb[NB];
  unsigned okm = 0u;      // bit r: element r of A exists, bit NA + r: of B
  for (int kc = 0; kc < Ktot; kc += TG_KTAB) {
    const int kchunk = min(TG_KTAB, Ktot - kc);
    __syncthreads();
    for (int k = tid; k < kchunk; k += 256) {
      offAk[k] = tg_off3(kc + k, d.K, d.sAk);
      offBk[k] = tg_off3(kc + k, d.K, d.sBk);
    }
    __syncthreads();
    const int nkt = (kchunk + TG_BK - 1) / TG_BK;
    auto load_regs = [&](int kt) {
      okm = 0u;
#pragma unroll
      for (int r = 0; r < NA; ++r) {
        const int e = tid + 256 * r;
        int ia, ka;
        if (a_ifast) { ia = e % BM; ka = e / BM; } else { ka = e & 15; ia = e >> 4; }
        const int kk = kt * TG_BK + ka, oa = offAi[ia];
        va[r] = A[max(oa, 0) + offAk[min(kk, kchunk - 1)]];      // unconditional, clamped; predicate at the LDS store (see tgemm_tile)
        okm |= ((oa >= 0 && kk < kchunk) ? 1u : 0u) << r;
      }
#pragma unroll
      for (int r = 0; r < NB; ++r) {
        const int e = tid + 256 * r;
        int jb, kb;
        if (b_jfast) { jb = e % BN; kb = e / BN; } else { kb = e & 15; jb = e >> 4; }
        const int kk = kt * TG_BK + kb, ob = offBj[jb];
        vb[r] = B[max(ob, 0) + offBk[min(kk, kchunk - 1)]];
        okm |= ((ob >= 0 && kk < kchunk) ? 1u : 0u) << (NA + r);
      }
    };
    auto store_regs = [&]() {
#pragma unroll
      for (int r = 0; r < NA; ++r) {
        const int e = tid + 256 * r;
        int ia, ka;
        if (a_ifast) { ia = e % BM; ka = e / BM; } else { ka = e & 15; ia = e >> 4; }
        As[ka][ia] = ((okm >> r) & 1u) ? (double)va[r] : 0.0;
      }
#pragma unroll
      for (int r = 0; r < NB; ++r) {
        const int e = tid + 256 * r;
        int jb, kb;
        if (b_jfast) { jb = e % BN; kb = e / BN; } else { kb = e & 15; jb = e >> 4; }
        Bs[kb][jb] = ((okm >> (NA + r)) & 1u) ? (double)vb[r] : 0.0;
      }
    };
    load_regs(0);
    store_regs();
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
      if (kt + 1 < nkt) load_regs(kt + 1);
#pragma unroll
      for (int kk = 0; kk < TG_BK; kk += 4) {
        const double a0 = As[kk + (lane >> 4)][wm * 32 + (lane & 15)], a1 = As[kk + (lane >> 4)][wm * 32 + 16 + (lane & 15)];
        const double b0 = Bs[kk + (lane >> 4)][wn * 32 + (lane & 15)], b1 = Bs[kk + (lane >> 4)][wn * 32 + 16 + (lane & 15)];
        acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
      }
      __syncthreads();
      if (kt + 1 < nkt) {
        store_regs();
        __syncthreads();
      }
    }
  }
  const double alpha = d.alpha;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int li = wm * 32 + a * 16 + (lane >> 4) + 4 * r, lj = wn * 32 + c * 16 + (lane & 15);
        const int oi = offCi[li], oj = offCj[lj];
        if (oi >= 0 && oj >= 0) {
          TC *p = C + oi + oj;
          double v = acc[a][c][r] * alpha;
          if (d.accumulate) v += (double)*p;
          *p = TC(v);
        }
      }
}

template <typename TA, typename TB, typename TC, int BM, int BN>
__global__ __launch_bounds__(256) void tgemm_skinny_f64_kernel(TGemmDesc d, const TA *__restrict__ Ag, const TB *__restrict__ Bg,
                                                               TC *__restrict__ Cg) {
  const int b = blockIdx.z;
  if (d.batch_flag && d.batch_flag[b] >= 0) return;
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    if (d.dI[s].p) { const int e = max(0, min(d.I[s], d.dI[s].p[b / d.dI[s].div] * d.dI[s].mul)); if (d.dI[s].mask) d.Imask[s] = e; else d.I[s] = e; }
    if (d.dJ[s].p) { const int e = max(0, min(d.J[s], d.dJ[s].p[b / d.dJ[s].div] * d.dJ[s].mul)); if (d.dJ[s].mask) d.Jmask[s] = e; else d.J[s] = e; }
    if (d.dK[s].p) d.K[s] = max(0, min(d.K[s], d.dK[s].p[b / d.dK[s].div] * d.dK[s].mul));
  }
  if ((int)blockIdx.y * BN >= d.Jtot()) return;
  int Itot = d.Itot(), Ktot = d.Ktot();
  if (d.dynI) Itot = max(0, min(Itot, d.dynI[b] * d.dynI_mul));
  if (d.dynK) Ktot = max(0, min(Ktot, d.dynK[b] * d.dynK_mul));
  if (d.flopc && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && b % d.flop_stride == 0) {
    atomicAdd(d.flopc, 2ull * d.flop_stride * Itot * d.Jtot() * Ktot);
    if (d.bytec)
      atomicAdd(d.bytec, (unsigned long long)d.flop_stride * ((unsigned long long)Itot * Ktot * sizeof(TA) + (unsigned long long)Ktot * d.Jtot() * sizeof(TB) +
                                                              (unsigned long long)Itot * d.Jtot() * sizeof(TC)));
  }
  for (int i0 = blockIdx.x * BM; i0 < Itot; i0 += gridDim.x * BM) {
    tgemm_tile_skinny<TA, TB, TC, BM, BN>(d, Ag, Bg, Cg, i0, Itot, Ktot);
    __syncthreads();
  }
}

// One block walks the I tiles blockIdx.x, blockIdx.x + gridDim.x, ... of its (j tile, batch entry):
// with a per-walker dynamic extent the grid is launched narrow (TG_DYN_GRIDX tiles) so that the
// launch does not consist of tens of thousands of blocks that exit at once.
constexpr int TG_DYN_GRIDX = 2;
template <typename TA, typename TB, typename TC, typename TAcc, bool USE_MFMA>
__global__ __launch_bounds__(256) void tgemm_kernel(TGemmDesc d, const TA *__restrict__ Ag,
                                                    const TB *__restrict__ Bg, TC *__restrict__ Cg) {
  const int b = blockIdx.z;
  if (d.batch_flag && d.batch_flag[b] >= 0) return;
#pragma unroll
  for (int s = 0; s < 3; ++s) {   // block-uniform: the walker's live extents replace / mask the static dims
    if (d.dI[s].p) { const int e = max(0, min(d.I[s], d.dI[s].p[b / d.dI[s].div] * d.dI[s].mul)); if (d.dI[s].mask) d.Imask[s] = e; else d.I[s] = e; }
    if (d.dJ[s].p) { const int e = max(0, min(d.J[s], d.dJ[s].p[b / d.dJ[s].div] * d.dJ[s].mul)); if (d.dJ[s].mask) d.Jmask[s] = e; else d.J[s] = e; }
    if (d.dK[s].p) d.K[s] = max(0, min(d.K[s], d.dK[s].p[b / d.dK[s].div] * d.dK[s].mul));
  }
  if ((int)blockIdx.y * TG_BN >= d.Jtot()) return;
  int Itot = d.Itot(), Ktot = d.Ktot();
  if (d.dynI) Itot = max(0, min(Itot, d.dynI[b] * d.dynI_mul));
  if (d.dynK) Ktot = max(0, min(Ktot, d.dynK[b] * d.dynK_mul));
  if (d.flopc && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && b % d.flop_stride == 0)
  {
    atomicAdd(d.flopc, (unsigned long long)(d.upper_only ? 1 : 2) * d.flop_stride * Itot * d.Jtot() * Ktot);
    if (d.bytec)
      atomicAdd(d.bytec, (unsigned long long)d.flop_stride *
                             ((unsigned long long)Itot * Ktot * sizeof(TA) + (unsigned long long)Ktot * d.Jtot() * sizeof(TB) +
                              (unsigned long long)Itot * d.Jtot() * sizeof(TC) / (d.upper_only ? 2 : 1)));
  }
  for (int i0 = blockIdx.x * TG_BM; i0 < Itot; i0 += gridDim.x * TG_BM) {   // block-uniform trip count
    if (d.upper_only && (int)(blockIdx.y + 1) * TG_BN <= i0) continue;
    tgemm_tile<TA, TB, TC, TAcc, USE_MFMA>(d, Ag, Bg, Cg, i0, Itot, Ktot);
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// Direct variant for the small contractions of the rank-adaptive absorption (live extents of a few
// tens): one WAVE per 32x32 tile of C, operands loaded from global memory straight into the MFMA
// operand registers (lane l holds A[i0 + l%32][k], B[k][j0 + l%32] -- the 32x32x2 operand layout),
// no LDS staging, no workgroup barrier.  A block is four independent waves that walk the tiles of
// the walker's (dynamic) extent.
//
// K is walked as (k0, k1) x rounds of 8 values of the innermost sub-index k2: in a round the lower
// half-wave takes k2 = 8r..8r+3 and the upper half 8r+4..8r+7, four MFMA steps (any pairing of k
// values with MFMA steps is valid as long as A and B agree).  An operand whose k2 is contiguous in
// memory (AVEC / BVEC) fetches its four values as one 16-byte load: the cost of these kernels is
// the number of memory requests, not bytes (each lane addresses its own row).
// f32 in / f32 out.  Same descriptor semantics as tgemm_kernel (dynK is not supported here).
// per-walker live extents replace / mask the static dims of a descriptor (block-uniform)
__device__ __forceinline__ void tg_apply_extents(TGemmDesc &d, const int b) {
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    if (d.dI[s].p) { const int e = max(0, min(d.I[s], d.dI[s].p[b / d.dI[s].div] * d.dI[s].mul)); if (d.dI[s].mask) d.Imask[s] = e; else d.I[s] = e; }
    if (d.dJ[s].p) { const int e = max(0, min(d.J[s], d.dJ[s].p[b / d.dJ[s].div] * d.dJ[s].mul)); if (d.dJ[s].mask) d.Jmask[s] = e; else d.J[s] = e; }
    if (d.dK[s].p) d.K[s] = max(0, min(d.K[s], d.dK[s].p[b / d.dK[s].div] * d.dK[s].mul));
  }
}

// The tile loop of the direct kernel: the block's four waves walk the 32x32 tiles of C = A B for one batch entry.
// d has its live extents applied; A, B, C are the entry's operand bases (generic pointers: an operand may live in LDS,
// which is how tgemm_chain_kernel keeps the intermediate of two chained contractions on chip).  K2s = static extent of
// k2 (vector loads stay inside it).
// quotient and remainder of idx / dv (0 <= idx < 2^22, dv > 0) through the float reciprocal rcp ~ 1 / dv: the estimate is off by
// at most one either way, two selects repair it -- a third of the instructions of the 32-bit integer division sequence
__device__ __forceinline__ int tg_fdivmod(const int idx, const int dv, const float rcp, int &rem) {
  int q = (int)((float)idx * rcp);
  int r = idx - q * dv;
  if (r < 0) { r += dv; --q; }
  if (r >= dv) { r -= dv; ++q; }
  rem = r;
  return q;
}

// Offsets are handled in BYTES as unsigned 32-bit values (an operand of one batch entry is far below 4 GB): a load is
// base pointer (uniform) + 32-bit lane offset, no 64-bit address arithmetic per element.
__device__ __forceinline__ float tg_ldf(const float *__restrict__ base, const unsigned boff) {
  return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + boff);
}
__device__ __forceinline__ float4 tg_ldf4(const float *__restrict__ base, const unsigned boff) {
  return *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(base) + boff);
}

constexpr int TG_ZERO_ROW = 0x40000000;   // flag in the C-row offset table: the row exists in C but its A row reads as zero

template <bool AVEC, bool BVEC>
__device__ __forceinline__ void tg_direct_body(const TGemmDesc &d, const float *__restrict__ A, const float *__restrict__ B,
                                               float *__restrict__ C, const int Itot, const int Jtot, const int K2s,
                                               int (*offCi_s)[32], const int tile0, const int tile_step,
                                               const float scale = 1.f, double *__restrict__ sumsq = nullptr) {
  // scale: multiplies alpha (per-entry scale of an operand that was left unnormalised); sumsq (optional): += squares of the
  // values this lane stores.
  // Instruction budget (SQ counters, round 2: 36 VALU instructions per MFMA in the chained kernel, the launches were
  // bound by VALU issue, not by memory): no integer division per lane (float-reciprocal split of the tile's row / column
  // index), K walked with uniform counters, loads unconditional at clamped addresses (rows and columns that do not exist
  // read row / column 0 and are never stored; k beyond the live extent is zeroed in the last round of a k2 run only),
  // the loads of round r+1 issued before and consumed after the MFMAs of round r.
  // (Round 6 measured a ring of three / four rounds in flight here and in the f64 body -- s_waitcnt vmcnt(10..23) in the loops instead
  // of 0..11 --: direct launches 416 -> 400 ms, chained launches and the headline unchanged (121.1 k): these loops do not wait for one
  // round trip per round; not kept.)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ntj = (Jtot + 31) >> 5, ntiles = ((Itot + 31) >> 5) * ntj;
  const float rntj = __builtin_amdgcn_rcpf((float)ntj);
  const float alpha = (float)d.alpha * scale;
  double ss = 0.0;     // (f64: the norm it feeds replaces an f64 reduction over the stored tensor)
  const int half = lane >> 5, l31 = lane & 31;
  const int K2 = d.K[2], K1 = d.K[1];
  const int nr8 = (K2 + 7) >> 3, nrounds = d.K[0] * K1 * nr8;
  const unsigned sA2b = 4u * d.sAk[2], sB2b = 4u * d.sBk[2];
  const float rI2 = __builtin_amdgcn_rcpf((float)d.I[2]), rI1 = __builtin_amdgcn_rcpf((float)d.I[1]);
  const float rJ2 = __builtin_amdgcn_rcpf((float)d.J[2]), rJ1 = __builtin_amdgcn_rcpf((float)d.J[1]);
  const int kh = 4 * half;
  const bool accumulate = d.accumulate != 0;

  for (int t = tile0 + wave; t < ntiles; t += tile_step) {
    int tj;
    const int ti = tg_fdivmod(t, ntj, rntj, tj);      // (round 6: the integer division was ~35 of a tile's ~400 vector instructions)
    const int i = ti * 32 + l31, j = tj * 32 + l31;
    unsigned oab, obb;    // byte offsets of this lane's A row / B column (0 when it does not exist)
    int ocj;              // element offset of column j in C, -1: not stored; TG_ZERO_ROW set: stored as zero
    {
      int i2, i1, j2, j1;
      const int qi = tg_fdivmod(i, d.I[2], rI2, i2);
      const int i0 = tg_fdivmod(qi, d.I[1], rI1, i1);
      const int qj = tg_fdivmod(j, d.J[2], rJ2, j2);
      const int j0 = tg_fdivmod(qj, d.J[1], rJ1, j1);
      const bool iv = i < Itot, jv = j < Jtot;
      const bool iz = i2 >= d.Imask[2] || i1 >= d.Imask[1] || i0 >= d.Imask[0];
      const bool jz = j2 >= d.Jmask[2] || j1 >= d.Jmask[1] || j0 >= d.Jmask[0];
      // (a row / column beyond a masked live extent is stored as zero whatever was multiplied: it reads row / column 0 too,
      // instead of dragging the dead part of the operand through the memory system)
      oab = (iv && !iz) ? 4u * (unsigned)(i0 * d.sAi[0] + i1 * d.sAi[1] + i2 * d.sAi[2]) : 0u;
      obb = (jv && !jz) ? 4u * (unsigned)(j0 * d.sBj[0] + j1 * d.sBj[1] + j2 * d.sBj[2]) : 0u;
      const int oci = i0 * d.sCi[0] + i1 * d.sCi[1] + i2 * d.sCi[2];
      if (half == 0) offCi_s[wave][l31] = iv ? (oci | (iz ? TG_ZERO_ROW : 0)) : -1;
      ocj = jv ? ((j0 * d.sCj[0] + j1 * d.sCj[1] + j2 * d.sCj[2]) | (jz ? TG_ZERO_ROW : 0)) : -1;
    }
    tg_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    int k0 = 0, k1 = 0, r8 = 0;          // uniform position of the round being loaded: (k0, k1), k2 = 8 r8 + 4 half + (0..3)
    unsigned kab = 0, kbb = 0;           // byte offsets of (k0, k1) in A and B
    auto advance = [&]() {
      if (++r8 == nr8) {
        r8 = 0;
        if (++k1 == K1) { k1 = 0; ++k0; }
        kab = 4u * (unsigned)(k0 * d.sAk[0] + k1 * d.sAk[1]);
        kbb = 4u * (unsigned)(k0 * d.sBk[0] + k1 * d.sBk[1]);
      }
    };
    auto load_raw = [&](float (&av)[4], float (&bv)[4]) {
      const int kq = 8 * r8 + kh;
      if constexpr (AVEC) {
        const float4 v = tg_ldf4(A, oab + kab + 4u * (unsigned)min(kq, K2s - 4));
        av[0] = v.x; av[1] = v.y; av[2] = v.z; av[3] = v.w;
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) av[q] = tg_ldf(A, oab + kab + (unsigned)min(kq + q, K2 - 1) * sA2b);
      }
      if constexpr (BVEC) {
        const float4 v = tg_ldf4(B, obb + kbb + 4u * (unsigned)min(kq, K2s - 4));
        bv[0] = v.x; bv[1] = v.y; bv[2] = v.z; bv[3] = v.w;
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) bv[q] = tg_ldf(B, obb + kbb + (unsigned)min(kq + q, K2 - 1) * sB2b);
      }
    };
    auto mask_k = [&](const int r8m, float (&av)[4], float (&bv)[4]) {   // only the last round of a k2 run can be partial
      if (8 * r8m + 8 > K2) {
        const int kq = 8 * r8m + kh;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const bool ok = kq + q < K2;
          av[q] = ok ? av[q] : 0.f;
          bv[q] = ok ? bv[q] : 0.f;
        }
      }
    };
    float a0[4], b0[4], a1[4], b1[4];
    auto mfma4 = [&](const float (&av)[4], const float (&bv)[4]) {
#pragma unroll
      for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bv[q], acc, 0, 0, 0);
    };
    if (nrounds > 0) {
      load_raw(a0, b0);
      mask_k(0, a0, b0);
      int rd = 0;
      // steady state without a conditional around the loads (the wait counters stay exact: the loads of the next round
      // are in flight while the MFMAs of this one issue)
      for (; rd + 2 < nrounds; rd += 2) {
        advance();
        const int r8b = r8;
        load_raw(a1, b1);
        mfma4(a0, b0);
        mask_k(r8b, a1, b1);
        advance();
        const int r8a = r8;
        load_raw(a0, b0);
        mfma4(a1, b1);
        mask_k(r8a, a0, b0);
      }
      if (rd + 1 < nrounds) {     // two rounds left
        advance();
        load_raw(a1, b1);
        mfma4(a0, b0);
        mask_k(r8, a1, b1);
        mfma4(a1, b1);
      } else {
        mfma4(a0, b0);
      }
    }
    // accumulator r of this lane = row 8 (r / 4) + 4 half + (r % 4), column l31 of the tile
    const bool jzero = (ocj & TG_ZERO_ROW) != 0;
    const int ocj_e = ocj & ~TG_ZERO_ROW;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int4 o4 = *reinterpret_cast<const int4 *>(&offCi_s[wave][8 * g + kh]);
      const int oi4[4] = {o4.x, o4.y, o4.z, o4.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int oi = oi4[e];
        if (oi >= 0 && ocj >= 0) {
          float *p = C + ((oi & ~TG_ZERO_ROW) + ocj_e);
          float v = ((oi & TG_ZERO_ROW) || jzero) ? 0.f : acc[4 * g + e] * alpha;
          if (accumulate) v += *p;
          *p = v;
          if (sumsq) ss = fma((double)v, (double)v, ss);
        }
      }
    }
  }
  if (sumsq) *sumsq += ss;
}

// ---------------------------------------------------------------------------------------------
// The same tile loop with FLOAT64 ACCUMULATION on the f64 matrix cores (round 5): f32 operands (global memory or LDS) are
// converted on the way into v_mfma_f64_16x16x4_f64, the f32 result is rounded once at the store.  For the contractions of the
// truncation pass whose f32 accumulation shows in the amplitude of a dense state (Z1 = A Y, Tt = W Z1 and Y = Tt V^T --
// columns of small sigma are differences of O(sigma_1) terms; draining f32 chains of 8 products into float64 registers removes a
// third of it only, the cancellation sits inside the 8-term sums too).  A wave owns the same 32 x 32 tile of C as 2 x 2 quadrants
// of 16 x 16 (lane l: rows / columns l % 16 and 16 + l % 16 of the tile, k group l / 16); a round is 8 values of k2 -- lane group
// g takes k2 = 8 r + 2 g + (0, 1), two MFMA steps of four quadrants each -- so a k2 run of 8 (the PEPS bond) wastes nothing.
// Half the matrix rate of the f32 body; used where the flops are few (the backward pair is an eighth of the forward pair).
template <bool AVEC, bool BVEC>
__device__ __forceinline__ void tg_direct_body_f64(const TGemmDesc &d, const float *__restrict__ A, const float *__restrict__ B,
                                                   float *__restrict__ C, const int Itot, const int Jtot, const int K2s,
                                                   int (*offCi_s)[32], const int tile0, const int tile_step,
                                                   const float scale = 1.f, double *__restrict__ sumsq = nullptr) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ntj = (Jtot + 31) >> 5, ntiles = ((Itot + 31) >> 5) * ntj;
  const float rntj = __builtin_amdgcn_rcpf((float)ntj);
  const double alpha = d.alpha * (double)scale;
  double ss = 0.0;
  const int g4 = lane >> 4, c16 = lane & 15;
  const int K2 = d.K[2], K1 = d.K[1];
  const int nr8 = (K2 + 7) >> 3, nrounds = d.K[0] * K1 * nr8;
  const unsigned sA2b = 4u * d.sAk[2], sB2b = 4u * d.sBk[2];
  const float rI2 = __builtin_amdgcn_rcpf((float)d.I[2]), rI1 = __builtin_amdgcn_rcpf((float)d.I[1]);
  const float rJ2 = __builtin_amdgcn_rcpf((float)d.J[2]), rJ1 = __builtin_amdgcn_rcpf((float)d.J[1]);
  const int kh = 2 * g4;
  const bool accumulate = d.accumulate != 0;

  for (int t = tile0 + wave; t < ntiles; t += tile_step) {
    int tj;
    const int ti = tg_fdivmod(t, ntj, rntj, tj);      // (round 6: the integer division was ~35 of a tile's ~400 vector instructions)
    unsigned oab[2], obb[2];
    int ocj[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int i = ti * 32 + 16 * q + c16, j = tj * 32 + 16 * q + c16;
      int i2, i1, j2, j1;
      const int qi = tg_fdivmod(i, d.I[2], rI2, i2);
      const int i0 = tg_fdivmod(qi, d.I[1], rI1, i1);
      const int qj = tg_fdivmod(j, d.J[2], rJ2, j2);
      const int j0 = tg_fdivmod(qj, d.J[1], rJ1, j1);
      const bool iv = i < Itot, jv = j < Jtot;
      const bool iz = i2 >= d.Imask[2] || i1 >= d.Imask[1] || i0 >= d.Imask[0];
      const bool jz = j2 >= d.Jmask[2] || j1 >= d.Jmask[1] || j0 >= d.Jmask[0];
      oab[q] = (iv && !iz) ? 4u * (unsigned)(i0 * d.sAi[0] + i1 * d.sAi[1] + i2 * d.sAi[2]) : 0u;
      obb[q] = (jv && !jz) ? 4u * (unsigned)(j0 * d.sBj[0] + j1 * d.sBj[1] + j2 * d.sBj[2]) : 0u;
      const int oci = i0 * d.sCi[0] + i1 * d.sCi[1] + i2 * d.sCi[2];
      if (g4 == q) offCi_s[wave][16 * q + c16] = iv ? (oci | (iz ? TG_ZERO_ROW : 0)) : -1;
      ocj[q] = jv ? ((j0 * d.sCj[0] + j1 * d.sCj[1] + j2 * d.sCj[2]) | (jz ? TG_ZERO_ROW : 0)) : -1;
    }
    tg_f64x4 acc[2][2];
#pragma unroll
    for (int qa = 0; qa < 2; ++qa)
#pragma unroll
      for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[qa][qb][r] = 0.0;

    int k0 = 0, k1 = 0, r8 = 0;
    unsigned kab = 0, kbb = 0;
    auto advance = [&]() {
      if (++r8 == nr8) {
        r8 = 0;
        if (++k1 == K1) { k1 = 0; ++k0; }
        kab = 4u * (unsigned)(k0 * d.sAk[0] + k1 * d.sAk[1]);
        kbb = 4u * (unsigned)(k0 * d.sBk[0] + k1 * d.sBk[1]);
      }
    };
    // av[q][e]: row quadrant q, k2 = 8 r8 + kh + e
    auto load_raw = [&](float (&av)[2][2], float (&bv)[2][2]) {
      const int kq = 8 * r8 + kh;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        if constexpr (AVEC) {
          const float2 v = *reinterpret_cast<const float2 *>(reinterpret_cast<const char *>(A) + (oab[q] + kab + 4u * (unsigned)min(kq, K2s - 2)));
          av[q][0] = v.x; av[q][1] = v.y;
        } else {
#pragma unroll
          for (int e = 0; e < 2; ++e) av[q][e] = tg_ldf(A, oab[q] + kab + (unsigned)min(kq + e, K2 - 1) * sA2b);
        }
        if constexpr (BVEC) {
          const float2 v = *reinterpret_cast<const float2 *>(reinterpret_cast<const char *>(B) + (obb[q] + kbb + 4u * (unsigned)min(kq, K2s - 2)));
          bv[q][0] = v.x; bv[q][1] = v.y;
        } else {
#pragma unroll
          for (int e = 0; e < 2; ++e) bv[q][e] = tg_ldf(B, obb[q] + kbb + (unsigned)min(kq + e, K2 - 1) * sB2b);
        }
      }
    };
    auto mask_k = [&](const int r8m, float (&av)[2][2], float (&bv)[2][2]) {   // only the last round of a k2 run can be partial
      if (8 * r8m + 8 > K2) {
        const int kq = 8 * r8m + kh;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const bool ok = kq + e < K2;
#pragma unroll
          for (int q = 0; q < 2; ++q) { av[q][e] = ok ? av[q][e] : 0.f; bv[q][e] = ok ? bv[q][e] : 0.f; }
        }
      }
    };
    auto mfma8 = [&](const float (&av)[2][2], const float (&bv)[2][2]) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const double a0 = (double)av[0][e], a1 = (double)av[1][e], b0 = (double)bv[0][e], b1 = (double)bv[1][e];
        acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
      }
    };
    float a0[2][2], b0[2][2], a1[2][2], b1[2][2];
    if (nrounds > 0) {
      load_raw(a0, b0);
      mask_k(0, a0, b0);
      int rd = 0;
      for (; rd + 2 < nrounds; rd += 2) {
        advance();
        const int r8b = r8;
        load_raw(a1, b1);
        mfma8(a0, b0);
        mask_k(r8b, a1, b1);
        advance();
        const int r8a = r8;
        load_raw(a0, b0);
        mfma8(a1, b1);
        mask_k(r8a, a0, b0);
      }
      if (rd + 1 < nrounds) {
        advance();
        load_raw(a1, b1);
        mfma8(a0, b0);
        mask_k(r8, a1, b1);
        mfma8(a1, b1);
      } else {
        mfma8(a0, b0);
      }
    }
    // accumulator r of quadrant (qa, qb) = row 16 qa + g4 + 4 r, column 16 qb + c16 of the tile
#pragma unroll
    for (int qa = 0; qa < 2; ++qa)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int oi = offCi_s[wave][16 * qa + g4 + 4 * r];
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
          if (oi >= 0 && ocj[qb] >= 0) {
            float *p = C + ((oi & ~TG_ZERO_ROW) + (ocj[qb] & ~TG_ZERO_ROW));
            float v = ((oi & TG_ZERO_ROW) || (ocj[qb] & TG_ZERO_ROW)) ? 0.f : (float)(acc[qa][qb][r] * alpha);
            if (accumulate) v += *p;
            *p = v;
            if (sumsq) ss = fma((double)v, (double)v, ss);
          }
        }
      }
  }
  if (sumsq) *sumsq += ss;
}

template <bool AVEC, bool BVEC, bool ACC64 = false>
__global__ __launch_bounds__(256, ACC64 ? 4 : 6) void tgemm_direct_kernel(TGemmDesc d, const float *__restrict__ Ag,
                                                           const float *__restrict__ Bg, float *__restrict__ Cg) {
  __shared__ __attribute__((aligned(16))) int offCi_s[4][32];
  const int b = blockIdx.z;
  if (d.batch_flag && d.batch_flag[b] >= 0) return;
  const int K2s = d.K[2];                     // static extent of k2 (vector loads stay inside it)
  tg_apply_extents(d, b);
  int Itot = d.Itot();
  const int Jtot = d.Jtot();
  if (d.dynI) Itot = max(0, min(Itot, d.dynI[b] * d.dynI_mul));
  if (Itot <= 0 || Jtot <= 0) {
    if (d.scale_out && threadIdx.x == 0) { d.scale_out[b] = 1.f; if (d.norm_flag) d.norm_flag[b] = 1; }   // nothing stored: zero norm
    return;
  }
  if (d.flopc && threadIdx.x == 0 && blockIdx.x == 0 && b % d.flop_stride == 0)
  {
    atomicAdd(d.flopc, 2ull * d.flop_stride * Itot * Jtot * d.Ktot());
    if (d.bytec) atomicAdd(d.bytec, 4ull * d.flop_stride * ((unsigned long long)Itot * d.Ktot() + (unsigned long long)d.Ktot() * Jtot + (unsigned long long)Itot * Jtot));
  }
  long baseA = (long)(b / d.bdivA) * d.wA, baseB = (long)(b / d.bdivB) * d.wB;
  if (d.selA) baseA += (long)d.selA[(long)(b / d.seldivA) * d.selA_inc] * d.selA_mul;
  if (d.selB) baseB += (long)d.selB[(long)(b / d.seldivB) * d.selB_inc] * d.selB_mul;
  double ss = 0.0;
  if constexpr (ACC64)
    tg_direct_body_f64<AVEC, BVEC>(d, Ag + baseA, Bg + baseB, Cg + (long)(b / d.bdivC) * d.wC, Itot, Jtot, K2s, offCi_s,
                                   blockIdx.x * 4, gridDim.x * 4, d.scale_in ? d.scale_in[b] : 1.f, d.scale_out ? &ss : nullptr);
  else
    tg_direct_body<AVEC, BVEC>(d, Ag + baseA, Bg + baseB, Cg + (long)(b / d.bdivC) * d.wC, Itot, Jtot, K2s, offCi_s,
                               blockIdx.x * 4, gridDim.x * 4, d.scale_in ? d.scale_in[b] : 1.f, d.scale_out ? &ss : nullptr);
  if (d.scale_out) {     // (gridDim.x == 1: this block stored all of C[b])
    __shared__ double s_nred[4];
    double a = ss;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
    if ((threadIdx.x & 63) == 0) s_nred[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
      const double nrm = sqrt(s_nred[0] + s_nred[1] + s_nred[2] + s_nred[3]);
      if (!(nrm > 0.0) || !isfinite(nrm)) {
        d.scale_out[b] = 1.f;
        if (d.norm_flag) d.norm_flag[b] = 1;
      } else {
        // the log-scale takes the scale AS APPLIED (rounded to f32), not log(nrm): the difference, up to 2^-24 per site, is the same
        // at every repetition of a tensor and adds up coherently over the ~10^3 normalisations of one amplitude (round 5: the fused
        // form of Y sat 8e-6 off on the tiled state with it)
        const float sf = (float)(1.0 / nrm);
        d.scale_out[b] = sf;
        if (d.norm_log) d.norm_log[b] -= log((double)sf);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Two chained contractions with the intermediate resident in LDS:  C1 = A1 B1,  C2 = A2 C1'
// (C1' = C1 read with its sub-indices regrouped into the K and J of the second contraction).  One block per batch
// entry; the live extents of both descriptors are applied first, C1 is laid out compactly (row-major over its live
// I1 and J1 sub-indices) in the block's LDS, written by the tile loop of stage 1 and read as the B operand of stage 2.
// mapK / mapJ name, for every K / J sub-index of the second contraction, the sub-index of C1 it runs over:
// 0..2 = I1[s], 3..5 = J1[s], -1 = unused (extent 1).  An entry whose live C1 exceeds the LDS buffer writes
// flag[b] = -1 and returns (the caller runs the two contractions separately for those entries), else flag[b] = 0.
constexpr int TG_CHAIN_LDS_FLOATS = 6144;    // 24 KB: six blocks per CU (measured: 10240 / 3 blocks 437 ms, 8192 / 4 387, 7168 / 5 363, 6144 / 6 350, 4864 / 8 352 per two steps)
// (a second chain pass with a 48 KB buffer for the declined entries was measured slower than the two separate launches
// they fall back to: 416 ms vs 350 ms -- three blocks per CU hide too little latency)
struct TGemmChainMap { int mapK[3] = {-1, -1, -1}, mapJ[3] = {-1, -1, -1}; };

// F64: both stages accumulate in float64 on the f64 matrix cores (tg_direct_body_f64; the intermediate stays f32 in LDS)
// (two measured-and-rejected bodies of round 5 -- 16 x 16 x 4 tiles with dead quadrants skipped, two J tiles per wave in stage 2 -- were
// removed in round 6; HISTORY.md has their numbers)
// TRI: the triangular-carry form of the chunk loop (round 6; its own instantiation: the per-chunk bookkeeping cost the low-rank headline 2.6 % of this kernel)
template <bool AVEC1, bool BVEC1, bool AVEC2, int LDSF, int MINB, bool F64 = false, bool TRI = false>
__global__ __launch_bounds__(256, MINB) void tgemm_chain_kernel(TGemmDesc d1, TGemmDesc d2, TGemmChainMap mp, const float *__restrict__ A1g,
                                                            const float *__restrict__ B1g, const float *__restrict__ A2g,
                                                            float *__restrict__ C2g, int *__restrict__ flag, int only_flagged, int allow_chunks) {
  __shared__ __attribute__((aligned(16))) int offCi_s[4][32];
  __shared__ float s_mid[LDSF];
  const int b = blockIdx.x;
  if (only_flagged && flag[b] >= 0) return;     // second launch (larger buffer, fewer blocks per CU): declined entries only
  const int K2s1 = d1.K[2], K2s2 = d2.K[2];
  tg_apply_extents(d1, b);
  tg_apply_extents(d2, b);
  int I1 = d1.Itot();
  const int J1 = d1.Jtot();
  if (d1.dynI) I1 = max(0, min(I1, d1.dynI[b] * d1.dynI_mul));
  int I2 = d2.Itot();
  const int J2 = d2.Jtot();
  if (d2.dynI) I2 = max(0, min(I2, d2.dynI[b] * d2.dynI_mul));
  // compact LDS layout of C1 over the live dims of (I1 sub-indices, J1 sub-indices); a flattened dynI limit of stage 1
  // only leaves the rows beyond it unwritten (they are not read: stage 2 runs over the same live extents)
  int lds_stride[6];
  int chunk = d1.I[1], jsub = -1;   // C1 is produced chunk values of I1[1] at a time (all of them when it fits the buffer)
  {
    int st = 1;
    for (int s = 2; s >= 0; --s) { lds_stride[3 + s] = st; st *= d1.J[s]; }
    for (int s = 2; s >= 0; --s) { lds_stride[s] = st; st *= d1.I[s]; }
    if (st > LDSF) {
      // Larger live extents (states of higher rank): walk the middle I sub-index of stage 1 (the carry row m / the bond a)
      // in chunks whose slice of C1 fits -- it is a J sub-index of stage 2, so every chunk is a complete pair of
      // contractions on a slice of the result, and C1 still never leaves the chip.
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        if (mp.mapJ[s] == 1) jsub = (jsub < 0 && d2.Jmask[s] == 0x7fffffff) ? s : 3;
        if (mp.mapK[s] == 1) jsub = 3;
      }
      const int per = lds_stride[1];    // floats of C1 per value of I1[1]
      if (!(allow_chunks & 1) || d1.I[0] != 1 || d1.dynI || d2.dynI || jsub < 0 || jsub > 2 || per > LDSF || d1.Imask[1] != 0x7fffffff) {
        if (threadIdx.x == 0) flag[b] = -1;
        return;
      }
      chunk = LDSF / per;
    }
  }
  if (threadIdx.x == 0) flag[b] = 0;
  if (I1 <= 0 || J1 <= 0 || I2 <= 0 || J2 <= 0) return;
  const bool tri_req = TRI && (allow_chunks & 2) && chunk < d1.I[1];     // (executed flops of the triangular form: counted chunk by chunk below)
  if (d1.flopc && threadIdx.x == 0 && b % d1.flop_stride == 0) {
    if (!tri_req) atomicAdd(d1.flopc, 2ull * d1.flop_stride * ((unsigned long long)I1 * J1 * d1.Ktot() + (unsigned long long)I2 * J2 * d2.Ktot()));
    if (d1.bytec)
      atomicAdd(d1.bytec, 4ull * d1.flop_stride * ((unsigned long long)I1 * d1.Ktot() + (unsigned long long)d1.Ktot() * J1 +
                                                    (unsigned long long)I2 * d2.Ktot() + (unsigned long long)I2 * J2));
  }
#pragma unroll
  for (int s = 0; s < 3; ++s) { d1.sCi[s] = lds_stride[s]; d1.sCj[s] = lds_stride[3 + s]; }
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    d2.sBk[s] = mp.mapK[s] >= 0 ? lds_stride[mp.mapK[s]] : 0;
    d2.sBj[s] = mp.mapJ[s] >= 0 ? lds_stride[mp.mapJ[s]] : 0;
  }
  long baseA1 = (long)b * d1.wA, baseB1 = (long)b * d1.wB, baseA2 = (long)b * d2.wA;
  if (d1.selA) baseA1 += (long)d1.selA[(long)(b / d1.seldivA) * d1.selA_inc] * d1.selA_mul;
  if (d1.selB) baseB1 += (long)d1.selB[(long)(b / d1.seldivB) * d1.selB_inc] * d1.selB_mul;
  if (d2.selA) baseA2 += (long)d2.selA[(long)(b / d2.seldivA) * d2.selA_inc] * d2.selA_mul;
  d1.accumulate = 0;
  const float in_scale = d1.scale_in ? d1.scale_in[b] : 1.f;   // an operand of stage 1 was left unnormalised by its producer
  if (chunk >= d1.I[1]) {
    if constexpr (F64) {
      tg_direct_body_f64<AVEC1, BVEC1>(d1, A1g + baseA1, B1g + baseB1, s_mid, I1, J1, K2s1, offCi_s, 0, 4, in_scale);
      __syncthreads();
      tg_direct_body_f64<AVEC2, false>(d2, A2g + baseA2, s_mid, C2g + (long)b * d2.wC, I2, J2, K2s2, offCi_s, 0, 4);
    } else {
      tg_direct_body<AVEC1, BVEC1>(d1, A1g + baseA1, B1g + baseB1, s_mid, I1, J1, K2s1, offCi_s, 0, 4, in_scale);
      __syncthreads();
      tg_direct_body<AVEC2, false>(d2, A2g + baseA2, s_mid, C2g + (long)b * d2.wC, I2, J2, K2s2, offCi_s, 0, 4);
    }
    return;
  }
  const int n1 = d1.I[1];
  const int sA1 = d1.sAi[1], sC2 = jsub == 0 ? d2.sCj[0] : jsub == 1 ? d2.sCj[1] : d2.sCj[2];
  // Triangular A operand of stage 1 (round 6, allow_chunks & 2): the carry R[m][l][a] of a dense walker is a Cholesky factor with its
  // rows compacted -- row m is zero in the columns (l, a) before m.  A chunk starts at row c0, so its l-blocks before c0 / a_static hold
  // zeros only: stage 1 skips them (the A operand starts at l0) and stage 2, whose contracted index runs over (l, p), starts there too.
  // On 220 live rows of 256 columns that is 37 % of the MFMAs of both stages.
  int ksub = -1;
#pragma unroll
  for (int s = 0; s < 3; ++s) if (mp.mapK[s] == 2) ksub = s;
  const int L1 = d1.I[2], sAl = d1.sAi[2];
  const int sA2l = ksub == 0 ? d2.sAk[0] : ksub == 1 ? d2.sAk[1] : d2.sAk[2];
  const bool tri = TRI && (allow_chunks & 2) && ksub >= 0 && sAl > 0 && d1.Imask[2] == 0x7fffffff;
  unsigned long long fl_exec = 0;
  for (int c0 = 0; c0 < n1; c0 += chunk) {
    const int cn = min(chunk, n1 - c0);
    d1.I[1] = cn;
#pragma unroll
    for (int s = 0; s < 3; ++s)      // (no dynamic indexing: the descriptors stay in registers)
      if (s == jsub) d2.J[s] = cn;
    long oA1 = (long)c0 * sA1, oA2 = 0;
    if constexpr (TRI) {
      const int l0 = tri ? min(c0 / sAl, L1 - 1) : 0;
      d1.I[2] = L1 - l0;
#pragma unroll
      for (int s = 0; s < 3; ++s)
        if (tri && s == ksub) d2.K[s] = L1 - l0;
      oA1 += (long)l0 * sAl;
      oA2 = (long)l0 * sA2l;
      if (tri_req) fl_exec += 2ull * ((unsigned long long)d1.Itot() * J1 * d1.Ktot() + (unsigned long long)I2 * d2.Jtot() * d2.Ktot());
    }
    if (c0) __syncthreads();     // stage 2 of the chunk before has read the buffer
    if constexpr (F64) {
      tg_direct_body_f64<AVEC1, BVEC1>(d1, A1g + baseA1 + oA1, B1g + baseB1, s_mid, d1.Itot(), J1, K2s1, offCi_s, 0, 4, in_scale);
      __syncthreads();
      tg_direct_body_f64<AVEC2, false>(d2, A2g + baseA2 + oA2, s_mid, C2g + (long)b * d2.wC + (long)c0 * sC2, I2, d2.Jtot(), K2s2, offCi_s, 0, 4);
    } else {
      tg_direct_body<AVEC1, BVEC1>(d1, A1g + baseA1 + oA1, B1g + baseB1, s_mid, d1.Itot(), J1, K2s1, offCi_s, 0, 4, in_scale);
      __syncthreads();
      tg_direct_body<AVEC2, false>(d2, A2g + baseA2 + oA2, s_mid, C2g + (long)b * d2.wC + (long)c0 * sC2, I2, d2.Jtot(), K2s2, offCi_s, 0, 4);
    }
  }
  if (tri_req && d1.flopc && threadIdx.x == 0 && b % d1.flop_stride == 0) atomicAdd(d1.flopc, fl_exec * d1.flop_stride);
}

// ---------------------------------------------------------------------------------------------
// THREE chained contractions, both intermediates resident in LDS (round 4: one BTen growth step = one launch):
//     C1 = A1 B1,   C2 = A2 C1'  (C1' = C1 regrouped as the B operand of stage 2, as above),   C3 = C2' B3
// (C2' = C2 regrouped as the A operand of stage 3: mp3.mapI / mapK name, for every I / K sub-index of stage 3, the sub-index of C2
// it runs over, 0..2 = I2[s], 3..5 = J2[s]).  The middle I sub-index of stage 1 (I1[1]) must be a J sub-index of stage 2 AND an I
// sub-index of stage 3 (mp3.chunkI): when the live intermediates exceed the two buffers it is walked in chunks, every chunk a
// complete triple of contractions on a slice of the result.  Rows of C3 beyond the live extent of that sub-index are stored as
// zeros when d3 masks it (the new environment tensor is written in full).  Entries that cannot be chunked write flag[b] = -1.
struct TGemmChain3Map { int mapI[3] = {-1, -1, -1}, mapK[3] = {-1, -1, -1}; int chunkI = -1; };

template <bool AVEC1, bool BVEC1, bool AVEC2, int LDSF, int MINB>
__global__ __launch_bounds__(256, MINB) void tgemm_chain3_kernel(TGemmDesc d1, TGemmDesc d2, TGemmDesc d3, TGemmChainMap mp, TGemmChain3Map mp3,
                                                             const float *__restrict__ A1g, const float *__restrict__ B1g,
                                                             const float *__restrict__ A2g, const float *__restrict__ B3g,
                                                             float *__restrict__ C3g, int *__restrict__ flag,
                                                             const int *__restrict__ skip) {
  __shared__ __attribute__((aligned(16))) int offCi_s[4][32];
  __shared__ float s_mid1[LDSF];
  __shared__ float s_mid2[LDSF];
  const int b = blockIdx.x;
  if (skip && skip[b]) return;        // the caller does not need this entry (its C3 stays undefined)
  const int K2s1 = d1.K[2], K2s2 = d2.K[2], K2s3 = d3.K[2];
  const int x_static = mp3.chunkI == 0 ? d3.I[0] : mp3.chunkI == 1 ? d3.I[1] : d3.I[2];
  tg_apply_extents(d1, b);
  tg_apply_extents(d2, b);
  tg_apply_extents(d3, b);
  const int I1 = d1.Itot(), J1 = d1.Jtot(), I2 = d2.Itot(), J2 = d2.Jtot(), J3 = d3.Jtot();
  int jsub = -1;
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    if (mp.mapJ[s] == 1) jsub = (jsub < 0 && d2.Jmask[s] == 0x7fffffff) ? s : 3;
    if (mp.mapK[s] == 1) jsub = 3;
  }
  int lds1[6], lds2[6];
  int st1 = 1, st2 = 1;
  for (int s = 2; s >= 0; --s) { lds1[3 + s] = st1; st1 *= d1.J[s]; }
  for (int s = 2; s >= 0; --s) { lds1[s] = st1; st1 *= d1.I[s]; }
  for (int s = 2; s >= 0; --s) { lds2[3 + s] = st2; st2 *= d2.J[s]; }
  for (int s = 2; s >= 0; --s) { lds2[s] = st2; st2 *= d2.I[s]; }
  const int n1 = d1.I[1];                                   // live extent of the chunked sub-index
  const int per1 = lds1[1], per2 = n1 > 0 ? st2 / n1 : 0;   // floats of C1 / C2 per value of it
  if (jsub < 0 || jsub > 2 || mp3.chunkI < 0 || d1.I[0] != 1 || d1.dynI || d2.dynI || d3.dynI || d1.Imask[1] != 0x7fffffff ||
      per1 > LDSF || per2 > LDSF || I1 <= 0 || J1 <= 0 || I2 <= 0 || J2 <= 0 || J3 <= 0 || d3.Ktot() <= 0) {
    if (threadIdx.x == 0) flag[b] = -1;
    return;
  }
  if (threadIdx.x == 0) flag[b] = 0;
  const int chunk = min(n1, min(LDSF / per1, LDSF / per2));
  const int sC3 = mp3.chunkI == 0 ? d3.sCi[0] : mp3.chunkI == 1 ? d3.sCi[1] : d3.sCi[2];
  if (d1.flopc && threadIdx.x == 0 && b % d1.flop_stride == 0) {
    const unsigned long long i3 = (unsigned long long)(d3.Itot() / max(1, x_static)) * n1;
    atomicAdd(d1.flopc, 2ull * d1.flop_stride * ((unsigned long long)I1 * J1 * d1.Ktot() + (unsigned long long)I2 * J2 * d2.Ktot() + i3 * J3 * d3.Ktot()));
    if (d1.bytec)
      atomicAdd(d1.bytec, 4ull * d1.flop_stride * ((unsigned long long)I1 * d1.Ktot() + (unsigned long long)d1.Ktot() * J1 +
                                                    (unsigned long long)I2 * d2.Ktot() + (unsigned long long)d3.Ktot() * J3 + i3 * J3));
  }
  long baseA1 = (long)b * d1.wA, baseB1 = (long)b * d1.wB, baseA2 = (long)b * d2.wA, baseB3 = (long)(b / d3.bdivB) * d3.wB;
  if (d1.selA) baseA1 += (long)d1.selA[(long)(b / d1.seldivA) * d1.selA_inc] * d1.selA_mul;
  if (d1.selB) baseB1 += (long)d1.selB[(long)(b / d1.seldivB) * d1.selB_inc] * d1.selB_mul;
  if (d2.selA) baseA2 += (long)d2.selA[(long)(b / d2.seldivA) * d2.selA_inc] * d2.selA_mul;
  d1.accumulate = 0; d2.accumulate = 0;
  const float in_scale = d1.scale_in ? d1.scale_in[b] : 1.f;
  float *C3 = C3g + (long)b * d3.wC;
  const int sA1 = d1.sAi[1];
  for (int c0 = 0; c0 < n1; c0 += chunk) {
    const int cn = min(chunk, n1 - c0);
    d1.I[1] = cn;
#pragma unroll
    for (int s = 0; s < 3; ++s) {      // (no dynamic indexing: the descriptors stay in registers)
      if (s == jsub) d2.J[s] = cn;
      if (s == mp3.chunkI) { d3.I[s] = cn; d3.Imask[s] = 0x7fffffff; }
    }
    // compact layouts of the two intermediates over the live extents of this chunk
    {
      int st = 1;
      for (int s = 2; s >= 0; --s) { lds1[3 + s] = st; st *= d1.J[s]; }
      for (int s = 2; s >= 0; --s) { lds1[s] = st; st *= d1.I[s]; }
      st = 1;
      for (int s = 2; s >= 0; --s) { lds2[3 + s] = st; st *= d2.J[s]; }
      for (int s = 2; s >= 0; --s) { lds2[s] = st; st *= d2.I[s]; }
    }
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      d1.sCi[s] = lds1[s]; d1.sCj[s] = lds1[3 + s];
      d2.sBk[s] = mp.mapK[s] >= 0 ? lds1[mp.mapK[s]] : 0;
      d2.sBj[s] = mp.mapJ[s] >= 0 ? lds1[mp.mapJ[s]] : 0;
      d2.sCi[s] = lds2[s]; d2.sCj[s] = lds2[3 + s];
      d3.sAi[s] = mp3.mapI[s] >= 0 ? lds2[mp3.mapI[s]] : 0;
      d3.sAk[s] = mp3.mapK[s] >= 0 ? lds2[mp3.mapK[s]] : 0;
    }
    if (c0) __syncthreads();     // stage 3 of the chunk before has read the second buffer
    tg_direct_body<AVEC1, BVEC1>(d1, A1g + baseA1 + (long)c0 * sA1, B1g + baseB1, s_mid1, d1.Itot(), J1, K2s1, offCi_s, 0, 4, in_scale);
    __syncthreads();
    tg_direct_body<AVEC2, false>(d2, A2g + baseA2, s_mid1, s_mid2, I2, d2.Jtot(), K2s2, offCi_s, 0, 4);
    __syncthreads();
    tg_direct_body<false, false>(d3, s_mid2, B3g + baseB3, C3 + (long)c0 * sC3, d3.Itot(), J3, K2s3, offCi_s, 0, 4);
  }
  // rows of the result beyond the live extent of the chunked sub-index: zeros (it is the slowest sub-index of C3 in use)
  if (n1 < x_static) {
    const long lo = (long)n1 * sC3, hi = (long)x_static * sC3;
    for (long e = lo + threadIdx.x; e < hi; e += 256) C3[e] = 0.f;
  }
}

// stage 2 of the chain must read exactly what stage 1 wrote: same live extents on the shared sub-indices (the caller
// sets the same TgDyn on both), no masks on them.  Returns false when the static shapes rule the chain out.
// Returns 0 when the static shapes rule the chain out, 1 when launched (entries may be declined: flag -1), 2 when launched and
// no entry can be declined (every slice of the intermediate fits the buffer: the caller skips the fallback launches).
inline int tgemm_chain_launch(hipStream_t s, const TGemmDesc &d1_in, const TGemmDesc &d2_in, const TGemmChainMap &mp,
                              const float *A1, const float *B1, const float *A2, float *C2, int *flag, int allow_chunks = 0, int dense = 0,
                              int f64acc = 0, int tri = 0);

inline int tgemm_chain3_launch(hipStream_t s, const TGemmDesc &d1_in, const TGemmDesc &d2_in, const TGemmDesc &d3_in, const TGemmChainMap &mp,
                               const TGemmChain3Map &mp3, const float *A1, const float *B1, const float *A2, const float *B3, float *C3, int *flag,
                               const int *skip = nullptr);

bool tgemm_use_mfma();

// device counter the launches of the current profiling bracket add their contracted flops to (engine.h prof_begin)
inline thread_local unsigned long long *tg_flop_counter = nullptr;
inline thread_local unsigned long long *tg_byte_counter = nullptr;

// whether tgemm_launch<float ...> takes the wave-per-tile kernel with ONE block per batch entry for this descriptor: the
// condition under which scale_in / scale_out may be set
inline bool tgemm_one_block_direct(const TGemmDesc &d) {
  constexpr int direct_mode = 1;
  constexpr int gx_dyn = 1;
  bool any_dyn = d.dynI != nullptr;
  for (int q = 0; q < 3; ++q) any_dyn = any_dyn || d.dI[q].p || d.dJ[q].p || d.dK[q].p;
  return tgemm_use_mfma() && !d.dynK && direct_mode == 1 && any_dyn && gx_dyn == 1 && d.bdivC == 1 && !d.accumulate &&
         !d.batch_flag && d.nbatch > 0 && d.Itot() > 0 && d.Jtot() > 0 && !d.prefer_tiled;
}

template <typename TA, typename TB, typename TC, typename TAcc>
void tgemm_launch(hipStream_t s, const TGemmDesc &d_in, const TA *A, const TB *B, TC *C) {
  if (d_in.nbatch <= 0 || d_in.Itot() <= 0 || d_in.Jtot() <= 0) return;
  TGemmDesc d = d_in;
  PG_REQUIRE(d.nbatch <= 65535, 1, "walkers x candidates exceeds 65535 (grid z limit): use a smaller walker batch");
  d.flopc = tg_flop_counter;
  d.bytec = tg_byte_counter;
  d.flop_stride = d.nbatch >= 256 ? 64 : 1;   // one atomic per 64 walkers: a same-address atomic per block costs ~10 %
  bool fused_norm_ok = false;
  int gx = (d.Itot() + TG_BM - 1) / TG_BM;
  const bool dyn_i = d.dynI || (d.dI[0].p && !d.dI[0].mask) || (d.dI[1].p && !d.dI[1].mask) || (d.dI[2].p && !d.dI[2].mask);
  if (dyn_i && gx > TG_DYN_GRIDX) gx = TG_DYN_GRIDX;
  dim3 grid(gx, (d.Jtot() + TG_BN - 1) / TG_BN, d.nbatch);
  if constexpr (sizeof(TA) == 4 && sizeof(TB) == 4 && sizeof(TC) == 4 && sizeof(TAcc) == 4) {
    // small per-walker extents (rank-adaptive absorption): wave-per-tile kernel without LDS staging
    constexpr int direct_mode = 1;
    constexpr bool no_vec = false;
    const bool any_dyn = dyn_i || d.dK[0].p || d.dK[1].p || d.dK[2].p || d.dJ[0].p || d.dJ[1].p || d.dJ[2].p;
    if (tgemm_use_mfma() && !d.dynK && !d.prefer_tiled && (direct_mode == 2 || (direct_mode == 1 && any_dyn))) {
      // 16-byte loads along k2 where the operand is contiguous there and every other offset keeps the alignment
      auto al4 = [](long v) { return (v & 3) == 0; };
      const bool avec = !no_vec && d.sAk[2] == 1 && al4(d.K[2]) && al4(d.sAi[0]) && al4(d.sAi[1]) && al4(d.sAi[2]) &&
                        al4(d.sAk[0]) && al4(d.sAk[1]) && al4(d.wA) && al4(d.selA_mul) && (((uintptr_t)A) & 15) == 0;
      const bool bvec = !no_vec && d.sBk[2] == 1 && al4(d.K[2]) && al4(d.sBj[0]) && al4(d.sBj[1]) && al4(d.sBj[2]) &&
                        al4(d.sBk[0]) && al4(d.sBk[1]) && al4(d.wB) && al4(d.selB_mul) && (((uintptr_t)B) & 15) == 0;
      // (the wave-per-tile body splits row / column indices through a float reciprocal and addresses in 32-bit bytes)
      PG_REQUIRE(d.Itot() < (1 << 22) && d.Jtot() < (1 << 22), 1, "tensor GEMM: more than 2^22 rows / columns in one batch entry");
      fused_norm_ok = true;
      const int tiles = ((d.Itot() + 31) / 32) * ((d.Jtot() + 31) / 32);
      // with per-walker live extents the tile count is a few: one block (four waves) walks them; extra blocks
      // would only pay the chain of dependent loads (extents, selector, offsets) and exit
      constexpr int gx_dyn = 1;
      const dim3 gd(any_dyn ? std::min(gx_dyn, std::max(1, tiles / 4)) : (tiles >= 64 ? 4 : tiles >= 16 ? 2 : 1), 1,
                    d.nbatch);
      PG_REQUIRE(!d.scale_out || (gd.x == 1 && d.bdivC == 1 && !d.accumulate && !d.batch_flag), 5,
                 "tensor GEMM: the norm of the result needs one block per batch entry");
      const float *Af = (const float *)A, *Bf = (const float *)B;
      float *Cf = (float *)C;
      if (d.acc64) {
        if (avec && bvec) hipLaunchKernelGGL((tgemm_direct_kernel<true, true, true>), gd, dim3(256), 0, s, d, Af, Bf, Cf);
        else if (avec) hipLaunchKernelGGL((tgemm_direct_kernel<true, false, true>), gd, dim3(256), 0, s, d, Af, Bf, Cf);
        else if (bvec) hipLaunchKernelGGL((tgemm_direct_kernel<false, true, true>), gd, dim3(256), 0, s, d, Af, Bf, Cf);
        else hipLaunchKernelGGL((tgemm_direct_kernel<false, false, true>), gd, dim3(256), 0, s, d, Af, Bf, Cf);
      }
      else if (avec && bvec) hipLaunchKernelGGL((tgemm_direct_kernel<true, true>), gd, dim3(256), 0, s, d, Af, Bf, Cf);
      else if (avec) hipLaunchKernelGGL((tgemm_direct_kernel<true, false>), gd, dim3(256), 0, s, d, Af, Bf, Cf);
      else if (bvec) hipLaunchKernelGGL((tgemm_direct_kernel<false, true>), gd, dim3(256), 0, s, d, Af, Bf, Cf);
      else hipLaunchKernelGGL((tgemm_direct_kernel<false, false>), gd, dim3(256), 0, s, d, Af, Bf, Cf);
      PG_CHECK_HIP(hipGetLastError());
      return;
    }
  }
  (void)fused_norm_ok;
  PG_REQUIRE(!d.scale_out && !d.scale_in, 5, "tensor GEMM: scale_in / scale_out need the wave-per-tile kernel");
  if constexpr (std::is_same<TAcc, double>::value && !is_cplx<TA>::value && !is_cplx<TB>::value && !is_cplx<TC>::value) {
    // skinny float64-accumulated products (one side <= 32): 128 x 32 / 32 x 128 block tiles, every wave on a live quadrant
    constexpr bool no_skinny = false;
    if (tgemm_use_mfma() && !no_skinny && !d.upper_only && d.Itot() > 1 && d.Jtot() > 1) {
      if (d.Jtot() <= 32 && d.Itot() >= 64) {
        int gxs = (d.Itot() + 127) / 128;
        if (dyn_i && gxs > TG_DYN_GRIDX) gxs = TG_DYN_GRIDX;
        hipLaunchKernelGGL((tgemm_skinny_f64_kernel<TA, TB, TC, 128, 32>), dim3(gxs, 1, d.nbatch), dim3(256), 0, s, d, A, B, C);
        PG_CHECK_HIP(hipGetLastError());
        return;
      }
      if (d.Itot() <= 32 && d.Jtot() >= 64) {
        hipLaunchKernelGGL((tgemm_skinny_f64_kernel<TA, TB, TC, 32, 128>), dim3(1, (d.Jtot() + 127) / 128, d.nbatch), dim3(256), 0, s, d, A, B, C);
        PG_CHECK_HIP(hipGetLastError());
        return;
      }
    }
  }
  if constexpr (is_cplx<TAcc>::value) {
    // complex element type: four real v_mfma_f64_16x16x4_f64 products per tile from the interleaved LDS operands
    // (the same tiling on the vector ALUs was round 2's path: USE_MFMA = false)
    constexpr bool no_cmfma = false;
    if constexpr (std::is_same<TAcc, c128>::value) {
      if (tgemm_use_mfma() && !no_cmfma) {
        hipLaunchKernelGGL((tgemm_kernel<TA, TB, TC, TAcc, true>), grid, dim3(256), 0, s, d, A, B, C);
        PG_CHECK_HIP(hipGetLastError());
        return;
      }
    }
    hipLaunchKernelGGL((tgemm_kernel<TA, TB, TC, TAcc, false>), grid, dim3(256), 0, s, d, A, B, C);
  } else {
    if (tgemm_use_mfma())
      hipLaunchKernelGGL((tgemm_kernel<TA, TB, TC, TAcc, true>), grid, dim3(256), 0, s, d, A, B, C);
    else
      hipLaunchKernelGGL((tgemm_kernel<TA, TB, TC, TAcc, false>), grid, dim3(256), 0, s, d, A, B, C);
  }
  PG_CHECK_HIP(hipGetLastError());
}

inline int tgemm_chain_launch(hipStream_t s, const TGemmDesc &d1_in, const TGemmDesc &d2_in, const TGemmChainMap &mp,
                              const float *A1, const float *B1, const float *A2, float *C2, int *flag, int allow_chunks, int dense, int f64acc,
                              int tri) {
  // tri: the A operand of stage 1 is a row-compacted upper-triangular factor (see the chunk loop of the kernel)
  const int akf = (allow_chunks ? 1 : 0) | (tri ? 2 : 0);
  if (!tgemm_use_mfma() || d1_in.dynK || d2_in.dynK || d1_in.nbatch != d2_in.nbatch || d1_in.nbatch <= 0) return 0;
  if (d1_in.bdivA != 1 || d1_in.bdivB != 1 || d2_in.bdivA != 1 || d2_in.bdivC != 1) return 0;
  TGemmDesc d1 = d1_in, d2 = d2_in;
  if (d1.Itot() >= (1 << 22) || d1.Jtot() >= (1 << 22) || d2.Itot() >= (1 << 22) || d2.Jtot() >= (1 << 22)) return 0;
  d1.flopc = tg_flop_counter; d1.bytec = tg_byte_counter;
  d1.flop_stride = d1.nbatch >= 256 ? 64 : 1;
  constexpr bool no_vec = false;
  auto al4 = [](long v) { return (v & 3) == 0; };
  const bool avec1 = !no_vec && d1.sAk[2] == 1 && al4(d1.K[2]) && al4(d1.sAi[0]) && al4(d1.sAi[1]) && al4(d1.sAi[2]) &&
                     al4(d1.sAk[0]) && al4(d1.sAk[1]) && al4(d1.wA) && al4(d1.selA_mul) && (((uintptr_t)A1) & 15) == 0;
  const bool bvec1 = !no_vec && d1.sBk[2] == 1 && al4(d1.K[2]) && al4(d1.sBj[0]) && al4(d1.sBj[1]) && al4(d1.sBj[2]) &&
                     al4(d1.sBk[0]) && al4(d1.sBk[1]) && al4(d1.wB) && al4(d1.selB_mul) && (((uintptr_t)B1) & 15) == 0;
  const bool avec2 = !no_vec && d2.sAk[2] == 1 && al4(d2.K[2]) && al4(d2.sAi[0]) && al4(d2.sAi[1]) && al4(d2.sAi[2]) &&
                     al4(d2.sAk[0]) && al4(d2.sAk[1]) && al4(d2.wA) && al4(d2.selA_mul) && (((uintptr_t)A2) & 15) == 0;
  const dim3 g(d1.nbatch), blk(256);
  // dense walker batch (hint of the caller): the intermediate is walked in chunks anyway; a 32 KB buffer holds four values of
  // the chunked sub-index = 32 full rows per stage-1 tile and eight balanced stage-2 tiles (24 KB: 24 rows, six tiles), at four
  // blocks per CU instead of six (0 / 8192 / 16384 floats were measured in round 5: HISTORY item 13)
  int ldsf = (dense && allow_chunks) ? 8192 : TG_CHAIN_LDS_FLOATS;
  if (f64acc && ldsf > 8192) ldsf = 8192;       // (the float64-accumulating form is built for the two default buffer sizes)
#define PG_CHAIN(a1, b1, a2)                                                                                                   \
  do {                                                                                                                         \
    if (f64acc && ldsf >= 8192) hipLaunchKernelGGL((tgemm_chain_kernel<a1, b1, a2, 8192, 4, true>), g, blk, 0, s, d1, d2, mp, A1, B1, A2, C2, flag, 0, akf); \
    else if (f64acc) hipLaunchKernelGGL((tgemm_chain_kernel<a1, b1, a2, TG_CHAIN_LDS_FLOATS, 4, true>), g, blk, 0, s, d1, d2, mp, A1, B1, A2, C2, flag, 0, akf); \
    else if (ldsf == 8192 && tri) hipLaunchKernelGGL((tgemm_chain_kernel<a1, b1, a2, 8192, 4, false, true>), g, blk, 0, s, d1, d2, mp, A1, B1, A2, C2, flag, 0, akf); \
    else if (ldsf == 8192) hipLaunchKernelGGL((tgemm_chain_kernel<a1, b1, a2, 8192, 4>), g, blk, 0, s, d1, d2, mp, A1, B1, A2, C2, flag, 0, akf); \
    else hipLaunchKernelGGL((tgemm_chain_kernel<a1, b1, a2, TG_CHAIN_LDS_FLOATS, 6>), g, blk, 0, s, d1, d2, mp, A1, B1, A2, C2, flag, 0, akf); \
  } while (0)
  if (avec1 && bvec1 && avec2) PG_CHAIN(true, true, true);
  else if (avec1 && bvec1) PG_CHAIN(true, true, false);
  else if (avec1 && avec2) PG_CHAIN(true, false, true);
  else if (avec1) PG_CHAIN(true, false, false);
  else if (bvec1 && avec2) PG_CHAIN(false, true, true);
  else if (bvec1) PG_CHAIN(false, true, false);
  else if (avec2) PG_CHAIN(false, false, true);
  else PG_CHAIN(false, false, false);
#undef PG_CHAIN
  PG_CHECK_HIP(hipGetLastError());
  // the kernel's own test, on the static extents (live extents are never larger)
  int jsub = -1;
  for (int q = 0; q < 3; ++q) {
    if (mp.mapJ[q] == 1) jsub = (jsub < 0 && !(d2.dJ[q].p && d2.dJ[q].mask)) ? q : 3;
    if (mp.mapK[q] == 1) jsub = 3;
  }
  const long whole = (long)d1.Itot() * d1.Jtot(), per = (long)d1.I[2] * d1.Jtot();
  const bool chunkable = allow_chunks && d1.I[0] == 1 && !d1.dynI && !d2.dynI && jsub >= 0 && jsub <= 2 &&
                         !(d1.dI[1].p && d1.dI[1].mask) && per <= ldsf;
  return (whole <= ldsf || chunkable) ? 2 : 1;
}

// Returns 0 when the static shapes rule the three-stage chain out (nothing launched), 2 when launched and no entry can be declined
// (the caller needs no fallback); it never launches a chain that could decline an entry.
inline int tgemm_chain3_launch(hipStream_t s, const TGemmDesc &d1_in, const TGemmDesc &d2_in, const TGemmDesc &d3_in, const TGemmChainMap &mp,
                               const TGemmChain3Map &mp3, const float *A1, const float *B1, const float *A2, const float *B3, float *C3, int *flag,
                               const int *skip) {
  if (!tgemm_use_mfma() || d1_in.dynK || d2_in.dynK || d3_in.dynK || d1_in.nbatch != d2_in.nbatch || d1_in.nbatch != d3_in.nbatch ||
      d1_in.nbatch <= 0)
    return 0;
  if (d1_in.bdivA != 1 || d1_in.bdivB != 1 || d2_in.bdivA != 1 || d2_in.bdivC != 1 || d3_in.bdivA != 1 || d3_in.bdivC != 1) return 0;
  if (d1_in.dynI || d2_in.dynI || d3_in.dynI || d3_in.accumulate || d3_in.scale_in || d3_in.scale_out || d3_in.selA || d3_in.selB) return 0;
  TGemmDesc d1 = d1_in, d2 = d2_in, d3 = d3_in;
  if (d1.Itot() >= (1 << 22) || d1.Jtot() >= (1 << 22) || d2.Itot() >= (1 << 22) || d2.Jtot() >= (1 << 22) || d3.Itot() >= (1 << 22) ||
      d3.Jtot() >= (1 << 22))
    return 0;
  constexpr int lds3 = 4096;     // (measured: 8192 walkers of the headline state 15.9 k sweeps/s at 4096 floats x 2 / four blocks per CU, 15.0 k at 6656 / three, 15.7 k without the third stage)
  const int ldsf = lds3;
  // the kernel's own test, on the static extents (live extents are never larger)
  int jsub = -1;
  for (int q = 0; q < 3; ++q) {
    if (mp.mapJ[q] == 1) jsub = (jsub < 0 && !(d2.dJ[q].p && d2.dJ[q].mask)) ? q : 3;
    if (mp.mapK[q] == 1) jsub = 3;
  }
  if (jsub < 0 || jsub > 2 || mp3.chunkI < 0 || mp3.chunkI > 2 || d1.I[0] != 1 || (d1.dI[1].p && d1.dI[1].mask)) return 0;
  if (mp3.mapI[mp3.chunkI] != 3 + jsub) return 0;                              // the chunked sub-index of stage 3 is that J of stage 2
  for (int q = 0; q < mp3.chunkI; ++q) if (d3.I[q] != 1) return 0;               // ... and the slowest I sub-index of C3 in use
  const long per1 = (long)d1.I[2] * d1.Jtot(), per2 = (long)d2.Itot() * d2.Jtot() / (d2.J[jsub] > 0 ? d2.J[jsub] : 1);
  if (per1 > ldsf || per2 > ldsf) return 0;
  d1.flopc = tg_flop_counter; d1.bytec = tg_byte_counter;
  d1.flop_stride = d1.nbatch >= 256 ? 64 : 1;
  constexpr bool no_vec = false;
  auto al4 = [](long v) { return (v & 3) == 0; };
  const bool avec1 = !no_vec && d1.sAk[2] == 1 && al4(d1.K[2]) && al4(d1.sAi[0]) && al4(d1.sAi[1]) && al4(d1.sAi[2]) &&
                     al4(d1.sAk[0]) && al4(d1.sAk[1]) && al4(d1.wA) && al4(d1.selA_mul) && (((uintptr_t)A1) & 15) == 0;
  const bool bvec1 = !no_vec && d1.sBk[2] == 1 && al4(d1.K[2]) && al4(d1.sBj[0]) && al4(d1.sBj[1]) && al4(d1.sBj[2]) &&
                     al4(d1.sBk[0]) && al4(d1.sBk[1]) && al4(d1.wB) && al4(d1.selB_mul) && (((uintptr_t)B1) & 15) == 0;
  const bool avec2 = !no_vec && d2.sAk[2] == 1 && al4(d2.K[2]) && al4(d2.sAi[0]) && al4(d2.sAi[1]) && al4(d2.sAi[2]) &&
                     al4(d2.sAk[0]) && al4(d2.sAk[1]) && al4(d2.wA) && al4(d2.selA_mul) && (((uintptr_t)A2) & 15) == 0;
  const dim3 g(d1.nbatch), blk(256);
#define PG_CHAIN3(a1, b1, a2)                                                                                                               \
  do {                                                                                                                                      \
    hipLaunchKernelGGL((tgemm_chain3_kernel<a1, b1, a2, 4096, 4>), g, blk, 0, s, d1, d2, d3, mp, mp3, A1, B1, A2, B3, C3, flag, skip);   \
  } while (0)
  if (avec1 && bvec1 && avec2) PG_CHAIN3(true, true, true);
  else if (avec1 && bvec1) PG_CHAIN3(true, true, false);
  else if (avec1 && avec2) PG_CHAIN3(true, false, true);
  else if (avec1) PG_CHAIN3(true, false, false);
  else if (bvec1 && avec2) PG_CHAIN3(false, true, true);
  else if (bvec1) PG_CHAIN3(false, true, false);
  else if (avec2) PG_CHAIN3(false, false, true);
  else PG_CHAIN3(false, false, false);
#undef PG_CHAIN3
  PG_CHECK_HIP(hipGetLastError());
  return 2;
}

}  // namespace pepsgpu
