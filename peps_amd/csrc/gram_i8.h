// Exact-integer Gram matrix on the i8 matrix cores (round 4) -- the forward Gram of dense walkers, same contract as
// gram_cols_lds_kernel (gram.h):
//
//     G[b] = P[b]^T P[b],   P = K x n float32 (K live rows per walker, 192 < n <= 256 columns, row stride ld), G = n x n float64
//
// Why.  The Gram has to carry the f32 data without squaring their rounding (the Cholesky factor of G is the carry of the next
// site, cond(P) ~ 1e5..1e6 on a real PEPS), so it ran on v_mfma_f64_16x16x4_f64: 64 cycles per 4 k of a 16 x 16 tile, the floor of
// the dense step.  A float32 is a 24-bit integer times a power of two.  Per block of 64 rows every column gets ONE
// power of two (from its largest magnitude in the block), its elements become integers |n| <= 2^22 (round to nearest: a backward
// perturbation of P of 2^-23 of the column's block maximum, the size of the f32 rounding P already carries), and
// n = d0 2^16 + d1 2^8 + d2 with three SIGNED BYTES.  Then
//
//     sum_k n_ki n_kj = sum_{a,b} 2^(32 - 8 (a + b)) sum_k d_a,ki d_b,kj
//
// and each of the nine byte products is one v_mfma_i32_16x16x64_i8 (16 cycles for 64 k, exact int32 accumulation): 144 cycles per
// 64 rows of a tile against 1024 on the f64 pipe.  The five weight classes S_0..S_4 (|S_c| < 2^22) are merged in integer
// arithmetic (U = 256 S_0 + S_1, W = 256 S_2 + S_3), converted to float64 exactly (T = 2^24 U + 2^8 W + S_4 < 2^53) and added into
// the float64 accumulator of the tile scaled by the two column exponents of the block: the result is the float64 Gram of the
// fixed-point image of P, every integer step exact.
//
// Shape of the work: one workgroup per walker (8 or 12 waves: template parameter NW); the 136 upper 16 x 16 tiles dealt in runs of
// 17 (or 12 / 11) per wave, their float64 accumulators in registers for the whole walker (136 or 96 VGPRs; the per-tile transients
// of the 16 x 16 x 64 form are 44).
// Per 64-row block: the rows arrive in LDS by LDS-DMA (global_load_lds_dwordx4: one row of 256 floats per wave-instruction, no
// staging registers, in flight during the MFMAs of the block before); thread (column j, half h) then reads rows 32 h .. 32 h + 31 of
// its column, the column maximum goes through LDS, and the digits are laid down as three byte planes D[plane][column][k]
// (k contiguous: a lane's MFMA operand is one ds_read_b128; any bijection of the 64 k onto (lane group, byte) serves, the A and the
// B operand use the same one).
#pragma once
#include "common.h"      // (included from the middle of gram.h: the launchers follow it there)

namespace pepsgpu {

typedef int gi_i32x4 __attribute__((ext_vector_type(4)));

constexpr int GI_KB = 64;                     // rows per block = k of one v_mfma_i32_16x16x64_i8
constexpr int GI_PITCH = 80;                  // bytes per column and plane (64 + 16: the 16 lanes of a b128 phase on distinct banks)
constexpr int GI_PLANE = 256 * GI_PITCH;
constexpr int GI_BUF = 3 * GI_PLANE;
constexpr int GI_RAW = GI_KB * 256 * 4;       // the float32 rows of the next block (LDS-DMA image: one row = one wave-instruction)
inline size_t gram_cols_i8_smem_bytes() { return GI_BUF + GI_RAW + 256 * sizeof(int) + 2 * 256 * sizeof(float) + 16; }

// ROWS: the row Gram of the truncation input instead (gram_rows_f64_kernel of gram.h: G = M M^T, M = n x K row-major with row stride K,
// n = nrows[b] live rows, the contracted index runs along the rows) -- the same kernel with the roles of the two indices of the
// operand exchanged: "column" j of the text above is row j of M, a block is 64 consecutive k of every row.
// DBG: timing-only variants of scripts/gram_i8_bench.hip (1 no drain, 2 no digit pass after block 0, 4 no MFMA).
// NW: waves of the workgroup (8: seventeen tiles per wave, two waves per SIMD; 12: twelve / eleven tiles per wave, three waves per
// SIMD -- fewer accumulators per wave, one more wave per SIMD to fill the LDS / MFMA latencies of the tile loop; the digit pass is
// done by the first eight waves either way).
template <int NW, int W> struct GiTile {       // tile t of wave W: the upper triangle of the 16 x 16 tile grid, rows first, dealt in runs
  static constexpr int per = 136 / NW, extra = 136 % NW;
  static constexpr int count = per + (W < extra ? 1 : 0);
  static constexpr int first = W * per + (W < extra ? W : extra);
  static constexpr int start(int x) { return 16 * x - x * (x - 1) / 2; }
  static constexpr int x(int t) { int r = 0; while (r < 15 && start(r + 1) <= first + t) ++r; return r; }
  static constexpr int c(int t) { return x(t) + (first + t - start(x(t))); }
};

template <typename T, bool ROWS = false, int DBG = 0, int NW = 8>
__global__ __launch_bounds__(64 * NW, NW == 12 ? 3 : 2) void gram_cols_i8_kernel(const T *__restrict__ Pg, long wP, int n, int ld,
                                                              const int *__restrict__ kdyn, int kdyn_mul, int kmax,
                                                              double *__restrict__ Gg, long wG, int ldg,
                                                              const int *__restrict__ run_flag, int inner,
                                                              const int *__restrict__ inner_live,
                                                              const int *__restrict__ nrows,
                                                              unsigned long long *__restrict__ flopc,
                                                              unsigned long long *__restrict__ bytec, int flop_stride, int sym = 0) {
  // sym: the tiles above the diagonal are written to their mirror positions too (G symmetric in memory: the pivoted factorisation of
  // chol_pivot.h reads whole rows)
  static_assert(sizeof(T) == 4, "f32 input");
  extern __shared__ __attribute__((aligned(16))) unsigned char gi_smem[];
  unsigned char *dig = gi_smem;
  float *raw = reinterpret_cast<float *>(gi_smem + GI_BUF);
  int *exps = reinterpret_cast<int *>(gi_smem + GI_BUF + GI_RAW);
  float *pmax = reinterpret_cast<float *>(gi_smem + GI_BUF + GI_RAW + 256 * sizeof(int));
  // non-finite input: the fixed-point image of a NaN / Inf is finite garbage (fmaxf drops a NaN from the column maximum, the float add
  // of the rounding trick is reinterpreted bitwise), so the kernel looks for it itself -- x * 0 is NaN exactly for NaN and Inf -- and
  // poisons the diagonal of G with NaN at the end: the Cholesky that follows flags the walker, as with the float64 Gram this replaces
  int *bad_s = reinterpret_cast<int *>(gi_smem + GI_BUF + GI_RAW + 256 * sizeof(int) + 2 * 256 * sizeof(float));
  const int b = blockIdx.x;
  if (run_flag && run_flag[b] >= 0) return;
  if (threadIdx.x == 0) *bad_s = 0;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int K = ROWS ? kmax : (kdyn ? max(0, min(kmax, kdyn[b] * kdyn_mul)) : kmax);
  if (ROWS) {
    n = min(n, nrows[b]);
    if (n <= 0) return;
  }
  if (run_flag) flop_stride = 1;
  if (flopc && tid == 0 && b % flop_stride == 0) {
    atomicAdd(flopc, (unsigned long long)flop_stride * n * n * K);
    if (bytec) atomicAdd(bytec, (unsigned long long)flop_stride * ((unsigned long long)K * n * sizeof(T) + (unsigned long long)n * n * 4));
  }
  const T *P = Pg + (long)b * wP;
  double *G = Gg + (long)b * wG;
  const int ilive = inner_live ? min(inner, inner_live[b]) : inner;
  const int c16 = lane & 15, g4 = lane >> 4;
  const int j = tid & 255, kh = (wave >> 2) & 1;
  const bool layer = NW == 8 || wave < 8;          // (the digit pass: 512 threads)
  const bool col_ok = layer && j < n && (ROWS || (j % inner) < ilive);          // dead / absent column: never written in P
  const int nb = (K + GI_KB - 1) / GI_KB;
  // cols: rows 8 wave .. 8 wave + 7 of block blk, lane l fetches floats 4 l .. 4 l + 3 of the row (rows beyond K / columns beyond n: a
  // clamped address, masked when the image is read).  rows: the image is [row of M][64 k]; lane l of instruction q fetches for row
  // 32 wave + 4 q + l / 16 the k-chunk (l % 16) ^ (row % 16) -- the image of a row is a permutation of its sixteen 16-byte chunks, so that
  // the sixteen lanes of a ds_read_b128 phase (consecutive rows, same k) hit sixteen different bank groups.
  auto issue = [&](int blk) __attribute__((always_inline)) {
    if (NW > 8 && wave >= 8) return;
    if constexpr (ROWS) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int jr = 32 * wave + 4 * q + (lane >> 4);
        const int kc = blk * GI_KB + 4 * ((lane & 15) ^ (jr & 15));
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(P + (long)min(jr, n - 1) * ld + min(kc, K - 4)),
                                         (__attribute__((address_space(3))) void *)(raw + (32 * wave + 4 * q) * 64), 16, 0, 0);
      }
    } else {
      const int cs = 4 * lane < n ? 4 * lane : 0;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int rl = 8 * wave + q, r = min(blk * GI_KB + rl, K - 1);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(P + (long)r * ld + cs),
                                         (__attribute__((address_space(3))) void *)(raw + rl * 256), 16, 0, 0);
      }
    }
  };
  // digits of block blk from the LDS image of its rows: one pass for the column maxima, a second one (sixteen rows at a time, the
  // accumulators leave few registers) for the digits
  auto lay = [&](int blk) __attribute__((always_inline)) {
    const int r0 = blk * GI_KB + 32 * kh;
    // element q (0..31) of this thread's half column in the image
    auto at = [&](int q) __attribute__((always_inline)) -> float {
      if constexpr (ROWS) return raw[j * 64 + 4 * (((32 * kh + q) >> 2) ^ (j & 15)) + (q & 3)];
      else return raw[(32 * kh + q) * 256 + j];
    };
    const int qlive = col_ok ? min(32, K - r0) : 0;             // rows of this half that exist (the others: clamped copies, masked)
    float m = 0.f, chk = 0.f;
#pragma unroll
    for (int q = 0; q < 32; ++q) {
      const float x = q < qlive ? at(q) : 0.f;
      m = fmaxf(m, fabsf(x));
      chk = fmaf(x, 0.f, chk);
    }
    if (chk != 0.f) *bad_s = 1;       // (NaN != 0; benign race: every writer stores 1)
    if (layer) pmax[256 * kh + j] = m;
    __syncthreads();
    m = fmaxf(pmax[j], pmax[256 + j]);
    const int e = max(30, (int)(__float_as_uint(m) >> 23));   // biased exponent of the block maximum (|x| < 2^(e - 126))
    if (kh == 0 && layer) exps[j] = e - 148;                           // x = n 2^(e - 148), |n| <= 2^22
    const float sc = __uint_as_float((unsigned)(275 - e) << 23);
    unsigned char *dst = dig + j * GI_PITCH + 32 * kh;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      unsigned pl[3][4];
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        unsigned w[4];
#pragma unroll
        for (int z = 0; z < 4; ++z) {
          const int q = 16 * hh + 4 * q4 + z;
          const float x = q < qlive ? at(q) : 0.f;
          // round to nearest integer by the float add: 1.5 * 2^23 + n has n (two's complement, |n| <= 2^22) in its mantissa bits
          const unsigned yb = __float_as_uint(fmaf(x, sc, 12582912.f));
          w[z] = (yb + (0x808080u - 0x4B400000u)) ^ 0x808080u;       // bytes 0, 1, 2 = the signed digits d2, d1, d0 of n
        }
        const unsigned t01l = __builtin_amdgcn_perm(w[1], w[0], 0x05010400u), t01h = __builtin_amdgcn_perm(w[1], w[0], 0x0c0c0602u);
        const unsigned t23l = __builtin_amdgcn_perm(w[3], w[2], 0x05010400u), t23h = __builtin_amdgcn_perm(w[3], w[2], 0x0c0c0602u);
        pl[2][q4] = __builtin_amdgcn_perm(t23l, t01l, 0x05040100u);
        pl[1][q4] = __builtin_amdgcn_perm(t23l, t01l, 0x07060302u);
        pl[0][q4] = __builtin_amdgcn_perm(t23h, t01h, 0x05040100u);
      }
      if (layer) {
#pragma unroll
        for (int p = 0; p < 3; ++p)
          *reinterpret_cast<uint4 *>(dst + p * GI_PLANE + 16 * hh) = make_uint4(pl[p][0], pl[p][1], pl[p][2], pl[p][3]);
      }
    }
    __syncthreads();
  };
  auto run = [&](auto wc) __attribute__((always_inline)) {
    constexpr int W = decltype(wc)::value;
    using Tl = GiTile<NW, W>;
    constexpr int NT = Tl::count;
    double acc[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[t][r] = 0.0;
    if (nb > 0) {
      issue(0);
      __syncthreads();                    // (the fence of the barrier waits for the LDS-DMA: vmcnt(0))
      lay(0);
    }
    for (int blk = 0; blk < nb; ++blk) {
      if (blk + 1 < nb) issue(blk + 1);   // in flight during the MFMAs of this block
      const unsigned char *src = dig + c16 * GI_PITCH + 16 * g4;
      const int *ex = exps;
      gi_i32x4 a[3], bq[3], ei;
      int ejb;
      auto fetch_b = [&](int c, gi_i32x4(&q)[3], int &e) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < 3; ++p) q[p] = *reinterpret_cast<const gi_i32x4 *>(src + p * GI_PLANE + 16 * c * GI_PITCH);
        e = ex[16 * c + c16] + 1023;
      };
      auto fetch_a = [&](int x, gi_i32x4(&q)[3], gi_i32x4 &e) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < 3; ++p) q[p] = *reinterpret_cast<const gi_i32x4 *>(src + p * GI_PLANE + 16 * x * GI_PITCH);
        e = *reinterpret_cast<const gi_i32x4 *>(ex + 16 * x + 4 * g4);
      };
      fetch_a(Tl::x(0), a, ei);
      fetch_b(Tl::c(0), bq, ejb);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int c = Tl::c(t);
        const bool new_row = t + 1 < NT && Tl::x(t + 1 < NT ? t + 1 : t) != Tl::x(t);
        if (t > 0) fetch_b(c, bq, ejb);
        // (wave-uniform) tile beyond the columns of this launch.  The branch also keeps the tiles apart: as one basic block (n == 256 as
        // a template parameter) the seventeen tiles are scheduled over each other and spill 1.6 KB per lane.
        if (16 * c < n) {
          const gi_i32x4 z4 = {0, 0, 0, 0};
          gi_i32x4 s0, s1, s2, s3, s4;
          if (DBG & 4) {
            s0 = a[0] + bq[0]; s1 = a[1] + bq[1]; s2 = a[2] + bq[2]; s3 = a[0] - bq[1]; s4 = a[1] - bq[2];
          } else {
            s0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[0], bq[0], z4, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[0], bq[1], z4, 0, 0, 0);
            s2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[0], bq[2], z4, 0, 0, 0);
            s3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[1], bq[2], z4, 0, 0, 0);
            s4 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[2], bq[2], z4, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[1], bq[0], s1, 0, 0, 0);
            s2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[1], bq[1], s2, 0, 0, 0);
            s3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[2], bq[1], s3, 0, 0, 0);
            s2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[2], bq[0], s2, 0, 0, 0);
          }
          // C layout of the 16 x 16 int32 tile: column = lane & 15, row = 4 (lane >> 4) + register
          if (DBG & 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[t][r] += (double)(s0[r] + s1[r] + s2[r] + s3[r] + s4[r]);
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int u = (s0[r] << 8) + s1[r], w2 = (s2[r] << 8) + s3[r];
              const double tt = fma((double)u, 16777216.0, fma((double)w2, 256.0, (double)s4[r]));
              // 2^(e_i + e_j) assembled in the exponent field: one integer instruction instead of v_ldexp_f64 + v_add_f64
              acc[t][r] = fma(tt, __hiloint2double((ei[r] + ejb) << 20, 0), acc[t][r]);
            }
          }
        }
        if (t + 1 < NT) {
          if (new_row) fetch_a(Tl::x(t + 1 < NT ? t + 1 : t), a, ei);      // (two or three times per block: not prefetched)
        }
      }
      __syncthreads();                    // every wave is done with the digits of this block; the next rows have landed
      if (blk + 1 < nb && !(DBG & 2)) lay(blk + 1);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * Tl::x(t) + 4 * g4 + r, jj = 16 * Tl::c(t) + c16;
        if (i < n && jj < n) G[(long)i * ldg + jj] = acc[t][r];
      }
    if (sym) {      // a lane's four values are consecutive in the mirrored row: 32-byte runs, four lanes fill a 128-byte line
#pragma unroll
      for (int t = 0; t < NT; ++t)
        if (Tl::x(t) != Tl::c(t)) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = 16 * Tl::x(t) + 4 * g4 + r, jj = 16 * Tl::c(t) + c16;
            if (i < n && jj < n) G[(long)jj * ldg + i] = acc[t][r];
          }
        }
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
    case 7: run(std::integral_constant<int, 7>{}); break;
    case 8: run(std::integral_constant<int, (NW > 8 ? 8 : 0)>{}); break;
    case 9: run(std::integral_constant<int, (NW > 8 ? 9 : 0)>{}); break;
    case 10: run(std::integral_constant<int, (NW > 8 ? 10 : 0)>{}); break;
    default: run(std::integral_constant<int, (NW > 8 ? 11 : 0)>{}); break;
  }
  __syncthreads();
  if (*bad_s && tid < n) G[(long)tid * ldg + tid] = __longlong_as_double(0x7ff8000000000000ll);
}

}  // namespace pepsgpu
