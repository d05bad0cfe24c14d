// Register-resident one-sided Jacobi for the bulk chi-truncation blocks (up to 256 x 256, f32).
//
// The 256 KiB working matrix of the C4 workload (12x12, D=8, chi=32: M_i = R_i T_i is 256 x 256)
// fits neither the 160 KiB LDS nor one wave, but it fits the register file of one CU: 8 waves x
// 32 rows x 256 columns = 128 VGPRs per lane (2 waves per SIMD, 256-VGPR budget, no spills).
// Rows are grouped in 16 blocks of 16; wave j holds the pair (top[j], bottom[j]) of a round-robin
// tournament over blocks.  One "super-round" = every wave orthogonalises the 256 row pairs
// between its two blocks entirely in registers (dot products by DPP / permlane-swap all-reduce),
// then the blocks move one step round the circle through LDS (9 x 16 KiB slots).  15 super-rounds
// = one sweep over all row pairs (and one full turn of the circle, so every block is back in its
// own place); the pairs inside a block are done once per sweep.  Each row crosses LDS once per 16
// rotations, so the LDS write bandwidth -- the limit of an LDS-resident Jacobi, which rewrites
// 2 KiB per rotation -- is off the critical path.
//
// Same mathematics as jacobi_rows_kernel (linalg.h): threshold 2*sqrt(len)*eps, noise floor
// NOISE_C*eps*|M|_F, no row swapping (select_rows_kernel sorts by norm afterwards).
#pragma once
#include "linalg.h"

namespace pepsgpu {

typedef float jr_f2 __attribute__((ext_vector_type(2)));

template <int CTRL>
__device__ __forceinline__ float jr_dpp_add(float v) {
  // old = 0 with bound_ctrl: every lane of these controls (quad_perm, the mirrors) has a valid source, so `old` is never used --
  // but passing the value itself as `old` (rounds 1-4) ties the result to a copy of it and keeps the compiler's DPP combiner from
  // folding the move into the add: three instructions (v_mov, v_mov_dpp, v_add) instead of one v_add_f32_dpp per step
  const int r = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true);
  return v + __builtin_bit_cast(float, r);
}

// sum over the 64 lanes, result in every lane: 4 DPP adds inside a row of 16, then the two
// gfx950 half-swaps across rows
__device__ __forceinline__ float jr_allsum(float v) {
  v = jr_dpp_add<0xB1>(v);    // quad_perm [1,0,3,2]
  v = jr_dpp_add<0x4E>(v);    // quad_perm [2,3,0,1]
  v = jr_dpp_add<0x141>(v);   // row_half_mirror
  v = jr_dpp_add<0x140>(v);   // row_mirror
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

struct JrRow { jr_f2 lo, hi; };   // columns 4*lane .. 4*lane+3

__device__ __forceinline__ float jr_dot(const JrRow &x, const JrRow &y) {
  const jr_f2 p = x.lo * y.lo + x.hi * y.hi;
  return p.x + p.y;
}

// Branch-free Hestenes rotation.  Parameters in f32 with hardware rcp/sqrt/rsq: an error in
// (c, s) only scales BOTH rows by sqrt(c^2+s^2) (they stay orthogonal) and rows are normalised at
// the end.  A pair below threshold gets t = 0, i.e. the identity.
__device__ __forceinline__ int jr_apply(JrRow &x, JrRow &y, float &nx, float &ny, const float g, const float tol2,
                                        const float floor2) {
  const bool go = g * g > tol2 * nx * ny && nx > floor2 && ny > floor2;
  if (!__any(go)) return 0;     // wave-uniform: no lane rotates this pair (late sweeps: most pairs) -- the kernels are VALU bound
  const float zeta = (ny - nx) * __builtin_amdgcn_rcpf(2.f * g);
  const float az = fabsf(zeta);
  float t = copysignf(__builtin_amdgcn_rcpf(az + __builtin_amdgcn_sqrtf(fmaf(az, az, 1.f))), zeta);
  t = go ? t : 0.f;
  const float cs = __builtin_amdgcn_rsqf(fmaf(t, t, 1.f)), sn = cs * t;
  const jr_f2 c2 = {cs, cs}, s2 = {sn, sn};
  const jr_f2 xl = x.lo, xh = x.hi, yl = y.lo, yh = y.hi;
  const jr_f2 syl = s2 * yl, syh = s2 * yh, sxl = s2 * xl, sxh = s2 * xh;    // (each update may overwrite its own operand: jrx_apply)
  x.lo = __builtin_elementwise_fma(c2, xl, -syl);
  x.hi = __builtin_elementwise_fma(c2, xh, -syh);
  y.lo = __builtin_elementwise_fma(c2, yl, sxl);
  y.hi = __builtin_elementwise_fma(c2, yh, sxh);
  const float tg = t * g;
  nx = fmaxf(nx - tg, 0.f);
  ny = ny + tg;
  return go ? 1 : 0;
}

constexpr int JR_BR = 16;        // rows per block
constexpr int JR_NW = 8;         // waves
constexpr int JR_SLOTS = JR_NW + 1;

__device__ __forceinline__ void jacobi_rows_reg256_body(float4 (*xch)[JR_BR][64], const int walker, float *__restrict__ Mg, long wM, int m,
                                                        int len, int ld, int max_sweeps, int *__restrict__ sweeps_out,
                                                        const int *__restrict__ mdyn, int mdyn_mul, int skip_small);

__global__ __launch_bounds__(512) void jacobi_rows_reg256_kernel(float *__restrict__ Mg, long wM, int m, int len, int ld,
                                                                 int max_sweeps, int *__restrict__ sweeps_out,
                                                                 const int *__restrict__ mdyn, int mdyn_mul,
                                                                 int skip_small) {
  __shared__ float4 xch[JR_SLOTS][JR_BR][64];   // 9 x 16 KiB
  jacobi_rows_reg256_body(xch, blockIdx.x, Mg, wM, m, len, ld, max_sweeps, sweeps_out, mdyn, mdyn_mul, skip_small);
}

// The same kernel over a LIST of walkers (list[0 .. *count)), a small fixed grid whose blocks walk the list: when the list is
// empty (the usual case where it is used: walkers whose compressed factor kept more than 128 rows) the launch costs a few
// microseconds instead of the ~0.9 ms that 2048 blocks of 144 KB LDS cost even when every block returns at once.
__global__ __launch_bounds__(512) void jacobi_rows_reg256_list_kernel(float *__restrict__ Mg, long wM, int m, int len, int ld,
                                                                      int max_sweeps, int *__restrict__ sweeps_out,
                                                                      const int *__restrict__ mdyn, int mdyn_mul, int skip_small,
                                                                      const int *__restrict__ list, const int *__restrict__ count) {
  __shared__ float4 xch[JR_SLOTS][JR_BR][64];
  const int n = *count;
  for (int q = blockIdx.x; q < n; q += gridDim.x) {
    jacobi_rows_reg256_body(xch, list[q], Mg, wM, m, len, ld, max_sweeps, sweeps_out, mdyn, mdyn_mul, skip_small);
    __syncthreads();
  }
}

__device__ __forceinline__ void jacobi_rows_reg256_body(float4 (*xch)[JR_BR][64], const int walker, float *__restrict__ Mg, long wM, int m,
                                                        int len, int ld, int max_sweeps, int *__restrict__ sweeps_out,
                                                        const int *__restrict__ mdyn, int mdyn_mul, int skip_small) {
  if (mdyn) m = max(0, min(m, mdyn[walker] * mdyn_mul));   // rows that exist for this walker
  if (skip_small && m <= max(skip_small, 2 * JR_BR)) return;   // skip_small = 1: the one-wave kernels took this walker;
                                                                // > 32: also the walkers of the preconditioned mid route
  __shared__ float xnorm[JR_SLOTS][JR_BR];
  __shared__ float s_n2[256];
  __shared__ short s_perm[256];
  __shared__ double s_fro[JR_NW];
  __shared__ int s_rot, s_live0;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  float *M = Mg + (long)walker * wM;

  // ---- prepass: row norms, noise floor, rows ranked by decreasing norm ----
  // Row order is free (select_rows_kernel sorts afterwards), so the rows are taken in rank
  // order: the live rows (norm above the floor) fill the first blocks and the tournament runs
  // over those blocks only -- cost ~ (numerical rank)^2 instead of m^2.
  if (tid == 0) s_live0 = 0;
  for (int r = w; r < 256; r += JR_NW) {
    float n2 = 0.f;
    if (r < m) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = 4 * lane + q;
        const float v = c < len ? M[(long)r * ld + c] : 0.f;
        n2 = fmaf(v, v, n2);
      }
    }
    n2 = jr_allsum(n2);
    if (lane == 0) s_n2[r] = n2;
  }
  __syncthreads();
  {
    double f = 0.0;
    for (int r = tid; r < 256; r += 512) f += (double)s_n2[r];
    f = wave_sum(f);
    if (lane == 0) s_fro[w] = f;
    __syncthreads();
    if (tid == 0) { double t = 0.0; for (int k = 0; k < JR_NW; ++k) t += s_fro[k]; s_fro[0] = t; }
    __syncthreads();
  }
  const float floor2 = (float)(NOISE_C * NOISE_C * (double)Eps<float>::v * (double)Eps<float>::v * s_fro[0]);
  const float tol2 = 4.f * (float)len * Eps<float>::v * Eps<float>::v;   // (2 sqrt(len) eps)^2
  if (tid < 256) {
    const float v = s_n2[tid];
    int rk = 0;
    for (int q = 0; q < 256; ++q) {
      const float u = s_n2[q];
      rk += (u > v) || (u == v && q < tid);
    }
    s_perm[rk] = (short)tid;
    if (v > floor2) atomicAdd(&s_live0, 1);
  }
  __syncthreads();
  const int live0 = s_live0;
  const int nbl = (live0 + JR_BR - 1) / JR_BR;            // live blocks
  const int nwv = nbl <= 2 ? 1 : (nbl + 1) / 2;           // waves taking part in the tournament
  const bool active = w < nwv;

  JrRow a[JR_BR], b[JR_BR];
  float na[JR_BR], nb[JR_BR];
  auto load_block = [&](JrRow(&blk)[JR_BR], int bid) {
#pragma unroll
    for (int i = 0; i < JR_BR; ++i) {
      const int pos = bid * JR_BR + i;
      const int r = (active && pos < nbl * JR_BR) ? (int)s_perm[pos] : m;
      float v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = 4 * lane + q;
        v[q] = (r < m && c < len) ? M[(long)r * ld + c] : 0.f;
      }
      blk[i].lo = jr_f2{v[0], v[1]};
      blk[i].hi = jr_f2{v[2], v[3]};
    }
  };
  load_block(a, w);
  load_block(b, nwv + w);

  int sweep = 0;
  for (; sweep < max_sweeps; ++sweep) {
    if (tid == 0) s_rot = 0;
    // exact norms at sweep start
#pragma unroll
    for (int i = 0; i < JR_BR; ++i) {
      na[i] = jr_allsum(jr_dot(a[i], a[i]));
      nb[i] = jr_allsum(jr_dot(b[i], b[i]));
    }
    int rot = 0;
    // pairs inside each block, once per sweep: circle method on the 16 rows of a block.  The
    // rows are physically rotated in registers so that the pairing is always (i, 15-i) with
    // static indices inside a ROLLED loop (a fully unrolled tournament makes the scheduler
    // hoist across rounds and spill).
#pragma unroll 1
    for (int r = 0; r < (active ? JR_BR - 1 : 0); ++r) {
      float ga[JR_BR / 2], gb[JR_BR / 2];
#pragma unroll
      for (int p = 0; p < JR_BR / 2; ++p) {
        ga[p] = jr_allsum(jr_dot(a[p], a[JR_BR - 1 - p]));
        gb[p] = jr_allsum(jr_dot(b[p], b[JR_BR - 1 - p]));
      }
#pragma unroll
      for (int p = 0; p < JR_BR / 2; ++p) {
        rot += jr_apply(a[p], a[JR_BR - 1 - p], na[p], na[JR_BR - 1 - p], ga[p], tol2, floor2);
        rot += jr_apply(b[p], b[JR_BR - 1 - p], nb[p], nb[JR_BR - 1 - p], gb[p], tol2, floor2);
      }
      {   // rows 1..15 move one place round the circle (row 0 fixed); 15 rounds = identity
        const JrRow ta = a[JR_BR - 1], tb = b[JR_BR - 1];
        const float fa = na[JR_BR - 1], fb = nb[JR_BR - 1];
#pragma unroll
        for (int i = JR_BR - 1; i >= 2; --i) { a[i] = a[i - 1]; b[i] = b[i - 1]; na[i] = na[i - 1]; nb[i] = nb[i - 1]; }
        a[1] = ta; b[1] = tb; na[1] = fa; nb[1] = fb;
      }
    }
    __syncthreads();   // s_rot reset visible; LDS slots free
    for (int sr = 0; sr < 2 * nwv - 1; ++sr) {
      // A block whose rows are all below the noise floor takes part in no rotation (jr_apply
      // would return the identity for every one of its pairs): skip its 256 pairs outright.
      float mxa = na[0], mxb = nb[0];
#pragma unroll
      for (int i = 1; i < JR_BR; ++i) { mxa = fmaxf(mxa, na[i]); mxb = fmaxf(mxb, nb[i]); }
      const bool live = active && mxa > floor2 && mxb > floor2;
#pragma unroll 1
      for (int t = 0; t < (live ? JR_BR : 0); ++t) {
        // the 16 pairs (a[i], b[i]) of a round are disjoint: dots + reductions first, then rotations
        float g[JR_BR];
#pragma unroll
        for (int i = 0; i < JR_BR; ++i) g[i] = jr_allsum(jr_dot(a[i], b[i]));
#pragma unroll
        for (int i = 0; i < JR_BR; ++i) rot += jr_apply(a[i], b[i], na[i], nb[i], g[i], tol2, floor2);
        {   // b rows shift by one place (cyclic); 16 rounds = identity
          const JrRow tb = b[0];
          const float fb = nb[0];
#pragma unroll
          for (int i = 0; i < JR_BR - 1; ++i) { b[i] = b[i + 1]; nb[i] = nb[i + 1]; }
          b[JR_BR - 1] = tb; nb[JR_BR - 1] = fb;
        }
      }
      if (nwv == 1) continue;   // two blocks only: nothing moves
      // ---- move the blocks one step round the circle (top[0] fixed) ----
      // phase 1: bottom[j] <- bottom[j+1], bottom[last] <- top[last]; old bottom[0] stays in slot 0
      if (active) {
#pragma unroll
        for (int i = 0; i < JR_BR; ++i) xch[w][i][lane] = make_float4(b[i].lo.x, b[i].lo.y, b[i].hi.x, b[i].hi.y);
        if (lane == 0) {
#pragma unroll
          for (int i = 0; i < JR_BR; ++i) xnorm[w][i] = nb[i];
        }
        if (w == nwv - 1) {
#pragma unroll
          for (int i = 0; i < JR_BR; ++i) xch[nwv][i][lane] = make_float4(a[i].lo.x, a[i].lo.y, a[i].hi.x, a[i].hi.y);
          if (lane == 0) {
#pragma unroll
            for (int i = 0; i < JR_BR; ++i) xnorm[nwv][i] = na[i];
          }
        }
      }
      __syncthreads();
      if (active) {
        const int src = (w == nwv - 1) ? nwv : w + 1;
#pragma unroll
        for (int i = 0; i < JR_BR; ++i) {
          const float4 t4 = xch[src][i][lane];
          b[i].lo = jr_f2{t4.x, t4.y};
          b[i].hi = jr_f2{t4.z, t4.w};
          nb[i] = xnorm[src][i];
        }
      }
      __syncthreads();
      // phase 2: top[j] <- top[j-1] (j >= 2), top[1] <- old bottom[0] (still in slot 0)
      if (active && w >= 1 && w <= nwv - 2) {
#pragma unroll
        for (int i = 0; i < JR_BR; ++i) xch[w][i][lane] = make_float4(a[i].lo.x, a[i].lo.y, a[i].hi.x, a[i].hi.y);
        if (lane == 0) {
#pragma unroll
          for (int i = 0; i < JR_BR; ++i) xnorm[w][i] = na[i];
        }
      }
      __syncthreads();
      if (active && w >= 1) {
        const int src = w - 1;   // w == 1 reads slot 0 = old bottom[0]
#pragma unroll
        for (int i = 0; i < JR_BR; ++i) {
          const float4 t4 = xch[src][i][lane];
          a[i].lo = jr_f2{t4.x, t4.y};
          a[i].hi = jr_f2{t4.z, t4.w};
          na[i] = xnorm[src][i];
        }
      }
      __syncthreads();
    }
    if (lane == 0 && rot) atomicAdd(&s_rot, rot);
    __syncthreads();
    const int total = s_rot;
    __syncthreads();
    if (total == 0) { ++sweep; break; }
  }
  // 2*nwv-1 exchanges = one full turn of the circle: at the end of every sweep each block is back
  // in its initial (wave, slot), so every row returns to the address it was loaded from
  auto store_block = [&](JrRow(&blk)[JR_BR], int bid) {
#pragma unroll
    for (int i = 0; i < JR_BR; ++i) {
      const int pos = bid * JR_BR + i;
      const int r = (active && pos < nbl * JR_BR) ? (int)s_perm[pos] : m;
      const float v[4] = {blk[i].lo.x, blk[i].lo.y, blk[i].hi.x, blk[i].hi.y};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = 4 * lane + q;
        if (r < m && c < len) M[(long)r * ld + c] = v[q];
      }
    }
  };
  store_block(a, w);
  store_block(b, nwv + w);
  if (tid == 0 && sweeps_out) sweeps_out[walker] = sweep | (live0 << 8);
}


// ---------------------------------------------------------------------------------------------
// The same tournament for the PRECONDITIONED mid-rank blocks (Engine::absorb_impl, "mid" route): the matrix is the
// triangular factor B (B^T B = M M^T, at most 128 x 128) instead of M itself, so rows are at most 128 long (CPL = 2
// columns per lane) and NW waves x 2 blocks x 16 rows cover it: a quarter of the registers and LDS of the 256 x 256
// kernel, several walkers resident per CU.  Skips walkers without rows (mdyn == 0: not on this route).
// A row piece is stored as register PAIRS (round 4, late): the rotation runs on v_pk_mul / v_pk_fma_f32, which take 64-bit aligned
// register pairs -- with the columns as separate floats the allocator scattered them and gathered a pair in front of every packed
// instruction (a third of the VALU instructions of the tournament loops were v_mov).
template <int CPL> struct JrRowT {
  static_assert(CPL % 2 == 0, "columns per lane come in pairs");
  jr_f2 p[CPL / 2];
  __device__ __forceinline__ float get(int q) const { return p[q >> 1][q & 1]; }
  __device__ __forceinline__ void set(int q, float x) { p[q >> 1][q & 1] = x; }
};

template <int CPL>
__device__ __forceinline__ float jrx_dot(const JrRowT<CPL> &x, const JrRowT<CPL> &y) {
  jr_f2 acc = x.p[0] * y.p[0];
#pragma unroll
  for (int k = 1; k < CPL / 2; ++k) acc = __builtin_elementwise_fma(x.p[k], y.p[k], acc);
  return acc[0] + acc[1];
}

template <int CPL>
__device__ __forceinline__ int jrx_apply(JrRowT<CPL> &x, JrRowT<CPL> &y, float &nx, float &ny, const float g, const float tol2,
                                         const float floor2) {
  const bool go = g * g > tol2 * nx * ny && nx > floor2 && ny > floor2;
  #ifndef JRX_BRANCHFREE   // (-DJRX_BRANCHFREE: no early exit -- 13 % fewer instructions in the tournament loops and the four pairs of a step free
                        //  to overlap their rcp / sqrt / rsq chains, against the arithmetic of the pairs no lane rotates; NOT measured: DESIGN 8)
  if (!__any(go)) return 0;
#endif
  const float zeta = (ny - nx) * __builtin_amdgcn_rcpf(2.f * g);
  const float az = fabsf(zeta);
  float t = copysignf(__builtin_amdgcn_rcpf(az + __builtin_amdgcn_sqrtf(fmaf(az, az, 1.f))), zeta);
  t = go ? t : 0.f;
  const float cs = __builtin_amdgcn_rsqf(fmaf(t, t, 1.f)), sn = cs * t;
  const jr_f2 cs2 = {cs, cs}, sn2 = {sn, sn};
#pragma unroll
  for (int k = 0; k < CPL / 2; ++k) {                  // both updates can overwrite their own operand: no copies at the join below
    const jr_f2 xv = x.p[k], yv = y.p[k];
    const jr_f2 sy = sn2 * yv, sx = sn2 * xv;
    x.p[k] = __builtin_elementwise_fma(cs2, xv, -sy);
    y.p[k] = __builtin_elementwise_fma(cs2, yv, sx);
  }
  const float tg = t * g;
  nx = fmaxf(nx - tg, 0.f);
  ny = ny + tg;
  return go ? 1 : 0;
}

// (the 64-lanes-per-row tournament kernel of round 3, jacobi_rows_regx_kernel, was replaced by the sixteen-lanes-per-row form below
// and removed in round 6; the row type and the pair helpers above are shared with it)

// ---------------------------------------------------------------------------------------------
// The mid-route tournament with SIXTEEN LANES PER ROW.  The rows of the triangular factor are at most 128 long: spread over
// 64 lanes (the kernel above) a pair costs two FMAs per lane and a six-step 64-lane reduction -- the reduction is the
// kernel.  Here a row lives in one DPP row of 16 lanes (CPL = 8 columns per lane), a wave instruction works on FOUR pairs
// side by side (one per 16-lane group) and a dot product ends with four DPP steps inside the group.  The players of the
// tournament are the 16-lane groups: 4 NW players, two blocks of JG_RB = 4 rows each (NW = 2: up to 64 rows, NW = 4: 128);
// blocks move between players through LDS exactly as the wave-sized blocks of the kernel above do (same circle method, same
// rotation, threshold, noise floor and sorting by norm), only the unit is a quarter of a wave.
constexpr int JG_RB = 4;

__device__ __forceinline__ float jg_sum16(float v) {
  v = jr_dpp_add<0xB1>(v);    // quad_perm [1,0,3,2]
  v = jr_dpp_add<0x4E>(v);    // quad_perm [2,3,0,1]
  v = jr_dpp_add<0x141>(v);   // row_half_mirror
  v = jr_dpp_add<0x140>(v);   // row_mirror
  return v;
}

template <int NW, int CPL>
__global__ __launch_bounds__(NW * 64, CPL <= 8 ? 4 : 2) void jacobi_rows_grp_kernel(float *__restrict__ Mg, long wM, int m, int len, int ld,
                                                                  int max_sweeps, int *__restrict__ sweeps_out,
                                                                  const int *__restrict__ mdyn, int mdyn_mul,
                                                                  int lo_rows = 0, int hi_rows = 1 << 30) {
  // lo_rows < rows <= min(hi_rows, MAXR): the size class of this instantiation (other launches take the rest); the row length
  // covered is 16 CPL columns -- a square factor of r rows is given to the instantiation whose CPL just covers r
  constexpr int NP = 4 * NW, SLOTS = NP + 1, MAXR = NP * 2 * JG_RB, NT = NW * 64;
  extern __shared__ float jg_dyn[];                       // exchange slots (dynamic: 70 KB at NW = 4, CPL = 16)
  float (*xch)[JG_RB][CPL][16] = reinterpret_cast<float (*)[JG_RB][CPL][16]>(jg_dyn);   // [SLOTS][JG_RB][CPL][16]
  __shared__ float xnorm[SLOTS][JG_RB];
  __shared__ float s_n2[MAXR];
  __shared__ short s_perm[MAXR];
  __shared__ double s_fro[NW];
  __shared__ int s_rot, s_live0;
  if (mdyn) m = max(0, min(m, mdyn[blockIdx.x] * mdyn_mul));
  if (m <= lo_rows || m > MAXR || m > hi_rows) return;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l16 = lane & 15;
  const int w = wv * 4 + (lane >> 4);                     // the player this lane belongs to
  float *M = Mg + (long)blockIdx.x * wM;
  if (tid == 0) s_live0 = 0;
  for (int r = wv; r < MAXR; r += NW) {                   // squared row norms: one wave per row
    float n2 = 0.f;
    if (r < m)
      for (int c = lane; c < len; c += 64) { const float v = M[(long)r * ld + c]; n2 = fmaf(v, v, n2); }
    n2 = jr_allsum(n2);
    if (lane == 0) s_n2[r] = n2;
  }
  __syncthreads();
  {
    double f = 0.0;
    for (int r = tid; r < MAXR; r += NT) f += (double)s_n2[r];
    f = wave_sum(f);
    if (lane == 0) s_fro[wv] = f;
    __syncthreads();
    if (tid == 0) { double t = 0.0; for (int k = 0; k < NW; ++k) t += s_fro[k]; s_fro[0] = t; }
    __syncthreads();
  }
  const float floor2 = (float)(NOISE_C * NOISE_C * (double)Eps<float>::v * (double)Eps<float>::v * s_fro[0]);
  const float tol2 = 4.f * (float)len * Eps<float>::v * Eps<float>::v;
  for (int r = tid; r < MAXR; r += NT) {                  // rows sorted by norm: the live ones come first
    const float v = s_n2[r];
    int rk = 0;
    for (int q = 0; q < MAXR; ++q) {
      const float u = s_n2[q];
      rk += (u > v) || (u == v && q < r);
    }
    s_perm[rk] = (short)r;
    if (v > floor2) atomicAdd(&s_live0, 1);
  }
  __syncthreads();
  const int live0 = s_live0;
  const int nbl = (live0 + JG_RB - 1) / JG_RB;            // blocks that hold live rows
  const int np = nbl <= 2 ? 1 : (nbl + 1) / 2;            // players that take part
  const bool active = w < np;

  JrRowT<CPL> a[JG_RB], b[JG_RB];
  float na[JG_RB], nb[JG_RB];
  auto load_block = [&](JrRowT<CPL>(&blk)[JG_RB], int bid) {
#pragma unroll
    for (int i = 0; i < JG_RB; ++i) {
      const int pos = bid * JG_RB + i;
      const int r = (active && pos < nbl * JG_RB) ? (int)s_perm[pos] : m;
#pragma unroll
      for (int q = 0; q < CPL; ++q) {
        const int c = CPL * l16 + q;
        blk[i].set(q, (r < m && c < len) ? M[(long)r * ld + c] : 0.f);
      }
    }
  };
  load_block(a, w);
  load_block(b, np + w);

  int sweep = 0;
  for (; sweep < max_sweeps; ++sweep) {
    if (tid == 0) s_rot = 0;
#pragma unroll
    for (int i = 0; i < JG_RB; ++i) {
      na[i] = jg_sum16(jrx_dot<CPL>(a[i], a[i]));
      nb[i] = jg_sum16(jrx_dot<CPL>(b[i], b[i]));
    }
    int rot = 0;
    // pairs inside the two blocks of a player (circle method on JG_RB rows; a player without rows holds zeros: no rotation)
#pragma unroll
    for (int r = 0; r < JG_RB - 1; ++r) {
      float ga[JG_RB / 2], gb[JG_RB / 2];
#pragma unroll
      for (int p = 0; p < JG_RB / 2; ++p) {
        ga[p] = jg_sum16(jrx_dot<CPL>(a[p], a[JG_RB - 1 - p]));
        gb[p] = jg_sum16(jrx_dot<CPL>(b[p], b[JG_RB - 1 - p]));
      }
#pragma unroll
      for (int p = 0; p < JG_RB / 2; ++p) {
        rot += jrx_apply<CPL>(a[p], a[JG_RB - 1 - p], na[p], na[JG_RB - 1 - p], ga[p], tol2, floor2);
        rot += jrx_apply<CPL>(b[p], b[JG_RB - 1 - p], nb[p], nb[JG_RB - 1 - p], gb[p], tol2, floor2);
      }
      {
        const JrRowT<CPL> ta = a[JG_RB - 1], tb = b[JG_RB - 1];
        const float fa = na[JG_RB - 1], fb = nb[JG_RB - 1];
#pragma unroll
        for (int i = JG_RB - 1; i >= 2; --i) { a[i] = a[i - 1]; b[i] = b[i - 1]; na[i] = na[i - 1]; nb[i] = nb[i - 1]; }
        a[1] = ta; b[1] = tb; na[1] = fa; nb[1] = fb;
      }
    }
    __syncthreads();
    for (int sr = 0; sr < 2 * np - 1; ++sr) {
      // every row of block a against every row of block b: JG_RB steps of JG_RB disjoint pairs (b turns by one row a step).
      // Unrolled (round 4): the turn of b is register renaming instead of 36 v_mov per step -- a fifth of the instructions of the
      // kernel were moves; Jacobi category of the real leg 1 745 -> 1 602 ms per two steps (the three-step loop above stays rolled:
      // unrolled too it spills 17 registers at the four blocks per CU of the 8-column form for 1 % more)
#pragma unroll
      for (int t = 0; t < JG_RB; ++t) {
        float g[JG_RB];
#pragma unroll
        for (int i = 0; i < JG_RB; ++i) g[i] = jg_sum16(jrx_dot<CPL>(a[i], b[i]));
#pragma unroll
        for (int i = 0; i < JG_RB; ++i) rot += jrx_apply<CPL>(a[i], b[i], na[i], nb[i], g[i], tol2, floor2);
        {
          const JrRowT<CPL> tb = b[0];
          const float fb = nb[0];
#pragma unroll
          for (int i = 0; i < JG_RB - 1; ++i) { b[i] = b[i + 1]; nb[i] = nb[i + 1]; }
          b[JG_RB - 1] = tb; nb[JG_RB - 1] = fb;
        }
      }
      if (np == 1) continue;
      // the circle turns by one player: bottom blocks move down (w <- w + 1, the last takes the top of the last), top
      // blocks move up (w <- w - 1, player 1 takes the old bottom of player 0); player 0 keeps its top block
      if (active) {
#pragma unroll
        for (int i = 0; i < JG_RB; ++i)
#pragma unroll
          for (int q = 0; q < CPL; ++q) xch[w][i][q][l16] = b[i].get(q);
        if (l16 == 0) {
#pragma unroll
          for (int i = 0; i < JG_RB; ++i) xnorm[w][i] = nb[i];
        }
        if (w == np - 1) {
#pragma unroll
          for (int i = 0; i < JG_RB; ++i)
#pragma unroll
            for (int q = 0; q < CPL; ++q) xch[np][i][q][l16] = a[i].get(q);
          if (l16 == 0) {
#pragma unroll
            for (int i = 0; i < JG_RB; ++i) xnorm[np][i] = na[i];
          }
        }
      }
      __syncthreads();
      if (active) {
        const int src = (w == np - 1) ? np : w + 1;
#pragma unroll
        for (int i = 0; i < JG_RB; ++i) {
#pragma unroll
          for (int q = 0; q < CPL; ++q) b[i].set(q, xch[src][i][q][l16]);
          nb[i] = xnorm[src][i];
        }
      }
      __syncthreads();
      if (active && w >= 1 && w <= np - 2) {
#pragma unroll
        for (int i = 0; i < JG_RB; ++i)
#pragma unroll
          for (int q = 0; q < CPL; ++q) xch[w][i][q][l16] = a[i].get(q);
        if (l16 == 0) {
#pragma unroll
          for (int i = 0; i < JG_RB; ++i) xnorm[w][i] = na[i];
        }
      }
      __syncthreads();
      if (active && w >= 1) {
        const int src = w - 1;   // w == 1 reads slot 0 = old bottom[0]
#pragma unroll
        for (int i = 0; i < JG_RB; ++i) {
#pragma unroll
          for (int q = 0; q < CPL; ++q) a[i].set(q, xch[src][i][q][l16]);
          na[i] = xnorm[src][i];
        }
      }
      __syncthreads();
    }
    if (l16 == 0 && rot) atomicAdd(&s_rot, rot);
    __syncthreads();
    const int total = s_rot;
    __syncthreads();
    if (total == 0) { ++sweep; break; }
  }
  // 2 np - 1 exchanges = one full turn of the circle: every block is back with the player and slot it was loaded into
  auto store_block = [&](JrRowT<CPL>(&blk)[JG_RB], int bid) {
#pragma unroll
    for (int i = 0; i < JG_RB; ++i) {
      const int pos = bid * JG_RB + i;
      const int r = (active && pos < nbl * JG_RB) ? (int)s_perm[pos] : m;
#pragma unroll
      for (int q = 0; q < CPL; ++q) {
        const int c = CPL * l16 + q;
        if (r < m && c < len) M[(long)r * ld + c] = blk[i].get(q);
      }
    }
  };
  store_block(a, w);
  store_block(b, np + w);
  if (tid == 0 && sweeps_out) sweeps_out[blockIdx.x] = sweep | (live0 << 8);
}

template <int NW, int CPL>
inline void launch_jacobi_grp(hipStream_t s, int nbatch, float *M, long wM, int m, int len, int ld, int max_sweeps, int *sweeps_out,
                              const int *mdyn, int mdyn_mul, int lo_rows, int hi_rows = 1 << 30) {
  const size_t smem = sizeof(float) * (size_t)(4 * NW + 1) * JG_RB * CPL * 16;
  allow_dynamic_lds(reinterpret_cast<const void *>(&jacobi_rows_grp_kernel<NW, CPL>), smem);
  hipLaunchKernelGGL((jacobi_rows_grp_kernel<NW, CPL>), dim3(nbatch), dim3(NW * 64), smem, s, M, wM, m, len, ld, max_sweeps, sweeps_out,
                     mdyn, mdyn_mul, lo_rows, hi_rows);
}

// ---------------------------------------------------------------------------------------------
// Small-rank variant: a walker whose carry has at most 32 existing rows (rank-adaptive absorption:
// the usual case away from full rank) is a 32 x 256 problem that ONE wave holds in registers, so
// four walkers share a 256-thread workgroup and nothing crosses LDS or a workgroup barrier.  The
// launch covers every walker; a wave whose walker has more rows returns at once and the 8-wave
// kernel above (which returns at once for the small walkers) takes it.  Same rotation, threshold
// and noise floor as above.
constexpr int JR_SMALL_ROWS = 2 * JR_BR;

// circle-method sweep over the first NB rows of one block (NB even): NB-1 rounds of NB/2 disjoint
// pairs (p, NB-1-p); rows 1..NB-1 rotate physically so that the indices stay static
template <int NB>
__device__ __forceinline__ int jr_intra(JrRow (&a)[JR_BR], float (&na)[JR_BR], const float tol2, const float floor2) {
  int rot = 0;
#pragma unroll 1
  for (int r = 0; r < NB - 1; ++r) {
    float ga[NB / 2];
#pragma unroll
    for (int p = 0; p < NB / 2; ++p) ga[p] = jr_allsum(jr_dot(a[p], a[NB - 1 - p]));
#pragma unroll
    for (int p = 0; p < NB / 2; ++p) rot += jr_apply(a[p], a[NB - 1 - p], na[p], na[NB - 1 - p], ga[p], tol2, floor2);
    const JrRow ta = a[NB - 1];
    const float fa = na[NB - 1];
#pragma unroll
    for (int i = NB - 1; i >= 2; --i) { a[i] = a[i - 1]; na[i] = na[i - 1]; }
    a[1] = ta; na[1] = fa;
  }
  return rot;
}

__global__ __launch_bounds__(256) void jacobi_rows_small_kernel(float *__restrict__ Mg, long wM, int m, int len, int ld,
                                                                int max_sweeps, int *__restrict__ sweeps_out,
                                                                const int *__restrict__ mdyn, int mdyn_mul, int nwalkers,
                                                                int skip_tiny = 0) {
  const int lane = threadIdx.x & 63;
  const int walker = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (walker >= nwalkers) return;
  const int mm = mdyn ? max(0, min(m, mdyn[walker] * mdyn_mul)) : m;
  if (mm > JR_SMALL_ROWS || (skip_tiny && mm <= JR_BR)) return;   // jacobi_rows_tiny_kernel took the <= 16-row walkers
  float *M = Mg + (long)walker * wM;
  JrRow a[JR_BR], b[JR_BR];
  float na[JR_BR], nb[JR_BR];
  auto load_row = [&](int r) {
    JrRow x;
    float v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = 4 * lane + q;
      v[q] = (r < mm && c < len) ? M[(long)r * ld + c] : 0.f;
    }
    x.lo = jr_f2{v[0], v[1]};
    x.hi = jr_f2{v[2], v[3]};
    return x;
  };
  const bool has_b = mm > JR_BR;
#pragma unroll
  for (int i = 0; i < JR_BR; ++i) a[i] = load_row(i);
#pragma unroll
  for (int i = 0; i < JR_BR; ++i) b[i] = load_row(has_b ? JR_BR + i : mm);
  double fro = 0.0;
#pragma unroll
  for (int i = 0; i < JR_BR; ++i) {
    na[i] = jr_allsum(jr_dot(a[i], a[i]));
    nb[i] = jr_allsum(jr_dot(b[i], b[i]));
    fro += (double)na[i] + (double)nb[i];
  }
  const float floor2 = (float)(NOISE_C * NOISE_C * (double)Eps<float>::v * (double)Eps<float>::v * fro);
  const float tol2 = 4.f * (float)len * Eps<float>::v * Eps<float>::v;
  int sweep = 0;
  for (; sweep < max_sweeps; ++sweep) {
    if (sweep) {   // exact norms at sweep start
#pragma unroll
      for (int i = 0; i < JR_BR; ++i) na[i] = jr_allsum(jr_dot(a[i], a[i]));
      if (has_b) {
#pragma unroll
        for (int i = 0; i < JR_BR; ++i) nb[i] = jr_allsum(jr_dot(b[i], b[i]));
      }
    }
    int rot = 0;
    if (mm <= 8) {
      rot += jr_intra<8>(a, na, tol2, floor2);
    } else {
      rot += jr_intra<JR_BR>(a, na, tol2, floor2);
      if (has_b) {
        rot += jr_intra<JR_BR>(b, nb, tol2, floor2);
#pragma unroll 1
        for (int t = 0; t < JR_BR; ++t) {
          float g[JR_BR];
#pragma unroll
          for (int i = 0; i < JR_BR; ++i) g[i] = jr_allsum(jr_dot(a[i], b[i]));
#pragma unroll
          for (int i = 0; i < JR_BR; ++i) rot += jr_apply(a[i], b[i], na[i], nb[i], g[i], tol2, floor2);
          const JrRow tb = b[0];
          const float fb = nb[0];
#pragma unroll
          for (int i = 0; i < JR_BR - 1; ++i) { b[i] = b[i + 1]; nb[i] = nb[i + 1]; }
          b[JR_BR - 1] = tb; nb[JR_BR - 1] = fb;
        }
      }
    }
    if (rot == 0) { ++sweep; break; }
  }
  // every rotation loop above is a whole number of turns: each row is back in its own register
  auto store_row = [&](const JrRow &x, int r) {
    const float v[4] = {x.lo.x, x.lo.y, x.hi.x, x.hi.y};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = 4 * lane + q;
      if (r < mm && c < len) M[(long)r * ld + c] = v[q];
    }
  };
#pragma unroll
  for (int i = 0; i < JR_BR; ++i) store_row(a[i], i);
  if (has_b) {
#pragma unroll
    for (int i = 0; i < JR_BR; ++i) store_row(b[i], JR_BR + i);
  }
  if (lane == 0 && sweeps_out) sweeps_out[walker] = sweep | (mm << 8);
}


// At most 16 existing rows (the usual numerical rank of the carry on smooth states): one block of
// 16 rows per wave, a third of the registers of the 32-row kernel, so that every walker of a large
// batch is resident at once.  The tournament runs over 8, 12 or 16 rows.
__global__ __launch_bounds__(256, 3) void jacobi_rows_tiny_kernel(float *__restrict__ Mg, long wM, int m, int len, int ld,
                                                               int max_sweeps, int *__restrict__ sweeps_out,
                                                               const int *__restrict__ mdyn, int mdyn_mul, int nwalkers) {
  const int lane = threadIdx.x & 63;
  const int walker = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (walker >= nwalkers) return;
  const int mm = mdyn ? max(0, min(m, mdyn[walker] * mdyn_mul)) : m;
  if (mm > JR_BR) return;
  float *M = Mg + (long)walker * wM;
  JrRow a[JR_BR];
  float na[JR_BR];
  double fro = 0.0;
#pragma unroll
  for (int i = 0; i < JR_BR; ++i) {
    float v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = 4 * lane + q;
      v[q] = (i < mm && c < len) ? M[(long)i * ld + c] : 0.f;
    }
    a[i].lo = jr_f2{v[0], v[1]};
    a[i].hi = jr_f2{v[2], v[3]};
  }
#pragma unroll
  for (int i = 0; i < JR_BR; ++i) {
    na[i] = jr_allsum(jr_dot(a[i], a[i]));
    fro += (double)na[i];
  }
  const float floor2 = (float)(NOISE_C * NOISE_C * (double)Eps<float>::v * (double)Eps<float>::v * fro);
  const float tol2 = 4.f * (float)len * Eps<float>::v * Eps<float>::v;
  int sweep = 0;
  for (; sweep < max_sweeps; ++sweep) {
    if (sweep) {
#pragma unroll
      for (int i = 0; i < JR_BR; ++i) na[i] = jr_allsum(jr_dot(a[i], a[i]));
    }
    int rot;
    if (mm <= 6) rot = jr_intra<6>(a, na, tol2, floor2);
    else if (mm <= 8) rot = jr_intra<8>(a, na, tol2, floor2);
    else if (mm <= 10) rot = jr_intra<10>(a, na, tol2, floor2);
    else if (mm <= 12) rot = jr_intra<12>(a, na, tol2, floor2);
    else if (mm <= 14) rot = jr_intra<14>(a, na, tol2, floor2);
    else rot = jr_intra<JR_BR>(a, na, tol2, floor2);
    if (rot == 0) { ++sweep; break; }
  }
#pragma unroll
  for (int i = 0; i < JR_BR; ++i) {
    const float v[4] = {a[i].lo.x, a[i].lo.y, a[i].hi.x, a[i].hi.y};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = 4 * lane + q;
      if (i < mm && c < len) M[(long)i * ld + c] = v[q];
    }
  }
  if (lane == 0 && sweeps_out) sweeps_out[walker] = sweep | (mm << 8);
}


// Four walkers per wave: a row of at most 128 elements lives in one DPP row of 16 lanes (CPL = 8 columns per lane, 4 when
// len <= 64), a dot product ends with four DPP steps inside the lane group and every wave instruction works on four
// walkers' pairs at once (the two-walkers-per-wave kernel it replaced, removed in round 6: five steps, two walkers).  Same rotation, threshold and noise floor.
template <int NB, int CPL>
__device__ __forceinline__ int jr_intra16(JrRowT<CPL> (&a)[JR_BR], float (&na)[JR_BR], const float tol2, const float floor2) {
  int rot = 0;
#ifdef JR_INTRA_TWOSTEP
  // (default off, NOT measured -- DESIGN 8, lead 2.)  Two steps per trip: the second step is written with the indices of the arrangement
  // BEFORE the turn (a'[i] = a[T(i)], T(0) = 0, T(1) = NB - 1, T(i) = i - 1), then ONE double turn: half the register moves of the loop.
  auto step = [&](auto turned) {
    constexpr bool TURN = decltype(turned)::value;
    auto T = [](int i) { return !TURN ? i : (i == 0 ? 0 : (i == 1 ? NB - 1 : i - 1)); };
    float ga[NB / 2];
#pragma unroll
    for (int p = 0; p < NB / 2; ++p) ga[p] = jg_sum16(jrx_dot<CPL>(a[T(p)], a[T(NB - 1 - p)]));
#pragma unroll
    for (int p = 0; p < NB / 2; ++p) rot += jrx_apply<CPL>(a[T(p)], a[T(NB - 1 - p)], na[T(p)], na[T(NB - 1 - p)], ga[p], tol2, floor2);
  };
#pragma unroll 1
  for (int r = 0; r + 1 < NB - 1; r += 2) {
    step(std::false_type());
    step(std::true_type());
    const JrRowT<CPL> t1 = a[NB - 2], t2 = a[NB - 1];
    const float f1 = na[NB - 2], f2 = na[NB - 1];
#pragma unroll
    for (int i = NB - 1; i >= 3; --i) { a[i] = a[i - 2]; na[i] = na[i - 2]; }
    a[1] = t1; na[1] = f1; a[2] = t2; na[2] = f2;
  }
  {   // NB - 1 is odd: the last step and a single turn (NB - 1 turns in all: every row is back in its slot)
    step(std::false_type());
    const JrRowT<CPL> ta = a[NB - 1];
    const float fa = na[NB - 1];
#pragma unroll
    for (int i = NB - 1; i >= 2; --i) { a[i] = a[i - 1]; na[i] = na[i - 1]; }
    a[1] = ta; na[1] = fa;
  }
#else
#pragma unroll 1
  for (int r = 0; r < NB - 1; ++r) {
    float ga[NB / 2];
#pragma unroll
    for (int p = 0; p < NB / 2; ++p) ga[p] = jg_sum16(jrx_dot<CPL>(a[p], a[NB - 1 - p]));
#pragma unroll
    for (int p = 0; p < NB / 2; ++p) rot += jrx_apply<CPL>(a[p], a[NB - 1 - p], na[p], na[NB - 1 - p], ga[p], tol2, floor2);
    const JrRowT<CPL> ta = a[NB - 1];
    const float fa = na[NB - 1];
#pragma unroll
    for (int i = NB - 1; i >= 2; --i) { a[i] = a[i - 1]; na[i] = na[i - 1]; }
    a[1] = ta; na[1] = fa;
  }
#endif
  return rot;
}

template <int CPL>
__global__ __launch_bounds__(256, CPL <= 4 ? 3 : 2) void jacobi_rows_tiny4_kernel(float *__restrict__ Mg, long wM, int m, int len, int ld,
                                                                               int max_sweeps, int *__restrict__ sweeps_out,
                                                                               const int *__restrict__ mdyn, int mdyn_mul, int nwalkers,
                                                                               JrSelect sel = JrSelect()) {
  const int lane = threadIdx.x & 63, l16 = lane & 15;
  const int walker = blockIdx.x * 16 + (threadIdx.x >> 6) * 4 + (lane >> 4);
  const bool have = walker < nwalkers;
  int mm = have ? (mdyn ? max(0, min(m, mdyn[walker] * mdyn_mul)) : m) : 0;
  if (mm > JR_BR) mm = 0;                                     // the 32-row / 8-wave kernels take this walker
  int mm_max = max(mm, __shfl_xor(mm, 16, 64));               // wave-uniform: the four walkers of the wave
  mm_max = max(mm_max, __shfl_xor(mm_max, 32, 64));
  if (mm_max == 0) return;
  float *M = Mg + (long)(have ? walker : 0) * wM;
  JrRowT<CPL> a[JR_BR];
  float na[JR_BR];
  float fro = 0.f;
#pragma unroll
  for (int i = 0; i < JR_BR; ++i) {
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
      const int c = CPL * l16 + q;
      a[i].set(q, (i < mm && c < len) ? M[(long)i * ld + c] : 0.f);
    }
  }
#pragma unroll
  for (int i = 0; i < JR_BR; ++i) {
    na[i] = jg_sum16(jrx_dot<CPL>(a[i], a[i]));
    fro += na[i];
  }
  const float floor2 = (float)(NOISE_C * NOISE_C * (double)Eps<float>::v * (double)Eps<float>::v) * fro;
  const float tol2 = 4.f * (float)len * Eps<float>::v * Eps<float>::v;
  int sweep = 0;
  for (; sweep < max_sweeps; ++sweep) {
    if (sweep) {
#pragma unroll
      for (int i = 0; i < JR_BR; ++i)
        if (i < mm_max) na[i] = jg_sum16(jrx_dot<CPL>(a[i], a[i]));   // rows beyond every walker's count are zero
    }
    int rot;
    if (mm_max <= 6) rot = jr_intra16<6, CPL>(a, na, tol2, floor2);
    else if (mm_max <= 8) rot = jr_intra16<8, CPL>(a, na, tol2, floor2);
    else if (mm_max <= 10) rot = jr_intra16<10, CPL>(a, na, tol2, floor2);
    else if (mm_max <= 12) rot = jr_intra16<12, CPL>(a, na, tol2, floor2);
    else if (mm_max <= 14) rot = jr_intra16<14, CPL>(a, na, tol2, floor2);
    else rot = jr_intra16<JR_BR, CPL>(a, na, tol2, floor2);
    if (!__any(rot != 0)) { ++sweep; break; }
  }
  if (l16 == 0 && mm > 0 && sweeps_out) sweeps_out[walker] = sweep | (mm << 8);
  if (!sel.V) {
#pragma unroll
    for (int i = 0; i < JR_BR; ++i) {
#pragma unroll
      for (int q = 0; q < CPL; ++q) {
        const int c = CPL * l16 + q;
        if (i < mm && c < len) M[(long)i * ld + c] = a[i].get(q);
      }
    }
    return;
  }
  // ---- select: the rows are mutually orthogonal, sigma_i = |row_i| (select_rows_kernel, same rules) ----
  if (mm == 0) return;                                     // (not this kernel's walker: select_rows_kernel takes it)
  float *V = sel.V + (long)walker * sel.wV;
  double n2[JR_BR];        // f64 sums of the exact f32 squares, as select_rows_kernel forms them
  double fro2 = 0.0;
#pragma unroll
  for (int i = 0; i < JR_BR; ++i) {
    double t = 0.0;
#pragma unroll
    for (int q = 0; q < CPL; ++q) t = fma((double)a[i].get(q), (double)a[i].get(q), t);
    t += lw_dpp_f64<0xB1>(t);
    t += lw_dpp_f64<0x4E>(t);
    t += lw_dpp_f64<0x141>(t);
    t += lw_dpp_f64<0x140>(t);
    n2[i] = i < mm ? t : -1.0;                                     // rows that do not exist rank after every real row
    fro2 += i < mm ? t : 0.0;
  }
  const double nfloor2 = 4.0 * NOISE_C * NOISE_C * (double)Eps<float>::v * (double)Eps<float>::v * fro2;   // (2 NOISE_C eps |M|_F)^2
  int rank[JR_BR];
#pragma unroll
  for (int i = 0; i < JR_BR; ++i) {
    int rk = 0;
#pragma unroll
    for (int j = 0; j < JR_BR; ++j) rk += (n2[j] > n2[i]) || (n2[j] == n2[i] && j < i);
    rank[i] = rk;
  }
  int kcut = sel.k;
  if (sel.trunc_err > 0.0) {
    int kept = mm;
    double err = 0.0;
    while (kept > 0) {
      if (kept <= sel.dmin && kept <= sel.k) break;
      double sk2 = 0.0;                                      // squared singular value of rank kept - 1
#pragma unroll
      for (int i = 0; i < JR_BR; ++i) sk2 = rank[i] == kept - 1 ? n2[i] : sk2;
      const double wgt = fro2 > 0.0 ? sk2 / fro2 : 0.0;
      if (kept > sel.k || (kept > sel.dmin && err + wgt < sel.trunc_err)) { err += wgt; --kept; }
      else break;
    }
    kcut = max(kept, 1);
  }
  int klive = 0;
#pragma unroll
  for (int i = 0; i < JR_BR; ++i) {
    if (i < mm && rank[i] < kcut) {
      const bool live = n2[i] > nfloor2;
      const float inv = live ? (float)jr_rsq64((double)n2[i]) : 0.f;     // numerically zero direction -> zero row of Vt
      klive += live ? 1 : 0;
#pragma unroll
      for (int q = 0; q < CPL; ++q) {
        const int c = CPL * l16 + q;
        if (c < len) V[(long)rank[i] * len + c] = a[i].get(q) * inv;
      }
    }
  }
  // rows of Vt from the kept count to k are zero
  const int nkeep = min(kcut, mm);
  for (int e = nkeep * len + l16; e < sel.k * len; e += 16) V[e] = 0.f;
  if (l16 == 0 && sel.klive_out) sel.klive_out[walker] = klive;
}


// ---------------------------------------------------------------------------------------------
// The short-row kernel for the FLOAT64 element type: at most JR_BR rows of at most LPR * CPL elements, a row in LPR lanes
// (16: four walkers per wave, rows <= 64 long; 32: two walkers, rows <= 128), rows in registers, DPP reductions inside the
// lane group.  The f64 mode had only the general kernel (1024 threads and LDS or global memory per walker): two thirds of
// its step.  Same Hestenes rotation, threshold (4 len eps^2) and noise floor as the f32 kernels, in f64 arithmetic.
template <int CPL> struct JrRowD { double v[CPL]; };

template <int LPR>
__device__ __forceinline__ double jd_sum(double v) {
  v += lw_dpp_f64<0xB1>(v);
  v += lw_dpp_f64<0x4E>(v);
  v += lw_dpp_f64<0x141>(v);
  v += lw_dpp_f64<0x140>(v);
  if constexpr (LPR == 32) {
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const auto pl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto ph = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    v = __hiloint2double((int)ph[0], (int)pl[0]) + __hiloint2double((int)ph[1], (int)pl[1]);
  }
  return v;
}

template <int CPL>
__device__ __forceinline__ double jd_dot(const JrRowD<CPL> &x, const JrRowD<CPL> &y) {
  double p = x.v[0] * y.v[0];
#pragma unroll
  for (int q = 1; q < CPL; ++q) p = fma(x.v[q], y.v[q], p);
  return p;
}

template <int CPL>
__device__ __forceinline__ int jd_apply(JrRowD<CPL> &x, JrRowD<CPL> &y, double &nx, double &ny, const double g, const double tol2,
                                        const double floor2) {
  const bool go = g * g > tol2 * nx * ny && nx > floor2 && ny > floor2;
  if (!__any(go)) return 0;
  const double zeta = (ny - nx) * jr_rcp64(2.0 * (go ? g : 1.0));
  const double az = fabs(zeta);
  const double wz = fma(az, az, 1.0);
  double t = copysign(jr_rcp64(az + wz * jr_rsq64(wz)), zeta);
  t = go ? t : 0.0;
  const double cs = jr_rsq64(fma(t, t, 1.0)), sn = cs * t;
#pragma unroll
  for (int q = 0; q < CPL; ++q) {
    const double xv = x.v[q], yv = y.v[q];
    x.v[q] = cs * xv - sn * yv;
    y.v[q] = sn * xv + cs * yv;
  }
  const double tg = t * g;
  nx = fmax(nx - tg, 0.0);
  ny = ny + tg;
  return go ? 1 : 0;
}

template <int NB, int CPL, int LPR>
__device__ __forceinline__ int jd_intra(JrRowD<CPL> (&a)[JR_BR], double (&na)[JR_BR], const double tol2, const double floor2) {
  int rot = 0;
#pragma unroll 1
  for (int r = 0; r < NB - 1; ++r) {
    double ga[NB / 2];
#pragma unroll
    for (int p = 0; p < NB / 2; ++p) ga[p] = jd_sum<LPR>(jd_dot<CPL>(a[p], a[NB - 1 - p]));
#pragma unroll
    for (int p = 0; p < NB / 2; ++p) rot += jd_apply<CPL>(a[p], a[NB - 1 - p], na[p], na[NB - 1 - p], ga[p], tol2, floor2);
    const JrRowD<CPL> ta = a[NB - 1];
    const double fa = na[NB - 1];
#pragma unroll
    for (int i = NB - 1; i >= 2; --i) { a[i] = a[i - 1]; na[i] = na[i - 1]; }
    a[1] = ta; na[1] = fa;
  }
  return rot;
}

template <int CPL, int LPR>
__global__ __launch_bounds__(256, 2) void jacobi_rows_tiny_f64_kernel(double *__restrict__ Mg, long wM, int m, int len, int ld,
                                                                      int max_sweeps, int *__restrict__ sweeps_out,
                                                                      const int *__restrict__ mdyn, int mdyn_mul, int nwalkers) {
  constexpr int WPW = 64 / LPR;                               // walkers per wave
  const int lane = threadIdx.x & 63, ll = lane & (LPR - 1);
  const int walker = blockIdx.x * (4 * WPW) + (threadIdx.x >> 6) * WPW + lane / LPR;
  const bool have = walker < nwalkers;
  int mm = have ? (mdyn ? max(0, min(m, mdyn[walker] * mdyn_mul)) : m) : 0;
  if (mm > JR_BR) mm = 0;                                     // the general kernel takes this walker
  int mm_max = mm;
#pragma unroll
  for (int o = LPR; o < 64; o <<= 1) mm_max = max(mm_max, __shfl_xor(mm_max, o, 64));   // wave-uniform
  if (mm_max == 0) return;
  double *M = Mg + (long)(have ? walker : 0) * wM;
  JrRowD<CPL> a[JR_BR];
  double na[JR_BR];
  double fro = 0.0;
#pragma unroll
  for (int i = 0; i < JR_BR; ++i) {
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
      const int c = CPL * ll + q;
      a[i].v[q] = (i < mm && c < len) ? M[(long)i * ld + c] : 0.0;
    }
  }
#pragma unroll
  for (int i = 0; i < JR_BR; ++i) {
    na[i] = jd_sum<LPR>(jd_dot<CPL>(a[i], a[i]));
    fro += na[i];
  }
  const double floor2 = NOISE_C * NOISE_C * eps_rt<double>() * eps_rt<double>() * fro;
  const double tol2 = 4.0 * (double)len * eps_rt<double>() * eps_rt<double>();
  int sweep = 0;
  for (; sweep < max_sweeps; ++sweep) {
    if (sweep) {
#pragma unroll
      for (int i = 0; i < JR_BR; ++i)
        if (i < mm_max) na[i] = jd_sum<LPR>(jd_dot<CPL>(a[i], a[i]));
    }
    int rot;
    if (mm_max <= 4) rot = jd_intra<4, CPL, LPR>(a, na, tol2, floor2);
    else if (mm_max <= 8) rot = jd_intra<8, CPL, LPR>(a, na, tol2, floor2);
    else if (mm_max <= 12) rot = jd_intra<12, CPL, LPR>(a, na, tol2, floor2);
    else rot = jd_intra<JR_BR, CPL, LPR>(a, na, tol2, floor2);
    if (!__any(rot != 0)) { ++sweep; break; }
  }
#pragma unroll
  for (int i = 0; i < JR_BR; ++i) {
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
      const int c = CPL * ll + q;
      if (i < mm && c < len) M[(long)i * ld + c] = a[i].v[q];
    }
  }
  if (ll == 0 && mm > 0 && sweeps_out) sweeps_out[walker] = sweep | (mm << 8);
}

}  // namespace pepsgpu
