// f64-accumulated Gram matrix of the columns of P for the forward pass of the absorption (gfx950):
//
//     G[b] = P[b]^T P[b],   P = K x n (K live rows per walker, n <= 256 columns, row stride ld), G = n x n float64
//
// It replaces the R factor of qlten::QR at bmps_impl.h:821 together with the Cholesky kernels of linalg.h (only
// R^T R = P^T P is ever needed downstream).  The products of two f32 values are exact in f64, so the Gram carries the
// f32 data without squaring their rounding; v_mfma_f64_16x16x4_f64 does the accumulation.
//
// Shape of the work: per walker a few hundred rows and 64..256 columns -- too small for an LDS-tiled GEMM to amortise
// its barriers (tgemm_kernel reaches 11 TF here), so there is no LDS and no barrier at all: one WAVE owns one 64 x 64
// block of G (16 accumulator tiles of 16 x 16 in registers) and streams the rows of P straight from global memory / L2
// into the MFMA operand layout (lane l holds P[4 ks + l / 16][c0 + l % 16]: four 64-byte row segments per load, the A
// and the B operand of a tile are the same kind of load).  8 loads feed 16 MFMAs per step of four rows; the next step's
// loads are in flight during the MFMAs.  Only blocks on or above the diagonal are computed (the Cholesky reads the upper
// triangle and the diagonal blocks); a workgroup is four independent waves = four blocks of one walker.
#pragma once
#include "common.h"

namespace pepsgpu {

typedef double gr_f64x4 __attribute__((ext_vector_type(4)));

template <typename T>
__global__ __launch_bounds__(256, 2) void gram_cols_f64_kernel(const T *__restrict__ Pg, long wP, int n, int ld,
                                                               const int *__restrict__ kdyn, int kdyn_mul, int kmax,
                                                               double *__restrict__ Gg, long wG,
                                                               const int *__restrict__ run_flag, int inner,
                                                               const int *__restrict__ inner_live,
                                                               unsigned long long *__restrict__ flopc,
                                                               unsigned long long *__restrict__ bytec, int flop_stride, int nbatch) {
  // one WAVE = one 64 x 64 block of one walker; the waves of the launch are numbered across walkers (round 3: with ten blocks
  // per walker and four waves per workgroup a per-walker grid left two of twelve wave slots idle)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int nb = (n + 63) >> 6, nblk = nb * (nb + 1) / 2;
  const long gw = (long)blockIdx.x * 4 + wave;
  const int b = (int)(gw / nblk);
  if (b >= nbatch) return;
  if (run_flag && run_flag[b] >= 0) return;               // only the entries a cheaper kernel has declined
  const int t = (int)(gw - (long)b * nblk);
  int bi = 0, rem = t;
  while (rem >= nb - bi) { rem -= nb - bi; ++bi; }
  const int bj = bi + rem;                                // bi <= bj: on or above the diagonal
  const int K = kdyn ? max(0, min(kmax, kdyn[b] * kdyn_mul)) : kmax;
  if (run_flag) flop_stride = 1;      // partial participation: every running entry counts itself (no sampling)
  if (flopc && t == 0 && lane == 0 && b % flop_stride == 0) {
    atomicAdd(flopc, (unsigned long long)flop_stride * n * n * K);                 // = 2 n n K / 2 (upper triangle)
    if (bytec) atomicAdd(bytec, (unsigned long long)flop_stride * ((unsigned long long)K * n * sizeof(T) + (unsigned long long)n * n * 4));
  }
  const T *P = Pg + (long)b * wP;
  double *G = Gg + (long)b * wG;
  const int r4 = lane >> 4, c16 = lane & 15;
  const bool diag = bi == bj;
  // columns are (outer, inner) with `inner` fastest; inner_live[b] (optional) = live extent of the inner index (live bond of
  // the boundary MPS): the columns beyond it were never written and are read as zeros
  const int ilive = inner_live ? min(inner, inner_live[b]) : inner;
  int cola[4], colb[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    cola[c] = bi * 64 + 16 * c + c16;
    colb[c] = bj * 64 + 16 * c + c16;
    if (cola[c] % inner >= ilive) cola[c] = n;      // dead column: masked like a column beyond n
    if (colb[c] % inner >= ilive) colb[c] = n;
  }
  gr_f64x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[a][c][r] = 0.0;

  const int nks = (K + 3) >> 2;
  // Round 6: the k loop as straight-line code (one instantiation per kind of block: diagonal / off-diagonal), every load unconditional
  // at a clamped address (a row beyond K reads the last one, a dead column reads column 0; the operand is zeroed by a select), the
  // rotation of PF steps kept by scheduling fences.  Rounds 3-5 had `(ok && col < n) ? pr[col] : 0` and `if (cur < nks) step(...)`:
  // exec-mask branches with an s_waitcnt vmcnt(0) at every join -- the "PF steps deep" pipeline had one step in flight.  (Round 3
  // measured unconditional loads + select at 334 -> 553 ms: that form kept the raw and the selected values alive; here the select
  // happens in place, right before the MFMAs.)
  bool oka[4], okb[4];
  unsigned offa[4], offb[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    oka[c] = cola[c] < n; okb[c] = colb[c] < n;
    offa[c] = oka[c] ? (unsigned)cola[c] : 0u; offb[c] = okb[c] ? (unsigned)colb[c] : 0u;
  }
  auto run = [&](auto DIAG_) {
    constexpr bool DIAG = decltype(DIAG_)::value;
    constexpr int PF = 4;
    T av[PF][4], bv[PF][DIAG ? 1 : 4];
    auto load = [&](int ks, T(&a)[4], T(&bb)[DIAG ? 1 : 4]) {
      const unsigned row = (unsigned)min(4 * ks + r4, K - 1) * (unsigned)ld;
#pragma unroll
      for (int c = 0; c < 4; ++c) a[c] = P[row + offa[c]];
      if constexpr (!DIAG) {
#pragma unroll
        for (int c = 0; c < 4; ++c) bb[c] = P[row + offb[c]];
      }
    };
#pragma unroll
    for (int p = 0; p < PF - 1; ++p) load(p, av[p], bv[p]);
    for (int ks = 0; ks < nks; ks += PF) {
#pragma unroll
      for (int p = 0; p < PF; ++p) {
        const int cur = ks + p;
        load(cur + PF - 1, av[(p + PF - 1) % PF], bv[(p + PF - 1) % PF]);
        __builtin_amdgcn_sched_barrier(0);
        const bool rok = 4 * cur + r4 < K;      // (k-steps beyond nks: every row is beyond K, the MFMAs add zeros)
        double ad[4], bd[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) ad[c] = (rok && oka[c]) ? (double)av[p][c] : 0.0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          if constexpr (DIAG) bd[c] = ad[c];
          else bd[c] = (rok && okb[c]) ? (double)bv[p][c] : 0.0;
        }
        // a diagonal block needs its tiles on or above the diagonal only (ten of sixteen; the Cholesky reads the upper triangle)
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int c = 0; c < 4; ++c)
            if (!DIAG || c >= a) acc[a][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(ad[a], bd[c], acc[a][c], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  if (K > 0) {
    if (diag) run(std::true_type()); else run(std::false_type());
  }
  // accumulator layout of v_mfma_f64_16x16x4_f64: acc[r] = C[(lane >> 4) + 4 r][lane & 15]
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = bi * 64 + 16 * a + r4 + 4 * r, j = bj * 64 + 16 * c + c16;
        if (i < n && j < n && (!diag || c >= a)) G[(long)i * n + j] = acc[a][c][r];
      }
}

// ---------------------------------------------------------------------------------------------
// The same Gram for DENSE walkers with 129..256 columns, one WORKGROUP per walker (round 3).  The streaming kernel above gives
// every 64 x 64 block of G its own wave, so a walker's rows of P are fetched by four different waves: 10 GB of fabric traffic per
// launch of 2048 dense walkers against 3.1 GB of P (profiles/r03_pmc_FETCH_SIZE_c4_f32_real_nw2048.txt), 2.2 TB/s beside 50 % of
// the f64 MFMA peak.  Here the rows of P pass through LDS once, in chunks of GL_KC rows (float4 loads, dead columns and rows
// beyond K zeroed on the way in), and the eight waves of the workgroup hold all 136 tiles of the upper triangle in their
// accumulators, seventeen tiles per wave.  Operands are read from LDS as ds_read_b32 at a row pitch of 272 floats (the four k-lanes of an operand land on
// different banks).  Two chunk buffers: the next chunk is in flight (registers) during the MFMAs of the current one.
constexpr int GL_KC = 32, GL_PITCH = 272;
// tile t (0..16) of wave W: the (17 W + t)-th tile of the upper triangle of the 16 x 16 tile grid, rows first
template <int W> struct GlTile {
  static constexpr int start(int x) { return 16 * x - x * (x - 1) / 2; }
  static constexpr int x(int t) { int r = 0; while (r < 15 && start(r + 1) <= 17 * W + t) ++r; return r; }
  static constexpr int c(int t) { return x(t) + (17 * W + t - start(x(t))); }
};
inline size_t gram_cols_lds_smem_bytes() { return sizeof(float) * 2 * GL_KC * GL_PITCH; }

template <typename T>
__global__ __launch_bounds__(512, 2) void gram_cols_lds_kernel(const T *__restrict__ Pg, long wP, int n, int ld,
                                                               const int *__restrict__ kdyn, int kdyn_mul, int kmax,
                                                               double *__restrict__ Gg, long wG,
                                                               const int *__restrict__ run_flag, int inner,
                                                               const int *__restrict__ inner_live,
                                                               unsigned long long *__restrict__ flopc,
                                                               unsigned long long *__restrict__ bytec, int flop_stride) {
  static_assert(sizeof(T) == 4, "f32 input");
  extern __shared__ float gl_smem[];
  const int b = blockIdx.x;
  if (run_flag && run_flag[b] >= 0) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = kdyn ? max(0, min(kmax, kdyn[b] * kdyn_mul)) : kmax;
  if (run_flag) flop_stride = 1;
  if (flopc && tid == 0 && b % flop_stride == 0) {
    atomicAdd(flopc, (unsigned long long)flop_stride * n * n * K);
    if (bytec) atomicAdd(bytec, (unsigned long long)flop_stride * ((unsigned long long)K * n * sizeof(T) + (unsigned long long)n * n * 4));
  }
  const T *P = Pg + (long)b * wP;
  double *G = Gg + (long)b * wG;
  const int ilive = inner_live ? min(inner, inner_live[b]) : inner;
  const int i16 = lane & 15, k4 = lane >> 4;
  // ---- chunk loader: 32 rows x 256 columns = 2048 float4, four per thread ----
  const int nch = (K + GL_KC - 1) / GL_KC;
  float4 pv[4];
  auto issue = [&](int ch) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = tid + 512 * q;              // float4 index: row = e / 64, column group = e % 64
      const int r = ch * GL_KC + (e >> 6), c = 4 * (e & 63);
      const bool ok = r < K && c < n;
      const float4 v = ok ? *reinterpret_cast<const float4 *>(P + (long)r * ld + c) : make_float4(0.f, 0.f, 0.f, 0.f);
      pv[q] = v;
    }
  };
  auto lay = [&](int buf) {
    float *dst = gl_smem + buf * GL_KC * GL_PITCH;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = tid + 512 * q;
      const int rr = e >> 6, c = 4 * (e & 63);
      float v[4] = {pv[q].x, pv[q].y, pv[q].z, pv[q].w};
#pragma unroll
      for (int z = 0; z < 4; ++z)
        if ((c + z) % inner >= ilive || c + z >= n) v[z] = 0.f;      // dead / absent column: never written in P
      *reinterpret_cast<float4 *>(dst + rr * GL_PITCH + c) = make_float4(v[0], v[1], v[2], v[3]);
    }
  };
  // The 136 tiles (x, c >= x) of the 16 x 16 tile grid, row after row, dealt in runs of 17 to the eight waves: every wave
  // issues the same number of MFMAs.  A wave's run spans two or three tile rows; per k-step it converts the <= 16 operand
  // segments its tiles touch (the reads of the segments it does not use are dead code after unrolling) and issues 17 MFMAs.
  auto run = [&](auto wc) {
    constexpr int W = decltype(wc)::value;
    gr_f64x4 acc[17];
#pragma unroll
    for (int t = 0; t < 17; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[t][r] = 0.0;
    if (nch > 0) { issue(0); lay(0); }
    __syncthreads();
    for (int ch = 0; ch < nch; ++ch) {
      if (ch + 1 < nch) issue(ch + 1);
      const float *src = gl_smem + (ch & 1) * GL_KC * GL_PITCH;
#pragma unroll 2
      for (int s2 = 0; s2 < GL_KC / 4; ++s2) {
        const float *row = src + (4 * s2 + k4) * GL_PITCH + i16;
        double seg[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) seg[q] = (double)row[16 * q];
#pragma unroll
        for (int t = 0; t < 17; ++t)
          acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(seg[GlTile<W>::x(t)], seg[GlTile<W>::c(t)], acc[t], 0, 0, 0);
      }
      if (ch + 1 < nch) lay((ch + 1) & 1);
      __syncthreads();
    }
    // store: acc[r] = C[(lane >> 4) + 4 r][lane & 15]
#pragma unroll
    for (int t = 0; t < 17; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * GlTile<W>::x(t) + k4 + 4 * r, j = 16 * GlTile<W>::c(t) + i16;
        if (i < n && j < n) G[(long)i * n + j] = acc[t][r];
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

}  // namespace pepsgpu
#include "gram_i8.h"
namespace pepsgpu {

template <typename T>
inline void launch_gram_cols_f64(hipStream_t s, int nbatch, const T *P, long wP, int n, int ld, const int *kdyn, int kdyn_mul,
                                 int kmax, double *G, const int *run_flag, int inner, const int *inner_live,
                                 unsigned long long *flopc, unsigned long long *bytec) {
  if (nbatch <= 0 || n <= 0) return;
  PG_REQUIRE(nbatch <= 65535, 1, "walker batch exceeds 65535 (grid y limit)");
  if constexpr (sizeof(T) == 4) {
    constexpr bool no_lds_gram = false;
    // dense walkers (hint of the caller: kmax rows, 193..256 columns = four 64-column blocks): P through LDS once per walker
    if (!no_lds_gram && n > 192 && n <= 256 && ld % 4 == 0 && wP % 4 == 0 && (((uintptr_t)P) & 15) == 0 && kmax >= 256) {
      // round 4: the same Gram as exact integer arithmetic on the i8 matrix cores (gram_i8.h), PEPSGPU_NO_I8_GRAM=1 for the f64 form
      static const bool no_i8_gram = getenv("PEPSGPU_NO_I8_GRAM") != nullptr;
      if (!no_i8_gram) {
        const size_t smem8 = gram_cols_i8_smem_bytes();
        // twelve waves (three per SIMD, 12 / 11 tiles each): 1.61 ms against 1.85 ms with eight on 2048 walkers x 1536 rows
        allow_dynamic_lds(reinterpret_cast<const void *>(&gram_cols_i8_kernel<T, false, 0, 12>), smem8);
        hipLaunchKernelGGL((gram_cols_i8_kernel<T, false, 0, 12>), dim3(nbatch), dim3(768), smem8, s, P, wP, n, ld, kdyn, kdyn_mul, kmax, G, (long)n * n, n,
                           run_flag, inner > 0 ? inner : 1, inner_live, (const int *)nullptr, flopc, bytec, nbatch >= 256 ? 64 : 1);
        PG_CHECK_HIP(hipGetLastError());
        return;
      }
      const size_t smem = gram_cols_lds_smem_bytes();
      allow_dynamic_lds(reinterpret_cast<const void *>(&gram_cols_lds_kernel<T>), smem);
      hipLaunchKernelGGL(gram_cols_lds_kernel<T>, dim3(nbatch), dim3(512), smem, s, P, wP, n, ld, kdyn, kdyn_mul, kmax, G, (long)n * n,
                         run_flag, inner > 0 ? inner : 1, inner_live, flopc, bytec, nbatch >= 256 ? 64 : 1);
      PG_CHECK_HIP(hipGetLastError());
      return;
    }
  }
  const int nb = (n + 63) / 64, nblk = nb * (nb + 1) / 2;
  const long nwaves = (long)nblk * nbatch;
  hipLaunchKernelGGL(gram_cols_f64_kernel<T>, dim3((unsigned)((nwaves + 3) / 4)), dim3(256), 0, s, P, wP, n, ld, kdyn, kdyn_mul, kmax, G,
                     (long)n * n, run_flag, inner > 0 ? inner : 1, inner_live, flopc, bytec, nbatch >= 256 ? 64 : 1, nbatch);
  PG_CHECK_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
// Row Gram of the truncation input (round 3: the route for walkers with more than 128 live carry rows):
//
//     G[b] = M[b] M[b]^T,   M = n x K row-major (n = nrows[b] <= 256 live rows, K = row length, multiple of 16, row stride K),
//     G = ldg x ldg float64, upper blocks only
//
// Same shape of work as above (one wave = one 64 x 64 block of G in 16 accumulator tiles, no LDS, no barrier), but the
// contracted index runs ALONG the rows of M: lane (c16, r4) loads M[i0 + c16][16 s + 4 r4 .. + 3] as one 16-byte vector and
// feeds element t of it to MFMA t of the step -- the k index a lane supplies to an MFMA is any bijection of the 16 k of the
// step as long as the A and the B operand use the same one.  Per 16 k: 8 vector loads per lane, 64 MFMAs.
template <typename T>
__global__ __launch_bounds__(256, 2) void gram_rows_f64_kernel(const T *__restrict__ Mg, long wM, int K, const int *__restrict__ nrows,
                                                               double *__restrict__ Gg, long wG, int ldg,
                                                               const int *__restrict__ run_flag,
                                                               unsigned long long *__restrict__ flopc,
                                                               unsigned long long *__restrict__ bytec) {
  static_assert(sizeof(T) == 4, "f32 input");
  const int b = blockIdx.y;
  if (run_flag && run_flag[b] >= 0) return;
  const int n = nrows[b];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int nb = (n + 63) >> 6, nblk = nb * (nb + 1) / 2;
  const int t = blockIdx.x * 4 + wave;
  if (t >= nblk) return;
  int bi = 0, rem = t;
  while (rem >= nb - bi) { rem -= nb - bi; ++bi; }
  const int bj = bi + rem;
  if (flopc && t == 0 && lane == 0) {
    atomicAdd(flopc, (unsigned long long)n * n * K);
    if (bytec) atomicAdd(bytec, (unsigned long long)n * K * 4ull + (unsigned long long)n * n * 4ull);
  }
  const T *M = Mg + (long)b * wM;
  double *G = Gg + (long)b * wG;
  const int r4 = lane >> 4, c16 = lane & 15;
  const bool diag = bi == bj;
  typedef float gr_f32x4 __attribute__((ext_vector_type(4)));
  const gr_f32x4 *pa[4], *pb[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {       // rows beyond n: clamped address, the products land in rows / columns that are not stored
    pa[c] = reinterpret_cast<const gr_f32x4 *>(M + (long)min(bi * 64 + 16 * c + c16, n - 1) * K + 4 * r4);
    pb[c] = reinterpret_cast<const gr_f32x4 *>(M + (long)min(bj * 64 + 16 * c + c16, n - 1) * K + 4 * r4);
  }
  gr_f64x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[a][c][r] = 0.0;
  const int ns = K >> 4;
  gr_f32x4 av[2][4], bv[2][4];
  auto load = [&](int s, gr_f32x4(&x)[4], gr_f32x4(&y)[4]) {
#pragma unroll
    for (int c = 0; c < 4; ++c) x[c] = pa[c][4 * s];
    if (!diag) {
#pragma unroll
      for (int c = 0; c < 4; ++c) y[c] = pb[c][4 * s];
    }
  };
  auto step = [&](const gr_f32x4(&x)[4], const gr_f32x4(&y)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      double ad[4], bd[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) ad[c] = (double)x[c][q];
#pragma unroll
      for (int c = 0; c < 4; ++c) bd[c] = diag ? ad[c] : (double)y[c][q];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[a][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(ad[a], bd[c], acc[a][c], 0, 0, 0);
    }
  };
  if (ns > 0) load(0, av[0], bv[0]);
  for (int s = 0; s < ns; s += 2) {
    if (s + 1 < ns) load(s + 1, av[1], bv[1]);
    step(av[0], bv[0]);
    if (s + 1 < ns) {
      if (s + 2 < ns) load(s + 2, av[0], bv[0]);
      step(av[1], bv[1]);
    }
  }
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = bi * 64 + 16 * a + r4 + 4 * r, j = bj * 64 + 16 * c + c16;
        if (i < n && j < n) G[(long)i * ldg + j] = acc[a][c][r];
      }
}

inline bool gram_rows_i8_ok(const void *M, int nmax) {
  // round 4: exact integer arithmetic on the i8 matrix cores (gram_i8.h, ROWS form), PEPSGPU_NO_I8_GRAM=1 for the f64 form
  static const bool no_i8_gram = getenv("PEPSGPU_NO_I8_GRAM") != nullptr || false;
  return !no_i8_gram && nmax > 128 && nmax <= 256 && (((uintptr_t)M) & 15) == 0;
}

// sym (i8 form only, gram_rows_i8_ok): both triangles of G are written
template <typename T>
inline void launch_gram_rows_f64(hipStream_t s, int nbatch, const T *M, long wM, int K, int nmax, const int *nrows, double *G, long wG,
                                 int ldg, const int *run_flag, unsigned long long *flopc, unsigned long long *bytec, int sym = 0) {
  if (nbatch <= 0 || nmax <= 0) return;
  PG_REQUIRE(nbatch <= 65535, 1, "walker batch exceeds 65535 (grid y limit)");
  PG_REQUIRE(K % 16 == 0 && wM % 4 == 0, 1, "row Gram: row length must be a multiple of 16");
  if constexpr (sizeof(T) == 4) {
    if (gram_rows_i8_ok(M, nmax)) {
      const size_t smem8 = gram_cols_i8_smem_bytes();
      allow_dynamic_lds(reinterpret_cast<const void *>(&gram_cols_i8_kernel<T, true>), smem8);
      hipLaunchKernelGGL((gram_cols_i8_kernel<T, true>), dim3(nbatch), dim3(512), smem8, s, M, wM, nmax, K, (const int *)nullptr, 1, K, G, wG, ldg,
                         run_flag, 1, (const int *)nullptr, nrows, flopc, bytec, 1, sym);
      PG_CHECK_HIP(hipGetLastError());
      return;
    }
  }
  PG_REQUIRE(!sym, 1, "row Gram: the symmetric form exists on the i8 kernel only");
  const int nb = (nmax + 63) / 64, nblk = nb * (nb + 1) / 2;
  hipLaunchKernelGGL(gram_rows_f64_kernel<T>, dim3((nblk + 3) / 4, nbatch), dim3(256), 0, s, M, wM, K, nrows, G, wG, ldg, run_flag, flopc,
                     bytec);
  PG_CHECK_HIP(hipGetLastError());
}

}  // namespace pepsgpu
