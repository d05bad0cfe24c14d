// Row absorption and small helpers of Engine<T> (see engine.h for the algorithm statement).
#pragma once
#include "engine.h"
#include "jacobi_reg.h"

namespace pepsgpu {

__global__ void add_logs_kernel(double *acc, const double *a, const double *b, const double *c, const double *d, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) acc[i] += (a ? a[i] : 0.0) + (b ? b[i] : 0.0) + (c ? c[i] : 0.0) + (d ? d[i] : 0.0);
}

template <typename T>
void Engine<T>::add_logs(double *acc, const double *a, const double *b, const double *c, const double *d) {
  hipLaunchKernelGGL(add_logs_kernel, dim3((nw_ + 255) / 256), dim3(256), 0, stream_, acc, a, b, c, d, nw_);
}
template <typename T>
void Engine<T>::add_log(double *acc, const double *a) {
  add_logs(acc, a, nullptr, nullptr, nullptr);
}

// Jacobi dispatch: LDS-resident generic kernel when the block fits, register-resident kernel for
// the f32 bulk blocks (<= 256 x 256), global-memory generic kernel otherwise (f64 bulk blocks).
// mid_hi > 0: walkers with 32 < rows <= mid_hi are on the preconditioned mid route (absorb_impl) and are skipped here.
template <typename T>
bool Engine<T>::jacobi_small_ok(int len, int m, const int *mdyn) {
  if constexpr (sizeof(T) == 4) return len <= 256 && (mdyn || m <= JR_SMALL_ROWS);
  return false;
}

template <typename T>
bool Engine<T>::launch_jacobi(T *M, long wM, int m, int len, int use_lds, size_t need, const int *mdyn, int mdyn_mul, int mid_hi,
                              const JrSelect *sel, int rows_cap) {
  int small = 0;
  bool sel_used = false;
  if constexpr (sizeof(T) == 4) {
    // walkers whose block has at most 32 existing rows: one wave each; the kernels below return at once for those walkers
    if (jacobi_small_ok(len, m, mdyn)) {
      small = 1;
      // (small batches: giving every walker with <= 32 rows a wave of its own on the sixteen-lanes-per-row tournament -- four pairs
      // per instruction instead of the pairs of a walker one after the other -- was measured as a latency measure: one walker
      // 18.0 -> 19.6 ms per amplitude, 2048 walkers 34.4 -> 40.4 ms: the exchange rounds and the ranking prologue of that kernel
      // cost more than the shorter pair chain saves)
      // walkers with <= 16 rows first (low register count: all of them resident at once)
      if (len <= 64) {          // short rows (shrunk bonds): four walkers per wave, 16 lanes x 4 columns
        hipLaunchKernelGGL((jacobi_rows_tiny4_kernel<4>), dim3((nw_ + 15) / 16), dim3(256), 0, stream_, (float *)M, wM, m, len, len, 40,
                           sweeps_, mdyn, mdyn_mul, nw_, sel ? *sel : JrSelect());
        sel_used = sel != nullptr;
      } else if (len <= 128) {  // ... 16 lanes x 8 columns
        hipLaunchKernelGGL((jacobi_rows_tiny4_kernel<8>), dim3((nw_ + 15) / 16), dim3(256), 0, stream_, (float *)M, wM, m, len, len, 40,
                           sweeps_, mdyn, mdyn_mul, nw_, sel ? *sel : JrSelect());
        sel_used = sel != nullptr;
      } else
        hipLaunchKernelGGL(jacobi_rows_tiny_kernel, dim3((nw_ + 3) / 4), dim3(256), 0, stream_, (float *)M, wM, m, len, len, 40,
                           sweeps_, mdyn, mdyn_mul, nw_);
      PG_CHECK_HIP(hipGetLastError());
      if (m <= JR_BR || (rows_cap > 0 && rows_cap <= JR_BR)) return sel_used;
      // walkers with 17..32 rows: the sixteen-lanes-per-row tournament with one wave per walker (four players of two blocks of
      // four rows, four pairs per wave instruction) or, for short rows, the one-pair-per-instruction kernel; measured, jacobi
      // category per step of 4096 walkers: full-rank state 58.7 -> 54.6 ms, real state 515 -> 500 ms
      if (len > 128)
        launch_jacobi_grp<1, 16>(stream_, nw_, (float *)M, wM, m, len, len, 40, sweeps_, mdyn, mdyn_mul, JR_BR, JR_SMALL_ROWS);
      else if (len > 64)
        launch_jacobi_grp<1, 8>(stream_, nw_, (float *)M, wM, m, len, len, 40, sweeps_, mdyn, mdyn_mul, JR_BR, JR_SMALL_ROWS);
      else
        hipLaunchKernelGGL(jacobi_rows_small_kernel, dim3((nw_ + 3) / 4), dim3(256), 0, stream_, (float *)M, wM, m, len, len, 40,
                           sweeps_, mdyn, mdyn_mul, nw_, 1);
      PG_CHECK_HIP(hipGetLastError());
      if (m <= JR_SMALL_ROWS || (rows_cap > 0 && rows_cap <= JR_SMALL_ROWS)) return sel_used;
    }
    if (mid_hi && m <= mid_hi) return sel_used;        // every remaining walker is on the mid route
    const int skip = mid_hi ? mid_hi : small;          // rows <= max(skip, 32) are taken elsewhere
    if (!use_lds && m <= 256 && len <= 256) {
      hipLaunchKernelGGL(jacobi_rows_reg256_kernel, dim3(nw_), dim3(512), 0, stream_, (float *)M, wM, m, len, len, 40,
                         sweeps_, mdyn, mdyn_mul, skip);
      PG_CHECK_HIP(hipGetLastError());
      return sel_used;
    }
    hipLaunchKernelGGL(jacobi_rows_kernel<T>, dim3(nw_), dim3(1024), use_lds ? need : 0, stream_, M, wM, m, len, len, 40,
                       use_lds, sweeps_, mdyn, mdyn_mul, skip);
    PG_CHECK_HIP(hipGetLastError());
    return sel_used;
  }
  int skip_le = 0;
  if constexpr (std::is_same<T, double>::value) {
    // f64: walkers with at most JR_BR live rows of at most 128 elements in the register kernel (16 or 32 lanes per row)
    if (len <= 128 && (mdyn || m <= JR_BR)) {
      if (len <= 64)
        hipLaunchKernelGGL((jacobi_rows_tiny_f64_kernel<4, 16>), dim3((nw_ + 15) / 16), dim3(256), 0, stream_, (double *)M, wM, m, len, len,
                           40, sweeps_, mdyn, mdyn_mul, nw_);
      else
        hipLaunchKernelGGL((jacobi_rows_tiny_f64_kernel<4, 32>), dim3((nw_ + 7) / 8), dim3(256), 0, stream_, (double *)M, wM, m, len, len,
                           40, sweeps_, mdyn, mdyn_mul, nw_);
      PG_CHECK_HIP(hipGetLastError());
      if (m <= JR_BR) return sel_used;
      skip_le = JR_BR;
    }
  }
  // (float64 tail of the function: a 64 x 256 block of doubles is 131.6 KB, 0.6 KB above JACOBI_LDS_MAX -- the second row of every
  // stack at C4 ran its 18-20 sweeps from global memory; one block per CU with the block in LDS is the better trade up to 136 KB)
  if (!use_lds && need <= 136 * 1024) {
    use_lds = 1;
    allow_dynamic_lds(reinterpret_cast<const void *>(&jacobi_rows_kernel<T>), need);
  }
  // the static block does not fit LDS: 64 KB of dynamic LDS (two blocks per CU as before) for the walkers whose live rows do
  if (!use_lds && mdyn) {
    // (136 KB at one block per CU for blocks like C5's 144 x 145 doubles was measured: C5 f64 2 273 -> 2 109 amp/s -- not adopted)
    constexpr int CAP = 64 * 1024;
    allow_dynamic_lds(reinterpret_cast<const void *>(&jacobi_rows_kernel<T>), (size_t)CAP);
    hipLaunchKernelGGL(jacobi_rows_kernel<T>, dim3(nw_), dim3(1024), CAP, stream_, M, wM, m, len, len, 40, 2, sweeps_, mdyn, mdyn_mul,
                       small, skip_le, CAP);
    PG_CHECK_HIP(hipGetLastError());
    return sel_used;
  }
  hipLaunchKernelGGL(jacobi_rows_kernel<T>, dim3(nw_), dim3(1024), use_lds ? need : 0, stream_, M, wM, m, len, len, 40,
                     use_lds, sweeps_, mdyn, mdyn_mul, small, skip_le);
  PG_CHECK_HIP(hipGetLastError());
  return sel_used;
}

// BMPS::MultiplyMPO with SVD compression (bmps_impl.h:404-437, :756-862, :225-263), Q-less form.
// `num` = index of the absorbed row (UP/DOWN) or column (LEFT/RIGHT); sites are visited in the
// storage order of the BMPS (reversed for UP/RIGHT, bmps_impl.h:694-699).
//
// Rank-adaptive carry: the Cholesky kernel drops the rows of R_{i+1} that are numerically zero
// (below the rounding of the T-typed data) and reports the live count per walker; every launch
// that runs over the carry index m (X = R.A, P = X.W, the Gram's K index, M = R.T, the Jacobi)
// takes that count as a per-walker dynamic extent.  Buffers keep their static (worst case) shape.
// The new bonds get the static size chi unless the absorbed BMPS shows that far fewer states are alive
// (bond b of the new BMPS sits above bond b of the absorbed one and grows by a few states per row): then the
// static size is the previous row's maximum live count plus a margin.  After the absorption one small read-back
// checks that no walker filled a shrunk bond; if one did, the absorption is repeated at full size (rare).
template <typename T>
void Engine<T>::absorb(int pos, int num) {
  if constexpr (kCplx) {
    // complex element type: the plain static-shape form of the same algorithm (engine_cplx.h); variational schemes: engine_var.h
    BMPSDev out = (scheme_ != 0 && mps_len(pos) > 2) ? absorb_variational(pos, num, bmps_[pos].back()) : absorb_simple(pos, num, bmps_[pos].back());
    bmps_[pos].push_back(std::move(out));
  } else {
    if (scheme_ != 0 && mps_len(pos) > 2) {   // bmps_impl.h:419-430: N == 2 always takes the SVD path
      BMPSDev out = absorb_variational(pos, num, bmps_[pos].back());
      bmps_[pos].push_back(std::move(out));
      return;
    }
    BMPSDev out = absorb_svd(pos, num, bmps_[pos].back());
    bmps_[pos].push_back(std::move(out));
  }
}

template <typename T>
typename Engine<T>::BMPSDev Engine<T>::absorb_svd(int pos, int num, const BMPSDev &in) {
  constexpr bool no_shrink = false;
  ArenaScope scope(arena_);   // a throw inside returns every temporary and the half-built BMPS to the arena
  BMPSDev out;
  // A row whose hint-sized attempt had to be redone (its bonds grow faster than the margin: the first rows of a dense state)
  // goes straight to the full size the next time the same row is absorbed (the walkers of the next step look like these);
  // forgotten with the state (state_upload).
  char &redo_seen = redo_seen_[pos][num];
  // A hinted attempt that fails leaves walkers half processed (a walker the skipped fallback would have taken carries no live
  // rows, so its Y vanishes and the norm kernels raise its sticky flag): the persistent walker flags are snapshot before such an
  // attempt and put back when it is redone -- only the attempt that produced the result may flag a walker.
  int *flag_keep = nullptr;
  if (!redo_seen) {
    flag_keep = (int *)arena_.alloc(sizeof(int) * (size_t)nw_);
    PG_CHECK_HIP(hipMemcpyAsync(flag_keep, flag_, sizeof(int) * (size_t)nw_, hipMemcpyDeviceToDevice, stream_));
  }
  // (a throw while a kernel of the truncation route runs on the side stream: that kernel still reads and writes buffers the
  // ArenaScope is about to hand back -- wait for it first)
  auto impl = [&](bool full) {
    try { return absorb_impl(pos, num, full, in, out); }
    catch (...) { (void)hipStreamSynchronize(side_stream_); throw; }
  };
  if (redo_seen || !impl(no_shrink)) {
    if (!redo_seen) {
      ++n_redo_; free_bmps(out); out = BMPSDev();
      PG_CHECK_HIP(hipMemcpyAsync(flag_, flag_keep, sizeof(int) * (size_t)nw_, hipMemcpyDeviceToDevice, stream_));
    }
    redo_seen = 1;
    PG_REQUIRE(impl(true), 5, "MultiplyMPO: internal error (full-size absorption reported clipping)");
  }
  if (flag_keep) arena_.free(flag_keep);
  return out;
}

// hint of the row absorbed before: the carry at site i ran at a hundred or more live rows (a dense walker batch)
template <typename BM>
static inline int hint_dense_carry(const BM &in, int i) {
  return in.depth >= 3 && (int)in.mlmax.size() > i && in.mlmax[i] > 96;
}

template <typename T>
bool Engine<T>::absorb_impl(int pos, int num, bool full_bonds, const BMPSDev &in, BMPSDev &out) {
  const int N = mps_len(pos);
  const std::vector<DTen<T>> cur = in.t;
  const double *cur_log = in.logscale;
  // live bond dimensions of the absorbing BMPS (per walker, device) and of the one being built:
  // every contraction below runs over the live part of a bond only; persistent tensors stay zero padded
  static const bool bond_adapt = true && getenv("PEPSGPU_NO_RANK_ADAPT") == nullptr;
  std::vector<int *> clive = in.live;
  clive.resize(N + 1, nullptr);
  if (!bond_adapt) std::fill(clive.begin(), clive.end(), nullptr);
  std::vector<int *> kn(N + 1, nullptr);
  std::vector<int> cur_kmax = in.kmax;
  cur_kmax.resize(N + 1, -1);
  std::vector<int> kstat(N + 1, 0), kfull(N + 1, 0);   // static size chosen / full static size of each new bond
  PG_REQUIRE((int)cur.size() == N, 3, "MultiplyMPO: MPS/MPO length mismatch");
  auto site_rc = [&](int i, int &r, int &c) {
    switch (pos) {
      case DOWN: r = num; c = i; break;
      case UP: r = num; c = N - 1 - i; break;
      case LEFT: r = i; c = num; break;
      default: r = N - 1 - i; c = num; break;
    }
  };
  const int ll = (pos + 3) % 4, lp = pos, lr = (pos + 1) % 4, lu = (pos + 2) % 4;
  static const bool adaptive = getenv("PEPSGPU_NO_RANK_ADAPT") == nullptr;
  constexpr int chain_chunks = 1;
  // error-budget experiments (scripts/error_budget.py): contractions of the f32 engine with float64 accumulation, by stage
  // (1: X / P, 2: Z1 / Tt, 4: M = R Tt, 8: Y = Tt V^T; the separate LDS-tiled launches on the f64 matrix cores)
  static const int acc64 = (sizeof(T) == 4 && getenv("PEPSGPU_ACC64")) ? atoi(getenv("PEPSGPU_ACC64")) : 0;

  // ---------------- forward: R_{i+1} from P_i = R_i (A_i x W_i) ----------------
  std::vector<int> assume_fused(N + 1, 0);   // per carry: the Gram + Cholesky fallback of the fused factor was not launched (hint)
  std::vector<DTen<T>> R(N);
  std::vector<int *> mdyn(N, nullptr);     // live rows of R[i] = mdyn[i][w] * mmul[i] (nullptr: all rows)
  std::vector<int> mmul(N, 1);
  // R[i] is a Cholesky factor with compacted rows (row j zero before column j): the contractions that read it skip its zero blocks
  // (round 6; PEPSGPU_TRI=0: off).  Set where R[i] comes out of the Gram / Cholesky branch below (every kernel of it keeps the column
  // order), not where a walker may keep its rows of P.
  static const bool use_tri = getenv("PEPSGPU_TRI") == nullptr || atoi(getenv("PEPSGPU_TRI")) != 0;
  std::vector<char> R_tri(N, 0);
  R[0] = ones3();
  for (int i = 0; i + 1 < N; ++i) {
    int r, c, dd[4], st[4];
    site_rc(i, r, c);
    site_dims(r, c, dd);
    site_strides(r, c, st);
    const DTen<T> &A = cur[i];
    const int m = R[i].d[0], l = R[i].d[1], a = R[i].d[2];
    const int p = A.d[1], a2 = A.d[2];
    const int l2 = dd[lr], u = dd[lu];
    PG_REQUIRE(l == dd[ll] && a == A.d[0] && p == dd[lp], 3, "MultiplyMPO: bond dimension mismatch");
    // X[m,l,p,a2] = sum_a R[m,l,a] A[a,p,a2]                      (bmps_impl.h:806)
    // P[m,u,l2,a2] = sum_{l,p} X[m,l,p,a2] W[l,p,l2,u]           (bmps_impl.h:807 + :815-817)
    DTen<T> X = alloc_ten(m * l, p, a2);
    DTen<T> P = alloc_ten(m, u, l2, a2);
    {
      TGemmDesc gx, gp;
      gx.I[1] = m; gx.I[2] = l; gx.sAi[1] = l * a; gx.sAi[2] = a; gx.sCi[1] = l * p * a2; gx.sCi[2] = p * a2;
      gx.K[2] = a; gx.sAk[2] = 1; gx.sBk[2] = p * a2;
      gx.J[1] = p; gx.J[2] = a2; gx.sBj[1] = a2; gx.sBj[2] = 1; gx.sCj[1] = a2; gx.sCj[2] = 1;
      gx.wA = R[i].n; gx.wB = A.n; gx.wC = X.n; gx.nbatch = nw_;
      gx.dI[1].p = mdyn[i]; gx.dI[1].mul = mmul[i];   // live carry rows
      gx.dK[2].p = clive[i];            // live part of the bond to the left of A
      gx.dJ[2].p = clive[i + 1];        // ... and to its right
      // site tensor as the A operand: the lanes of a tile run along (m, a2), contiguous in X and in P
      gp.I[1] = l2; gp.I[2] = u; gp.sAi[1] = st[lr]; gp.sAi[2] = st[lu]; gp.sCi[1] = a2; gp.sCi[2] = l2 * a2;
      gp.K[1] = l; gp.K[2] = p; gp.sAk[1] = st[ll]; gp.sAk[2] = st[lp]; gp.sBk[1] = p * a2; gp.sBk[2] = a2;
      gp.J[1] = m; gp.J[2] = a2; gp.sBj[1] = l * p * a2; gp.sBj[2] = 1; gp.sCj[1] = u * l2 * a2; gp.sCj[2] = 1;
      gp.wB = X.n; gp.wC = P.n; gp.nbatch = nw_;
      gp.dJ[1].p = mdyn[i]; gp.dJ[1].mul = mmul[i];
      gp.dJ[2].p = clive[i + 1];
      const double flx = 2.0 * nw_ * (double)(m * l) * a * (double)(p * a2);
      const double flp = 2.0 * nw_ * (double)(m * a2) * (double)(l * p) * (double)(l2 * u);
      int *chain_flag = nullptr;
      int chained = 0;
      if constexpr (sizeof(T) == 4) {
        constexpr bool no_chain = false;
        if (!no_chain && !(acc64 & 1)) {
          // both contractions in one launch, X stays in LDS; walkers whose live X does not fit are flagged and take the
          // two separate launches below
          chain_flag = (int *)arena_.alloc(sizeof(int) * nw_);
          TGemmDesc g2 = gp;
          const SiteSel ss = cfg_site(r, c);
          g2.selA = ss.sel; g2.selA_mul = slot_; g2.selA_inc = ss.inc; g2.seldivA = 1; g2.wA = 0;
          TGemmChainMap mp;
          mp.mapK[1] = 2; mp.mapK[2] = 4;      // K2 = (l, p): l = I1[2], p = J1[1]
          mp.mapJ[1] = 1; mp.mapJ[2] = 5;      // J2 = (m, a2): m = I1[1], a2 = J1[2]
          prof_begin(PROF_CHAIN, flx + flp, flx + flp);
          chained = tgemm_chain_launch(stream_, gx, g2, mp, (const float *)R[i].p, (const float *)A.p,
                                       (const float *)sel_base(ss), (float *)P.p, chain_flag, chain_chunks, hint_dense_carry(in, i), 0,
                                       R_tri[i] ? 1 : 0);
          prof_end();
          if (!chained) { arena_.free(chain_flag); chain_flag = nullptr; }
        }
      }
      if (chained < 2) {   // the two separate launches: for the entries the chain declined (all of them when it did not run)
        gx.batch_flag = chain_flag; gp.batch_flag = chain_flag;
        prof_begin(PROF_CONTRACT, chain_flag ? 0.0 : flx, chain_flag ? 0.0 : flx);
        if (acc64 & 1) tgemm_launch<T, T, T, Acc>(stream_, gx, R[i].p, A.p, X.p);
        else tgemm_launch<T, T, T, T>(stream_, gx, R[i].p, A.p, X.p);
        prof_end();
        prof_begin(PROF_CONTRACT, chain_flag ? 0.0 : flp, chain_flag ? 0.0 : flp);
        launch_site_gemm_a(gp, cfg_site(r, c), 1, X.p, P.p, (acc64 & 1) != 0);
        prof_end();
      }
      if (chain_flag) arena_.free(chain_flag);
    }
    free_ten(X);
    inject(INJ_P, P.p, P.n);
    const int rows = m * u, cols = l2 * a2;
    // hint from the row absorbed before: its carry at the next site ran above the small rank cap of the factor kernels
    const bool hint_dense = in.depth >= 3 && (int)in.mlmax.size() > i + 1 && in.mlmax[i + 1] > 14;
    constexpr int FUSED_KCAP = sizeof(T) == 4 ? 96 : 48;   // rows of P a thread of the fused kernel holds in registers
    constexpr bool no_fused = false;
    if (rows < cols && adaptive && !no_fused && cols <= 256 && rows >= 16 && rows <= FUSED_KCAP) {
      // Fewer rows than columns, but already more rows than the usual numerical rank: compress now
      // (gram_chol_lowrank_kernel) instead of letting the carry grow by the factor u per site until it
      // reaches the column count.  Walkers whose rank exceeds the kernel's cap keep their rows of P.
      R[i + 1] = alloc_ten(cols, l2, a2);
      int *ml = (int *)arena_.alloc(sizeof(int) * nw_);
      prof_begin(PROF_CHOL, nw_ * 2.0 * (2.0 * cols * (double)rows * rows - 2.0 / 3.0 * (double)rows * rows * rows), 0.0);
      int *flist = (int *)arena_.alloc(sizeof(int) * (nw_ + 1));
      launch_gram_chol_lowrank<T, FUSED_KCAP>(stream_, nw_, (const T *)P.p, P.n, cols, (const int *)mdyn[i], mmul[i] * u, rows,
                                              R[i + 1].p, R[i + 1].n, ml, a2, (const int *)clive[i + 1], 1, hint_dense, flist);
      arena_.free(flist);
      hipLaunchKernelGGL(adopt_rows_flagged_kernel<T>, dim3(nw_), dim3(256), 0, stream_, (const T *)P.p, P.n, cols,
                         (const int *)mdyn[i], mmul[i] * u, rows, R[i + 1].p, R[i + 1].n, ml, a2, (const int *)clive[i + 1]);
      PG_CHECK_HIP(hipGetLastError());
      prof_end();
      mdyn[i + 1] = ml;
      mmul[i + 1] = 1;
      free_ten(P);
    } else if (rows < cols) {
      // economy QR would return R = Q^T P with rows x cols; any R with R^T R = P^T P serves
      // (rows == cols goes through the Cholesky: a triangular carry makes the Jacobi converge 3x faster)
      P.d[0] = rows; P.d[1] = l2; P.d[2] = a2; P.d[3] = 1;
      // reference op here: QR of the (rows x cols) block, rows < cols (swap R,C)
      prof_begin(PROF_NORM, nw_ * 2.0 * (2.0 * cols * (double)rows * rows - 2.0 / 3.0 * (double)rows * rows * rows), 0.0);
      if (clive[i + 1]) {
        hipLaunchKernelGGL(zero_dead_cols_kernel<T>, dim3(nw_), dim3(256), 0, stream_, P.p, P.n, cols, (const int *)mdyn[i],
                           mmul[i] * u, rows, a2, (const int *)clive[i + 1], (const int *)nullptr);
        PG_CHECK_HIP(hipGetLastError());
      }
      normalize(P.p, P.n, P.n, nw_, nullptr, mdyn[i], mmul[i] * u * cols);
      prof_end();
      R[i + 1] = P;
      mdyn[i + 1] = mdyn[i];                 // live rows of P = live rows of R_i times u (m is P's outer index)
      mmul[i + 1] = mmul[i] * u;
    } else {
      double *G = nullptr;
      R[i + 1] = alloc_ten(cols, l2, a2);
      int *ml = adaptive ? (int *)arena_.alloc(sizeof(int) * nw_) : nullptr;
      // Low-rank walkers: the factor straight from the live rows of P, no Gram matrix in memory
      // (gram_chol_lowrank_kernel); it flags the walkers it cannot take (ml = -1) and the Gram GEMM
      // and the Cholesky kernels below then run for those only.
      const bool fused = ml && !no_fused && cols <= 256 && (mdyn[i] || rows <= FUSED_KCAP);
      // Hint of the row absorbed before: its carry stayed at <= 24 rows on both sides of this site, well inside what the fused
      // factor covers (rank 32, 288 rows) -- the launches for the walkers it would flag (Gram, low-rank and blocked Cholesky:
      // ~66 us per site on an empty list) are not issued.  Verified after the absorption: a walker left flagged (ml < 0) fails
      // the attempt and the absorption is redone with every launch (absorb_svd), as for the other hints.
      constexpr bool no_skip_fb = false;
      static const bool force_skip_fb = getenv("PEPSGPU_FORCE_SKIP_FALLBACK") != nullptr;     // tests: a wrong hint
      const bool skip_fb = fused && !full_bonds && !no_skip_fb && sizeof(T) == 4 &&
                           (force_skip_fb || (in.depth >= 3 && (int)in.mlmax.size() > i + 1 && in.mlmax[i] >= 0 && in.mlmax[i] <= 24 &&
                                              in.mlmax[i + 1] >= 0 && in.mlmax[i + 1] <= 24));
      if (fused) {
        // more live rows than one pass holds (moderate rank): fold the rows of P in over up to four passes
        // (covers K <= KCAP + 3 (KCAP - 32) rows); walkers beyond that, or of rank > 32, are flagged
        constexpr int max_pass = 4;
        const int npass = (mdyn[i] && rows > FUSED_KCAP) ? std::max(1, max_pass) : 1;
        prof_begin(PROF_CHOL, 0.0, 0.0);
        int *flist = (int *)arena_.alloc(sizeof(int) * (nw_ + 1));
        launch_gram_chol_lowrank<T, FUSED_KCAP>(stream_, nw_, (const T *)P.p, P.n, cols, (const int *)mdyn[i], mmul[i] * u, rows,
                                                R[i + 1].p, R[i + 1].n, ml, a2, (const int *)clive[i + 1], npass, hint_dense, flist);
        arena_.free(flist);
        prof_end();
      }
      if (skip_fb) assume_fused[i + 1] = 1;
      else {
      G = (double *)arena_.alloc(sizeof(double) * (size_t)cols * cols * nw_);
      constexpr bool no_gd = false;
      const bool gram_direct = !no_gd && cols >= 32 && cols <= 256;
      if (clive[i + 1] && !gram_direct) {   // the Gram GEMM reads whole rows: define the never-written columns (flagged walkers only)
        hipLaunchKernelGGL(zero_dead_cols_kernel<T>, dim3(nw_), dim3(256), 0, stream_, P.p, P.n, cols, (const int *)mdyn[i],
                           mmul[i] * u, rows, a2, (const int *)clive[i + 1], (const int *)(fused ? ml : nullptr));
        PG_CHECK_HIP(hipGetLastError());
      }
      {
        TGemmDesc g;
        g.I[2] = cols; g.sAi[2] = 1; g.sCi[2] = cols;
        g.K[2] = rows; g.sAk[2] = cols; g.sBk[2] = cols;
        g.J[2] = cols; g.sBj[2] = 1; g.sCj[2] = 1;
        g.wA = P.n; g.wB = P.n; g.wC = (long)cols * cols; g.nbatch = nw_;
        g.dynK = mdyn[i]; g.dynK_mul = mmul[i] * u;
        g.upper_only = 1;                       // the Cholesky reads the upper triangle only
        g.batch_flag = fused ? ml : nullptr;
        // algorithmic flops of the op this replaces: geqrf + orgqr of (rows x cols)
        // (executed flops of this category are counted on the device only: the launch runs for the flagged walkers)
        prof_begin(PROF_GRAM, nw_ * 2.0 * (2.0 * rows * (double)cols * cols - 2.0 / 3.0 * (double)cols * cols * cols), 0.0);
        if (gram_direct)   // wave-per-block streaming kernel (gram.h): no LDS, no barrier; dead columns masked at the load
          launch_gram_cols_f64<T>(stream_, nw_, (const T *)P.p, P.n, cols, cols, (const int *)mdyn[i], mmul[i] * u, rows, G,
                                  (const int *)(fused ? ml : nullptr), a2, (const int *)clive[i + 1], tg_flop_counter,
                                  tg_byte_counter);
        else
          tgemm_launch<T, T, double, double>(stream_, g, P.p, P.p, G);
        prof_end();
      }
      const size_t smem = chol_smem_bytes(cols);
      PG_REQUIRE(smem <= 150 * 1024 && cols < 32768, 1, "Cholesky panel does not fit LDS (D*chi too large)");
      allow_dynamic_lds(reinterpret_cast<const void *>(&chol_upper_kernel<T>), smem);
      prof_begin(PROF_CHOL, 0.0, 0.0);   // (executed flops of this category: the MFMA flops of the fused Gram kernels, counted on the device)
      constexpr bool no_lowrank = false;
      // (hint from the row absorbed before: when its carry at this site ran well above the cap, every walker would spend 32
      // steps here only to be handed on; the blocked kernel takes any rank)
      const bool above_cap = in.depth >= 3 && (int)in.mlmax.size() > i + 1 && in.mlmax[i + 1] > CH_LR_CAP + 8;
      const bool lowrank = ml && !no_lowrank && cols <= 256 * CH_LR_Q && !above_cap;
      if (lowrank) {   // walkers of rank <= CH_LR_CAP finish here; the others are flagged for the blocked kernel
        const size_t lsm = chol_lowrank_smem_bytes(cols);
        allow_dynamic_lds(reinterpret_cast<const void *>(&chol_lowrank_kernel<T>), lsm);
        hipLaunchKernelGGL(chol_lowrank_kernel<T>, dim3(nw_), dim3(256), lsm, stream_, (const double *)G, (long)cols * cols,
                           cols, R[i + 1].p, R[i + 1].n, ml, fused ? 1 : 0);
        PG_CHECK_HIP(hipGetLastError());
      }
      launch_chol_upper<T>(stream_, nw_, G, (long)cols * cols, cols, R[i + 1].p, R[i + 1].n, ml, (lowrank || fused) ? 1 : 0);
      prof_end();
      }
      if (dbg_sweeps_ && ml) {   // diagnostics: numerical rank of the carry (forces a sync)
        std::vector<int> h(nw_);
        PG_CHECK_HIP(hipMemcpyAsync(h.data(), ml, nw_ * sizeof(int), hipMemcpyDeviceToHost, stream_));
        PG_CHECK_HIP(hipStreamSynchronize(stream_));
        for (int v : h) { live_sum_ += v; live_full_ += cols; live_max_ = std::max<long>(live_max_, v); }
      }
      mdyn[i + 1] = ml;
      mmul[i + 1] = 1;
      R_tri[i + 1] = use_tri && ml != nullptr;
      arena_.free(G);
      free_ten(P);
    }
    inject(INJ_R, R[i + 1].p, R[i + 1].n);
  }

  // ---------------- backward: truncate right to left ----------------
  out.t.resize(N);
  out.logscale = (double *)arena_.alloc(sizeof(double) * nw_);
  PG_CHECK_HIP(hipMemcpyAsync(out.logscale, cur_log, sizeof(double) * nw_, hipMemcpyDeviceToDevice, stream_));
  DTen<T> Y = ones3();   // [l2, a2, k2]
  std::vector<int> assume_rows(N, 0);   // per site: the live-row cap the Jacobi launches relied on (0: none)
  float *yscale = nullptr;   // 1 / |Y| per walker when Y was left unnormalised by the launch that wrote it (y_scaled)
  bool y_scaled = false;
  for (int i = N - 1; i >= 0; --i) {
    int r, c, dd[4], st[4];
    site_rc(i, r, c);
    site_dims(r, c, dd);
    site_strides(r, c, st);
    const DTen<T> &A = cur[i];
    const int a = A.d[0], p = A.d[1], a2 = A.d[2];
    const int l = dd[ll], l2 = dd[lr], u = dd[lu];
    const int k2 = Y.d[2];
    PG_REQUIRE(Y.d[0] == l2 && Y.d[1] == a2 && p == dd[lp], 3, "MultiplyMPO: bond dimension mismatch (backward)");
    // "Precise" sites (f32 engine): where the carry is not of low rank -- the row absorbed before ran more than 24 live
    // carry rows at this site, or gives no hint yet (the first three rows of a stack) -- the places where f32 rounding showed in the
    // amplitude get float64-grade arithmetic: the backward pair Z1 = A Y, Tt = W Z1 (round 5) and Y = Tt V^T accumulate in float64 on
    // the f64 matrix cores (columns of small sigma are differences of O(sigma_1) terms), and the rows of Vt are made orthonormal by a
    // Newton-Schulz step in float64 (ortho_rows_kernel).  The low-rank headline state keeps the f32 forms.
    bool precise_site = false;
    if constexpr (sizeof(T) == 4) {
      static const int precise = getenv("PEPSGPU_PRECISE") ? atoi(getenv("PEPSGPU_PRECISE")) : 1;    // 0 never, 1 auto, 2 always
      // (rows whose predecessor gives no hint yet -- the first three of a stack -- go by what the SAME row of the SAME stack showed
      // the last time it was absorbed, carry_seen_: unknown on a fresh state -> precise)
      const int seen = carry_seen_[pos][num];
      const bool hinted = in.depth >= 3 && (int)in.mlmax.size() > i && in.mlmax[i] >= 0;
      precise_site = precise == 2 || (precise == 1 && (hinted ? in.mlmax[i] > 24 : (seen < 0 || seen > 24)));
    }
    // the backward pair of a precise site runs on the float64-accumulating chained kernel (round 5; f32 in round 4)
    constexpr int tt_mode = 1;
    const bool tt_f64 = sizeof(T) == 4 && precise_site && tt_mode != 0;
    // Z1[a,p,l2,k2] = sum_{a2} A[a,p,a2] Y[l2,a2,k2]
    // Tt[l,a,u,k2] = sum_{p,l2} Z1[a,p,l2,k2] W[l,p,l2,u]
    DTen<T> Z1 = alloc_ten(a, p, l2, k2);
    DTen<T> Tt = alloc_ten(l, a, u, k2);
    // Layout of Tt (internal to this site step: written once, read by M = R Tt and by Y = Tt V^T): with the partially live
    // bond k2 innermost the live part of a (l, a) slice is u runs of k2_live floats (40 bytes in 64-byte requests); with the
    // full leg u innermost it is ONE run of k2_live * u floats.  Not at i == 0, where Tt becomes the first tensor (u, k2).
    constexpr bool tt_swap = true;
    const bool tsw = tt_swap && i > 0 && sizeof(T) == 4;
    {
      TGemmDesc gz, gt;
      gz.I[1] = a; gz.I[2] = p; gz.sAi[1] = p * a2; gz.sAi[2] = a2; gz.sCi[1] = p * l2 * k2; gz.sCi[2] = l2 * k2;
      gz.K[2] = a2; gz.sAk[2] = 1; gz.sBk[2] = k2;
      gz.J[1] = l2; gz.J[2] = k2; gz.sBj[1] = a2 * k2; gz.sBj[2] = 1; gz.sCj[1] = k2; gz.sCj[2] = 1;
      gz.wA = A.n; gz.wB = Y.n; gz.wC = Z1.n; gz.nbatch = nw_;
      if constexpr (sizeof(T) == 4) { if (y_scaled) gz.scale_in = yscale; }
      gz.dI[1].p = clive[i];                  // live bonds: a (rows of A), a2 (contracted), k2 (new bond to the right)
      gz.dK[2].p = clive[i + 1];
      gz.dJ[2].p = kn[i + 1];
      // site tensor as the A operand: the lanes of a tile run along (a, k2), contiguous in Z1 and in Tt
      gt.I[1] = l; gt.I[2] = u; gt.sAi[1] = st[ll]; gt.sAi[2] = st[lu]; gt.sCi[1] = a * u * k2; gt.sCi[2] = k2;
      gt.K[1] = p; gt.K[2] = l2; gt.sAk[1] = st[lp]; gt.sAk[2] = st[lr]; gt.sBk[1] = l2 * k2; gt.sBk[2] = k2;
      gt.J[1] = a; gt.J[2] = k2; gt.sBj[1] = p * l2 * k2; gt.sBj[2] = 1; gt.sCj[1] = u * k2; gt.sCj[2] = 1;
      if (tsw) { gt.sCi[2] = 1; gt.sCj[2] = u; }    // Tt[l, a, k2, u]: the fully live leg u innermost (see tsw above)
      gt.wB = Z1.n; gt.wC = Tt.n; gt.nbatch = nw_;
      gt.dJ[1].p = clive[i];
      gt.dJ[2].p = kn[i + 1]; gt.dJ[2].mask = (i == 0);   // i == 0: Tt becomes the (persistent, zero padded) first tensor
      const double flz = 2.0 * nw_ * (double)(a * p) * a2 * (double)(l2 * k2);
      const double flt = 2.0 * nw_ * (double)(a * k2) * (double)(p * l2) * (double)(l * u);
      int *chain_flag = nullptr;
      int chained = 0;
      if constexpr (sizeof(T) == 4) {
        constexpr bool no_chain = false;
        if (!no_chain && !(acc64 & 2)) {   // Z1 stays in LDS (see the forward pair)
          chain_flag = (int *)arena_.alloc(sizeof(int) * nw_);
          TGemmDesc g2 = gt;
          const SiteSel ss = cfg_site(r, c);
          g2.selA = ss.sel; g2.selA_mul = slot_; g2.selA_inc = ss.inc; g2.seldivA = 1; g2.wA = 0;
          TGemmChainMap mp;
          mp.mapK[1] = 2; mp.mapK[2] = 4;      // K2 = (p, l2): p = I1[2], l2 = J1[1]
          mp.mapJ[1] = 1; mp.mapJ[2] = 5;      // J2 = (a, k2): a = I1[1], k2 = J1[2]
          prof_begin(PROF_CHAIN, 0.0, flz + flt);
          chained = tgemm_chain_launch(stream_, gz, g2, mp, (const float *)A.p, (const float *)Y.p,
                                       (const float *)sel_base(ss), (float *)Tt.p, chain_flag, chain_chunks, hint_dense_carry(in, i),
                                       tt_f64 ? 1 : 0);
          prof_end();
          if (!chained) { arena_.free(chain_flag); chain_flag = nullptr; }
        }
      }
      if (chained < 2) {
        gz.batch_flag = chain_flag; gt.batch_flag = chain_flag;
        prof_begin(PROF_CONTRACT, 0.0, chain_flag ? 0.0 : flz);
        gz.acc64 = tt_f64; gt.acc64 = tt_f64;     // (entries the chain declined: the wave-per-tile kernel honours it)
        if (acc64 & 2) tgemm_launch<T, T, T, Acc>(stream_, gz, A.p, Y.p, Z1.p);
        else tgemm_launch<T, T, T, T>(stream_, gz, A.p, Y.p, Z1.p);
        prof_end();
        prof_begin(PROF_CONTRACT, 0.0, chain_flag ? 0.0 : flt);
        launch_site_gemm_a(gt, cfg_site(r, c), 1, Z1.p, Tt.p, (acc64 & 2) != 0);
        prof_end();
      }
      if (chain_flag) arena_.free(chain_flag);
    }
    free_ten(Z1);
    free_ten(Y);
    inject(INJ_T, Tt.p, Tt.n);
    if (i == 0) {
      PG_REQUIRE(l == 1 && a == 1, 3, "MultiplyMPO: left boundary bond is not trivial");
      Tt.d[0] = 1; Tt.d[1] = u; Tt.d[2] = k2; Tt.d[3] = 1;
      prof_begin(PROF_NORM, 0.0, 0.0);
      normalize(Tt.p, Tt.n, Tt.n, nw_, out.logscale);
      prof_end();
      inject(INJ_V, Tt.p, Tt.n);
      out.t[0] = Tt;
      break;
    }
    // M[m,(u,k2)] = sum_{(l,a)} R_i[m,(l,a)] Tt[(l,a),(u,k2)]
    const int m = R[i].d[0], la = l * a, uk = u * k2;
    bool dense_site = false;
    PG_REQUIRE(R[i].d[1] == l && R[i].d[2] == a, 3, "MultiplyMPO: carry dimension mismatch");
    DTen<T> M = alloc_ten(m, uk, 1);
    {
      TGemmDesc g;
      g.I[2] = m; g.sAi[2] = la; g.sCi[2] = uk;
      g.K[1] = l; g.K[2] = a; g.sAk[1] = a; g.sAk[2] = 1; g.sBk[1] = a * uk; g.sBk[2] = uk;
      g.J[1] = u; g.J[2] = k2; g.sBj[1] = k2; g.sBj[2] = 1; g.sCj[1] = k2; g.sCj[2] = 1;
      if (tsw) { g.sBj[1] = 1; g.sBj[2] = u; }
      g.wA = R[i].n; g.wB = Tt.n; g.wC = M.n; g.nbatch = nw_;
      g.dynI = mdyn[i]; g.dynI_mul = mmul[i];
      g.dK[2].p = clive[i];
      g.dJ[2].p = kn[i + 1]; g.dJ[2].mask = 1;   // the Jacobi reads whole rows of M: dead columns are written as zeros
      // dense carry at this site (hint of the row absorbed before): the LDS-tiled kernel
      constexpr bool no_tiled_hint = false;
      dense_site = !no_tiled_hint && in.depth >= 3 && (int)in.mlmax.size() > i && in.mlmax[i] > 96 && la >= 128 && uk >= 128;
      g.prefer_tiled = dense_site;
      prof_begin(PROF_CONTRACT, 0.0, 2.0 * nw_ * (double)m * la * (double)uk);
      bool mg_done = false;
      if constexpr (sizeof(T) == 4) {
        // dense carry: the workgroup-per-walker kernel (mgemm_dense.h): R and Tt through LDS once, eight waves x 32 columns
        constexpr bool no_mgd = false;
        if (dense_site && !no_mgd && !(acc64 & 4) && m > 128 && mgemm_dense_ok(m, la, a, u, k2, R[i].n, Tt.n, R[i].p, Tt.p)) {
          launch_mgemm_dense(stream_, nw_, (const float *)R[i].p, R[i].n, (const float *)Tt.p, Tt.n, (float *)M.p, M.n, m, la, a, u, k2,
                             tsw ? 1 : 0, (const int *)mdyn[i], mmul[i], (const int *)clive[i], (const int *)kn[i + 1], tg_flop_counter,
                             tg_byte_counter, R_tri[i] ? 1 : 0);
          mg_done = true;
        }
      }
      if (!mg_done) {
        if (acc64 & 4) tgemm_launch<T, T, T, Acc>(stream_, g, R[i].p, Tt.p, M.p);
        else tgemm_launch<T, T, T, T>(stream_, g, R[i].p, Tt.p, M.p);
      }
      prof_end();
      inject(INJ_M, M.p, M.n);
    }
    // rows of M -> mutually orthogonal (sigma_k v_k^T)
    //
    // Mid route (f32, 32 < live rows <= 128: the usual size of the carry on states of higher rank): the Jacobi runs on
    // the triangular factor B of the small Gram matrix instead of on M itself,
    //     G = M M^T (f64 MFMA, ml x ml),  B^T B = G (Cholesky),  rows of B --Jacobi--> sigma_k u_k^T,
    //     Vt = rows of (U^T M) normalised,
    // the preconditioned one-sided Jacobi SVD (Drmac / Veselic): rows are ml <= 128 long instead of u * k2, the
    // triangular factor converges in about half the sweeps, and four walkers share a CU.  sigma and Vt are those of M:
    // select_rows_kernel sees the same singular values, the truncation rule is unchanged.
    // Round 3: the route reaches 256 live rows.  A state of the rank of a real PEPS carries ~190-240 live rows, but M = R Tt is
    // numerically of rank ~60-100 at the f32 floor (the singular values of the truncation input fall by five orders of magnitude
    // over the first 32): the Cholesky of M M^T drops the dependent rows, the Jacobi runs on the <= 128 live rows of B (256 long)
    // instead of on the 240 rows of M (19 sweeps of the 256 x 256 register kernel: 80 % of the step before).
    constexpr bool no_dense_mid = false;
    // hint from the row absorbed before: no walker came near 128 live rows at this site -> the route keeps its <= 128-row form
    // (walkers that do exceed 128 rows are then taken by the general kernels: time, never correctness)
    const bool hint_le128 = !full_bonds && in.depth >= 3 && (int)in.mlmax.size() > i && in.mlmax[i] >= 0 && in.mlmax[i] + 12 <= 128;
    const int MID_HI = (m > 128 && !no_dense_mid && !hint_le128) ? 256 : 128;
    bool mid = false;
    if constexpr (sizeof(T) == 4) {
      static const bool no_mid = getenv("PEPSGPU_NO_MIDROUTE") != nullptr;
      // hint from the row absorbed before (the carry rank grows by a few states per row): no walker near 32 live rows
      // at this site -> skip the route's launches; walkers that do exceed 32 rows are then taken by the general kernels
      const bool near = in.depth < 3 || (int)in.mlmax.size() <= i || in.mlmax[i] < 0 || in.mlmax[i] > 24;
      mid = !no_mid && adaptive && m > JR_SMALL_ROWS && uk <= 1024 && near;
    }
    int *midflag = nullptr, *nmid = nullptr, *mB = nullptr;
    int *flagA = nullptr, *rowsA = nullptr, *flag2 = nullptr, *rows2 = nullptr, *mB2 = nullptr;   // two-level form (below)
    int *big_list = nullptr;    // walkers whose first factor kept more than 128 rows (+ their count behind the list)
    bool side_pending = false;  // a kernel of this site runs on the side stream
    bool two_level = false;
    int *ortho_skip = nullptr;  // walkers whose rows of V are orthonormal to float64 accuracy already (rows_qr.h): flag < 0
    bool pivoted = false;       // the first factor of the two-level form came from chol_pivot_kernel: at most 64 rows per walker
    DTen<T> Bt, Ut, B2;
    const int GS = std::min(m, MID_HI);
    if (mid) {
      midflag = (int *)arena_.alloc(sizeof(int) * nw_);
      nmid = (int *)arena_.alloc(sizeof(int) * nw_);
      mB = (int *)arena_.alloc(sizeof(int) * nw_);
      PG_CHECK_HIP(hipMemsetAsync(mB, 0, sizeof(int) * nw_, stream_));
      const int lo = jacobi_small_ok(uk, m, mdyn[i]) ? JR_SMALL_ROWS : 0;
      hipLaunchKernelGGL(mid_route_flag_kernel, dim3((nw_ + 255) / 256), dim3(256), 0, stream_, (const int *)mdyn[i], mmul[i], m, lo,
                         MID_HI, nw_, midflag, nmid);
      PG_CHECK_HIP(hipGetLastError());
      Bt = alloc_ten(GS, GS, 1);
      prof_begin(PROF_TRUNC_GRAM, 0.0, 0.0);
      constexpr bool no_fused_mid = false;
      if constexpr (sizeof(T) == 4) {
        if (!no_fused_mid)   // G = M M^T and its Cholesky in one kernel, G resident in LDS (trunc_mid.h)
          launch_mid_gram_chol<T>(stream_, nw_, (const T *)M.p, M.n, uk, (const int *)nmid, (const int *)midflag, GS, Bt.p, Bt.n, mB);
      }
      const bool fused_mid_ran = sizeof(T) == 4 && !no_fused_mid;
      if (!fused_mid_ran || GS > 128) {
        // the walkers the fused kernel does not take (more than 128 live rows; all of the route without it): Gram through HBM
        int *hiflag = midflag, *nhi = nmid;
        if (fused_mid_ran) {
          hiflag = (int *)arena_.alloc(sizeof(int) * nw_);
          nhi = (int *)arena_.alloc(sizeof(int) * nw_);
          hipLaunchKernelGGL(mid_route_flag_kernel, dim3((nw_ + 255) / 256), dim3(256), 0, stream_, (const int *)mdyn[i], mmul[i], m, 128,
                             MID_HI, nw_, hiflag, nhi);
          PG_CHECK_HIP(hipGetLastError());
        }
        double *Gm = (double *)arena_.alloc(sizeof(double) * (size_t)GS * GS * nw_);
        TGemmDesc g;
        g.I[2] = m; g.sAi[2] = uk; g.sCi[2] = GS;
        g.K[2] = uk; g.sAk[2] = 1; g.sBk[2] = 1;
        g.J[2] = m; g.sBj[2] = uk; g.sCj[2] = 1;
        g.wA = M.n; g.wB = M.n; g.wC = (long)GS * GS; g.nbatch = nw_;
        g.dI[2].p = nhi; g.dJ[2].p = nhi;
        g.upper_only = 1;
        g.batch_flag = hiflag;
        constexpr bool no_rowgram = false;
        constexpr bool no_two_level = false;
        bool rowgram = false;
        if constexpr (sizeof(T) == 4) {
          if (!no_rowgram && uk % 16 == 0 && M.n % 4 == 0 && m <= 256) {   // streaming wave-per-block kernel (gram.h)
            // Round 6: the first compression as a diagonally PIVOTED factorisation stopped after pivot_cap rows (chol_pivot.h): the
            // truncation keeps chi of the directions, the pivot order puts the dominant ones first -- no walker keeps more than 64 rows,
            // so the second level below never sees the 65..128-row class nor the > 128-row stragglers.  PEPSGPU_PIVOT_CHOL=0: the full
            // factorisation in the natural order (round 3-5); = 56 (default) / 64: the cap (measured, real leg at 8192 walkers: 2 371 amp/s
            // without, 2 626 with 64 rows at two blocks per SIMD, 2 699 with 56 at three; graded subspace error of the prototype 3.7e-8 / 6e-8
            // median against 2.1e-7 of the unpivoted factor).
            static const int pivot_cap = getenv("PEPSGPU_PIVOT_CHOL") ? atoi(getenv("PEPSGPU_PIVOT_CHOL")) : 56;
            // (the cap leaves chi + 24 rows of oversampling: 56 rows up to chi = 32 -- three blocks per SIMD --, 64 up to chi = 40)
            const int kf = std::min(chi_, std::min(m, uk));
            const int kcap = (kf + 24 <= std::min(64, pivot_cap)) ? std::min(64, pivot_cap) : 64;
            pivoted = pivot_cap > 0 && fused_mid_ran && GS > 128 && !no_two_level && gram_rows_i8_ok(M.p, m) && kf + 24 <= kcap;
            launch_gram_rows_f64<T>(stream_, nw_, (const T *)M.p, M.n, uk, m, (const int *)nhi, Gm, (long)GS * GS, GS,
                                    (const int *)hiflag, tg_flop_counter, tg_byte_counter, pivoted ? 1 : 0);
            rowgram = true;
            if (pivoted)
              launch_chol_pivot<T>(stream_, nw_, (const double *)Gm, (long)GS * GS, GS, Bt.p, Bt.n, mB, GS, (const int *)nhi, 1, (const int *)hiflag,
                                   kcap);
          }
        }
        if (!rowgram) tgemm_launch<T, T, double, double>(stream_, g, M.p, M.p, Gm);
        if (!pivoted) {
          const size_t smem = chol_smem_bytes(GS);
          allow_dynamic_lds(reinterpret_cast<const void *>(&chol_upper_kernel<T>), smem);
          launch_chol_upper<T>(stream_, nw_, Gm, (long)GS * GS, GS, Bt.p, Bt.n, mB, 0, GS, (const int *)nhi, 1, (const int *)hiflag);
        }
        arena_.free(Gm);
        // Second level (walkers with more than 128 live rows of M whose factor B kept at most 128 rows -- the usual case: the
        // truncation input of a real PEPS is of numerical rank 60-100): the rows of B are as long as M has live rows (up to
        // 256), so the same compression is applied once more, B2^T B2 = B B^T (r x r, LDS resident: the fused kernel with B in
        // the place of M), and the Jacobi runs on the r x r factor B2 (rows <= 128 long: the sixteen-lanes-per-row tournament at
        // its fast size).  Rotated rows of B2 = sigma_k w_k^T (w: left singular vectors of B); sigma_k u_k^T = w_k^T B.
        if constexpr (sizeof(T) == 4) {
          if (fused_mid_ran && GS > 128 && !no_two_level) {
            two_level = true;
            flagA = (int *)arena_.alloc(sizeof(int) * nw_);
            rowsA = (int *)arena_.alloc(sizeof(int) * nw_);
            flag2 = (int *)arena_.alloc(sizeof(int) * nw_);
            rows2 = (int *)arena_.alloc(sizeof(int) * nw_);
            mB2 = (int *)arena_.alloc(sizeof(int) * nw_);
            PG_CHECK_HIP(hipMemsetAsync(mB2, 0, sizeof(int) * nw_, stream_));
            big_list = (int *)arena_.alloc(sizeof(int) * (nw_ + 1));
            PG_CHECK_HIP(hipMemsetAsync(big_list + nw_, 0, sizeof(int), stream_));     // the count sits behind the list
            hipLaunchKernelGGL(mid_split_kernel, dim3((nw_ + 255) / 256), dim3(256), 0, stream_, (const int *)midflag, (const int *)hiflag,
                               (const int *)mB, 128, nw_, flagA, rowsA, flag2, rows2, big_list, big_list + nw_);
            PG_CHECK_HIP(hipGetLastError());
            B2 = alloc_ten(128, 128, 1);
            launch_mid_gram_chol<T>(stream_, nw_, (const T *)Bt.p, Bt.n, GS, (const int *)rows2, (const int *)flag2, 128, B2.p, B2.n, mB2);
            if (dbg_sweeps_ && getenv("PEPSGPU_DEBUG_VERBOSE")) {   // diagnostics: rows kept by the two compressions
              std::vector<int> h1(nw_), h2(nw_), hm(nw_);
              PG_CHECK_HIP(hipMemcpyAsync(h1.data(), mB, nw_ * sizeof(int), hipMemcpyDeviceToHost, stream_));
              PG_CHECK_HIP(hipMemcpyAsync(h2.data(), mB2, nw_ * sizeof(int), hipMemcpyDeviceToHost, stream_));
              PG_CHECK_HIP(hipMemcpyAsync(hm.data(), nmid, nw_ * sizeof(int), hipMemcpyDeviceToHost, stream_));
              PG_CHECK_HIP(hipStreamSynchronize(stream_));
              long s0 = 0, s1 = 0, s2 = 0, x0 = 0, x1 = 0, x2 = 0;
              for (int w = 0; w < nw_; ++w) { s0 += hm[w]; s1 += h1[w]; s2 += h2[w]; x0 = std::max<long>(x0, hm[w]); x1 = std::max<long>(x1, h1[w]); x2 = std::max<long>(x2, h2[w]); }
              fprintf(stderr, "[pepsgpu] trunc site %d: live rows of M mean %.1f max %ld -> B mean %.1f max %ld -> B2 mean %.1f max %ld\n", i,
                      (double)s0 / nw_, x0, (double)s1 / nw_, x1, (double)s2 / nw_, x2);
            }
          }
        }
        if (fused_mid_ran) { arena_.free(hiflag); arena_.free(nhi); }
      }
      prof_end();
    }
    // static size of the new bond and the tensor it leads to (before the Jacobi: the kernel of the walkers with few live rows
    // selects and normalises their rows into V itself)
    const int k_full = std::min(chi_, std::min(m, uk));
    int k = k_full;
    if (!full_bonds && bond_adapt && cur_kmax[i] >= 0) {
      const int want = cur_kmax[i] + std::max(2, cur_kmax[i] / 4);
      k = std::min(k_full, (want + 3) & ~3);
    }
    kstat[i] = k; kfull[i] = k_full;
    PG_REQUIRE(m <= 1024, 1, "bond dimension too large for select_rows_kernel");
    DTen<T> V = alloc_ten(k, u, k2);
    if (bond_adapt) kn[i] = (int *)arena_.alloc(sizeof(int) * nw_);
    // ---- float64 engine, dense site: two-level preconditioned truncation with oversampling (round 5) ----------------------------
    // The f64 mode on a dense state spent 98 % of its time in the general one-sided Jacobi on the 256 x 256 block M (25 amp/s at C4:
    // with the rows in global memory a sweep is 255 passes over the matrix).  The Gram-preconditioned route of the f32 engine cannot
    // be taken over as it is: the Cholesky of M M^T in float64 perturbs the boundary between the kept direction chi and the discarded
    // direction chi + 1 by ~3e-14 s_1^2 / (s_chi^2 - s_chi+1^2), i.e. ~5e-9 per truncation at s_chi / s_1 = 2e-5 -- too much for the
    // 1e-8 parity of this mode.  With OVERSAMPLING it can: the two Gram + Cholesky compressions (B^T B = M M^T, B2^T B2 = B B^T) and
    // the Jacobi on the small factor B2 only have to deliver a subspace U of kq = 2 chi dimensions that CONTAINS the top-chi left
    // singular subspace -- the mixing that matters is then between direction chi and direction 2 chi + 1, smaller by
    // s_2chi+1 / s_chi+1 and with a gap of s_chi^2 (~1e-10 per truncation on the real state) -- and the exact top-chi singular
    // vectors inside it come from an accurate float64 Jacobi on Z = U^T M, kq x uk (Rayleigh-Ritz on M itself).  Both Jacobi problems
    // (<= 128 x 128 and 64 x 256 doubles) live in LDS.  Walkers whose factors keep fewer than kq (or more than 128) rows take the
    // general kernels as before (rflag = 0); trunc_err > 0 keeps the general path (the truncation rule wants every singular value).
    int *rflag = nullptr, *fbrows = nullptr, *early = nullptr, *fb_early = nullptr, *lateflag = nullptr;
    if constexpr (std::is_same<T, double>::value) {
      static const bool no_route = getenv("PEPSGPU_NO_F64_DENSE_ROUTE") != nullptr;
      // oversampled subspace: 2 chi directions, at most three quarters of the rank M can have (the right-edge sites are 256 x 64)
      const int kq = std::min(2 * k_full, (3 * std::min(m, uk)) / 4);
      // Round 6: the subspace from a diagonally PIVOTED factorisation of G = M M^T stopped after kq rows (chol_pivot.h; measured on
      // the truncation inputs of the real state in float64, scripts/proto_subspace.py: the kept sigma_k v_k lost by the subspace of 64
      // pivot rows 2.5e-10 median / 4.4e-9 max of sigma_1, 56 rows 1.8e-9 / 1.9e-8), made orthonormal by a Cholesky-QR2 in float64
      // (chol_solve_rows_kernel: U = L^-1 B twice), sharpened by ONE step of subspace iteration on M itself (U <- orth(orth(U M) M^T): the
      // part outside shrinks by (sigma_kq+1 / sigma_chi)^2 ~ 3e-3) and followed by the same accurate Jacobi on Z = U M as before.
      // No Gram-resolution cliff (a pivoted factor simply stops at the numerical rank: C5's synthetic state keeps 30-47 directions and
      // stays on the route) and no Jacobi on a 128 x 128 factor.  PEPSGPU_F64_PIVOT=0: the two-Cholesky route of round 5.
      static const int f64_pivot = getenv("PEPSGPU_F64_PIVOT") ? atoi(getenv("PEPSGPU_F64_PIVOT")) : 1;
      constexpr int f64_pivot_mlo = 63;     // blocks of 64 .. 256 rows (measured, real state at 2 048 walkers: 443 amp/s with the route above 128 rows only, 490 from 64)
      if (f64_pivot && !no_route && adaptive && trunc_err_ == 0.0 && m > f64_pivot_mlo && m <= 256 && uk <= 256 && uk % 4 == 0 && kq <= 64 &&
          kq >= k_full + 8 && i > 0) {
        const int GSd = m;
        const int gb = (nw_ + 255) / 256;
        rflag = (int *)arena_.alloc(sizeof(int) * nw_);
        fbrows = (int *)arena_.alloc(sizeof(int) * nw_);
        int *rowsM = (int *)arena_.alloc(sizeof(int) * nw_), *mB1 = (int *)arena_.alloc(sizeof(int) * nw_);
        double *resid = (double *)arena_.alloc(sizeof(double) * nw_);
        PG_CHECK_HIP(hipMemsetAsync(mB1, 0, sizeof(int) * nw_, stream_));
        hipLaunchKernelGGL(f64_route_init_kernel, dim3(gb), dim3(256), 0, stream_, (const int *)mdyn[i], mmul[i], m, nw_, rowsM, rflag);
        PG_CHECK_HIP(hipGetLastError());
        prof_begin(PROF_TRUNC_GRAM, 0.0, 0.0);
        double *Gm = (double *)arena_.alloc(sizeof(double) * (size_t)GSd * GSd * nw_);
        DTen<T> Bq = alloc_ten(64, GSd, 1), Zt = alloc_ten(64, uk, 1);
        double *Sq = (double *)arena_.alloc(sizeof(double) * 64 * 64 * (size_t)nw_);
        auto gram_rows = [&](const DTen<T> &X, int len, const int *rows, int rmax, double *S, int ldS, bool full, const int *lenlive) {
          TGemmDesc g;      // S = X X^T over the live rows (upper triangle unless `full`)
          g.I[2] = rmax; g.sAi[2] = len; g.sCi[2] = ldS;
          g.K[2] = len; g.sAk[2] = 1; g.sBk[2] = 1;
          g.J[2] = rmax; g.sBj[2] = len; g.sCj[2] = 1;
          g.wA = X.n; g.wB = X.n; g.wC = (long)ldS * ldS; g.nbatch = nw_;
          g.dI[2].p = rows; g.dJ[2].p = rows;
          g.dK[2].p = lenlive;
          g.upper_only = full ? 0 : 1;
          g.batch_flag = rflag;
          tgemm_launch<T, T, double, double>(stream_, g, X.p, X.p, S);
        };
        gram_rows(M, uk, rowsM, m, Gm, GSd, true, nullptr);           // both triangles: the pivoted factorisation reads whole rows
        // The factorisation only SELECTS rows of M here (pivot order, down to the rounding noise of G: thresh_scale 0): what the
        // Gram cannot resolve (directions below 2.4e-7 sigma_1 -- C5's synthetic state has ~25 above it for chi = 24: taking the factor
        // itself as the basis left the f64 amplitude at 2.9e-7) comes from the rows themselves, Gram-Schmidt'ed in float64.
        const int slots = chol_pivot_slots(kq);
        int *piv = (int *)arena_.alloc(sizeof(int) * (size_t)slots * nw_);
        launch_chol_pivot<T>(stream_, nw_, (const double *)Gm, (long)GSd * GSd, GSd, Bq.p, Bq.n, mB1, GSd, (const int *)rowsM, 1,
                             (const int *)rflag, kq, resid, 0.0, piv);
        arena_.free(Gm);
        hipLaunchKernelGGL(gather_rows_kernel<T>, dim3(64, nw_), dim3(256), 0, stream_, (const T *)M.p, M.n, uk, (const int *)piv, slots,
                           (const int *)mB1, Zt.p, Zt.n, (const int *)rflag);
        PG_CHECK_HIP(hipGetLastError());
        arena_.free(piv);
        auto orthonormalise = [&](DTen<T> &X, int len, const int *lenlive) {     // Cholesky-QR2 of the mB1 rows of X (in place)
          for (int pass = 0; pass < 2; ++pass) {
            gram_rows(X, len, mB1, 64, Sq, 64, false, lenlive);
            hipLaunchKernelGGL(chol_solve_rows_kernel, dim3(nw_), dim3(256), 0, stream_, (const double *)Sq, 64L * 64, 64, (double *)X.p, X.n,
                               len, (const int *)mB1, (const int *)rflag);
            PG_CHECK_HIP(hipGetLastError());
          }
        };
        orthonormalise(Zt, uk, nullptr);                            // Q0: the selected rows of M, orthonormal (right space)
        prof_end();
        prof_begin(PROF_TRUNC_APPLY, 0.0, 0.0);
        auto times_mt = [&](const DTen<T> &Q, DTen<T> &Uout) {      // U = Q M^T (rows of Q: uk long; rows of U: GSd long, zeros beyond the live rows of M)
          TGemmDesc g;
          g.I[2] = 64; g.sAi[2] = uk; g.sCi[2] = GSd;
          g.K[2] = uk; g.sAk[2] = 1; g.sBk[2] = 1;
          g.J[2] = m; g.sBj[2] = uk; g.sCj[2] = 1;
          g.wA = Q.n; g.wB = M.n; g.wC = Uout.n; g.nbatch = nw_;
          g.dI[2].p = mB1;
          g.dJ[2].p = rowsM; g.dJ[2].mask = 1;
          g.batch_flag = rflag;
          tgemm_launch<T, T, T, double>(stream_, g, Q.p, M.p, Uout.p);
        };
        times_mt(Zt, Bq);                                           // the pivoted factor itself, from M: B = Q0 M^T
        prof_end();
        prof_begin(PROF_TRUNC_GRAM, 0.0, 0.0);
        orthonormalise(Bq, GSd, rowsM);
        prof_end();
        prof_begin(PROF_TRUNC_APPLY, 0.0, 0.0);
        auto times_m = [&](const DTen<T> &U, DTen<T> &Zout) {       // Z = U M (rows of U: GSd long, live part rowsM)
          TGemmDesc g;
          g.I[2] = 64; g.sAi[2] = GSd; g.sCi[2] = uk;
          g.K[2] = m; g.sAk[2] = 1; g.sBk[2] = uk;
          g.J[2] = uk; g.sBj[2] = 1; g.sCj[2] = 1;
          g.wA = U.n; g.wB = M.n; g.wC = Zout.n; g.nbatch = nw_;
          g.dI[2].p = mB1;
          g.dK[2].p = rowsM;
          g.batch_flag = rflag;
          tgemm_launch<T, T, T, double>(stream_, g, U.p, M.p, Zout.p);
        };
        times_m(Bq, Zt);
        prof_end();
        prof_begin(PROF_TRUNC_GRAM, 0.0, 0.0);
        orthonormalise(Zt, uk, nullptr);      // (each half step re-orthonormalised: U M M^T has the SQUARED condition, 1e12 -- no Gram survives it)
        prof_end();
        prof_begin(PROF_TRUNC_APPLY, 0.0, 0.0);
        times_mt(Zt, Bq);                                           // one step of subspace iteration: U <- orth(orth(U M) M^T)
        prof_end();
        prof_begin(PROF_TRUNC_GRAM, 0.0, 0.0);
        orthonormalise(Bq, GSd, rowsM);
        prof_end();
        prof_begin(PROF_TRUNC_APPLY, 0.0, 0.0);
        times_m(Bq, Zt);
        prof_end();
        prof_begin(PROF_JACOBI, 0.0, 0.0);
        {   // the accurate SVD inside the subspace: one-sided Jacobi on the <= kq rows of Z, LDS resident
          const size_t needz = sizeof(T) * (size_t)kq * (uk | 1);
          allow_dynamic_lds(reinterpret_cast<const void *>(&jacobi_rows_kernel<T>), needz);
          hipLaunchKernelGGL(jacobi_rows_kernel<T>, dim3(nw_), dim3(1024), needz, stream_, Zt.p, Zt.n, kq, uk, uk, 40, 1, sweeps_,
                             (const int *)mB1, 1, 0, 0);
          PG_CHECK_HIP(hipGetLastError());
        }
        prof_end();
        prof_begin(PROF_SELECT, 0.0, 0.0);
        hipLaunchKernelGGL(select_rows_kernel<T>, dim3(nw_), dim3(256), 0, stream_, (const T *)Zt.p, Zt.n, kq, uk, uk, k, V.p, V.n,
                           (T *)nullptr, 0L, (const int *)mB1, 1, kn[i], 0.0, chi_min_, (double *)nullptr, (const int *)rflag, 1);
        PG_CHECK_HIP(hipGetLastError());
        // guard: a cap that cut into the spectrum (resid > 0) is priced by what one step of subspace iteration leaves of it,
        // resid (sigma_1 / sigma_chi)^2; a walker above the tolerance takes the general kernel
        hipLaunchKernelGGL(f64_pivot_guard_kernel<double>, dim3(nw_), dim3(256), 0, stream_, (const double *)Zt.p, Zt.n, uk, (const int *)mB1,
                           k_full, (const double *)resid, 3e-2, rflag);
        PG_CHECK_HIP(hipGetLastError());
        prof_end();
        if (dbg_sweeps_ && getenv("PEPSGPU_DEBUG_VERBOSE")) {
          std::vector<int> hf(nw_), hk(nw_);
          std::vector<double> hr(nw_);
          PG_CHECK_HIP(hipMemcpyAsync(hf.data(), rflag, nw_ * sizeof(int), hipMemcpyDeviceToHost, stream_));
          PG_CHECK_HIP(hipMemcpyAsync(hk.data(), mB1, nw_ * sizeof(int), hipMemcpyDeviceToHost, stream_));
          PG_CHECK_HIP(hipMemcpyAsync(hr.data(), resid, nw_ * sizeof(double), hipMemcpyDeviceToHost, stream_));
          PG_CHECK_HIP(hipStreamSynchronize(stream_));
          long on = 0, sk = 0; double rmx = 0.0;
          for (int w = 0; w < nw_; ++w) { on += hf[w] < 0; sk += hk[w]; rmx = std::max(rmx, hr[w]); }
          fprintf(stderr, "[pepsgpu] f64 pivoted route site %d (m = %d, uk = %d, kq = %d): %ld of %d walkers on the route, pivot rows mean %.1f, residual pivot max %.2e\n",
                  i, m, uk, kq, on, nw_, (double)sk / nw_, rmx);
        }
        lateflag = (int *)arena_.alloc(sizeof(int) * nw_);
        hipLaunchKernelGGL(f64_route_fallback_kernel, dim3(gb), dim3(256), 0, stream_, (const int *)rflag, (const int *)rowsM, nw_, fbrows,
                           (const int *)nullptr, lateflag);
        PG_CHECK_HIP(hipGetLastError());
        free_ten(Bq); free_ten(Zt);
        arena_.free(Sq); arena_.free(rowsM); arena_.free(mB1); arena_.free(resid);
      } else if (!no_route && adaptive && trunc_err_ == 0.0 && m > 128 && m <= 256 && uk <= 256 && kq <= 64 && kq >= k_full + 8 && i > 0) {
        const int GSd = m;
        // a walker stays on the route with as few as chi + 4 directions above the resolution of a Gram: the guard prices what its
        // factors dropped (C5: the synthetic fermionic state keeps 30-47 of kq = 48; real state: the edge sites)
        const int route_lo = std::min(kq, k_full + 4);
        rflag = (int *)arena_.alloc(sizeof(int) * nw_);
        fbrows = (int *)arena_.alloc(sizeof(int) * nw_);
        int *rowsM = (int *)arena_.alloc(sizeof(int) * nw_), *mB1 = (int *)arena_.alloc(sizeof(int) * nw_);
        int *mB2 = (int *)arena_.alloc(sizeof(int) * nw_), *kW = (int *)arena_.alloc(sizeof(int) * nw_);
        PG_CHECK_HIP(hipMemsetAsync(mB1, 0, sizeof(int) * nw_, stream_));
        PG_CHECK_HIP(hipMemsetAsync(mB2, 0, sizeof(int) * nw_, stream_));
        PG_CHECK_HIP(hipMemsetAsync(kW, 0, sizeof(int) * nw_, stream_));
        const int gb = (nw_ + 255) / 256;
        hipLaunchKernelGGL(f64_route_init_kernel, dim3(gb), dim3(256), 0, stream_, (const int *)mdyn[i], mmul[i], m, nw_, rowsM, rflag);
        PG_CHECK_HIP(hipGetLastError());
        const bool rdbg = dbg_sweeps_ && getenv("PEPSGPU_DEBUG_VERBOSE");
        long stage_on[3] = {0, 0, 0}, stage_hi = 0, stage_lo = 0;
        auto count_on = [&](int st, const int *rows_after) {     // diagnostics: walkers still on the route after a stage
          if (!rdbg) return;
          std::vector<int> hf(nw_), hr(nw_);
          PG_CHECK_HIP(hipMemcpyAsync(hf.data(), rflag, nw_ * sizeof(int), hipMemcpyDeviceToHost, stream_));
          if (rows_after) PG_CHECK_HIP(hipMemcpyAsync(hr.data(), rows_after, nw_ * sizeof(int), hipMemcpyDeviceToHost, stream_));
          PG_CHECK_HIP(hipStreamSynchronize(stream_));
          for (int w = 0; w < nw_; ++w) stage_on[st] += hf[w] < 0;
          (void)hr;
        };
        prof_begin(PROF_TRUNC_GRAM, 0.0, 0.0);
        double *Gm = (double *)arena_.alloc(sizeof(double) * (size_t)GSd * GSd * nw_);
        DTen<T> B1 = alloc_ten(GSd, GSd, 1);
        {   // G = M M^T over the live rows, upper triangle
          TGemmDesc g;
          g.I[2] = m; g.sAi[2] = uk; g.sCi[2] = GSd;
          g.K[2] = uk; g.sAk[2] = 1; g.sBk[2] = 1;
          g.J[2] = m; g.sBj[2] = uk; g.sCj[2] = 1;
          g.wA = M.n; g.wB = M.n; g.wC = (long)GSd * GSd; g.nbatch = nw_;
          g.dI[2].p = rowsM; g.dJ[2].p = rowsM;
          g.upper_only = 1;
          tgemm_launch<T, T, double, double>(stream_, g, M.p, M.p, Gm);
        }
        // (tscale: the pivot threshold of the first factorisation for EVERY walker -- an experiment constant of round 5: fewer kept
        // rows = smaller Jacobi problems, priced by the guard)
        constexpr double tscale = 1.0;
        launch_chol_upper<T>(stream_, nw_, Gm, (long)GSd * GSd, GSd, B1.p, B1.n, mB1, 0, GSd, (const int *)rowsM, 1, (const int *)nullptr, tscale);
        // Second chance for the walkers whose factor kept more than 128 rows (1-3 of 1 024 per site on the real state -- each of them
        // would otherwise cost a whole general Jacobi, ~40 ms per site whatever the batch): the Gram again (the factorisation works in
        // place) and the factor with the pivot threshold REDO_SCALE times higher, i.e. directions below sqrt(REDO_SCALE) 2.4e-7 s_1
        // dropped; the guard prices exactly that for them.  Who still keeps more than 128 rows leaves the route.
        constexpr double REDO_SCALE = 64.0;
        int *redo = (int *)arena_.alloc(sizeof(int) * nw_), *lvl = (int *)arena_.alloc(sizeof(int) * nw_);
        hipLaunchKernelGGL(f64_route_redo_kernel, dim3(gb), dim3(256), 0, stream_, (const int *)mB1, 128, nw_, redo, lvl);
        PG_CHECK_HIP(hipGetLastError());
        {
          TGemmDesc g;
          g.I[2] = m; g.sAi[2] = uk; g.sCi[2] = GSd;
          g.K[2] = uk; g.sAk[2] = 1; g.sBk[2] = 1;
          g.J[2] = m; g.sBj[2] = uk; g.sCj[2] = 1;
          g.wA = M.n; g.wB = M.n; g.wC = (long)GSd * GSd; g.nbatch = nw_;
          g.dI[2].p = rowsM; g.dJ[2].p = rowsM;
          g.upper_only = 1;
          g.batch_flag = redo;
          tgemm_launch<T, T, double, double>(stream_, g, M.p, M.p, Gm);
        }
        launch_chol_upper<T>(stream_, nw_, Gm, (long)GSd * GSd, GSd, B1.p, B1.n, mB1, 0, GSd, (const int *)rowsM, 1, (const int *)redo, tscale * REDO_SCALE);
        // ... and a third one at REDO_SCALE^2 for what is still above 128 rows (flat spectra: the guard decides whether that is good enough)
        hipLaunchKernelGGL(f64_route_redo_kernel, dim3(gb), dim3(256), 0, stream_, (const int *)mB1, 128, nw_, redo, lvl, 2);
        PG_CHECK_HIP(hipGetLastError());
        {
          TGemmDesc g;
          g.I[2] = m; g.sAi[2] = uk; g.sCi[2] = GSd;
          g.K[2] = uk; g.sAk[2] = 1; g.sBk[2] = 1;
          g.J[2] = m; g.sBj[2] = uk; g.sCj[2] = 1;
          g.wA = M.n; g.wB = M.n; g.wC = (long)GSd * GSd; g.nbatch = nw_;
          g.dI[2].p = rowsM; g.dJ[2].p = rowsM;
          g.upper_only = 1;
          g.batch_flag = redo;
          tgemm_launch<T, T, double, double>(stream_, g, M.p, M.p, Gm);
        }
        launch_chol_upper<T>(stream_, nw_, Gm, (long)GSd * GSd, GSd, B1.p, B1.n, mB1, 0, GSd, (const int *)rowsM, 1, (const int *)redo,
                             tscale * REDO_SCALE * REDO_SCALE);
        arena_.free(redo);
        arena_.free(Gm);
        // walkers whose first factor kept more than 128 or fewer than kq rows leave the route
        if (rdbg) {
          std::vector<int> hr(nw_);
          PG_CHECK_HIP(hipMemcpyAsync(hr.data(), mB1, nw_ * sizeof(int), hipMemcpyDeviceToHost, stream_));
          PG_CHECK_HIP(hipStreamSynchronize(stream_));
          for (int w = 0; w < nw_; ++w) { stage_hi += hr[w] > 128; stage_lo += hr[w] < route_lo; }
        }
        hipLaunchKernelGGL(f64_route_check_kernel, dim3(gb), dim3(256), 0, stream_, rflag, mB1, route_lo, 128, nw_);
        PG_CHECK_HIP(hipGetLastError());
        count_on(0, nullptr);
        // The few walkers that leave here (1-3 of 1 024 per site with more than 128 rows, some tens at the edge sites) each cost a whole
        // general Jacobi from global memory, ~50 ms per site whatever the batch: it starts NOW on the side stream, beside the route.
        constexpr bool no_side = false;
        if (!no_side) {
          early = (int *)arena_.alloc(sizeof(int) * nw_);
          fb_early = (int *)arena_.alloc(sizeof(int) * nw_);
          PG_CHECK_HIP(hipMemcpyAsync(early, rflag, sizeof(int) * nw_, hipMemcpyDeviceToDevice, stream_));
          hipLaunchKernelGGL(f64_route_fallback_kernel, dim3(gb), dim3(256), 0, stream_, (const int *)rflag, (const int *)rowsM, nw_, fb_early,
                             (const int *)nullptr, (int *)nullptr);
          PG_CHECK_HIP(hipGetLastError());
          PG_CHECK_HIP(hipEventRecord(ev_fork_, stream_));
          PG_CHECK_HIP(hipStreamWaitEvent(side_stream_, ev_fork_, 0));
          constexpr int CAPS = 64 * 1024;
          allow_dynamic_lds(reinterpret_cast<const void *>(&jacobi_rows_kernel<T>), (size_t)CAPS);
          hipLaunchKernelGGL(jacobi_rows_kernel<T>, dim3(nw_), dim3(1024), CAPS, side_stream_, M.p, M.n, m, uk, uk, 40, 2, sweeps_,
                             (const int *)fb_early, 1, 0, 0, CAPS);
          hipLaunchKernelGGL(select_rows_kernel<T>, dim3(nw_), dim3(256), 0, side_stream_, (const T *)M.p, M.n, m, uk, uk, k, V.p, V.n,
                             (T *)nullptr, 0L, (const int *)mdyn[i], mmul[i], kn[i], trunc_err_, chi_min_, (double *)nullptr,
                             (const int *)early, 0);
          PG_CHECK_HIP(hipGetLastError());
          PG_CHECK_HIP(hipEventRecord(ev_join_, side_stream_));
        }
        double *G2 = (double *)arena_.alloc(sizeof(double) * (size_t)128 * 128 * nw_);
        DTen<T> B2 = alloc_ten(128, 128, 1);
        {   // G2 = B B^T (r x r, r = mB1 <= 128), the rows of B are GSd long (zero beyond the live rows of M)
          TGemmDesc g;
          g.I[2] = 128; g.sAi[2] = GSd; g.sCi[2] = 128;
          g.K[2] = GSd; g.sAk[2] = 1; g.sBk[2] = 1;
          g.J[2] = 128; g.sBj[2] = GSd; g.sCj[2] = 1;
          g.wA = B1.n; g.wB = B1.n; g.wC = 128L * 128; g.nbatch = nw_;
          g.dI[2].p = mB1; g.dJ[2].p = mB1;
          g.dK[2].p = rowsM;
          g.upper_only = 1;
          g.batch_flag = rflag;
          tgemm_launch<T, T, double, double>(stream_, g, B1.p, B1.p, G2);
        }
        launch_chol_upper<T>(stream_, nw_, G2, 128L * 128, 128, B2.p, B2.n, mB2, 0, 128, (const int *)mB1, 1, (const int *)rflag);
        arena_.free(G2);
        hipLaunchKernelGGL(f64_route_check_kernel, dim3(gb), dim3(256), 0, stream_, rflag, mB2, route_lo, 128, nw_);
        PG_CHECK_HIP(hipGetLastError());
        count_on(1, nullptr);
        prof_end();
        // rotated rows of B2 = sigma_q w_q^T (LDS-resident Jacobi: 128 x 129 doubles)
        prof_begin(PROF_JACOBI, 0.0, 0.0);
        {
          const size_t need2 = sizeof(T) * (size_t)128 * 129;
          allow_dynamic_lds(reinterpret_cast<const void *>(&jacobi_rows_kernel<T>), need2);
          hipLaunchKernelGGL(jacobi_rows_kernel<T>, dim3(nw_), dim3(1024), need2, stream_, B2.p, B2.n, 128, 128, 128, 40, 1, sweeps_,
                             (const int *)mB2, 1, 0, 0);
          PG_CHECK_HIP(hipGetLastError());
        }
        prof_end();
        DTen<T> Wt = alloc_ten(kq, 128, 1), T1 = alloc_ten(kq, GSd, 1), Uq = alloc_ten(kq, GSd, 1), Zt = alloc_ten(kq, uk, 1);
        prof_begin(PROF_SELECT, 0.0, 0.0);
        hipLaunchKernelGGL(select_rows_kernel<T>, dim3(nw_), dim3(256), 0, stream_, (const T *)B2.p, B2.n, 128, 128, 128, kq, Wt.p, Wt.n,
                           (T *)nullptr, 0L, (const int *)mB2, 1, kW, 0.0, 0, (double *)nullptr, (const int *)rflag, 1);
        PG_CHECK_HIP(hipGetLastError());
        prof_end();
        prof_begin(PROF_TRUNC_APPLY, 0.0, 0.0);
        {   // sigma_q u_q^T = w_q^T B
          TGemmDesc g;
          g.I[2] = kq; g.sAi[2] = 128; g.sCi[2] = GSd;
          g.K[2] = 128; g.sAk[2] = 1; g.sBk[2] = GSd;
          g.J[2] = GSd; g.sBj[2] = 1; g.sCj[2] = 1;
          g.wA = Wt.n; g.wB = B1.n; g.wC = T1.n; g.nbatch = nw_;
          g.dK[2].p = mB1;
          g.batch_flag = rflag;
          tgemm_launch<T, T, T, double>(stream_, g, Wt.p, B1.p, T1.p);
        }
        prof_end();
        prof_begin(PROF_SELECT, 0.0, 0.0);
        hipLaunchKernelGGL(select_rows_kernel<T>, dim3(nw_), dim3(256), 0, stream_, (const T *)T1.p, T1.n, kq, GSd, GSd, kq, Uq.p, Uq.n,
                           (T *)nullptr, 0L, (const int *)kW, 1, (int *)nullptr, 0.0, 0, (double *)nullptr, (const int *)rflag, 1);
        PG_CHECK_HIP(hipGetLastError());
        prof_end();
        prof_begin(PROF_TRUNC_APPLY, 0.0, 0.0);
        {   // Z = U^T M (kq x uk): its rows span the oversampled subspace exactly (float64 product with M itself)
          TGemmDesc g;
          g.I[2] = kq; g.sAi[2] = GSd; g.sCi[2] = uk;
          g.K[2] = m; g.sAk[2] = 1; g.sBk[2] = uk;
          g.J[2] = uk; g.sBj[2] = 1; g.sCj[2] = 1;
          g.wA = Uq.n; g.wB = M.n; g.wC = Zt.n; g.nbatch = nw_;
          g.dK[2].p = rowsM;
          g.batch_flag = rflag;
          tgemm_launch<T, T, T, double>(stream_, g, Uq.p, M.p, Zt.p);
        }
        prof_end();
        prof_begin(PROF_JACOBI, 0.0, 0.0);
        {   // the accurate SVD inside the subspace: one-sided Jacobi on the kq rows of Z, LDS resident (64 x 257 doubles)
          const size_t needz = sizeof(T) * (size_t)kq * (uk | 1);
          allow_dynamic_lds(reinterpret_cast<const void *>(&jacobi_rows_kernel<T>), needz);
          hipLaunchKernelGGL(jacobi_rows_kernel<T>, dim3(nw_), dim3(1024), needz, stream_, Zt.p, Zt.n, kq, uk, uk, 40, 1, sweeps_,
                             (const int *)kW, 1, 0, 0);
          PG_CHECK_HIP(hipGetLastError());
        }
        prof_end();
        prof_begin(PROF_SELECT, 0.0, 0.0);
        hipLaunchKernelGGL(select_rows_kernel<T>, dim3(nw_), dim3(256), 0, stream_, (const T *)Zt.p, Zt.n, kq, uk, uk, k, V.p, V.n,
                           (T *)nullptr, 0L, (const int *)kW, 1, kn[i], 0.0, chi_min_, (double *)nullptr, (const int *)rflag, 1);
        PG_CHECK_HIP(hipGetLastError());
        prof_end();
        // guard (f64_route_guard_kernel): a spectrum that falls to the resolution of a Gram inside the subspace leaves the route
        constexpr double guard_tol = 1e-10;
        hipLaunchKernelGGL(f64_route_guard_kernel<double>, dim3(nw_), dim3(256), 0, stream_, (const double *)Zt.p, Zt.n, uk, (const int *)kW, k_full,
                           guard_tol, rflag, kq, (const int *)lvl, 5.7e-14 * tscale * REDO_SCALE, 5.7e-14 * tscale * REDO_SCALE * REDO_SCALE,
                           5.7e-14 * tscale, tscale > 1.0 ? 1 : 0);
        PG_CHECK_HIP(hipGetLastError());
        if (dbg_sweeps_ && getenv("PEPSGPU_DEBUG_VERBOSE")) {   // diagnostics: who stays on the route, rows kept by the two compressions
          std::vector<int> hf(nw_), h0(nw_), hk(nw_);
          PG_CHECK_HIP(hipMemcpyAsync(hf.data(), rflag, nw_ * sizeof(int), hipMemcpyDeviceToHost, stream_));
          PG_CHECK_HIP(hipMemcpyAsync(h0.data(), rowsM, nw_ * sizeof(int), hipMemcpyDeviceToHost, stream_));
          PG_CHECK_HIP(hipMemcpyAsync(hk.data(), kW, nw_ * sizeof(int), hipMemcpyDeviceToHost, stream_));
          PG_CHECK_HIP(hipStreamSynchronize(stream_));
          long on = 0, s0 = 0, sk = 0, x0 = 0;
          for (int w = 0; w < nw_; ++w) { on += hf[w] < 0; s0 += h0[w]; sk += hf[w] < 0 ? hk[w] : 0; x0 = std::max<long>(x0, h0[w]); }
          fprintf(stderr, "[pepsgpu] f64 dense route site %d (m = %d, uk = %d, kq = %d): %ld of %d walkers on the route (after the first factor %ld: %ld above 128 rows, %ld below kq; after the second %ld), live rows of M mean %.1f max %ld, kept directions mean %.1f\n",
                  i, m, uk, kq, on, nw_, stage_on[0], stage_hi, stage_lo, stage_on[1], (double)s0 / nw_, x0, on ? (double)sk / on : 0.0);
        }
        // the others: the general kernels below on their live rows (the route's walkers count zero rows there)
        lateflag = (int *)arena_.alloc(sizeof(int) * nw_);
        hipLaunchKernelGGL(f64_route_fallback_kernel, dim3(gb), dim3(256), 0, stream_, (const int *)rflag, (const int *)rowsM, nw_, fbrows,
                           (const int *)early, lateflag);
        PG_CHECK_HIP(hipGetLastError());
        free_ten(B1); free_ten(B2); free_ten(Wt); free_ten(T1); free_ten(Uq); free_ten(Zt);
        arena_.free(rowsM); arena_.free(mB1); arena_.free(mB2); arena_.free(kW); arena_.free(lvl);
      }
    }
    bool sel_done = false;
    {
      const size_t need = sizeof(T) * (size_t)m * (uk | 1);
      const int use_lds = need <= JACOBI_LDS_MAX;
      if (use_lds) allow_dynamic_lds(reinterpret_cast<const void *>(&jacobi_rows_kernel<T>), need);
      {   // reference op: gesdd of the (m x uk) block: 4 r c^2 + 22 c^3, r >= c
        const double rr = std::max(m, uk), cc = std::min(m, uk);
        // category 3 = register kernel (bulk blocks), 7 = generic LDS/global kernel (edge blocks)
        const bool bulk = sizeof(T) == 4 && ((!use_lds && m <= 256 && uk <= 256) || mid);
        prof_begin(bulk ? PROF_JACOBI : PROF_JACOBI_EDGE, nw_ * (4.0 * rr * cc * cc + 22.0 * cc * cc * cc), 0.0);
      }
      JrSelect jsel;
      constexpr bool no_jsel = false;
      if constexpr (sizeof(T) == 4) {
        if (!no_jsel && kn[i]) { jsel.V = (float *)V.p; jsel.wV = V.n; jsel.k = k; jsel.klive_out = kn[i]; jsel.trunc_err = trunc_err_; jsel.dmin = chi_min_; }
      }
      // Size classes of the Jacobi kernels above the carry rank of the row absorbed before (+ margin) are not launched (each
      // is a launch of nw blocks that return at once); the live counts read back at the end of this absorption verify it,
      // a miss redoes the absorption without hints (absorb_svd).
      int rows_cap = 0;
      if constexpr (sizeof(T) == 4) {
        constexpr bool no_hint_skip = false;
        static const int force_cap = getenv("PEPSGPU_FORCE_ROWS_CAP") ? atoi(getenv("PEPSGPU_FORCE_ROWS_CAP")) : 0;   // tests: a wrong hint
        if (!full_bonds && !no_hint_skip && adaptive && mdyn[i] && !mid) {
          if (force_cap) rows_cap = force_cap;
          else if (in.depth >= 3 && (int)in.mlmax.size() > i && in.mlmax[i] >= 0)
            rows_cap = in.mlmax[i] + 3 <= JR_BR ? JR_BR : (in.mlmax[i] + 6 <= JR_SMALL_ROWS ? JR_SMALL_ROWS : 0);
        }
      }
      assume_rows[i] = rows_cap;
      sel_done = launch_jacobi(M.p, M.n, m, uk, use_lds, need, rflag ? fbrows : mdyn[i], rflag ? 1 : mmul[i], mid ? MID_HI : 0,
                               jsel.V ? &jsel : nullptr, rows_cap);
      if constexpr (sizeof(T) == 4) {
        if (mid) {
          // <= 64 live rows: two waves per walker, else four; rows of 16 lanes, four pairs per wave instruction (jacobi_rows_grp_kernel)
          if (GS <= 128) {
            launch_jacobi_grp<2, 8>(stream_, nw_, (float *)Bt.p, Bt.n, GS, GS, GS, 40, sweeps_, (const int *)mB, 1, 0);
            if (GS > 64) launch_jacobi_grp<4, 8>(stream_, nw_, (float *)Bt.p, Bt.n, GS, GS, GS, 40, sweeps_, (const int *)mB, 1, 64);
          } else if (two_level) {
            // on B itself: the walkers with at most 128 live rows of M (B at most 128 columns wide) and, on the 256 x 256
            // register kernel, those whose factor kept more than 128 rows; on B2: everybody else
            launch_jacobi_grp<2, 8>(stream_, nw_, (float *)Bt.p, Bt.n, GS, 128, GS, 40, sweeps_, (const int *)rowsA, 1, 0);
            launch_jacobi_grp<4, 8>(stream_, nw_, (float *)Bt.p, Bt.n, GS, 128, GS, 40, sweeps_, (const int *)rowsA, 1, 64);
            // (2048 blocks of 144 KB LDS cost ~0.9 ms even when every block returns at once: a small grid walks the list of the
            // walkers that need it, usually empty)
            // ... on the side stream: the few blocks run beside the launches below (which touch other walkers) instead of holding
            // the whole device for ~0.8 ms; joined before the rows of B are selected
            if (!pivoted) {      // (a pivoted first factor keeps at most 64 rows: the list is empty by construction)
              PG_CHECK_HIP(hipEventRecord(ev_fork_, stream_));
              PG_CHECK_HIP(hipStreamWaitEvent(side_stream_, ev_fork_, 0));
              hipLaunchKernelGGL(jacobi_rows_reg256_list_kernel, dim3(std::min(nw_, 128)), dim3(512), 0, side_stream_, (float *)Bt.p, Bt.n, GS, GS,
                                 GS, 40, sweeps_, (const int *)rowsA, 1, 128, (const int *)big_list, (const int *)(big_list + nw_));
              PG_CHECK_HIP(hipEventRecord(ev_join_, side_stream_));
              side_pending = true;
            }
            // (size classes by row length -- <2,4> for r <= 64, <3,5>, <3,6>, <4,8> -- were measured in round 3: 533 -> 576 ms per
            // step of 4096 dense walkers; the tournament is bound by its exchange / reduction latency, not by the FMAs of a pair)
            launch_jacobi_grp<2, 8>(stream_, nw_, (float *)B2.p, B2.n, 128, 128, 128, 40, sweeps_, (const int *)mB2, 1, 0);
            if (!pivoted) launch_jacobi_grp<4, 8>(stream_, nw_, (float *)B2.p, B2.n, 128, 128, 128, 40, sweeps_, (const int *)mB2, 1, 64);
          } else {
            // rows of B up to 256 long (sixteen columns per lane); more than 128 live rows of B: the 256 x 256 register kernel
            launch_jacobi_grp<2, 16>(stream_, nw_, (float *)Bt.p, Bt.n, GS, GS, GS, 40, sweeps_, (const int *)mB, 1, 0);
            launch_jacobi_grp<4, 16>(stream_, nw_, (float *)Bt.p, Bt.n, GS, GS, GS, 40, sweeps_, (const int *)mB, 1, 64);
            hipLaunchKernelGGL(jacobi_rows_reg256_kernel, dim3(nw_), dim3(512), 0, stream_, (float *)Bt.p, Bt.n, GS, GS, GS, 40,
                               sweeps_, (const int *)mB, 1, 128);
          }
          PG_CHECK_HIP(hipGetLastError());
        }
      }
      prof_end();
      ++n_jacobi_;
      if (dbg_sweeps_) {   // diagnostics only: per-launch sweep counts (forces a sync)
        std::vector<int> hs(nw_);
        PG_CHECK_HIP(hipMemcpyAsync(hs.data(), sweeps_, nw_ * sizeof(int), hipMemcpyDeviceToHost, stream_));
        PG_CHECK_HIP(hipStreamSynchronize(stream_));
        long mx = 0, live = 0, sw_sum = 0, live_mx = 0;
        for (int v : hs) { mx = std::max<long>(mx, v & 0xFF); sw_sum += v & 0xFF; live += v >> 8; live_mx = std::max<long>(live_mx, v >> 8); }
        jacobi_sweeps_sum_ += mx;
        jacobi_sweeps_max_ = std::max(jacobi_sweeps_max_, mx);
        if (getenv("PEPSGPU_DEBUG_VERBOSE"))
          fprintf(stderr, "[pepsgpu] jacobi m=%d len=%d sweeps max=%ld mean=%.2f live_rows_mean=%.1f live_rows_max=%ld\n", m, uk, mx,
                  (double)sw_sum / nw_, (double)live / nw_, live_mx);
      }
    }
    prof_begin(PROF_SELECT, 0.0, 0.0);
    // (behind the rank hint "no walker above 16 rows" the short-row Jacobi has selected every walker itself: the launch would
    // return at once for all of them -- 15 us x 160 sites per step of 49 152 walkers; a miss is caught by the same read-back)
    constexpr bool no_sel_skip = false;
    const bool skip_select = sel_done && !mid && !no_sel_skip && assume_rows[i] > 0 && assume_rows[i] <= JR_BR;
    if (!skip_select)
      hipLaunchKernelGGL(select_rows_kernel<T>, dim3(nw_), dim3(256), 0, stream_, (const T *)M.p, M.n, m, uk, uk, k, V.p,
                         V.n, (T *)nullptr, 0L, (const int *)mdyn[i], mmul[i], kn[i], trunc_err_, chi_min_, (double *)nullptr,
                         (const int *)(rflag ? lateflag : midflag), 0, sel_done ? JR_BR : 0);
    PG_CHECK_HIP(hipGetLastError());
    if (rflag) {
      if (early) {      // the side stream's walkers: joined before anything reads V / kn of this site
        PG_CHECK_HIP(hipStreamWaitEvent(stream_, ev_join_, 0));
        // (their buffers go back to the arena: it hands them out to launches on stream_ only, which are ordered behind the join)
        arena_.free(early); arena_.free(fb_early);
        early = nullptr; fb_early = nullptr;
      }
      arena_.free(rflag); arena_.free(fbrows); arena_.free(lateflag);
      rflag = nullptr; fbrows = nullptr; lateflag = nullptr;
    }
    if (mid) {
      // sigma_k u_k^T = the rotated rows of B: the chi largest, normalised -> U^T (k x GS), kB = how many are live
      if (side_pending) { PG_CHECK_HIP(hipStreamWaitEvent(stream_, ev_join_, 0)); side_pending = false; }
      int *kB = (int *)arena_.alloc(sizeof(int) * nw_);
      PG_CHECK_HIP(hipMemsetAsync(kB, 0, sizeof(int) * nw_, stream_));
      Ut = alloc_ten(k, GS, 1);
      hipLaunchKernelGGL(select_rows_kernel<T>, dim3(nw_), dim3(256), 0, stream_, (const T *)Bt.p, Bt.n, GS, GS, GS, k, Ut.p,
                         Ut.n, (T *)nullptr, 0L, (const int *)(two_level ? rowsA : mB), 1, kB, trunc_err_, chi_min_, (double *)nullptr,
                         (const int *)(two_level ? flagA : midflag), 1);
      PG_CHECK_HIP(hipGetLastError());
      prof_end();
      if (two_level) {
        // W = the chi largest rotated rows of B2, normalised (truncation rule applied here); U^T = rows of W B, normalised
        int *kB2 = (int *)arena_.alloc(sizeof(int) * nw_);
        PG_CHECK_HIP(hipMemsetAsync(kB2, 0, sizeof(int) * nw_, stream_));
        DTen<T> W = alloc_ten(k, 128, 1), T1 = alloc_ten(k, GS, 1);
        prof_begin(PROF_SELECT, 0.0, 0.0);
        hipLaunchKernelGGL(select_rows_kernel<T>, dim3(nw_), dim3(256), 0, stream_, (const T *)B2.p, B2.n, 128, 128, 128, k, W.p,
                           W.n, (T *)nullptr, 0L, (const int *)mB2, 1, kB2, trunc_err_, chi_min_, (double *)nullptr,
                           (const int *)flag2, 1);
        PG_CHECK_HIP(hipGetLastError());
        prof_end();
        {
          TGemmDesc g;
          g.I[2] = k; g.sAi[2] = 128; g.sCi[2] = GS;
          g.K[2] = 128; g.sAk[2] = 1; g.sBk[2] = GS;
          g.J[2] = GS; g.sBj[2] = 1; g.sCj[2] = 1;
          g.wA = W.n; g.wB = Bt.n; g.wC = T1.n; g.nbatch = nw_;
          g.dK[2].p = rows2;
          g.batch_flag = flag2;
          prof_begin(PROF_TRUNC_APPLY, 0.0, 0.0);
          tgemm_launch<T, T, T, double>(stream_, g, W.p, Bt.p, T1.p);
          prof_end();
        }
        prof_begin(PROF_SELECT, 0.0, 0.0);
        hipLaunchKernelGGL(select_rows_kernel<T>, dim3(nw_), dim3(256), 0, stream_, (const T *)T1.p, T1.n, k, GS, GS, k, Ut.p,
                           Ut.n, (T *)nullptr, 0L, (const int *)kB2, 1, kB, 0.0, 0, (double *)nullptr, (const int *)flag2, 1);
        PG_CHECK_HIP(hipGetLastError());
        prof_end();
        free_ten(W); free_ten(T1); free_ten(B2);
        arena_.free(kB2); arena_.free(flagA); arena_.free(rowsA); arena_.free(flag2); arena_.free(rows2); arena_.free(mB2); arena_.free(big_list);
      }
      // V' = U^T M (k x uk): row q is sigma_q v_q^T up to the rounding of u_q -- an error of 1e-7 in u_q brings in the
      // dominant directions with weight 1e-7 sigma_1, which is NOT small against a row of size sigma_q << sigma_1.  So the k
      // rows are not normalised as they come: they are handed to the one-sided Jacobi once more (a k-row problem: the
      // one-wave kernels), which restores their mutual orthogonality relative to each row's own norm in one or two
      // sweeps; what is left is contamination by the discarded directions only, of relative size 1e-7.
      DTen<T> Vp = alloc_ten(k, u, k2);
      {
        TGemmDesc g;
        g.I[2] = k; g.sAi[2] = GS; g.sCi[2] = uk;
        g.K[2] = m; g.sAk[2] = 1; g.sBk[2] = uk;
        g.J[2] = uk; g.sBj[2] = 1; g.sCj[2] = 1;
        g.wA = Ut.n; g.wB = M.n; g.wC = Vp.n; g.nbatch = nw_;
        g.dK[2].p = nmid;
        g.batch_flag = midflag;
        prof_begin(PROF_TRUNC_APPLY, 0.0, 0.0);
        tgemm_launch<T, T, T, double>(stream_, g, Ut.p, M.p, Vp.p);   // f64 accumulation: small sigma_q are differences
        prof_end();
      }
      // Round 6: only the span of the k rows leaves the site, so instead of the polishing Jacobi + select_rows (+ the Newton-Schulz step of
      // precise sites) the rows are made orthonormal in float64 in one launch (rows_qr.h: Cholesky-QR of the unit-scaled rows in their
      // order, i.e. Gram-Schmidt from the dominant direction down; live count by the same floor).  PEPSGPU_ROWS_QR=0: rounds 3-5.
      bool qr_done = false;
      if constexpr (sizeof(T) == 4) {
        static const int rows_qr = getenv("PEPSGPU_ROWS_QR") ? atoi(getenv("PEPSGPU_ROWS_QR")) : 1;
        if (rows_qr && kn[i] && rows_qr_ok(k, uk)) {
          prof_begin(PROF_SELECT, 0.0, 0.0);
          launch_rows_qr(stream_, nw_, (const float *)Vp.p, Vp.n, k, uk, (const int *)kB, (float *)V.p, V.n, kn[i], (const int *)midflag);
          prof_end();
          qr_done = true;
        }
      }
      if (!qr_done) {
        {
          const size_t need = sizeof(T) * (size_t)k * (uk | 1);
          const int use_lds = need <= JACOBI_LDS_MAX;
          if (use_lds) allow_dynamic_lds(reinterpret_cast<const void *>(&jacobi_rows_kernel<T>), need);
          prof_begin(PROF_JACOBI, 0.0, 0.0);
          launch_jacobi(Vp.p, Vp.n, k, uk, use_lds, need, kB, 1);      // walkers off the route have kB = 0 rows
          prof_end();
        }
        prof_begin(PROF_SELECT, 0.0, 0.0);   // normalise, count the live rows; the truncation rule was applied on B already
        hipLaunchKernelGGL(select_rows_kernel<T>, dim3(nw_), dim3(256), 0, stream_, (const T *)Vp.p, Vp.n, k, uk, uk, k, V.p,
                           V.n, (T *)nullptr, 0L, (const int *)kB, 1, kn[i], 0.0, 0, (double *)nullptr, (const int *)midflag, 1);
        PG_CHECK_HIP(hipGetLastError());
        prof_end();
      }
      free_ten(Bt); free_ten(Ut); free_ten(Vp);
      if (qr_done) ortho_skip = midflag; else arena_.free(midflag);     // (the walkers rows_qr took are orthonormal already)
      arena_.free(nmid); arena_.free(kB);
      arena_.free(mB);
    } else {
      prof_end();
    }
    free_ten(M);
    if constexpr (sizeof(T) == 4) {
      constexpr int ortho = 1;
      const size_t osm = ortho_rows_smem(k, uk);
      if (ortho && precise_site && k >= 2 && k <= 64 && osm <= 96 * 1024) {
        allow_dynamic_lds(reinterpret_cast<const void *>(&ortho_rows_kernel), osm);
        prof_begin(PROF_SELECT, 0.0, 0.0);
        hipLaunchKernelGGL(ortho_rows_kernel, dim3(nw_), dim3(256), osm, stream_, (float *)V.p, V.n, k, uk, (const int *)kn[i], uk + 1,
                           (const int *)ortho_skip);
        PG_CHECK_HIP(hipGetLastError());
        prof_end();
      }
    }
    if (ortho_skip) { arena_.free(ortho_skip); ortho_skip = nullptr; }
    inject(INJ_V, V.p, V.n);
    out.t[i] = V;
    // Ynew[(l,a),q] = sum_{(u,k2)} Tt[(l,a),(u,k2)] V[q,(u,k2)]
    DTen<T> Yn = alloc_ten(l, a, k);
    {
      TGemmDesc g;
      g.I[1] = l; g.I[2] = a; g.sAi[1] = a * uk; g.sAi[2] = uk; g.sCi[1] = a * k; g.sCi[2] = k;
      g.K[1] = u; g.K[2] = k2; g.sAk[1] = k2; g.sAk[2] = 1; g.sBk[1] = k2; g.sBk[2] = 1;
      g.J[2] = k; g.sBj[2] = uk; g.sCj[2] = 1;
      g.wA = Tt.n; g.wB = V.n; g.wC = Yn.n; g.nbatch = nw_;
      g.dI[2].p = clive[i]; g.dI[2].mask = 1;    // Yn is normalised as a whole: written in full, zeros beyond the live bonds
      g.dK[2].p = kn[i + 1];
      if (tsw) {   // K = (k2, u): u contiguous in Tt (vector loads), k2 contiguous in V
        g.K[1] = k2; g.K[2] = u; g.sAk[1] = u; g.sAk[2] = 1; g.sBk[1] = 1; g.sBk[2] = k2;
        g.dK[2].p = nullptr; g.dK[1].p = kn[i + 1];
      }
      g.dJ[2].p = kn[i]; g.dJ[2].mask = 1;
      g.prefer_tiled = dense_site;
      constexpr bool y_tiled = false;     // experiments: Y on the LDS-tiled f32 kernel
      if (y_tiled) g.prefer_tiled = true;
      // Y on precise sites: 2 = float64 accumulation on the LDS-tiled kernel (f64 matrix cores) + separate normalisation (round 4);
      // 1 = the wave-per-tile kernel with float64 accumulation (tg_direct_body_f64, round 5; the norm stays fused into the launch);
      // 0 = the f32 chain of round 3
      // (measured, round 5, real state at C4, n = 256 vs the f64 mode: mode 1 max 6.8e-6 / median 1.67e-6 at 2 235 amp/s (4 096
      // walkers), mode 2 8.0e-6 / 1.81e-6 at 2 200)
      constexpr int y_mode = 1;
      bool y_f64 = false;
      if constexpr (sizeof(T) == 4) {
        g.acc64 = (precise_site && y_mode == 1) ? 1 : 0;
        y_f64 = precise_site && y_mode == 2;
        // (the wave-per-tile kernel is the one that honours acc64: round 4 left prefer_tiled set on dense sites, so its "mode 1"
        // measurement ran the LDS-tiled f32 kernel there -- the "drain removes a third only" of HISTORY 3e was that, not the drain)
        if (g.acc64) g.prefer_tiled = false;
      }
      // reference op: res[i-1] . (u s)  (bmps_impl.h:254): 2 (m_{i-1} D_u) m_i k_i
      int rp, cp, ddp[4];
      site_rc(i - 1, rp, cp);
      site_dims(rp, cp, ddp);
      // The norm of Yn comes out of the launch that writes it (squares of the stored values, summed in registers) as a
      // per-walker scale that the contraction reading Yn at the next site applies to its own result: no pass over Yn.
      bool fused_norm = false;
      if constexpr (sizeof(T) == 4) {
        constexpr bool no_fn = false;
        if (!no_fn && !acc64 && !y_tiled && !y_f64 && bond_adapt && kn[i] && tgemm_one_block_direct(g)) {
          if (!yscale) yscale = (float *)arena_.alloc(sizeof(float) * nw_);
          g.scale_out = yscale; g.norm_log = out.logscale; g.norm_flag = flag_;
          fused_norm = true;
        }
      }
      prof_begin(PROF_CONTRACT, 2.0 * nw_ * (double)R[i - 1].d[0] * ddp[lu] * (double)m * k, 2.0 * nw_ * (double)la * uk * (double)k);
      if ((acc64 & 8) || y_f64) tgemm_launch<T, T, T, Acc>(stream_, g, Tt.p, V.p, Yn.p);
      else tgemm_launch<T, T, T, T>(stream_, g, Tt.p, V.p, Yn.p);
      prof_end();
      y_scaled = fused_norm;
    }
    if (!y_scaled) {
      prof_begin(PROF_NORM, 0.0, 0.0);
      normalize(Yn.p, Yn.n, Yn.n, nw_, out.logscale);
      prof_end();
    }
    inject(INJ_Y, Yn.p, Yn.n);
    free_ten(Tt);
    Y = Yn;
  }
  if (yscale) arena_.free(yscale);
  out.live = kn;
  out.kmax.assign(N + 1, -1);
  out.mlmax.assign(N, -1);
  out.depth = in.depth + 1;
  bool ok = true;
  if (bond_adapt) {   // one small read-back per absorption: the maximum live count of every new bond and of every carry
    const int ntab = 3 * N + 1;
    std::vector<const int *> htab(ntab, nullptr);
    for (int b = 0; b <= N; ++b) htab[b] = kn[b];
    for (int i = 0; i < N; ++i) htab[N + 1 + i] = mdyn[i];
    for (int i = 0; i < N; ++i) htab[2 * N + 1 + i] = assume_fused[i] ? mdyn[i] : nullptr;   // (read as "any entry negative")
    std::vector<int> hmax(ntab, -1);
    const int **dtab = (const int **)arena_.alloc(sizeof(int *) * ntab);
    int *dmax = (int *)arena_.alloc(sizeof(int) * ntab);
    PG_CHECK_HIP(hipMemcpyAsync(dtab, htab.data(), sizeof(int *) * ntab, hipMemcpyHostToDevice, stream_));
    hipLaunchKernelGGL(max_over_walkers_kernel, dim3(ntab), dim3(256), 0, stream_, (const int *const *)dtab, nw_, dmax, 2 * N + 1);
    PG_CHECK_HIP(hipGetLastError());
    PG_CHECK_HIP(hipMemcpyAsync(hmax.data(), dmax, sizeof(int) * ntab, hipMemcpyDeviceToHost, stream_));
    PG_CHECK_HIP(hipStreamSynchronize(stream_));
    arena_.free(dtab); arena_.free(dmax);
    for (int b = 0; b <= N; ++b) out.kmax[b] = hmax[b];
    for (int i = 0; i < N; ++i) out.mlmax[i] = mdyn[i] ? std::min(R[i].d[0], hmax[N + 1 + i] * mmul[i]) : R[i].d[0];
    if (!ovr_on_) {      // (a BMPSWalker's foreign MPO says nothing about the network's own row)
      int mx = 0;
      for (int i = 0; i < N; ++i) mx = std::max(mx, out.mlmax[i]);
      carry_seen_[pos][num] = mx;
    }
    for (int i = 1; i < N; ++i)
      if (kstat[i] < kfull[i] && out.kmax[i] >= kstat[i]) ok = false;   // a walker filled a shrunk bond: maybe clipped
    for (int i = 0; i < N; ++i)
      if (assume_rows[i] > 0 && out.mlmax[i] > assume_rows[i]) ok = false;   // a rank hint was missed: rows left unrotated
    for (int i = 0; i < N; ++i)
      if (assume_fused[i] && hmax[2 * N + 1 + i] > 0) ok = false;            // a walker the fused factor flagged had no fallback
  }
  for (auto &t : R) arena_.free(t.p);
  {   // the dynamic-extent arrays (several R_i may share one)
    int *last = nullptr;
    for (int *p : mdyn)
      if (p && p != last) { arena_.free(p); last = p; }
  }
  if (ok) ++n_absorb_;
  return ok;
}

}  // namespace pepsgpu
