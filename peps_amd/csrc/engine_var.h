// Variational compression of BMPS x MPO on the device: CompressMPSScheme::VARIATION2Site / VARIATION1Site
// (bmps.h:31-35; BMPS::MultiplyMPO2SiteVariationalCompress_ bmps_impl.h:864-995,
// MultiplyMPO1SiteVariationalCompress_ :997-1172, MakeVariationalInitGuess_ :1174-1212,
// MakeEnvironmentBoundaries_ / GrowRightEnvironments_ :701-743).  Bosonic (the reference asserts !IsFermionic()).
// Element types: real, and -- round 5 -- complex: the Dag() of the reference's environments is a conjugated operand of the
// GEMM (EinView::cj), the SVD / QR are the complex row Jacobi of linalg_cplx.h on static shapes (no live extents, no
// compressing factor: parity-grade like the complex absorption; the reference runs this combination in its Z2 Ising test,
// test_bmps_contractor.cpp:663-673).
//
// Same state as the reference up to the bond gauge:
//  * every contraction is one strided tensor GEMM (ein() below builds the descriptor from leg names; no
//    transposed copy is ever made);
//  * SVD(theta) -> row Jacobi of theta (right move: the rotated rows are sigma_k v_k^T) or of theta^T
//    (left move: sigma_k u_k^T), top rows selected and normalised by select_rows_kernel with the
//    (trunc_err, D_min, D_max) rule; u.s of the final step = t1 . renv (theta itself is not kept);
//  * QR(t2) of the one-site scheme -> the normalised rotated rows of t2^T: an orthonormal basis of the same
//    column space (a rank-deficient t2 leaves zero rows where Householder QR would put arbitrary vectors;
//    the product state has no weight there);
//  * environments are normalised after every growth step and carry a per-walker log-scale; the convergence
//    criteria (sum |s - s_last| / s_0, |r - r_last| / r) are evaluated on the host from the per-walker values,
//    one small read-back per sweep pair.  Walkers move in lockstep: the sweeps stop when EVERY walker meets the
//    criterion (a walker that converged earlier keeps being refined; the reference would stop it there).
#pragma once
#include "engine_impl.h"
#include "linalg_cplx.h"

namespace pepsgpu {

// a tensor operand of ein(): base pointer, batch stride and named legs (dim, stride); `site` picks the
// projected site tensor of (r, c) by the walker's configuration instead of a per-walker buffer
template <typename T>
struct EinView {
  const T *p = nullptr;
  long w = 0;
  int n = 0;
  char nm[6];
  int dim[6], st[6];
  bool site = false;
  int r = 0, c = 0;
  // per-walker live extent of a leg (device array, nullptr = the static dim).  On an operand: the leg is read up to the
  // live extent only.  On the result: msk = 0 the index space is compacted (elements beyond are not written: only ein()
  // calls given the same extent may read the tensor), msk = 1 the static tiling is kept and zeros are written beyond.
  const int *live[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  int msk[6] = {0, 0, 0, 0, 0, 0};
  bool conj = false;        // complex element types: the operand enters conjugated (Dag() of the reference); no-op for real types
  EinView &cj() { conj = true; return *this; }
  int find(char ch) const {
    for (int i = 0; i < n; ++i) if (nm[i] == ch) return i;
    return -1;
  }
  EinView &dyn(char ch, const int *l, int mask = 0) {
    const int i = find(ch);
    if (i >= 0) { live[i] = l; msk[i] = mask; }
    return *this;
  }
};

template <typename T>
static EinView<T> ein_view(const T *p, long w, const char *legs, std::initializer_list<int> dims) {
  EinView<T> v;
  v.p = p; v.w = w; v.n = (int)dims.size();
  int i = 0;
  for (int d : dims) { v.nm[i] = legs[i]; v.dim[i] = d; ++i; }
  long s = 1;
  for (int k = v.n - 1; k >= 0; --k) { v.st[k] = (int)s; s *= v.dim[k]; }
  return v;
}

// C[legs of c] = sum over the legs shared by a and b and absent from c.  a supplies the I side, b the J side.
template <typename T>
void Engine<T>::ein(const EinView<T> &a, const EinView<T> &b, const EinView<T> &c, T *cp) {
  struct Sub { int dim, s0, s1; const int *live; int mask; };   // dim, stride in the first / second tensor of the group
  std::vector<Sub> gi, gj, gk;
  auto live_of = [&](char ch) -> const int * {   // the live extent of a leg may be named on any of the three views
    for (const EinView<T> *v : {&a, &b, &c}) { const int i = v->find(ch); if (i >= 0 && v->live[i]) return v->live[i]; }
    return nullptr;
  };
  for (int x = 0; x < a.n; ++x) {
    if (a.dim[x] == 1) continue;
    const int ic = c.find(a.nm[x]), ib = b.find(a.nm[x]);
    PG_REQUIRE((ic >= 0) != (ib >= 0), 5, "ein: a leg of A must be either kept or contracted");
    if (ic >= 0) { PG_REQUIRE(c.dim[ic] == a.dim[x], 5, "ein: dim mismatch (A,C)"); gi.push_back({a.dim[x], a.st[x], c.st[ic], live_of(a.nm[x]), c.msk[ic]}); }
    else { PG_REQUIRE(b.dim[ib] == a.dim[x], 5, "ein: dim mismatch (A,B)"); gk.push_back({a.dim[x], a.st[x], b.st[ib], live_of(a.nm[x]), 0}); }
  }
  for (int x = 0; x < b.n; ++x) {
    if (b.dim[x] == 1 || a.find(b.nm[x]) >= 0) continue;
    const int ic = c.find(b.nm[x]);
    PG_REQUIRE(ic >= 0 && c.dim[ic] == b.dim[x], 5, "ein: a free leg of B is missing in C");
    gj.push_back({b.dim[x], b.st[x], c.st[ic], live_of(b.nm[x]), c.msk[ic]});
  }
  auto pack = [](std::vector<Sub> &g, bool by_second) {
    // innermost last: order by decreasing stride of the tensor that is written (C) resp. read contiguously (A for K)
    std::sort(g.begin(), g.end(), [&](const Sub &x, const Sub &y) { return by_second ? x.s1 > y.s1 : x.s0 > y.s0; });
    // merge neighbours that are contiguous in both tensors
    for (size_t i = 0; i + 1 < g.size();) {
      if (!g[i].live && !g[i + 1].live && g[i].s0 == g[i + 1].s0 * g[i + 1].dim && g[i].s1 == g[i + 1].s1 * g[i + 1].dim) {
        g[i + 1].dim *= g[i].dim;
        g.erase(g.begin() + i);
      } else ++i;
    }
    PG_REQUIRE(g.size() <= 3, 5, "ein: more than three sub-indices in one group");
  };
  pack(gi, true); pack(gj, true); pack(gk, false);
  TGemmDesc g;
  for (size_t x = 0; x < gi.size(); ++x) { const int o = 3 - (int)gi.size() + (int)x; g.I[o] = gi[x].dim; g.sAi[o] = gi[x].s0; g.sCi[o] = gi[x].s1; g.dI[o].p = gi[x].live; g.dI[o].mask = gi[x].mask; }
  for (size_t x = 0; x < gj.size(); ++x) { const int o = 3 - (int)gj.size() + (int)x; g.J[o] = gj[x].dim; g.sBj[o] = gj[x].s0; g.sCj[o] = gj[x].s1; g.dJ[o].p = gj[x].live; g.dJ[o].mask = gj[x].mask; }
  for (size_t x = 0; x < gk.size(); ++x) { const int o = 3 - (int)gk.size() + (int)x; g.K[o] = gk[x].dim; g.sAk[o] = gk[x].s0; g.sBk[o] = gk[x].s1; g.dK[o].p = gk[x].live; }
  g.wA = a.w; g.wB = b.w; g.wC = c.w; g.nbatch = nw_;
  PG_REQUIRE(!(a.site && a.conj) && !(b.site && b.conj), 5, "ein: a site operand cannot be conjugated");
  g.conjA = a.conj ? 1 : 0;
  g.conjB = b.conj ? 1 : 0;
  const double fl = 2.0 * nw_ * (double)g.Itot() * g.Jtot() * g.Ktot();
  prof_begin(PROF_CONTRACT, fl, fl);   // (executed flops: counted on the device over the live extents)
  PG_REQUIRE(!(a.site && b.site), 5, "ein: two site operands");
  if (a.site) launch_site_gemm_a(g, cfg_site(a.r, a.c), 1, b.p, cp);
  else if (b.site) launch_site_gemm(g, cfg_site(b.r, b.c), 1, a.p, cp);
  else tgemm_launch<T, T, T, T>(stream_, g, a.p, b.p, cp);
  prof_end();
}

// rows of M (m x len, contiguous) -> mutually orthogonal; the k rows of largest norm, normalised -> V (k x len);
// optionally their norms -> S (k per walker)
inline bool svd_rows_compresses(int m, int len) {
  constexpr bool no_compress = false;
  return !no_compress && len <= 256 && m > 16;
}

template <typename T>
DTen<T> Engine<T>::svd_rows(DTen<T> &M, int m, int len, int k, double terr, int dmin, T *S, const int *mdyn, int mmul, int *kn_out,
                            int inner, const int *inner_live) {
  // mdyn (optional): the first mdyn[w] * mmul rows of M exist (the rest was never written); kn_out: rows kept per walker.
  // inner / inner_live (optional, only when svd_rows_compresses(m, len)): columns are (outer, inner) and only the first
  // inner_live[w] values of the inner index exist -- the factor kernel reads those columns only, the others are zero in
  // everything downstream.
  PG_REQUIRE(m <= 1024, 1, "bond dimension too large for select_rows_kernel");
  if constexpr (kCplx) {
    // complex: rotated rows of M = sigma_k v_k^H (the rows of V^H = `vt` of qlten::SVD); static shapes, every row exists
    PG_REQUIRE(!mdyn && !inner_live, 5, "svd_rows: the complex element type has no live extents");
    prof_begin(7, 0.0, 0.0);
    hipLaunchKernelGGL(jacobi_rows_cplx_kernel<T>, dim3(nw_), dim3(1024), 0, stream_, M.p, M.n, m, len, len, 60, sweeps_);
    PG_CHECK_HIP(hipGetLastError());
    prof_end();
    ++n_jacobi_;
    DTen<T> V = alloc_ten(k, len, 1);
    prof_begin(PROF_SELECT, 0.0, 0.0);
    hipLaunchKernelGGL(select_rows_kernel<T>, dim3(nw_), dim3(256), 0, stream_, (const T *)M.p, M.n, m, len, len, k, V.p, V.n,
                       S, (long)k, (const int *)nullptr, 1, kn_out, terr, dmin, (double *)nullptr);
    PG_CHECK_HIP(hipGetLastError());
    prof_end();
    return V;
  } else {
  // Rank compression first, as in the absorption: R with R^T R = M^T M from the Gram-free factor (the right singular
  // vectors and the singular values of R are those of M), then the Jacobi runs on the few live rows of R instead of
  // the m rows of M.  Walkers the factor declines (rank above its cap) keep their rows of M.
  constexpr int KC = sizeof(T) == 4 ? 96 : 48;
  PG_REQUIRE(!inner_live || svd_rows_compresses(m, len), 5, "svd_rows: live columns need the compressing path");
  DTen<T> Rf;
  int *ml = nullptr;
  T *src = M.p;
  long wsrc = M.n;
  int msrc = m;
  if (svd_rows_compresses(m, len)) {
    msrc = std::max(m, len);
    Rf = alloc_ten(msrc, len, 1);
    ml = (int *)arena_.alloc(sizeof(int) * nw_);
    prof_begin(PROF_CHOL, 0.0, 0.0);
    if (inner_live)   // the factor kernel writes the data columns of its (at most CH_LR_CAP) rows only: define the others
      PG_CHECK_HIP(hipMemset2DAsync(Rf.p, sizeof(T) * (size_t)Rf.n, 0, sizeof(T) * (size_t)std::min(msrc, CH_LR_CAP) * len, nw_, stream_));
    launch_gram_chol_lowrank<T, KC>(stream_, nw_, (const T *)M.p, M.n, len, mdyn, mmul, m, Rf.p, Rf.n, ml, inner_live ? inner : 1,
                                    inner_live, 4);
    hipLaunchKernelGGL(adopt_rows_flagged_kernel<T>, dim3(nw_), dim3(256), 0, stream_, (const T *)M.p, M.n, len,
                       mdyn, mmul, m, Rf.p, Rf.n, ml, inner_live ? inner : 1, inner_live);
    PG_CHECK_HIP(hipGetLastError());
    prof_end();
    src = Rf.p; wsrc = Rf.n;
  }
  const size_t need = sizeof(T) * (size_t)msrc * (len | 1);
  const int use_lds = need <= JACOBI_LDS_MAX;
  if (use_lds) allow_dynamic_lds(reinterpret_cast<const void *>(&jacobi_rows_kernel<T>), need);
  prof_begin(7, 0.0, 0.0);
  const int *rdyn = ml ? ml : mdyn;       // live rows of what the Jacobi runs on: the factor, or M itself
  const int rmul = ml ? 1 : mmul;
  launch_jacobi(src, wsrc, msrc, len, use_lds, need, rdyn, rmul);
  prof_end();
  ++n_jacobi_;
  DTen<T> V = alloc_ten(k, len, 1);
  prof_begin(PROF_SELECT, 0.0, 0.0);
  hipLaunchKernelGGL(select_rows_kernel<T>, dim3(nw_), dim3(256), 0, stream_, (const T *)src, wsrc, msrc, len, len, k, V.p, V.n,
                     S, (long)k, rdyn, rmul, kn_out, terr, dmin, (double *)nullptr);
  PG_CHECK_HIP(hipGetLastError());
  prof_end();
  if (ml) { arena_.free(ml); free_ten(Rf); }
  return V;
  }
}

// BMPS truncated to bond dimension <= kmax: Centralize(N-1) + RightCanonicalizeTruncate(i, 1, kmax, 0)
// (bmps_impl.h:1196-1200) in the Q-less form of the absorption: forward R_{i+1}^T R_{i+1} = (R_i A_i)^T (R_i A_i),
// backward M_i = R_i (A_i Y_{i+1}), rows -> Vt_i, Y_i = (A_i Y_{i+1}) Vt_i^T.
template <typename T>
typename Engine<T>::BMPSDev Engine<T>::truncate_bmps(const BMPSDev &in, int kmax) {
  const int N = (int)in.t.size();
  std::vector<DTen<T>> R(N);
  R[0] = ones3();   // (m = 1, a = 1)
  R[0].d[1] = 1;
  for (int i = 0; i + 1 < N; ++i) {
    const DTen<T> &A = in.t[i];
    const int m = R[i].d[0], a = A.d[0], p = A.d[1], b = A.d[2];
    DTen<T> P = alloc_ten(m, p, b);
    ein(ein_view<T>(R[i].p, R[i].n, "ma", {m, a}), ein_view<T>(A.p, A.n, "apb", {a, p, b}), ein_view<T>(P.p, P.n, "mpb", {m, p, b}), P.p);
    const int rows = m * p, cols = b;
    Acc *G = (Acc *)arena_.alloc(sizeof(Acc) * (size_t)cols * cols * nw_);
    {
      TGemmDesc g;          // G = P^H P
      g.I[2] = cols; g.sAi[2] = 1; g.sCi[2] = cols;
      g.K[2] = rows; g.sAk[2] = cols; g.sBk[2] = cols;
      g.J[2] = cols; g.sBj[2] = 1; g.sCj[2] = 1;
      g.wA = P.n; g.wB = P.n; g.wC = (long)cols * cols; g.nbatch = nw_;
      g.conjA = 1;
      prof_begin(PROF_GRAM, 0.0, 2.0 * nw_ * (double)cols * cols * rows);
      tgemm_launch<T, T, Acc, Acc>(stream_, g, P.p, P.p, G);
      prof_end();
    }
    R[i + 1] = alloc_ten(cols, cols, 1);
    prof_begin(PROF_CHOL, 0.0, 0.0);
    if constexpr (kCplx) {
      PG_REQUIRE(cols <= 1024, 1, "bond dimension too large for the complex Cholesky kernel");
      hipLaunchKernelGGL(chol_upper_cplx_kernel<T>, dim3(nw_), dim3(1024), 0, stream_, (c128 *)G, (long)cols * cols, cols, R[i + 1].p,
                         R[i + 1].n, (int *)nullptr);
    } else {
      const size_t smem = chol_smem_bytes(cols);
      allow_dynamic_lds(reinterpret_cast<const void *>(&chol_upper_kernel<T>), smem);
      hipLaunchKernelGGL(chol_upper_kernel<T>, dim3(nw_), dim3(256), smem, stream_, G, (long)cols * cols, cols, R[i + 1].p,
                         R[i + 1].n, (int *)nullptr, 0);
    }
    PG_CHECK_HIP(hipGetLastError());
    prof_end();
    arena_.free(G);
    free_ten(P);
  }
  BMPSDev out;
  out.t.resize(N);
  out.logscale = (double *)arena_.alloc(sizeof(double) * nw_);
  PG_CHECK_HIP(hipMemcpyAsync(out.logscale, in.logscale, sizeof(double) * nw_, hipMemcpyDeviceToDevice, stream_));
  DTen<T> Y = ones3();   // (b, y)
  int y = 1;
  for (int i = N - 1; i >= 0; --i) {
    const DTen<T> &A = in.t[i];
    const int a = A.d[0], p = A.d[1], b = A.d[2];
    DTen<T> Tt = alloc_ten(a, p, y);
    ein(ein_view<T>(A.p, A.n, "apb", {a, p, b}), ein_view<T>(Y.p, Y.n, "by", {b, y}), ein_view<T>(Tt.p, Tt.n, "apy", {a, p, y}), Tt.p);
    free_ten(Y);
    if (i == 0) {
      normalize(Tt.p, Tt.n, Tt.n, nw_, out.logscale);
      out.t[0] = Tt;
      break;
    }
    const int m = R[i].d[0];
    DTen<T> M = alloc_ten(m, p, y);
    ein(ein_view<T>(R[i].p, R[i].n, "ma", {m, a}), ein_view<T>(Tt.p, Tt.n, "apy", {a, p, y}), ein_view<T>(M.p, M.n, "mpy", {m, p, y}), M.p);
    const int k = std::min(kmax, std::min(m, p * y));
    DTen<T> V = svd_rows(M, m, p * y, k, 0.0, k, nullptr);
    free_ten(M);
    V.d[0] = k; V.d[1] = p; V.d[2] = y;
    out.t[i] = V;
    DTen<T> Yn = alloc_ten(a, k, 1);
    ein(ein_view<T>(Tt.p, Tt.n, "apy", {a, p, y}), ein_view<T>(V.p, V.n, "npy", {k, p, y}).cj(), ein_view<T>(Yn.p, Yn.n, "an", {a, k}), Yn.p);   // Y = T Vt^H
    normalize(Yn.p, Yn.n, Yn.n, nw_, out.logscale);
    free_ten(Tt);
    Y = Yn;
    y = k;
  }
  for (auto &t : R) arena_.free(t.p);
  return out;
}

template <typename T>
typename Engine<T>::BMPSDev Engine<T>::absorb_variational(int pos, int num, const BMPSDev &in) {
  const int N = mps_len(pos);
  PG_REQUIRE((int)in.t.size() == N && N > 2, 3, "MultiplyMPO (variational): MPS/MPO length mismatch");
  ArenaScope scope(arena_);
  const int ll = (pos + 3) % 4, lp = pos, lr = (pos + 1) % 4, lu = (pos + 2) % 4;   // pre, position, next, opposite legs
  struct Site { int r, c, e, p, f, u; char nm[5]; };
  std::vector<Site> S(N);
  for (int i = 0; i < N; ++i) {
    int r, c, dd[4];
    switch (pos) {
      case DOWN: r = num; c = i; break;
      case UP: r = num; c = N - 1 - i; break;
      case LEFT: r = i; c = num; break;
      default: r = N - 1 - i; c = num; break;
    }
    site_dims(r, c, dd);
    S[i] = Site{r, c, dd[ll], dd[lp], dd[lr], dd[lu], {0, 0, 0, 0, 0}};
  }
  // site tensor of site i with its legs (L, D, R, U) named by role: e = pre, p = position, f = next, `un` = opposite
  auto site_view = [&](int i, char un) {
    EinView<T> v;
    int dd[4], st[4];
    site_dims(S[i].r, S[i].c, dd);
    site_strides(S[i].r, S[i].c, st);
    v.n = 4; v.site = true; v.r = S[i].r; v.c = S[i].c; v.p = nullptr; v.w = 0;
    for (int l = 0; l < 4; ++l) { v.dim[l] = dd[l]; v.st[l] = st[l]; }
    v.nm[ll] = 'e'; v.nm[lp] = 'p'; v.nm[lr] = 'f'; v.nm[lu] = un;
    return v;
  };

  // ---- initial guess: (BMPS truncated to bond 2) x MPO, SVD-compressed (MakeVariationalInitGuess_) ----
  BMPSDev small = truncate_bmps(in, 2);
  const int save_min = chi_min_;
  const double save_err = trunc_err_;
  if (scheme_ == 2) { chi_min_ = chi_; trunc_err_ = 0.0; }   // bmps_impl.h:1012: (Dmax, Dmax, 0.0)
  BMPSDev res;
  try {
    if constexpr (kCplx) res = absorb_simple(pos, num, small);
    else res = absorb_svd(pos, num, small);
  } catch (...) {
    chi_min_ = save_min; trunc_err_ = save_err;
    throw;
  }
  chi_min_ = save_min; trunc_err_ = save_err;
  free_bmps(small);
  // Per-walker live extents of every bond (device arrays, nullptr = the static dim): il[i] / rl[i] = bond to the left of
  // site i of the absorbing BMPS / of the result.  Contractions run over the live parts only (ein(): compact legs on
  // tensors only ein() reads, masked -- zeros written -- legs where a whole-row kernel or a persistent tensor follows);
  // every truncation returns the live count of the bond it made.
  constexpr bool no_adapt_env = false;
  const bool no_adapt = no_adapt_env || kCplx;      // (the complex kernels have no live extents)
  std::vector<int *> il(in.live.begin(), in.live.end()), rl = res.live;
  il.resize(N + 1, nullptr);
  rl.resize(N + 1, nullptr);
  std::vector<int *> owned;                     // every live array made here or taken over from the initial guess
  for (int *l : rl) if (l) owned.push_back(l);
  if (no_adapt) { std::fill(il.begin(), il.end(), nullptr); std::fill(rl.begin(), rl.end(), nullptr); }
  auto new_live = [&]() -> int * {
    if (no_adapt) return nullptr;
    int *l = (int *)arena_.alloc(sizeof(int) * nw_);
    owned.push_back(l);
    return l;
  };
  res.live.clear();
  res.kmax.clear();
  std::vector<DTen<T>> &B = res.t;   // res tensors (k, u, q)

  struct Env { DTen<T> t; double *log; int *nl; };   // nl: live extent of the leg on the result's bond (k resp. q / n)
  auto new_log = [&](const double *a) {
    double *l = (double *)arena_.alloc(sizeof(double) * nw_);
    if (a) PG_CHECK_HIP(hipMemcpyAsync(l, a, sizeof(double) * nw_, hipMemcpyDeviceToDevice, stream_));
    else PG_CHECK_HIP(hipMemsetAsync(l, 0, sizeof(double) * nw_, stream_));
    return l;
  };
  auto free_env = [&](Env &e) { arena_.free(e.t.p); arena_.free(e.log); };
  std::vector<Env> lenvs, renvs;
  lenvs.push_back(Env{ones3(), new_log(nullptr), nullptr});   // (k, e, a)
  renvs.push_back(Env{ones3(), new_log(nullptr), nullptr});   // (b, f, q)

  // t1[k,u,b,f] = sum_{a,e,p} lenv[k,e,a] A_i[a,p,b] W_i[e,p,f,u]        (bmps_impl.h:896-897)
  auto half_left = [&](int i, const Env &le) {
    const DTen<T> &A = in.t[i];
    const int k = le.t.d[0], e = S[i].e, a = A.d[0], p = A.d[1], b = A.d[2], f = S[i].f, u = S[i].u;
    PG_REQUIRE(le.t.d[1] == e && le.t.d[2] == a && p == S[i].p, 3, "variational compression: bond mismatch (left)");
    DTen<T> t0 = alloc_ten(k * e, p, b);
    ein(ein_view<T>(le.t.p, le.t.n, "kea", {k, e, a}).dyn('k', le.nl).dyn('a', il[i]), ein_view<T>(A.p, A.n, "apb", {a, p, b}).dyn('b', il[i + 1]),
        ein_view<T>(t0.p, t0.n, "kepb", {k, e, p, b}), t0.p);
    DTen<T> t1 = alloc_ten(k, u, b, f);
    ein(ein_view<T>(t0.p, t0.n, "kepb", {k, e, p, b}).dyn('k', le.nl).dyn('b', il[i + 1]), site_view(i, 'u'),
        ein_view<T>(t1.p, t1.n, "kubf", {k, u, b, f}), t1.p);
    free_ten(t0);
    return t1;
  };
  // t3[v,q,b,f] = sum_{c,p,g} A_j[b,p,c] renv[c,g,q] W_j[f,p,g,v]   (legs of site j: e -> 'f', next -> 'g', opposite -> 'v')
  auto half_right = [&](int j, const Env &re) {
    const DTen<T> &A = in.t[j];
    const int q = re.t.d[2], g = S[j].f, a = A.d[0], p = A.d[1], c = A.d[2], e = S[j].e, v = S[j].u;
    PG_REQUIRE(re.t.d[0] == c && re.t.d[1] == g && p == S[j].p, 3, "variational compression: bond mismatch (right)");
    DTen<T> t2 = alloc_ten(a, p, g, q);
    ein(ein_view<T>(A.p, A.n, "bpc", {a, p, c}).dyn('b', il[j]).dyn('c', il[j + 1]), ein_view<T>(re.t.p, re.t.n, "cgq", {c, g, q}).dyn('q', re.nl),
        ein_view<T>(t2.p, t2.n, "bpgq", {a, p, g, q}), t2.p);
    EinView<T> w = site_view(j, 'v');
    w.nm[ll] = 'f'; w.nm[lr] = 'g';
    DTen<T> t3 = alloc_ten(v, q, a, e);
    ein(ein_view<T>(t2.p, t2.n, "bpgq", {a, p, g, q}).dyn('b', il[j]).dyn('q', re.nl), w, ein_view<T>(t3.p, t3.n, "vqbf", {v, q, a, e}), t3.p);
    free_ten(t2);
    return t3;
  };
  // renv'[b,f,n] = sum_{v,q} t3[v,q,b,f] Bj[n,v,q]                  (bmps_impl.h:739, :939)
  // (j = site of t3 / Bj; nl = live extent of Bj's left bond n.  b, outermost in the result, is compacted: the live part
  // of renv' is a prefix, n is written in full -- zeros beyond nl -- for the normalisation over that prefix)
  auto grow_right = [&](int j, const DTen<T> &t3, const DTen<T> &Bj, const Env &re, int *nl) {
    const int v = t3.d[0], q = t3.d[1], b = t3.d[2], f = t3.d[3], n = Bj.d[0];
    Env o{alloc_ten(b, f, n), new_log(re.log), nl};
    ein(ein_view<T>(t3.p, t3.n, "vqbf", {v, q, b, f}).dyn('q', re.nl).dyn('b', il[j]), ein_view<T>(Bj.p, Bj.n, "nvq", {n, v, q}).cj(),
        ein_view<T>(o.t.p, o.t.n, "bfn", {b, f, n}).dyn('n', nl, 1), o.t.p);       // Dag(res[j]) (:739)
    normalize(o.t.p, o.t.n, o.t.n, nw_, o.log, il[j], f * n);
    return o;
  };
  // lenv'[n,f,b] = sum_{k,u} Ut[n,k,u] t1[k,u,b,f]                  (bmps_impl.h:916-918)
  // (i = site of t1; nl = live extent of the new bond n: compacted prefix, b masked)
  auto grow_left = [&](int i, const DTen<T> &t1, const DTen<T> &Ut, int n, const Env &le, int *nl, bool uk_order = false) {
    const int k = t1.d[0], u = t1.d[1], b = t1.d[2], f = t1.d[3];
    Env o{alloc_ten(n, f, b), new_log(le.log), nl};
    EinView<T> utv = uk_order ? ein_view<T>(Ut.p, Ut.n, "nuk", {n, u, k}) : ein_view<T>(Ut.p, Ut.n, "nku", {n, k, u});
    ein(utv.dyn('n', nl).dyn('k', le.nl).cj(), ein_view<T>(t1.p, t1.n, "kubf", {k, u, b, f}),      // Dag(u) (:916-918)
        ein_view<T>(o.t.p, o.t.n, "nfb", {n, f, b}).dyn('b', il[i + 1], 1), o.t.p);
    normalize(o.t.p, o.t.n, o.t.n, nw_, o.log, nl, f * b);
    return o;
  };
  // right environments of sites N-1 .. 2 from the initial guess (GrowRightEnvironments_)
  for (int i = N - 1; i > 1; --i) {
    DTen<T> t3 = half_right(i, renvs.back());
    renvs.push_back(grow_right(i, t3, B[i], renvs.back(), rl[i]));
    free_ten(t3);
  }

  // two-site update of bond (i, i+1); left_move: only the left environment grows (the u of the SVD), else
  // B[i+1] = vt and the right environment grows.  Sout: singular values (padded to chi_), slog: scale of theta
  auto two_site = [&](int i, bool left_move, double terr, int dmin, T *Sout, double *slog) {
    DTen<T> t1 = half_left(i, lenvs.back());
    DTen<T> t3 = half_right(i + 1, renvs.back());
    const int k = t1.d[0], u = t1.d[1], b = t1.d[2], f = t1.d[3], v = t3.d[0], q = t3.d[1];
    PG_REQUIRE(t3.d[2] == b && t3.d[3] == f, 3, "variational compression: bond mismatch (two-site)");
    const int rows = left_move ? v * q : k * u, len = left_move ? k * u : v * q;
    const int kn = std::min(chi_, std::min(rows, len));
    DTen<T> th = alloc_ten(rows, len, 1);
    const Env &le = lenvs.back(), &re = renvs.back();
    // rows of theta: the live bond outermost (a prefix of rows exists); columns: the live bond innermost, left out when the
    // compressing factor kernel reads theta (it takes the live columns only), else written in full (zeros)
    // (leaving the dead columns out -- inner_live of svd_rows -- was measured: the GEMM saves what the extra zero fill of
    // the factor costs, 3.5 k vs 3.6 k amplitudes/s at C4: the masked form is kept)
    const int cmask = 1;
    if (left_move)
      ein(ein_view<T>(t3.p, t3.n, "vqbf", {v, q, b, f}).dyn('q', re.nl).dyn('b', il[i + 1]), ein_view<T>(t1.p, t1.n, "kubf", {k, u, b, f}),
          ein_view<T>(th.p, th.n, "qvuk", {q, v, u, k}).dyn('k', le.nl, cmask), th.p);
    else
      ein(ein_view<T>(t1.p, t1.n, "kubf", {k, u, b, f}).dyn('k', le.nl).dyn('b', il[i + 1]), ein_view<T>(t3.p, t3.n, "vqbf", {v, q, b, f}),
          ein_view<T>(th.p, th.n, "kuvq", {k, u, v, q}).dyn('q', re.nl, cmask), th.p);
    if (Sout) PG_CHECK_HIP(hipMemsetAsync(Sout, 0, sizeof(T) * (size_t)chi_ * nw_, stream_));
    int *nl = new_live();
    const int *cl = left_move ? le.nl : re.nl;     // live extent of the inner column index
    DTen<T> V = svd_rows(th, rows, len, kn, terr, std::min(dmin, kn), Sout, left_move ? re.nl : le.nl, left_move ? v : u, nl,
                         left_move ? k : q, cmask ? nullptr : cl);
    free_ten(th);
    if (slog) {
      PG_CHECK_HIP(hipMemsetAsync(slog, 0, sizeof(double) * nw_, stream_));
      add_logs(slog, lenvs.back().log, renvs.back().log, nullptr, nullptr);
    }
    if (left_move) {
      Env ne = grow_left(i, t1, V, kn, lenvs.back(), nl, true);
      lenvs.push_back(ne);
      free_env(renvs.back());
      renvs.pop_back();
      free_ten(V);
    } else {
      V.d[0] = kn; V.d[1] = v; V.d[2] = q;
      arena_.free(B[i + 1].p);
      B[i + 1] = V;
      rl[i + 1] = nl;
      Env ne = grow_right(i + 1, t3, V, renvs.back(), nl);
      renvs.push_back(ne);
      free_env(lenvs.back());
      lenvs.pop_back();
    }
    free_ten(t1);
    free_ten(t3);
    return kn;
  };
  // the final two-site step at bond (0, 1): B[1] = vt, renv grows, B[0] = t1 . renv = u s   (bmps_impl.h:960-988)
  auto close_at_zero = [&](double terr, int dmin, bool keep_env) {
    DTen<T> t1 = half_left(0, lenvs.back());
    DTen<T> t3 = half_right(1, renvs.back());
    const int k = t1.d[0], u = t1.d[1], b = t1.d[2], f = t1.d[3], v = t3.d[0], q = t3.d[1];
    const int rows = k * u, len = v * q, kn = std::min(chi_, std::min(rows, len));
    DTen<T> th = alloc_ten(rows, len, 1);
    const Env &le = lenvs.back(), &re = renvs.back();
    ein(ein_view<T>(t1.p, t1.n, "kubf", {k, u, b, f}).dyn('k', le.nl).dyn('b', il[1]), ein_view<T>(t3.p, t3.n, "vqbf", {v, q, b, f}),
        ein_view<T>(th.p, th.n, "kuvq", {k, u, v, q}).dyn('q', re.nl, 1), th.p);
    int *nl = new_live();
    DTen<T> V = svd_rows(th, rows, len, kn, terr, std::min(dmin, kn), nullptr, le.nl, u, nl);
    free_ten(th);
    V.d[0] = kn; V.d[1] = v; V.d[2] = q;
    arena_.free(B[1].p);
    B[1] = V;
    rl[1] = nl;
    Env ne = grow_right(1, t3, V, renvs.back(), nl);
    free_ten(t3);
    DTen<T> B0 = alloc_ten(k, u, kn);
    ein(ein_view<T>(t1.p, t1.n, "kubf", {k, u, b, f}).dyn('k', le.nl).dyn('b', il[1]), ein_view<T>(ne.t.p, ne.t.n, "bfn", {b, f, kn}),
        ein_view<T>(B0.p, B0.n, "kun", {k, u, kn}).dyn('n', nl, 1), B0.p);
    free_ten(t1);
    arena_.free(B[0].p);
    B[0] = B0;
    // scale of the state: |B0| exp(log renv) (the left boundary is 1)
    PG_CHECK_HIP(hipMemcpyAsync(res.logscale, in.logscale, sizeof(double) * nw_, hipMemcpyDeviceToDevice, stream_));
    add_log(res.logscale, ne.log);
    if (keep_env) renvs.push_back(ne);
    else free_env(ne);
  };

  std::vector<double> h_log(nw_), h_log_last(nw_);
  if (scheme_ == 1) {
    // ---------------- two-site sweeps (bmps_impl.h:888-959) ----------------
    T *Sd = (T *)arena_.alloc(sizeof(T) * (size_t)chi_ * nw_);
    double *slog = (double *)arena_.alloc(sizeof(double) * nw_);
    std::vector<T> h_s((size_t)chi_ * nw_), h_last;
    int ks_last = -1;
    for (int it = 0; it < iter_max_; ++it) {
      for (int i = 0; i < N - 2; ++i) two_site(i, true, trunc_err_, chi_min_, nullptr, nullptr);
      int ks = 0;
      for (int i = N - 2; i > 0; --i) ks = two_site(i, false, trunc_err_, chi_min_, i == 1 ? Sd : nullptr, i == 1 ? slog : nullptr);
      ++n_var_iters_;
      PG_CHECK_HIP(hipMemcpyAsync(h_s.data(), Sd, sizeof(T) * (size_t)ks * nw_, hipMemcpyDeviceToHost, stream_));
      PG_CHECK_HIP(hipMemcpyAsync(h_log.data(), slog, sizeof(double) * nw_, hipMemcpyDeviceToHost, stream_));
      PG_CHECK_HIP(hipStreamSynchronize(stream_));
      bool all = it > 0 && ks == ks_last;
      if (all) {
        for (int w = 0; w < nw_ && all; ++w) {
          const T *s = &h_s[(size_t)w * ks], *sl = &h_last[(size_t)w * ks];
          auto sv = [](const T &x) -> double { if constexpr (kCplx) return (double)x.re; else return (double)x; };   // singular values are real
          int n = 0, nl = 0;
          for (int x = 0; x < ks; ++x) { n += sv(s[x]) != 0.0; nl += sv(sl[x]) != 0.0; }
          if (n != nl || !(sv(s[0]) > 0.0)) { all = false; break; }   // bond dimension changed (bmps_impl.h:946)
          const double rel = std::exp(h_log_last[w] - h_log[w]);
          double diff = 0.0;
          for (int x = 0; x < ks; ++x) diff += std::fabs(sv(s[x]) - sv(sl[x]) * rel);
          if (!(diff / sv(s[0]) < conv_tol_)) all = false;
        }
      }
      if (all) break;
      h_last = h_s; h_log_last = h_log; ks_last = ks;
    }
    arena_.free(Sd);
    arena_.free(slog);
    close_at_zero(trunc_err_, chi_min_, false);
  } else {
    // ---------------- one-site scheme (bmps_impl.h:1021-1166) ----------------
    for (int i = 0; i < N - 2; ++i) two_site(i, true, trunc_err_, chi_, nullptr, nullptr);      // SVD(.., Dmax, Dmax)
    for (int i = N - 2; i > 0; --i) two_site(i, false, trunc_err_, chi_, nullptr, nullptr);
    close_at_zero(trunc_err_, chi_min_, true);                                                     // renv of sites >= 1 kept
    double *rlog = (double *)arena_.alloc(sizeof(double) * nw_);
    for (int it = 0; it < iter_max_; ++it) {
      for (int i = 0; i + 1 < N; ++i) {          // right-moving: Q of t2 = t1 . renv
        DTen<T> t1 = half_left(i, lenvs.back());
        const int k = t1.d[0], u = t1.d[1], b = t1.d[2], f = t1.d[3], q = renvs.back().t.d[2];
        DTen<T> t2 = alloc_ten(q, k, u);
        const Env &le = lenvs.back(), &re = renvs.back();
        ein(ein_view<T>(re.t.p, re.t.n, "bfn", {b, f, q}).dyn('n', re.nl).dyn('b', il[i + 1]), ein_view<T>(t1.p, t1.n, "kubf", {k, u, b, f}),
            ein_view<T>(t2.p, t2.n, "nku", {q, k, u}).dyn('k', le.nl, 1), t2.p);
        const int kn = std::min(q, k * u);
        int *nl = new_live();
        DTen<T> Qt = svd_rows(t2, q, k * u, kn, 0.0, kn, nullptr, re.nl, 1, nl);
        free_ten(t2);
        Env ne = grow_left(i, t1, Qt, kn, lenvs.back(), nl);
        lenvs.push_back(ne);
        free_env(renvs.back());
        renvs.pop_back();
        free_ten(Qt);
        free_ten(t1);
      }
      for (int i = N - 1; i > 0; --i) {          // left-moving: Q of t2 = lenv . t3
        DTen<T> t3 = half_right(i, renvs.back());
        const int v = t3.d[0], q = t3.d[1], b = t3.d[2], f = t3.d[3], k = lenvs.back().t.d[0];
        DTen<T> t2 = alloc_ten(k, v, q);
        const Env &le = lenvs.back(), &re = renvs.back();
        ein(ein_view<T>(le.t.p, le.t.n, "kfb", {k, f, b}).dyn('k', le.nl).dyn('b', il[i]), ein_view<T>(t3.p, t3.n, "vqbf", {v, q, b, f}),
            ein_view<T>(t2.p, t2.n, "kvq", {k, v, q}).dyn('q', re.nl, 1), t2.p);
        if (i == 1) {   // |r| of the last QR = |t2| exp(log lenv + log renv)   (bmps_impl.h:1149)
          PG_CHECK_HIP(hipMemsetAsync(rlog, 0, sizeof(double) * nw_, stream_));
          add_logs(rlog, lenvs.back().log, renvs.back().log, nullptr, nullptr);
          normalize(t2.p, t2.n, t2.n, nw_, rlog, le.nl, v * q);
        }
        const int kn = std::min(k, v * q);
        int *nl = new_live();
        DTen<T> Qt = svd_rows(t2, k, v * q, kn, 0.0, kn, nullptr, le.nl, 1, nl);
        free_ten(t2);
        Qt.d[0] = kn; Qt.d[1] = v; Qt.d[2] = q;
        arena_.free(B[i].p);
        B[i] = Qt;
        rl[i] = nl;
        Env ne = grow_right(i, t3, Qt, renvs.back(), nl);
        renvs.push_back(ne);
        free_env(lenvs.back());
        lenvs.pop_back();
        free_ten(t3);
      }
      ++n_var_iters_;
      PG_CHECK_HIP(hipMemcpyAsync(h_log.data(), rlog, sizeof(double) * nw_, hipMemcpyDeviceToHost, stream_));
      PG_CHECK_HIP(hipStreamSynchronize(stream_));
      bool all = it > 0;
      for (int w = 0; w < nw_ && all; ++w)
        if (std::fabs(1.0 - std::exp(h_log_last[w] - h_log[w])) > conv_tol_) all = false;   // |r - r_last| / |r|
      if (all) break;
      h_log_last = h_log;
    }
    arena_.free(rlog);
    // res[0] = t1 . renv   (bmps_impl.h:1158-1164)
    DTen<T> t1 = half_left(0, lenvs.back());
    const int k = t1.d[0], u = t1.d[1], b = t1.d[2], f = t1.d[3], q = renvs.back().t.d[2];
    DTen<T> B0 = alloc_ten(k, u, q);
    ein(ein_view<T>(t1.p, t1.n, "kubf", {k, u, b, f}).dyn('k', lenvs.back().nl).dyn('b', il[1]),
        ein_view<T>(renvs.back().t.p, renvs.back().t.n, "bfn", {b, f, q}), ein_view<T>(B0.p, B0.n, "kun", {k, u, q}).dyn('n', renvs.back().nl, 1), B0.p);
    free_ten(t1);
    arena_.free(B[0].p);
    B[0] = B0;
    PG_CHECK_HIP(hipMemcpyAsync(res.logscale, in.logscale, sizeof(double) * nw_, hipMemcpyDeviceToDevice, stream_));
    add_log(res.logscale, renvs.back().log);
  }
  normalize(B[0].p, B[0].n, B[0].n, nw_, res.logscale);
  for (auto &e : lenvs) free_env(e);
  for (auto &e : renvs) free_env(e);
  // the live extents of the result's bonds go with it (the next absorption contracts over them); the others are returned
  res.live = rl;
  res.live.resize(N + 1, nullptr);
  for (int *l : owned)
    if (std::find(res.live.begin(), res.live.end(), l) == res.live.end()) arena_.free(l);
  ++n_absorb_;
  return res;
}

}  // namespace pepsgpu
