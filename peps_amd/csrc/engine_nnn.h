// Two-row (rank-4) environments and the next-nearest / third-neighbour / sqrt(5) replacement traces
// of BMPSContractor (bmps_contractor_init.h:130-186, bmps_contractor_grow.h:375-527,
// bmps_contractor_helpers.h:12-180, bmps_contractor_trace.h:207-536; bosonic branches), batched over
// walkers x candidates.  Everything is one primitive -- the BTen2 growth step of
// GrowBTen2StepAfterTransposedMPOTens (helpers.h:174-177) with two site selectors -- plus a
// 4-index dot; the Transpose calls of the reference are strides of the tensor GEMM.
#pragma once
#include "engine.h"

namespace pepsgpu {

enum { LEFTUP_TO_RIGHTDOWN = 0, LEFTDOWN_TO_RIGHTUP = 1 };   // basic.h:89-92 DIAGONAL_DIR

// out[x, o1, o2, y] from bt[c, b1, b2, b3], mps1[x, p1, c], site1, site2, mps2[b3, n2, y]:
//   tmp1[x,p1,b1,b2,b3] = sum_c      mps1 . bt
//   tmp2[b2,b3,x,o1,n1] = sum_{p1,b1} tmp1 . site1[(post+3)<-p1, post<-b1 | (post+2)->o1, (post+1)->n1]
//   tmp3[b3,x,o1,n2,o2] = sum_{n1,b2} tmp2 . site2[(post+3)<-n1, post<-b2 | (post+1)->n2, (post+2)->o2]
//   out                 = sum_{b3,n2} tmp3 . mps2
template <typename T>
typename Engine<T>::BTenDev Engine<T>::bten2_step(int post, const BTenDev &bt, const DTen<T> &mps1, const SiteSel &s1,
                                                  const SiteSel &s2, const DTen<T> &mps2, int ncand, int bt_ncand,
                                                  bool normalise) {
  ArenaScope scope(arena_);
  const int nb = nw_ * ncand, nb1 = nw_ * bt_ncand;
  PG_REQUIRE(ncand % bt_ncand == 0 && (!normalise || ncand == 1), 1, "BTen2 step: bad candidate batching");
  int d1[4], st1[4], d2[4], st2[4];
  site_dims(s1.r, s1.c, d1); site_strides(s1.r, s1.c, st1);
  site_dims(s2.r, s2.c, d2); site_strides(s2.r, s2.c, st2);
  const int lc = (post + 3) % 4, lb = post, ln = (post + 1) % 4, lo = (post + 2) % 4;
  const int x = mps1.d[0], p1 = mps1.d[1], cdim = mps1.d[2];
  const int b1 = bt.t.d[1], b2 = bt.t.d[2], b3 = bt.t.d[3];
  const int o1 = d1[lo], n1 = d1[ln], n2 = d2[ln], o2 = d2[lo], y = mps2.d[2];
  PG_REQUIRE(cdim == bt.t.d[0] && p1 == d1[lc] && b1 == d1[lb] && n1 == d2[lc] && b2 == d2[lb] && mps2.d[0] == b3 &&
                 mps2.d[1] == n2, 3, "BTen2 step: bond dimension mismatch between environment tensors");
  DTen<T> tmp1 = alloc_ten(x * p1, b1, b2, b3, nb1);
  {
    TGemmDesc g;
    g.I[2] = x * p1; g.sAi[2] = cdim; g.sCi[2] = b1 * b2 * b3;
    g.K[2] = cdim; g.sAk[2] = 1; g.sBk[2] = b1 * b2 * b3;
    g.J[2] = b1 * b2 * b3; g.sBj[2] = 1; g.sCj[2] = 1;
    g.wA = mps1.n; g.bdivA = bt_ncand; g.wB = bt.t.n; g.wC = tmp1.n; g.nbatch = nb1;
    tgemm_launch<T, T, T, T>(stream_, g, mps1.p, bt.t.p, tmp1.p);
  }
  // tmp1 layout [x][p1][b1][b2][b3]
  DTen<T> tmp2 = alloc_ten(b2 * b3, x, o1, n1, nb);
  {
    TGemmDesc g;
    g.I[0] = b2; g.I[1] = b3; g.I[2] = x;
    g.sAi[0] = b3; g.sAi[1] = 1; g.sAi[2] = p1 * b1 * b2 * b3;
    g.sCi[0] = b3 * x * o1 * n1; g.sCi[1] = x * o1 * n1; g.sCi[2] = o1 * n1;
    g.K[1] = p1; g.K[2] = b1; g.sAk[1] = b1 * b2 * b3; g.sAk[2] = b2 * b3; g.sBk[1] = st1[lc]; g.sBk[2] = st1[lb];
    g.J[1] = o1; g.J[2] = n1; g.sBj[1] = st1[lo]; g.sBj[2] = st1[ln]; g.sCj[1] = n1; g.sCj[2] = 1;
    g.wA = tmp1.n; g.bdivA = ncand / bt_ncand; g.wC = tmp2.n; g.nbatch = nb;
    launch_site_gemm(g, s1, ncand, tmp1.p, tmp2.p);
  }
  // tmp2 layout [b2][b3][x][o1][n1]
  DTen<T> tmp3 = alloc_ten(b3 * x, o1, n2, o2, nb);
  {
    TGemmDesc g;
    g.I[0] = b3; g.I[1] = x; g.I[2] = o1;
    g.sAi[0] = x * o1 * n1; g.sAi[1] = o1 * n1; g.sAi[2] = n1;
    g.sCi[0] = x * o1 * n2 * o2; g.sCi[1] = o1 * n2 * o2; g.sCi[2] = n2 * o2;
    g.K[1] = n1; g.K[2] = b2; g.sAk[1] = 1; g.sAk[2] = b3 * x * o1 * n1; g.sBk[1] = st2[lc]; g.sBk[2] = st2[lb];
    g.J[1] = n2; g.J[2] = o2; g.sBj[1] = st2[ln]; g.sBj[2] = st2[lo]; g.sCj[1] = o2; g.sCj[2] = 1;
    g.wA = tmp2.n; g.wC = tmp3.n; g.nbatch = nb;
    launch_site_gemm(g, s2, ncand, tmp2.p, tmp3.p);
  }
  // tmp3 layout [b3][x][o1][n2][o2]
  BTenDev o;
  o.t = alloc_ten(x, o1, o2, y, nb);
  {
    TGemmDesc g;
    g.I[0] = x; g.I[1] = o1; g.I[2] = o2;
    g.sAi[0] = o1 * n2 * o2; g.sAi[1] = n2 * o2; g.sAi[2] = 1;
    g.sCi[0] = o1 * o2 * y; g.sCi[1] = o2 * y; g.sCi[2] = y;
    g.K[1] = b3; g.K[2] = n2; g.sAk[1] = x * o1 * n2 * o2; g.sAk[2] = o2; g.sBk[1] = n2 * y; g.sBk[2] = y;
    g.J[2] = y; g.sBj[2] = 1; g.sCj[2] = 1;
    g.wA = tmp3.n; g.wB = mps2.n; g.bdivB = ncand; g.wC = o.t.n; g.nbatch = nb;
    tgemm_launch<T, T, T, T>(stream_, g, tmp3.p, mps2.p, o.t.p);
  }
  free_ten(tmp1); free_ten(tmp2); free_ten(tmp3);
  o.logscale = nullptr;
  if (normalise) {
    o.logscale = (double *)arena_.alloc(sizeof(double) * nw_);
    PG_CHECK_HIP(hipMemcpyAsync(o.logscale, bt.logscale, sizeof(double) * nw_, hipMemcpyDeviceToDevice, stream_));
    normalize(o.t.p, o.t.n, o.t.n, nw_, o.logscale);
  }
  return o;
}

// out[(w,cand)] = sum a[i,j,k,l] b[l,k,j,i] * exp(lsum[w])     (trace.h:280, :321, :534)
template <typename T>
void Engine<T>::finish_dot4(const DTen<T> &a, const DTen<T> &b, int nc, double *lsum, double *out) {
  PG_REQUIRE(a.d[0] == b.d[3] && a.d[1] == b.d[2] && a.d[2] == b.d[1] && a.d[3] == b.d[0], 3,
             "trace: two-row environment bond mismatch");
  const int nb = nw_ * nc;
  Acc *res = (Acc *)arena_.alloc(sizeof(Acc) * nb);
  TGemmDesc g;
  // The tensor GEMM addresses three K sub-indices; the fourth (j, a site bond of size <= D) is a
  // short loop of accumulating launches.
  const int I = a.d[0], J = a.d[1], Kd = a.d[2], L = a.d[3];
  g.K[0] = I; g.K[1] = Kd; g.K[2] = L;
  g.sAk[0] = J * Kd * L; g.sAk[1] = L; g.sAk[2] = 1;
  g.sBk[0] = 1; g.sBk[1] = J * I; g.sBk[2] = Kd * J * I;
  g.wA = a.n; g.wB = b.n; g.wC = 1; g.nbatch = nb;
  for (int j = 0; j < J; ++j) {
    g.accumulate = j > 0;
    tgemm_launch<T, T, Acc, Acc>(stream_, g, a.p + (long)j * Kd * L, b.p + (long)j * I, res);
  }
  std::vector<Acc> h(nb);
  std::vector<double> hl(nw_);
  PG_CHECK_HIP(hipMemcpyAsync(h.data(), res, nb * sizeof(Acc), hipMemcpyDeviceToHost, stream_));
  PG_CHECK_HIP(hipMemcpyAsync(hl.data(), lsum, nw_ * sizeof(double), hipMemcpyDeviceToHost, stream_));
  PG_CHECK_HIP(hipStreamSynchronize(stream_));
  for (int i = 0; i < nb; ++i) {
    const double sc = std::exp(hl[i / nc]);
    if constexpr (kCplx) { out[2 * i] = h[i].re * sc; out[2 * i + 1] = h[i].im * sc; }
    else out[i] = h[i] * sc;
  }
  arena_.free(res);
}

template <typename T>
int *Engine<T>::upload_cand(int ncand, int ncols, const int32_t *cand) {
  if (ncand <= 0) return nullptr;
  const size_t cnt = (size_t)nw_ * ncand * ncols;
  for (size_t i = 0; i < cnt; ++i) PG_REQUIRE(cand[i] >= 0 && cand[i] < dp_, 4, "candidate state out of range");
  int *d = (int *)arena_.alloc(cnt * sizeof(int));
  PG_CHECK_HIP(hipMemcpyAsync(d, cand, cnt * sizeof(int), hipMemcpyHostToDevice, stream_));
  return d;
}

template <typename T>
void Engine<T>::init_bten2(int pos, int slice) {   // init.h:130-186
  require_ready();
  (void)slice;
  clear_bten2(pos, 0);
  BTenDev b;
  b.t = alloc_ten(1, 1, 1, 1);
  hipLaunchKernelGGL(fill_kernel<T>, dim3((nw_ + 255) / 256), dim3(256), 0, stream_, b.t.p, (long)nw_, T(1));
  b.logscale = zeros_f64();
  bten2_[pos].push_back(b);
}

template <typename T>
void Engine<T>::grow_full_bten2(int pos, int slice, int remain, int init) {   // grow.h:375-470
  require_ready();
  if (init) init_bten2(pos, slice);
  PG_REQUIRE(bten2_size(pos) > 0, 3, "GrowFullBTen2: BTen2 not initialised");
  const int n = (pos == DOWN || pos == UP) ? Ly_ : Lx_;
  const int pre = (pos + 3) % 4, nxt = (pos + 1) % 4;
  int s_pre, s_nxt;   // slices of the two BMPS
  switch (pos) {
    case DOWN: s_pre = slice; s_nxt = slice + 1; break;        // LEFT(col1), RIGHT(col2)
    case RIGHT: s_pre = slice + 1; s_nxt = slice; break;       // DOWN(row2), UP(row1)
    case UP: s_pre = slice + 1; s_nxt = slice; break;          // RIGHT(col2), LEFT(col1)
    default: s_pre = slice; s_nxt = slice + 1; break;          // UP(row1), DOWN(row2)
  }
  const BMPSDev &b1 = bmps_at_slice(pre, s_pre);
  const BMPSDev &b2 = bmps_at_slice(nxt, s_nxt);
  for (int i = bten2_size(pos) - 1; i < n - remain; ++i) {
    SitePick p1, p2;
    switch (pos) {
      case DOWN: p1 = {n - 1 - i, slice, -1}; p2 = {n - 1 - i, slice + 1, -1}; break;
      case RIGHT: p1 = {slice + 1, n - 1 - i, -1}; p2 = {slice, n - 1 - i, -1}; break;
      case UP: p1 = {i, slice + 1, -1}; p2 = {i, slice, -1}; break;
      default: p1 = {slice, i, -1}; p2 = {slice + 1, i, -1}; break;
    }
    BTenDev nb = bten2_step(pos, bten2_[pos].back(), b1.t[n - i - 1], pick(p1, nullptr, 0), pick(p2, nullptr, 0), b2.t[i],
                            1, 1, true);
    bten2_[pos].push_back(nb);
  }
}

template <typename T>
void Engine<T>::grow_bten2_step(int pos, int slice) {   // grow.h:472-515, helpers.h:41-92
  require_ready();
  const int pre = (pos + 3) % 4, nxt = (pos + 1) % 4;
  const int bs = bten2_size(pos);
  PG_REQUIRE(bs > 0, 3, "GrowBTen2Step: BTen2 not initialised");
  int n, i1, i2;
  SitePick p1, p2;
  switch (pos) {
    case DOWN: n = Ly_; p1 = {n - bs, slice, -1}; p2 = {n - bs, slice + 1, -1}; i1 = slice; i2 = Lx_ - 1 - (slice + 1); break;
    case UP: n = Ly_; p1 = {bs - 1, slice + 1, -1}; p2 = {bs - 1, slice, -1}; i1 = Lx_ - 1 - (slice + 1); i2 = slice; break;
    case LEFT: n = Lx_; p1 = {slice, bs - 1, -1}; p2 = {slice + 1, bs - 1, -1}; i1 = slice; i2 = Ly_ - 1 - (slice + 1); break;
    default: n = Lx_; p1 = {slice + 1, n - bs, -1}; p2 = {slice, n - bs, -1}; i1 = Ly_ - 1 - (slice + 1); i2 = slice; break;
  }
  PG_REQUIRE(bs <= n && i1 >= 0 && i1 < bmps_size(pre) && i2 >= 0 && i2 < bmps_size(nxt), 3,
             "GrowBTen2Step: BMPS environment missing");
  BTenDev nb = bten2_step(pos, bten2_[pos].back(), bmps_[pre][i1].t[n - bs], pick(p1, nullptr, 0), pick(p2, nullptr, 0),
                          bmps_[nxt][i2].t[bs - 1], 1, 1, true);
  bten2_[pos].push_back(nb);
}

template <typename T>
void Engine<T>::shift_bten2_window(int pos, int slice) {   // grow.h:523-527
  PG_REQUIRE(bten2_size(pos) > 0, 3, "ShiftBTen2Window: BTen2 empty");
  clear_bten2(pos, bten2_size(pos) - 1);
  grow_bten2_step((pos + 2) % 4, slice);
}

// ReplaceNNNSiteTrace (trace.h:207-324).  cand[w][k][2] = states of (ten_left, ten_right).
template <typename T>
void Engine<T>::replace_nnn_trace(int row1, int col1, int dir, int orient, int ncand, const int32_t *cand, double *out) {
  require_ready();
  ArenaScope scope(arena_);
  const int row2 = row1 + 1, col2 = col1 + 1;
  PG_REQUIRE(row1 >= 0 && col1 >= 0 && row2 < Ly_ && col2 < Lx_, 1, "ReplaceNNNSiteTrace: plaquette outside the lattice");
  const int nc = ncand > 0 ? ncand : 1;
  int *dc = upload_cand(ncand, 2, cand);
  const int cl = ncand > 0 ? 0 : -1, cr = ncand > 0 ? 1 : -1;
  double *lsum = zeros_f64();
  BTenDev a, b;
  if (orient == HORIZONTAL) {
    const BMPSDev &up = bmps_at_slice(UP, row1), &dn = bmps_at_slice(DOWN, row2);
    PG_REQUIRE(bten2_size(LEFT) > col1, 3, "ReplaceNNNSiteTrace: LEFT BTen2 missing");
    const BTenDev &lb = bten2_[LEFT][col1], &rb = bten2_at_slice(RIGHT, col2);
    SitePick t0, t1, t2, t3;   // (row1,col1), (row2,col1), (row2,col2), (row1,col2)
    if (dir == LEFTUP_TO_RIGHTDOWN) { t0 = {row1, col1, cl}; t1 = {row2, col1, -1}; t2 = {row2, col2, cr}; t3 = {row1, col2, -1}; }
    else { t0 = {row1, col1, -1}; t1 = {row2, col1, cl}; t2 = {row2, col2, -1}; t3 = {row1, col2, cr}; }
    a = bten2_step(LEFT, lb, at_logical(up, UP, col1), pick(t0, dc, 2), pick(t1, dc, 2), at_logical(dn, DOWN, col1), nc, 1, false);
    b = bten2_step(RIGHT, rb, at_logical(dn, DOWN, col2), pick(t2, dc, 2), pick(t3, dc, 2), at_logical(up, UP, col2), nc, 1, false);
    add_logs(lsum, up.logscale, dn.logscale, lb.logscale, rb.logscale);
  } else {
    const BMPSDev &lf = bmps_at_slice(LEFT, col1), &rt = bmps_at_slice(RIGHT, col2);
    PG_REQUIRE(bten2_size(UP) > row1, 3, "ReplaceNNNSiteTrace: UP BTen2 missing");
    const BTenDev &tb = bten2_[UP][row1], &bb = bten2_at_slice(DOWN, row2);
    SitePick m0, m1, m2, m3;   // (row2,col1), (row2,col2), (row1,col1), (row1,col2)
    if (dir == LEFTUP_TO_RIGHTDOWN) { m0 = {row2, col1, -1}; m1 = {row2, col2, cr}; m2 = {row1, col1, cl}; m3 = {row1, col2, -1}; }
    else { m0 = {row2, col1, cl}; m1 = {row2, col2, -1}; m2 = {row1, col1, -1}; m3 = {row1, col2, cr}; }
    a = bten2_step(DOWN, bb, at_logical(lf, LEFT, row2), pick(m0, dc, 2), pick(m1, dc, 2), at_logical(rt, RIGHT, row2), nc, 1, false);
    b = bten2_step(UP, tb, at_logical(rt, RIGHT, row1), pick(m3, dc, 2), pick(m2, dc, 2), at_logical(lf, LEFT, row1), nc, 1, false);
    add_logs(lsum, lf.logscale, rt.logscale, tb.logscale, bb.logscale);
  }
  finish_dot4(a.t, b.t, nc, lsum, out);
  free_ten(a.t); free_ten(b.t);
  arena_.free(lsum);
  if (dc) arena_.free(dc);
}

// ---- fermionic diagonal hop against twisted environments (round 5; see the declaration of bten2_inactive_ in engine.h) ----
template <typename T>
void Engine<T>::bten2_select_set(int set) {
  PG_REQUIRE(set == 0 || set == 1, 1, "BTen2 set must be 0 or 1");
  if (set == bten2_active_) return;
  for (int p = 0; p < 4; ++p) std::swap(bten2_[p], bten2_inactive_[p]);
  bten2_active_ = set;
}

// states = [walker][N] extended states of row (HORIZONTAL) / column (VERTICAL) `num`; nullptr clears the override.  While it is
// set, every kernel that selects a site tensor of that slice by configuration reads this table (the walkers' own table elsewhere).
template <typename T>
void Engine<T>::cfg_override_slice(int orient, int num, const int32_t *states) {
  require_ready();
  if (!states) {
    if (cfg_ovr_tab_) { arena_.free(cfg_ovr_tab_); cfg_ovr_tab_ = nullptr; }
    ovr_on_ = false; ovr_cfg_ = nullptr;
    return;
  }
  PG_REQUIRE(orient == HORIZONTAL || orient == VERTICAL, 1, "bad orientation");
  const bool hor = orient == HORIZONTAL;
  const int N = hor ? Lx_ : Ly_, lim = hor ? Ly_ : Lx_;
  PG_REQUIRE(num >= 0 && num < lim, 1, "configuration override: slice outside the lattice");
  PG_REQUIRE(!ovr_on_ || cfg_ovr_tab_, 3, "configuration override: a BMPSWalker operation is in progress");
  std::vector<int> tab(hcfg_);
  for (int w = 0; w < nw_; ++w)
    for (int j = 0; j < N; ++j) {
      const int st = states[(size_t)w * N + j];
      PG_REQUIRE(st >= 0 && st < dp_, 4, "configuration override: state exceeds the physical dimension");
      tab[(size_t)w * Ly_ * Lx_ + (hor ? num * Lx_ + j : j * Lx_ + num)] = st;
    }
  if (!cfg_ovr_tab_) cfg_ovr_tab_ = (int *)arena_.alloc(sizeof(int) * tab.size());
  PG_CHECK_HIP(hipMemcpyAsync(cfg_ovr_tab_, tab.data(), sizeof(int) * tab.size(), hipMemcpyHostToDevice, stream_));
  PG_CHECK_HIP(hipStreamSynchronize(stream_));
  ovr_hor_ = hor; ovr_num_ = num; ovr_cfg_ = cfg_ovr_tab_; ovr_tens_ = nullptr; ovr_nt_ = 0; ovr_on_ = true;
}

// The plaquette (row1, col1) .. (row1 + 1, col1 + 1) closed with FOUR replaced tensors between the LEFT BTen2 of set `left_set` and
// the RIGHT BTen2 of set `right_set` (row BMPS: HORIZONTAL form of ReplaceNNNSiteTrace, trace.h:207-324).
// cand[w][k][4] = states of (row1, col1), (row2, col1), (row2, col2), (row1, col2); ncand = 0: the walkers' own states.
template <typename T>
void Engine<T>::replace_plaquette_trace(int row1, int col1, int ncand, const int32_t *cand, int left_set, int right_set, double *out) {
  require_ready();
  ArenaScope scope(arena_);
  const int row2 = row1 + 1, col2 = col1 + 1;
  PG_REQUIRE(row1 >= 0 && col1 >= 0 && row2 < Ly_ && col2 < Lx_, 1, "plaquette outside the lattice");
  PG_REQUIRE((left_set == 0 || left_set == 1) && (right_set == 0 || right_set == 1), 1, "BTen2 set must be 0 or 1");
  const int nc = ncand > 0 ? ncand : 1;
  int *dc = upload_cand(ncand, 4, cand);
  auto col_of = [&](int k) { return ncand > 0 ? k : -1; };
  double *lsum = zeros_f64();
  const std::vector<BTenDev> &ls = (left_set == bten2_active_ ? bten2_ : bten2_inactive_)[LEFT];
  const std::vector<BTenDev> &rs = (right_set == bten2_active_ ? bten2_ : bten2_inactive_)[RIGHT];
  const int kr = Lx_ - 1 - col2;
  PG_REQUIRE((int)ls.size() > col1, 3, "plaquette trace: LEFT BTen2 of that set missing");
  PG_REQUIRE(kr >= 0 && kr < (int)rs.size(), 3, "plaquette trace: RIGHT BTen2 of that set missing");
  const BMPSDev &up = bmps_at_slice(UP, row1), &dn = bmps_at_slice(DOWN, row2);
  const BTenDev &lb = ls[col1], &rb = rs[kr];
  const SitePick t0{row1, col1, col_of(0)}, t1{row2, col1, col_of(1)}, t2{row2, col2, col_of(2)}, t3{row1, col2, col_of(3)};
  // (the candidate table overrides the configuration override: a replaced site never reads either configuration table)
  BTenDev a = bten2_step(LEFT, lb, at_logical(up, UP, col1), pick(t0, dc, 4), pick(t1, dc, 4), at_logical(dn, DOWN, col1), nc, 1, false);
  BTenDev b = bten2_step(RIGHT, rb, at_logical(dn, DOWN, col2), pick(t2, dc, 4), pick(t3, dc, 4), at_logical(up, UP, col2), nc, 1, false);
  add_logs(lsum, up.logscale, dn.logscale, lb.logscale, rb.logscale);
  finish_dot4(a.t, b.t, nc, lsum, out);
  free_ten(a.t); free_ten(b.t);
  arena_.free(lsum);
  if (dc) arena_.free(dc);
}

// ReplaceTNNSiteTrace (trace.h:326-423).  cand[w][k][3] = states of the three consecutive sites.
template <typename T>
void Engine<T>::replace_tnn_trace(int row, int col, int orient, int ncand, const int32_t *cand, double *out) {
  require_ready();
  ArenaScope scope(arena_);
  const int nc = ncand > 0 ? ncand : 1;
  PG_REQUIRE(row >= 0 && col >= 0 && (orient == HORIZONTAL ? (col + 2 < Lx_ && row < Ly_) : (row + 2 < Ly_ && col < Lx_)), 1,
             "ReplaceTNNSiteTrace: sites outside the lattice");
  int *dc = upload_cand(ncand, 3, cand);
  double *lsum = zeros_f64();
  BTenDev cur;
  const DTen<T> *closing;
  if (orient == HORIZONTAL) {
    const BMPSDev &up = bmps_at_slice(UP, row), &dn = bmps_at_slice(DOWN, row);
    PG_REQUIRE(bten_size(LEFT) > col, 3, "ReplaceTNNSiteTrace: LEFT BTen missing");
    const BTenDev &rb = bten_at_slice(RIGHT, col + 2);
    BTenDev prev = bten_[LEFT][col];
    for (int k = 0; k < 3; ++k) {
      SitePick s{row, col + k, ncand > 0 ? k : -1};
      cur = bten_step(LEFT, prev, at_logical(up, UP, col + k), pick(s, dc, 3), at_logical(dn, DOWN, col + k), nc, false,
                      k == 0 ? 1 : nc);
      if (k > 0) free_ten(prev.t);
      prev = cur;
    }
    add_logs(lsum, up.logscale, dn.logscale, bten_[LEFT][col].logscale, rb.logscale);
    closing = &rb.t;
  } else {
    const BMPSDev &lf = bmps_at_slice(LEFT, col), &rt = bmps_at_slice(RIGHT, col);
    PG_REQUIRE(bten_size(UP) > row, 3, "ReplaceTNNSiteTrace: UP BTen missing");
    const BTenDev &bb = bten_at_slice(DOWN, row + 2);
    BTenDev prev = bten_[UP][row];
    for (int k = 0; k < 3; ++k) {
      SitePick s{row + k, col, ncand > 0 ? k : -1};
      cur = bten_step(UP, prev, at_logical(rt, RIGHT, row + k), pick(s, dc, 3), at_logical(lf, LEFT, row + k), nc, false,
                      k == 0 ? 1 : nc);
      if (k > 0) free_ten(prev.t);
      prev = cur;
    }
    add_logs(lsum, lf.logscale, rt.logscale, bten_[UP][row].logscale, bb.logscale);
    closing = &bb.t;
  }
  finish_dot(cur.t, nc, *closing, 1, nc, lsum, out);
  free_ten(cur.t);
  arena_.free(lsum);
  if (dc) arena_.free(dc);
}

// ReplaceSqrt5DistTwoSiteTrace (trace.h:425-536).  cand[w][k][2] = states of (ten_left, ten_right).
template <typename T>
void Engine<T>::replace_sqrt5_trace(int row1, int col1, int dir, int orient, int ncand, const int32_t *cand, double *out) {
  require_ready();
  ArenaScope scope(arena_);
  const int nc = ncand > 0 ? ncand : 1;
  int *dc = upload_cand(ncand, 2, cand);
  const int cl = ncand > 0 ? 0 : -1, cr = ncand > 0 ? 1 : -1;
  double *lsum = zeros_f64();
  BTenDev a, b, c;
  if (orient == HORIZONTAL) {
    const int row2 = row1 + 1, col2 = col1 + 1, col3 = col1 + 2;
    PG_REQUIRE(row1 >= 0 && col1 >= 0 && row2 < Ly_ && col3 < Lx_, 1, "ReplaceSqrt5DistTwoSiteTrace: sites outside the lattice");
    const BMPSDev &up = bmps_at_slice(UP, row1), &dn = bmps_at_slice(DOWN, row2);
    PG_REQUIRE(bten2_size(LEFT) > col1, 3, "ReplaceSqrt5DistTwoSiteTrace: LEFT BTen2 missing");
    const BTenDev &lb = bten2_[LEFT][col1], &rb = bten2_at_slice(RIGHT, col3);
    SitePick m0, m1, m4, m5;   // (row1,col1), (row2,col1), (row1,col3), (row2,col3)
    if (dir == LEFTUP_TO_RIGHTDOWN) { m0 = {row1, col1, cl}; m1 = {row2, col1, -1}; m4 = {row1, col3, -1}; m5 = {row2, col3, cr}; }
    else { m0 = {row1, col1, -1}; m1 = {row2, col1, cl}; m4 = {row1, col3, cr}; m5 = {row2, col3, -1}; }
    const SitePick m2{row1, col2, -1}, m3{row2, col2, -1};
    a = bten2_step(LEFT, lb, at_logical(up, UP, col1), pick(m0, dc, 2), pick(m1, dc, 2), at_logical(dn, DOWN, col1), nc, 1, false);
    b = bten2_step(RIGHT, rb, at_logical(dn, DOWN, col3), pick(m5, dc, 2), pick(m4, dc, 2), at_logical(up, UP, col3), nc, 1, false);
    c = bten2_step(LEFT, a, at_logical(up, UP, col2), pick(m2, dc, 2), pick(m3, dc, 2), at_logical(dn, DOWN, col2), nc, nc, false);
    add_logs(lsum, up.logscale, dn.logscale, lb.logscale, rb.logscale);
  } else {
    const int row2 = row1 + 1, row3 = row1 + 2, col2 = col1 + 1;
    PG_REQUIRE(row1 >= 0 && col1 >= 0 && row3 < Ly_ && col2 < Lx_, 1, "ReplaceSqrt5DistTwoSiteTrace: sites outside the lattice");
    const BMPSDev &lf = bmps_at_slice(LEFT, col1), &rt = bmps_at_slice(RIGHT, col2);
    PG_REQUIRE(bten2_size(UP) > row1, 3, "ReplaceSqrt5DistTwoSiteTrace: UP BTen2 missing");
    const BTenDev &tb = bten2_[UP][row1], &bb = bten2_at_slice(DOWN, row3);
    SitePick m0, m1, m4, m5;   // (row3,col1), (row3,col2), (row1,col1), (row1,col2)
    if (dir == LEFTUP_TO_RIGHTDOWN) { m0 = {row3, col1, -1}; m1 = {row3, col2, cr}; m4 = {row1, col1, cl}; m5 = {row1, col2, -1}; }
    else { m0 = {row3, col1, cl}; m1 = {row3, col2, -1}; m4 = {row1, col1, -1}; m5 = {row1, col2, cr}; }
    const SitePick m2{row2, col1, -1}, m3{row2, col2, -1};
    a = bten2_step(DOWN, bb, at_logical(lf, LEFT, row3), pick(m0, dc, 2), pick(m1, dc, 2), at_logical(rt, RIGHT, row3), nc, 1, false);
    b = bten2_step(UP, tb, at_logical(rt, RIGHT, row1), pick(m5, dc, 2), pick(m4, dc, 2), at_logical(lf, LEFT, row1), nc, 1, false);
    c = bten2_step(DOWN, a, at_logical(lf, LEFT, row2), pick(m2, dc, 2), pick(m3, dc, 2), at_logical(rt, RIGHT, row2), nc, nc, false);
    add_logs(lsum, lf.logscale, rt.logscale, tb.logscale, bb.logscale);
  }
  finish_dot4(c.t, b.t, nc, lsum, out);
  free_ten(a.t); free_ten(b.t); free_ten(c.t);
  arena_.free(lsum);
  if (dc) arena_.free(dc);
}

}  // namespace pepsgpu
