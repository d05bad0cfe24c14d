// BMPSWalker on the device (bmps_contractor.h:357-646, bmps/impl/bmps_walker.h:13-465, bten_operations.h:60-277).
//
// The reference forks ONE boundary MPS out of a contractor stack; it evolves on its own -- through rows of the network
// (EvolveStep) or through ANY TransferMPO the caller hands over (Evolve) --, is closed against a named boundary of the opposite
// stack (ContractRow) and keeps its own LEFT / RIGHT BTen caches for multi-site traces on one row.  Here a walker object holds
// the fork for every Monte-Carlo walker of the context at once (a deep copy of the top BMPS: tensors [walker][elements], live
// bond counts, log-scales) plus the two BTen caches; every operation is the same batched launch sequence the contractor's own
// stacks use (absorb_* / bten_step / finish_dot), with the MPO named in one of three ways (pepsgpu_walker_set_mpo):
//   * slice `num` of the network under the walkers' current configurations,
//   * slice `num` with a per-walker state for every site along it (an excited row),
//   * explicit site tensors along the slice (one set shared by the walkers, or one per walker): NOT a row of the network.
// The row operations support the UP walker / DOWN opposite pair only, as the reference does (bmps_walker.h:114-118).
#pragma once
#include "engine.h"

namespace pepsgpu {

template <typename T>
typename Engine<T>::BMPSDev Engine<T>::copy_bmps(const BMPSDev &b) {
  BMPSDev o;
  o.kmax = b.kmax; o.mlmax = b.mlmax; o.depth = b.depth;
  for (const auto &t : b.t) {
    DTen<T> c = t;
    c.p = (T *)arena_.alloc(sizeof(T) * (size_t)t.n * nw_);
    PG_CHECK_HIP(hipMemcpyAsync(c.p, t.p, sizeof(T) * (size_t)t.n * nw_, hipMemcpyDeviceToDevice, stream_));
    o.t.push_back(c);
  }
  for (int *l : b.live) {
    int *c = nullptr;
    if (l) {
      c = (int *)arena_.alloc(sizeof(int) * nw_);
      PG_CHECK_HIP(hipMemcpyAsync(c, l, sizeof(int) * nw_, hipMemcpyDeviceToDevice, stream_));
    }
    o.live.push_back(c);
  }
  o.logscale = (double *)arena_.alloc(sizeof(double) * nw_);
  PG_CHECK_HIP(hipMemcpyAsync(o.logscale, b.logscale, sizeof(double) * nw_, hipMemcpyDeviceToDevice, stream_));
  return o;
}

template <typename T>
typename Engine<T>::WalkerDev &Engine<T>::walker_ref(int id) {
  auto it = walkers_.find(id);
  PG_REQUIRE(it != walkers_.end(), 1, "BMPSWalker: no such walker");
  return it->second;
}

template <typename T>
void Engine<T>::walker_free_bten(WalkerDev &w) {
  for (auto *v : {&w.btl, &w.btr})
    for (auto &b : *v) { arena_.free(b.t.p); if (b.logscale) arena_.free(b.logscale); }
  w.btl.clear(); w.btr.clear();
  w.lcol = 0; w.rcol = 0;
}

template <typename T>
void Engine<T>::walker_free_mpo(WalkerDev &w) {
  if (w.mpo_cfg) arena_.free(w.mpo_cfg);
  if (w.mpo_tens) arena_.free(w.mpo_tens);
  w.mpo_cfg = nullptr; w.mpo_tens = nullptr; w.mpo_nt = 0; w.mpo_num = -1;
}

template <typename T>
void Engine<T>::walkers_clear() {
  for (auto &kv : walkers_) {
    walker_free_bten(kv.second);
    walker_free_mpo(kv.second);
    free_bmps(kv.second.b);
  }
  walkers_.clear();
}

// GetWalker (bmps_walker.h:51-58): a copy of the top of the stack (level < 0), or BMPSWalker(tn, stack[level], pos, level + 1,
// params) as the structure-factor mixin builds its main walker from the vacuum (structure_factor_measurement_mixin.h:121-122)
template <typename T>
int Engine<T>::walker_create(int pos, int level) {
  require_ready();
  PG_REQUIRE(bmps_size(pos) > 0, 3, "GetWalker: cannot create a walker from an empty BMPS stack");
  PG_REQUIRE(level < bmps_size(pos), 1, "BMPSWalker: level outside the stack");
  const int lv = level < 0 ? bmps_size(pos) - 1 : level;
  WalkerDev w;
  w.pos = pos;
  w.stack = lv + 1;
  w.b = copy_bmps(bmps_[pos][lv]);
  const int id = next_walker_id_++;
  walkers_.emplace(id, std::move(w));
  return id;
}

// copy construction of a walker (`auto excited_walker = main_walker;`): the BMPS and the counters.  The MPO named by walker_set_mpo is
// NOT copied (the reference's walker holds no MPO: Evolve takes it as an argument) -- the copy starts on the network's own slices --
// and its BTen caches start empty
template <typename T>
int Engine<T>::walker_clone(int id) {
  require_ready();
  WalkerDev &src = walker_ref(id);
  WalkerDev w;
  w.pos = src.pos;
  w.stack = src.stack;
  w.b = copy_bmps(src.b);
  const int nid = next_walker_id_++;
  walkers_.emplace(nid, std::move(w));
  return nid;
}

template <typename T>
void Engine<T>::walker_destroy(int id) {
  WalkerDev &w = walker_ref(id);
  walker_free_bten(w);
  walker_free_mpo(w);
  free_bmps(w.b);
  walkers_.erase(id);
}

template <typename T>
void Engine<T>::walker_info(int id, int *pos, int *stack, int *lcol, int *rcol) {
  WalkerDev &w = walker_ref(id);
  if (pos) *pos = w.pos;
  if (stack) *stack = w.stack;
  if (lcol) *lcol = w.lcol;
  if (rcol) *rcol = w.rcol;
}

// the TransferMPO of the calls that follow
template <typename T>
void Engine<T>::walker_set_mpo(int id, int num, const int32_t *states, const double *tensors, int n_tensors) {
  require_ready();
  WalkerDev &w = walker_ref(id);
  const bool hor = (w.pos == UP || w.pos == DOWN);
  const int N = mps_len(w.pos), lim = hor ? Ly_ : Lx_;
  PG_REQUIRE(num >= 0 && num < lim, 1, "BMPSWalker: MPO slice outside the lattice");
  PG_REQUIRE(!(states && tensors), 1, "BMPSWalker: name the MPO by states OR by tensors");
  PG_REQUIRE(!tensors || n_tensors == 1 || n_tensors == nw_, 1, "BMPSWalker: explicit MPO tensors come as one set or one set per walker");
  // validate and build BEFORE the walker's current MPO is released: a bad argument leaves the walker as it was
  std::vector<int> tab;
  if (states) {
    // a full configuration table with the slice replaced (same layout as the walkers' own table: every kernel's selector works)
    tab = hcfg_;
    for (int wk = 0; wk < nw_; ++wk)
      for (int j = 0; j < N; ++j) {
        const int s = states[(size_t)wk * N + j];
        PG_REQUIRE(s >= 0 && s < dp_, 4, "BMPSWalker: MPO state exceeds the physical dimension");
        const int r = hor ? num : j, c = hor ? j : num;
        tab[(size_t)wk * Ly_ * Lx_ + r * Lx_ + c] = s;
      }
  }
  walker_free_mpo(w);
  w.mpo_num = num;
  if (states) {
    w.mpo_cfg = (int *)arena_.alloc(sizeof(int) * tab.size());
    PG_CHECK_HIP(hipMemcpyAsync(w.mpo_cfg, tab.data(), sizeof(int) * tab.size(), hipMemcpyHostToDevice, stream_));
    PG_CHECK_HIP(hipStreamSynchronize(stream_));
  } else if (tensors) {
    // host [nt][N][L][D][R][U] zero padded to D^4 -> device [N][nt][slot], compact inside the slot (as the SITPS)
    std::vector<T> buf((size_t)N * n_tensors * slot_, T(0));
    for (int j = 0; j < N; ++j) {
      int dd[4];
      site_dims(hor ? num : j, hor ? j : num, dd);
      for (int q = 0; q < n_tensors; ++q) {
        const size_t src0 = ((size_t)q * N + j) * slot_, dst0 = ((size_t)j * n_tensors + q) * slot_;
        size_t o = 0;
        for (int a = 0; a < dd[0]; ++a)
          for (int b = 0; b < dd[1]; ++b)
            for (int c = 0; c < dd[2]; ++c)
              for (int e = 0; e < dd[3]; ++e) {
                const size_t src = src0 + (((size_t)a * D_ + b) * D_ + c) * D_ + e;
                if constexpr (kCplx) {
                  typedef typename real_of<T>::type R;
                  buf[dst0 + o++] = T(R(tensors[2 * src]), R(tensors[2 * src + 1]));
                } else {
                  buf[dst0 + o++] = T(tensors[src]);
                }
              }
      }
    }
    w.mpo_tens = (T *)arena_.alloc(sizeof(T) * buf.size());
    w.mpo_nt = n_tensors;
    PG_CHECK_HIP(hipMemcpyAsync(w.mpo_tens, buf.data(), sizeof(T) * buf.size(), hipMemcpyHostToDevice, stream_));
    PG_CHECK_HIP(hipStreamSynchronize(stream_));
    ensure_iota();
  }
}

template <typename T>
void Engine<T>::ensure_iota() {
  if (iota_) return;
  std::vector<int> h(maxw_ + 1);
  for (int i = 0; i < maxw_; ++i) h[i + 1] = i;
  h[0] = 0;                                   // iota_[0] alone = the selector of a shared tensor set (stride 0)
  iota_ = (int *)arena_.alloc(sizeof(int) * h.size());
  PG_CHECK_HIP(hipMemcpyAsync(iota_, h.data(), sizeof(int) * h.size(), hipMemcpyHostToDevice, stream_));
  PG_CHECK_HIP(hipStreamSynchronize(stream_));
}

// scope guard: cfg_site / site selection of slice w.mpo_num follow the walker's MPO while it lives
template <typename T>
struct Engine<T>::MpoScope {
  Engine<T> &e;
  MpoScope(Engine<T> &eng, const WalkerDev &w) : e(eng) {
    PG_REQUIRE(w.mpo_num >= 0, 3, "BMPSWalker: no MPO set (pepsgpu_walker_set_mpo)");
    PG_REQUIRE(!e.cfg_ovr_tab_, 3, "BMPSWalker: a configuration override is set (pepsgpu_cfg_override_slice): clear it first");
    e.ovr_hor_ = (w.pos == UP || w.pos == DOWN);
    e.ovr_num_ = w.mpo_num;
    e.ovr_cfg_ = w.mpo_cfg;
    e.ovr_tens_ = w.mpo_tens;
    e.ovr_nt_ = w.mpo_nt;
    e.ovr_on_ = true;
  }
  ~MpoScope() { e.ovr_on_ = false; e.ovr_cfg_ = nullptr; e.ovr_tens_ = nullptr; }
};

template <typename T>
typename Engine<T>::BMPSDev Engine<T>::absorb_any(int pos, int num, const BMPSDev &in) {
  if constexpr (kCplx) {
    if (scheme_ != 0 && mps_len(pos) > 2) return absorb_variational(pos, num, in);
    return absorb_simple(pos, num, in);
  } else {
    if (scheme_ != 0 && mps_len(pos) > 2) return absorb_variational(pos, num, in);
    return absorb_svd(pos, num, in);
  }
}

// Evolve(mpo) (bmps_walker.h:13-21): the stack-size counter is NOT advanced
template <typename T>
void Engine<T>::walker_evolve(int id) {
  require_ready();
  WalkerDev &w = walker_ref(id);
  MpoScope scope(*this, w);
  // (the routing hints of absorb_svd are keyed by (pos, num): a foreign MPO must not teach them anything about the network's row)
  const char keep = redo_seen_[w.pos][w.mpo_num];
  BMPSDev out;
  try { out = absorb_any(w.pos, w.mpo_num, w.b); } catch (...) { redo_seen_[w.pos][w.mpo_num] = keep; throw; }
  redo_seen_[w.pos][w.mpo_num] = keep;
  free_bmps(w.b);
  w.b = std::move(out);
  walker_free_bten(w);
}

// EvolveStep (bmps_walker.h:23-49)
template <typename T>
void Engine<T>::walker_evolve_step(int id) {
  require_ready();
  WalkerDev &w = walker_ref(id);
  PG_REQUIRE(w.stack > 0, 3, "BMPSWalker::EvolveStep: empty walker");
  int num;
  if (w.pos == UP || w.pos == LEFT) num = w.stack - 1;
  else if (w.pos == DOWN) num = Ly_ - w.stack;
  else num = Lx_ - w.stack;
  if (w.pos == UP && num >= Ly_ - 1) return;
  if (w.pos == LEFT && num >= Lx_ - 1) return;
  PG_REQUIRE(num >= 0, 3, "BMPSWalker::EvolveStep: no slice left to absorb");
  BMPSDev out = absorb_any(w.pos, num, w.b);
  free_bmps(w.b);
  w.b = std::move(out);
  w.stack++;
  walker_free_bten(w);
}

template <typename T>
const typename Engine<T>::BMPSDev &Engine<T>::walker_opposite(const WalkerDev &w, int opp_level, const char *what) {
  PG_REQUIRE(w.pos == UP, 1, "BMPSWalker: unsupported direction pair (walker must be UP, opposite must be DOWN)");
  PG_REQUIRE(opp_level >= 0 && opp_level < bmps_size(DOWN), 3, "BMPSWalker: opposite boundary not available (DOWN stack level)");
  (void)what;
  return bmps_[DOWN][opp_level];
}

template <typename T>
void Engine<T>::walker_grow_left(WalkerDev &w, const BMPSDev &opp) {
  const int N = Lx_;
  if (w.btl.empty()) {
    PG_REQUIRE(w.lcol == 0, 3, "BMPSWalker::GrowBTenLeftStep: bten_left_ is empty but col != 0");
    BTenDev v; v.t = ones3(); v.logscale = zeros_f64();
    w.btl.push_back(v);
  }
  PG_REQUIRE(w.lcol < N, 3, "BMPSWalker::GrowBTenLeftStep: Cannot grow beyond N.");
  const int col = w.lcol, j1 = N - 1 - col;
  auto lv = [](const BMPSDev &b, int j) -> const int * { return (int)b.live.size() > j ? b.live[j] : nullptr; };
  BTenDev nb = bten_step(LEFT, w.btl.back(), w.b.t[j1], cfg_site(w.mpo_num, col), opp.t[col], 1, true, 1, lv(w.b, j1), lv(w.b, j1 + 1),
                         lv(opp, col), lv(opp, col + 1));
  w.btl.push_back(nb);
  w.lcol++;
}

template <typename T>
void Engine<T>::walker_grow_right(WalkerDev &w, const BMPSDev &opp) {
  const int N = Lx_;
  PG_REQUIRE(!w.btr.empty(), 3, "BMPSWalker::GrowBTenRightStep: Right BTen cache is empty. Call InitBTenRight first.");
  PG_REQUIRE(w.rcol > 0, 3, "BMPSWalker::GrowBTenRightStep: Cannot grow further left. col is already 0.");
  const int col = w.rcol - 1, j2 = N - 1 - col;
  auto lv = [](const BMPSDev &b, int j) -> const int * { return (int)b.live.size() > j ? b.live[j] : nullptr; };
  BTenDev nb = bten_step(RIGHT, w.btr.back(), opp.t[col], cfg_site(w.mpo_num, col), w.b.t[j2], 1, true, 1, lv(opp, col), lv(opp, col + 1),
                         lv(w.b, j2), lv(w.b, j2 + 1));
  w.btr.push_back(nb);
  w.rcol--;
}

template <typename T>
void Engine<T>::walker_init_bten(int id, int opp_level, int side, int target_col) {
  require_ready();
  WalkerDev &w = walker_ref(id);
  const BMPSDev &opp = walker_opposite(w, opp_level, "InitBTen");
  MpoScope scope(*this, w);
  PG_REQUIRE(target_col >= 0, 1, "BMPSWalker::InitBTen: negative column");
  const int N = Lx_;
  if (side == LEFT) {
    for (auto &b : w.btl) { arena_.free(b.t.p); arena_.free(b.logscale); }
    w.btl.clear(); w.lcol = 0;
    BTenDev v; v.t = ones3(); v.logscale = zeros_f64();
    w.btl.push_back(v);
    while (w.lcol < target_col && w.lcol < N) walker_grow_left(w, opp);
  } else {
    PG_REQUIRE(side == RIGHT, 1, "BMPSWalker::InitBTen: position must be LEFT or RIGHT");
    for (auto &b : w.btr) { arena_.free(b.t.p); arena_.free(b.logscale); }
    w.btr.clear(); w.rcol = N;
    BTenDev v; v.t = ones3(); v.logscale = zeros_f64();
    w.btr.push_back(v);
    while (w.rcol > target_col + 1 && w.rcol > 0) walker_grow_right(w, opp);
  }
}

template <typename T>
void Engine<T>::walker_grow_bten_step(int id, int opp_level, int side) {
  require_ready();
  WalkerDev &w = walker_ref(id);
  const BMPSDev &opp = walker_opposite(w, opp_level, "GrowBTenStep");
  MpoScope scope(*this, w);
  if (side == LEFT) walker_grow_left(w, opp);
  else { PG_REQUIRE(side == RIGHT, 1, "BMPSWalker: position must be LEFT or RIGHT"); walker_grow_right(w, opp); }
}

// ShiftBTenWindow (bmps_walker.h:326-350)
template <typename T>
void Engine<T>::walker_shift_bten_window(int id, int opp_level, int side) {
  require_ready();
  WalkerDev &w = walker_ref(id);
  const BMPSDev &opp = walker_opposite(w, opp_level, "ShiftBTenWindow");
  MpoScope scope(*this, w);
  if (side == LEFT) {
    PG_REQUIRE(!w.btl.empty(), 3, "BMPSWalker::ShiftBTenWindow: Left BTen cache is empty.");
    PG_REQUIRE(!w.btr.empty(), 3, "BMPSWalker::GrowBTenRightStep: Right BTen cache is empty. Call InitBTenRight first.");
    arena_.free(w.btl.back().t.p); arena_.free(w.btl.back().logscale);
    w.btl.pop_back(); w.lcol--;
    walker_grow_right(w, opp);
  } else {
    PG_REQUIRE(side == RIGHT, 1, "BMPSWalker: position must be LEFT or RIGHT");
    PG_REQUIRE(!w.btr.empty(), 3, "BMPSWalker::ShiftBTenWindow: Right BTen cache is empty.");
    arena_.free(w.btr.back().t.p); arena_.free(w.btr.back().logscale);
    w.btr.pop_back(); w.rcol++;
    walker_grow_left(w, opp);
  }
}

// the replacement site of a trace: the MPO's own tensor (nothing given), a component of the SITPS per walker (states), or an
// explicit tensor (one shared or one per walker, host layout [nt][L][D][R][U] padded)
template <typename T>
typename Engine<T>::SiteSel Engine<T>::walker_site(const WalkerDev &w, int col, const int32_t *states, int sstride, const double *tensor,
                                                   int n_tensors, long tstride, std::vector<void *> &tmp) {
  if (!states && !tensor) return cfg_site(w.mpo_num, col);
  if (states) {
    std::vector<int> h(nw_);
    for (int k = 0; k < nw_; ++k) {
      h[k] = states[(size_t)k * sstride];
      PG_REQUIRE(h[k] >= 0 && h[k] < dp_, 4, "BMPSWalker: replacement state exceeds the physical dimension");
    }
    int *d = (int *)arena_.alloc(sizeof(int) * nw_);
    tmp.push_back(d);
    PG_CHECK_HIP(hipMemcpyAsync(d, h.data(), sizeof(int) * nw_, hipMemcpyHostToDevice, stream_));
    PG_CHECK_HIP(hipStreamSynchronize(stream_));
    SiteSel s{w.mpo_num, col, d, 1};
    s.base = sitps_ + (long)(w.mpo_num * Lx_ + col) * dp_ * slot_;     // the SITPS itself, whatever the MPO is made of
    return s;
  }
  PG_REQUIRE(n_tensors == 1 || n_tensors == nw_, 1, "BMPSWalker: a replacement tensor comes once or once per walker");
  int dd[4];
  site_dims(w.mpo_num, col, dd);
  std::vector<T> buf((size_t)n_tensors * slot_, T(0));
  for (int q = 0; q < n_tensors; ++q) {
    size_t o = 0;
    for (int a = 0; a < dd[0]; ++a)
      for (int b = 0; b < dd[1]; ++b)
        for (int c = 0; c < dd[2]; ++c)
          for (int e = 0; e < dd[3]; ++e) {
            const size_t src = (size_t)q * tstride + (((size_t)a * D_ + b) * D_ + c) * D_ + e;
            if constexpr (kCplx) {
              typedef typename real_of<T>::type R;
              buf[(size_t)q * slot_ + o++] = T(R(tensor[2 * src]), R(tensor[2 * src + 1]));
            } else {
              buf[(size_t)q * slot_ + o++] = T(tensor[src]);
            }
          }
  }
  T *d = (T *)arena_.alloc(sizeof(T) * buf.size());
  tmp.push_back(d);
  PG_CHECK_HIP(hipMemcpyAsync(d, buf.data(), sizeof(T) * buf.size(), hipMemcpyHostToDevice, stream_));
  PG_CHECK_HIP(hipStreamSynchronize(stream_));
  ensure_iota();
  SiteSel s{w.mpo_num, col, n_tensors == 1 ? iota_ : iota_ + 1, n_tensors == 1 ? 0 : 1};
  s.base = d;
  return s;
}

// TraceWithBTen (bmps_walker.h:352-392) / TraceWithTwoSiteBTen (:394-463); two_site: states = [n][2], tensors = [nt][2][D^4]
template <typename T>
void Engine<T>::walker_trace(int id, int opp_level, int site_col, int two_site, const int32_t *states, const double *tensors, int n_tensors,
                             double *out) {
  require_ready();
  WalkerDev &w = walker_ref(id);
  const BMPSDev &opp = walker_opposite(w, opp_level, "TraceWithBTen");
  MpoScope scope(*this, w);
  ArenaScope ascope(arena_);
  const int N = Lx_;
  PG_REQUIRE(site_col >= 0 && site_col + (two_site ? 1 : 0) < N, 1, "BMPSWalker::TraceWithBTen: site column out of bounds");
  PG_REQUIRE(!w.btl.empty() && !w.btr.empty(), 3, "BMPSWalker::TraceWithBTen: BTen caches not initialized.");
  PG_REQUIRE(w.lcol >= site_col, 3, "BMPSWalker::TraceWithBTen: Left BTen insufficient.");
  const int last = site_col + (two_site ? 1 : 0);
  PG_REQUIRE(w.rcol <= last + 1, 3, "BMPSWalker::TraceWithBTen: Right BTen insufficient.");
  const int right_idx = N - 1 - last;
  PG_REQUIRE(site_col < (int)w.btl.size() && right_idx < (int)w.btr.size(), 3, "BMPSWalker::TraceWithBTen: BTen index out of bounds.");
  std::vector<void *> tmp;
  const int ns = two_site ? 2 : 1;
  auto lv = [](const BMPSDev &b, int j) -> const int * { return (int)b.live.size() > j ? b.live[j] : nullptr; };
  const BTenDev &lb = w.btl[site_col], &rb = w.btr[right_idx];
  SiteSel sa = walker_site(w, site_col, states, ns, tensors, n_tensors, (long)ns * slot_, tmp);
  const int ja = N - 1 - site_col;
  BTenDev t2 = bten_step(LEFT, lb, w.b.t[ja], sa, opp.t[site_col], 1, false, 1, lv(w.b, ja), lv(w.b, ja + 1), lv(opp, site_col),
                         lv(opp, site_col + 1));
  if (two_site) {
    SiteSel sb = walker_site(w, site_col + 1, states ? states + 1 : nullptr, ns, tensors ? tensors + slot_ * kOut : nullptr, n_tensors,
                             (long)ns * slot_, tmp);
    const int jb = N - 2 - site_col;
    BTenDev t3 = bten_step(LEFT, t2, w.b.t[jb], sb, opp.t[site_col + 1], 1, false, 1, lv(w.b, jb), lv(w.b, jb + 1), lv(opp, site_col + 1),
                           lv(opp, site_col + 2));
    free_ten(t2.t);
    t2 = t3;
  }
  double *lsum = zeros_f64();
  add_logs(lsum, w.b.logscale, opp.logscale, lb.logscale, rb.logscale);
  finish_dot(t2.t, 1, rb.t, 1, 1, lsum, out);
  free_ten(t2.t);
  arena_.free(lsum);
  for (void *p : tmp) arena_.free(p);
}

// ContractRow (bmps_walker.h:60-214): <walker | mpo | opposite>.  The reference multiplies the columns right to left into a
// six-leg accumulator; the value is the same closed network the BTen route contracts (its own tests assert exactly that,
// test_bmps_contractor.cpp:1233-1274), so it is computed as LEFT environment over the first N - 1 columns, vacuum on the right,
// trace at the last column -- without touching the walker's caches.
template <typename T>
void Engine<T>::walker_contract_row(int id, int opp_level, double *out) {
  require_ready();
  WalkerDev &w = walker_ref(id);
  (void)walker_opposite(w, opp_level, "ContractRow");
  std::vector<BTenDev> kl, kr;
  kl.swap(w.btl); kr.swap(w.btr);
  const int lc = w.lcol, rc = w.rcol;
  w.lcol = 0; w.rcol = 0;
  auto restore = [&]() {
    walker_free_bten(w);
    w.btl.swap(kl); w.btr.swap(kr);
    w.lcol = lc; w.rcol = rc;
  };
  try {
    walker_init_bten(id, opp_level, LEFT, Lx_ - 1);
    walker_init_bten(id, opp_level, RIGHT, Lx_ - 1);
    walker_trace(id, opp_level, Lx_ - 1, 0, nullptr, nullptr, 0, out);
  } catch (...) { restore(); throw; }
  restore();
}

template <typename T>
void Engine<T>::walker_clear_bten(int id) { walker_free_bten(walker_ref(id)); }

template <typename T>
void Engine<T>::walker_get_tensor(int id, int idx, int *dims, double *out, double *logscale) {
  WalkerDev &w = walker_ref(id);
  PG_REQUIRE(idx >= 0 && idx < (int)w.b.t.size(), 1, "BMPS tensor index out of range");
  const DTen<T> &t = w.b.t[idx];
  dims[0] = t.d[0]; dims[1] = t.d[1]; dims[2] = t.d[2];
  if (out) {
    std::vector<T> h((size_t)t.n * nw_);
    PG_CHECK_HIP(hipMemcpyAsync(h.data(), t.p, h.size() * sizeof(T), hipMemcpyDeviceToHost, stream_));
    PG_CHECK_HIP(hipStreamSynchronize(stream_));
    for (size_t i = 0; i < h.size(); ++i) {
      if constexpr (kCplx) { out[2 * i] = (double)h[i].re; out[2 * i + 1] = (double)h[i].im; }
      else out[i] = (double)h[i];
    }
  }
  if (logscale) {
    PG_CHECK_HIP(hipMemcpyAsync(logscale, w.b.logscale, nw_ * sizeof(double), hipMemcpyDeviceToHost, stream_));
    PG_CHECK_HIP(hipStreamSynchronize(stream_));
  }
}

}  // namespace pepsgpu
