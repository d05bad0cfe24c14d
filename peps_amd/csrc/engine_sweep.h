// One row (or column) of nearest-neighbour exchange updates of the Monte-Carlo sweep on the device
// (MCUpdateSquareNNUpdateBaseOBC::operator() + MCUpdateSquareNNExchangeOBC::TwoSiteNNUpdateLocalImpl, square_nn_updater.h:25-83,
// :142-189; TPSWaveFunctionComponent::UpdateLocal, wave_function_component.h:345-378).
//
// The host-driven form pays, per bond, a candidate upload, the read-back of psi', the Metropolis test on the host, an upload of
// the whole configuration table for the accepted walkers and two stream synchronisations: 264 bonds x ~3 ms per sweep of 8192
// walkers, 3-4 x the contraction work (7.6 k sweeps/s against 29 k amplitude-equivalents).  Here ONE call runs a
// whole slice: InitBTen + GrowFullBTen of the slice, then per bond the replacement trace of the exchanged pair (candidate table
// built on the device), the Metropolis test, the exchange in the device's configuration table, and the BTen window shift -- no
// host round trip inside the slice.
//
// Identical chains: the reference draws a uniform deviate only when the spins differ and |psi'| < |psi| (square_nn_updater.h:
// 160-170).  The host hands over, per walker, the NEXT `n_uniform` deviates of that walker's std::mt19937 stream (drawn ahead into
// a queue, qlpeps_gpu.h: UniformQueue); the kernel consumes them in order and reports how many it took, the host pops exactly
// those -- the deviates a walker consumes are the ones the reference's updater would have drawn, in the same order.
//
// Environment bookkeeping: UpdateLocal erases the environments that cross an updated site (EraseEnvsAfterUpdate).  Inside a slice
// pass that erase never removes anything the pass still holds (LEFT covers [0, col), RIGHT covers (col + 1, N): checked against
// the reference's flow), so it is applied unconditionally -- the decision "did any walker accept" would cost a read-back per bond.
#pragma once
#include "engine.h"

namespace pepsgpu {

// amplitude helpers, uniform over the real (double) and the complex (cplx<double>) accumulation type
__device__ __forceinline__ double sw_abs(double x) { return fabs(x); }
__device__ __forceinline__ double sw_abs(const cplx<double> &x) { return sqrt(x.re * x.re + x.im * x.im); }
__device__ __forceinline__ double sw_scaled(double x, double s) { return x * s; }
__device__ __forceinline__ cplx<double> sw_scaled(const cplx<double> &x, double s) { return cplx<double>(x.re * s, x.im * s); }

// The candidate of the "exchange" move on the bond (s1, s2).  tab == nullptr: cand[w] = (cfg[w][s2], cfg[w][s1]), the exchanged
// pair (bosons).  tab [dp * dp][2] (round 6): the pair the move proposes for the states (a, b) = tab[a * dp + b] -- a fermionic state
// lives on the device as extended states (state + d * variant, pepsgpu.h) and the exchange of two sites adjacent in the mode order
// changes their variants by a rule that is local in (a, b) (TPSWaveFunctionComponent::DeviceStatesNN of the host layer tabulates it).
// same[w] = 1 when the move is the identity (square_nn_updater.h:149-151 returns before any contraction, and so do the environment
// steps of the replacement trace here).
__global__ void sweep_swap_cand_kernel(const int *__restrict__ cfg, int sites, int s1, int s2, int *__restrict__ cand,
                                       int *__restrict__ same, int n, const int *__restrict__ tab = nullptr, int dp = 0) {
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n) return;
  const int a = cfg[(long)w * sites + s1], b = cfg[(long)w * sites + s2];
  const int c1 = tab ? tab[2 * (a * dp + b)] : b, c2 = tab ? tab[2 * (a * dp + b) + 1] : a;
  cand[2 * w] = c1;
  cand[2 * w + 1] = c2;
  same[w] = (c1 == a && c2 == b);
}

// Metropolis test of the exchange (square_nn_updater.h:149-170) and the accepted exchange itself
template <typename AccT>
__global__ void sweep_metropolis_exchange_kernel(int *__restrict__ cfg, int sites, int s1, int s2, const AccT *__restrict__ res,
                                                 const double *__restrict__ lsum, AccT *__restrict__ amp,
                                                 const double *__restrict__ uni, int nu, int *__restrict__ uptr,
                                                 int *__restrict__ acc, int *__restrict__ overrun, int *__restrict__ acc_now, int n,
                                                 const int *__restrict__ cand, const int *__restrict__ same) {
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n) return;
  acc_now[w] = 0;                                         // (1: this bond's exchange was accepted)
  if (same[w]) return;                                    // :149-151
  const AccT psi_b = sw_scaled(AccT(res[w]), exp(lsum[w]));
  const double pa = sw_abs(amp[w]), pb = sw_abs(psi_b);
  bool exchange;
  if (pb >= pa) exchange = true;
  else {
    const double div = pb / pa;
    const int q = uptr[w];
    if (q >= nu) { *overrun = 1; return; }               // (cannot happen: the host hands over one deviate per bond of the slice)
    exchange = uni[(long)w * nu + q] < div * div;
    uptr[w] = q + 1;
  }
  if (exchange) {
    cfg[(long)w * sites + s1] = cand[2 * w];
    cfg[(long)w * sites + s2] = cand[2 * w + 1];
    amp[w] = psi_b;
    acc[w] += 1;
    acc_now[w] = 1;
  }
}

// every pair of states of the bond as the candidate table of one replacement trace: cand[w][k] = (k / dim, k % dim)
__global__ void sweep_all_cand_kernel(int *__restrict__ cand, int dim, int n) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const int nc = dim * dim;
  if (e >= n * nc) return;
  const int k = e % nc;
  cand[2 * e] = k / dim;
  cand[2 * e + 1] = k % dim;
}

// MCUpdateSquareNNFullSpaceUpdateOBC::TwoSiteNNUpdateLocalImpl (square_nn_updater.h:253-293) for one bond: the weights
// |psi_k / psi|^2 of the dim^2 states of the pair (the current one keeps its stored amplitude), SuwaTodoStateUpdate
// (suwa_todo_update.h:53-112) with the walker's next deviate, and the accepted move.  The deviate: the reference draws
// std::uniform_real_distribution<long double> from a std::mt19937, i.e. generate_canonical<long double, 64> = two 32-bit words,
// (w0 + 2^32 w1) / 2^64 -- the host hands over the two raw words per bond (always two: the count does not depend on the data), so a
// walker consumes its engine exactly as the reference's updater does.  The arithmetic here is float64 where the reference's is
// long double: a decision differs only when the deviate falls within 2^-53 of a boundary of the cumulative weights.
constexpr int SW_MAXC = 16;
// SuwaTodoStateUpdate (suwa_todo_update.h:53-112) for nc <= SW_MAXC states: wt (modified: the largest weight is swapped to the front, as
// the reference does), init = the current state, (w0, w1) = the two engine words of the long double draw.  Returns the new state.
__device__ __forceinline__ int sw_suwa_todo_decide(double *wt, int nc, int init, unsigned w0, unsigned w1) {
  int mx = 0;
  for (int k = 1; k < nc; ++k)
    if (wt[k] > wt[mx]) mx = k;                           // std::max_element: the first of equal maxima
  if (mx != 0) { const double t = wt[0]; wt[0] = wt[mx]; wt[mx] = t; }
  if (init == mx) init = 0;
  else if (init == 0) init = mx;
  double cs[SW_MAXC];
  cs[0] = wt[0];
  for (int k = 1; k < nc; ++k) cs[k] = cs[k - 1] + wt[k];
  const double S = cs[nc - 1];
  const double s_im1 = init == 0 ? 0.0 : cs[init - 1];
  double start = s_im1 + wt[0];
  if (start >= S) start -= S;
  double u = ((double)w0 + 4294967296.0 * (double)w1) * 5.421010862427522e-20;      // / 2^64
  if (u >= 1.0) u = 0.9999999999999999;
  const double hi = nextafter(start + wt[init], start);
  double x = u * (hi - start) + start;
  if (x >= S) x -= S;
  int fin = 0;
  while (fin < nc && !(cs[fin] > x)) ++fin;               // std::upper_bound
  if (fin >= nc) fin = nc - 1;
  if (mx != 0) {
    if (fin == 0) fin = mx;
    else if (fin == mx) fin = 0;
  }
  return fin;
}

// a chain of the update on one weight vector (kernel test: the reference's unit cases against the host's long double form)
__global__ void sweep_suwa_todo_chain_kernel(const double *__restrict__ weights, int nc, int init, const unsigned *__restrict__ words, int steps,
                                             int *__restrict__ out) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  int st = init;
  for (int s = 0; s < steps; ++s) {
    double wt[SW_MAXC];
    for (int k = 0; k < nc; ++k) wt[k] = weights[k];
    st = sw_suwa_todo_decide(wt, nc, st, words[2 * s], words[2 * s + 1]);
    out[s] = st;
  }
}

template <typename AccT>
__global__ void sweep_suwa_todo_kernel(int *__restrict__ cfg, int sites, int s1, int s2, int dim, const AccT *__restrict__ res,
                                       const double *__restrict__ lsum, AccT *__restrict__ amp, const unsigned *__restrict__ words,
                                       int words_per_walker, int bond, int *__restrict__ acc, int *__restrict__ acc_now, int n) {
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n) return;
  const int nc = dim * dim;
  const int c1 = cfg[(long)w * sites + s1], c2 = cfg[(long)w * sites + s2];
  const int init0 = c1 * dim + c2;
  const double sc = exp(lsum[w]);
  const AccT a0 = amp[w];
  const double pa = sw_abs(a0);
  double wt[SW_MAXC];
  for (int k = 0; k < nc; ++k) {
    const double r = (k == init0) ? 1.0 : sw_abs(sw_scaled(AccT(res[(long)w * nc + k]), sc)) / pa;
    wt[k] = r * r;
  }
  const int fin = sw_suwa_todo_decide(wt, nc, init0, words[(long)w * words_per_walker + 2 * bond], words[(long)w * words_per_walker + 2 * bond + 1]);
  const int changed = fin != init0;
  acc_now[w] = changed;
  if (changed) {
    cfg[(long)w * sites + s1] = fin / dim;
    cfg[(long)w * sites + s2] = fin % dim;
    amp[w] = sw_scaled(AccT(res[(long)w * nc + fin]), sc);
    acc[w] += 1;
  }
}

// dst[w] = src[w] for the walkers with take[w] != 0 (len elements per walker)
template <typename T>
__global__ void sweep_take_kernel(T *__restrict__ dst, const T *__restrict__ src, long len, const int *__restrict__ take) {
  const int w = blockIdx.y;
  if (!take[w]) return;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < len; e += (long)gridDim.x * blockDim.x) dst[(long)w * len + e] = src[(long)w * len + e];
}

__global__ void sweep_gather_slice_kernel(const int *__restrict__ cfg, int sites, int first, int stride, int len, int *__restrict__ out, int n) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n * len) return;
  const int w = e / len, j = e - w * len;
  out[e] = cfg[(long)w * sites + first + j * stride];
}

// psi'[w][j] = res[w] exp(lsum[w]) into column j of a [n][stride] table
// (same[w] != 0: the exchange was the identity and its trace was skipped -- psi' = psi, column 0 of the table)
template <typename AccT>
__global__ void sweep_store_value_kernel(const AccT *__restrict__ res, const double *__restrict__ lsum, double *__restrict__ out,
                                         int stride, int j, const int *__restrict__ same, int n) {
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w < n) out[(long)w * stride + j] = (same && same[w]) ? out[(long)w * stride] : (double)res[w] * exp(lsum[w]);
}

// One row / column of the energy evaluation (SquareNNNModelEnergySolver::CalEnergyAndHolesImpl, square_nnn_energy_solver.h:
// 116-200 row pass, bond_traversal_mixin.h:120-144 column pass) for models whose nearest-neighbour off-diagonal term is the
// exchange of the two site states (XXZ, J1-J2, t-J ...): InitBTen + GrowFullBTen, psi of the slice, and for every bond the
// amplitude of the configuration with the two states exchanged -- all on the device, ONE read-back per slice (psi [n] and
// psi_ex [n][N-1]); with punch_holes the hole of every site of the slice is stored in HBM on the way (PunchHole, :163).
template <typename T>
void Engine<T>::nn_exchange_slice(int orient, int slice, int punch_holes, double *psi_out, double *psi_ex_out) {
  require_ready();
  if constexpr (kCplx) {
    PG_REQUIRE(false, 1, "device-side energy slice: real element types only (complex contexts use the per-bond calls)");
  } else {
    PG_REQUIRE(orient == HORIZONTAL || orient == VERTICAL, 1, "bad orientation");
    const int N = orient == HORIZONTAL ? Lx_ : Ly_, lim = orient == HORIZONTAL ? Ly_ : Lx_;
    PG_REQUIRE(slice >= 0 && slice < lim, 1, "slice outside the lattice");
    const int sites = Ly_ * Lx_;
    if (punch_holes && !holes_) {
      holes_ = (T *)arena_.alloc(sizeof(T) * (size_t)maxw_ * Ly_ * Lx_ * slot_);
      holes_ls_ = (double *)arena_.alloc(sizeof(double) * (size_t)maxw_ * Ly_ * Lx_);
    }
    double *dval = (double *)arena_.alloc(sizeof(double) * (size_t)nw_ * N);     // column 0: psi, columns 1..N-1: psi_ex of bond j-1
    int *dcand = (int *)arena_.alloc(sizeof(int) * 3 * (size_t)nw_);
    int *dsame = dcand + 2 * (size_t)nw_;
    auto release = [&]() { arena_.free(dval); arena_.free(dcand); };
    try {
      const int lo = orient == HORIZONTAL ? LEFT : UP, hi = orient == HORIZONTAL ? RIGHT : DOWN;
      const int remain = punch_holes ? 1 : 2;         // :143 GrowFullBTen(RIGHT, row, 1, true) with holes, 2 in the column pass
      init_bten(lo, slice);
      grow_full_bten(hi, slice, remain, 1);
      const int gb = (nw_ + 255) / 256;
      {
        double *lsum = nullptr;
        Acc *res = nn_trace_device(orient == HORIZONTAL ? slice : 0, orient == HORIZONTAL ? 0 : slice, orient, 1, nullptr, &lsum);
        hipLaunchKernelGGL(sweep_store_value_kernel<Acc>, dim3(gb), dim3(256), 0, stream_, (const Acc *)res, (const double *)lsum, dval, N, 0, (const int *)nullptr, nw_);
        PG_CHECK_HIP(hipGetLastError());
        arena_.free(res); arena_.free(lsum);
      }
      for (int j = 0; j < N; ++j) {
        const int r1 = orient == HORIZONTAL ? slice : j, c1 = orient == HORIZONTAL ? j : slice;
        if (punch_holes) punch_hole(r1, c1, orient, nullptr);
        if (j + 1 < N) {
          const int r2 = orient == HORIZONTAL ? slice : j + 1, c2 = orient == HORIZONTAL ? j + 1 : slice;
          hipLaunchKernelGGL(sweep_swap_cand_kernel, dim3(gb), dim3(256), 0, stream_, (const int *)cfg_, sites, r1 * Lx_ + c1, r2 * Lx_ + c2,
                             dcand, dsame, nw_);
          PG_CHECK_HIP(hipGetLastError());
          double *lsum = nullptr;
          Acc *res = nn_trace_device(r1, c1, orient, 1, dcand, &lsum, dsame);
          hipLaunchKernelGGL(sweep_store_value_kernel<Acc>, dim3(gb), dim3(256), 0, stream_, (const Acc *)res, (const double *)lsum, dval, N,
                             j + 1, (const int *)dsame, nw_);
          PG_CHECK_HIP(hipGetLastError());
          arena_.free(res); arena_.free(lsum);
          if (remain == 1 || j + 2 < N) shift_bten_window(hi);
        }
      }
      std::vector<double> h((size_t)nw_ * N);
      PG_CHECK_HIP(hipMemcpyAsync(h.data(), dval, sizeof(double) * h.size(), hipMemcpyDeviceToHost, stream_));
      PG_CHECK_HIP(hipStreamSynchronize(stream_));
      for (int w = 0; w < nw_; ++w) {
        psi_out[w] = h[(size_t)w * N];
        for (int j = 0; j + 1 < N; ++j) psi_ex_out[(size_t)w * (N - 1) + j] = h[(size_t)w * N + j + 1];
      }
    } catch (...) { release(); throw; }
    release();
  }
}

template <typename T>
void Engine<T>::grow_bten_step_reuse(int pos, BTenDev &half, const int *take) {
  require_ready();
  int pre = (pos + 3) % 4, nxt = (pos + 1) % 4;
  int bs = bten_size(pos);
  PG_REQUIRE(bs > 0, 3, "GrowBTenStep: BTen not initialised");
  int n, r, c;
  switch (pos) {
    case DOWN: c = bmps_size(LEFT) - 1; n = Ly_; r = n - bs; break;
    case UP: c = bmps_size(LEFT) - 1; n = Ly_; r = bs - 1; break;
    case LEFT: r = bmps_size(UP) - 1; n = Lx_; c = bs - 1; break;
    default: r = bmps_size(UP) - 1; n = Lx_; c = n - bs; break;
  }
  PG_REQUIRE(bs <= n && !bmps_[pre].empty() && !bmps_[nxt].empty(), 3, "GrowBTenStep: environment missing");
  SiteSel sel = cfg_site(r, c);
  const BMPSDev &b1 = bmps_[pre].back(), &b2 = bmps_[nxt].back();
  auto lv = [](const BMPSDev &b, int j) -> const int * { return (int)b.live.size() > j ? b.live[j] : nullptr; };
  const BTenDev &bt = bten_[pos].back();
  BTenDev nb = bten_step(pos, bt, b1.t[n - bs], sel, b2.t[bs - 1], 1, false, 1, lv(b1, n - bs), lv(b1, n - bs + 1), lv(b2, bs - 1),
                         lv(b2, bs), take);
  try {      // (nb is not owned by anything yet: an error below must not leave it in the arena; `half` stays the caller's until consumed)
    PG_REQUIRE(nb.t.n == half.t.n, 3, "GrowBTenStep: the kept half step has another shape");
    hipLaunchKernelGGL(sweep_take_kernel<T>, dim3((unsigned)std::min<long>(8, (nb.t.n + 255) / 256), nw_), dim3(256), 0, stream_, nb.t.p,
                       (const T *)half.t.p, (long)nb.t.n, take);
    PG_CHECK_HIP(hipGetLastError());
    free_ten(half.t);
    half.t.p = nullptr;
    nb.logscale = (double *)arena_.alloc(sizeof(double) * nw_);
    PG_CHECK_HIP(hipMemcpyAsync(nb.logscale, bt.logscale, sizeof(double) * nw_, hipMemcpyDeviceToDevice, stream_));
    normalize(nb.t.p, nb.t.n, nb.t.n, nw_, nb.logscale);
  } catch (...) {
    if (nb.t.p) arena_.free(nb.t.p);
    if (nb.logscale) arena_.free(nb.logscale);
    throw;
  }
  bten_[pos].push_back(nb);
}

// One slice of a sweep of a two-site updater on the device.  mode 0: the exchange move (pair_table == nullptr: swap of the two
// states; else the tabulated pair, see sweep_swap_cand_kernel) with the Metropolis test; mode 1: the full-space move (Suwa-Todo over
// the phys_dim^2 states of the pair, sweep_suwa_todo_kernel).  amp_inout: [n] amplitudes of the element type (complex: interleaved).
template <typename T>
void Engine<T>::sweep_slice_impl(int mode, int orient, int slice, int n_uniform, const double *uniforms, const int32_t *pair_table,
                                 int phys_dim, const uint32_t *words, double *amp_inout, int32_t *consumed_out, int32_t *accepted_out,
                                 int32_t *slice_states_out) {
  require_ready();
  PG_REQUIRE(orient == HORIZONTAL || orient == VERTICAL, 1, "bad orientation");
  const int N = orient == HORIZONTAL ? Lx_ : Ly_, lim = orient == HORIZONTAL ? Ly_ : Lx_;
  PG_REQUIRE(slice >= 0 && slice < lim, 1, "slice outside the lattice");
  if (mode == 0) PG_REQUIRE(n_uniform >= N - 1 && uniforms, 1, "one uniform deviate per bond of the slice is needed");
  else PG_REQUIRE(phys_dim >= 1 && phys_dim <= dp_ && phys_dim * phys_dim <= SW_MAXC && words, 1, "full-space slice: 1 <= phys_dim, phys_dim^2 <= 16, two engine words per bond");
  const int sites = Ly_ * Lx_;
  const int nc = mode == 0 ? 1 : phys_dim * phys_dim;
  const int nwords = 2 * (N - 1);
  // persistent-for-the-call buffers (outside the ArenaScope'd operations they bracket)
  Acc *damp = (Acc *)arena_.alloc(sizeof(Acc) * nw_);
  double *duni = mode == 0 ? (double *)arena_.alloc(sizeof(double) * (size_t)nw_ * n_uniform) : nullptr;
  unsigned *dwords = mode == 1 ? (unsigned *)arena_.alloc(sizeof(unsigned) * (size_t)nw_ * nwords) : nullptr;
  int *dptr = (int *)arena_.alloc(sizeof(int) * (2 * (size_t)nw_ + 1));
  int *dacc = dptr + nw_, *dover = dptr + 2 * nw_;
  int *dcand = (int *)arena_.alloc(sizeof(int) * (2 * (size_t)nc + 2) * (size_t)nw_);
  int *dsame = dcand + 2 * (size_t)nc * nw_, *dnow = dsame + nw_;
  int *dslice = (int *)arena_.alloc(sizeof(int) * (size_t)nw_ * N);
  int *dtab = (mode == 0 && pair_table) ? (int *)arena_.alloc(sizeof(int) * 2 * (size_t)dp_ * dp_) : nullptr;
  BTenDev half;       // the kept half step of the bond in flight (owned here until grow_bten_step_reuse consumes it)
  half.t.p = nullptr;
  auto release = [&]() {
    if (half.t.p) { arena_.free(half.t.p); half.t.p = nullptr; }
    arena_.free(damp); arena_.free(dptr); arena_.free(dcand); arena_.free(dslice);
    if (duni) arena_.free(duni);
    if (dwords) arena_.free(dwords);
    if (dtab) arena_.free(dtab);
  };
  try {
    PG_CHECK_HIP(hipMemcpyAsync(damp, amp_inout, sizeof(Acc) * nw_, hipMemcpyHostToDevice, stream_));
    if (duni) PG_CHECK_HIP(hipMemcpyAsync(duni, uniforms, sizeof(double) * (size_t)nw_ * n_uniform, hipMemcpyHostToDevice, stream_));
    if (dwords) PG_CHECK_HIP(hipMemcpyAsync(dwords, words, sizeof(unsigned) * (size_t)nw_ * nwords, hipMemcpyHostToDevice, stream_));
    if (dtab) {
      for (int e = 0; e < 2 * dp_ * dp_; ++e) PG_REQUIRE(pair_table[e] >= 0 && pair_table[e] < dp_, 4, "pair table: state out of range");
      PG_CHECK_HIP(hipMemcpyAsync(dtab, pair_table, sizeof(int) * 2 * (size_t)dp_ * dp_, hipMemcpyHostToDevice, stream_));
    }
    PG_CHECK_HIP(hipMemsetAsync(dptr, 0, sizeof(int) * (2 * (size_t)nw_ + 1), stream_));
    const int lo = orient == HORIZONTAL ? LEFT : UP, hi = orient == HORIZONTAL ? RIGHT : DOWN;
    init_bten(lo, slice);
    grow_full_bten(hi, slice, 2, 1);
    const int gb = (nw_ + 255) / 256;
    if (mode == 1) {
      hipLaunchKernelGGL(sweep_all_cand_kernel, dim3((nw_ * nc + 255) / 256), dim3(256), 0, stream_, dcand, phys_dim, nw_);
      PG_CHECK_HIP(hipGetLastError());
    }
    for (int j = 0; j + 1 < N; ++j) {
      const int r1 = orient == HORIZONTAL ? slice : j, c1 = orient == HORIZONTAL ? j : slice;
      const int r2 = orient == HORIZONTAL ? slice : j + 1, c2 = orient == HORIZONTAL ? j + 1 : slice;
      const int s1 = r1 * Lx_ + c1, s2 = r2 * Lx_ + c2;
      double *lsum = nullptr;
      if (mode == 0) {
        hipLaunchKernelGGL(sweep_swap_cand_kernel, dim3(gb), dim3(256), 0, stream_, (const int *)cfg_, sites, s1, s2, dcand, dsame, nw_,
                           (const int *)dtab, dp_);
        PG_CHECK_HIP(hipGetLastError());
        // the left half of the replacement trace IS the next environment tensor of the walkers that accept the exchange (the same
        // kernel on the same operands): it is kept, and the growth step behind the Metropolis test runs for the others only
        constexpr bool no_reuse = false;
        const bool reuse = j + 2 < N && !no_reuse;
        Acc *res = nn_trace_device(r1, c1, orient, 1, dcand, &lsum, dsame, reuse ? &half : nullptr);
        hipLaunchKernelGGL(sweep_metropolis_exchange_kernel<Acc>, dim3(gb), dim3(256), 0, stream_, cfg_, sites, s1, s2, (const Acc *)res,
                           (const double *)lsum, damp, (const double *)duni, n_uniform, dptr, dacc, dover, dnow, nw_, (const int *)dcand,
                           (const int *)dsame);
        PG_CHECK_HIP(hipGetLastError());
        arena_.free(res);
        arena_.free(lsum);
        erase_envs_after_update(r1, c1);
        erase_envs_after_update(r2, c2);
        if (reuse) {
          clear_bten(hi, bten_size(hi) - 1);
          grow_bten_step_reuse(lo, half, dnow);
        } else if (j + 2 < N) {
          shift_bten_window(hi);
        }
      } else {
        Acc *res = nn_trace_device(r1, c1, orient, nc, dcand, &lsum);
        hipLaunchKernelGGL(sweep_suwa_todo_kernel<Acc>, dim3(gb), dim3(256), 0, stream_, cfg_, sites, s1, s2, phys_dim, (const Acc *)res,
                           (const double *)lsum, damp, (const unsigned *)dwords, nwords, j, dacc, dnow, nw_);
        PG_CHECK_HIP(hipGetLastError());
        arena_.free(res);
        arena_.free(lsum);
        erase_envs_after_update(r1, c1);
        erase_envs_after_update(r2, c2);
        if (j + 2 < N) shift_bten_window(hi);
      }
    }
    hipLaunchKernelGGL(sweep_gather_slice_kernel, dim3((nw_ * N + 255) / 256), dim3(256), 0, stream_, (const int *)cfg_, sites,
                       orient == HORIZONTAL ? slice * Lx_ : slice, orient == HORIZONTAL ? 1 : Lx_, N, dslice, nw_);
    PG_CHECK_HIP(hipGetLastError());
    std::vector<int> hs((size_t)nw_ * N), hp(2 * (size_t)nw_ + 1);
    PG_CHECK_HIP(hipMemcpyAsync(amp_inout, damp, sizeof(Acc) * nw_, hipMemcpyDeviceToHost, stream_));
    PG_CHECK_HIP(hipMemcpyAsync(hs.data(), dslice, sizeof(int) * hs.size(), hipMemcpyDeviceToHost, stream_));
    PG_CHECK_HIP(hipMemcpyAsync(hp.data(), dptr, sizeof(int) * hp.size(), hipMemcpyDeviceToHost, stream_));
    PG_CHECK_HIP(hipStreamSynchronize(stream_));
    PG_REQUIRE(hp[2 * (size_t)nw_] == 0, 5, "device-side slice sweep: uniform deviates exhausted");
    for (int w = 0; w < nw_; ++w) {
      if (consumed_out) consumed_out[w] = hp[w];
      accepted_out[w] = hp[nw_ + w];
      for (int j = 0; j < N; ++j) {
        const int v = hs[(size_t)w * N + j];
        const int r = orient == HORIZONTAL ? slice : j, c = orient == HORIZONTAL ? j : slice;
        hcfg_[(size_t)w * sites + r * Lx_ + c] = v;          // host mirror of the configuration table
        if (slice_states_out) slice_states_out[(size_t)w * N + j] = v;
      }
    }
  } catch (...) {
    // the device table may hold moves the host has not seen: bring the mirror back in step before reporting the failure
    (void)hipMemcpy(hcfg_.data(), cfg_, sizeof(int) * (size_t)nw_ * sites, hipMemcpyDeviceToHost);
    release();
    throw;
  }
  release();
}

}  // namespace pepsgpu
