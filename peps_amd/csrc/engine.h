// Walker-batched boundary-MPS engine (device side of the C ABI in include/pepsgpu.h).
//
// All walkers of a context advance in lockstep through the same contractor calls; every tensor
// of the reference's per-walker BMPSContractor (bmps_contractor.h:187-232) becomes one device
// buffer [walker][elements].  The projected TensorNetwork2D (tensor_network_2d_basic_impl.h:24-74)
// is never materialised: kernels pick T[r][c][config[w][r][c]] out of the shared SITPS buffer.
//
// Row absorption (BMPS::MultiplyMPO, bmps_impl.h:756-862 + :225-263) is restated "Q-less":
//   forward   P_i = R_i (A_i x W_i)            (same two contractions as :806-807)
//             R_{i+1}: R^T R = P_i^T P_i       (f64 Gram + Cholesky; equals the R of QR :821)
//   backward  T_i = (A_i x W_i) Y_{i+1},  M_i = R_i T_i  (= Q_i . (u s)_{i+1} of :254)
//             rows of M_i -> Jacobi -> Vt_i (chi largest),  Y_i = T_i Vt_i^T
//   res[i] = Vt_i, res[0] = T_0.  The result equals the reference's up to the bond gauge
//   (validated against the op-for-op oracle); see DESIGN.md.
#pragma once
#include <cmath>
#include <memory>
#include "common.h"
#include "linalg.h"
#include "tgemm.h"
#include "gram.h"
#include "trunc_mid.h"
#include "mgemm_dense.h"
#include "chol_pivot.h"
#include "rows_qr.h"

namespace pepsgpu {

enum { LEFT = 0, DOWN = 1, RIGHT = 2, UP = 3 };      // include/qlpeps/basic.h:58-63
enum { HORIZONTAL = 0, VERTICAL = 1 };               // include/qlpeps/basic.h:19-22

// kernel categories of the event profile (pepsgpu_profile_read)
enum { PROF_CONTRACT = 0, PROF_GRAM = 1, PROF_CHOL = 2, PROF_JACOBI = 3, PROF_SELECT = 4, PROF_NORM = 5,
       PROF_ENV = 6, PROF_JACOBI_EDGE = 7, PROF_TRUNC_GRAM = 8, PROF_TRUNC_APPLY = 9,
       PROF_CHAIN = 10,   // tgemm_chain_kernel alone (one bracket = one launch of that kernel): the per-kernel roofline of bench.py
       PROF_NCAT = 11 };

struct EngineBase {
  virtual ~EngineBase() {}
  std::string last_error;
  int dtype = 0;
  int device_id = 0;   // every C-ABI call makes this the calling thread's current HIP device first (capi.hip CTX_CALL)
  virtual void *stream_handle() = 0;                                        // hipStream_t of every launch of this context
  virtual void grad_device_ptr(void **so, void **seo, long *n_elems) = 0;   // HBM-resident f64 accumulators (grad_reset)
  // --- contractor surface (all walkers in lockstep) ---
  virtual void state_upload(const void *host, int host_dtype) = 0;
  // the flat SITPS buffer in HBM (pepsgpu_bcast_state: broadcast over the communicator instead of N host uploads);
  // state_adopted(): the buffer was written from outside (broadcast): what state_upload does besides the copy
  virtual void state_device_ptr(void **p, size_t *bytes) = 0;
  virtual void state_adopted() = 0;
  virtual void set_configs(int n, const int32_t *cfg) = 0;
  virtual void get_configs(int32_t *out) = 0;
  virtual int n_walkers() const = 0;
  virtual void init_bmps(int pos) = 0;
  virtual int bmps_size(int pos) const = 0;
  virtual int bten_size(int pos) const = 0;
  virtual void grow_bmps_step(int pos) = 0;
  virtual void grow_full_bmps(int pos) = 0;
  virtual void grow_bmps_for_row(int row) = 0;
  virtual void grow_bmps_for_col(int col) = 0;
  virtual void shift_bmps_window(int pos) = 0;
  virtual void delete_inner_bmps(int pos) = 0;
  virtual void bmps_park(int pos, int keep) = 0;
  virtual void bmps_unpark(int pos) = 0;
  virtual void generate_bmps_approach(int pos) = 0;
  virtual void init_bten(int pos, int slice) = 0;
  virtual void grow_full_bten(int pos, int slice, int remain, int init) = 0;
  virtual void grow_bten_step(int pos) = 0;
  virtual void shift_bten_window(int pos) = 0;
  virtual void truncate_bten(int pos, int len) = 0;
  virtual void init_bten2(int pos, int slice) = 0;
  virtual void grow_full_bten2(int pos, int slice, int remain, int init) = 0;
  virtual void grow_bten2_step(int pos, int slice) = 0;
  virtual void shift_bten2_window(int pos, int slice) = 0;
  virtual int bten2_size(int pos) const = 0;
  virtual void replace_nnn_trace(int row, int col, int diag_dir, int orient, int ncand, const int32_t *cand, double *out) = 0;
  virtual void bten2_select_set(int set) = 0;
  virtual void cfg_override_slice(int orient, int num, const int32_t *states) = 0;
  virtual void replace_plaquette_trace(int row, int col, int ncand, const int32_t *cand, int left_set, int right_set, double *out) = 0;
  virtual void replace_tnn_trace(int row, int col, int orient, int ncand, const int32_t *cand, double *out) = 0;
  virtual void replace_sqrt5_trace(int row, int col, int diag_dir, int orient, int ncand, const int32_t *cand, double *out) = 0;
  virtual void trace(int row, int col, int dir, double *out) = 0;
  virtual void replace_nn_trace(int row, int col, int dir, int ncand, const int32_t *cand, double *out) = 0;
  virtual void replace_one_trace(int row, int col, int orient, int ncand, const int32_t *cand, double *out) = 0;
  virtual void punch_hole(int row, int col, int orient, double *out) = 0;
  virtual void update_local(int nsites, const int32_t *sites, const int32_t *new_states, const uint8_t *mask) = 0;
  virtual void erase_envs_after_update(int row, int col) = 0;
  virtual void evaluate_amplitude(double *out) = 0;
  virtual void get_bmps_tensor(int pos, int level, int idx, int *dims, double *out, double *logscale) = 0;
  virtual void sync() = 0;
  // SetTruncateParams (bmps_contractor.h:216): scheme 0 SVD, 1 two-site, 2 one-site variational (bmps.h:31-35)
  virtual void set_truncate_params(int chi_min, int chi_max, double trunc_err, int scheme, double conv_tol, int iter_max) = 0;
  virtual void read_flags(int32_t *out) = 0;
  virtual size_t device_bytes() const = 0;
  virtual void stats(double *out, int n) = 0;
  virtual void grad_reset() = 0;
  virtual void grad_accumulate(const double *psi, const double *eloc, int exact_sum, const int32_t *states = nullptr) = 0;
  virtual void grad_read(double *so, double *seo) = 0;
  virtual void sr_begin(int max_samples) = 0;
  virtual void sr_append(const double *psi) = 0;
  virtual int sr_count() const = 0;
  virtual void sr_sum(double *out) = 0;
  virtual void sr_matvec(const double *v, double mean_dot_v, double scale, double *out) = 0;
  virtual void sr_matvec_cplx(const double *v, double mean_dot_v_re, double mean_dot_v_im, double scale, double *out) = 0;
  virtual void sr_cg_solve(const double *b, const double *x0, double diag_shift, int max_iter, double rel_tol, double abs_tol,
                           int recompute_interval, double ortho_threshold, double *x_out, double *residual_norm,
                           int *iterations, int *reason) = 0;
  virtual void sr_gram(const void *remote_o, const int32_t *remote_cfg, int n_remote, double *out) = 0;
  virtual void sr_weighted_sum(const double *y, double *out) = 0;
  virtual void sr_copy_samples(void *dst_o, int32_t *dst_cfg) = 0;
  // mode 0: exchange move (pair_table nullable), mode 1: full-space move (Suwa-Todo over phys_dim^2 states); engine_sweep.h
  virtual void sweep_slice_impl(int mode, int orient, int slice, int n_uniform, const double *uniforms, const int32_t *pair_table,
                                int phys_dim, const uint32_t *words, double *amp_inout, int32_t *consumed_out, int32_t *accepted_out,
                                int32_t *slice_states_out) = 0;
  virtual void nn_exchange_slice(int orient, int slice, int punch_holes, double *psi_out, double *psi_ex_out) = 0;
  // BMPSWalker (bmps_contractor.h:357-646)
  virtual int walker_create(int pos, int level) = 0;
  virtual int walker_clone(int id) = 0;
  virtual void walker_destroy(int id) = 0;
  virtual void walker_info(int id, int *pos, int *stack, int *lcol, int *rcol) = 0;
  virtual void walker_set_mpo(int id, int num, const int32_t *states, const double *tensors, int n_tensors) = 0;
  virtual void walker_evolve(int id) = 0;
  virtual void walker_evolve_step(int id) = 0;
  virtual void walker_contract_row(int id, int opp_level, double *out) = 0;
  virtual void walker_init_bten(int id, int opp_level, int side, int target_col) = 0;
  virtual void walker_grow_bten_step(int id, int opp_level, int side) = 0;
  virtual void walker_shift_bten_window(int id, int opp_level, int side) = 0;
  virtual void walker_trace(int id, int opp_level, int site_col, int two_site, const int32_t *states, const double *tensors,
                            int n_tensors, double *out) = 0;
  virtual void walker_clear_bten(int id) = 0;
  virtual void walker_get_tensor(int id, int idx, int *dims, double *out, double *logscale) = 0;
  virtual void profile_enable(int on) = 0;
  virtual void profile_read(double *out) = 0;   // [PROF_NCAT = 11][5]: ms, launches, algorithmic flops, executed flops, operand+result bytes
};

template <typename T> struct EinView;

template <typename T>
struct DTen {
  T *p = nullptr;
  int d[4] = {1, 1, 1, 1};
  long n = 0;   // elements per walker (= batch stride)
  int rank = 3;
};

template <typename T>
class Engine : public EngineBase {
 public:
  typedef typename acc64_of<T>::type Acc;              // float64 accumulation type: double, or complex<double>
  static constexpr bool kCplx = is_cplx<T>::value;     // complex element type: scalar results are (re, im) pairs
  static constexpr int kOut = kCplx ? 2 : 1;           // doubles per scalar in every host output array
  Engine(int device, int Ly, int Lx, int D, int dphys, int chi_min, int chi_max, double trunc_err, int max_walkers)
      : Ly_(Ly), Lx_(Lx), D_(D), dp_(dphys), chi_min_(chi_min), chi_(chi_max), trunc_err_(trunc_err),
        maxw_(max_walkers) {
    PG_REQUIRE(Ly >= 2 && Lx >= 2, 1, "lattice must be at least 2x2");
    PG_REQUIRE(D >= 1 && dphys >= 1 && chi_max >= 1 && max_walkers >= 1, 1, "bad dimensions");
    PG_REQUIRE(trunc_err >= 0.0 && trunc_err < 1.0, 1, "trunc_err must be in [0, 1)");
    PG_CHECK_HIP(hipSetDevice(device));
    device_ = device;
    device_id = device;
    PG_CHECK_HIP(hipStreamCreate(&stream_));
    PG_CHECK_HIP(hipStreamCreateWithFlags(&side_stream_, hipStreamNonBlocking));
    PG_CHECK_HIP(hipEventCreateWithFlags(&ev_fork_, hipEventDisableTiming));
    PG_CHECK_HIP(hipEventCreateWithFlags(&ev_join_, hipEventDisableTiming));
    slot_ = (long)D * D * D * D;
    sitps_ = (T *)arena_.alloc(sizeof(T) * slot_ * dp_ * Ly * Lx);
    cfg_ = (int *)arena_.alloc(sizeof(int) * (size_t)maxw_ * Ly * Lx);
    flag_ = (int *)arena_.alloc(sizeof(int) * (size_t)maxw_);
    sweeps_ = (int *)arena_.alloc(sizeof(int) * (size_t)maxw_);
    { const char *e = getenv("PEPSGPU_DEBUG_SWEEPS"); dbg_sweeps_ = e && e[0] == '1'; }
    if constexpr (std::is_same<T, double>::value) {
      // error-budget experiments (scripts/error_budget.py): the float64 engine with ONE intermediate rounded to float32
      // where the float32 engine stores it (letters of PEPSGPU_INJECT_F32: S state, P, R carry, T = Tt, M, V, Y, E environments)
      // and / or with the noise floors of the float32 engine (PEPSGPU_F64_EPS)
      if (const char *e = getenv("PEPSGPU_INJECT_F32"))
        for (; *e; ++e) {
          const char *all = "SPRTMVYE";
          const char *q = strchr(all, *e);
          if (q) inject_ |= 1 << (int)(q - all);
        }
      double eps = 1.1102230246251565e-16;
      if (const char *e = getenv("PEPSGPU_F64_EPS")) eps = atof(e);
      PG_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_eps64_rt), &eps, sizeof(double)));
    }
    if constexpr (std::is_same<T, float>::value) {
      double eps = 5.9604644775390625e-8;
      if (const char *e = getenv("PEPSGPU_F32_EPS")) eps = atof(e);
      PG_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_eps32_rt), &eps, sizeof(double)));
    }
    PG_CHECK_HIP(hipMemsetAsync(flag_, 0, sizeof(int) * (size_t)maxw_, stream_));
    dtype = kCplx ? 3 : (sizeof(T) == 4 ? 0 : 1);
    for (int q = 0; q < 4; ++q) { redo_seen_[q].assign(std::max(Ly_, Lx_) + 1, 0); carry_seen_[q].assign(std::max(Ly_, Lx_) + 1, -1); }
  }
  ~Engine() override {
    (void)hipStreamSynchronize(side_stream_);
    (void)hipStreamSynchronize(stream_);
    arena_.release();
    (void)hipEventDestroy(ev_fork_); (void)hipEventDestroy(ev_join_);
    (void)hipStreamDestroy(side_stream_);
    (void)hipStreamDestroy(stream_);
  }

  // ------------------------------------------------------------------------------------------
  void site_dims(int r, int c, int *dd) const {
    dd[0] = c == 0 ? 1 : D_;
    dd[1] = r == Ly_ - 1 ? 1 : D_;
    dd[2] = c == Lx_ - 1 ? 1 : D_;
    dd[3] = r == 0 ? 1 : D_;
  }
  void site_strides(int r, int c, int *ss) const {
    int dd[4];
    site_dims(r, c, dd);
    ss[3] = 1; ss[2] = dd[3]; ss[1] = dd[2] * dd[3]; ss[0] = dd[1] * dd[2] * dd[3];
  }

  void state_device_ptr(void **p, size_t *bytes) override {
    *p = sitps_;
    *bytes = sizeof(T) * (size_t)slot_ * dp_ * Ly_ * Lx_;
  }
  void state_adopted() override {
    PG_CHECK_HIP(hipStreamSynchronize(stream_));
    for (int q = 0; q < 4; ++q) { redo_seen_[q].assign(std::max(Ly_, Lx_) + 1, 0); carry_seen_[q].assign(std::max(Ly_, Lx_) + 1, -1); }
    have_state_ = true;
  }
  void state_upload(const void *host, int host_dtype) override {
    for (int q = 0; q < 4; ++q) { redo_seen_[q].assign(std::max(Ly_, Lx_) + 1, 0); carry_seen_[q].assign(std::max(Ly_, Lx_) + 1, -1); }
    // host layout [row][col][s][L][D][R][U] zero padded to D^4; stored compact in each slot
    std::vector<T> buf((size_t)slot_ * dp_ * Ly_ * Lx_, T(0));
    for (int r = 0; r < Ly_; ++r)
      for (int c = 0; c < Lx_; ++c) {
        int dd[4];
        site_dims(r, c, dd);
        for (int s = 0; s < dp_; ++s) {
          size_t base = ((size_t)(r * Lx_ + c) * dp_ + s) * slot_;
          size_t o = 0;
          for (int a = 0; a < dd[0]; ++a)
            for (int b = 0; b < dd[1]; ++b)
              for (int cc = 0; cc < dd[2]; ++cc)
                for (int e = 0; e < dd[3]; ++e) {
                  size_t src = base + (((size_t)a * D_ + b) * D_ + cc) * D_ + e;
                  if constexpr (kCplx) {
                    typedef typename real_of<T>::type R;
                    if (host_dtype == 3) buf[base + o++] = T(R(((const double *)host)[2 * src]), R(((const double *)host)[2 * src + 1]));
                    else buf[base + o++] = T(R(host_dtype == 0 ? (double)((const float *)host)[src] : ((const double *)host)[src]));
                  } else {
                    PG_REQUIRE(host_dtype == 0 || host_dtype == 1, 1, "a real context takes float32 / float64 state buffers");
                    double v = host_dtype == 0 ? (double)((const float *)host)[src] : ((const double *)host)[src];
                    if (inject_ & INJ_S) v = (double)(float)v;
                    buf[base + o++] = T(v);
                  }
                }
        }
      }
    PG_CHECK_HIP(hipMemcpyAsync(sitps_, buf.data(), buf.size() * sizeof(T), hipMemcpyHostToDevice, stream_));
    PG_CHECK_HIP(hipStreamSynchronize(stream_));
    have_state_ = true;
  }

  void set_configs(int n, const int32_t *cfg) override {
    PG_REQUIRE(n >= 1 && n <= maxw_, 1, "walker count out of range");
    for (long i = 0; i < (long)n * Ly_ * Lx_; ++i)
      PG_REQUIRE(cfg[i] >= 0 && cfg[i] < dp_, 4, "configuration value exceeds physical dimension");
    nw_ = n;
    hcfg_.assign(cfg, cfg + (size_t)n * Ly_ * Lx_);
    PG_CHECK_HIP(hipMemcpyAsync(cfg_, hcfg_.data(), hcfg_.size() * sizeof(int), hipMemcpyHostToDevice, stream_));
    PG_CHECK_HIP(hipMemsetAsync(flag_, 0, sizeof(int) * (size_t)maxw_, stream_));
    PG_CHECK_HIP(hipStreamSynchronize(stream_));
    // Init(tn): bmps_contractor_init.h:25-32
    walkers_clear();     // a BMPSWalker is a fork of the old configurations' stacks
    for (int p = 0; p < 4; ++p) {
      for (auto &b : parked_[p]) free_bmps(b);
      parked_[p].clear();
      clear_bmps(p, 0);
      clear_bten(p, 0);
      clear_bten2(p, 0);
      init_bmps(p);
    }
    clear_bten2_sets();
  }
  void get_configs(int32_t *out) override { std::copy(hcfg_.begin(), hcfg_.end(), out); }
  int n_walkers() const override { return nw_; }

  // ------------------------------------------------------------------------------------------
  struct BMPSDev {
    std::vector<DTen<T>> t;
    double *logscale = nullptr;
    // live[b][w] (b = 0..N): number of non-zero states of bond b (between tensors b-1 and b) for
    // walker w; the tensors stay zero padded to their static shape.  nullptr = the static dimension.
    std::vector<int *> live;
    // kmax[b] = max over the walkers of live[b] (host copy; -1 = unknown): the next absorption sizes the static
    // shape of its new bond b from it instead of chi
    std::vector<int> kmax;
    // mlmax[i] = max over the walkers of the live carry rows at site i of the absorption that built this BMPS (-1 =
    // unknown).  A performance hint only: the next absorption skips the launches of the mid-rank truncation route at
    // the sites where no walker came near it (the general kernels take whatever was mispredicted).
    std::vector<int> mlmax;
    int depth = 0;   // rows absorbed so far (0 = the vacuum boundary): the carry rank can grow by the factor D per row at first
  };
  struct BTenDev {
    DTen<T> t;
    double *logscale = nullptr;
  };

  int mps_len(int pos) const { return (pos == DOWN || pos == UP) ? Lx_ : Ly_; }
  int bmps_size(int pos) const override { return (int)bmps_[pos].size(); }
  int bten_size(int pos) const override { return (int)bten_[pos].size(); }

  void init_bmps(int pos) override {   // bmps_contractor_init.h:34-70, bmps_impl.h:60-96
    PG_REQUIRE(nw_ > 0, 3, "set walker configurations first");
    PG_REQUIRE(bmps_[pos].empty(), 3, "InitBMPS: stack not empty");
    BMPSDev b;
    int n = mps_len(pos);
    for (int i = 0; i < n; ++i) b.t.push_back(ones3());
    b.live.assign(n + 1, nullptr);
    b.logscale = zeros_f64();
    bmps_[pos].push_back(std::move(b));
  }

  void grow_bmps_step(int pos) override {   // bmps_contractor_grow.h:32-47
    require_ready();
    int existed = bmps_size(pos);
    PG_REQUIRE(existed > 0, 3, "GrowBMPSStep: stack empty");
    int num;
    if (pos == UP || pos == LEFT) num = existed - 1;
    else if (pos == DOWN) num = Ly_ - existed;
    else num = Lx_ - existed;
    int lim = (pos == UP || pos == DOWN) ? Ly_ : Lx_;
    PG_REQUIRE(num >= 0 && num < lim, 3, "GrowBMPSStep: no slice left to absorb");
    absorb(pos, num);
  }
  void grow_full_bmps(int pos) override {   // grow.h:49-86
    int existed = bmps_size(pos);
    PG_REQUIRE(existed > 0, 3, "GrowFullBMPS: stack empty");
    int lim = (pos == UP || pos == DOWN) ? Ly_ : Lx_;
    for (int k = existed; k < lim; ++k) grow_bmps_step(pos);
  }
  void grow_bmps_for_row(int row) override {   // grow.h:88-104
    while (Ly_ - bmps_size(DOWN) > row) grow_bmps_step(DOWN);
    while (bmps_size(UP) - 1 < row) grow_bmps_step(UP);
  }
  void grow_bmps_for_col(int col) override {   // grow.h:106-122
    while (Lx_ - bmps_size(RIGHT) > col) grow_bmps_step(RIGHT);
    while (bmps_size(LEFT) - 1 < col) grow_bmps_step(LEFT);
  }
  void shift_bmps_window(int pos) override {   // grow.h:143-148
    PG_REQUIRE(bmps_size(pos) > 0, 3, "ShiftBMPSWindow: stack empty");
    clear_bmps(pos, bmps_size(pos) - 1);
    grow_bmps_step((pos + 2) % 4);
  }
  void delete_inner_bmps(int pos) override {   // bmps_contractor.h:320-324
    if (bmps_size(pos) > 1) clear_bmps(pos, 1);
  }
  // BMPSWalker support (bmps/impl/bmps_walker.h): the reference forks a BMPS out of a stack and contracts rows against an
  // explicitly named environment of the opposite stack.  Here the stack itself is the walker; parking hides the levels
  // above `keep` (nothing is copied or freed) so that level keep-1 is the top every BTen / trace call sees, unparking
  // drops whatever was grown meanwhile and puts the hidden levels back.
  void bmps_park(int pos, int keep) override {
    PG_REQUIRE(parked_[pos].empty(), 3, "bmps_park: this stack is already parked");
    PG_REQUIRE(keep >= 1 && keep <= bmps_size(pos), 1, "bmps_park: level outside the stack");
    parked_keep_[pos] = keep;
    auto &v = bmps_[pos];
    parked_[pos].assign(std::make_move_iterator(v.begin() + keep), std::make_move_iterator(v.end()));
    v.erase(v.begin() + keep, v.end());
  }
  void bmps_unpark(int pos) override {
    if (parked_[pos].empty()) return;
    PG_REQUIRE(bmps_size(pos) >= parked_keep_[pos], 3, "bmps_unpark: the parked stack was truncated below its park level");
    clear_bmps(pos, parked_keep_[pos]);
    for (auto &b : parked_[pos]) bmps_[pos].push_back(std::move(b));
    parked_[pos].clear();
  }
  void generate_bmps_approach(int pos) override {   // grow.h:11-17
    delete_inner_bmps(pos);
    grow_full_bmps((pos + 2) % 4);
  }

  // ------------------------------------------------------------------------------------------
  void init_bten(int pos, int slice) override {   // init.h:72-120
    require_ready();
    (void)slice;
    clear_bten(pos, 0);
    BTenDev b;
    b.t = ones3();
    b.logscale = zeros_f64();
    bten_[pos].push_back(b);
  }
  void truncate_bten(int pos, int len) override {   // init.h:122-128
    if (bten_size(pos) > len) clear_bten(pos, len);
  }
  // two-row environments and next-nearest / third-neighbour traces: engine_nnn.h
  int bten2_size(int pos) const override { return (int)bten2_[pos].size(); }
  void init_bten2(int pos, int slice) override;
  void grow_full_bten2(int pos, int slice, int remain, int init) override;
  void grow_bten2_step(int pos, int slice) override;
  void shift_bten2_window(int pos, int slice) override;
  void replace_nnn_trace(int row, int col, int diag_dir, int orient, int ncand, const int32_t *cand, double *out) override;
  void bten2_select_set(int set) override;
  void cfg_override_slice(int orient, int num, const int32_t *states) override;
  void replace_plaquette_trace(int row, int col, int ncand, const int32_t *cand, int left_set, int right_set, double *out) override;
  void replace_tnn_trace(int row, int col, int orient, int ncand, const int32_t *cand, double *out) override;
  void replace_sqrt5_trace(int row, int col, int diag_dir, int orient, int ncand, const int32_t *cand, double *out) override;

  const BMPSDev &bmps_at_slice(int pos, int idx) const {   // bmps_contractor.h:985-999
    int k = idx;
    if (pos == DOWN) k = Ly_ - 1 - idx;
    if (pos == RIGHT) k = Lx_ - 1 - idx;
    PG_REQUIRE(k >= 0 && k < (int)bmps_[pos].size(), 3, "BMPS environment not available for this slice");
    return bmps_[pos][k];
  }
  const BTenDev &bten_at_slice(int pos, int idx) const {   // bmps_contractor.h:1003-1009
    int k = idx;
    if (pos == DOWN) k = Ly_ - 1 - idx;
    if (pos == RIGHT) k = Lx_ - 1 - idx;
    PG_REQUIRE(k >= 0 && k < (int)bten_[pos].size(), 3, "BTen environment not available for this slice");
    return bten_[pos][k];
  }
  static const DTen<T> &at_logical(const BMPSDev &b, int pos, int col) {   // bmps.h:214-227
    int n = (int)b.t.size();
    return (pos == UP || pos == RIGHT) ? b.t[n - 1 - col] : b.t[col];
  }

  // live extents (device arrays, nullptr = static) of the first and the third leg of at_logical(b, pos, col)
  static void live_at_logical(const BMPSDev &b, int pos, int col, const int *&first, const int *&third) {
    const int n = (int)b.t.size(), j = (pos == UP || pos == RIGHT) ? n - 1 - col : col;
    first = (int)b.live.size() > j ? b.live[j] : nullptr;
    third = (int)b.live.size() > j + 1 ? b.live[j + 1] : nullptr;
  }

  void grow_full_bten(int pos, int slice, int remain, int init) override {   // grow.h:243-373
    require_ready();
    if (init) init_bten(pos, slice);
    PG_REQUIRE(bten_size(pos) > 0, 3, "GrowFullBTen: BTen not initialised");
    int n = (pos == DOWN || pos == UP) ? Ly_ : Lx_;
    int pre = (pos + 3) % 4, nxt = (pos + 1) % 4;
    const BMPSDev &b1 = bmps_at_slice(pre, slice);
    const BMPSDev &b2 = bmps_at_slice(nxt, slice);
    for (int i = bten_size(pos) - 1; i < n - remain; ++i) {
      int r, c;
      switch (pos) {
        case DOWN: r = n - i - 1; c = slice; break;
        case UP: r = i; c = slice; break;
        case LEFT: r = slice; c = i; break;
        default: r = slice; c = n - i - 1; break;
      }
      SiteSel sel = cfg_site(r, c);
      const int j1 = n - i - 1, j2 = i;
      auto lv = [](const BMPSDev &b, int j) -> const int * { return (int)b.live.size() > j ? b.live[j] : nullptr; };
      BTenDev nb = bten_step(pos, bten_[pos].back(), b1.t[j1], sel, b2.t[j2], 1, true, 1, lv(b1, j1), lv(b1, j1 + 1), lv(b2, j2),
                             lv(b2, j2 + 1));
      bten_[pos].push_back(nb);
    }
  }

  void grow_bten_step(int pos) override {   // grow.h:529-582
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
    BTenDev nb = bten_step(pos, bten_[pos].back(), b1.t[n - bs], sel, b2.t[bs - 1], 1, true, 1, lv(b1, n - bs), lv(b1, n - bs + 1),
                           lv(b2, bs - 1), lv(b2, bs));
    bten_[pos].push_back(nb);
  }
  // GrowBTenStep for the slice sweep (engine_sweep.h): walkers with take[w] != 0 adopt `half` (the un-normalised step tensor the
  // replacement trace already computed with their NEW state of the site), the step runs for the others only; consumes `half`
  void grow_bten_step_reuse(int pos, BTenDev &half, const int *take);
  void shift_bten_window(int pos) override {   // grow.h:517-521
    PG_REQUIRE(bten_size(pos) > 0, 3, "ShiftBTenWindow: BTen empty");
    clear_bten(pos, bten_size(pos) - 1);
    grow_bten_step((pos + 2) % 4);
  }

  // ------------------------------------------------------------------------------------------
  // amplitudes: trace.h:11-28, :90-205, :30-88
  void trace(int row, int col, int dir, double *out) override {
    replace_nn_trace(row, col, dir, 0, nullptr, out);
  }

  void replace_nn_trace(int row, int col, int dir, int ncand, const int32_t *cand, double *out) override {
    require_ready();
    ArenaScope scope(arena_);
    const int nc = ncand > 0 ? ncand : 1;
    int *dcand = nullptr;
    if (ncand > 0) {
      size_t cnt = (size_t)nw_ * ncand * 2;
      for (size_t i = 0; i < cnt; ++i) PG_REQUIRE(cand[i] >= 0 && cand[i] < dp_, 4, "candidate state out of range");
      dcand = (int *)arena_.alloc(cnt * sizeof(int));
      PG_CHECK_HIP(hipMemcpyAsync(dcand, cand, cnt * sizeof(int), hipMemcpyHostToDevice, stream_));
    }
    double *lsum = nullptr;
    Acc *res = nn_trace_device(row, col, dir, nc, dcand, &lsum);
    finish_read(res, nw_ * nc, nc, lsum, out);
    arena_.free(res);
    arena_.free(lsum);
    if (dcand) arena_.free(dcand);
  }

  // ReplaceNNSiteTrace with everything left on the device: candidate table dcand [walker][nc][2] (nullptr: the configurations),
  // result res [walker x nc] (mantissa) and lsum [walker] (log-scale): psi' = res exp(lsum).  Caller frees both.
  // skip (optional, per walker, nc == 1): nonzero = the result of this walker is not needed (its entry of res is undefined)
  // keep_t2 (optional): receives the half-step tensor of the FIRST site (LEFT / UP environment grown over it with the candidate
  // state, not normalised) instead of freeing it -- the caller frees it (free_ten) or hands it to grow_bten_step_reuse
  Acc *nn_trace_device(int row, int col, int dir, int nc, const int *dcand, double **lsum_out, const int *skip = nullptr,
                       BTenDev *keep_t2 = nullptr) {
    int rb = row + (dir == VERTICAL), cb = col + (dir == HORIZONTAL);
    PG_REQUIRE(row >= 0 && col >= 0 && rb < Ly_ && cb < Lx_, 1, "ReplaceNNSiteTrace: bond outside the lattice");
    SiteSel sa = cfg_site(row, col), sb = cfg_site(rb, cb);
    if (dcand) {
      sa.sel = dcand; sa.inc = 2; sa.base = nullptr; sa.per_walker = false;
      sb.sel = dcand + 1; sb.inc = 2; sb.base = nullptr; sb.per_walker = false;
    }
    BTenDev t2, t5;
    double *lsum = zeros_f64();
    if (dir == HORIZONTAL) {
      const BMPSDev &up = bmps_at_slice(UP, row), &dn = bmps_at_slice(DOWN, row);
      PG_REQUIRE(bten_size(LEFT) > col, 3, "ReplaceNNSiteTrace: LEFT BTen missing");
      const int *a1, *a3, *b1, *b3;
      live_at_logical(up, UP, col, a1, a3); live_at_logical(dn, DOWN, col, b1, b3);
      t2 = bten_step(LEFT, bten_[LEFT][col], at_logical(up, UP, col), sa, at_logical(dn, DOWN, col), nc, false, 1, a1, a3, b1, b3, skip);
      live_at_logical(dn, DOWN, cb, a1, a3); live_at_logical(up, UP, cb, b1, b3);
      t5 = bten_step(RIGHT, bten_at_slice(RIGHT, cb), at_logical(dn, DOWN, cb), sb, at_logical(up, UP, cb), nc, false, 1, a1, a3, b1, b3, skip);
      add_logs(lsum, up.logscale, dn.logscale, bten_[LEFT][col].logscale, bten_at_slice(RIGHT, cb).logscale);
    } else {
      const BMPSDev &lf = bmps_at_slice(LEFT, col), &rt = bmps_at_slice(RIGHT, col);
      PG_REQUIRE(bten_size(UP) > row, 3, "ReplaceNNSiteTrace: UP BTen missing");
      const int *a1, *a3, *b1, *b3;
      live_at_logical(rt, RIGHT, row, a1, a3); live_at_logical(lf, LEFT, row, b1, b3);
      t2 = bten_step(UP, bten_[UP][row], at_logical(rt, RIGHT, row), sa, at_logical(lf, LEFT, row), nc, false, 1, a1, a3, b1, b3, skip);
      live_at_logical(lf, LEFT, rb, a1, a3); live_at_logical(rt, RIGHT, rb, b1, b3);
      t5 = bten_step(DOWN, bten_at_slice(DOWN, rb), at_logical(lf, LEFT, rb), sb, at_logical(rt, RIGHT, rb), nc, false, 1, a1, a3, b1, b3, skip);
      add_logs(lsum, lf.logscale, rt.logscale, bten_[UP][row].logscale, bten_at_slice(DOWN, rb).logscale);
    }
    Acc *res = finish_dot_device(t2.t, nc, t5.t, nc, nc);
    if (keep_t2) *keep_t2 = t2; else free_ten(t2.t);
    free_ten(t5.t);
    *lsum_out = lsum;
    return res;
  }

  void replace_one_trace(int row, int col, int orient, int ncand, const int32_t *cand, double *out) override {
    require_ready();
    ArenaScope scope(arena_);
    PG_REQUIRE(row >= 0 && col >= 0 && row < Ly_ && col < Lx_, 1, "ReplaceOneSiteTrace: site outside the lattice");
    const int nc = ncand > 0 ? ncand : 1;
    SiteSel sa = cfg_site(row, col);
    int *dcand = nullptr;
    if (ncand > 0) {
      size_t cnt = (size_t)nw_ * ncand;
      for (size_t i = 0; i < cnt; ++i) PG_REQUIRE(cand[i] >= 0 && cand[i] < dp_, 4, "candidate state out of range");
      dcand = (int *)arena_.alloc(cnt * sizeof(int));
      PG_CHECK_HIP(hipMemcpyAsync(dcand, cand, cnt * sizeof(int), hipMemcpyHostToDevice, stream_));
      sa.sel = dcand; sa.inc = 1; sa.per_walker = false;
    }
    BTenDev t2;
    double *lsum = zeros_f64();
    const DTen<T> *other;
    if (orient == HORIZONTAL) {
      const BMPSDev &up = bmps_at_slice(UP, row), &dn = bmps_at_slice(DOWN, row);
      PG_REQUIRE(bten_size(LEFT) > col, 3, "ReplaceOneSiteTrace: LEFT BTen missing");
      const BTenDev &rb = bten_at_slice(RIGHT, col);
      t2 = bten_step(LEFT, bten_[LEFT][col], at_logical(up, UP, col), sa, at_logical(dn, DOWN, col), nc, false);
      add_logs(lsum, up.logscale, dn.logscale, bten_[LEFT][col].logscale, rb.logscale);
      other = &rb.t;
    } else {
      const BMPSDev &lf = bmps_at_slice(LEFT, col), &rt = bmps_at_slice(RIGHT, col);
      PG_REQUIRE(bten_size(UP) > row, 3, "ReplaceOneSiteTrace: UP BTen missing");
      const BTenDev &db = bten_at_slice(DOWN, row);
      t2 = bten_step(UP, bten_[UP][row], at_logical(rt, RIGHT, row), sa, at_logical(lf, LEFT, row), nc, false);
      add_logs(lsum, lf.logscale, rt.logscale, bten_[UP][row].logscale, db.logscale);
      other = &db.t;
    }
    finish_dot(t2.t, nc, *other, 1, nc, lsum, out);
    free_ten(t2.t);
    arena_.free(lsum);
    if (dcand) arena_.free(dcand);
  }

  void punch_hole(int row, int col, int orient, double *out) override {   // grow.h:150-183
    require_ready();
    if (out == nullptr && !holes_) {   // the resident hole store is persistent: allocated outside the scope below
      holes_ = (T *)arena_.alloc(sizeof(T) * (size_t)maxw_ * Ly_ * Lx_ * slot_);
      holes_ls_ = (double *)arena_.alloc(sizeof(double) * (size_t)maxw_ * Ly_ * Lx_);
    }
    ArenaScope scope(arena_);
    const DTen<T> *left, *down, *right, *up;
    double *lsum = zeros_f64();
    if (orient == HORIZONTAL) {
      const BMPSDev &ub = bmps_at_slice(UP, row), &db = bmps_at_slice(DOWN, row);
      PG_REQUIRE(bten_size(LEFT) > col, 3, "PunchHole: LEFT BTen missing");
      up = &at_logical(ub, UP, col); down = &at_logical(db, DOWN, col);
      left = &bten_[LEFT][col].t; right = &bten_at_slice(RIGHT, col).t;
      add_logs(lsum, ub.logscale, db.logscale, bten_[LEFT][col].logscale, bten_at_slice(RIGHT, col).logscale);
    } else {
      const BMPSDev &lb = bmps_at_slice(LEFT, col), &rb = bmps_at_slice(RIGHT, col);
      PG_REQUIRE(bten_size(UP) > row, 3, "PunchHole: UP BTen missing");
      left = &at_logical(lb, LEFT, row); right = &at_logical(rb, RIGHT, row);
      up = &bten_[UP][row].t; down = &bten_at_slice(DOWN, row).t;
      add_logs(lsum, lb.logscale, rb.logscale, bten_[UP][row].logscale, bten_at_slice(DOWN, row).logscale);
    }
    // tmp1[l0,l1,d1,d2] = sum_c left[l0,l1,c] down[c,d1,d2]
    DTen<T> tmp1 = alloc_ten(left->d[0], left->d[1], down->d[1], down->d[2]);
    {
      TGemmDesc g;
      g.I[2] = left->d[0] * left->d[1]; g.sAi[2] = left->d[2]; g.sCi[2] = down->d[1] * down->d[2];
      g.K[2] = left->d[2]; g.sAk[2] = 1; g.sBk[2] = down->d[1] * down->d[2];
      g.J[2] = down->d[1] * down->d[2]; g.sBj[2] = 1; g.sCj[2] = 1;
      g.wA = left->n; g.wB = down->n; g.wC = tmp1.n; g.nbatch = nw_;
      tgemm_launch<T, T, T, T>(stream_, g, left->p, down->p, tmp1.p);
    }
    DTen<T> tmp2 = alloc_ten(right->d[0], right->d[1], up->d[1], up->d[2]);
    {
      TGemmDesc g;
      g.I[2] = right->d[0] * right->d[1]; g.sAi[2] = right->d[2]; g.sCi[2] = up->d[1] * up->d[2];
      g.K[2] = right->d[2]; g.sAk[2] = 1; g.sBk[2] = up->d[1] * up->d[2];
      g.J[2] = up->d[1] * up->d[2]; g.sBj[2] = 1; g.sCj[2] = 1;
      g.wA = right->n; g.wB = up->n; g.wC = tmp2.n; g.nbatch = nw_;
      tgemm_launch<T, T, T, T>(stream_, g, right->p, up->p, tmp2.p);
    }
    // res[l1,d1,r1,u1] = sum_{l0,d2} tmp1[l0,l1,d1,d2] tmp2[r0=d2, r1, u1, u2=l0]
    int l0 = tmp1.d[0], l1 = tmp1.d[1], d1 = tmp1.d[2], d2 = tmp1.d[3];
    int r0 = tmp2.d[0], r1 = tmp2.d[1], u1 = tmp2.d[2], u2 = tmp2.d[3];
    PG_REQUIRE(l0 == u2 && d2 == r0, 3, "PunchHole: environment bond mismatch");
    DTen<T> res = alloc_ten(l1, d1, r1, u1);
    {
      TGemmDesc g;
      g.I[1] = l1; g.I[2] = d1; g.sAi[1] = d1 * d2; g.sAi[2] = d2; g.sCi[1] = d1 * r1 * u1; g.sCi[2] = r1 * u1;
      g.K[1] = l0; g.K[2] = d2; g.sAk[1] = l1 * d1 * d2; g.sAk[2] = 1; g.sBk[1] = 1; g.sBk[2] = r1 * u1 * u2;
      g.J[1] = r1; g.J[2] = u1; g.sBj[1] = u1 * u2; g.sBj[2] = u2; g.sCj[1] = u1; g.sCj[2] = 1;
      g.wA = tmp1.n; g.wB = tmp2.n; g.wC = res.n; g.nbatch = nw_;
      tgemm_launch<T, T, T, T>(stream_, g, tmp1.p, tmp2.p, res.p);
    }
    if (out == nullptr) {
      // store mode: the hole stays on the device (mantissa in its D^4 slot, compact, + log-scale)
      // for grad_accumulate -- mc_energy_grad_evaluator.h:257-278 without the PCIe round trip
      const int sites = Ly_ * Lx_, site = row * Lx_ + col;
      hipLaunchKernelGGL(store_hole_kernel<T>, dim3((unsigned)((res.n + 255) / 256), nw_), dim3(256), 0, stream_,
                         (const T *)res.p, res.n, holes_ + (long)site * slot_, (long)sites * slot_, (const double *)lsum,
                         holes_ls_ + site, sites);
      PG_CHECK_HIP(hipGetLastError());
      free_ten(tmp1); free_ten(tmp2); free_ten(res);
      arena_.free(lsum);
      return;
    }
    std::vector<T> h((size_t)res.n * nw_);
    std::vector<double> hl(nw_);
    PG_CHECK_HIP(hipMemcpyAsync(h.data(), res.p, h.size() * sizeof(T), hipMemcpyDeviceToHost, stream_));
    PG_CHECK_HIP(hipMemcpyAsync(hl.data(), lsum, nw_ * sizeof(double), hipMemcpyDeviceToHost, stream_));
    PG_CHECK_HIP(hipStreamSynchronize(stream_));
    // output padded to D^4 per walker, leg order (L,D,R,U)
    const long slot = slot_;
    std::fill(out, out + (size_t)slot * nw_ * kOut, 0.0);
    for (int w = 0; w < nw_; ++w) {
      double sc = std::exp(hl[w]);
      size_t o = 0;
      for (int a = 0; a < l1; ++a)
        for (int b = 0; b < d1; ++b)
          for (int c = 0; c < r1; ++c)
            for (int e = 0; e < u1; ++e) {
              const size_t dst = (size_t)w * slot + (((size_t)a * D_ + b) * D_ + c) * D_ + e;
              const T v = h[(size_t)w * res.n + o++];
              if constexpr (kCplx) { out[2 * dst] = (double)v.re * sc; out[2 * dst + 1] = (double)v.im * sc; }
              else out[dst] = (double)v * sc;
            }
    }
    free_ten(tmp1); free_ten(tmp2); free_ten(res);
    arena_.free(lsum);
  }

  // ------------------------------------------------------------------------------------------
  // Device-resident gradient accumulators S_O = sum w O*, S_EO = sum w E_loc O* over everything
  // accumulated since grad_reset(): layout [row][col][s][D^4 slot] (compact inside the slot).
  void grad_reset() override {
    const size_t n = (size_t)Ly_ * Lx_ * dp_ * slot_ * kOut;   // complex: interleaved (re, im) pairs
    if (!so_) {
      so_ = (double *)arena_.alloc(sizeof(double) * n);
      seo_ = (double *)arena_.alloc(sizeof(double) * n);
    }
    PG_CHECK_HIP(hipMemsetAsync(so_, 0, sizeof(double) * n, stream_));
    PG_CHECK_HIP(hipMemsetAsync(seo_, 0, sizeof(double) * n, stream_));
  }
  // psi[w], eloc[w] from the host (the solver's scalars).  MC: O* = hole / psi (weight 1)
  // (mc_energy_grad_evaluator.h:266); exact summation: |psi|^2 O* = psi * hole
  // (exact_summation_energy_evaluator.h:231).
  // states (optional, [walker][row][col]): the component of each site the walker's hole belongs to, when that is not the
  // configuration the device holds at this moment (fermionic states: the extended state of the row-major decoration the
  // holes were punched in, while the walkers may have moved on to the column-major pass since).
  void grad_accumulate(const double *psi, const double *eloc, int exact_sum, const int32_t *states = nullptr) override {
    require_ready();
    if constexpr (kCplx) grad_accumulate_cplx(psi, eloc, exact_sum, states);
    else grad_accumulate_real(psi, eloc, exact_sum, states);
  }
  // Complex element type (psi, eloc = interleaved (re, im) pairs).  The reference stores Dag(hole) (square_nnn_energy_solver.h:163)
  // and accumulates O* = conj(1 / psi) Dag(hole) with weight 1 (mc_energy_grad_evaluator.h:245-278) or psi Dag(hole) under the
  // exact-summation weight (exact_summation_energy_evaluator.h:228-240), and E_loc^* O*.  Both factors have the phase
  // psi / |psi|: v = conj(hole) (psi / |psi|) |psi|^(-1 or +1);  S_O += v,  S_EO += conj(E_loc) v.
  void grad_accumulate_cplx(const double *psi, const double *eloc, int exact_sum, const int32_t *states) {
    PG_REQUIRE(holes_ != nullptr, 3, "grad_accumulate: no holes stored (pepsgpu_punch_hole with out == NULL)");
    if (!so_) grad_reset();
    std::vector<double> h(5 * (size_t)nw_);
    for (int w = 0; w < nw_; ++w) {
      const double a = std::hypot(psi[2 * w], psi[2 * w + 1]);
      PG_REQUIRE(a != 0.0, 5, "Wavefunction amplitude is near zero, causing division by zero.");
      h[w] = (exact_sum ? 1.0 : -1.0) * std::log(a);
      h[nw_ + w] = psi[2 * w] / a;
      h[2 * nw_ + w] = psi[2 * w + 1] / a;
      h[3 * nw_ + w] = eloc[2 * w];
      h[4 * nw_ + w] = -eloc[2 * w + 1];          // conj(E_loc)
    }
    double *d = (double *)arena_.alloc(sizeof(double) * h.size());
    PG_CHECK_HIP(hipMemcpyAsync(d, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice, stream_));
    PG_CHECK_HIP(hipStreamSynchronize(stream_));
    const int sites = Ly_ * Lx_;
    int *dstates = nullptr;
    if (states) {
      for (size_t q = 0; q < (size_t)nw_ * sites; ++q)
        PG_REQUIRE(states[q] >= 0 && states[q] < dp_, 1, "grad_accumulate: state index out of range");
      dstates = (int *)arena_.alloc(sizeof(int) * (size_t)nw_ * sites);
      PG_CHECK_HIP(hipMemcpyAsync(dstates, states, sizeof(int) * (size_t)nw_ * sites, hipMemcpyHostToDevice, stream_));
      PG_CHECK_HIP(hipStreamSynchronize(stream_));
    }
    if constexpr (kCplx) {
      hipLaunchKernelGGL(grad_accumulate_cplx_kernel<T>, dim3((unsigned)((slot_ + 255) / 256), sites), dim3(256), 0, stream_,
                         (const T *)holes_, (const double *)holes_ls_, (const int *)(dstates ? dstates : cfg_), (const double *)d,
                         (const double *)(d + nw_), (const double *)(d + 2 * nw_), (const double *)(d + 3 * nw_),
                         (const double *)(d + 4 * nw_), so_, seo_, nw_, sites, slot_, dp_);
      PG_CHECK_HIP(hipGetLastError());
    }
    arena_.free(d);
    if (dstates) arena_.free(dstates);
  }
  void grad_accumulate_real(const double *psi, const double *eloc, int exact_sum, const int32_t *states) {
    PG_REQUIRE(holes_ != nullptr, 3, "grad_accumulate: no holes stored (pepsgpu_punch_hole with out == NULL)");
    if (!so_) grad_reset();
    std::vector<double> h(3 * (size_t)nw_);
    for (int w = 0; w < nw_; ++w) {
      PG_REQUIRE(psi[w] != 0.0, 5, "Wavefunction amplitude is near zero, causing division by zero.");
      h[w] = (exact_sum ? 1.0 : -1.0) * std::log(std::fabs(psi[w]));
      h[nw_ + w] = psi[w] < 0 ? -1.0 : 1.0;
      h[2 * nw_ + w] = eloc[w];
    }
    double *d = (double *)arena_.alloc(sizeof(double) * h.size());
    PG_CHECK_HIP(hipMemcpyAsync(d, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice, stream_));
    PG_CHECK_HIP(hipStreamSynchronize(stream_));
    const int sites = Ly_ * Lx_;
    int *dstates = nullptr;
    if (states) {
      for (size_t q = 0; q < (size_t)nw_ * sites; ++q)
        PG_REQUIRE(states[q] >= 0 && states[q] < dp_, 1, "grad_accumulate: state index out of range");
      dstates = (int *)arena_.alloc(sizeof(int) * (size_t)nw_ * sites);
      PG_CHECK_HIP(hipMemcpyAsync(dstates, states, sizeof(int) * (size_t)nw_ * sites, hipMemcpyHostToDevice, stream_));
      PG_CHECK_HIP(hipStreamSynchronize(stream_));
    }
    hipLaunchKernelGGL(grad_accumulate_kernel<T>, dim3((unsigned)((slot_ + 255) / 256), sites), dim3(256), 0, stream_,
                       (const T *)holes_, (const double *)holes_ls_, (const int *)(dstates ? dstates : cfg_), (const double *)d,
                       (const double *)(d + nw_), (const double *)(d + 2 * nw_), so_, seo_, nw_, sites, slot_, dp_);
    PG_CHECK_HIP(hipGetLastError());
    arena_.free(d);
    if (dstates) arena_.free(dstates);
  }
  void *stream_handle() override { return (void *)stream_; }
  void grad_device_ptr(void **so, void **seo, long *n_elems) override {
    if (!so_) grad_reset();
    *so = so_; *seo = seo_;
    *n_elems = (long)Ly_ * Lx_ * dp_ * slot_ * kOut;   // doubles (complex: interleaved pairs): what an all-reduce sums
  }
  // ---- stochastic-reconfiguration sample store (engine_sr.h) ----
  void sr_begin(int max_samples) override;
  void sr_append(const double *psi) override;
  int sr_count() const override { return sr_n_; }
  void sr_sum(double *out) override;
  void sr_matvec(const double *v, double mean_dot_v, double scale, double *out) override;
  void sr_matvec_cplx(const double *v, double mean_dot_v_re, double mean_dot_v_im, double scale, double *out) override;
  void sr_cg_solve(const double *b, const double *x0, double diag_shift, int max_iter, double rel_tol, double abs_tol,
                   int recompute_interval, double ortho_threshold, double *x_out, double *residual_norm, int *iterations,
                   int *reason) override;
  void sr_gram(const void *remote_o, const int32_t *remote_cfg, int n_remote, double *out) override;
  void sr_weighted_sum(const double *y, double *out) override;
  void sr_copy_samples(void *dst_o, int32_t *dst_cfg) override;
  void sr_release();
  void sr_convert(const double *src, double *dst, bool to_compact) const;

  // out layout = state upload layout [row][col][s][L][D][R][U] zero padded to D
  void grad_read(double *so, double *seo) override {
    PG_REQUIRE(so_ != nullptr, 3, "grad_read: nothing accumulated");
    const size_t n = (size_t)Ly_ * Lx_ * dp_ * slot_;
    std::vector<double> h(n * kOut);
    for (int pass = 0; pass < 2; ++pass) {
      double *dst = pass ? seo : so;
      PG_CHECK_HIP(hipMemcpyAsync(h.data(), pass ? seo_ : so_, n * kOut * sizeof(double), hipMemcpyDeviceToHost, stream_));
      PG_CHECK_HIP(hipStreamSynchronize(stream_));
      std::fill(dst, dst + n * kOut, 0.0);
      for (int r = 0; r < Ly_; ++r)
        for (int c = 0; c < Lx_; ++c) {
          int dd[4];
          site_dims(r, c, dd);
          for (int s = 0; s < dp_; ++s) {
            const size_t base = ((size_t)(r * Lx_ + c) * dp_ + s) * slot_;
            size_t o = 0;
            for (int a = 0; a < dd[0]; ++a)
              for (int b = 0; b < dd[1]; ++b)
                for (int cc = 0; cc < dd[2]; ++cc)
                  for (int e = 0; e < dd[3]; ++e) {
                    const size_t q = base + (((size_t)a * D_ + b) * D_ + cc) * D_ + e;
                    for (int z = 0; z < kOut; ++z) dst[kOut * q + z] = h[kOut * (base + o) + z];
                    ++o;
                  }
          }
        }
    }
  }

  // ------------------------------------------------------------------------------------------
  void erase_envs_after_update(int row, int col) override {   // trace.h:538-589
    // (an update while the twisted set is selected: the TRUE environments are the inactive ones -- back to set 0 first, so that what
    // is freed below is the scratch set, and the slice override, stale for the updated slice, goes with it; pepsgpu.h says so)
    if (bten2_active_ != 0) bten2_select_set(0);
    if (cfg_ovr_tab_) { arena_.free(cfg_ovr_tab_); cfg_ovr_tab_ = nullptr; ovr_on_ = false; ovr_cfg_ = nullptr; }
    if (bmps_size(LEFT) > col + 1) clear_bmps(LEFT, col + 1);
    if (bmps_size(UP) > row + 1) clear_bmps(UP, row + 1);
    if (bmps_size(DOWN) > Ly_ - row) clear_bmps(DOWN, Ly_ - row);
    if (bmps_size(RIGHT) > Lx_ - col) clear_bmps(RIGHT, Lx_ - col);
    if (bten_size(LEFT) > col + 1) clear_bten(LEFT, col + 1);
    if (bten_size(UP) > row + 1) clear_bten(UP, row + 1);
    if (bten_size(RIGHT) > Lx_ - col) clear_bten(RIGHT, Lx_ - col);
    if (bten_size(DOWN) > Ly_ - row) clear_bten(DOWN, Ly_ - row);
    if (bten2_size(LEFT) > col + 1) clear_bten2(LEFT, col + 1);
    if (bten2_size(UP) > row + 1) clear_bten2(UP, row + 1);
    if (bten2_size(RIGHT) > Lx_ - col) clear_bten2(RIGHT, Lx_ - col);
    if (bten2_size(DOWN) > Ly_ - row) clear_bten2(DOWN, Ly_ - row);
    for (int p = 0; p < 4; ++p) {     // (the second set is scratch of one row-pair traversal: any update drops it)
      for (auto &b : bten2_inactive_[p]) { arena_.free(b.t.p); arena_.free(b.logscale); }
      bten2_inactive_[p].clear();
    }
  }

  // wave_function_component.h:345-378 for the walkers with mask[w] != 0 (all walkers share the
  // cache bookkeeping: an environment is dropped if ANY walker changed the site it crosses).
  void update_local(int nsites, const int32_t *sites, const int32_t *new_states, const uint8_t *mask) override {
    require_ready();
    // validate everything first: a failure must leave the host and device configurations untouched and in step
    for (int k = 0; k < nsites; ++k)
      PG_REQUIRE(sites[2 * k] >= 0 && sites[2 * k] < Ly_ && sites[2 * k + 1] >= 0 && sites[2 * k + 1] < Lx_, 1,
                 "UpdateLocal: site outside the lattice");
    for (int w = 0; w < nw_; ++w) {
      if (mask && !mask[w]) continue;
      for (int k = 0; k < nsites; ++k) {
        const int s = new_states[(size_t)w * nsites + k];
        PG_REQUIRE(s >= 0 && s < dp_, 4, "UpdateLocal: configuration value exceeds physical dimension");
      }
    }
    bool any = false;
    for (int w = 0; w < nw_; ++w) {
      if (mask && !mask[w]) continue;
      for (int k = 0; k < nsites; ++k) {
        hcfg_[(size_t)w * Ly_ * Lx_ + sites[2 * k] * Lx_ + sites[2 * k + 1]] = new_states[(size_t)w * nsites + k];
        any = true;
      }
    }
    if (!any) return;
    PG_CHECK_HIP(hipMemcpyAsync(cfg_, hcfg_.data(), hcfg_.size() * sizeof(int), hipMemcpyHostToDevice, stream_));
    PG_CHECK_HIP(hipStreamSynchronize(stream_));
    for (int k = 0; k < nsites; ++k) erase_envs_after_update(sites[2 * k], sites[2 * k + 1]);
  }

  void evaluate_amplitude(double *out) override {   // wave_function_component.h:187-212
    grow_bmps_for_row(0);
    grow_full_bten(RIGHT, 0, 2, 1);
    init_bten(LEFT, 0);
    trace(0, 0, HORIZONTAL, out);
  }

  void get_bmps_tensor(int pos, int level, int idx, int *dims, double *out, double *logscale) override {
    PG_REQUIRE(level >= 0 && level < bmps_size(pos), 1, "BMPS level out of range");
    const BMPSDev &b = bmps_[pos][level];
    PG_REQUIRE(idx >= 0 && idx < (int)b.t.size(), 1, "BMPS tensor index out of range");
    const DTen<T> &t = b.t[idx];
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
      PG_CHECK_HIP(hipMemcpyAsync(logscale, b.logscale, nw_ * sizeof(double), hipMemcpyDeviceToHost, stream_));
      PG_CHECK_HIP(hipStreamSynchronize(stream_));
    }
  }

  void sync() override { PG_CHECK_HIP(hipStreamSynchronize(stream_)); }
  void set_truncate_params(int chi_min, int chi_max, double trunc_err, int scheme, double conv_tol, int iter_max) override {
    PG_REQUIRE(chi_min >= 0 && chi_max >= 1 && chi_min <= chi_max, 1, "D_min > D_max");
    PG_REQUIRE(trunc_err >= 0.0 && trunc_err < 1.0, 1, "trunc_err must be in [0, 1)");
    PG_REQUIRE(scheme >= 0 && scheme <= 2, 1, "unknown CompressMPSScheme");
    PG_REQUIRE(scheme == 0 || (iter_max >= 1 && conv_tol >= 0.0), 1,
               "variational compression needs convergence_tol and iter_max (bmps.h:81-97)");
    chi_min_ = chi_min; chi_ = chi_max; trunc_err_ = trunc_err;
    scheme_ = scheme; conv_tol_ = conv_tol; iter_max_ = iter_max;
    for (int q = 0; q < 4; ++q) { redo_seen_[q].assign(std::max(Ly_, Lx_) + 1, 0); carry_seen_[q].assign(std::max(Ly_, Lx_) + 1, -1); }   // the routing hints belong to the old parameters
  }
  void read_flags(int32_t *out) override {
    PG_CHECK_HIP(hipMemcpyAsync(out, flag_, nw_ * sizeof(int), hipMemcpyDeviceToHost, stream_));
    PG_CHECK_HIP(hipStreamSynchronize(stream_));
  }
  size_t device_bytes() const override { return arena_.total_bytes(); }
  void stats(double *out, int n) override {
    double v[9] = {(double)n_absorb_, (double)n_jacobi_, (double)jacobi_sweeps_sum_, (double)arena_.total_bytes(),
                   (double)jacobi_sweeps_max_, live_sum_, live_full_, (double)live_max_, (double)n_redo_};
    for (int i = 0; i < n && i < 9; ++i) out[i] = v[i];
  }
  hipStream_t stream() const { return stream_; }
  // ---- Monte-Carlo sweep of one row / column of bonds on the device (engine_sweep.h) ----
  void sweep_slice_impl(int mode, int orient, int slice, int n_uniform, const double *uniforms, const int32_t *pair_table, int phys_dim,
                        const uint32_t *words, double *amp_inout, int32_t *consumed_out, int32_t *accepted_out,
                        int32_t *slice_states_out) override;
  void nn_exchange_slice(int orient, int slice, int punch_holes, double *psi_out, double *psi_ex_out) override;
  // ---- BMPSWalker (engine_walker.h) ----
  int walker_create(int pos, int level) override;
  int walker_clone(int id) override;
  void walker_destroy(int id) override;
  void walker_info(int id, int *pos, int *stack, int *lcol, int *rcol) override;
  void walker_set_mpo(int id, int num, const int32_t *states, const double *tensors, int n_tensors) override;
  void walker_evolve(int id) override;
  void walker_evolve_step(int id) override;
  void walker_contract_row(int id, int opp_level, double *out) override;
  void walker_init_bten(int id, int opp_level, int side, int target_col) override;
  void walker_grow_bten_step(int id, int opp_level, int side) override;
  void walker_shift_bten_window(int id, int opp_level, int side) override;
  void walker_trace(int id, int opp_level, int site_col, int two_site, const int32_t *states, const double *tensors, int n_tensors,
                    double *out) override;
  void walker_clear_bten(int id) override;
  void walker_get_tensor(int id, int idx, int *dims, double *out, double *logscale) override;

  // ---- per-kernel timing with HIP events on the launch stream (bench.py roofline leg) ----
  void profile_enable(int on) override {
    prof_resolve();
    prof_on_ = on != 0;
    if (prof_on_ && !flopc_) {   // persistent: allocated here, never inside an ArenaScope'd operation
      flopc_ = (unsigned long long *)arena_.alloc(sizeof(unsigned long long) * 2 * PROF_NCAT);
      PG_CHECK_HIP(hipMemsetAsync(flopc_, 0, sizeof(unsigned long long) * 2 * PROF_NCAT, stream_));
    }
  }
  void profile_read(double *out) override {
    prof_resolve();
    // flops the tensor GEMMs of each category actually contracted (live extents), counted on the device
    unsigned long long hc[2 * PROF_NCAT] = {0};
    if (flopc_) {
      PG_CHECK_HIP(hipMemcpyAsync(hc, flopc_, sizeof(hc), hipMemcpyDeviceToHost, stream_));
      PG_CHECK_HIP(hipMemsetAsync(flopc_, 0, sizeof(hc), stream_));
      PG_CHECK_HIP(hipStreamSynchronize(stream_));
    }
    for (int c = 0; c < PROF_NCAT; ++c) {
      out[5 * c + 0] = prof_ms_[c]; out[5 * c + 1] = (double)prof_n_[c];
      out[5 * c + 2] = prof_alg_[c]; out[5 * c + 3] = hc[c] ? (double)hc[c] : prof_exec_[c];
      out[5 * c + 4] = (double)hc[PROF_NCAT + c];
      prof_ms_[c] = 0; prof_n_[c] = 0; prof_alg_[c] = 0; prof_exec_[c] = 0;
    }
  }
  void prof_begin(int cat, double alg_flops, double exec_flops) {
    if (!prof_on_) return;
    ProfRec r;
    // A bracket that follows another one directly starts at that one's end event (an event costs ~2.3 us on the stream: 4 000 of
    // them were 9 ms of the 441 ms headline step): the few unbracketed operations in between (flag memsets, list kernels) are
    // charged to the bracket that follows.  Not for the chained contraction, whose pair brackets exactly its launch (roofline).
    constexpr bool share = true;
    r.a_shared = share && prof_chain_ok_ && cat != PROF_CHAIN && !prof_.empty();
    if (r.a_shared) r.a = prof_.back().b;
    else if (!ev_pool_.empty()) { r.a = ev_pool_.back(); ev_pool_.pop_back(); } else PG_CHECK_HIP(hipEventCreate(&r.a));
    if (!ev_pool_.empty()) { r.b = ev_pool_.back(); ev_pool_.pop_back(); } else PG_CHECK_HIP(hipEventCreate(&r.b));
    r.cat = cat; r.alg = alg_flops; r.exec = exec_flops;
    if (!r.a_shared) PG_CHECK_HIP(hipEventRecord(r.a, stream_));
    prof_chain_ok_ = false;
    prof_.push_back(r);
    tg_flop_counter = flopc_ + cat;
    tg_byte_counter = flopc_ + PROF_NCAT + cat;
  }
  void prof_end() {
    tg_flop_counter = nullptr;
    tg_byte_counter = nullptr;
    if (!prof_on_) return;
    PG_CHECK_HIP(hipEventRecord(prof_.back().b, stream_));
    prof_chain_ok_ = true;
  }
  void prof_resolve() {
    if (prof_.empty()) return;
    PG_CHECK_HIP(hipStreamSynchronize(stream_));
    for (auto &r : prof_) {
      float ms = 0.f;
      PG_CHECK_HIP(hipEventElapsedTime(&ms, r.a, r.b));
      prof_ms_[r.cat] += ms; prof_n_[r.cat] += 1; prof_alg_[r.cat] += r.alg; prof_exec_[r.cat] += r.exec;
      if (!r.a_shared) ev_pool_.push_back(r.a);
      ev_pool_.push_back(r.b);
    }
    prof_.clear();
    prof_chain_ok_ = false;
  }

 private:
  struct SiteSel {
    int r, c;
    const int *sel;   // device selector (configuration or candidate table)
    int inc;          // selector stride per batch entry
    const T *base = nullptr;   // tensor store the selector indexes (nullptr: the SITPS slot of site (r, c))
    // a configuration table is indexed by WALKER (batch entry / candidates), a candidate table by batch entry.  (Until round 5 this
    // was read off inc == Ly * Lx, which a four-column candidate table on a 2 x 2 lattice also satisfies.)
    bool per_walker = false;
  };
  // The site tensor of (r, c) under the walkers' configurations -- or, while a BMPSWalker operation runs (MpoScope,
  // engine_walker.h), under the walker's MPO on its slice: another configuration table, or explicit tensors.
  SiteSel cfg_site(int r, int c) const {
    if (ovr_on_ && (ovr_hor_ ? r : c) == ovr_num_) {
      if (ovr_tens_) {
        SiteSel s{r, c, ovr_nt_ == 1 ? iota_ : iota_ + 1, ovr_nt_ == 1 ? 0 : 1};
        s.base = ovr_tens_ + (long)(ovr_hor_ ? c : r) * ovr_nt_ * slot_;
        return s;
      }
      if (ovr_cfg_) { SiteSel s{r, c, ovr_cfg_ + r * Lx_ + c, Ly_ * Lx_}; s.per_walker = true; return s; }
    }
    SiteSel s{r, c, cfg_ + r * Lx_ + c, Ly_ * Lx_};
    s.per_walker = true;
    return s;
  }
  const T *site_base(int r, int c) const { return sitps_ + (long)(r * Lx_ + c) * dp_ * slot_; }
  const T *sel_base(const SiteSel &ss) const { return ss.base ? ss.base : site_base(ss.r, ss.c); }

  enum { INJ_S = 1, INJ_P = 2, INJ_R = 4, INJ_T = 8, INJ_M = 16, INJ_V = 32, INJ_Y = 64, INJ_E = 128 };
  // float64 engine only: round a stored intermediate to float32 (error budget by stage; no-op unless PEPSGPU_INJECT_F32 names it)
  void inject(int bit, T *p, long n_per_walker, int nb = -1) {
    if constexpr (std::is_same<T, double>::value) {
      if (!(inject_ & bit) || !p) return;
      const long n = n_per_walker * (nb < 0 ? nw_ : nb);
      hipLaunchKernelGGL(round_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream_, p, n);
    }
  }

  void require_ready() const {
    prof_chain_ok_ = false;   // (entry of an API call: a bracket does not reach back into the previous call)
    PG_REQUIRE(have_state_, 3, "no state uploaded (pepsgpu_state_upload)");
    PG_REQUIRE(nw_ > 0, 3, "no walker configurations set (pepsgpu_walkers_set_configs)");
  }

  DTen<T> alloc_ten(int d0, int d1, int d2, int d3 = 1, int nb = -1) {
    DTen<T> t;
    t.d[0] = d0; t.d[1] = d1; t.d[2] = d2; t.d[3] = d3;
    t.n = (long)d0 * d1 * d2 * d3;
    t.p = (T *)arena_.alloc(sizeof(T) * (size_t)t.n * (nb < 0 ? nw_ : nb));
    return t;
  }
  void free_ten(DTen<T> &t) { arena_.free(t.p); t.p = nullptr; }
  DTen<T> ones3() {
    DTen<T> t = alloc_ten(1, 1, 1);
    hipLaunchKernelGGL(fill_kernel<T>, dim3((nw_ + 255) / 256), dim3(256), 0, stream_, t.p, (long)nw_, T(1));
    return t;
  }
  double *zeros_f64() {
    double *p = (double *)arena_.alloc(sizeof(double) * nw_);
    PG_CHECK_HIP(hipMemsetAsync(p, 0, sizeof(double) * nw_, stream_));
    return p;
  }
  void free_bmps(BMPSDev &b) {
    for (auto &t : b.t) if (t.p) arena_.free(t.p);
    for (int *l : b.live) if (l) arena_.free(l);
    if (b.logscale) arena_.free(b.logscale);
    b.t.clear(); b.live.clear(); b.logscale = nullptr;
  }
  void clear_bmps(int pos, int keep) {
    auto &v = bmps_[pos];
    while ((int)v.size() > keep) {
      free_bmps(v.back());
      v.pop_back();
    }
  }
  void clear_bten(int pos, int keep) {
    auto &v = bten_[pos];
    while ((int)v.size() > keep) {
      arena_.free(v.back().t.p);
      arena_.free(v.back().logscale);
      v.pop_back();
    }
  }

  void normalize(T *x, long n, long stride, int nb, double *logscale, const int *ndyn = nullptr, int ndyn_mul = 1) {
    constexpr bool no_wave = false;
    if (n <= 4096 && nb >= 64 && !no_wave)   // short tensors, many walkers: one wave per walker
      hipLaunchKernelGGL(normalize_wave_kernel<T>, dim3((nb + 3) / 4), dim3(256), 0, stream_, x, stride, (int)n, logscale, flag_,
                         ndyn, ndyn_mul, nb);
    else
      hipLaunchKernelGGL(normalize_kernel<T>, dim3(nb), dim3(256), 0, stream_, x, stride, (int)n, logscale, flag_, ndyn,
                         ndyn_mul);
  }

  // acc[w] += a[w] + b[w] + c[w] + d[w]
  void add_logs(double *acc, const double *a, const double *b, const double *c, const double *d);
  void add_log(double *acc, const double *a);

  // res[(w,cand)] = sum t2[a,b,c] t5[c,b,a]  (device; caller frees)
  Acc *finish_dot_device(const DTen<T> &t2, int nc2, const DTen<T> &t5, int nc5, int nc) {
    PG_REQUIRE(t2.d[0] == t5.d[2] && t2.d[1] == t5.d[1] && t2.d[2] == t5.d[0], 3, "trace: environment bond mismatch");
    const int nb = nw_ * nc;
    Acc *res = (Acc *)arena_.alloc(sizeof(Acc) * nb);
    // dedicated dot kernel (linalg.h: trace_dot_kernel; the generic tensor GEMM ran this 1 x 1 output at 1.36 ms per call: HISTORY.md)
    const size_t bytes = sizeof(T) * (size_t)t5.n;
    const int use_lds = bytes <= 64 * 1024;
    if (use_lds) allow_dynamic_lds(reinterpret_cast<const void *>(&trace_dot_kernel<T, Acc>), bytes);
    hipLaunchKernelGGL((trace_dot_kernel<T, Acc>), dim3(nb), dim3(256), use_lds ? bytes : 0, stream_, (const T *)t2.p, t2.n, nc / nc2,
                       (const T *)t5.p, t5.n, nc / nc5, t2.d[0], t2.d[1], t2.d[2], res, use_lds);
    PG_CHECK_HIP(hipGetLastError());
    return res;
  }
  // out[i] = res[i] * exp(lsum[i / nc])  (to the host)
  void finish_read(const Acc *res, int nb, int nc, const double *lsum, double *out) {
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
  }
  // out[(w,cand)] = sum t2[a,b,c] t5[c,b,a] * exp(lsum[w])
  void finish_dot(const DTen<T> &t2, int nc2, const DTen<T> &t5, int nc5, int nc, double *lsum, double *out) {
    Acc *res = finish_dot_device(t2, nc2, t5, nc5, nc);
    finish_read(res, nw_ * nc, nc, lsum, out);
    arena_.free(res);
  }

  // One BTen growth step with an explicit site selector (grow.h:577-579 and the half-steps of
  // trace.h:129-131 / :149-156): out[x, s_opp, y] from bten[c,b1,b2], mps1[x,p1,c], site, mps2[b2,s1,y].
  // Batch = walker x ncand (environment tensors shared by the candidates of one walker).
  // bt_ncand: the input BTen is batched over walker x bt_ncand candidates (chains of steps with
  // replaced tensors, ReplaceTNNSiteTrace); 1 = one BTen per walker.
  // vx / vc / vb / vy (optional, per walker): live extents of mps1's bonds x, c and of mps2's bonds b2, y.  The contractions
  // then run over the live parts only (the tensors are zero beyond them); the new BTen is written in full, zeros included.
  BTenDev bten_step(int post, const BTenDev &bt, const DTen<T> &mps1, const SiteSel &ss, const DTen<T> &mps2,
                    int ncand, bool normalise, int bt_ncand = 1, const int *vx = nullptr, const int *vc = nullptr,
                    const int *vb = nullptr, const int *vy = nullptr, const int *skip = nullptr) {
    ArenaScope scope(arena_);
    constexpr bool no_live_env = false;
    if (no_live_env || sizeof(T) != 4) vx = vc = vb = vy = nullptr;
    const int nb = nw_ * ncand, nb1 = nw_ * bt_ncand;
    PG_REQUIRE(ncand % bt_ncand == 0 && (!normalise || ncand == 1), 1, "BTen step: bad candidate batching");
    int dd[4], st[4];
    site_dims(ss.r, ss.c, dd);
    site_strides(ss.r, ss.c, st);
    const int lc = (post + 3) % 4, lb = post, l1 = (post + 1) % 4, l2 = (post + 2) % 4;
    const int x = mps1.d[0], p1 = mps1.d[1], cdim = mps1.d[2];
    const int b1 = bt.t.d[1], b2 = bt.t.d[2];
    const int s1 = dd[l1], s2 = dd[l2], y = mps2.d[2];
    PG_REQUIRE(cdim == bt.t.d[0] && p1 == dd[lc] && b1 == dd[lb] && mps2.d[0] == b2 && mps2.d[1] == s1, 3,
               "BTen step: bond dimension mismatch between environment tensors");
    // f32, one candidate per walker (growth steps, the half steps of the sweeps / energy slices): the first two contractions
    // run as ONE chained launch with tmp1 resident in LDS (tgemm_chain_kernel, as the absorption's X -> P pair): one launch and
    // the HBM round trip of tmp1 less per step; entries whose live tmp1 does not fit take the separate launches below
    int *bt_chain_flag = nullptr;
    int bt_chained = 0;
    DTen<T> tmp2c;
    if constexpr (sizeof(T) == 4) {
      constexpr bool no_btc = false;
      if (!no_btc && ncand == 1 && bt_ncand == 1 && x * p1 > 1 && b1 * b2 > 1) {
        tmp2c = alloc_ten(b2, x, s1, s2, nb);
        bt_chain_flag = (int *)arena_.alloc(sizeof(int) * nb);
        TGemmDesc g1, g2;
        g1.I[1] = x; g1.I[2] = p1; g1.sAi[1] = p1 * cdim; g1.sAi[2] = cdim; g1.sCi[1] = p1 * b1 * b2; g1.sCi[2] = b1 * b2;
        g1.K[2] = cdim; g1.sAk[2] = 1; g1.sBk[2] = b1 * b2;
        g1.J[1] = b1; g1.J[2] = b2; g1.sBj[1] = b2; g1.sBj[2] = 1; g1.sCj[1] = b2; g1.sCj[2] = 1;
        g1.wA = mps1.n; g1.wB = bt.t.n; g1.wC = (long)x * p1 * b1 * b2; g1.nbatch = nb;
        g1.dI[1].p = vx; g1.dK[2].p = vc; g1.dJ[2].p = vb;
        // tmp2[b2,x,s1,s2] = sum_{p1,b1} site[p1,b1,s1,s2] tmp1[x,p1,b1,b2]:  I2 = (s1, s2), K2 = (p1, b1), J2 = (x, b2)
        g2.I[1] = s1; g2.I[2] = s2; g2.sAi[1] = st[l1]; g2.sAi[2] = st[l2]; g2.sCi[1] = s2; g2.sCi[2] = 1;
        g2.K[1] = p1; g2.K[2] = b1; g2.sAk[1] = st[lc]; g2.sAk[2] = st[lb]; g2.sBk[1] = b1 * b2; g2.sBk[2] = b2;
        g2.J[1] = x; g2.J[2] = b2; g2.sBj[1] = p1 * b1 * b2; g2.sBj[2] = 1; g2.sCj[1] = s1 * s2; g2.sCj[2] = x * s1 * s2;
        g2.wB = g1.wC; g2.wC = tmp2c.n; g2.nbatch = nb;
        g2.dJ[1].p = vx; g2.dJ[2].p = vb;
        g2.selA = ss.sel; g2.selA_mul = slot_; g2.selA_inc = ss.inc; g2.seldivA = 1; g2.wA = 0;
        TGemmChainMap mp;
        mp.mapK[1] = 2; mp.mapK[2] = 4;      // K2 = (p1, b1): p1 = I1[2], b1 = J1[1]
        mp.mapJ[1] = 1; mp.mapJ[2] = 5;      // J2 = (x, b2):  x = I1[1],  b2 = J1[2]
        const double fl = 2.0 * nb * ((double)(x * p1) * cdim * (double)(b1 * b2) + (double)(b2 * x) * (double)(p1 * b1) * (double)(s1 * s2));
        // round 4: all three contractions in one launch (tgemm_chain3_kernel: tmp1 and tmp2 resident in LDS, the bond x walked in
        // chunks when the live intermediates exceed the buffers) (the two-stage chain + separate launch of round 3 was a switch until round 6)
        constexpr bool no_bt3 = false;
        // ... when the bond x is walked in at most three chunks of the 4096-float buffers (static extents; they follow the live bonds
        // through the bond shrink): with the bonds of a real state (x = b2 = 32: sixteen chunks of two) the three-stage kernel is 2 %
        // slower than the two-stage chain + separate launch (368 against 375 sweeps/s at 2048 walkers)
        const long per_x = (long)p1 * b1 * b2;
        const bool few_chunks = per_x <= 4096 && (x + (4096 / per_x) - 1) / (4096 / per_x) <= 3;
        if (!no_bt3 && few_chunks) {
          TGemmDesc g3;
          g3.I[1] = x; g3.I[2] = s2; g3.sCi[1] = s2 * y; g3.sCi[2] = y;
          g3.K[1] = b2; g3.K[2] = s1; g3.sBk[1] = s1 * y; g3.sBk[2] = y;
          g3.J[2] = y; g3.sBj[2] = 1; g3.sCj[2] = 1;
          g3.wB = mps2.n; g3.nbatch = nb;
          g3.dI[1].p = vx; g3.dI[1].mask = 1;     // the new BTen is written in full
          g3.dK[1].p = vb;
          g3.dJ[2].p = vy; g3.dJ[2].mask = 1;
          TGemmChain3Map mp3;
          mp3.mapI[1] = 4; mp3.mapI[2] = 2;    // I3 = (x, s2):  x = J2[1],  s2 = I2[2]
          mp3.mapK[1] = 5; mp3.mapK[2] = 1;    // K3 = (b2, s1): b2 = J2[2], s1 = I2[1]
          mp3.chunkI = 1;
          BTenDev o3;
          o3.t = alloc_ten(x, s2, y, 1, nb);
          g3.wC = o3.t.n;
          const double fl3 = fl + 2.0 * nb * (double)(x * s2) * (double)(b2 * s1) * (double)y;
          prof_begin(PROF_ENV, fl3, fl3);
          const int done = tgemm_chain3_launch(stream_, g1, g2, g3, mp, mp3, (const float *)mps1.p, (const float *)bt.t.p,
                                               (const float *)sel_base(ss), (const float *)mps2.p, (float *)o3.t.p, bt_chain_flag, skip);
          prof_end();
          if (done == 2) {
            arena_.free(bt_chain_flag);
            free_ten(tmp2c);
            inject(INJ_E, o3.t.p, o3.t.n, nb);
            o3.logscale = nullptr;
            if (normalise) {
              o3.logscale = (double *)arena_.alloc(sizeof(double) * nw_);
              PG_CHECK_HIP(hipMemcpyAsync(o3.logscale, bt.logscale, sizeof(double) * nw_, hipMemcpyDeviceToDevice, stream_));
              normalize(o3.t.p, o3.t.n, o3.t.n, nw_, o3.logscale);
            }
            return o3;
          }
          free_ten(o3.t);
        }
        prof_begin(PROF_ENV, fl, fl);
        bt_chained = tgemm_chain_launch(stream_, g1, g2, mp, (const float *)mps1.p, (const float *)bt.t.p, (const float *)sel_base(ss),
                                        (float *)tmp2c.p, bt_chain_flag, 1, 0);
        prof_end();
        if (!bt_chained) { arena_.free(bt_chain_flag); bt_chain_flag = nullptr; }
      }
    }
    // tmp1[x,p1,b1,b2] = sum_c mps1[x,p1,c] bten[c,b1,b2]     (per walker)
    DTen<T> tmp1;
    if (bt_chained < 2) tmp1 = alloc_ten(x, p1, b1, b2, nb1);
    if (bt_chained < 2) {
      TGemmDesc g;
      g.I[2] = x * p1; g.sAi[2] = cdim; g.sCi[2] = b1 * b2;
      g.K[2] = cdim; g.sAk[2] = 1; g.sBk[2] = b1 * b2;
      g.J[2] = b1 * b2; g.sBj[2] = 1; g.sCj[2] = 1;
      if (vx || vc || vb) {   // (x, p1) and (b1, b2) as separate sub-indices: the live bonds are x and b2
        g.I[1] = x; g.I[2] = p1; g.sAi[1] = p1 * cdim; g.sAi[2] = cdim; g.sCi[1] = p1 * b1 * b2; g.sCi[2] = b1 * b2;
        g.J[1] = b1; g.J[2] = b2; g.sBj[1] = b2; g.sBj[2] = 1; g.sCj[1] = b2; g.sCj[2] = 1;
        g.dI[1].p = vx; g.dI[1].div = bt_ncand; g.dK[2].p = vc; g.dK[2].div = bt_ncand; g.dJ[2].p = vb; g.dJ[2].div = bt_ncand;
      }
      g.wA = mps1.n; g.bdivA = bt_ncand; g.wB = bt.t.n; g.wC = tmp1.n; g.nbatch = nb1;
      g.batch_flag = bt_chain_flag;
      const double fl = 2.0 * nb1 * (double)(x * p1) * cdim * (double)(b1 * b2);
      prof_begin(PROF_ENV, bt_chained ? 0.0 : fl, bt_chained ? 0.0 : fl);
      tgemm_launch<T, T, T, T>(stream_, g, mps1.p, bt.t.p, tmp1.p);
      prof_end();
    }
    // tmp2[b2,x,s1,s2] = sum_{p1,b1} tmp1[x,p1,b1,b2] site[lc<-p1, lb<-b1, l1->s1, l2->s2]
    DTen<T> tmp2 = bt_chained ? tmp2c : alloc_ten(b2, x, s1, s2, nb);
    if (bt_chained < 2) {
      TGemmDesc g;
      g.I[1] = b2; g.I[2] = x; g.sAi[1] = 1; g.sAi[2] = p1 * b1 * b2; g.sCi[1] = x * s1 * s2; g.sCi[2] = s1 * s2;
      g.K[1] = p1; g.K[2] = b1; g.sAk[1] = b1 * b2; g.sAk[2] = b2; g.sBk[1] = st[lc]; g.sBk[2] = st[lb];
      g.J[1] = s1; g.J[2] = s2; g.sBj[1] = st[l1]; g.sBj[2] = st[l2]; g.sCj[1] = s2; g.sCj[2] = 1;
      g.wA = tmp1.n; g.bdivA = ncand / bt_ncand; g.wC = tmp2.n; g.nbatch = nb;
      g.dI[1].p = vb; g.dI[1].div = ncand; g.dI[2].p = vx; g.dI[2].div = ncand;
      g.batch_flag = bt_chain_flag;
      const double fl = 2.0 * nb * (double)(b2 * x) * (double)(p1 * b1) * (double)(s1 * s2);
      prof_begin(PROF_ENV, bt_chained ? 0.0 : fl, bt_chained ? 0.0 : fl);
      launch_site_gemm(g, ss, ncand, tmp1.p, tmp2.p);
      prof_end();
    }
    if (bt_chain_flag) arena_.free(bt_chain_flag);
    // out[x,s2,y] = sum_{b2,s1} tmp2[b2,x,s1,s2] mps2[b2,s1,y]
    BTenDev o;
    o.t = alloc_ten(x, s2, y, 1, nb);
    {
      TGemmDesc g;
      g.I[1] = x; g.I[2] = s2; g.sAi[1] = s1 * s2; g.sAi[2] = 1; g.sCi[1] = s2 * y; g.sCi[2] = y;
      g.K[1] = b2; g.K[2] = s1; g.sAk[1] = x * s1 * s2; g.sAk[2] = s2; g.sBk[1] = s1 * y; g.sBk[2] = y;
      g.J[2] = y; g.sBj[2] = 1; g.sCj[2] = 1;
      g.wA = tmp2.n; g.wB = mps2.n; g.bdivB = ncand; g.wC = o.t.n; g.nbatch = nb;
      g.dI[1].p = vx; g.dI[1].div = ncand; g.dI[1].mask = 1;     // the new BTen is written in full
      g.dK[1].p = vb; g.dK[1].div = ncand;
      g.dJ[2].p = vy; g.dJ[2].div = ncand; g.dJ[2].mask = 1;
      const double fl = 2.0 * nb * (double)(x * s2) * (double)(b2 * s1) * (double)y;
      prof_begin(PROF_ENV, fl, fl);
      tgemm_launch<T, T, T, T>(stream_, g, tmp2.p, mps2.p, o.t.p);
      prof_end();
    }
    if (tmp1.p) free_ten(tmp1);
    free_ten(tmp2);
    inject(INJ_E, o.t.p, o.t.n, nb);
    o.logscale = nullptr;
    if (normalise) {
      o.logscale = (double *)arena_.alloc(sizeof(double) * nw_);
      PG_CHECK_HIP(hipMemcpyAsync(o.logscale, bt.logscale, sizeof(double) * nw_, hipMemcpyDeviceToDevice, stream_));
      normalize(o.t.p, o.t.n, o.t.n, nw_, o.logscale);
    }
    return o;
  }

  // site-tensor GEMM: B operand = sitps slot chosen by a selector.  Configuration selectors are
  // indexed by walker (b / ncand), candidate tables by batch entry.
  void launch_site_gemm(TGemmDesc &g, const SiteSel &ss, int ncand, const T *A, T *C) {
    g.selB = ss.sel;
    g.selB_mul = slot_;
    const bool per_walker = ss.per_walker;
    if (per_walker) {
      // emulate sel[(b / ncand) * inc] with inc applied to the walker index
      g.selB_inc = ss.inc;
      g.seldivB = ncand;
    } else {
      g.selB_inc = ss.inc;
      g.seldivB = 1;
    }
    g.wB = 0;
    tgemm_launch<T, T, T, T>(stream_, g, A, sel_base(ss), C);
  }

  // same with the site tensor as the A operand (C[(site legs), (...)]: lanes of the MFMA tile then run
  // along the other operand's contiguous index, which keeps the stores of C coalesced)
  void launch_site_gemm_a(TGemmDesc &g, const SiteSel &ss, int ncand, const T *B, T *C, bool acc64 = false) {
    g.selA = ss.sel;
    g.selA_mul = slot_;
    g.selA_inc = ss.inc;
    g.seldivA = ss.per_walker ? ncand : 1;
    g.wA = 0;
    if (acc64) tgemm_launch<T, T, T, Acc>(stream_, g, sel_base(ss), B, C);     // (experiments: float64 accumulation)
    else tgemm_launch<T, T, T, T>(stream_, g, sel_base(ss), B, C);
  }

  void absorb(int pos, int num);
  BMPSDev absorb_simple(int pos, int num, const BMPSDev &in);   // engine_cplx.h
  BMPSDev absorb_svd(int pos, int num, const BMPSDev &in);
  bool absorb_impl(int pos, int num, bool full_bonds, const BMPSDev &in, BMPSDev &out);
  // ---- variational compression schemes (engine_var.h) ----
  BMPSDev absorb_variational(int pos, int num, const BMPSDev &in);
  BMPSDev truncate_bmps(const BMPSDev &in, int kmax);
  void ein(const EinView<T> &a, const EinView<T> &b, const EinView<T> &c, T *cp);
  DTen<T> svd_rows(DTen<T> &M, int m, int len, int k, double terr, int dmin, T *S, const int *mdyn = nullptr, int mmul = 1,
                   int *kn_out = nullptr, int inner = 1, const int *inner_live = nullptr);
  // ---- two-row environments and NNN / TNN / sqrt5 traces (engine_nnn.h) ----
  void clear_bten2(int pos, int keep) {
    auto &v = bten2_[pos];
    while ((int)v.size() > keep) {
      arena_.free(v.back().t.p);
      arena_.free(v.back().logscale);
      v.pop_back();
    }
  }
  const BTenDev &bten2_at_slice(int pos, int idx) const {   // bmps_contractor.h:1012-1018
    int k = idx;
    if (pos == DOWN) k = Ly_ - 1 - idx;
    if (pos == RIGHT) k = Lx_ - 1 - idx;
    PG_REQUIRE(k >= 0 && k < (int)bten2_[pos].size(), 3, "BTen2 environment not available for this slice");
    return bten2_[pos][k];
  }
  struct SitePick { int r, c, cand; };   // cand < 0: the walker's configuration; else column of the candidate table
  SiteSel pick(const SitePick &s, const int *dcand, int ncols) const {
    if (s.cand < 0) return cfg_site(s.r, s.c);
    return SiteSel{s.r, s.c, dcand + s.cand, ncols};
  }
  BTenDev bten2_step(int post, const BTenDev &bt, const DTen<T> &mps1, const SiteSel &s1, const SiteSel &s2,
                     const DTen<T> &mps2, int ncand, int bt_ncand, bool normalise);
  void finish_dot4(const DTen<T> &a, const DTen<T> &b, int nc, double *lsum, double *out);
  int *upload_cand(int ncand, int ncols, const int32_t *cand);
  // sel (optional, f32): the kernel of the walkers with at most JR_BR live rows selects / normalises their rows into Vt
  // itself; returns true when it did (select_rows_kernel then skips those walkers)
  // rows_cap (0 = none): the caller vouches that no walker has more live rows (hint of the row absorbed before, verified
  // against the live counts read back at the end of the absorption): the kernels of the larger size classes are not launched
  bool launch_jacobi(T *M, long wM, int m, int len, int use_lds, size_t need, const int *mdyn, int mdyn_mul, int mid_hi = 0,
                     const JrSelect *sel = nullptr, int rows_cap = 0);
  static bool jacobi_small_ok(int len, int m, const int *mdyn);

  int Ly_, Lx_, D_, dp_, chi_min_, chi_;
  double trunc_err_;
  int maxw_, nw_ = 0, device_ = 0;
  long slot_;
  bool have_state_ = false;
  hipStream_t stream_;
  // side stream for rare-walker kernels that would otherwise serialise the main stream behind a handful of long blocks
  // (fork after ev_fork_, joined through ev_join_ before their results are read)
  hipStream_t side_stream_ = nullptr;
  hipEvent_t ev_fork_ = nullptr, ev_join_ = nullptr;
  Arena arena_;
  T *sitps_ = nullptr;
  int *cfg_ = nullptr, *flag_ = nullptr;
  std::vector<int> hcfg_;
  std::vector<BMPSDev> bmps_[4];
  std::vector<BTenDev> bten_[4];
  std::vector<BTenDev> bten2_[4];   // two-row (rank-4) environments, bten_set2_ of the reference (the ACTIVE set)
  // Second BTen2 set + a persistent one-slice configuration override (round 5): the environment-reusing diagonal hop of a
  // FERMIONIC state.  In the sign-decorated form a hop across a plaquette diagonal flips the variant of every site between its
  // two ends in the mode order, i.e. of row r right of the plaquette and of row r + 1 left of it: the hopped amplitude is a local
  // replacement against "twisted" environments -- the LEFT BTen2 grown with row r + 1 under flipped variants, the RIGHT one with
  // row r flipped.  bten2_select_set swaps which set the init / grow / shift calls work on; cfg_override_slice makes cfg_site
  // read one row (column) from another table while they do; replace_plaquette_trace closes a plaquette with four replaced
  // tensors between a LEFT environment of one set and a RIGHT environment of the other.
  std::vector<BTenDev> bten2_inactive_[4];
  int bten2_active_ = 0;
  int *cfg_ovr_tab_ = nullptr;       // [walker][Ly * Lx]: the walkers' table with one slice replaced (cfg_override_slice)
  void clear_bten2_sets() {           // every invalidation of the environments drops both sets and the override
    for (int p = 0; p < 4; ++p) {
      clear_bten2(p, 0);
      for (auto &b : bten2_inactive_[p]) { arena_.free(b.t.p); arena_.free(b.logscale); }
      bten2_inactive_[p].clear();
    }
    if (bten2_active_) bten2_active_ = 0;
    if (cfg_ovr_tab_) { arena_.free(cfg_ovr_tab_); cfg_ovr_tab_ = nullptr; ovr_on_ = false; ovr_cfg_ = nullptr; }
  }
  struct ProfRec { hipEvent_t a, b; int cat; double alg, exec; bool a_shared; };
  mutable bool prof_chain_ok_ = false;   // the last profiling call was a prof_end of this API call: its event can open the next bracket
  std::vector<ProfRec> prof_;
  std::vector<hipEvent_t> ev_pool_;
  bool prof_on_ = false;
  double prof_ms_[PROF_NCAT] = {0}, prof_alg_[PROF_NCAT] = {0}, prof_exec_[PROF_NCAT] = {0};
  long prof_n_[PROF_NCAT] = {0};
  std::vector<BMPSDev> parked_[4];
  int parked_keep_[4] = {0, 0, 0, 0};
  int scheme_ = 0, iter_max_ = 0;
  double conv_tol_ = 0.0;
  long n_var_iters_ = 0;
  long n_absorb_ = 0, n_jacobi_ = 0, jacobi_sweeps_sum_ = 0, jacobi_sweeps_max_ = 0;
  std::vector<char> redo_seen_[4];        // per stack position and row / column: its hint-sized absorption was redone before
  std::vector<int> carry_seen_[4];        // ... and the largest live carry of its last absorption (-1: not absorbed yet on this state)
  long n_redo_ = 0;                       // absorptions done twice: a shrunk bond was filled, or a rank hint of the row before was missed
  double live_sum_ = 0, live_full_ = 0;   // diagnostics: sum of live carry rows / sum of carry sizes
  long live_max_ = 0;                     // ... and the largest live carry of any walker (> 32: the dense route ran)
  T *holes_ = nullptr;                    // resident hole store [walker][site][D^4]
  double *holes_ls_ = nullptr;            // its log-scales [walker][site]
  double *so_ = nullptr, *seo_ = nullptr; // gradient accumulators
  int *sweeps_ = nullptr;
  unsigned long long *flopc_ = nullptr;   // device flop counters [PROF_NCAT] then byte counters [PROF_NCAT]
  T *sr_o_ = nullptr;                      // O* samples [sample][site][D^4]
  int *sr_cfg_ = nullptr;                  // their configurations [sample][site]
  int *sr_ne_ = nullptr;                   // elements of the (compact) site tensor per site
  double *sr_delta_ = nullptr, *sr_v_ = nullptr, *sr_out_ = nullptr;
  int sr_cap_ = 0, sr_n_ = 0;
  std::vector<uint32_t> sr_map_c_, sr_map_p_;   // compact <-> padded element index of every stored tensor element
  // ---- BMPSWalker objects (engine_walker.h) ----
  struct WalkerDev {
    BMPSDev b;                       // the forked boundary MPS (all walkers of the context)
    int pos = UP, stack = 0;         // evolution direction, layers absorbed (+1: the vacuum)
    std::vector<BTenDev> btl, btr;   // walker-owned BTen caches; btl[k] covers columns [0, k), btr[k] columns [N - k, N)
    int lcol = 0, rcol = 0;          // left edge (exclusive upper bound) / right edge (exclusive lower bound)
    int mpo_num = -1;                // the current TransferMPO: slice of the network ...
    int *mpo_cfg = nullptr;          // ... under another configuration table [walker][Ly * Lx] (nullptr: the walkers' own)
    T *mpo_tens = nullptr;           // ... or explicit tensors [site along the slice][mpo_nt][D^4 slot]
    int mpo_nt = 0;
  };
  struct MpoScope;
  std::map<int, WalkerDev> walkers_;
  int next_walker_id_ = 1;
  int *iota_ = nullptr;              // [0, 0, 1, 2, ...]: selector of explicit tensor sets (iota_ : shared, iota_ + 1 : per walker)
  bool ovr_on_ = false, ovr_hor_ = true;
  int ovr_num_ = -1, ovr_nt_ = 0;
  const int *ovr_cfg_ = nullptr;
  const T *ovr_tens_ = nullptr;
  BMPSDev copy_bmps(const BMPSDev &b);
  BMPSDev absorb_any(int pos, int num, const BMPSDev &in);
  WalkerDev &walker_ref(int id);
  void walker_free_bten(WalkerDev &w);
  void walker_free_mpo(WalkerDev &w);
  void walkers_clear();
  void ensure_iota();
  const BMPSDev &walker_opposite(const WalkerDev &w, int opp_level, const char *what);
  void walker_grow_left(WalkerDev &w, const BMPSDev &opp);
  void walker_grow_right(WalkerDev &w, const BMPSDev &opp);
  SiteSel walker_site(const WalkerDev &w, int col, const int32_t *states, int sstride, const double *tensor, int n_tensors, long tstride,
                      std::vector<void *> &tmp);

  bool dbg_sweeps_ = false;
  int inject_ = 0;                         // PEPSGPU_INJECT_F32 mask (float64 engine, experiments)
};

}  // namespace pepsgpu
