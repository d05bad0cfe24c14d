// extern "C" boundary of libpepsgpu.so (declared in include/pepsgpu.h).
#include <cstring>
#include "../../include/pepsgpu.h"
#include "engine_impl.h"
#include "engine_nnn.h"
#include "engine_sr.h"
#include "engine_var.h"
#include "engine_cplx.h"
#include "engine_walker.h"
#include "engine_sweep.h"
#include "comm.h"

using namespace pepsgpu;

namespace pepsgpu {
bool tgemm_use_mfma() {
  static int v = -1;
  if (v < 0) {
    const char *e = getenv("PEPSGPU_NO_MFMA");
    v = (e && e[0] == '1') ? 0 : 1;
  }
  return v == 1;
}
}  // namespace pepsgpu

struct pepsgpu_ctx {
  EngineBase *eng = nullptr;
  Comm comm;
  std::string err;
};

static thread_local std::string g_err;

template <typename F>
static int guarded(pepsgpu_ctx *ctx, F &&f) {
  try {
    f();
    return PEPSGPU_OK;
  } catch (const Error &e) {
    if (ctx) ctx->err = e.what(); else g_err = e.what();
    return e.code;
  } catch (const std::exception &e) {
    if (ctx) ctx->err = e.what(); else g_err = e.what();
    return PEPSGPU_EHIP;
  }
}

// every entry makes the context's GPU the calling thread's current device first: a context is bound to one device, the
// caller (another context, torch, another thread) may have changed the thread's device since the last call
#define CTX_CALL(body)                                     \
  if (!ctx || !ctx->eng) return PEPSGPU_EINVAL;            \
  return guarded(ctx, [&]() { PG_CHECK_HIP(hipSetDevice(ctx->eng->device_id)); body; })

extern "C" {

const char *pepsgpu_version(void) { return "pepsgpu 0.1 (gfx950)"; }

int pepsgpu_ctx_create(pepsgpu_ctx **out, int device, int dtype, int rows, int cols, int D, int phys_dim,
                       int chi_min, int chi_max, double trunc_err, int scheme, int max_walkers) {
  if (!out) return PEPSGPU_EINVAL;
  *out = nullptr;
  pepsgpu_ctx *ctx = new pepsgpu_ctx;
  int rc = guarded(nullptr, [&]() {
    PG_REQUIRE(scheme >= PEPSGPU_SVD_COMPRESS && scheme <= PEPSGPU_VARIATION1SITE, 1, "unknown CompressMPSScheme");
    PG_REQUIRE(chi_min <= chi_max, 1, "D_min > D_max");
    int ndev = 0;
    PG_CHECK_HIP(hipGetDeviceCount(&ndev));
    PG_REQUIRE(ndev > 0 && device >= 0 && device < ndev, 2, "no such HIP device");
    // walkers x candidates is the z extent of the contraction grids (65535); refuse here, not at the first launch
    PG_REQUIRE(max_walkers >= 1 && max_walkers <= 65535, 1, "max_walkers must be in [1, 65535] (grid z limit)");
    if (dtype == PEPSGPU_F32)
      ctx->eng = new Engine<float>(device, rows, cols, D, phys_dim, chi_min, chi_max, trunc_err, max_walkers);
    else if (dtype == PEPSGPU_F64)
      ctx->eng = new Engine<double>(device, rows, cols, D, phys_dim, chi_min, chi_max, trunc_err, max_walkers);
    else if (dtype == PEPSGPU_C128)
      ctx->eng = new Engine<c128>(device, rows, cols, D, phys_dim, chi_min, chi_max, trunc_err, max_walkers);
    else
      throw Error(1, "unknown dtype");
    // variational schemes: convergence_tol / iter_max default to 1e-10 / 10 until pepsgpu_set_truncate_params sets them
    if (scheme != PEPSGPU_SVD_COMPRESS) ctx->eng->set_truncate_params(chi_min, chi_max, trunc_err, scheme, 1e-10, 10);
  });
  if (rc != PEPSGPU_OK) {
    delete ctx;
    return rc;
  }
  *out = ctx;
  return PEPSGPU_OK;
}

int pepsgpu_set_truncate_params(pepsgpu_ctx *ctx, int chi_min, int chi_max, double trunc_err, int scheme,
                                double convergence_tol, int iter_max) {
  CTX_CALL(ctx->eng->set_truncate_params(chi_min, chi_max, trunc_err, scheme, convergence_tol, iter_max));
}

void pepsgpu_ctx_destroy(pepsgpu_ctx *ctx) {
  if (!ctx) return;
  if (ctx->eng) (void)hipSetDevice(ctx->eng->device_id);
  ctx->comm.destroy();
  delete ctx->eng;
  delete ctx;
}

const char *pepsgpu_last_error(pepsgpu_ctx *ctx) { return ctx ? ctx->err.c_str() : g_err.c_str(); }

int pepsgpu_state_upload(pepsgpu_ctx *ctx, const void *p, int host_dtype) {
  CTX_CALL(PG_REQUIRE(p != nullptr, 1, "null state buffer"); ctx->eng->state_upload(p, host_dtype));
}
int pepsgpu_walkers_set_configs(pepsgpu_ctx *ctx, int n, const int32_t *cfg) {
  CTX_CALL(PG_REQUIRE(cfg != nullptr, 1, "null configuration buffer"); ctx->eng->set_configs(n, cfg));
}
int pepsgpu_walkers_get_configs(pepsgpu_ctx *ctx, int32_t *out) { CTX_CALL(ctx->eng->get_configs(out)); }
int pepsgpu_n_walkers(pepsgpu_ctx *ctx) { return (ctx && ctx->eng) ? ctx->eng->n_walkers() : -1; }

static void check_pos(int pos) { PG_REQUIRE(pos >= 0 && pos < 4, 1, "bad BMPSPOSITION"); }

int pepsgpu_grow_bmps_step(pepsgpu_ctx *ctx, int pos) { CTX_CALL(check_pos(pos); ctx->eng->grow_bmps_step(pos)); }
int pepsgpu_grow_full_bmps(pepsgpu_ctx *ctx, int pos) { CTX_CALL(check_pos(pos); ctx->eng->grow_full_bmps(pos)); }
int pepsgpu_grow_bmps_for_row(pepsgpu_ctx *ctx, int row) { CTX_CALL(ctx->eng->grow_bmps_for_row(row)); }
int pepsgpu_grow_bmps_for_col(pepsgpu_ctx *ctx, int col) { CTX_CALL(ctx->eng->grow_bmps_for_col(col)); }
int pepsgpu_shift_bmps_window(pepsgpu_ctx *ctx, int pos) { CTX_CALL(check_pos(pos); ctx->eng->shift_bmps_window(pos)); }
int pepsgpu_delete_inner_bmps(pepsgpu_ctx *ctx, int pos) { CTX_CALL(check_pos(pos); ctx->eng->delete_inner_bmps(pos)); }
int pepsgpu_sweep_slice_exchange(pepsgpu_ctx *ctx, int orientation, int slice, int n_uniform, const double *uniforms,
                                 double *amplitude_inout, int32_t *consumed_out, int32_t *accepted_out, int32_t *slice_states_out) {
  CTX_CALL(PG_REQUIRE(uniforms && amplitude_inout && consumed_out && accepted_out, 1, "null buffer");
           ctx->eng->sweep_slice_impl(0, orientation, slice, n_uniform, uniforms, nullptr, 0, nullptr, amplitude_inout, consumed_out,
                                      accepted_out, slice_states_out));
}
int pepsgpu_sweep_slice_exchange_tab(pepsgpu_ctx *ctx, int orientation, int slice, int n_uniform, const double *uniforms,
                                     const int32_t *pair_table, double *amplitude_inout, int32_t *consumed_out, int32_t *accepted_out,
                                     int32_t *slice_states_out) {
  CTX_CALL(PG_REQUIRE(uniforms && amplitude_inout && consumed_out && accepted_out, 1, "null buffer");
           ctx->eng->sweep_slice_impl(0, orientation, slice, n_uniform, uniforms, pair_table, 0, nullptr, amplitude_inout, consumed_out,
                                      accepted_out, slice_states_out));
}
int pepsgpu_sweep_slice_fullspace(pepsgpu_ctx *ctx, int orientation, int slice, int phys_dim, const uint32_t *engine_words,
                                  double *amplitude_inout, int32_t *accepted_out, int32_t *slice_states_out) {
  CTX_CALL(PG_REQUIRE(engine_words && amplitude_inout && accepted_out, 1, "null buffer");
           ctx->eng->sweep_slice_impl(1, orientation, slice, 0, nullptr, nullptr, phys_dim, engine_words, amplitude_inout, nullptr,
                                      accepted_out, slice_states_out));
}
int pepsgpu_nn_exchange_slice(pepsgpu_ctx *ctx, int orientation, int slice, int punch_holes, double *psi_out, double *psi_exchanged_out) {
  CTX_CALL(PG_REQUIRE(psi_out && psi_exchanged_out, 1, "null buffer");
           ctx->eng->nn_exchange_slice(orientation, slice, punch_holes, psi_out, psi_exchanged_out));
}
int pepsgpu_walker_create(pepsgpu_ctx *ctx, int pos, int level, int *walker_out) {
  CTX_CALL(check_pos(pos); PG_REQUIRE(walker_out != nullptr, 1, "null output"); *walker_out = ctx->eng->walker_create(pos, level));
}
int pepsgpu_walker_clone(pepsgpu_ctx *ctx, int walker, int *walker_out) {
  CTX_CALL(PG_REQUIRE(walker_out != nullptr, 1, "null output"); *walker_out = ctx->eng->walker_clone(walker));
}
int pepsgpu_walker_destroy(pepsgpu_ctx *ctx, int walker) { CTX_CALL(ctx->eng->walker_destroy(walker)); }
int pepsgpu_walker_info(pepsgpu_ctx *ctx, int walker, int *pos_out, int *stack_size_out, int *bten_left_col_out, int *bten_right_col_out) {
  CTX_CALL(ctx->eng->walker_info(walker, pos_out, stack_size_out, bten_left_col_out, bten_right_col_out));
}
int pepsgpu_walker_set_mpo(pepsgpu_ctx *ctx, int walker, int num, const int32_t *states, const double *tensors, int n_tensors) {
  CTX_CALL(ctx->eng->walker_set_mpo(walker, num, states, tensors, n_tensors));
}
int pepsgpu_walker_evolve(pepsgpu_ctx *ctx, int walker) { CTX_CALL(ctx->eng->walker_evolve(walker)); }
int pepsgpu_walker_evolve_step(pepsgpu_ctx *ctx, int walker) { CTX_CALL(ctx->eng->walker_evolve_step(walker)); }
int pepsgpu_walker_contract_row(pepsgpu_ctx *ctx, int walker, int opp_level, double *out) {
  CTX_CALL(PG_REQUIRE(out != nullptr, 1, "null output"); ctx->eng->walker_contract_row(walker, opp_level, out));
}
int pepsgpu_walker_init_bten(pepsgpu_ctx *ctx, int walker, int opp_level, int position, int target_col) {
  CTX_CALL(ctx->eng->walker_init_bten(walker, opp_level, position, target_col));
}
int pepsgpu_walker_grow_bten_step(pepsgpu_ctx *ctx, int walker, int opp_level, int position) {
  CTX_CALL(ctx->eng->walker_grow_bten_step(walker, opp_level, position));
}
int pepsgpu_walker_shift_bten_window(pepsgpu_ctx *ctx, int walker, int opp_level, int position) {
  CTX_CALL(ctx->eng->walker_shift_bten_window(walker, opp_level, position));
}
int pepsgpu_walker_trace_with_bten(pepsgpu_ctx *ctx, int walker, int opp_level, int site_col, int two_site, const int32_t *site_states,
                                   const double *site_tensors, int n_tensors, double *out) {
  CTX_CALL(PG_REQUIRE(out != nullptr, 1, "null output");
           PG_REQUIRE(!(site_states && site_tensors), 1, "name the replacement by states OR by tensors");
           ctx->eng->walker_trace(walker, opp_level, site_col, two_site, site_states, site_tensors, n_tensors, out));
}
int pepsgpu_walker_clear_bten(pepsgpu_ctx *ctx, int walker) { CTX_CALL(ctx->eng->walker_clear_bten(walker)); }
int pepsgpu_walker_get_bmps_tensor(pepsgpu_ctx *ctx, int walker, int idx, int *dims_out, double *data_out, double *logscale_out) {
  CTX_CALL(PG_REQUIRE(dims_out != nullptr, 1, "null dims"); ctx->eng->walker_get_tensor(walker, idx, dims_out, data_out, logscale_out));
}
int pepsgpu_bmps_park(pepsgpu_ctx *ctx, int pos, int keep_levels) { CTX_CALL(check_pos(pos); ctx->eng->bmps_park(pos, keep_levels)); }
int pepsgpu_bmps_unpark(pepsgpu_ctx *ctx, int pos) { CTX_CALL(check_pos(pos); ctx->eng->bmps_unpark(pos)); }
int pepsgpu_generate_bmps_approach(pepsgpu_ctx *ctx, int pos) {
  CTX_CALL(check_pos(pos); ctx->eng->generate_bmps_approach(pos));
}
int pepsgpu_bmps_stack_size(pepsgpu_ctx *ctx, int pos) {
  if (!ctx || !ctx->eng || pos < 0 || pos > 3) return -1;
  return ctx->eng->bmps_size(pos);
}
int pepsgpu_bten_stack_size(pepsgpu_ctx *ctx, int pos) {
  if (!ctx || !ctx->eng || pos < 0 || pos > 3) return -1;
  return ctx->eng->bten_size(pos);
}
int pepsgpu_get_bmps_tensor(pepsgpu_ctx *ctx, int pos, int level, int idx, int *dims, double *data, double *ls) {
  CTX_CALL(check_pos(pos); ctx->eng->get_bmps_tensor(pos, level, idx, dims, data, ls));
}

int pepsgpu_init_bten(pepsgpu_ctx *ctx, int pos, int slice) { CTX_CALL(check_pos(pos); ctx->eng->init_bten(pos, slice)); }
int pepsgpu_grow_full_bten(pepsgpu_ctx *ctx, int pos, int slice, int remain, int init) {
  CTX_CALL(check_pos(pos); ctx->eng->grow_full_bten(pos, slice, remain, init));
}
int pepsgpu_grow_bten_step(pepsgpu_ctx *ctx, int pos) { CTX_CALL(check_pos(pos); ctx->eng->grow_bten_step(pos)); }
int pepsgpu_shift_bten_window(pepsgpu_ctx *ctx, int pos) { CTX_CALL(check_pos(pos); ctx->eng->shift_bten_window(pos)); }
int pepsgpu_truncate_bten(pepsgpu_ctx *ctx, int pos, int len) { CTX_CALL(check_pos(pos); ctx->eng->truncate_bten(pos, len)); }

int pepsgpu_trace(pepsgpu_ctx *ctx, int row, int col, int dir, double *out) {
  CTX_CALL(ctx->eng->trace(row, col, dir, out));
}
int pepsgpu_replace_nn_trace(pepsgpu_ctx *ctx, int row, int col, int dir, int ncand, const int32_t *cand, double *out) {
  CTX_CALL(PG_REQUIRE(ncand >= 1 && cand, 1, "need >= 1 candidate"); ctx->eng->replace_nn_trace(row, col, dir, ncand, cand, out));
}
int pepsgpu_replace_one_trace(pepsgpu_ctx *ctx, int row, int col, int orient, int ncand, const int32_t *cand, double *out) {
  CTX_CALL(PG_REQUIRE(ncand >= 1 && cand, 1, "need >= 1 candidate"); ctx->eng->replace_one_trace(row, col, orient, ncand, cand, out));
}
int pepsgpu_init_bten2(pepsgpu_ctx *ctx, int pos, int slice) { CTX_CALL(check_pos(pos); ctx->eng->init_bten2(pos, slice)); }
int pepsgpu_grow_full_bten2(pepsgpu_ctx *ctx, int pos, int slice, int remain, int init) {
  CTX_CALL(check_pos(pos); ctx->eng->grow_full_bten2(pos, slice, remain, init));
}
int pepsgpu_grow_bten2_step(pepsgpu_ctx *ctx, int pos, int slice) {
  CTX_CALL(check_pos(pos); ctx->eng->grow_bten2_step(pos, slice));
}
int pepsgpu_shift_bten2_window(pepsgpu_ctx *ctx, int pos, int slice) {
  CTX_CALL(check_pos(pos); ctx->eng->shift_bten2_window(pos, slice));
}
int pepsgpu_bten2_stack_size(pepsgpu_ctx *ctx, int pos) {
  if (!ctx || !ctx->eng || pos < 0 || pos > 3) return -1;
  return ctx->eng->bten2_size(pos);
}
int pepsgpu_bten2_select_set(pepsgpu_ctx *ctx, int set) { CTX_CALL(ctx->eng->bten2_select_set(set)); }
int pepsgpu_cfg_override_slice(pepsgpu_ctx *ctx, int orient, int num, const int32_t *states) {
  CTX_CALL(ctx->eng->cfg_override_slice(orient, num, states));
}
int pepsgpu_replace_plaquette_trace(pepsgpu_ctx *ctx, int row, int col, int ncand, const int32_t *cand, int left_set, int right_set,
                                    double *out) {
  CTX_CALL(PG_REQUIRE(out && (ncand == 0 || cand), 1, "null argument");
           ctx->eng->replace_plaquette_trace(row, col, ncand, cand, left_set, right_set, out));
}
int pepsgpu_replace_nnn_trace(pepsgpu_ctx *ctx, int row, int col, int nnn_dir, int orient, int ncand, const int32_t *cand,
                              double *out) {
  CTX_CALL(PG_REQUIRE((ncand == 0 || cand) && ncand >= 0 && out, 1, "bad candidate table");
           PG_REQUIRE(nnn_dir == 0 || nnn_dir == 1, 1, "bad diagonal direction");
           ctx->eng->replace_nnn_trace(row, col, nnn_dir, orient, ncand, cand, out));
}
int pepsgpu_replace_tnn_trace(pepsgpu_ctx *ctx, int row, int col, int orient, int ncand, const int32_t *cand, double *out) {
  CTX_CALL(PG_REQUIRE((ncand == 0 || cand) && ncand >= 0 && out, 1, "bad candidate table");
           ctx->eng->replace_tnn_trace(row, col, orient, ncand, cand, out));
}
int pepsgpu_replace_sqrt5_trace(pepsgpu_ctx *ctx, int row, int col, int link_dir, int orient, int ncand,
                                const int32_t *cand, double *out) {
  CTX_CALL(PG_REQUIRE((ncand == 0 || cand) && ncand >= 0 && out, 1, "bad candidate table");
           PG_REQUIRE(link_dir == 0 || link_dir == 1, 1, "bad diagonal direction");
           ctx->eng->replace_sqrt5_trace(row, col, link_dir, orient, ncand, cand, out));
}
int pepsgpu_punch_hole(pepsgpu_ctx *ctx, int row, int col, int orient, double *out) {
  CTX_CALL(ctx->eng->punch_hole(row, col, orient, out));
}
int pepsgpu_grad_reset(pepsgpu_ctx *ctx) { CTX_CALL(ctx->eng->grad_reset()); }
int pepsgpu_grad_accumulate_states(pepsgpu_ctx *ctx, const double *psi, const double *eloc, int exact_sum, const int32_t *states) {
  CTX_CALL(PG_REQUIRE(psi && eloc, 1, "null psi / eloc"); ctx->eng->grad_accumulate(psi, eloc, exact_sum, states));
}
int pepsgpu_grad_accumulate(pepsgpu_ctx *ctx, const double *psi, const double *eloc, int exact_sum) {
  CTX_CALL(PG_REQUIRE(psi && eloc, 1, "null psi / eloc"); ctx->eng->grad_accumulate(psi, eloc, exact_sum));
}
int pepsgpu_grad_read(pepsgpu_ctx *ctx, double *so, double *seo) {
  CTX_CALL(PG_REQUIRE(so && seo, 1, "null output"); ctx->eng->grad_read(so, seo));
}
int pepsgpu_grad_device_ptr(pepsgpu_ctx *ctx, void **so, void **seo, long *n_elems) {
  CTX_CALL(PG_REQUIRE(so && seo && n_elems, 1, "null output"); ctx->eng->grad_device_ptr(so, seo, n_elems));
}
int pepsgpu_comm_unique_id(void *id128_out) {
  return guarded(nullptr, [&]() {
    PG_REQUIRE(id128_out != nullptr, 1, "null output");
    ncclUniqueId id;
    PG_CHECK_RCCL(RcclApi::get().GetUniqueId(&id));
    memcpy(id128_out, &id, sizeof(id));
  });
}
int pepsgpu_comm_init(pepsgpu_ctx *ctx, int nranks, int rank, const void *id128) {
  CTX_CALL(ctx->comm.init(nranks, rank, id128));
}
int pepsgpu_comm_size(pepsgpu_ctx *ctx) { return ctx ? ctx->comm.nranks : -1; }
int pepsgpu_comm_rank(pepsgpu_ctx *ctx) { return ctx ? ctx->comm.rank : -1; }
int pepsgpu_comm_destroy(pepsgpu_ctx *ctx) { CTX_CALL(ctx->comm.destroy()); }
int pepsgpu_allreduce(pepsgpu_ctx *ctx, void *buf, long n, int dtype, int op, int on_device) {
  CTX_CALL(PG_REQUIRE(buf != nullptr || n == 0, 1, "null buffer"); PG_REQUIRE(n >= 0, 1, "negative count");
           ctx->comm.allreduce((hipStream_t)ctx->eng->stream_handle(), buf, (size_t)n, dtype, op, on_device != 0);
           if (on_device) ctx->eng->sync());
}
int pepsgpu_bcast_state(pepsgpu_ctx *ctx, int root) {
  CTX_CALL(void *p = nullptr; size_t bytes = 0; ctx->eng->state_device_ptr(&p, &bytes);
           PG_REQUIRE(ctx->comm.nranks == 1 || ctx->comm.comm != nullptr, 3, "pepsgpu_bcast_state: no communicator (pepsgpu_comm_init)");
           ctx->comm.bcast((hipStream_t)ctx->eng->stream_handle(), p, bytes, root);
           ctx->eng->state_adopted());
}
int pepsgpu_grad_allreduce(pepsgpu_ctx *ctx) {
  CTX_CALL(void *so = nullptr; void *seo = nullptr; long n = 0; ctx->eng->grad_device_ptr(&so, &seo, &n);
           hipStream_t s = (hipStream_t)ctx->eng->stream_handle();
           ctx->comm.allreduce(s, so, (size_t)n, 1, 0, true); ctx->comm.allreduce(s, seo, (size_t)n, 1, 0, true);
           ctx->eng->sync());
}
int pepsgpu_sr_begin(pepsgpu_ctx *ctx, int max_samples) { CTX_CALL(ctx->eng->sr_begin(max_samples)); }
int pepsgpu_sr_append(pepsgpu_ctx *ctx, const double *psi) { CTX_CALL(PG_REQUIRE(psi, 1, "null psi"); ctx->eng->sr_append(psi)); }
int pepsgpu_sr_count(pepsgpu_ctx *ctx) { return (ctx && ctx->eng) ? ctx->eng->sr_count() : -1; }
int pepsgpu_sr_sum(pepsgpu_ctx *ctx, double *out) { CTX_CALL(PG_REQUIRE(out, 1, "null output"); ctx->eng->sr_sum(out)); }
int pepsgpu_sr_matvec_c128(pepsgpu_ctx *ctx, const double *v, double mean_dot_v_re, double mean_dot_v_im, double scale, double *out) {
  CTX_CALL(PG_REQUIRE(v && out, 1, "null vector"); ctx->eng->sr_matvec_cplx(v, mean_dot_v_re, mean_dot_v_im, scale, out));
}
int pepsgpu_sr_matvec(pepsgpu_ctx *ctx, const double *v, double mean_dot_v, double scale, double *out) {
  CTX_CALL(PG_REQUIRE(v && out, 1, "null vector"); ctx->eng->sr_matvec(v, mean_dot_v, scale, out));
}
int pepsgpu_sr_cg_solve(pepsgpu_ctx *ctx, const double *b, const double *x0, double diag_shift, int max_iter,
                        double relative_tolerance, double absolute_tolerance, int residual_recompute_interval,
                        double orthogonality_threshold, double *x_out, double *residual_norm, int *iterations, int *reason) {
  CTX_CALL(PG_REQUIRE(b && x_out && residual_norm && iterations && reason, 1, "null argument");
           PG_REQUIRE(max_iter >= 0 && relative_tolerance >= 0.0 && absolute_tolerance >= 0.0, 1, "bad CG parameters");
           ctx->eng->sr_cg_solve(b, x0, diag_shift, max_iter, relative_tolerance, absolute_tolerance, residual_recompute_interval,
                                 orthogonality_threshold, x_out, residual_norm, iterations, reason));
}
int pepsgpu_sr_gram(pepsgpu_ctx *ctx, const void *remote_samples_dev, const int32_t *remote_configs_dev, int n_remote, double *out) {
  CTX_CALL(PG_REQUIRE(out, 1, "null output"); ctx->eng->sr_gram(remote_samples_dev, remote_configs_dev, n_remote, out));
}
int pepsgpu_sr_weighted_sum(pepsgpu_ctx *ctx, const double *y, double *out) {
  CTX_CALL(PG_REQUIRE(y && out, 1, "null argument"); ctx->eng->sr_weighted_sum(y, out));
}
int pepsgpu_sr_copy_samples(pepsgpu_ctx *ctx, void *dst_samples_dev, int32_t *dst_configs_dev) {
  CTX_CALL(PG_REQUIRE(dst_samples_dev && dst_configs_dev, 1, "null destination"); ctx->eng->sr_copy_samples(dst_samples_dev, dst_configs_dev));
}
int pepsgpu_update_local(pepsgpu_ctx *ctx, int nsites, const int32_t *sites, const int32_t *ns, const uint8_t *mask) {
  CTX_CALL(ctx->eng->update_local(nsites, sites, ns, mask));
}
int pepsgpu_erase_envs_after_update(pepsgpu_ctx *ctx, int row, int col) {
  CTX_CALL(ctx->eng->erase_envs_after_update(row, col));
}
int pepsgpu_evaluate_amplitude(pepsgpu_ctx *ctx, double *out) { CTX_CALL(ctx->eng->evaluate_amplitude(out)); }
int pepsgpu_walker_flags(pepsgpu_ctx *ctx, int32_t *out) { CTX_CALL(ctx->eng->read_flags(out)); }
int pepsgpu_sync(pepsgpu_ctx *ctx) { CTX_CALL(ctx->eng->sync()); }
int pepsgpu_stats(pepsgpu_ctx *ctx, double *out, int n) { CTX_CALL(ctx->eng->stats(out, n)); }
int pepsgpu_profile_enable(pepsgpu_ctx *ctx, int on) { CTX_CALL(ctx->eng->profile_enable(on)); }
int pepsgpu_profile_read(pepsgpu_ctx *ctx, double *out) { CTX_CALL(ctx->eng->profile_read(out)); }

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// diagnostics
template <typename TI, typename TO, typename TAcc>
static void diag_tgemm_t(const TGemmDesc &d, const void *A, size_t na, const void *B, size_t nb, void *C, size_t nc) {
  TI *dA, *dB;
  TO *dC;
  PG_CHECK_HIP(hipMalloc(&dA, na * sizeof(TI)));
  PG_CHECK_HIP(hipMalloc(&dB, nb * sizeof(TI)));
  PG_CHECK_HIP(hipMalloc(&dC, nc * sizeof(TO)));
  PG_CHECK_HIP(hipMemcpy(dA, A, na * sizeof(TI), hipMemcpyHostToDevice));
  PG_CHECK_HIP(hipMemcpy(dB, B, nb * sizeof(TI), hipMemcpyHostToDevice));
  PG_CHECK_HIP(hipMemcpy(dC, C, nc * sizeof(TO), hipMemcpyHostToDevice));
  tgemm_launch<TI, TI, TO, TAcc>(0, d, dA, dB, dC);
  PG_CHECK_HIP(hipDeviceSynchronize());
  PG_CHECK_HIP(hipMemcpy(C, dC, nc * sizeof(TO), hipMemcpyDeviceToHost));
  (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dC);
}

extern "C" int pepsgpu_diag_tgemm(int dtype_in, int dtype_out, const int *di, int n_ints, const void *A, size_t na, const void *B,
                       size_t nb, void *C, size_t nc, int nbatch, long wA, long wB, long wC) {
  return guarded(nullptr, [&]() {
    PG_REQUIRE(n_ints == 27, 1, "descriptor needs 27 ints");
    TGemmDesc d;
    for (int k = 0; k < 3; ++k) {
      d.I[k] = di[k]; d.J[k] = di[3 + k]; d.K[k] = di[6 + k];
      d.sAi[k] = di[9 + k]; d.sAk[k] = di[12 + k]; d.sBk[k] = di[15 + k];
      d.sBj[k] = di[18 + k]; d.sCi[k] = di[21 + k]; d.sCj[k] = di[24 + k];
    }
    d.wA = wA; d.wB = wB; d.wC = wC; d.nbatch = nbatch;
    // PEPSGPU_DIAG_TGEMM_MODE (read per call; f32 only): 1 = the wave-per-tile kernel (a per-entry live extent equal to the static one
    // routes there), 2 = the same with float64 accumulation on the f64 matrix cores (tg_direct_body_f64)
    const char *mode_s = getenv("PEPSGPU_DIAG_TGEMM_MODE");
    const int mode = mode_s ? atoi(mode_s) : 0;
    int *dext = nullptr;
    if (mode && dtype_in == 0 && dtype_out == 0) {
      std::vector<int> h(nbatch, d.I[2]);
      PG_CHECK_HIP(hipMalloc(&dext, sizeof(int) * nbatch));
      PG_CHECK_HIP(hipMemcpy(dext, h.data(), sizeof(int) * nbatch, hipMemcpyHostToDevice));
      d.dI[2].p = dext;
      d.acc64 = mode == 2;
    }
    struct Guard { int *p; ~Guard() { if (p) (void)hipFree(p); } } guard{dext};
    if (dtype_in == 0 && dtype_out == 0) diag_tgemm_t<float, float, float>(d, A, na, B, nb, C, nc);
    else if (dtype_in == 1 && dtype_out == 1) diag_tgemm_t<double, double, double>(d, A, na, B, nb, C, nc);
    else if (dtype_in == 0 && dtype_out == 1) diag_tgemm_t<float, double, double>(d, A, na, B, nb, C, nc);
    else throw Error(1, "unsupported dtype combination");
  });
}

// Forward pair of an absorption through tgemm_chain_kernel exactly as Engine::absorb_impl sets it up:
//   X[m,l,p,a2] = sum_a R[m,l,a] A[a,p,a2]  (kept in LDS),   P[m,u,l2,a2] = sum_{l,p} W[l,p,l2,u] X[m,l,p,a2]
// dims = {m, l, a, p, a2, l2, u} (static), live = per entry {m_live, a_live, a2_live}; W is one tensor per entry.
// flags_out[b] = 0 done, -1 the live X exceeds the LDS buffer (P[b] untouched).
extern "C" int pepsgpu_diag_tgemm_chain(const int *dims, const int32_t *live, int nbatch, const float *R, const float *A,
                                        const float *W, float *P_out, int32_t *flags_out) {
  return guarded(nullptr, [&]() {
    const int m = dims[0], l = dims[1], a = dims[2], p = dims[3], a2 = dims[4], l2 = dims[5], u = dims[6];
    const size_t nR = (size_t)m * l * a, nA = (size_t)a * p * a2, nW = (size_t)l * p * l2 * u, nP = (size_t)m * u * l2 * a2;
    float *dR, *dA, *dW, *dP;
    int *dl, *dflag;
    PG_CHECK_HIP(hipMalloc(&dR, nR * nbatch * sizeof(float)));
    PG_CHECK_HIP(hipMalloc(&dA, nA * nbatch * sizeof(float)));
    PG_CHECK_HIP(hipMalloc(&dW, nW * nbatch * sizeof(float)));
    PG_CHECK_HIP(hipMalloc(&dP, nP * nbatch * sizeof(float)));
    PG_CHECK_HIP(hipMalloc(&dl, 3 * (size_t)nbatch * sizeof(int)));
    PG_CHECK_HIP(hipMalloc(&dflag, nbatch * sizeof(int)));
    PG_CHECK_HIP(hipMemcpy(dR, R, nR * nbatch * sizeof(float), hipMemcpyHostToDevice));
    PG_CHECK_HIP(hipMemcpy(dA, A, nA * nbatch * sizeof(float), hipMemcpyHostToDevice));
    PG_CHECK_HIP(hipMemcpy(dW, W, nW * nbatch * sizeof(float), hipMemcpyHostToDevice));
    PG_CHECK_HIP(hipMemset(dP, 0, nP * nbatch * sizeof(float)));
    std::vector<int> hl(3 * (size_t)nbatch);      // [m_live | a_live | a2_live] as three arrays
    for (int b = 0; b < nbatch; ++b)
      for (int q = 0; q < 3; ++q) hl[(size_t)q * nbatch + b] = live[3 * b + q];
    PG_CHECK_HIP(hipMemcpy(dl, hl.data(), hl.size() * sizeof(int), hipMemcpyHostToDevice));
    const int *ml = dl, *al = dl + nbatch, *a2l = dl + 2 * nbatch;
    TGemmDesc gx, gp;
    gx.I[1] = m; gx.I[2] = l; gx.sAi[1] = l * a; gx.sAi[2] = a; gx.sCi[1] = l * p * a2; gx.sCi[2] = p * a2;
    gx.K[2] = a; gx.sAk[2] = 1; gx.sBk[2] = p * a2;
    gx.J[1] = p; gx.J[2] = a2; gx.sBj[1] = a2; gx.sBj[2] = 1; gx.sCj[1] = a2; gx.sCj[2] = 1;
    gx.wA = (long)nR; gx.wB = (long)nA; gx.wC = 0; gx.nbatch = nbatch;
    gx.dI[1].p = ml; gx.dK[2].p = al; gx.dJ[2].p = a2l;
    gp.I[1] = l2; gp.I[2] = u; gp.sAi[1] = u; gp.sAi[2] = 1; gp.sCi[1] = a2; gp.sCi[2] = l2 * a2;
    gp.K[1] = l; gp.K[2] = p; gp.sAk[1] = p * l2 * u; gp.sAk[2] = l2 * u;
    gp.J[1] = m; gp.J[2] = a2; gp.sCj[1] = u * l2 * a2; gp.sCj[2] = 1;
    gp.wA = (long)nW; gp.wC = (long)nP; gp.nbatch = nbatch;
    gp.dJ[1].p = ml; gp.dJ[2].p = a2l;
    TGemmChainMap mp;
    mp.mapK[1] = 2; mp.mapK[2] = 4;
    mp.mapJ[1] = 1; mp.mapJ[2] = 5;
    const char *f64_s = getenv("PEPSGPU_DIAG_CHAIN_F64");      // (read per call) 1: the float64-accumulating form of both stages
    const char *tri_s = getenv("PEPSGPU_DIAG_TRI");            // (read per call) 1: R is a row-compacted triangular factor, its zero blocks are skipped
    PG_REQUIRE(tgemm_chain_launch(0, gx, gp, mp, dR, dA, dW, dP, dflag, 1, tri_s && atoi(tri_s) ? 1 : 0, f64_s && atoi(f64_s) ? 1 : 0,
                                  tri_s && atoi(tri_s) ? 1 : 0), 1, "chain launch refused");
    PG_CHECK_HIP(hipDeviceSynchronize());
    PG_CHECK_HIP(hipMemcpy(P_out, dP, nP * nbatch * sizeof(float), hipMemcpyDeviceToHost));
    PG_CHECK_HIP(hipMemcpy(flags_out, dflag, nbatch * sizeof(int), hipMemcpyDeviceToHost));
    (void)hipFree(dR); (void)hipFree(dA); (void)hipFree(dW); (void)hipFree(dP); (void)hipFree(dl); (void)hipFree(dflag);
  });
}

template <typename T>
static void diag_chol_t(const double *G, int n, int nbatch, void *Rout) {
  double *dG;
  T *dR;
  size_t ne = (size_t)n * n * nbatch;
  PG_CHECK_HIP(hipMalloc(&dG, ne * sizeof(double)));
  PG_CHECK_HIP(hipMalloc(&dR, ne * sizeof(T)));
  PG_CHECK_HIP(hipMemcpy(dG, G, ne * sizeof(double), hipMemcpyHostToDevice));
  launch_chol_upper<T>(0, nbatch, dG, (long)n * n, n, dR, (long)n * n, (int *)nullptr);
  PG_CHECK_HIP(hipDeviceSynchronize());
  PG_CHECK_HIP(hipMemcpy(Rout, dR, ne * sizeof(T), hipMemcpyDeviceToHost));
  (void)hipFree(dG); (void)hipFree(dR);
}
// rank-adaptive path of the absorption: chol_lowrank_kernel, then chol_upper_kernel for the flagged walkers
template <typename T>
static void diag_chol_adaptive_t(const double *G, int n, int nbatch, void *Rout, int32_t *mlive) {
  double *dG;
  T *dR;
  int *dml;
  size_t ne = (size_t)n * n * nbatch;
  PG_REQUIRE(n <= 256 * CH_LR_Q, 1, "n too large for the low-rank Cholesky");
  PG_CHECK_HIP(hipMalloc(&dG, ne * sizeof(double)));
  PG_CHECK_HIP(hipMalloc(&dR, ne * sizeof(T)));
  PG_CHECK_HIP(hipMalloc(&dml, nbatch * sizeof(int)));
  PG_CHECK_HIP(hipMemcpy(dG, G, ne * sizeof(double), hipMemcpyHostToDevice));
  PG_CHECK_HIP(hipMemset(dR, 0, ne * sizeof(T)));
  const size_t lsm = chol_lowrank_smem_bytes(n), smem = chol_smem_bytes(n);
  allow_dynamic_lds(reinterpret_cast<const void *>(&chol_lowrank_kernel<T>), lsm);
  allow_dynamic_lds(reinterpret_cast<const void *>(&chol_upper_kernel<T>), smem);
  hipLaunchKernelGGL(chol_lowrank_kernel<T>, dim3(nbatch), dim3(256), lsm, 0, (const double *)dG, (long)n * n, n, dR,
                     (long)n * n, dml);
  PG_CHECK_HIP(hipGetLastError());
  launch_chol_upper<T>(0, nbatch, dG, (long)n * n, n, dR, (long)n * n, dml, 1);
  PG_CHECK_HIP(hipGetLastError());
  PG_CHECK_HIP(hipDeviceSynchronize());
  PG_CHECK_HIP(hipMemcpy(Rout, dR, ne * sizeof(T), hipMemcpyDeviceToHost));
  PG_CHECK_HIP(hipMemcpy(mlive, dml, nbatch * sizeof(int), hipMemcpyDeviceToHost));
  (void)hipFree(dG); (void)hipFree(dR); (void)hipFree(dml);
}
extern "C" int pepsgpu_diag_chol_adaptive(int dtype_out, const double *G, int n, int nbatch, void *R_out, int32_t *mlive_out) {
  return guarded(nullptr, [&]() {
    if (dtype_out == 0) diag_chol_adaptive_t<float>(G, n, nbatch, R_out, mlive_out);
    else diag_chol_adaptive_t<double>(G, n, nbatch, R_out, mlive_out);
  });
}
// gram_chol_lowrank_kernel alone: P = [nbatch][K][n] of type T; mlive_out[b] = -1 where it declines
template <typename T, int KCAP>
static void diag_gram_chol_t(const void *P, int K, int n, int nbatch, void *Rout, int32_t *mlive) {
  T *dP, *dR;
  int *dml;
  PG_REQUIRE(n <= 256, 1, "n too large for the fused low-rank kernel");
  PG_CHECK_HIP(hipMalloc(&dP, (size_t)K * n * nbatch * sizeof(T)));
  PG_CHECK_HIP(hipMalloc(&dR, (size_t)n * n * nbatch * sizeof(T)));
  PG_CHECK_HIP(hipMalloc(&dml, nbatch * sizeof(int)));
  PG_CHECK_HIP(hipMemcpy(dP, P, (size_t)K * n * nbatch * sizeof(T), hipMemcpyHostToDevice));
  PG_CHECK_HIP(hipMemset(dR, 0, (size_t)n * n * nbatch * sizeof(T)));
  const int npass = K > KCAP ? 4 : 1;          // the multi-pass use of the absorption when a pass cannot hold all rows
  // per-walker live row count = rows up to the last non-zero one, as the absorption passes it (mixed counts in one launch:
  // the short first pass, the walkers it hands on and the multi-pass path all run)
  std::vector<int> hk(nbatch, 0);
  for (int b = 0; b < nbatch; ++b)
    for (int r = 0; r < K; ++r) {
      const T *row = (const T *)P + ((size_t)b * K + r) * n;
      for (int c = 0; c < n; ++c)
        if (row[c] != T(0)) { hk[b] = r + 1; break; }
    }
  int *dk;
  PG_CHECK_HIP(hipMalloc(&dk, nbatch * sizeof(int)));
  PG_CHECK_HIP(hipMemcpy(dk, hk.data(), nbatch * sizeof(int), hipMemcpyHostToDevice));
  launch_gram_chol_lowrank<T, KCAP>(0, nbatch, (const T *)dP, (long)K * n, n, (const int *)dk, 1, K, dR, (long)n * n, dml, 1,
                                    (const int *)nullptr, npass);
  PG_CHECK_HIP(hipDeviceSynchronize());
  (void)hipFree(dk);
  PG_CHECK_HIP(hipDeviceSynchronize());
  PG_CHECK_HIP(hipMemcpy(Rout, dR, (size_t)n * n * nbatch * sizeof(T), hipMemcpyDeviceToHost));
  PG_CHECK_HIP(hipMemcpy(mlive, dml, nbatch * sizeof(int), hipMemcpyDeviceToHost));
  (void)hipFree(dP); (void)hipFree(dR); (void)hipFree(dml);
}
extern "C" int pepsgpu_diag_gram_chol(int dtype, const void *P, int K, int n, int nbatch, void *R_out, int32_t *mlive_out) {
  return guarded(nullptr, [&]() {
    if (dtype == 0) diag_gram_chol_t<float, 96>(P, K, n, nbatch, R_out, mlive_out);
    else diag_gram_chol_t<double, 48>(P, K, n, nbatch, R_out, mlive_out);
  });
}
// gram_cols_f64_kernel alone: P = [nbatch][K][n] of type T, klive[b] (optional) = live rows; G_out = [nbatch][n][n] float64,
// blocks on or above the diagonal (64 x 64 granularity) written, the rest left at zero
template <typename T>
static void diag_gram_cols_t(const void *P, int K, int n, int nbatch, const int32_t *klive, double *G) {
  T *dP;
  double *dG;
  int *dk = nullptr;
  PG_CHECK_HIP(hipMalloc(&dP, (size_t)K * n * nbatch * sizeof(T)));
  PG_CHECK_HIP(hipMalloc(&dG, (size_t)n * n * nbatch * sizeof(double)));
  PG_CHECK_HIP(hipMemcpy(dP, P, (size_t)K * n * nbatch * sizeof(T), hipMemcpyHostToDevice));
  PG_CHECK_HIP(hipMemset(dG, 0, (size_t)n * n * nbatch * sizeof(double)));
  if (klive) {
    PG_CHECK_HIP(hipMalloc(&dk, nbatch * sizeof(int)));
    PG_CHECK_HIP(hipMemcpy(dk, klive, nbatch * sizeof(int), hipMemcpyHostToDevice));
  }
  launch_gram_cols_f64<T>(0, nbatch, (const T *)dP, (long)K * n, n, n, (const int *)dk, 1, K, dG, (const int *)nullptr, 1,
                          (const int *)nullptr, (unsigned long long *)nullptr, (unsigned long long *)nullptr);
  PG_CHECK_HIP(hipDeviceSynchronize());
  PG_CHECK_HIP(hipMemcpy(G, dG, (size_t)n * n * nbatch * sizeof(double), hipMemcpyDeviceToHost));
  (void)hipFree(dP); (void)hipFree(dG); if (dk) (void)hipFree(dk);
}
extern "C" int pepsgpu_diag_gram_cols(int dtype, const void *P, int K, int n, int nbatch, const int32_t *klive, double *G_out) {
  return guarded(nullptr, [&]() {
    PG_REQUIRE(K >= 1 && n >= 1 && n <= 256 && nbatch >= 1, 1, "bad sizes");
    if (dtype == 0) diag_gram_cols_t<float>(P, K, n, nbatch, klive, G_out); else diag_gram_cols_t<double>(P, K, n, nbatch, klive, G_out);
  });
}
// gram_rows_f64_kernel alone: G[b] = M[b] M[b]^T for the first nrows[b] rows of M (n x K f32, K % 16 == 0), upper 64 x 64 blocks
extern "C" int pepsgpu_diag_gram_rows(const float *M, int n, int K, int nbatch, const int32_t *nrows, double *G_out) {
  return guarded(nullptr, [&]() {
    PG_REQUIRE(n >= 1 && n <= 256 && K >= 16 && K % 16 == 0 && nbatch >= 1 && nrows, 1, "bad sizes");
    float *dM; double *dG; int *dn;
    const size_t ne = (size_t)n * K * nbatch;
    PG_CHECK_HIP(hipMalloc(&dM, ne * sizeof(float)));
    PG_CHECK_HIP(hipMalloc(&dG, (size_t)n * n * nbatch * sizeof(double)));
    PG_CHECK_HIP(hipMalloc(&dn, nbatch * sizeof(int)));
    PG_CHECK_HIP(hipMemcpy(dM, M, ne * sizeof(float), hipMemcpyHostToDevice));
    PG_CHECK_HIP(hipMemcpy(dn, nrows, nbatch * sizeof(int), hipMemcpyHostToDevice));
    PG_CHECK_HIP(hipMemset(dG, 0, (size_t)n * n * nbatch * sizeof(double)));
    launch_gram_rows_f64<float>(0, nbatch, dM, (long)n * K, K, n, dn, dG, (long)n * n, n, nullptr, nullptr, nullptr);
    PG_CHECK_HIP(hipDeviceSynchronize());
    PG_CHECK_HIP(hipMemcpy(G_out, dG, (size_t)n * n * nbatch * sizeof(double), hipMemcpyDeviceToHost));
    (void)hipFree(dM); (void)hipFree(dG); (void)hipFree(dn);
  });
}
// mgemm_dense_kernel alone: M[b] = R[b] Tt[b] with per-walker live rows / a / k2 (nullable), Tt stored [la][u][k2] or [la][k2][u]
extern "C" int pepsgpu_diag_mgemm_dense(const float *R, const float *Tt, int m, int la, int a_dim, int u_dim, int k2_dim, int tt_u_inner,
                                        int nbatch, const int32_t *m_live, const int32_t *a_live, const int32_t *k2_live, float *M_out) {
  return guarded(nullptr, [&]() {
    const int uk = u_dim * k2_dim;
    float *dR, *dT, *dM;
    int *dl[3] = {nullptr, nullptr, nullptr};
    const int32_t *hl[3] = {m_live, a_live, k2_live};
    PG_CHECK_HIP(hipMalloc(&dR, (size_t)m * la * nbatch * sizeof(float)));
    PG_CHECK_HIP(hipMalloc(&dT, (size_t)la * uk * nbatch * sizeof(float)));
    PG_CHECK_HIP(hipMalloc(&dM, (size_t)m * uk * nbatch * sizeof(float)));
    PG_REQUIRE(mgemm_dense_ok(m, la, a_dim, u_dim, k2_dim, (long)m * la, (long)la * uk, dR, dT), 1, "shape not supported by mgemm_dense_kernel");
    PG_CHECK_HIP(hipMemcpy(dR, R, (size_t)m * la * nbatch * sizeof(float), hipMemcpyHostToDevice));
    PG_CHECK_HIP(hipMemcpy(dT, Tt, (size_t)la * uk * nbatch * sizeof(float), hipMemcpyHostToDevice));
    PG_CHECK_HIP(hipMemset(dM, 0xFF, (size_t)m * uk * nbatch * sizeof(float)));      // NaN pattern: rows beyond the live count stay untouched
    for (int q = 0; q < 3; ++q)
      if (hl[q]) {
        PG_CHECK_HIP(hipMalloc(&dl[q], nbatch * sizeof(int)));
        PG_CHECK_HIP(hipMemcpy(dl[q], hl[q], nbatch * sizeof(int), hipMemcpyHostToDevice));
      }
    launch_mgemm_dense(0, nbatch, dR, (long)m * la, dT, (long)la * uk, dM, (long)m * uk, m, la, a_dim, u_dim, k2_dim, tt_u_inner, dl[0], 1, dl[1],
                       dl[2], nullptr, nullptr, getenv("PEPSGPU_DIAG_TRI") && atoi(getenv("PEPSGPU_DIAG_TRI")) ? 1 : 0);
    PG_CHECK_HIP(hipDeviceSynchronize());
    PG_CHECK_HIP(hipMemcpy(M_out, dM, (size_t)m * uk * nbatch * sizeof(float), hipMemcpyDeviceToHost));
    (void)hipFree(dR); (void)hipFree(dT); (void)hipFree(dM);
    for (int q = 0; q < 3; ++q) if (dl[q]) (void)hipFree(dl[q]);
  });
}
// PEPSGPU_CG_STATS=1: read and reset the per-phase counters of colgram_dense_kernel (trunc_mid.h: cg_stats_dev), launches of >= 1024 walkers
extern "C" int pepsgpu_diag_cg_stats(double *out16) {
  return guarded(nullptr, [&]() {
    unsigned long long h[16] = {0};
    unsigned long long *d = cg_stats_dev();
    if (d) {
      PG_CHECK_HIP(hipDeviceSynchronize());
      PG_CHECK_HIP(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
      PG_CHECK_HIP(hipMemset(d, 0, sizeof(h)));
    }
    for (int i = 0; i < 16; ++i) out16[i] = (double)h[i];
  });
}
// The two LDS-resident Gram + Cholesky kernels alone (trunc_mid.h), f32:
//   which = 0: mid_gram_chol_kernel    X = [nbatch][n][K]  (rows of M),   R^T R = X X^T  (n x n),  nlive[b] = live rows of X
//   which = 1: colgram_dense_kernel    X = [nbatch][K][n]  (rows of P),   R^T R = X^T X  (n x n),  nlive[b] = live rows of X (<= K)
// R_out = [nbatch][n][n] (rows beyond mlive_out[b] untouched = NaN pattern), mlive_out[b] = rows of the factor.
extern "C" int pepsgpu_diag_lds_gram_chol(int which, const float *X, int n, int K, int nbatch, const int32_t *nlive, float *R_out,
                                          int32_t *mlive_out) {
  return guarded(nullptr, [&]() {
    PG_REQUIRE((which == 0 || which == 1) && n >= 1 && n <= 128 && K >= 1 && nbatch >= 1 && nlive, 1, "bad sizes");
    float *dX, *dR; int *dn, *dm;
    const size_t ne = (size_t)n * K * nbatch, nr = (size_t)n * n * nbatch;
    PG_CHECK_HIP(hipMalloc(&dX, ne * sizeof(float)));
    PG_CHECK_HIP(hipMalloc(&dR, nr * sizeof(float)));
    PG_CHECK_HIP(hipMalloc(&dn, nbatch * sizeof(int)));
    PG_CHECK_HIP(hipMalloc(&dm, nbatch * sizeof(int)));
    PG_CHECK_HIP(hipMemcpy(dX, X, ne * sizeof(float), hipMemcpyHostToDevice));
    PG_CHECK_HIP(hipMemcpy(dn, nlive, nbatch * sizeof(int), hipMemcpyHostToDevice));
    PG_CHECK_HIP(hipMemset(dR, 0xFF, nr * sizeof(float)));
    PG_CHECK_HIP(hipMemset(dm, 0xFF, nbatch * sizeof(int)));
    if (which == 0)
      launch_mid_gram_chol<float>(0, nbatch, dX, (long)n * K, K, dn, nullptr, n, dR, (long)n * n, dm);
    else
      launch_colgram_chol<float>(0, nbatch, dX, (long)n * K, n, dn, 1, K, dR, (long)n * n, dm, 1, nullptr, -4, true);
    PG_CHECK_HIP(hipDeviceSynchronize());
    PG_CHECK_HIP(hipMemcpy(R_out, dR, nr * sizeof(float), hipMemcpyDeviceToHost));
    PG_CHECK_HIP(hipMemcpy(mlive_out, dm, nbatch * sizeof(int), hipMemcpyDeviceToHost));
    (void)hipFree(dX); (void)hipFree(dR); (void)hipFree(dn); (void)hipFree(dm);
  });
}
// The first compression of the dense truncation route as it runs (round 6): G = X X^T on the i8 matrix cores with both triangles written
// (gram_i8.h, sym) + the diagonally pivoted factorisation stopped after kcap rows (chol_pivot.h).  X = [nbatch][n][K] f32 (rows of M),
// 128 < n <= 256, K a multiple of 16; nlive[b] = live rows.  R_out = [nbatch][kcap][n] (rows in pivot order; rows beyond mlive_out[b]
// untouched = NaN pattern).
extern "C" int pepsgpu_diag_chol_pivot(const float *X, int n, int K, int nbatch, const int32_t *nlive, int kcap, float *R_out,
                                       int32_t *mlive_out) {
  return guarded(nullptr, [&]() {
    PG_REQUIRE(n > 128 && n <= 256 && K >= 16 && K % 16 == 0 && nbatch >= 1 && nlive && kcap >= 1 && kcap <= 64, 1, "bad sizes");
    float *dX, *dR; int *dn, *dm; double *dG;
    const size_t ne = (size_t)n * K * nbatch, nr = (size_t)64 * n * nbatch;
    PG_CHECK_HIP(hipMalloc(&dX, ne * sizeof(float)));
    PG_CHECK_HIP(hipMalloc(&dR, nr * sizeof(float)));
    PG_CHECK_HIP(hipMalloc(&dG, (size_t)n * n * nbatch * sizeof(double)));
    PG_CHECK_HIP(hipMalloc(&dn, nbatch * sizeof(int)));
    PG_CHECK_HIP(hipMalloc(&dm, nbatch * sizeof(int)));
    PG_CHECK_HIP(hipMemcpy(dX, X, ne * sizeof(float), hipMemcpyHostToDevice));
    PG_CHECK_HIP(hipMemcpy(dn, nlive, nbatch * sizeof(int), hipMemcpyHostToDevice));
    PG_CHECK_HIP(hipMemset(dR, 0xFF, nr * sizeof(float)));
    PG_CHECK_HIP(hipMemset(dG, 0xFF, (size_t)n * n * nbatch * sizeof(double)));
    PG_CHECK_HIP(hipMemset(dm, 0xFF, nbatch * sizeof(int)));
    PG_REQUIRE(gram_rows_i8_ok(dX, n), 1, "the i8 row Gram does not take this shape");
    launch_gram_rows_f64<float>(0, nbatch, dX, (long)n * K, K, n, dn, dG, (long)n * n, n, nullptr, nullptr, nullptr, 1);
    launch_chol_pivot<float>(0, nbatch, dG, (long)n * n, n, dR, (long)64 * n, dm, n, dn, 1, nullptr, kcap);
    PG_CHECK_HIP(hipDeviceSynchronize());
    std::vector<float> h(nr);
    PG_CHECK_HIP(hipMemcpy(h.data(), dR, nr * sizeof(float), hipMemcpyDeviceToHost));
    for (int b = 0; b < nbatch; ++b)
      memcpy(R_out + (size_t)b * kcap * n, h.data() + (size_t)b * 64 * n, sizeof(float) * (size_t)kcap * n);
    PG_CHECK_HIP(hipMemcpy(mlive_out, dm, nbatch * sizeof(int), hipMemcpyDeviceToHost));
    (void)hipFree(dX); (void)hipFree(dR); (void)hipFree(dG); (void)hipFree(dn); (void)hipFree(dm);
  });
}
// rows_qr_kernel alone (round 6): X = [nbatch][k][len] f32 nearly orthogonal rows by decreasing norm, klive[b] = rows that exist;
// V_out = [nbatch][k][len] orthonormal rows spanning the same space (live rows first, the rest zero), klive_out[b] = their count.
extern "C" int pepsgpu_diag_rows_qr(const float *X, int k, int len, int nbatch, const int32_t *klive, float *V_out, int32_t *klive_out) {
  return guarded(nullptr, [&]() {
    PG_REQUIRE(rows_qr_ok(k, len) && nbatch >= 1 && klive, 1, "bad sizes");
    float *dX, *dV; int *dn, *dm;
    const size_t ne = (size_t)k * len * nbatch;
    PG_CHECK_HIP(hipMalloc(&dX, ne * sizeof(float)));
    PG_CHECK_HIP(hipMalloc(&dV, ne * sizeof(float)));
    PG_CHECK_HIP(hipMalloc(&dn, nbatch * sizeof(int)));
    PG_CHECK_HIP(hipMalloc(&dm, nbatch * sizeof(int)));
    PG_CHECK_HIP(hipMemcpy(dX, X, ne * sizeof(float), hipMemcpyHostToDevice));
    PG_CHECK_HIP(hipMemcpy(dn, klive, nbatch * sizeof(int), hipMemcpyHostToDevice));
    PG_CHECK_HIP(hipMemset(dV, 0xFF, ne * sizeof(float)));
    PG_CHECK_HIP(hipMemset(dm, 0xFF, nbatch * sizeof(int)));
    launch_rows_qr(0, nbatch, dX, (long)k * len, k, len, dn, dV, (long)k * len, dm, nullptr);
    PG_CHECK_HIP(hipDeviceSynchronize());
    PG_CHECK_HIP(hipMemcpy(V_out, dV, ne * sizeof(float), hipMemcpyDeviceToHost));
    PG_CHECK_HIP(hipMemcpy(klive_out, dm, nbatch * sizeof(int), hipMemcpyDeviceToHost));
    (void)hipFree(dX); (void)hipFree(dV); (void)hipFree(dn); (void)hipFree(dm);
  });
}
// The device's SuwaTodoStateUpdate alone (round 6): a chain of `steps` updates on one weight vector (n <= 16 states), fed with the
// raw 32-bit outputs of the caller's std::mt19937 (two per step, the reference's long double draw); out_chain[s] = the state after step s.
extern "C" int pepsgpu_diag_suwa_todo(const double *weights, int n, int init, const uint32_t *words, int steps, int32_t *out_chain) {
  return guarded(nullptr, [&]() {
    PG_REQUIRE(weights && words && out_chain && n >= 1 && n <= SW_MAXC && init >= 0 && init < n && steps >= 1, 1, "bad arguments");
    double *dw; unsigned *dd; int *dout;
    PG_CHECK_HIP(hipMalloc(&dw, n * sizeof(double)));
    PG_CHECK_HIP(hipMalloc(&dd, 2 * (size_t)steps * sizeof(unsigned)));
    PG_CHECK_HIP(hipMalloc(&dout, (size_t)steps * sizeof(int)));
    PG_CHECK_HIP(hipMemcpy(dw, weights, n * sizeof(double), hipMemcpyHostToDevice));
    PG_CHECK_HIP(hipMemcpy(dd, words, 2 * (size_t)steps * sizeof(unsigned), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(sweep_suwa_todo_chain_kernel, dim3(1), dim3(64), 0, 0, (const double *)dw, n, init, (const unsigned *)dd, steps, dout);
    PG_CHECK_HIP(hipGetLastError());
    PG_CHECK_HIP(hipDeviceSynchronize());
    PG_CHECK_HIP(hipMemcpy(out_chain, dout, (size_t)steps * sizeof(int), hipMemcpyDeviceToHost));
    (void)hipFree(dw); (void)hipFree(dd); (void)hipFree(dout);
  });
}
extern "C" int pepsgpu_diag_chol(int dtype_out, const double *G, int n, int nbatch, void *R_out) {
  return guarded(nullptr, [&]() {
    if (dtype_out == 0) diag_chol_t<float>(G, n, nbatch, R_out); else diag_chol_t<double>(G, n, nbatch, R_out);
  });
}

template <typename T>
static void diag_jacobi_t(void *M, int m, int len, int nbatch, int k, void *Vt, void *S, int force_global, int *sweeps) {
  T *dM, *dV, *dS;
  int *dsw;
  size_t ne = (size_t)m * len * nbatch;
  PG_CHECK_HIP(hipMalloc(&dM, ne * sizeof(T)));
  PG_CHECK_HIP(hipMalloc(&dV, (size_t)k * len * nbatch * sizeof(T)));
  PG_CHECK_HIP(hipMalloc(&dS, (size_t)k * nbatch * sizeof(T)));
  PG_CHECK_HIP(hipMalloc(&dsw, nbatch * sizeof(int)));
  PG_CHECK_HIP(hipMemcpy(dM, M, ne * sizeof(T), hipMemcpyHostToDevice));
  const size_t need = sizeof(T) * (size_t)m * (len | 1);
  const int use_lds = !force_global && need <= JACOBI_LDS_MAX;
  if (use_lds) allow_dynamic_lds(reinterpret_cast<const void *>(&jacobi_rows_kernel<T>), need);
  if (sizeof(T) == 4 && force_global == 2) {
    PG_REQUIRE(m <= 256 && len <= 256, 1, "register Jacobi handles up to 256 x 256");
    hipLaunchKernelGGL(jacobi_rows_reg256_kernel, dim3(nbatch), dim3(512), 0, 0, (float *)dM, (long)m * len, m, len, len,
                       40, dsw, (const int *)nullptr, 1, 0);
  } else if (sizeof(T) == 4 && force_global >= 5 && force_global <= 8) {
    // 16-lanes-per-row tournament: 5 = four waves (128 rows), 6 = two (64), rows up to 128 long; 7 / 8 = the same with rows up to 256 long
    const int wide = force_global >= 7, four = force_global == 5 || force_global == 7;
    PG_REQUIRE(m <= (four ? 128 : 64) && len <= (wide ? 256 : 128), 1, "grouped tournament handles up to 128 (64) x 128 (256)");
    if (force_global == 5) launch_jacobi_grp<4, 8>(0, nbatch, (float *)dM, (long)m * len, m, len, len, 40, dsw, (const int *)nullptr, 1, 0);
    else if (force_global == 6) launch_jacobi_grp<2, 8>(0, nbatch, (float *)dM, (long)m * len, m, len, len, 40, dsw, (const int *)nullptr, 1, 0);
    else if (force_global == 7) launch_jacobi_grp<4, 16>(0, nbatch, (float *)dM, (long)m * len, m, len, len, 40, dsw, (const int *)nullptr, 1, 0);
    else launch_jacobi_grp<2, 16>(0, nbatch, (float *)dM, (long)m * len, m, len, len, 40, dsw, (const int *)nullptr, 1, 0);
  } else if (sizeof(T) == 4 && (force_global == 9 || force_global == 10)) {
    // the same tournament with ONE wave per walker (up to 32 rows): 9 = rows up to 128 long, 10 = up to 256 (polish of 17..32 rows,
    // every walker of a small batch)
    PG_REQUIRE(m <= 32 && len <= (force_global == 9 ? 128 : 256), 1, "one-wave grouped tournament handles up to 32 x 128 (256)");
    if (force_global == 9) launch_jacobi_grp<1, 8>(0, nbatch, (float *)dM, (long)m * len, m, len, len, 40, dsw, (const int *)nullptr, 1, 0);
    else launch_jacobi_grp<1, 16>(0, nbatch, (float *)dM, (long)m * len, m, len, len, 40, dsw, (const int *)nullptr, 1, 0);
  } else if (sizeof(T) == 4 && force_global == 3) {   // one-wave-per-walker kernel (up to 32 x 256)
    PG_REQUIRE(m <= JR_SMALL_ROWS && len <= 256, 1, "small Jacobi handles up to 32 x 256");
    // per-walker live row count = rows up to the last non-zero one (mixed counts inside a launch, as in the absorption)
    std::vector<int> hm(nbatch, 0);
    for (int b = 0; b < nbatch; ++b)
      for (int r = 0; r < m; ++r) {
        const T *row = (const T *)M + ((size_t)b * m + r) * len;
        for (int c = 0; c < len; ++c)
          if (row[c] != T(0)) { hm[b] = r + 1; break; }
      }
    int *dm;
    PG_CHECK_HIP(hipMalloc(&dm, nbatch * sizeof(int)));
    PG_CHECK_HIP(hipMemcpy(dm, hm.data(), nbatch * sizeof(int), hipMemcpyHostToDevice));
    if (m <= JR_BR && len <= 64)      // as the absorption: short rows -> four walkers per wave, 16 lanes x 4 columns
      hipLaunchKernelGGL((jacobi_rows_tiny4_kernel<4>), dim3((nbatch + 15) / 16), dim3(256), 0, 0, (float *)dM, (long)m * len, m, len,
                         len, 40, dsw, (const int *)dm, 1, nbatch, JrSelect());
    else if (m <= JR_BR && len <= 128)   // ... 16 lanes x 8 columns
      hipLaunchKernelGGL((jacobi_rows_tiny4_kernel<8>), dim3((nbatch + 15) / 16), dim3(256), 0, 0, (float *)dM, (long)m * len, m, len,
                         len, 40, dsw, (const int *)dm, 1, nbatch, JrSelect());
    else if (m <= JR_BR)     // <= 16 rows -> tiny kernel, 17..32 -> small kernel
      hipLaunchKernelGGL(jacobi_rows_tiny_kernel, dim3((nbatch + 3) / 4), dim3(256), 0, 0, (float *)dM, (long)m * len, m, len,
                         len, 40, dsw, (const int *)dm, 1, nbatch);
    else
      hipLaunchKernelGGL(jacobi_rows_small_kernel, dim3((nbatch + 3) / 4), dim3(256), 0, 0, (float *)dM, (long)m * len, m, len,
                         len, 40, dsw, (const int *)dm, 1, nbatch, 0);
    PG_CHECK_HIP(hipDeviceSynchronize());
    (void)hipFree(dm);
  } else {
    hipLaunchKernelGGL(jacobi_rows_kernel<T>, dim3(nbatch), dim3(1024), use_lds ? need : 0, 0, dM, (long)m * len, m, len,
                       len, 40, use_lds, dsw, (const int *)nullptr, 1);
  }
  PG_CHECK_HIP(hipGetLastError());
  hipLaunchKernelGGL(select_rows_kernel<T>, dim3(nbatch), dim3(256), 0, 0, (const T *)dM, (long)m * len, m, len, len, k,
                     dV, (long)k * len, dS, (long)k, (const int *)nullptr, 1);
  PG_CHECK_HIP(hipGetLastError());
  PG_CHECK_HIP(hipDeviceSynchronize());
  PG_CHECK_HIP(hipMemcpy(M, dM, ne * sizeof(T), hipMemcpyDeviceToHost));
  PG_CHECK_HIP(hipMemcpy(Vt, dV, (size_t)k * len * nbatch * sizeof(T), hipMemcpyDeviceToHost));
  PG_CHECK_HIP(hipMemcpy(S, dS, (size_t)k * nbatch * sizeof(T), hipMemcpyDeviceToHost));
  if (sweeps) PG_CHECK_HIP(hipMemcpy(sweeps, dsw, nbatch * sizeof(int), hipMemcpyDeviceToHost));
  (void)hipFree(dM); (void)hipFree(dV); (void)hipFree(dS); (void)hipFree(dsw);
}
extern "C" int pepsgpu_diag_jacobi(int dtype, void *M, int m, int len, int nbatch, int k, void *Vt, void *S, int force_global,
                        int *sweeps) {
  return guarded(nullptr, [&]() {
    PG_REQUIRE(m <= 1024 && k <= m, 1, "bad sizes");
    if (dtype == 0) diag_jacobi_t<float>(M, m, len, nbatch, k, Vt, S, force_global, sweeps);
    else diag_jacobi_t<double>(M, m, len, nbatch, k, Vt, S, force_global, sweeps);
  });
}
