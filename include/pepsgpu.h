/* pepsgpu.h -- C ABI of the MI355X boundary-MPS contraction library (libpepsgpu.so).
 *
 * The reference (QuantumLiquids/PEPS v0.1.0) has NO FFI: its seam is the compile-time
 * `ContractorT` template parameter (include/qlpeps/vmc_basic/wave_function_component.h:136-140,
 * concepts :24-87) filled by `BMPSContractor<TenElemT,QNT>`
 * (include/qlpeps/two_dim_tn/tensor_network_2d/bmps/bmps_contractor.h:187-1027).
 * Each entry point below replaces one method of that class (cited per function), batched over
 * the Monte-Carlo walkers of one GPU: where the reference holds one TensorNetwork2D + one
 * BMPSContractor per MPI rank, a context holds `n` walkers that advance in lockstep.
 *
 * Conventions
 *   - plain pointers and sizes only; host buffers are caller-owned; no exceptions cross the ABI.
 *   - every call returns 0 on success or a PEPSGPU_E* code; pepsgpu_last_error(ctx) gives the text.
 *     The host wrappers re-raise them as the reference's C++ exceptions (bmps_impl.h:839-843,
 *     bmps_contractor.h:221-226, tensor_network_2d_basic_impl.h:37-66).
 *   - positions: LEFT=0, DOWN=1, RIGHT=2, UP=3 (include/qlpeps/basic.h:58-63);
 *     bond orientation: HORIZONTAL=0, VERTICAL=1 (basic.h:19-22).
 *   - site tensors have leg order (L, D, R, U) (tensor_network_2d.h:39-45).
 *   - amplitudes are returned as float64 (the device keeps per-environment log-scales).
 */
#ifndef PEPSGPU_H
#define PEPSGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pepsgpu_ctx pepsgpu_ctx;

enum {
  PEPSGPU_OK = 0,
  PEPSGPU_EINVAL = 1,     /* std::invalid_argument */
  PEPSGPU_EHIP = 2,       /* device/runtime failure */
  PEPSGPU_ESTATE = 3,     /* std::logic_error: environment / parameters not ready */
  PEPSGPU_ERANGE = 4,     /* std::out_of_range: configuration value >= physical dim */
  PEPSGPU_EEMPTY = 5      /* std::runtime_error: empty (zero) tensor, bmps_impl.h:839-843 */
};
/* element type of the device tensors (TenElemT of the reference: QLTEN_Double, QLTEN_Complex).  PEPSGPU_C128 = complex
 * float64 (std::complex<double> layout, interleaved re / im).  For a complex context EVERY scalar or tensor the calls
 * below return through a `double *` is an interleaved (re, im) pair: out_amp has 2 n doubles, a hole 2 D^4 per walker, a
 * BMPS tensor 2 x elements; pepsgpu_state_upload additionally accepts host_dtype = PEPSGPU_C128.  The complex type covers
 * SVD and (since round 5) variational compression, every contraction / trace / hole entry point, the walker calls, the gradient
 * accumulation (pepsgpu_grad_* below, with the reference's conjugations: psi, eloc and the accumulators are interleaved pairs) and
 * the SR / MinSR family (pepsgpu_sr_*) and (round 6) the device-side sweep slices; the device-side ENERGY slice
 * (pepsgpu_nn_exchange_slice) is real only and returns PEPSGPU_EINVAL. */
enum { PEPSGPU_F32 = 0, PEPSGPU_F64 = 1, PEPSGPU_C128 = 3 };
enum { PEPSGPU_LEFT = 0, PEPSGPU_DOWN = 1, PEPSGPU_RIGHT = 2, PEPSGPU_UP = 3 };
enum { PEPSGPU_HORIZONTAL = 0, PEPSGPU_VERTICAL = 1 };
enum { PEPSGPU_SVD_COMPRESS = 0, PEPSGPU_VARIATION2SITE = 1, PEPSGPU_VARIATION1SITE = 2 };   /* CompressMPSScheme, bmps.h:31-35 */

/* BMPSContractor(rows, cols) + SetTruncateParams(BMPSTruncateParams{D_min, D_max, trunc_err, scheme})
 * (bmps_contractor.h:187-226, bmps.h:47-98).  dtype = element type of the device tensors. */
int pepsgpu_ctx_create(pepsgpu_ctx **out, int device, int dtype, int rows, int cols, int D, int phys_dim,
                       int chi_min, int chi_max, double trunc_err, int scheme, int max_walkers);
/* BMPSContractor::SetTruncateParams (bmps_contractor.h:216) with the full BMPSTruncateParams (bmps.h:47-98):
 * D_min, D_max, trunc_err, scheme and, for the variational schemes, convergence_tol and iter_max
 * (BMPS::MultiplyMPO2SiteVariationalCompress_ / 1Site, bmps_impl.h:864-1172; bosonic only as in the reference).
 * Takes effect at the next absorption; existing BMPS stacks are kept.  A context created with a variational scheme
 * uses convergence_tol = 1e-10, iter_max = 10 until this is called. */
int pepsgpu_set_truncate_params(pepsgpu_ctx *ctx, int chi_min, int chi_max, double trunc_err, int scheme,
                                double convergence_tol, int iter_max);
void pepsgpu_ctx_destroy(pepsgpu_ctx *ctx);
const char *pepsgpu_last_error(pepsgpu_ctx *ctx);

/* SplitIndexTPS (two_dim_tn/tps/split_index_tps.h:80-606) as one flat host buffer
 * [row][col][s][L][D][R][U], every leg zero-padded to D; host_dtype = PEPSGPU_F32 / PEPSGPU_F64.
 * Replaces the MPI_Bcast of the state (mc_energy_grad_evaluator.h:161). */
int pepsgpu_state_upload(pepsgpu_ctx *ctx, const void *sitps_flat, int host_dtype);

/* TensorNetwork2D(sitps, config) + BMPSContractor::Init(tn) for n walkers
 * (tensor_network_2d_basic_impl.h:24-74, bmps_contractor_init.h:25-70).  configs = [n][rows][cols]. */
int pepsgpu_walkers_set_configs(pepsgpu_ctx *ctx, int n, const int32_t *configs);
int pepsgpu_walkers_get_configs(pepsgpu_ctx *ctx, int32_t *configs_out);
int pepsgpu_n_walkers(pepsgpu_ctx *ctx);

/* BMPS stacks -- bmps_contractor_grow.h */
int pepsgpu_grow_bmps_step(pepsgpu_ctx *ctx, int pos);          /* GrowBMPSStep(tn,pos)      :32-47   */
int pepsgpu_grow_full_bmps(pepsgpu_ctx *ctx, int pos);          /* GrowFullBMPS              :49-86   */
int pepsgpu_grow_bmps_for_row(pepsgpu_ctx *ctx, int row);       /* GrowBMPSForRow            :88-104  */
int pepsgpu_grow_bmps_for_col(pepsgpu_ctx *ctx, int col);       /* GrowBMPSForCol            :106-122 */
int pepsgpu_shift_bmps_window(pepsgpu_ctx *ctx, int pos);       /* ShiftBMPSWindow           :143-148 */
int pepsgpu_delete_inner_bmps(pepsgpu_ctx *ctx, int pos);       /* DeleteInnerBMPS  bmps_contractor.h:320-324 */
/* BMPSWalker (bmps/impl/bmps_walker.h:  a BMPS forked out of a stack, evolved row by row and contracted against an
 * explicitly named environment of the opposite stack).  The stack itself is the walker here: park hides the levels
 * above keep_levels (no copy) so that the BTen / trace calls see level keep_levels-1 as the top; unpark drops what was
 * grown meanwhile and restores the hidden levels. */
int pepsgpu_bmps_park(pepsgpu_ctx *ctx, int pos, int keep_levels);
int pepsgpu_bmps_unpark(pepsgpu_ctx *ctx, int pos);
/* One row (orientation = PEPSGPU_HORIZONTAL, slice = row) or column (PEPSGPU_VERTICAL, slice = column) of the Monte-Carlo sweep with
 * the nearest-neighbour EXCHANGE updater, entirely on the device: replaces, for that slice, the loop body of
 * MCUpdateSquareNNUpdateBaseOBC::operator() (square_nn_updater.h:41-55 / :62-76: InitBTen, GrowFullBTen(.., 2, true), then per bond
 * TwoSiteNNUpdateLocalImpl + ShiftBTenWindow) with MCUpdateSquareNNExchangeOBC::TwoSiteNNUpdateLocalImpl (:142-189: skip equal
 * spins, ReplaceNNSiteTrace of the exchanged pair, Metropolis on |psi'/psi|^2, UpdateLocal).  The BMPS pair of the slice must be in
 * place as for pepsgpu_init_bten (the caller keeps doing GenerateBMPSApproach / ShiftBMPSWindow between slices).
 *   uniforms        [n][n_uniform], n_uniform >= slice length - 1: per walker the NEXT deviates of its std::mt19937 +
 *                   uniform_real_distribution<double>(0, 1) stream, in drawing order; a walker consumes one only where the
 *                   reference draws one (spins differ and |psi'| < |psi|), consumed_out[w] says how many -- the chain is the
 *                   reference's chain when the caller pops exactly those from its queue;
 *   amplitude_inout [n] psi of every walker before / after the slice;  accepted_out [n] accepted exchanges;
 *   slice_states_out [n][slice length] (may be NULL) the configuration along the slice after the pass.
 * Every element type (round 6; a PEPSGPU_C128 context takes and returns interleaved (re, im) amplitudes, the test is on |psi'| / |psi|). */
int pepsgpu_sweep_slice_exchange(pepsgpu_ctx *ctx, int orientation, int slice, int n_uniform, const double *uniforms,
                                 double *amplitude_inout, int32_t *consumed_out, int32_t *accepted_out, int32_t *slice_states_out);
/* The same with the move given as a table (round 6): pair_table [phys_dim^2][2] (phys_dim = that of the context), the pair of states the
 * "exchange" proposes for the states (a, b) of a bond = pair_table[a * phys_dim + b]; NULL = the swap (b, a).  A fermionic state lives on
 * the device as EXTENDED states (state + d * variant: mode order x parity of the fermions before the site); the exchange of two sites
 * adjacent in the mode order changes their variants by a rule local in (a, b) -- TPSWaveFunctionComponent::DeviceStatesNN
 * (peps_amd/host/qlpeps_gpu.h) tabulated -- so MCUpdateSquareNNExchangeOBC on a fermionic state runs on the device too
 * (square_nn_updater.h:142-189 with the fermionic UpdateLocal of wave_function_component.h:345-378).  The amplitudes are those of
 * the decorated network of the current mode order (no sign factors: the Metropolis test sees moduli only; the caller restores
 * Sigma / Kappa from the configuration it gets back). */
int pepsgpu_sweep_slice_exchange_tab(pepsgpu_ctx *ctx, int orientation, int slice, int n_uniform, const double *uniforms,
                                     const int32_t *pair_table, double *amplitude_inout, int32_t *consumed_out, int32_t *accepted_out,
                                     int32_t *slice_states_out);
/* One row / column of MCUpdateSquareNNFullSpaceUpdateOBC (square_nn_updater.h:253-293) on the device: per bond the replacement trace of
 * all phys_dim^2 states of the pair (ReplaceNNSiteTrace with phys_dim^2 candidates), the weights |psi_k / psi|^2 (the current state
 * keeps its stored amplitude), SuwaTodoStateUpdate (suwa_todo_update.h:53-112) and UpdateLocal.
 *   engine_words [n][2 (slice length - 1)]: per walker the next raw 32-bit outputs of its std::mt19937 -- the reference draws
 *                std::uniform_real_distribution<long double>, i.e. two words per bond whatever the data, so the walker's engine is
 *                consumed exactly as by the reference's updater (identical chains; the kernel's arithmetic is float64 where
 *                SuwaTodoStateUpdate uses long double: a decision can differ only within 2^-53 of a boundary of the cumulative weights);
 *   phys_dim     states per site the move ranges over (phys_dim^2 <= 16); bosonic states (not the extended fermionic ones);
 *   amplitude_inout, accepted_out, slice_states_out as above (accepted = moves that changed the pair). */
int pepsgpu_sweep_slice_fullspace(pepsgpu_ctx *ctx, int orientation, int slice, int phys_dim, const uint32_t *engine_words,
                                  double *amplitude_inout, int32_t *accepted_out, int32_t *slice_states_out);

/* One row / column of the energy evaluation for models whose nearest-neighbour off-diagonal term exchanges the two site states (XXZ,
 * J1-J2, t-J ...): the slice part of SquareNNNModelEnergySolver::CalEnergyAndHolesImpl (square_nnn_energy_solver.h:142-200 row pass:
 * InitBTen, GrowFullBTen(RIGHT, row, 1, true), psi = Trace, per site PunchHole, per bond the ReplaceNNSiteTrace of the exchanged pair
 * + ShiftBTenWindow) and of the column pass (bond_traversal_mixin.h:120-144: GrowFullBTen(DOWN, col, 2, true), no holes) on the device
 * with ONE read-back: psi_out [n], psi_exchanged_out [n][slice length - 1] (amplitude of the configuration with the two sites of bond
 * j exchanged; for equal states it is psi).  punch_holes != 0: the holes of the slice's sites are stored in HBM as pepsgpu_punch_hole
 * with out == NULL does.  The BMPS pair of the slice must be in place.  Real element types only. */
int pepsgpu_nn_exchange_slice(pepsgpu_ctx *ctx, int orientation, int slice, int punch_holes, double *psi_out, double *psi_exchanged_out);

/* BMPSWalker as an object -- BMPSContractor::GetWalker / class BMPSWalker, bmps_contractor.h:357-646, bmps/impl/bmps_walker.h:13-465.
 * A walker holds the fork of the top BMPS of stack `pos` for every Monte-Carlo walker of the context (deep copy; the stacks are not
 * touched afterwards), its stack-size counter and its own LEFT / RIGHT BTen caches.  The TransferMPO the calls below absorb / sandwich
 * is named once with pepsgpu_walker_set_mpo: slice `num` of the network (a row for UP / DOWN walkers, a column for LEFT / RIGHT)
 *   states == NULL && tensors == NULL : under the walkers' current configurations,
 *   states  [n][N] (N = slice length)  : with the given SITPS component at every site of the slice, per walker (an excited row),
 *   tensors [n_tensors][N][D^4]        : explicit site tensors, float64 (interleaved pairs for PEPSGPU_C128), leg order (L, D, R, U)
 *                                        zero padded to D, leg dims of the slice's sites; n_tensors = 1 (shared) or n -- an MPO that is
 *                                        NOT a row of the context's network (Evolve(const TransferMPO &), bmps_walker.h:13-21).
 * The opposite boundary of the row operations is level `opp_level` of the DOWN stack (down_stack[opp_level] of the reference's tests:
 * the boundary that has absorbed opp_level rows from below); only the UP walker / DOWN opposite pair is supported, as in the reference
 * (bmps_walker.h:114-118: anything else -> PEPSGPU_EINVAL).  Errors of the reference's runtime_error checks -> PEPSGPU_ESTATE.
 * A walker dies with pepsgpu_walkers_set_configs (it is a fork of the old configurations' stacks). */
/* level < 0: GetWalker(tn, pos) :51-58 (fork of the top of the stack); level >= 0: BMPSWalker(tn, stack[level], pos, level + 1,
 * trunc_params), the constructor the structure-factor mixin uses on the vacuum (structure_factor_measurement_mixin.h:121-122) */
int pepsgpu_walker_create(pepsgpu_ctx *ctx, int pos, int level, int *walker_out);
/* copy construction (`auto excited_walker = main_walker;`, :134); the BTen caches of the copy start empty */
int pepsgpu_walker_clone(pepsgpu_ctx *ctx, int walker, int *walker_out);
int pepsgpu_walker_destroy(pepsgpu_ctx *ctx, int walker);
/* GetPosition / GetStackSize / GetBTenLeftCol / GetBTenRightCol (any pointer may be NULL) */
int pepsgpu_walker_info(pepsgpu_ctx *ctx, int walker, int *pos_out, int *stack_size_out, int *bten_left_col_out, int *bten_right_col_out);
int pepsgpu_walker_set_mpo(pepsgpu_ctx *ctx, int walker, int num, const int32_t *states, const double *tensors, int n_tensors);
int pepsgpu_walker_evolve(pepsgpu_ctx *ctx, int walker);                                     /* Evolve(mpo)               :13-21  */
int pepsgpu_walker_evolve_step(pepsgpu_ctx *ctx, int walker);                                /* EvolveStep()              :23-49  */
int pepsgpu_walker_contract_row(pepsgpu_ctx *ctx, int walker, int opp_level, double *out);   /* ContractRow(mpo, opp)     :60-214 */
int pepsgpu_walker_init_bten(pepsgpu_ctx *ctx, int walker, int opp_level, int position, int target_col);   /* InitBTenLeft / Right :216-272 */
int pepsgpu_walker_grow_bten_step(pepsgpu_ctx *ctx, int walker, int opp_level, int position);   /* GrowBTenLeftStep / RightStep :274-324 */
int pepsgpu_walker_shift_bten_window(pepsgpu_ctx *ctx, int walker, int opp_level, int position);   /* ShiftBTenWindow       :326-350 */
/* TraceWithBTen(site, site_col, opp) :352-392 (two_site = 0) / TraceWithTwoSiteBTen(site_a, site_b, site_col, mpo, opp) :394-463
 * (two_site = 1).  The replacement site(s): site_states [n] ([n][2]) = SITPS component per walker, or site_tensors
 * [n_tensors] ([n_tensors][2]) [D^4] explicit (layout as above), or both NULL = the MPO's own tensors.  out [n]. */
int pepsgpu_walker_trace_with_bten(pepsgpu_ctx *ctx, int walker, int opp_level, int site_col, int two_site, const int32_t *site_states,
                                   const double *site_tensors, int n_tensors, double *out);
int pepsgpu_walker_clear_bten(pepsgpu_ctx *ctx, int walker);                                  /* ClearBTen()                       */
/* GetBMPS()[idx] of the walker (as pepsgpu_get_bmps_tensor) */
int pepsgpu_walker_get_bmps_tensor(pepsgpu_ctx *ctx, int walker, int idx, int *dims_out, double *data_out, double *logscale_out);
int pepsgpu_generate_bmps_approach(pepsgpu_ctx *ctx, int pos);  /* GenerateBMPSApproach      :11-17   */
int pepsgpu_bmps_stack_size(pepsgpu_ctx *ctx, int pos);         /* GetBMPS(pos).size(); <0 on error */
/* GetBMPS(pos)[level][idx]: dims_out[3]; data_out [n][d0*d1*d2] float64 (may be NULL);
 * logscale_out [n] (may be NULL): the BMPS represents tensors * exp(logscale). Gauge differs
 * from the reference (only gauge-invariant contractions are comparable). */
int pepsgpu_get_bmps_tensor(pepsgpu_ctx *ctx, int pos, int level, int idx, int *dims_out, double *data_out,
                            double *logscale_out);

/* BTen (rank-3 environments of one row/column) -- bmps_contractor_init.h:72-128, grow.h:243-373,:517-582 */
int pepsgpu_init_bten(pepsgpu_ctx *ctx, int pos, int slice);                               /* InitBTen      */
int pepsgpu_grow_full_bten(pepsgpu_ctx *ctx, int pos, int slice, int remain_sites, int init); /* GrowFullBTen */
int pepsgpu_grow_bten_step(pepsgpu_ctx *ctx, int pos);                                     /* GrowBTenStep  */
int pepsgpu_shift_bten_window(pepsgpu_ctx *ctx, int pos);                                  /* ShiftBTenWindow */
int pepsgpu_truncate_bten(pepsgpu_ctx *ctx, int pos, int length);                          /* TruncateBTen  */
int pepsgpu_bten_stack_size(pepsgpu_ctx *ctx, int pos);

/* Scalar contractions -- bmps_contractor_trace.h.  out_amp = [n] (or [n][n_cand]) float64. */
int pepsgpu_trace(pepsgpu_ctx *ctx, int row, int col, int bond_dir, double *out_amp);       /* Trace :11-28 */
/* ReplaceNNSiteTrace(tn, site_a, site_b, dir, T_a[cand[..][0]], T_b[cand[..][1]])  :90-205.
 * cand_states = [n][n_cand][2] physical states put on (site_a, site_b). */
int pepsgpu_replace_nn_trace(pepsgpu_ctx *ctx, int row, int col, int bond_dir, int n_cand,
                             const int32_t *cand_states, double *out_amp);
/* ReplaceOneSiteTrace(tn, site, T[cand], mps_orient)  :30-88.  cand_states = [n][n_cand]. */
int pepsgpu_replace_one_trace(pepsgpu_ctx *ctx, int row, int col, int mps_orient, int n_cand,
                              const int32_t *cand_states, double *out_amp);
/* BTen2 (rank-4 environments of two adjacent rows/columns, bten_set2_) -- bmps_contractor_init.h:130-186,
 * bmps_contractor_grow.h:375-527.  slice_num1 = the first of the two rows (LEFT/RIGHT) or columns (UP/DOWN). */
int pepsgpu_init_bten2(pepsgpu_ctx *ctx, int pos, int slice_num1);                              /* InitBTen2  init.h:130-186 */
int pepsgpu_grow_full_bten2(pepsgpu_ctx *ctx, int pos, int slice_num1, int remain_sites, int init); /* GrowFullBTen2 grow.h:375-470 */
int pepsgpu_grow_bten2_step(pepsgpu_ctx *ctx, int pos, int slice_num1);                         /* GrowBTen2Step grow.h:472-515 */
int pepsgpu_shift_bten2_window(pepsgpu_ctx *ctx, int pos, int slice_num1);                      /* ShiftBTen2Window grow.h:523-527 */
int pepsgpu_bten2_stack_size(pepsgpu_ctx *ctx, int pos);
/* Diagonal direction of a next-nearest / sqrt(5) link (basic.h:89-92 DIAGONAL_DIR). */
#define PEPSGPU_LEFTUP_TO_RIGHTDOWN 0
#define PEPSGPU_LEFTDOWN_TO_RIGHTUP 1
/* ReplaceNNNSiteTrace(tn, left_up_site, nnn_dir, mps_orient, ten_left, ten_right)  trace.h:207-324.
 * (row, col) = upper-left corner of the plaquette; cand_states = [n][n_cand][2] states put on the
 * (left, right) end of the diagonal.  n_cand = 0: no replacement (the plaquette trace), out = [n]. */
int pepsgpu_replace_nnn_trace(pepsgpu_ctx *ctx, int row, int col, int nnn_dir, int mps_orient, int n_cand,
                              const int32_t *cand_states, double *out_amp);
/* Environment-reusing diagonal (NNN) hop of a FERMIONIC state (square_spinless_fermion.h:161-213 through
 * square_nnn_energy_solver.h:203-265).  The reference's graded ReplaceNNNSiteTrace carries the fermionic signs in the tensor
 * algebra; in the sign-decorated form of this library a diagonal hop also changes the decoration (variant) of every site between
 * its two ends in the row-major mode order -- row r right of the plaquette, row r + 1 left of it -- so the hopped amplitude is a
 * local replacement against "twisted" two-row environments.  Three calls provide them:
 *   pepsgpu_bten2_select_set(ctx, set)        set = 0 / 1: the BTen2 set init / grow / shift / replace_* work on from now on
 *   pepsgpu_cfg_override_slice(ctx, orient, num, states)   states = [n][N] extended states the kernels read for row (HORIZONTAL) /
 *                                              column (VERTICAL) `num` instead of the walkers' own until cleared (states = NULL)
 *   pepsgpu_replace_plaquette_trace(ctx, row, col, n_cand, cand, left_set, right_set, out)   the plaquette with upper-left corner
 *                                              (row, col) closed with four replaced tensors, cand = [n][n_cand][4] states of
 *                                              (row, col), (row+1, col), (row+1, col+1), (row, col+1), between the LEFT BTen2 of
 *                                              `left_set` and the RIGHT BTen2 of `right_set`; n_cand = 0: the walkers' own states.
 * Any configuration update or set_configs drops the second set and the override. */
int pepsgpu_bten2_select_set(pepsgpu_ctx *ctx, int set);
int pepsgpu_cfg_override_slice(pepsgpu_ctx *ctx, int orient, int num, const int32_t *states);
int pepsgpu_replace_plaquette_trace(pepsgpu_ctx *ctx, int row, int col, int n_cand, const int32_t *cand_states, int left_set,
                                    int right_set, double *out_amp);
/* ReplaceTNNSiteTrace(tn, site0, mps_orient, T0, T1, T2)  trace.h:326-423.  (row, col) = first of
 * three consecutive sites along mps_orient; cand_states = [n][n_cand][3]. */
int pepsgpu_replace_tnn_trace(pepsgpu_ctx *ctx, int row, int col, int mps_orient, int n_cand,
                              const int32_t *cand_states, double *out_amp);
/* ReplaceSqrt5DistTwoSiteTrace(tn, left_up_site, sqrt5link_dir, mps_orient, ten_left, ten_right)
 * trace.h:425-536.  (row, col) = upper-left corner of the 2x3 (HORIZONTAL) or 3x2 (VERTICAL) block;
 * cand_states = [n][n_cand][2] states on the (left, right) end of the link. */
int pepsgpu_replace_sqrt5_trace(pepsgpu_ctx *ctx, int row, int col, int link_dir, int mps_orient, int n_cand,
                                const int32_t *cand_states, double *out_amp);
/* PunchHole(tn, site, mps_orient)  grow.h:150-183.  out = [n][D][D][D][D] float64 (legs L,D,R,U, zero padded). */
int pepsgpu_punch_hole(pepsgpu_ctx *ctx, int row, int col, int mps_orient, double *out);
/* out == NULL: the hole of every walker stays on the device (resident hole store) for
 * pepsgpu_grad_accumulate.  The three calls below replace the per-sample accumulation loop
 * Ostar_sum += O*, ELocConj_Ostar_sum += E_loc^* O* (mc_energy_grad_evaluator.h:257-278; exact
 * summation: exact_summation_energy_evaluator.h:218-239) without moving the holes over PCIe:
 *   O*(site)[config_w(site)] = hole_w(site) / psi_w            (exact_sum = 0, weight 1)
 *   |psi|^2 O*               = psi_w * hole_w(site)            (exact_sum = 1)
 * psi, eloc = [n] host arrays (the solver's scalars).  pepsgpu_grad_read returns S_O and S_EO in the
 * state layout [row][col][s][L][D][R][U]; the cross-rank mean is one all-reduce of those buffers
 * (replaces MPIMeanTensor, statistics_tensor.h:37-79). */
int pepsgpu_grad_reset(pepsgpu_ctx *ctx);
int pepsgpu_grad_accumulate(pepsgpu_ctx *ctx, const double *psi, const double *eloc, int exact_sum);
/* The same with the component every hole belongs to named by the caller: states = [n][rows][cols], 0 <= state < d of
 * the context (NULL: the walkers' current configuration, as above).  Fermionic states (sign-decorated components,
 * INTEGRATION.md): the holes are those of the row-major decoration, states = the extended states of that decoration and
 * psi = the plain contraction value, whatever mode order the walkers have been moved to since the holes were punched
 * (replaces the same loop, mc_energy_grad_evaluator.h:257-278, for fermionic tensors). */
int pepsgpu_grad_accumulate_states(pepsgpu_ctx *ctx, const double *psi, const double *eloc, int exact_sum,
                                   const int32_t *states);
int pepsgpu_grad_read(pepsgpu_ctx *ctx, double *s_o_out, double *s_eo_out);

/* Stochastic reconfiguration: the O* samples stay in HBM and the S-matrix product of
 * SRSMatrix::operator* (optimizer/stochastic_reconfiguration_smatrix.h:37-99) is two sweeps over them.
 *   pepsgpu_sr_begin(max_samples)   allocate the store [sample][site][D^4]
 *   pepsgpu_sr_append(psi)          append O*_w = hole_w / psi_w of the current walkers (resident hole store of
 *                                   pepsgpu_punch_hole(.., NULL); replaces Ostar_samples.push_back, mc_energy_grad_evaluator.h:270)
 *   pepsgpu_sr_sum(out)             sum_i O*_i in the state layout (the caller divides by the total sample count -> Ostar_mean)
 *   pepsgpu_sr_matvec(v, mean_dot_v, scale, out)   out = scale * sum_i (O*_i . v - mean_dot_v) O*_i  (state layout, float64);
 *                                   the caller all-reduces over ranks and adds diag_shift * v. */
int pepsgpu_sr_begin(pepsgpu_ctx *ctx, int max_samples);
int pepsgpu_sr_append(pepsgpu_ctx *ctx, const double *psi);
int pepsgpu_sr_count(pepsgpu_ctx *ctx);
int pepsgpu_sr_sum(pepsgpu_ctx *ctx, double *sum_out);
int pepsgpu_sr_matvec(pepsgpu_ctx *ctx, const double *v, double mean_dot_v, double scale, double *out);
/* PEPSGPU_C128 contexts (SRSMatrix<QLTEN_Complex, QNT>): the sample store (begin / append / sum) takes psi and returns sums as
 * interleaved pairs; O*_i = conj(1 / psi_i) Dag(hole_i) (mc_energy_grad_evaluator.h:245-270); the product uses the positive-definite
 * pairing of SplitIndexTPS::operator* (split_index_tps.h:370-377):  out = scale * sum_i (<O*_i, v> - mean_dot_v) O*_i,
 * <a, b> = sum conj(a) b; v, out = state layout, interleaved pairs.  pepsgpu_sr_cg_solve works on PEPSGPU_C128 contexts too (b, x0, x_out
 * interleaved pairs; pap valid when Re > 0 and |Im| < 1e-10 as detail::pap_is_valid), and so do the MinSR blocks: pepsgpu_sr_gram returns
 * ip_ij = O*_i * O*_j = sum conj(O*_i) O*_j as interleaved pairs, pepsgpu_sr_weighted_sum takes complex weights (interleaved). */
int pepsgpu_sr_matvec_c128(pepsgpu_ctx *ctx, const double *v, double mean_dot_v_re, double mean_dot_v_im, double scale, double *out);
/* ConjugateGradientSolver (utility/conjugate_gradient_solver.h:181-276) on (S + diag_shift) x = b over the samples of THIS
 * context, every vector resident in HBM; parameters = ConjugateGradientParams (optimizer/optimizer_params.h:50-57).
 * b, x0 (NULL = 0) and x_out are host buffers in the state layout.  reason = CGTerminationReason:
 * 0 converged, 1 max iterations, 2 indefinite matrix, 3 numerical breakdown, 4 stagnated; on 1..4 x_out is the best iterate.
 * Multi-rank solves go through pepsgpu_sr_matvec + an all-reduce per product (peps_amd/sr.py). */
int pepsgpu_sr_cg_solve(pepsgpu_ctx *ctx, const double *b, const double *x0, double diag_shift, int max_iter,
                        double relative_tolerance, double absolute_tolerance, int residual_recompute_interval,
                        double orthogonality_threshold, double *x_out, double *residual_norm, int *iterations, int *reason);
/* MinSR building blocks (optimizer/minsr_tmatrix.h:53-147, optimizer_impl.h:1126-1215):
 *   pepsgpu_sr_gram          out[i][j] = <O*_i, O*_j'>, i over the local samples, j over a batch of n_remote samples given by
 *                            DEVICE pointers (layout of the local store: [sample][site][D^4] of the context's dtype and
 *                            int32 [sample][site]); remote == NULL: the local batch against itself.  One MFMA GEMM, f64 sums.
 *   pepsgpu_sr_weighted_sum  out = sum_i y[i] O*_i over the local samples (state layout) -- the back-substitution
 *   pepsgpu_sr_copy_samples  device-to-device copy of the local store into caller-owned device buffers (what the ring
 *                            exchange of MinSRTMatrix::Construct sends to the next rank) */
int pepsgpu_sr_gram(pepsgpu_ctx *ctx, const void *remote_samples_dev, const int32_t *remote_configs_dev, int n_remote, double *out);
int pepsgpu_sr_weighted_sum(pepsgpu_ctx *ctx, const double *y, double *out);
int pepsgpu_sr_copy_samples(pepsgpu_ctx *ctx, void *dst_samples_dev, int32_t *dst_configs_dev);

/* TPSWaveFunctionComponent::UpdateLocal (wave_function_component.h:345-378) for the walkers with
 * accept_mask[w] != 0 (NULL = all): config(site_k) = new_states[w][k], tn.UpdateSiteTensor,
 * contractor.EraseEnvsAfterUpdate(site_k) (trace.h:538-589).  sites = [n_sites][2] (row, col). */
int pepsgpu_update_local(pepsgpu_ctx *ctx, int n_sites, const int32_t *sites, const int32_t *new_states,
                         const uint8_t *accept_mask);
int pepsgpu_erase_envs_after_update(pepsgpu_ctx *ctx, int row, int col);                   /* trace.h:538-589 */

/* TPSWaveFunctionComponent::EvaluateAmplitude (wave_function_component.h:187-212):
 * GrowBMPSForRow(0); GrowFullBTen(RIGHT,0,2,true); InitBTen(LEFT,0); Trace({0,0},HORIZONTAL). */
int pepsgpu_evaluate_amplitude(pepsgpu_ctx *ctx, double *out_amp);

/* walker_flags[w] != 0: a boundary tensor of walker w vanished (reference throws, bmps_impl.h:839-843). */
int pepsgpu_walker_flags(pepsgpu_ctx *ctx, int32_t *flags_out);
int pepsgpu_sync(pepsgpu_ctx *ctx);
/* stats_out: [0] row absorptions, [1] Jacobi launches, [2] Jacobi sweeps (sum of per-launch maxima), [3] device bytes held,
 * [4] largest sweep count; with PEPSGPU_DEBUG_SWEEPS=1 at context creation also [5] sum of live carry rows, [6] sum of
 * carry sizes, [7] largest live carry of any walker (> 32: the dense Gram / Cholesky / full Jacobi route ran); [8] absorptions
 * done twice (a bond sized from the previous row's live count was filled, or a kernel size class skipped on the previous row's
 * carry rank was needed after all: performance hints only, results never depend on them) */
int pepsgpu_stats(pepsgpu_ctx *ctx, double *stats_out, int n);
/* Per-kernel timing with HIP events recorded on the launch stream (bench.py roofline leg).
 * out = [11][5]: {ms, launches, algorithmic flops, executed flops, operand + result bytes of the live extents} per category
 * 0 contraction GEMMs, 1 f64 Gram, 2 Cholesky, 3 Jacobi, 4 select, 5 normalise, 6 BTen/trace GEMMs, 7 Jacobi of edge blocks,
 * 8 Gram + Cholesky of the preconditioned truncation (mid-rank route), 9 its back-multiplication, 10 the chained contraction
 * kernel alone (tgemm_chain_kernel: one bracket per launch of that kernel; NOT included in category 0).
 * "algorithmic" = flops of the reference op the launch replaces (SURVEY.md 8d formulas). */
int pepsgpu_profile_enable(pepsgpu_ctx *ctx, int on);
int pepsgpu_profile_read(pepsgpu_ctx *ctx, double *out);

/* ---- the one exchange step of the path: sum over ranks of the energy / gradient accumulators ----
 * Replaces MPIMeanTensor per (site, component) (monte_carlo_tools/statistics_tensor.h:37-79), the
 * MPI_Send/Recv + reduce of S_O, S_EO (exact_summation_energy_evaluator.h:252-280, mc_energy_grad_evaluator.h:292-310),
 * the energy Gather (statistics.h:185-207) and the MPI_Allreduce(MAX) of acceptance rates
 * (mc_energy_grad_evaluator.h:405-410) by RCCL all-reduces over xGMI, one rank (= one context) per GPU.
 * The reference host owns the rendezvous (it has MPI): rank 0 calls pepsgpu_comm_unique_id and broadcasts the 128
 * bytes (MPI_Bcast; torch.distributed.broadcast_object_list in the Python host), every rank calls pepsgpu_comm_init.
 * A context without a communicator is a single rank: every reduction is the identity (as MPI with one rank). */
int pepsgpu_comm_unique_id(void *id128_out);                                     /* ncclGetUniqueId: 128 bytes */
int pepsgpu_comm_init(pepsgpu_ctx *ctx, int nranks, int rank, const void *id128); /* collective over the ranks */
int pepsgpu_comm_size(pepsgpu_ctx *ctx);
int pepsgpu_comm_rank(pepsgpu_ctx *ctx);
int pepsgpu_comm_destroy(pepsgpu_ctx *ctx);
/* in-place all-reduce of n elements on the context's stream, complete on return.  dtype: 0 f32, 1 f64, 2 int32;
 * op: 0 sum, 1 max; on_device != 0: buf is an HBM pointer of this context's GPU (no host hop), else a host buffer
 * staged through HBM. */
int pepsgpu_allreduce(pepsgpu_ctx *ctx, void *buf, long n, int dtype, int op, int on_device);
/* S_O and S_EO of pepsgpu_grad_accumulate summed over the ranks where they live (HBM), before pepsgpu_grad_read. */
int pepsgpu_grad_allreduce(pepsgpu_ctx *ctx);
/* Broadcast of the parameter buffer after an optimizer update: the flat SITPS of rank `root` (pepsgpu_state_upload there)
 * goes to the HBM state buffer of every rank of the context's communicator with one ncclBroadcast over xGMI -- no host upload
 * on the other ranks.  Replaces the per-tensor MPI_Bcast of SplitIndexTPS (two_dim_tn/tps/split_index_tps_impl.h:778-880,
 * called at algorithm/vmc_update/mc_energy_grad_evaluator.h:161).  Collective over the ranks; one rank: marks the state valid. */
int pepsgpu_bcast_state(pepsgpu_ctx *ctx, int root);
/* device pointers of the two float64 accumulators ([row][col][s][D^4 slot], n_elems each) for hosts that run their own
 * collective on them (torch.distributed backend "nccl" = RCCL: peps_amd/dist.py wraps them without a copy). */
int pepsgpu_grad_device_ptr(pepsgpu_ctx *ctx, void **so_dev, void **seo_dev, long *n_elems);

/* ---- diagnostics (unit tests of the kernels; not part of the reference surface) ---- */
int pepsgpu_diag_tgemm(int dtype_in, int dtype_out, const int *desc_ints, int n_ints, const void *A, size_t a_elems,
                       const void *B, size_t b_elems, void *C, size_t c_elems, int nbatch, long wA, long wB, long wC);
int pepsgpu_diag_chol(int dtype_out, const double *G, int n, int nbatch, void *R_out);
/* the streaming f64 Gram kernel of the forward pass (gram.h): P = [nbatch][K][n], klive (nullable) = live rows per entry */
int pepsgpu_diag_gram_cols(int dtype, const void *P, int K, int n, int nbatch, const int32_t *klive, double *G_out);
int pepsgpu_diag_gram_rows(const float *M, int n, int K, int nbatch, const int32_t *nrows, double *G_out);
int pepsgpu_diag_mgemm_dense(const float *R, const float *Tt, int m, int la, int a_dim, int u_dim, int k2_dim, int tt_u_inner, int nbatch,
                             const int32_t *m_live, const int32_t *a_live, const int32_t *k2_live, float *M_out);
/* diagnostics (PEPSGPU_CG_STATS=1): per-phase counters of colgram_dense_kernel, read and reset; see trunc_mid.h */
int pepsgpu_diag_cg_stats(double *out16);
/* the LDS-resident Gram + Cholesky kernels alone (f32): which = 0 rows form (X = [nbatch][n][K], R^T R = X X^T, nlive = live rows),
 * which = 1 column form (X = [nbatch][K][n], R^T R = X^T X, nlive = live rows of X); R_out = [nbatch][n][n], n <= 128 */
int pepsgpu_diag_lds_gram_chol(int which, const float *X, int n, int K, int nbatch, const int32_t *nlive, float *R_out, int32_t *mlive_out);
/* the first compression of the dense truncation route as it runs (round 6): i8 row Gram with both triangles + the diagonally pivoted
 * factorisation stopped after kcap <= 64 rows (chol_pivot.h); X = [nbatch][n][K] f32, 128 < n <= 256; R_out = [nbatch][kcap][n], rows in pivot order */
int pepsgpu_diag_chol_pivot(const float *X, int n, int K, int nbatch, const int32_t *nlive, int kcap, float *R_out, int32_t *mlive_out);
/* rows_qr_kernel alone (round 6): the k <= 32 nearly orthogonal rows X = [nbatch][k][len] (len <= 256, by decreasing norm) made orthonormal in
 * float64 (Cholesky-QR of the unit-scaled rows); V_out: live rows first, the rest zero; klive_out = their count */
int pepsgpu_diag_rows_qr(const float *X, int k, int len, int nbatch, const int32_t *klive, float *V_out, int32_t *klive_out);
/* the device's SuwaTodoStateUpdate alone (round 6, the decision of pepsgpu_sweep_slice_fullspace): a chain of `steps` updates on one
 * weight vector (n <= 16), words = 2 * steps raw outputs of the caller's std::mt19937; out_chain[s] = the state after step s */
int pepsgpu_diag_suwa_todo(const double *weights, int n, int init, const uint32_t *words, int steps, int32_t *out_chain);
/* the rank-adaptive pair used by the absorption: low-rank right-looking kernel, then the blocked
 * kernel for the walkers whose rank exceeds its cap; mlive_out[b] = rows of R_out[b] that exist */
/* the Gram-free low-rank kernel alone: P = [nbatch][K][n] (dtype), R^T R = P^T P; mlive_out[b] = -1 where
 * it declines (K or the rank above its caps) and the Gram + Cholesky pair has to run */
/* forward contraction pair of an absorption through the LDS-chained kernel (tgemm_chain_kernel); see capi.hip */
int pepsgpu_diag_tgemm_chain(const int *dims7, const int32_t *live3_per_entry, int nbatch, const float *R, const float *A,
                             const float *W, float *P_out, int32_t *flags_out);
int pepsgpu_diag_gram_chol(int dtype, const void *P, int K, int n, int nbatch, void *R_out, int32_t *mlive_out);
int pepsgpu_diag_chol_adaptive(int dtype_out, const double *G, int n, int nbatch, void *R_out, int32_t *mlive_out);
int pepsgpu_diag_jacobi(int dtype, void *M, int m, int len, int nbatch, int k, void *Vt_out, void *S_out,
                        int force_global, int *sweeps_out);
const char *pepsgpu_version(void);

#ifdef __cplusplus
}
#endif
#endif /* PEPSGPU_H */
