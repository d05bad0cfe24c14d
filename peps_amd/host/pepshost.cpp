// C shim over the C++ host layer (qlpeps_gpu.h) so that the Python tests / bench drive exactly the
// host code a reference user would compile (updaters, solvers, evaluators), not a re-implementation.
#include <cstring>
#include <memory>

#include "qlpeps_gpu.h"

using namespace qlpeps_gpu;

namespace {
thread_local std::string g_err;
// GPU of the contractors built below: one rank per GPU -> LOCAL_RANK unless pepshost_set_device says otherwise
int default_device() {
  const char *e = getenv("LOCAL_RANK");
  return e ? atoi(e) : 0;
}
int g_device = default_device();

template <typename F>
int guarded(F &&f) {
  try {
    f();
    return 0;
  } catch (const std::invalid_argument &e) { g_err = e.what(); return PEPSGPU_EINVAL;
  } catch (const std::out_of_range &e) { g_err = e.what(); return PEPSGPU_ERANGE;
  } catch (const std::logic_error &e) { g_err = e.what(); return PEPSGPU_ESTATE;
  } catch (const std::exception &e) { g_err = e.what(); return PEPSGPU_EEMPTY; }
}

SplitIndexTPS make_state(int rows, int cols, int D, int d, const double *flat) {
  SplitIndexTPS s(rows, cols, d, D);
  std::copy(flat, flat + s.flat().size(), s.flat().begin());
  return s;
}
// element-type generic form: `flat` = doubles, interleaved (re, im) pairs for QLTEN_Complex
template <typename TenElemT>
SplitIndexTPST<TenElemT> make_state_t(int rows, int cols, int D, int d, const double *flat) {
  SplitIndexTPST<TenElemT> s(rows, cols, d, D);
  const TenElemT *src = reinterpret_cast<const TenElemT *>(flat);
  std::copy(src, src + s.flat().size(), s.flat().begin());
  return s;
}
template <typename TenElemT>
void copy_out(const std::vector<TenElemT> &v, double *out) {
  if (out) std::copy(dptr(v.data()), dptr(v.data()) + v.size() * (ElemTraits<TenElemT>::is_complex ? 2 : 1), out);
}
// BMPSTruncateParams of the contractors built below: SVD(chi, chi, 0) unless pepshost_set_truncate_params changed
// D_min / trunc_err / scheme (D_max is always the call's chi)
struct TruncOverride { int d_min = -1; double trunc_err = 0.0; int scheme = 0; double tol = 0.0; int iters = 0; } g_trunc;
BMPSTruncateParams trunc_params(int chi) {
  const size_t dmin = g_trunc.d_min < 0 ? (size_t)chi : (size_t)std::min(g_trunc.d_min, chi);
  if (g_trunc.scheme == 1) return BMPSTruncateParams::Variational2Site(dmin, chi, g_trunc.trunc_err, g_trunc.tol, g_trunc.iters);
  if (g_trunc.scheme == 2) return BMPSTruncateParams::Variational1Site(dmin, chi, g_trunc.trunc_err, g_trunc.tol, g_trunc.iters);
  return BMPSTruncateParams::SVD(dmin, chi, g_trunc.trunc_err);
}
Configuration make_cfg(int n, int rows, int cols, const int32_t *cfg) {
  Configuration c(n, rows, cols);
  std::copy(cfg, cfg + (size_t)n * rows * cols, c.data());
  return c;
}

// ---- element-type generic bodies of the entry points below (TenElemT = QLTEN_Double / QLTEN_Complex) ----
template <typename TenElemT>
void energy_and_holes_impl(int rows, int cols, int D, int d, int chi, int dtype, const double *sitps_flat, int n,
                           const int32_t *configs, int model, const double *p, double *amplitudes_out,
                           double *energies_out, double *holes_out, double *psi_out, int *n_psi_out) {
  SplitIndexTPST<TenElemT> sitps = make_state_t<TenElemT>(rows, cols, D, d, sitps_flat);
  BMPSContractorT<TenElemT> contractor(rows, cols, D, d, trunc_params(chi), n, dtype, g_device);
  TPSWaveFunctionComponentT<TenElemT> comp(sitps, make_cfg(n, rows, cols, configs), contractor);
  EnergyAndHolesT<TenElemT> eh;
  if (model == 0) {
    SquareSpinOneHalfXXZModelOBC m(p[0], p[1], p[2]);
    eh = holes_out ? m.CalEnergyAndHoles<true>(sitps, comp) : m.CalEnergyAndHoles<false>(sitps, comp);
  } else if (model == 2) {
    SquareSpinOneHalfJ1J2XXZModelOBC m(p[0], p[1], p[2], p[3], p[4]);
    eh = holes_out ? m.CalEnergyAndHoles<true>(sitps, comp) : m.CalEnergyAndHoles<false>(sitps, comp);
  } else if (model == 3) {
    SpinOneHalfTriHeisenbergSqrPEPS m;
    eh = holes_out ? m.CalEnergyAndHoles<true>(sitps, comp) : m.CalEnergyAndHoles<false>(sitps, comp);
  } else if (model == 4) {
    SpinOneHalfTriJ1J2HeisenbergSqrPEPS m(p[0]);
    eh = holes_out ? m.CalEnergyAndHoles<true>(sitps, comp) : m.CalEnergyAndHoles<false>(sitps, comp);
  } else {
    TransverseFieldIsingSquareOBC m(p[0]);
    eh = holes_out ? m.CalEnergyAndHoles<true>(sitps, comp) : m.CalEnergyAndHoles<false>(sitps, comp);
  }
  constexpr int z = ElemTraits<TenElemT>::is_complex ? 2 : 1;
  copy_out(comp.amplitude, amplitudes_out);
  copy_out(eh.energy, energies_out);
  copy_out(eh.holes, holes_out);
  if (n_psi_out) *n_psi_out = (int)eh.psi_list.size();
  if (psi_out)
    for (size_t k = 0; k < eh.psi_list.size(); ++k) copy_out(eh.psi_list[k], psi_out + k * n * z);
}

template <typename TenElemT>
void exact_sum_partial_impl(int rows, int cols, int D, int d, int chi, int dtype, const double *sitps_flat,
                            const int32_t *all_configs, int n_configs, int model, const double *p, int rank, int size,
                            int batch, double *packed_out) {
  SplitIndexTPST<TenElemT> sitps = make_state_t<TenElemT>(rows, cols, D, d, sitps_flat);
  BMPSContractorT<TenElemT> contractor(rows, cols, D, d, trunc_params(chi), batch, dtype, g_device);
  std::vector<std::vector<int32_t>> all(n_configs);
  for (int i = 0; i < n_configs; ++i) all[i].assign(all_configs + (size_t)i * rows * cols, all_configs + (size_t)(i + 1) * rows * cols);
  std::vector<double> packed;
  auto capture = [&](std::vector<double> &v) { packed = v; };
  if (model == 0) {
    SquareSpinOneHalfXXZModelOBC m(p[0], p[1], p[2]);
    ExactSumEnergyEvaluator(sitps, all, contractor, m, rank, size, (size_t)batch, capture);
  } else if (model == 2) {
    SquareSpinOneHalfJ1J2XXZModelOBC m(p[0], p[1], p[2], p[3], p[4]);
    ExactSumEnergyEvaluator(sitps, all, contractor, m, rank, size, (size_t)batch, capture);
  } else if (model == 3) {
    SpinOneHalfTriHeisenbergSqrPEPS m;
    ExactSumEnergyEvaluator(sitps, all, contractor, m, rank, size, (size_t)batch, capture);
  } else if (model == 4) {
    SpinOneHalfTriJ1J2HeisenbergSqrPEPS m(p[0]);
    ExactSumEnergyEvaluator(sitps, all, contractor, m, rank, size, (size_t)batch, capture);
  } else {
    TransverseFieldIsingSquareOBC m(p[0]);
    ExactSumEnergyEvaluator(sitps, all, contractor, m, rank, size, (size_t)batch, capture);
  }
  std::copy(packed.begin(), packed.end(), packed_out);
}

template <typename TenElemT>
void mc_energy_grad_partial_impl(int rows, int cols, int D, int d, int chi, int dtype, const double *sitps_flat, int n,
                                 int32_t *configs, const uint64_t *seeds, int updater, int model, const double *p,
                                 int warmup_sweeps, int n_samples, double *packed_out, double *accept_out) {
  SplitIndexTPST<TenElemT> sitps = make_state_t<TenElemT>(rows, cols, D, d, sitps_flat);
  BMPSContractorT<TenElemT> contractor(rows, cols, D, d, trunc_params(chi), n, dtype, g_device);
  TPSWaveFunctionComponentT<TenElemT> comp(sitps, make_cfg(n, rows, cols, configs), contractor);
  std::vector<uint64_t> sd(seeds, seeds + n);
  MCUpdateSquareNNExchangeOBC ex(sd);
  MCUpdateSquareNNFullSpaceUpdateOBC fs(sd);
  std::vector<double> rates, acc_rate(n, 0.0);
  auto sweep = [&]() { if (updater == 0) ex(sitps, comp, rates); else fs(sitps, comp, rates); };
  for (int s = 0; s < warmup_sweeps; ++s) sweep();
  SquareSpinOneHalfXXZModelOBC xxz(p[0], p[1], p[2]);
  SquareSpinOneHalfJ1J2XXZModelOBC j1j2(p[0], p[1], p[2], p[3], p[4]);
  TransverseFieldIsingSquareOBC tfim(p[0]);
  GradAccumulatorT<TenElemT> acc(sitps);
  contractor.GradReset();
  for (int k = 0; k < n_samples; ++k) {
    sweep();
    for (int w = 0; w < n; ++w) acc_rate[w] += rates[w];
    EnergyAndHolesT<TenElemT> eh = model == 0   ? xxz.CalEnergyAndHoles<true>(sitps, comp, true)
                                   : model == 2 ? j1j2.CalEnergyAndHoles<true>(sitps, comp, true)
                                                : tfim.CalEnergyAndHoles<true>(sitps, comp, true);
    acc.AccumulateDevice(comp, eh, false);
  }
  if (n_samples > 0) acc.FetchDevice(contractor);
  std::vector<double> packed = acc.Pack();
  std::copy(packed.begin(), packed.end(), packed_out);
  std::copy(comp.config.data(), comp.config.data() + (size_t)n * rows * cols, configs);
  if (accept_out) for (int w = 0; w < n; ++w) accept_out[w] = n_samples ? acc_rate[w] / n_samples : 0.0;
}

template <typename TenElemT>
void mc_sweeps_impl(int rows, int cols, int D, int d, int chi, int dtype, const double *sitps_flat, int n,
                    int32_t *configs, const uint64_t *seeds, int updater, int n_sweeps, double *amplitudes_out,
                    double *accept_rates_out) {
  SplitIndexTPST<TenElemT> sitps = make_state_t<TenElemT>(rows, cols, D, d, sitps_flat);
  BMPSContractorT<TenElemT> contractor(rows, cols, D, d, trunc_params(chi), n, dtype, g_device);
  TPSWaveFunctionComponentT<TenElemT> comp(sitps, make_cfg(n, rows, cols, configs), contractor);
  std::vector<uint64_t> sd(seeds, seeds + n);
  std::vector<double> rates, acc(n, 0.0);
  if (updater == 0) {
    MCUpdateSquareNNExchangeOBC upd(sd);
    for (int s = 0; s < n_sweeps; ++s) { upd(sitps, comp, rates); for (int w = 0; w < n; ++w) acc[w] += rates[w]; }
  } else {
    MCUpdateSquareNNFullSpaceUpdateOBC upd(sd);
    for (int s = 0; s < n_sweeps; ++s) { upd(sitps, comp, rates); for (int w = 0; w < n; ++w) acc[w] += rates[w]; }
  }
  std::copy(comp.config.data(), comp.config.data() + (size_t)n * rows * cols, configs);
  copy_out(comp.amplitude, amplitudes_out);
  for (int w = 0; w < n; ++w) accept_rates_out[w] = n_sweeps ? acc[w] / n_sweeps : 0.0;
}
template <typename TenElemT>
void measure_impl(int rows, int cols, int D, int d, int chi, int dtype, const double *sitps_flat, int n, int32_t *configs,
                  const uint64_t *seeds, int updater, int model, const double *p, int warmup_sweeps, int n_samples,
                  int sweeps_between_samples, const char *dump_dir, char *keys_out, int keys_cap, double *values_out,
                  long values_cap, long *values_len) {
  SplitIndexTPST<TenElemT> sitps = make_state_t<TenElemT>(rows, cols, D, d, sitps_flat);
  BMPSContractorT<TenElemT> contractor(rows, cols, D, d, trunc_params(chi), n, dtype, g_device);
  TPSWaveFunctionComponentT<TenElemT> comp(sitps, make_cfg(n, rows, cols, configs), contractor);
  SquareSpinOneHalfXXZModelOBC xxz(p[0], p[1], p[2]);
  SquareSpinOneHalfJ1J2XXZModelOBC j1j2(p[0], p[1], p[2], p[3], p[4]);
  SpinOneHalfTriJ1J2HeisenbergSqrPEPS trij(p[0]);
  TransverseFieldIsingSquareOBC tfim(p[0]);
  if (model < 0 || model > 4 || model == 3) throw std::invalid_argument("pepshost_measure: model must be xxz, tfim, j1j2 or trij1j2");
  xxz.SetEnableStructureFactor(p[7] != 0.0);                  // params[7]: structure factor switch (xxz only)
  xxz.SetStructureFactorReferenceStackState(p[6] != 0.0);     // params[6]: the DOWN stack as the reference's traversal leaves it (K8; off)
  std::string keys;
  std::vector<double> vals;                                   // QLTEN_Complex: every number as a (re, im) pair
  auto push = [&](const auto &v) {
    vals.push_back(std::real(v));
    if constexpr (ElemTraits<TenElemT>::is_complex) vals.push_back(std::imag(v));
  };
  auto emit = [&](const std::string &key, const std::vector<TenElemT> &a, const std::vector<double> *b, size_t len) {
    keys += key + ":" + std::to_string(len) + ";";
    for (const auto &v : a) push(v);
    if (b) {
      if (b->empty()) for (size_t k = 0; k < a.size(); ++k) push(0.0);
      else for (double v : *b) push(v);
    }
  };
  PsiSummaryT<TenElemT> psi;
  if (n_samples <= 0) {
    ObservableMapT<TenElemT> obs = model == 0   ? xxz.EvaluateObservables(sitps, comp)
                                   : model == 1 ? tfim.EvaluateObservables(sitps, comp)
                                   : model == 2 ? j1j2.EvaluateObservables(sitps, comp)
                                                : trij.EvaluateObservables(sitps, comp);
    for (const auto &kv : obs.values) emit(kv.first, kv.second, nullptr, obs.len(kv.first));
    psi = model == 0   ? xxz.SquareNNModelMeasurementSolver<SquareSpinOneHalfXXZModelOBC>::template EvaluatePsiSummaryT<TenElemT>()
          : model == 1 ? tfim.template EvaluatePsiSummaryT<TenElemT>()
          : model == 2 ? j1j2.SquareNNNModelMeasurementSolver<SquareSpinOneHalfJ1J2XXZModelOBC>::template EvaluatePsiSummaryT<TenElemT>()
                       : trij.template EvaluatePsiSummaryT<TenElemT>();
  } else {
    std::vector<uint64_t> sd(seeds, seeds + n);
    MCMeasurementParams mp;
    mp.num_samples = n_samples; mp.num_warmup_sweeps = warmup_sweeps; mp.sweeps_between_samples = sweeps_between_samples;
    auto run = [&](auto &upd, auto &solver) {
      MCPEPSMeasurer<std::decay_t<decltype(upd)>, std::decay_t<decltype(solver)>, TenElemT> m(sitps, comp, mp, upd, solver);
      m.Execute();
      if (dump_dir && dump_dir[0]) m.DumpData(dump_dir);
      for (const auto &kv : m.ObservableRegistry()) emit(kv.first, kv.second.first, &kv.second.second, kv.second.first.size());
      psi.psi_mean.assign(n, TenElemT(0.0)); psi.psi_rel_err.assign(n, 0.0);
      for (int w = 0; w < n; ++w) { psi.psi_mean[w] = m.PsiSamples().back()[w].first; psi.psi_rel_err[w] = m.PsiSamples().back()[w].second; }
    };
    MCUpdateSquareNNExchangeOBC ex(sd);
    MCUpdateSquareNNFullSpaceUpdateOBC fs(sd);
    if (updater == 0 && model == 0) run(ex, xxz);
    else if (updater == 0 && model == 1) run(ex, tfim);
    else if (updater == 0 && model == 2) run(ex, j1j2);
    else if (updater == 0) run(ex, trij);
    else if (model == 0) run(fs, xxz);
    else if (model == 1) run(fs, tfim);
    else if (model == 2) run(fs, j1j2);
    else run(fs, trij);
    std::copy(comp.config.data(), comp.config.data() + (size_t)n * rows * cols, configs);
  }
  for (const auto &v : psi.psi_mean) push(v);
  for (double v : psi.psi_rel_err) push(v);
  if ((int)keys.size() + 1 > keys_cap || (long)vals.size() > values_cap) throw std::out_of_range("pepshost_measure: output buffer too small");
  std::copy(keys.begin(), keys.end(), keys_out);
  keys_out[keys.size()] = 0;
  std::copy(vals.begin(), vals.end(), values_out);
  *values_len = (long)vals.size();
}
template <typename TenElemT>
void exact_sum_measure_partial_impl(int rows, int cols, int D, int d, int chi, int dtype, const double *sitps_flat,
                                    const int32_t *all_configs, int n_configs, int model, const double *p, int rank, int size,
                                    int batch, char *keys_out, int keys_cap, double *values_out, long values_cap, long *values_len) {
  SplitIndexTPST<TenElemT> sitps = make_state_t<TenElemT>(rows, cols, D, d, sitps_flat);
  BMPSContractorT<TenElemT> contractor(rows, cols, D, d, trunc_params(chi), batch, dtype, g_device);
  std::vector<std::vector<int32_t>> all;
  if (n_configs >= 0) {
    all.resize(n_configs);
    for (int i = 0; i < n_configs; ++i) all[i].assign(all_configs + (size_t)i * rows * cols, all_configs + (size_t)(i + 1) * rows * cols);
  } else {
    all = GenerateAllBinaryConfigs(cols, rows);
  }
  if (model != 0 && model != 1 && model != 2) throw std::invalid_argument("pepshost_exact_sum_measure_partial: model must be xxz, tfim or j1j2");
  if (!all.empty() && (size_t)rank >= all.size()) {           // a rank without configurations contributes nothing
    if (keys_cap < 1 || values_cap < 1) throw std::out_of_range("pepshost_exact_sum_measure_partial: output buffer too small");
    keys_out[0] = 0; values_out[0] = 0.0; *values_len = 1;
    return;
  }
  std::vector<double> packed;
  auto capture = [&](std::vector<double> &v) { packed = v; v[0] = 1.0; };   // keep the raw sums; normalise in the caller
  std::map<std::string, std::vector<TenElemT>> res;
  if (model == 0) {
    SquareSpinOneHalfXXZModelOBC m(p[0], p[1], p[2]);
    m.SetEnableStructureFactor(p[7] != 0.0);
    res = ExactSumMeasurer(sitps, all, contractor, m, rank, size, (size_t)batch, capture);
  } else if (model == 1) {
    TransverseFieldIsingSquareOBC m(p[0]);
    res = ExactSumMeasurer(sitps, all, contractor, m, rank, size, (size_t)batch, capture);
  } else {
    SquareSpinOneHalfJ1J2XXZModelOBC m(p[0], p[1], p[2], p[3], p[4]);
    res = ExactSumMeasurer(sitps, all, contractor, m, rank, size, (size_t)batch, capture);
  }
  std::string keys;
  for (const auto &kv : res) keys += kv.first + ":" + std::to_string(kv.second.size()) + ";";
  if ((int)keys.size() + 1 > keys_cap || (long)packed.size() > values_cap) throw std::out_of_range("pepshost_exact_sum_measure_partial: output buffer too small");
  std::copy(keys.begin(), keys.end(), keys_out);
  keys_out[keys.size()] = 0;
  std::copy(packed.begin(), packed.end(), values_out);
  *values_len = (long)packed.size();
}

// ---- fermionic entry points, element-type generic (sitps_ext_flat: 4 d sign-decorated components per site) ----
template <typename TenElemT>
void fermion_energy_impl(int rows, int cols, int D, int d, const int32_t *nf, int chi, int dtype,
                         const double *sitps_ext_flat, int n, const int32_t *configs, int model, const double *prm,
                         double *amplitudes_out, double *energies_out, double *psi_out, int *n_psi_out) {
  SplitIndexTPST<TenElemT> sitps = make_state_t<TenElemT>(rows, cols, D, 4 * d, sitps_ext_flat);
  FermionDecoration dec;
  dec.nf.assign(nf, nf + d);
  BMPSContractorT<TenElemT> contractor(rows, cols, D, 4 * d, trunc_params(chi), n, dtype, g_device);
  TPSWaveFunctionComponentT<TenElemT> comp(sitps, make_cfg(n, rows, cols, configs), contractor, &dec);
  copy_out(comp.amplitude, amplitudes_out);
  EnergyAndHolesT<TenElemT> eh;
  if (model == 0) {
    SquareSpinlessFermion m(prm[0], prm[2], prm[1]);      // (t, t2, V): params = [t, V, t2, -]
    eh = m.CalEnergyAndHoles<false>(sitps, comp);
  } else {
    SquaretJVModel m(prm[0], 0.0, prm[1], prm[2], prm[3]);
    eh = m.CalEnergyAndHoles<false>(sitps, comp);
  }
  constexpr int z = ElemTraits<TenElemT>::is_complex ? 2 : 1;
  copy_out(eh.energy, energies_out);
  if (n_psi_out) *n_psi_out = (int)eh.psi_list.size();
  if (psi_out)
    for (size_t k = 0; k < eh.psi_list.size(); ++k) copy_out(eh.psi_list[k], psi_out + k * n * z);
}

template <typename TenElemT>
void fermion_exact_sum_partial_impl(int rows, int cols, int D, int d, const int32_t *nf, int chi, int dtype,
                                    const double *sitps_ext_flat, const int32_t *all_configs, int n_configs, double t,
                                    double V, int rank, int size, int batch, double *packed_out) {
  SplitIndexTPST<TenElemT> sitps = make_state_t<TenElemT>(rows, cols, D, 4 * d, sitps_ext_flat);
  FermionDecoration dec;
  dec.nf.assign(nf, nf + d);
  BMPSContractorT<TenElemT> contractor(rows, cols, D, 4 * d, trunc_params(chi), batch, dtype, g_device);
  std::vector<std::vector<int32_t>> all(n_configs);
  for (int i = 0; i < n_configs; ++i) all[i].assign(all_configs + (size_t)i * rows * cols, all_configs + (size_t)(i + 1) * rows * cols);
  std::vector<double> packed;
  auto capture = [&](std::vector<double> &v) { packed = v; };
  SquareSpinlessFermion m(t, V);
  ExactSumEnergyEvaluator(sitps, all, contractor, m, rank, size, (size_t)batch, capture, &dec);
  std::copy(packed.begin(), packed.end(), packed_out);
}

template <typename TenElemT>
void fermion_mc_sweeps_impl(int rows, int cols, int D, int d, const int32_t *nf, int chi, int dtype,
                            const double *sitps_ext_flat, int n, int32_t *configs, const uint64_t *seeds, int n_sweeps,
                            double *amplitudes_out, double *accept_out) {
  SplitIndexTPST<TenElemT> sitps = make_state_t<TenElemT>(rows, cols, D, 4 * d, sitps_ext_flat);
  FermionDecoration dec;
  dec.nf.assign(nf, nf + d);
  BMPSContractorT<TenElemT> contractor(rows, cols, D, 4 * d, trunc_params(chi), n, dtype, g_device);
  TPSWaveFunctionComponentT<TenElemT> comp(sitps, make_cfg(n, rows, cols, configs), contractor, &dec);
  std::vector<uint64_t> sd(seeds, seeds + n);
  MCUpdateSquareNNExchangeOBC ex(sd);
  std::vector<double> rates, acc(n, 0.0);
  for (int s = 0; s < n_sweeps; ++s) {
    ex(sitps, comp, rates);
    for (int w = 0; w < n; ++w) acc[w] += rates[w];
  }
  std::copy(comp.config.data(), comp.config.data() + (size_t)n * rows * cols, configs);
  copy_out(comp.amplitude, amplitudes_out);
  if (accept_out) for (int w = 0; w < n; ++w) accept_out[w] = n_sweeps ? acc[w] / n_sweeps : 0.0;
}

template <typename TenElemT>
void fermion_measure_energy_impl(int rows, int cols, int D, int d, const int32_t *nf, int chi, int dtype,
                                 const double *sitps_ext_flat, int n, int32_t *configs, const uint64_t *seeds, int warmup_sweeps,
                                 int n_samples, int sweeps_between, int model, const double *prm, double *energies_out,
                                 double *accept_out) {
  SplitIndexTPST<TenElemT> sitps = make_state_t<TenElemT>(rows, cols, D, 4 * d, sitps_ext_flat);
  FermionDecoration dec;
  dec.nf.assign(nf, nf + d);
  BMPSContractorT<TenElemT> contractor(rows, cols, D, 4 * d, trunc_params(chi), n, dtype, g_device);
  TPSWaveFunctionComponentT<TenElemT> comp(sitps, make_cfg(n, rows, cols, configs), contractor, &dec);
  std::vector<uint64_t> sd(seeds, seeds + n);
  MCUpdateSquareNNExchangeOBC ex(sd);
  std::vector<double> rates, acc(n, 0.0);
  for (int s = 0; s < warmup_sweeps; ++s) ex(sitps, comp, rates);
  comp.SetOrder(ROW_MAJOR);
  comp.EvaluateAmplitude();                                  // NormalizeStateOrder1: tps_sample_ = WaveFunctionComponentT(...) (:235-236)
  SquareSpinlessFermion spinless(prm[0], prm[2], prm[1]);     // model 0: (t, t2, V) from [t, V, t2, -]
  SquaretJVModel tj(prm[0], 0.0, prm[1], prm[2], prm[3]);     // model 1: [t, J, V, mu]
  constexpr int z = ElemTraits<TenElemT>::is_complex ? 2 : 1;
  for (int k = 0; k < n_samples; ++k) {
    for (int s = 0; s < sweeps_between; ++s) {
      ex(sitps, comp, rates);
      for (int w = 0; w < n; ++w) acc[w] += rates[w];
    }
    EnergyAndHolesT<TenElemT> eh = model == 0 ? spinless.CalEnergyAndHoles<false>(sitps, comp) : tj.CalEnergyAndHoles<false>(sitps, comp);
    copy_out(eh.energy, energies_out + (size_t)k * n * z);
  }
  std::copy(comp.config.data(), comp.config.data() + (size_t)n * rows * cols, configs);
  const int total = std::max(1, n_samples * sweeps_between);
  if (accept_out) for (int w = 0; w < n; ++w) accept_out[w] = acc[w] / total;
}
}  // namespace

extern "C" {

const char *pepshost_last_error(void) { return g_err.c_str(); }

// HIP device of every contractor the calls below build (default: LOCAL_RANK of the launcher, else 0)
int pepshost_set_device(int device) {
  return guarded([&]() {
    if (device < 0) throw std::invalid_argument("pepshost_set_device: negative device index");
    g_device = device;
  });
}
int pepshost_get_device(void) { return g_device; }

// d_min < 0: D_min = chi.  scheme: 0 SVD_COMPRESS, 1 VARIATION2Site, 2 VARIATION1Site (bmps.h:31-35)
int pepshost_set_truncate_params(int d_min, double trunc_err, int scheme, double convergence_tol, int iter_max) {
  return guarded([&]() {
    if (scheme < 0 || scheme > 2) throw std::invalid_argument("unknown CompressMPSScheme");
    g_trunc.d_min = d_min; g_trunc.trunc_err = trunc_err; g_trunc.scheme = scheme; g_trunc.tol = convergence_tol; g_trunc.iters = iter_max;
  });
}

// n_sweeps Monte-Carlo sweeps (square_nn_updater.h:30-81) of n walkers; updater 0 = NN exchange,
// 1 = NN full-space (Suwa-Todo).  configs are updated in place; one std::mt19937(seed[w]) per walker.
int pepshost_mc_sweeps(int rows, int cols, int D, int d, int chi, int dtype, const double *sitps_flat, int n,
                       int32_t *configs, const uint64_t *seeds, int updater, int n_sweeps, double *amplitudes_out,
                       double *accept_rates_out) {
  return guarded([&]() {
    mc_sweeps_impl<double>(rows, cols, D, d, chi, dtype, sitps_flat, n, configs, seeds, updater, n_sweeps, amplitudes_out, accept_rates_out);
  });
}
int pepshost_mc_sweeps_c128(int rows, int cols, int D, int d, int chi, const double *sitps_flat, int n,
                            int32_t *configs, const uint64_t *seeds, int updater, int n_sweeps, double *amplitudes_out,
                            double *accept_rates_out) {
  return guarded([&]() {
    mc_sweeps_impl<QLTEN_Complex>(rows, cols, D, d, chi, PEPSGPU_C128, sitps_flat, n, configs, seeds, updater, n_sweeps, amplitudes_out,
                                  accept_rates_out);
  });
}

// MonteCarloEngine (monte_carlo_engine.h:146-240, :340-414) on a walker batch: construction (configuration validity +
// rescue with the given amplitude window), WarmUp (warm-up sweeps with the NN exchange updater, sanity check,
// NormalizeStateOrder1).  In place: configs, sitps_flat (the scaled state).  out[0] = overall scale factor, out[1] = walkers
// rescued, out[2] = warmed up (0 / 1).
static int mc_engine_warmup_impl(int rows, int cols, int D, int d, int chi, int dtype, double *sitps_flat, int n, int32_t *configs,
                                 const uint64_t *seeds, int warmup_sweeps, int rescue_enabled, double amp_min, double amp_max,
                                 double *amplitudes_out, double *out, double (*max_over_ranks)(double),
                                 int (*exchange_valid_config)(int, int, int32_t *)) {
  return guarded([&]() {
    SplitIndexTPS sitps = make_state_t<double>(rows, cols, D, d, sitps_flat);
    BMPSContractor contractor(rows, cols, D, d, trunc_params(chi), n, dtype, g_device);
    TPSWaveFunctionComponent comp(sitps, make_cfg(n, rows, cols, configs), contractor);
    std::vector<uint64_t> sd(seeds, seeds + n);
    MCUpdateSquareNNExchangeOBC upd(sd);
    MonteCarloParams mp;
    mp.num_warmup_sweeps = (size_t)warmup_sweeps;
    ConfigurationRescueParams rp;
    rp.enabled = rescue_enabled != 0;
    if (amp_min > 0.0) rp.amplitude_min_threshold = amp_min;
    if (amp_max > 0.0) rp.amplitude_max_threshold = amp_max;
    std::function<double(double)> mx;
    std::function<int(int, int, int32_t *)> ex;
    if (max_over_ranks) mx = max_over_ranks;
    if (exchange_valid_config) ex = exchange_valid_config;
    MonteCarloEngine<MCUpdateSquareNNExchangeOBC> eng(sitps, comp, mp, upd, rp, mx, ex);
    const size_t rescued = eng.RescuedWalkers();
    eng.WarmUp();
    std::copy(comp.config.data(), comp.config.data() + (size_t)n * rows * cols, configs);
    std::copy(sitps.flat().begin(), sitps.flat().end(), sitps_flat);
    copy_out(comp.amplitude, amplitudes_out);
    out[0] = eng.LastScaleFactor(); out[1] = (double)rescued; out[2] = eng.IsWarmedUp() ? 1.0 : 0.0;
  });
}
int pepshost_mc_engine_warmup(int rows, int cols, int D, int d, int chi, int dtype, double *sitps_flat, int n, int32_t *configs,
                              const uint64_t *seeds, int warmup_sweeps, int rescue_enabled, double amp_min, double amp_max,
                              double *amplitudes_out, double *out) {
  return mc_engine_warmup_impl(rows, cols, D, d, chi, dtype, sitps_flat, n, configs, seeds, warmup_sweeps, rescue_enabled, amp_min, amp_max,
                               amplitudes_out, out, nullptr, nullptr);
}
// Several ranks (one process per GPU): max_over_ranks = the MPI_Allreduce(MAX) of NormalizeStateOrder1 (monte_carlo_engine.h:214-222),
// exchange_valid_config = the MPI_Allgather + MPI_BCast of EnsureConfigurationValidity (:344-387) -- both supplied by the caller
// (peps_amd/dist.py over torch.distributed); either may be NULL.
int pepshost_mc_engine_warmup_dist(int rows, int cols, int D, int d, int chi, int dtype, double *sitps_flat, int n, int32_t *configs,
                                   const uint64_t *seeds, int warmup_sweeps, int rescue_enabled, double amp_min, double amp_max,
                                   double *amplitudes_out, double *out, double (*max_over_ranks)(double),
                                   int (*exchange_valid_config)(int, int, int32_t *)) {
  return mc_engine_warmup_impl(rows, cols, D, d, chi, dtype, sitps_flat, n, configs, seeds, warmup_sweeps, rescue_enabled, amp_min, amp_max,
                               amplitudes_out, out, max_over_ranks, exchange_valid_config);
}

// CalEnergyAndHoles (model_energy_solver.h:32-126) for n configurations; model 0 = XXZ (p = jz, jxy,
// pinning00), 1 = TFIM (p[0] = h).  holes_out (nullable) = [n][rows][cols][D^4]; psi_out = [n_psi][n].
int pepshost_energy_and_holes(int rows, int cols, int D, int d, int chi, int dtype, const double *sitps_flat, int n,
                              const int32_t *configs, int model, const double *p, double *amplitudes_out,
                              double *energies_out, double *holes_out, double *psi_out, int *n_psi_out) {
  return guarded([&]() {
    energy_and_holes_impl<double>(rows, cols, D, d, chi, dtype, sitps_flat, n, configs, model, p, amplitudes_out, energies_out, holes_out,
                                  psi_out, n_psi_out);
  });
}
// the same for TenElemT = QLTEN_Complex: sitps_flat and every output scalar / tensor = interleaved (re, im) pairs
int pepshost_energy_and_holes_c128(int rows, int cols, int D, int d, int chi, const double *sitps_flat, int n,
                                   const int32_t *configs, int model, const double *p, double *amplitudes_out,
                                   double *energies_out, double *holes_out, double *psi_out, int *n_psi_out) {
  return guarded([&]() {
    energy_and_holes_impl<QLTEN_Complex>(rows, cols, D, d, chi, PEPSGPU_C128, sitps_flat, n, configs, model, p, amplitudes_out, energies_out,
                                         holes_out, psi_out, n_psi_out);
  });
}

// Rank-local part of MCEnergyGradEvaluator::Evaluate (mc_energy_grad_evaluator.h:205-310): warm-up
// sweeps, then n_samples x {one MC sweep, CalEnergyAndHoles, O* accumulation}; the holes and the
// tensor sums stay on the device.  packed_out as in pepshost_exact_sum_partial (weight 1 per sample);
// the caller all-reduces it over ranks and calls pepshost_exact_sum_finish.
int pepshost_mc_energy_grad_partial(int rows, int cols, int D, int d, int chi, int dtype, const double *sitps_flat, int n,
                                    int32_t *configs, const uint64_t *seeds, int updater, int model, const double *p,
                                    int warmup_sweeps, int n_samples, double *packed_out, double *accept_out) {
  return guarded([&]() {
    mc_energy_grad_partial_impl<double>(rows, cols, D, d, chi, dtype, sitps_flat, n, configs, seeds, updater, model, p, warmup_sweeps,
                                        n_samples, packed_out, accept_out);
  });
}
// QLTEN_Complex: packed_out = [S_O | S_EO (interleaved pairs) | sum w | Re, Im sum wE | sum w|E|^2 | samples], 4 m + 5 doubles
int pepshost_mc_energy_grad_partial_c128(int rows, int cols, int D, int d, int chi, const double *sitps_flat, int n,
                                         int32_t *configs, const uint64_t *seeds, int updater, int model, const double *p,
                                         int warmup_sweeps, int n_samples, double *packed_out, double *accept_out) {
  return guarded([&]() {
    mc_energy_grad_partial_impl<QLTEN_Complex>(rows, cols, D, d, chi, PEPSGPU_C128, sitps_flat, n, configs, seeds, updater, model, p,
                                               warmup_sweeps, n_samples, packed_out, accept_out);
  });
}

// EvaluateObservables of the XXZ (model 0) / TFIM (model 1, p[0] = h) / J1-J2 (model 2) / triangular J1-J2 (model 4, p[0] = j2) measurement solver on fixed configurations, or -- with
// n_samples > 0 -- a whole MCPEPSMeasurer run (warm-up, samples, statistics across the walkers, optional DumpData).
// Results come back through a flat buffer described by `keys_out` ("key:len;key:len;..."): for n_samples == 0 the
// per-walker values [key][walker][len], else [key][mean(len) | stderr(len)], followed by psi_mean[n], psi_rel_err[n]
// of the last sample.
int pepshost_measure(int rows, int cols, int D, int d, int chi, int dtype, const double *sitps_flat, int n, int32_t *configs,
                     const uint64_t *seeds, int updater, int model, const double *p, int warmup_sweeps, int n_samples,
                     int sweeps_between_samples, const char *dump_dir, char *keys_out, int keys_cap, double *values_out,
                     long values_cap, long *values_len) {
  return guarded([&]() {
    measure_impl<double>(rows, cols, D, d, chi, dtype, sitps_flat, n, configs, seeds, updater, model, p, warmup_sweeps, n_samples,
                         sweeps_between_samples, dump_dir, keys_out, keys_cap, values_out, values_cap, values_len);
  });
}
// QLTEN_Complex (ObservableMap<TenElemT> of the reference is complex for a complex state): sitps_flat and every number of values_out --
// observables, standard errors, psi_mean, psi_rel_err -- are interleaved (re, im) pairs; values_len counts doubles.
int pepshost_measure_c128(int rows, int cols, int D, int d, int chi, const double *sitps_flat, int n, int32_t *configs,
                          const uint64_t *seeds, int updater, int model, const double *p, int warmup_sweeps, int n_samples,
                          int sweeps_between_samples, const char *dump_dir, char *keys_out, int keys_cap, double *values_out,
                          long values_cap, long *values_len) {
  return guarded([&]() {
    measure_impl<QLTEN_Complex>(rows, cols, D, d, chi, PEPSGPU_C128, sitps_flat, n, configs, seeds, updater, model, p, warmup_sweeps, n_samples,
                                sweeps_between_samples, dump_dir, keys_out, keys_cap, values_out, values_cap, values_len);
  });
}

// Rank-local part of ExactSumEnergyEvaluatorMPI (exact_summation_energy_evaluator.h:173-245):
// packed_out = [S_O | S_EO | sum w | sum wE | sum wE^2 | samples], length 2*rows*cols*d*D^4 + 4.
int pepshost_exact_sum_partial(int rows, int cols, int D, int d, int chi, int dtype, const double *sitps_flat,
                               const int32_t *all_configs, int n_configs, int model, const double *p, int rank, int size,
                               int batch, double *packed_out) {
  return guarded([&]() {
    exact_sum_partial_impl<double>(rows, cols, D, d, chi, dtype, sitps_flat, all_configs, n_configs, model, p, rank, size, batch, packed_out);
  });
}
// QLTEN_Complex (layout of packed_out: see pepshost_mc_energy_grad_partial_c128)
int pepshost_exact_sum_partial_c128(int rows, int cols, int D, int d, int chi, const double *sitps_flat,
                                    const int32_t *all_configs, int n_configs, int model, const double *p, int rank, int size,
                                    int batch, double *packed_out) {
  return guarded([&]() {
    exact_sum_partial_impl<QLTEN_Complex>(rows, cols, D, d, chi, PEPSGPU_C128, sitps_flat, all_configs, n_configs, model, p, rank, size, batch,
                                          packed_out);
  });
}

// Rank-local part of ExactSumMeasurerMPI (exact_summation_measurer.h:103-257): un-normalised weighted sums of the
// registry observables over configurations rank, rank + size, ...   keys_out = "key:len;...", values_out =
// [sum w | key values in that order]; the caller sums values over ranks and divides by values[0].
// n_configs < 0: every binary configuration (GenerateAllBinaryConfigs).
int pepshost_exact_sum_measure_partial(int rows, int cols, int D, int d, int chi, int dtype, const double *sitps_flat,
                                       const int32_t *all_configs, int n_configs, int model, const double *p, int rank, int size,
                                       int batch, char *keys_out, int keys_cap, double *values_out, long values_cap, long *values_len) {
  return guarded([&]() {
    exact_sum_measure_partial_impl<double>(rows, cols, D, d, chi, dtype, sitps_flat, all_configs, n_configs, model, p, rank, size, batch,
                                           keys_out, keys_cap, values_out, values_cap, values_len);
  });
}
// QLTEN_Complex: values_out = [sum w (one double) | key values as (re, im) pairs]
int pepshost_exact_sum_measure_partial_c128(int rows, int cols, int D, int d, int chi, const double *sitps_flat,
                                            const int32_t *all_configs, int n_configs, int model, const double *p, int rank, int size,
                                            int batch, char *keys_out, int keys_cap, double *values_out, long values_cap, long *values_len) {
  return guarded([&]() {
    exact_sum_measure_partial_impl<QLTEN_Complex>(rows, cols, D, d, chi, PEPSGPU_C128, sitps_flat, all_configs, n_configs, model, p, rank, size,
                                                  batch, keys_out, keys_cap, values_out, values_cap, values_len);
  });
}

// energy = sum wE / sum w, gradient = (S_EO - E S_O)/sum w   (:286-295) from the rank-summed accumulators
int pepshost_exact_sum_finish(int rows, int cols, int D, int d, const double *packed, double *energy_out, double *grad_out) {
  return guarded([&]() {
    SplitIndexTPS like(rows, cols, d, D);
    GradAccumulator acc(like);
    std::vector<double> v(packed, packed + 2 * like.flat().size() + 4);
    acc.Unpack(v);
    auto res = acc.Finish();
    *energy_out = res.first;
    std::copy(res.second.flat().begin(), res.second.flat().end(), grad_out);
  });
}

// QLTEN_Complex: packed = 4 m + 5 doubles (see pepshost_mc_energy_grad_partial_c128); energy_out = (re, im), grad_out = interleaved pairs
int pepshost_exact_sum_finish_c128(int rows, int cols, int D, int d, const double *packed, double *energy_out, double *grad_out) {
  return guarded([&]() {
    SplitIndexTPST<QLTEN_Complex> like(rows, cols, d, D);
    GradAccumulatorT<QLTEN_Complex> acc(like);
    std::vector<double> v(packed, packed + 4 * like.flat().size() + 5);
    acc.Unpack(v);
    auto res = acc.Finish();
    energy_out[0] = res.first.real(); energy_out[1] = res.first.imag();
    copy_out(res.second.flat(), grad_out);
  });
}
// SplitIndexTPS<QLTEN_Complex>::Load / Dump of the reference's complex fixtures (interleaved complex128 payloads)
int pepshost_load_sitps_c128(const char *dir, int D, int *rows, int *cols, int *d, double *flat_out, size_t flat_cap) {
  return guarded([&]() {
    auto s = SplitIndexTPST<QLTEN_Complex>::Load(dir, D);
    *rows = (int)s.rows(); *cols = (int)s.cols(); *d = (int)s.PhysicalDim();
    if (flat_out) {
      if (flat_cap < 2 * s.flat().size()) throw std::invalid_argument("output buffer too small");
      copy_out(s.flat(), flat_out);
    }
  });
}
int pepshost_dump_sitps_c128(const char *dir, int rows, int cols, int D, int d, const double *flat) {
  return guarded([&]() { make_state_t<QLTEN_Complex>(rows, cols, D, d, flat).Dump(dir); });
}

// SplitIndexTPS::Load round trip for the reference's dump format (dense fixtures)
int pepshost_load_sitps(const char *dir, int D, int *rows, int *cols, int *d, double *flat_out, size_t flat_cap) {
  return guarded([&]() {
    SplitIndexTPS s = SplitIndexTPS::Load(dir, D);
    *rows = (int)s.rows(); *cols = (int)s.cols(); *d = (int)s.PhysicalDim();
    if (flat_out) {
      if (flat_cap < s.flat().size()) throw std::invalid_argument("output buffer too small");
      std::copy(s.flat().begin(), s.flat().end(), flat_out);
    }
  });
}

// Fermionic state (extended, sign-decorated components: peps_amd/fermion.py): E_loc of the spinless t-V model and
// the amplitudes for a batch of physical configurations.  sitps_ext_flat has 4 * d components per site.
// model 0: spinless t-V (params t, V); model 1: t-J-V (params t, J, V, mu)
int pepshost_fermion_energy(int rows, int cols, int D, int d, const int32_t *nf, int chi, int dtype,
                            const double *sitps_ext_flat, int n, const int32_t *configs, int model, const double *prm,
                            double *amplitudes_out, double *energies_out, double *psi_out, int *n_psi_out) {
  return guarded([&]() {
    fermion_energy_impl<double>(rows, cols, D, d, nf, chi, dtype, sitps_ext_flat, n, configs, model, prm, amplitudes_out, energies_out, psi_out,
                                n_psi_out);
  });
}
// QLTEN_Complex (SplitIndexTPS<QLTEN_Complex, fZ2QN>): the decorated components and every output as interleaved (re, im) pairs
int pepshost_fermion_energy_c128(int rows, int cols, int D, int d, const int32_t *nf, int chi,
                                 const double *sitps_ext_flat, int n, const int32_t *configs, int model, const double *prm,
                                 double *amplitudes_out, double *energies_out, double *psi_out, int *n_psi_out) {
  return guarded([&]() {
    fermion_energy_impl<QLTEN_Complex>(rows, cols, D, d, nf, chi, PEPSGPU_C128, sitps_ext_flat, n, configs, model, prm, amplitudes_out,
                                       energies_out, psi_out, n_psi_out);
  });
}

// ExactSumEnergyEvaluator on a fermionic state (spinless t-V): packed accumulators over the EXTENDED components
// (length 2 * rows*cols*4d*D^4 + 4); peps_amd.fermion.fold_gradient maps the finished gradient back
int pepshost_fermion_exact_sum_partial(int rows, int cols, int D, int d, const int32_t *nf, int chi, int dtype,
                                       const double *sitps_ext_flat, const int32_t *all_configs, int n_configs, double t,
                                       double V, int rank, int size, int batch, double *packed_out) {
  return guarded([&]() {
    fermion_exact_sum_partial_impl<double>(rows, cols, D, d, nf, chi, dtype, sitps_ext_flat, all_configs, n_configs, t, V, rank, size, batch,
                                           packed_out);
  });
}
// QLTEN_Complex: packed_out as pepshost_mc_energy_grad_partial_c128 (4 m + 5 doubles, m = rows*cols*4d*D^4); finish with
// pepshost_exact_sum_finish_c128 on d' = 4 d
int pepshost_fermion_exact_sum_partial_c128(int rows, int cols, int D, int d, const int32_t *nf, int chi,
                                            const double *sitps_ext_flat, const int32_t *all_configs, int n_configs, double t,
                                            double V, int rank, int size, int batch, double *packed_out) {
  return guarded([&]() {
    fermion_exact_sum_partial_impl<QLTEN_Complex>(rows, cols, D, d, nf, chi, PEPSGPU_C128, sitps_ext_flat, all_configs, n_configs, t, V, rank,
                                                  size, batch, packed_out);
  });
}

// MCUpdateSquareNNExchangeOBC on a fermionic state: n_sweeps sweeps, configurations updated in place
int pepshost_fermion_mc_sweeps(int rows, int cols, int D, int d, const int32_t *nf, int chi, int dtype,
                               const double *sitps_ext_flat, int n, int32_t *configs, const uint64_t *seeds, int n_sweeps,
                               double *amplitudes_out, double *accept_out) {
  return guarded([&]() {
    fermion_mc_sweeps_impl<double>(rows, cols, D, d, nf, chi, dtype, sitps_ext_flat, n, configs, seeds, n_sweeps, amplitudes_out, accept_out);
  });
}
int pepshost_fermion_mc_sweeps_c128(int rows, int cols, int D, int d, const int32_t *nf, int chi,
                                    const double *sitps_ext_flat, int n, int32_t *configs, const uint64_t *seeds, int n_sweeps,
                                    double *amplitudes_out, double *accept_out) {
  return guarded([&]() {
    fermion_mc_sweeps_impl<QLTEN_Complex>(rows, cols, D, d, nf, chi, PEPSGPU_C128, sitps_ext_flat, n, configs, seeds, n_sweeps, amplitudes_out,
                                          accept_out);
  });
}

// MCPEPSMeasurer's energy on a fermionic state with ONE random stream per walker over the whole run (monte_carlo_engine.h:146-176,
// monte_carlo_peps_measurer_impl.h:495-519): warmup_sweeps sweeps, the component rebuilt as NormalizeStateOrder1 does (the scale drops out
// of every ratio; the amplitude is evaluated afresh), then n_samples x {sweeps_between sweeps, E_loc}.  energies_out = [sample][walker].
// This is the call that reproduces the reference's fermionic regression value on the device (K9 of DESIGN 2: 6x6 fU1 t-J state, seed 42,
// -14.74320489110316; tests/test_gpu_fermion.py).
int pepshost_fermion_measure_energy(int rows, int cols, int D, int d, const int32_t *nf, int chi, int dtype,
                                    const double *sitps_ext_flat, int n, int32_t *configs, const uint64_t *seeds, int warmup_sweeps,
                                    int n_samples, int sweeps_between, int model, const double *prm, double *energies_out,
                                    double *accept_out) {
  return guarded([&]() {
    fermion_measure_energy_impl<double>(rows, cols, D, d, nf, chi, dtype, sitps_ext_flat, n, configs, seeds, warmup_sweeps, n_samples,
                                        sweeps_between, model, prm, energies_out, accept_out);
  });
}
int pepshost_fermion_measure_energy_c128(int rows, int cols, int D, int d, const int32_t *nf, int chi,
                                         const double *sitps_ext_flat, int n, int32_t *configs, const uint64_t *seeds, int warmup_sweeps,
                                         int n_samples, int sweeps_between, int model, const double *prm, double *energies_out,
                                         double *accept_out) {
  return guarded([&]() {
    fermion_measure_energy_impl<QLTEN_Complex>(rows, cols, D, d, nf, chi, PEPSGPU_C128, sitps_ext_flat, n, configs, seeds, warmup_sweeps,
                                               n_samples, sweeps_between, model, prm, energies_out, accept_out);
  });
}

// SplitIndexTPS::Dump (OBC leg dimensions) and the configuration{label} text files
int pepshost_dump_sitps(const char *dir, int rows, int cols, int D, int d, const double *flat) {
  return guarded([&]() {
    SplitIndexTPS s = make_state(rows, cols, D, d, flat);
    s.Dump(dir);
  });
}
int pepshost_dump_configuration(const char *dir, int label, int rows, int cols, const int32_t *config) {
  return guarded([&]() { DumpConfiguration(make_cfg(1, rows, cols, config), 0, dir, (size_t)label); });
}
// Configuration::Load (configuration.h:356-393): *loaded_out = 1 and the configuration in config_out, or *loaded_out = 0 (missing file, shape
// sidecar of another size, payload that does not parse) -- never an error code for those, as the reference never throws from Load.
// loaded_out == NULL: a configuration that cannot be loaded is an error (PEPSGPU_EEMPTY).
int pepshost_load_configuration2(const char *dir, int label, int rows, int cols, int32_t *config_out, int *loaded_out) {
  return guarded([&]() {
    Configuration c(1, rows, cols);
    const bool ok = LoadConfiguration(c, 0, dir, (size_t)label);
    if (loaded_out) *loaded_out = ok ? 1 : 0;
    else if (!ok) throw std::runtime_error(std::string("Configuration::Load failed: ") + dir + "/configuration" + std::to_string(label));
    if (ok) std::copy(c.data(), c.data() + (size_t)rows * cols, config_out);
  });
}
int pepshost_load_configuration(const char *dir, int label, int rows, int cols, int32_t *config_out) {
  return pepshost_load_configuration2(dir, label, rows, cols, config_out, nullptr);
}
// Configuration::StreamRead (configuration.h:446-455) of a text buffer: std::runtime_error (PEPSGPU_EEMPTY) when it holds too few numbers
int pepshost_configuration_from_text(const char *text, int rows, int cols, int32_t *config_out) {
  return guarded([&]() {
    Configuration c(1, rows, cols);
    std::istringstream iss(text);
    StreamReadConfiguration(c, 0, iss);
    std::copy(c.data(), c.data() + (size_t)rows * cols, config_out);
  });
}

// SuwaTodoStateUpdate (suwa_todo_update.h:53-112) alone, on the host: the chain state_{t+1} = SuwaTodoStateUpdate(state_t, weights, gen) of
// n_steps updates with gen = std::mt19937(seed).  No device call: what the reference's test_suwa_todo_update.cpp exercises (single state,
// zero weights, stationary distribution) and the deviate-for-deviate comparison with oracle/vmc.py run on the CPU suite.
int pepshost_suwa_todo_chain(int init_state, const double *weights, int n, uint64_t seed, long n_steps, int32_t *states_out) {
  return guarded([&]() {
    if (n <= 0 || init_state < 0 || init_state >= n) throw std::invalid_argument("pepshost_suwa_todo_chain: init_state outside the weights");
    std::vector<double> w(weights, weights + n);
    for (double x : w)
      if (!(x >= 0.0)) throw std::invalid_argument("pepshost_suwa_todo_chain: negative weight");
    if (!(w[init_state] > 0.0)) throw std::invalid_argument("pepshost_suwa_todo_chain: weights[init_state] must be positive");
    std::mt19937 gen((std::mt19937::result_type)seed);
    size_t state = (size_t)init_state;
    for (long t = 0; t < n_steps; ++t) {
      state = SuwaTodoStateUpdate(state, w, gen);
      states_out[t] = (int32_t)state;
    }
  });
}

}  // extern "C"
