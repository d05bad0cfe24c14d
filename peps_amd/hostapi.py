"""ctypes binding of libpepshost.so: the C++ host layer (peps_amd/host/qlpeps_gpu.h) that mirrors the
reference's updater / solver / evaluator surface on top of the pepsgpu C ABI."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libpepshost.so")

SYMBOLS = ["pepshost_last_error", "pepshost_mc_sweeps", "pepshost_energy_and_holes", "pepshost_exact_sum_partial", "pepshost_exact_sum_measure_partial",
           "pepshost_mc_energy_grad_partial", "pepshost_exact_sum_finish", "pepshost_load_sitps", "pepshost_dump_sitps",
           "pepshost_dump_configuration", "pepshost_load_configuration", "pepshost_fermion_energy",
           "pepshost_fermion_exact_sum_partial", "pepshost_fermion_mc_sweeps", "pepshost_measure",
           "pepshost_set_truncate_params", "pepshost_set_device", "pepshost_get_device",
           "pepshost_energy_and_holes_c128", "pepshost_exact_sum_partial_c128", "pepshost_exact_sum_finish_c128",
           "pepshost_mc_energy_grad_partial_c128", "pepshost_mc_sweeps_c128", "pepshost_load_sitps_c128", "pepshost_dump_sitps_c128",
           "pepshost_mc_engine_warmup", "pepshost_mc_engine_warmup_dist", "pepshost_suwa_todo_chain", "pepshost_load_configuration2", "pepshost_configuration_from_text",
           "pepshost_fermion_measure_energy", "pepshost_measure_c128", "pepshost_exact_sum_measure_partial_c128", "pepshost_fermion_energy_c128",
           "pepshost_fermion_exact_sum_partial_c128", "pepshost_fermion_mc_sweeps_c128", "pepshost_fermion_measure_energy_c128"]

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libpepshost.so not built: run __graft_entry__.build()")
        from . import capi
        capi.lib()                         # make sure libpepsgpu.so is resolvable first
        _lib = C.CDLL(LIB_PATH)
        _lib.pepshost_last_error.restype = C.c_char_p
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def _ck(rc):
    if rc != 0:
        msg = lib().pepshost_last_error().decode()
        raise {1: ValueError, 3: RuntimeError, 4: IndexError}.get(rc, RuntimeError)("pepshost error %d: %s" % (rc, msg))


MODEL_ID = {"xxz": 0, "tfim": 1, "j1j2": 2, "triangle": 3, "trij1j2": 4}   # params: xxz (jz, jxy, pinning00); tfim (h,); j1j2 (jz, jxy, jz2, jxy2, pinning00);
# triangle (SpinOneHalfTriHeisenbergSqrPEPS: none; energy_and_holes / exact_sum_partial); trij1j2 (SpinOneHalfTriJ1J2HeisenbergSqrPEPS: (j2,);
# energy_and_holes / exact_sum_partial / measure)


def _dims(flat):
    rows, cols, d, D = flat.shape[0], flat.shape[1], flat.shape[2], flat.shape[3]
    return rows, cols, d, D


def set_device(device=None):
    """GPU of every contractor the calls below build: one rank per GPU, so the default is LOCAL_RANK (else 0)."""
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0"))
    l = lib()
    l.pepshost_set_device.argtypes = [C.c_int]
    _ck(l.pepshost_set_device(int(device)))


def get_device():
    return lib().pepshost_get_device()


def set_truncate_params(d_min=-1, trunc_err=0.0, scheme=0, convergence_tol=0.0, iter_max=0):
    """BMPSTruncateParams used by every following call (D_max = the call's chi): SVD / Variational2Site / Variational1Site
    (bmps.h:47-98).  set_truncate_params() restores SVD(chi, chi, 0)."""
    l = lib()
    l.pepshost_set_truncate_params.argtypes = [C.c_int, C.c_double, C.c_int, C.c_double, C.c_int]
    _ck(l.pepshost_set_truncate_params(d_min, trunc_err, scheme, convergence_tol, iter_max))


def mc_sweeps(flat, configs, seeds, chi, updater="exchange", n_sweeps=1, dtype=1):
    flat = np.ascontiguousarray(flat, dtype=np.float64)
    rows, cols, d, D = _dims(flat)
    cfg = np.ascontiguousarray(configs, dtype=np.int32).copy()
    n = cfg.shape[0]
    sd = np.ascontiguousarray(seeds, dtype=np.uint64)
    amps = np.zeros(n)
    rates = np.zeros(n)
    _ck(lib().pepshost_mc_sweeps(rows, cols, D, d, chi, dtype, _p(flat, C.c_double), n, _p(cfg, C.c_int32),
                                 _p(sd, C.c_uint64), 0 if updater == "exchange" else 1, n_sweeps,
                                 _p(amps, C.c_double), _p(rates, C.c_double)))
    return cfg, amps, rates


def mc_engine_warmup(flat, configs, seeds, chi, warmup_sweeps=0, rescue=True, amp_min=0.0, amp_max=0.0, dtype=1, max_over_ranks=None,
                     exchange_valid_config=None):
    """MonteCarloEngine: configuration validity / rescue, warm-up, NormalizeStateOrder1 (monte_carlo_engine.h).
    Returns (scaled state, configs, amplitudes, overall scale factor, walkers rescued).
    Several ranks: max_over_ranks(x) -> max over the ranks (the MPI_Allreduce of NormalizeStateOrder1) and
    exchange_valid_config(n_invalid, have_valid, cfg) -> (total invalid | -1, cfg of the first valid rank) (the Allgather + BCast of
    EnsureConfigurationValidity); peps_amd.dist has both (allreduce_max, exchange_valid_configuration).  Both are collectives: every
    rank must pass them."""
    st = np.array(flat, dtype=np.float64, order="C")
    rows, cols, d, D = _dims(st)
    cfg = np.array(configs, dtype=np.int32, order="C")
    n = cfg.shape[0]
    sd = np.ascontiguousarray(seeds, dtype=np.uint64)
    amps, out = np.zeros(n), np.zeros(3)
    if max_over_ranks is None and exchange_valid_config is None:
        _ck(lib().pepshost_mc_engine_warmup(rows, cols, D, d, chi, dtype, _p(st, C.c_double), n, _p(cfg, C.c_int32), _p(sd, C.c_uint64),
                                            warmup_sweeps, int(rescue), C.c_double(amp_min), C.c_double(amp_max), _p(amps, C.c_double),
                                            _p(out, C.c_double)))
        return st, cfg, amps, float(out[0]), int(out[1])
    MAXF = C.CFUNCTYPE(C.c_double, C.c_double)
    EXF = C.CFUNCTYPE(C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32))
    sites = rows * cols

    def _ex(n_invalid, have_valid, ptr):
        mine = np.ctypeslib.as_array(ptr, shape=(sites,))
        tot, got = exchange_valid_config(int(n_invalid), bool(have_valid), mine.copy())
        mine[:] = np.asarray(got, dtype=np.int32).ravel()
        return int(tot)

    mx = MAXF(lambda x: float(max_over_ranks(float(x)))) if max_over_ranks is not None else C.cast(None, MAXF)
    ex = EXF(_ex) if exchange_valid_config is not None else C.cast(None, EXF)
    _ck(lib().pepshost_mc_engine_warmup_dist(rows, cols, D, d, chi, dtype, _p(st, C.c_double), n, _p(cfg, C.c_int32), _p(sd, C.c_uint64),
                                             warmup_sweeps, int(rescue), C.c_double(amp_min), C.c_double(amp_max), _p(amps, C.c_double),
                                             _p(out, C.c_double), mx, ex))
    return st, cfg, amps, float(out[0]), int(out[1])


def energy_and_holes(flat, configs, chi, model="xxz", params=(1.0, 1.0, 0.0), holes=True, dtype=1):
    flat = np.ascontiguousarray(flat, dtype=np.float64)
    rows, cols, d, D = _dims(flat)
    cfg = np.ascontiguousarray(configs, dtype=np.int32)
    n = cfg.shape[0]
    p = np.array(list(params) + [0.0] * 8, dtype=np.float64)
    amps, en = np.zeros(n), np.zeros(n)
    h = np.zeros((n, rows, cols, D, D, D, D)) if holes else None
    psi = np.zeros((rows + cols, n))
    npsi = C.c_int(0)
    _ck(lib().pepshost_energy_and_holes(rows, cols, D, d, chi, dtype, _p(flat, C.c_double), n, _p(cfg, C.c_int32),
                                        MODEL_ID[model], _p(p, C.c_double), _p(amps, C.c_double),
                                        _p(en, C.c_double), _p(h, C.c_double), _p(psi, C.c_double), C.byref(npsi)))
    return amps, en, h, psi[:npsi.value]


def exact_sum_partial(flat, all_configs, chi, model="xxz", params=(1.0, 1.0, 0.0), rank=0, size=1, batch=64, dtype=1):
    flat = np.ascontiguousarray(flat, dtype=np.float64)
    rows, cols, d, D = _dims(flat)
    cfg = np.ascontiguousarray(all_configs, dtype=np.int32)
    p = np.array(list(params) + [0.0] * 8, dtype=np.float64)
    packed = np.zeros(2 * flat.size + 4)
    _ck(lib().pepshost_exact_sum_partial(rows, cols, D, d, chi, dtype, _p(flat, C.c_double), _p(cfg, C.c_int32),
                                         cfg.shape[0], MODEL_ID[model], _p(p, C.c_double), rank, size, batch,
                                         _p(packed, C.c_double)))
    return packed


def exact_sum_measure_partial(flat, all_configs, chi, model="xxz", params=(1.0, 1.0, 0.0), rank=0, size=1, batch=64, dtype=1):
    """Rank-local part of ExactSumMeasurerMPI (exact_summation_measurer.h:103-257) through the C++ measurement solver:
    (dict key -> sum_S |psi(S)|^2 O_loc(S), sum_S |psi(S)|^2) over configurations rank, rank + size, ...  Sum both over
    the ranks and divide.  all_configs = None: every binary configuration (GenerateAllBinaryConfigs, :53-72).
    A complex `flat` runs the QLTEN_Complex instantiation (float64 arithmetic; `dtype` is ignored): complex sums."""
    cplx = np.iscomplexobj(flat)
    flat = np.ascontiguousarray(flat, dtype=np.complex128 if cplx else np.float64)
    rows, cols, d, D = _dims(flat)
    cfg = None if all_configs is None else np.ascontiguousarray(all_configs, dtype=np.int32)
    p = np.zeros(8, dtype=np.float64)
    p[:len(params)] = params
    cap = 4 * (rows * cols) ** 2 + 65536
    vals = np.zeros(cap, dtype=np.float64)
    keys = C.create_string_buffer(4096)
    nvals = C.c_long(0)
    l = lib()
    l.pepshost_exact_sum_measure_partial.argtypes = [C.c_int] * 6 + [C.POINTER(C.c_double), C.POINTER(C.c_int32), C.c_int, C.c_int,
                                                     C.POINTER(C.c_double), C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int,
                                                     C.POINTER(C.c_double), C.c_long, C.POINTER(C.c_long)]
    l.pepshost_exact_sum_measure_partial_c128.argtypes = [C.c_int] * 5 + l.pepshost_exact_sum_measure_partial.argtypes[6:]
    if cplx:
        _ck(l.pepshost_exact_sum_measure_partial_c128(rows, cols, D, d, chi, _cp(flat),
                                                      None if cfg is None else _p(cfg, C.c_int32), -1 if cfg is None else cfg.shape[0],
                                                      MODEL_ID[model], _p(p, C.c_double), rank, size, batch, keys, 4096,
                                                      _p(vals, C.c_double), cap, C.byref(nvals)))
    else:
        _ck(l.pepshost_exact_sum_measure_partial(rows, cols, D, d, chi, dtype, _p(flat, C.c_double),
                                                 None if cfg is None else _p(cfg, C.c_int32), -1 if cfg is None else cfg.shape[0],
                                                 MODEL_ID[model], _p(p, C.c_double), rank, size, batch, keys, 4096,
                                                 _p(vals, C.c_double), cap, C.byref(nvals)))
    out, off, z = {}, 1, 2 if cplx else 1
    for item in filter(None, keys.value.decode().split(";")):    # a rank without configurations reports no keys
        key, ln = item.split(":")
        seg = vals[off:off + z * int(ln)].copy()
        out[key] = seg.view(np.complex128) if cplx else seg
        off += z * int(ln)
    return out, float(vals[0])


def exact_sum_finish(packed, shape):
    rows, cols, d, D = shape[:4]
    packed = np.ascontiguousarray(packed, dtype=np.float64)
    e = C.c_double(0)
    grad = np.zeros((rows, cols, d, D, D, D, D))
    _ck(lib().pepshost_exact_sum_finish(rows, cols, D, d, _p(packed, C.c_double), C.byref(e), _p(grad, C.c_double)))
    return e.value, grad


def load_sitps(directory, D):
    rows, cols, d = C.c_int(0), C.c_int(0), C.c_int(0)
    _ck(lib().pepshost_load_sitps(directory.encode(), D, C.byref(rows), C.byref(cols), C.byref(d), None, 0))
    flat = np.zeros((rows.value, cols.value, d.value, D, D, D, D))
    _ck(lib().pepshost_load_sitps(directory.encode(), D, C.byref(rows), C.byref(cols), C.byref(d),
                                  _p(flat, C.c_double), flat.size))
    return flat


def mc_energy_grad_partial(flat, configs, seeds, chi, updater="exchange", model="xxz", params=(1.0, 1.0, 0.0),
                           warmup_sweeps=1, n_samples=1, dtype=1):
    """Rank-local MCEnergyGradEvaluator loop; returns (packed accumulators, final configs, accept rates)."""
    flat = np.ascontiguousarray(flat, dtype=np.float64)
    rows, cols, d, D = _dims(flat)
    cfg = np.ascontiguousarray(configs, dtype=np.int32).copy()
    n = cfg.shape[0]
    sd = np.ascontiguousarray(seeds, dtype=np.uint64)
    p = np.array(list(params) + [0.0] * 8, dtype=np.float64)
    packed = np.zeros(2 * flat.size + 4)
    acc = np.zeros(n)
    _ck(lib().pepshost_mc_energy_grad_partial(rows, cols, D, d, chi, dtype, _p(flat, C.c_double), n, _p(cfg, C.c_int32),
                                              _p(sd, C.c_uint64), 0 if updater == "exchange" else 1,
                                              MODEL_ID[model], _p(p, C.c_double), warmup_sweeps, n_samples,
                                              _p(packed, C.c_double), _p(acc, C.c_double)))
    return packed, cfg, acc


def measure(flat, configs, chi, model="xxz", params=(1.0, 1.0, 0.0), seeds=None, updater="exchange", warmup_sweeps=0,
            n_samples=0, sweeps_between_samples=1, dump_dir="", dtype=1):
    """Registry observables of the C++ measurement solver (SquareNNNModelMeasurementSolver::EvaluateObservables,
    square_nnn_model_measurement_solver.h) on fixed configurations (n_samples = 0: dict key -> [walker][len]) or a whole
    MCPEPSMeasurer run (n_samples > 0: dict key -> (mean[len], stderr[len]) across the walkers; configs updated in place;
    dump_dir: stats/*.csv + samples/psi.csv as the reference writes them).  Also returns the psi summary of the last
    sample: (psi_mean[walker], psi_rel_err[walker]).  A complex `flat` runs the QLTEN_Complex instantiation: complex values and means,
    real standard errors."""
    cplx = np.iscomplexobj(flat)
    flat = np.ascontiguousarray(flat, dtype=np.complex128 if cplx else np.float64)
    rows, cols, d, D = _dims(flat)
    cfg = np.ascontiguousarray(configs, dtype=np.int32)
    n = cfg.shape[0]
    sd = np.ascontiguousarray(np.zeros(n) if seeds is None else seeds, dtype=np.uint64)
    p = np.zeros(8, dtype=np.float64)
    p[:len(params)] = params
    cap = 8 * n * (rows * cols) ** 2 + 65536
    vals = np.zeros(cap, dtype=np.float64)
    keys = C.create_string_buffer(4096)
    nvals = C.c_long(0)
    l = lib()
    l.pepshost_measure.argtypes = [C.c_int] * 6 + [C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_uint64),
                                   C.c_int, C.c_int, C.POINTER(C.c_double), C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_char_p,
                                   C.c_int, C.POINTER(C.c_double), C.c_long, C.POINTER(C.c_long)]
    l.pepshost_measure_c128.argtypes = [C.c_int] * 5 + l.pepshost_measure.argtypes[6:]
    if cplx:
        _ck(l.pepshost_measure_c128(rows, cols, D, d, chi, _cp(flat), n, _p(cfg, C.c_int32), _p(sd, C.c_uint64),
                                    {"exchange": 0, "fullspace": 1}[updater], MODEL_ID[model], _p(p, C.c_double), warmup_sweeps,
                                    n_samples, sweeps_between_samples, dump_dir.encode(), keys, 4096, _p(vals, C.c_double), cap,
                                    C.byref(nvals)))
        vals = vals[:nvals.value].view(np.complex128)          # every number is a (re, im) pair
    else:
        _ck(l.pepshost_measure(rows, cols, D, d, chi, dtype, _p(flat, C.c_double), n, _p(cfg, C.c_int32), _p(sd, C.c_uint64),
                               {"exchange": 0, "fullspace": 1}[updater], MODEL_ID[model], _p(p, C.c_double), warmup_sweeps,
                               n_samples, sweeps_between_samples, dump_dir.encode(), keys, 4096, _p(vals, C.c_double), cap,
                               C.byref(nvals)))
    out, off = {}, 0
    for item in keys.value.decode().strip(";").split(";"):
        key, ln = item.split(":")
        ln = int(ln)
        if n_samples > 0:
            out[key] = (vals[off:off + ln].copy(), np.real(vals[off + ln:off + 2 * ln]).copy())
            off += 2 * ln
        else:
            out[key] = vals[off:off + n * ln].reshape(n, ln).copy()
            off += n * ln
    psi = (vals[off:off + n].copy(), np.real(vals[off + n:off + 2 * n]).copy())
    if n_samples > 0:
        configs[...] = cfg
    return out, psi


def dump_sitps(directory, flat):
    """SplitIndexTPS::Dump (split_index_tps_impl.h:300-330) of a padded OBC state."""
    flat = np.ascontiguousarray(flat, dtype=np.float64)
    rows, cols, d, D = _dims(flat)
    os.makedirs(directory, exist_ok=True)
    _ck(lib().pepshost_dump_sitps(directory.encode(), rows, cols, D, d, _p(flat, C.c_double)))


def dump_configuration(directory, label, config):
    cfg = np.ascontiguousarray(config, dtype=np.int32)
    _ck(lib().pepshost_dump_configuration(directory.encode(), label, cfg.shape[0], cfg.shape[1], _p(cfg, C.c_int32)))


def load_configuration(directory, label, rows, cols):
    out = np.zeros((rows, cols), dtype=np.int32)
    _ck(lib().pepshost_load_configuration(directory.encode(), label, rows, cols, _p(out, C.c_int32)))
    return out


def try_load_configuration(directory, label, rows, cols):
    """Configuration::Load (configuration.h:356-393): the configuration, or None where the reference returns false (missing file, `.shape`
    sidecar of another size, payload that does not parse) -- no exception for those."""
    out = np.zeros((rows, cols), dtype=np.int32)
    ok = C.c_int(0)
    _ck(lib().pepshost_load_configuration2(directory.encode(), label, rows, cols, _p(out, C.c_int32), C.byref(ok)))
    return out if ok.value else None


def configuration_from_text(text, rows, cols):
    """Configuration::StreamRead (configuration.h:446-455); RuntimeError when the text holds too few numbers"""
    out = np.zeros((rows, cols), dtype=np.int32)
    _ck(lib().pepshost_configuration_from_text(text.encode(), rows, cols, _p(out, C.c_int32)))
    return out


def fermion_energy(state, configs, chi, t, V=0.0, dtype=1, model="spinless", J=0.0, mu=0.0, t2=0.0):
    """C++ host layer on a fermionic state (peps_amd.fermion.FermionState): amplitudes (row-major mode order),
    E_loc of the spinless t-V model (model="spinless") or of the t-J-V model (model="tj"), psi along every route.
    A complex state (SplitIndexTPS<QLTEN_Complex, fZ2QN>) runs the complex instantiation in float64 (`dtype` ignored)."""
    cplx = state.is_complex
    et = np.complex128 if cplx else np.float64
    flat = np.ascontiguousarray(state.extended_flat(), dtype=et)
    rows, cols, D = flat.shape[0], flat.shape[1], flat.shape[3]
    cfg = np.ascontiguousarray(configs, dtype=np.int32)
    n = cfg.shape[0]
    nf = np.ascontiguousarray(state.nf, dtype=np.int32)
    amps, en = np.zeros(n, et), np.zeros(n, et)
    psi = np.zeros((rows + cols, n), et)
    npsi = C.c_int(0)
    prm = np.array([t, V, t2, 0.0] if model == "spinless" else [t, J, V, mu], dtype=np.float64)
    if cplx:
        _ck(lib().pepshost_fermion_energy_c128(rows, cols, D, state.d, _p(nf, C.c_int32), chi, _cp(flat), n,
                                               _p(cfg, C.c_int32), 0 if model == "spinless" else 1, _p(prm, C.c_double),
                                               _cp(amps), _cp(en), _cp(psi), C.byref(npsi)))
    else:
        _ck(lib().pepshost_fermion_energy(rows, cols, D, state.d, _p(nf, C.c_int32), chi, dtype, _p(flat, C.c_double), n,
                                          _p(cfg, C.c_int32), 0 if model == "spinless" else 1, _p(prm, C.c_double),
                                          _p(amps, C.c_double), _p(en, C.c_double), _p(psi, C.c_double), C.byref(npsi)))
    return amps, en, psi[:npsi.value]


def fermion_mc_sweeps(state, configs, seeds, chi, n_sweeps=1, dtype=1):
    cplx = state.is_complex
    et = np.complex128 if cplx else np.float64
    flat = np.ascontiguousarray(state.extended_flat(), dtype=et)
    rows, cols, D = flat.shape[0], flat.shape[1], flat.shape[3]
    cfg = np.ascontiguousarray(configs, dtype=np.int32).copy()
    n = cfg.shape[0]
    nf = np.ascontiguousarray(state.nf, dtype=np.int32)
    sd = np.ascontiguousarray(seeds, dtype=np.uint64)
    amps, rates = np.zeros(n, et), np.zeros(n)
    if cplx:
        _ck(lib().pepshost_fermion_mc_sweeps_c128(rows, cols, D, state.d, _p(nf, C.c_int32), chi, _cp(flat), n,
                                                  _p(cfg, C.c_int32), _p(sd, C.c_uint64), n_sweeps, _cp(amps), _p(rates, C.c_double)))
    else:
        _ck(lib().pepshost_fermion_mc_sweeps(rows, cols, D, state.d, _p(nf, C.c_int32), chi, dtype, _p(flat, C.c_double), n,
                                             _p(cfg, C.c_int32), _p(sd, C.c_uint64), n_sweeps, _p(amps, C.c_double),
                                             _p(rates, C.c_double)))
    return cfg, amps, rates


def fermion_measure_energy(state, configs, seeds, chi, warmup_sweeps, n_samples, sweeps_between=1, model="tj", t=1.0, J=0.0, V=0.0, mu=0.0,
                           t2=0.0, dtype=1):
    """MCPEPSMeasurer's energy samples on a fermionic state with ONE std::mt19937 stream per walker over warm-up, rebuild and samples
    (pepshost_fermion_measure_energy): (energies [sample][walker], final configurations, accept rates)."""
    cplx = state.is_complex
    et = np.complex128 if cplx else np.float64
    flat = np.ascontiguousarray(state.extended_flat(), dtype=et)
    rows, cols, D = flat.shape[0], flat.shape[1], flat.shape[3]
    cfg = np.ascontiguousarray(configs, dtype=np.int32).copy()
    n = cfg.shape[0]
    nf = np.ascontiguousarray(state.nf, dtype=np.int32)
    sd = np.ascontiguousarray(seeds, dtype=np.uint64)
    prm = np.array([t, J, V, mu] if model == "tj" else [t, V, t2, 0.0], dtype=np.float64)
    en, rates = np.zeros((n_samples, n), et), np.zeros(n)
    if cplx:
        _ck(lib().pepshost_fermion_measure_energy_c128(rows, cols, D, state.d, _p(nf, C.c_int32), chi, _cp(flat), n,
                                                       _p(cfg, C.c_int32), _p(sd, C.c_uint64), warmup_sweeps, n_samples, sweeps_between,
                                                       1 if model == "tj" else 0, _p(prm, C.c_double), _cp(en), _p(rates, C.c_double)))
    else:
        _ck(lib().pepshost_fermion_measure_energy(rows, cols, D, state.d, _p(nf, C.c_int32), chi, dtype, _p(flat, C.c_double), n,
                                                  _p(cfg, C.c_int32), _p(sd, C.c_uint64), warmup_sweeps, n_samples, sweeps_between,
                                                  1 if model == "tj" else 0, _p(prm, C.c_double), _p(en, C.c_double), _p(rates, C.c_double)))
    return en, cfg, rates


def fermion_exact_sum(state, all_configs, chi, t, V=0.0, batch=64, dtype=1):
    """ExactSumEnergyEvaluator on a fermionic state: (energy, gradient [rows][cols][d][D^4] with respect to the
    stored site-tensor components, zero on parity-forbidden entries)."""
    from . import fermion
    if state.is_complex:                    # QLTEN_Complex: packed = 4 m + 5 doubles, finished by pepshost_exact_sum_finish_c128 on 4 d components
        flat = np.ascontiguousarray(state.extended_flat(), dtype=np.complex128)
        rows, cols, D = flat.shape[0], flat.shape[1], flat.shape[3]
        cfg = np.ascontiguousarray(all_configs, dtype=np.int32)
        nf = np.ascontiguousarray(state.nf, dtype=np.int32)
        packed = np.zeros(4 * flat.size + 5)
        _ck(lib().pepshost_fermion_exact_sum_partial_c128(rows, cols, D, state.d, _p(nf, C.c_int32), chi, _cp(flat),
                                                          _p(cfg, C.c_int32), cfg.shape[0], C.c_double(t), C.c_double(V), 0, 1, batch,
                                                          _p(packed, C.c_double)))
        e = np.zeros(2)
        grad = np.zeros(flat.shape, dtype=np.complex128)
        _ck(lib().pepshost_exact_sum_finish_c128(rows, cols, D, 4 * state.d, _p(packed, C.c_double), _p(e, C.c_double), _cp(grad)))
        return complex(e[0], e[1]), fermion.fold_gradient(state, grad)
    flat = np.ascontiguousarray(state.extended_flat(), dtype=np.float64)
    rows, cols, D = flat.shape[0], flat.shape[1], flat.shape[3]
    cfg = np.ascontiguousarray(all_configs, dtype=np.int32)
    nf = np.ascontiguousarray(state.nf, dtype=np.int32)
    packed = np.zeros(2 * flat.size + 4)
    _ck(lib().pepshost_fermion_exact_sum_partial(rows, cols, D, state.d, _p(nf, C.c_int32), chi, dtype, _p(flat, C.c_double),
                                                 _p(cfg, C.c_int32), cfg.shape[0], C.c_double(t), C.c_double(V), 0, 1, batch,
                                                 _p(packed, C.c_double)))
    e, grad_ext = exact_sum_finish(packed, (rows, cols, 4 * state.d, D))
    return e, fermion.fold_gradient(state, grad_ext)


# ---- TenElemT = QLTEN_Complex (std::complex<double>): the same host classes instantiated for the complex element type ----
def _cflat(flat):
    return np.ascontiguousarray(flat, dtype=np.complex128)


def _cp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def load_sitps_complex(directory, D):
    rows, cols, d = C.c_int(0), C.c_int(0), C.c_int(0)
    _ck(lib().pepshost_load_sitps_c128(directory.encode(), D, C.byref(rows), C.byref(cols), C.byref(d), None, 0))
    flat = np.zeros((rows.value, cols.value, d.value, D, D, D, D), dtype=np.complex128)
    _ck(lib().pepshost_load_sitps_c128(directory.encode(), D, C.byref(rows), C.byref(cols), C.byref(d), _cp(flat), C.c_size_t(2 * flat.size)))
    return flat


def dump_sitps_complex(directory, flat):
    flat = _cflat(flat)
    rows, cols, d, D = _dims(flat)
    os.makedirs(directory, exist_ok=True)
    _ck(lib().pepshost_dump_sitps_c128(directory.encode(), rows, cols, D, d, _cp(flat)))


def energy_and_holes_complex(flat, configs, chi, model="xxz", params=(1.0, 1.0, 0.0), holes=True):
    """CalEnergyAndHoles of the C++ host layer on a COMPLEX state (BMPSContractorT<QLTEN_Complex>, PEPSGPU_C128 context):
    (amplitudes, energies, holes = Dag(hole) [n][rows][cols][D^4] or None, psi_list [n_psi][n]), all complex128."""
    flat = _cflat(flat)
    rows, cols, d, D = _dims(flat)
    cfg = np.ascontiguousarray(configs, dtype=np.int32)
    n = cfg.shape[0]
    p = np.array(list(params) + [0.0] * 8, dtype=np.float64)
    amps = np.zeros(n, dtype=np.complex128)
    en = np.zeros(n, dtype=np.complex128)
    hl = np.zeros((n, rows, cols, D, D, D, D), dtype=np.complex128) if holes else None
    psi = np.zeros((rows + cols, n), dtype=np.complex128)
    npsi = C.c_int(0)
    _ck(lib().pepshost_energy_and_holes_c128(rows, cols, D, d, chi, _cp(flat), n, _p(cfg, C.c_int32), MODEL_ID[model], _p(p, C.c_double),
                                             _cp(amps), _cp(en), _cp(hl) if holes else None, _cp(psi), C.byref(npsi)))
    return amps, en, hl, psi[:npsi.value]


def exact_sum_complex(flat, all_configs, chi, model="xxz", params=(1.0, 1.0, 0.0), size=1, batch=64):
    """ExactSumEnergyEvaluator on a complex state, the `size` rank-partials summed here: (energy complex, gradient complex128
    in the upload layout).  Holes resident in HBM, O* accumulation with the conjugations of
    exact_summation_energy_evaluator.h:228-240 on the device (pepsgpu_grad_accumulate for PEPSGPU_C128)."""
    flat = _cflat(flat)
    rows, cols, d, D = _dims(flat)
    cfg = np.ascontiguousarray(all_configs, dtype=np.int32)
    p = np.array(list(params) + [0.0] * 8, dtype=np.float64)
    m = flat.size
    tot = np.zeros(4 * m + 5)
    for rank in range(size):
        packed = np.zeros(4 * m + 5)
        _ck(lib().pepshost_exact_sum_partial_c128(rows, cols, D, d, chi, _cp(flat), _p(cfg, C.c_int32), cfg.shape[0], MODEL_ID[model],
                                                  _p(p, C.c_double), rank, size, batch, _p(packed, C.c_double)))
        tot += packed
    e = np.zeros(2)
    grad = np.zeros(flat.shape, dtype=np.complex128)
    _ck(lib().pepshost_exact_sum_finish_c128(rows, cols, D, d, _p(tot, C.c_double), _p(e, C.c_double), _cp(grad)))
    return complex(e[0], e[1]), grad


def mc_sweeps_complex(flat, configs, seeds, chi, updater="exchange", n_sweeps=1):
    flat = _cflat(flat)
    rows, cols, d, D = _dims(flat)
    cfg = np.ascontiguousarray(configs, dtype=np.int32).copy()
    n = cfg.shape[0]
    sd = np.ascontiguousarray(seeds, dtype=np.uint64)
    amps = np.zeros(n, dtype=np.complex128)
    rates = np.zeros(n)
    _ck(lib().pepshost_mc_sweeps_c128(rows, cols, D, d, chi, _cp(flat), n, _p(cfg, C.c_int32), _p(sd, C.c_uint64),
                                      0 if updater == "exchange" else 1, n_sweeps, _cp(amps), _p(rates, C.c_double)))
    return cfg, amps, rates


def mc_energy_grad_complex(flat, configs, seeds, chi, updater="exchange", model="xxz", params=(1.0, 1.0, 0.0), warmup_sweeps=1, n_samples=1):
    """MCEnergyGradEvaluator loop on a complex state: (energy, gradient, final configs, accept rates)"""
    flat = _cflat(flat)
    rows, cols, d, D = _dims(flat)
    cfg = np.ascontiguousarray(configs, dtype=np.int32).copy()
    n = cfg.shape[0]
    sd = np.ascontiguousarray(seeds, dtype=np.uint64)
    p = np.array(list(params) + [0.0] * 8, dtype=np.float64)
    packed = np.zeros(4 * flat.size + 5)
    acc = np.zeros(n)
    _ck(lib().pepshost_mc_energy_grad_partial_c128(rows, cols, D, d, chi, _cp(flat), n, _p(cfg, C.c_int32), _p(sd, C.c_uint64),
                                                   0 if updater == "exchange" else 1, MODEL_ID[model], _p(p, C.c_double), warmup_sweeps,
                                                   n_samples, _p(packed, C.c_double), _p(acc, C.c_double)))
    e = np.zeros(2)
    grad = np.zeros(flat.shape, dtype=np.complex128)
    _ck(lib().pepshost_exact_sum_finish_c128(rows, cols, D, d, _p(packed, C.c_double), _p(e, C.c_double), _cp(grad)))
    return complex(e[0], e[1]), grad, cfg, acc


def suwa_todo_chain(init_state, weights, seed, n_steps):
    """The chain of n_steps SuwaTodoStateUpdate calls (suwa_todo_update.h:53-112) with std::mt19937(seed), on the host alone (no GPU):
    states[t] for t = 1..n_steps."""
    w = np.ascontiguousarray(weights, dtype=np.float64)
    out = np.zeros(int(n_steps), dtype=np.int32)
    _ck(lib().pepshost_suwa_todo_chain(int(init_state), _p(w, C.c_double), int(w.size), C.c_uint64(int(seed)), C.c_long(int(n_steps)),
                                       _p(out, C.c_int32)))
    return out
