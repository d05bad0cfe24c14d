"""Fermionic (fZ2-graded) states on the device: sign-decorated components through the unchanged bosonic
engine (peps_amd/fermion.py), against the graded oracle (oracle/graded.py, oracle/fermion.py) and the
reference's 2x2 spinless-fermion known answers; BASELINE config C5 (8x8 spinless t-V, D=6, chi=24)."""
import itertools
import json
import os

import numpy as np
import pytest

from oracle import fermion as ofermion, graded
from oracle.bmps import BMPSTruncateParams
from oracle.graded import GT

pytestmark = pytest.mark.gpu


def _ctx(state, chi, dtype, n):
    from peps_amd import capi
    D = state.D
    ctx = capi.Context(state.rows, state.cols, D, 4 * state.d, chi, dtype=dtype, max_walkers=n)
    ctx.state_upload(state.extended_flat(D))
    return ctx


def _oracle_view(state):
    gts = [[[GT(state.tensors[r][c][s][..., None], list(state.par[r][c]) + [np.array([int(state.nf[s])])], [-1, 1, 1, -1, -1])
             for s in range(state.d)] for c in range(state.cols)] for r in range(state.rows)]
    return gts, ofermion.FermionSITPS(gts)


def _half_filling_configs():
    return np.array([np.array(p).reshape(2, 2) for p in sorted(set(itertools.permutations([0, 0, 1, 1])))])


@pytest.mark.parametrize("name,t2,e_ref", [("0.000000_doublelowest", 0.0, -2.0), ("0.000000_double_from_simple_update", 0.0, -1.98218053854),
                                           ("2.100000_doublelowest", 2.1, -4.2), ("2.100000_double_from_simple_update", 2.1, -4.1879072654),
                                           ("-2.500000_doublelowest", -2.5, -5.0), ("-2.500000_double_from_simple_update", -2.5, -4.98966397657)])
def test_k4_spinless_fermion_exact_sum_on_device(fixtures_dir, name, t2, e_ref):
    """reference known answers (test_exact_summation_evaluator.cpp:353-470, all six states): exact summation over the six
    half-filling configurations with amplitudes and hop ratios from the device (f64); the diagonal hop (t2) through
    fresh amplitudes with the Jordan-Wigner string of the row-major order."""
    from peps_amd import capi, fermion
    st = fermion.FermionState.load(os.path.join(fixtures_dir, "spinless_fermion_tps_t2_" + name))
    cfgs = _half_filling_configs()
    ctx = _ctx(st, 8, capi.F64, len(cfgs))
    amp = fermion.evaluate_amplitude(ctx, st, cfgs)
    e_loc, _ = fermion.spinless_fermion_energy(ctx, st, cfgs, 1.0, 0.0, t2)
    w = amp ** 2
    assert abs(np.sum(w * e_loc) / np.sum(w) - e_ref) < 1e-9
    if t2 != 0.0:                       # registry form: the bonds add up to the energy, per configuration
        obs, _ = fermion.spinless_fermion_observables(ctx, st, cfgs, 1.0, 0.0, t2)
        tot = sum(obs[k].sum(axis=1) for k in ("bond_energy_h", "bond_energy_v", "bond_energy_dr", "bond_energy_ur"))
        assert np.max(np.abs(tot - obs["energy"][:, 0])) < 1e-10 and np.max(np.abs(obs["energy"][:, 0] - e_loc)) < 1e-10
        assert np.count_nonzero(obs["bond_energy_dr"]) + np.count_nonzero(obs["bond_energy_ur"]) > 0
    # the C++ host layer's SquareSpinlessFermion(t, t2, V) (round 4: the diagonal hop from fresh batched amplitudes there too)
    from peps_amd import hostapi
    amps2, en2, _ = hostapi.fermion_energy(st, cfgs, 8, 1.0, 0.0, 1, t2=t2)
    assert np.max(np.abs(amps2 - amp)) < 1e-12 * np.max(np.abs(amp)) and np.max(np.abs(en2 - e_loc)) < 1e-9
    assert abs(np.sum(amps2 ** 2 * en2) / np.sum(amps2 ** 2) - e_ref) < 1e-9


@pytest.mark.parametrize("dt,tol", [("f64", 1e-10), ("f32", 2e-5)])
def test_exact_sum_measurer_reference_registry_on_device(fixtures_dir, dt, tol):
    """ExactSumMeasurerMPI known answers (tests/test_algorithm/test_exact_summation_measurer.cpp:205-240): energy,
    charge and the per-bond energies of the 2x2 simple-update state, amplitudes and hop ratios from the device;
    serial == 4-rank decomposition (:276-290)."""
    from peps_amd import capi, fermion
    MEASURER_GOLDEN = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "k4_exact_sum_measurer.json")))["observables"]
    st = fermion.FermionState.load(os.path.join(fixtures_dir, "spinless_fermion_tps_t2_0.000000_double_from_simple_update"))
    cfgs = _half_filling_configs()
    ctx = _ctx(st, 8, capi.F64 if dt == "f64" else capi.F32, len(cfgs))
    acc, w = fermion.exact_sum_measure(ctx, st, cfgs, 1.0, 0.0)
    assert set(acc) == set(MEASURER_GOLDEN)
    for key, want in MEASURER_GOLDEN.items():
        assert acc[key].shape == (len(want),), key
        assert np.max(np.abs(acc[key] / w - np.array(want))) < tol, key
    parts = [fermion.exact_sum_measure(ctx, st, cfgs, 1.0, 0.0, r, 4, batch=1) for r in range(4)]
    wp = sum(p[1] for p in parts)
    for key in acc:
        assert np.max(np.abs(sum(p[0][key] for p in parts) / wp - acc[key] / w)) < tol
    with pytest.raises(ValueError):
        fermion.exact_sum_measure(ctx, st, np.zeros((0, 2, 2), dtype=np.int64), 1.0)


@pytest.mark.parametrize("name", ["2.100000_double_from_simple_update", "-2.500000_doublelowest"])
def test_k4_amplitudes_match_graded_contraction(fixtures_dir, name):
    """device amplitudes (row- and column-major mode order) == full graded contraction, t2 != 0 fixtures"""
    from peps_amd import capi, fermion
    d = os.path.join(fixtures_dir, "spinless_fermion_tps_t2_" + name)
    st = fermion.FermionState.load(d)
    gts = ofermion.load_fermion_sitps(d)
    cfgs = _half_filling_configs()
    ctx = _ctx(st, 8, capi.F64, len(cfgs))
    amp = fermion.evaluate_amplitude(ctx, st, cfgs)
    ctx.set_configs(st.ext_config(cfgs, fermion.COL))
    amp_col = st.sigma(cfgs) * ctx.evaluate_amplitude()
    for k, cfg in enumerate(cfgs):
        g = graded.graded_amplitude_exact(gts, cfg)
        assert abs(amp[k] - g) < 1e-10 * max(1.0, abs(g))
        assert abs(amp_col[k] - st.kappa(cfg) * g) < 1e-10 * max(1.0, abs(g))


@pytest.mark.parametrize("dt,tol", [("f64", 1e-9), ("f32", 2e-5)])
def test_random_even_state_amplitude_and_energy(dt, tol):
    """4x4, D=4, chi=16 synthetic fermionic state: amplitudes and t-V local energies of random configurations
    (any filling) against the oracle; psi identical along every row / column route up to the kappa sign."""
    from peps_amd import capi, fermion
    st = fermion.random_even_state(4, 4, 4, seed=3)
    _, fs = _oracle_view(st)
    rng = np.random.default_rng(5)
    cfgs = rng.integers(0, 2, size=(6, 4, 4))
    cfgs[(4 * 4 - cfgs.sum(axis=(1, 2))) % 2 == 1, 0, 0] ^= 1        # even particle number: non-vanishing amplitude
    tp = BMPSTruncateParams.SVD(16, 16, 0.0)
    ctx = _ctx(st, 16, capi.F32 if dt == "f32" else capi.F64, len(cfgs))
    amp = fermion.evaluate_amplitude(ctx, st, cfgs)
    e_loc, psis = fermion.spinless_fermion_energy(ctx, st, cfgs, 1.0, 0.7)
    model = ofermion.SquareSpinlessFermionOBC(1.0, 0.0, 0.7)
    for k, cfg in enumerate(cfgs):
        a = fs.amplitude(cfg, tp)
        assert abs(amp[k] / a - 1) < tol
        e, _ = model.CalEnergy(fs, cfg, tp)
        assert abs(e_loc[k] - e) < tol * 10 * max(1.0, abs(e))
        sig, kap = st.sigma(cfg), st.kappa(cfg)
        assert np.max(np.abs(sig * psis[:4, k] / a - 1)) < tol * 10          # row routes
        assert np.max(np.abs(sig * kap * psis[4:, k] / a - 1)) < tol * 10    # column routes (column-major mode order)


C5_TOL = {"f64": (1e-7, 1e-7), "f32": (2e-5, 2e-5)}      # (amplitude, local energy), relative; measured: 1.2e-8 / 4.7e-9 and 7.2e-6 / 6.8e-6


@pytest.fixture(scope="module")
def c5_chain():
    """C5 state + EIGHT configurations as Monte-Carlo chains visit them (not the heaviest of a random pool):
    eight independent float64 device chains of the C++ host layer (NN exchange, std::mt19937 seeds 700..707), three sweeps from
    random half-filled starts; float64 oracle amplitude and t-V local energy of each."""
    from peps_amd import fermion, hostapi
    L, D, chi = 8, 6, 24
    st = fermion.random_even_state(L, L, D, seed=11)
    _, fs = _oracle_view(st)
    rng = np.random.default_rng(77)
    start = np.stack([rng.permutation(np.r_[np.zeros(32, dtype=int), np.ones(32, dtype=int)]).reshape(L, L) for _ in range(8)])
    cfgs, _, rates = hostapi.fermion_mc_sweeps(st, start, np.arange(8, dtype=np.uint64) + 700, chi, 3, 1)
    assert np.all(cfgs.sum(axis=(1, 2)) == 32) and np.all(rates > 0)
    tp = BMPSTruncateParams.SVD(chi, chi, 0.0)
    model = ofermion.SquareSpinlessFermionOBC(1.0, 0.0, 1.0)
    ref_a = np.array([fs.amplitude(c, tp) for c in cfgs])
    ref_e = np.array([model.CalEnergy(fs, c, tp)[0] for c in cfgs])
    return st, cfgs, ref_a, ref_e


@pytest.mark.parametrize("dt", ["f64", "f32"])
def test_c5_spinless_tV_8x8_d6_chi24(dt, c5_chain):
    """BASELINE config C5: 8x8 spinless-fermion t-V, Z2-graded tensors, D=6, chi=24: amplitude and local energy of EVERY one of
    eight chain-visited configurations against the f64 oracle at the full size.  The float64 device mode is the parity-grade path
    for fermions and holds north_star's 1e-6 on the energy with a decade to spare; fermionic amplitudes are sums with alternating
    signs, so the f32 mode is stated at 2e-5 on both (tolerances C5_TOL, printed with the measured values: the f32 energy is
    1e-7 in the median and 7e-6 on the worst of the eight configurations)."""
    from peps_amd import capi, fermion
    st, cfgs, ref_a, ref_e = c5_chain
    tol_amp, tol_e = C5_TOL[dt]
    ctx = _ctx(st, 24, capi.F32 if dt == "f32" else capi.F64, len(cfgs))
    amp = fermion.evaluate_amplitude(ctx, st, cfgs)
    e_loc, psis = fermion.spinless_fermion_energy(ctx, st, cfgs, 1.0, 1.0)
    assert np.all(ctx.walker_flags() == 0)
    ra, re = np.abs(amp / ref_a - 1), np.abs(e_loc / ref_e - 1)
    print("C5 %s: amplitude rel err max %.2e median %.2e; energy rel err max %.2e median %.2e (n = %d chain-visited configurations)"
          % (dt, ra.max(), np.median(ra), re.max(), np.median(re), len(cfgs)))
    assert ra.max() < tol_amp, ra
    assert re.max() < tol_e, re
    # size-independent property on every walker: all 2 L routes give the same |psi| up to the chi-truncation
    assert np.max(np.abs(np.abs(psis) / np.abs(amp)[None, :] - 1)) < (1e-3 if dt == "f32" else 1e-4)


@pytest.mark.parametrize("dt,tol", [(1, 1e-9), (0, 5e-5)])
def test_cpp_host_layer_fermion_energy(dt, tol):
    """C++ host layer (FermionDecoration + TPSWaveFunctionComponent + SquareSpinlessFermion through the CRTP
    energy solver of qlpeps_gpu.h): amplitudes and t-V local energies == Python product path == oracle."""
    from peps_amd import capi, fermion, hostapi
    st = fermion.random_even_state(4, 5, 4, seed=21) if False else fermion.random_even_state(4, 4, 4, seed=21)
    _, fs = _oracle_view(st)
    rng = np.random.default_rng(9)
    cfgs = rng.integers(0, 2, size=(5, 4, 4))
    cfgs[(16 - cfgs.sum(axis=(1, 2))) % 2 == 1, 0, 0] ^= 1
    amps, en, psi = hostapi.fermion_energy(st, cfgs, 16, 1.0, 0.5, dt)
    _, en_t2, _ = hostapi.fermion_energy(st, cfgs, 16, 1.0, 0.5, dt, t2=0.7)
    tp = BMPSTruncateParams.SVD(16, 16, 0.0)
    model, model_t2 = ofermion.SquareSpinlessFermionOBC(1.0, 0.0, 0.5), ofermion.SquareSpinlessFermionOBC(1.0, 0.7, 0.5)
    for k, cfg in enumerate(cfgs):
        a = fs.amplitude(cfg, tp)
        assert abs(amps[k] / a - 1) < tol
        e, _ = model.CalEnergy(fs, cfg, tp)
        assert abs(en[k] - e) < tol * 10 * max(1.0, abs(e))
        e2, _ = model_t2.CalEnergy(fs, cfg, tp)          # 4x4 with the diagonal hop: 18 plaquette diagonals, strings of up to 4 sites
        assert abs(en_t2[k] - e2) < tol * 10 * max(1.0, abs(e2)) and abs(e2 - e) > 1e-3
    assert psi.shape[0] == 8


def test_cpp_host_layer_fermion_mc_sweep():
    """MCUpdateSquareNNExchangeOBC on a fermionic state (f64): particle number conserved, moves accepted, and
    the carried amplitude equals a fresh evaluation of the final configuration in magnitude."""
    from peps_amd import capi, fermion, hostapi
    st = fermion.random_even_state(4, 4, 4, seed=22)
    rng = np.random.default_rng(3)
    cfgs = np.stack([rng.permutation(np.r_[np.zeros(8, dtype=int), np.ones(8, dtype=int)]).reshape(4, 4) for _ in range(6)])
    seeds = np.arange(6, dtype=np.uint64) + 40
    out_cfg, amps, rates = hostapi.fermion_mc_sweeps(st, cfgs, seeds, 16, 2, 1)
    assert np.all(out_cfg.sum(axis=(1, 2)) == 8)
    assert np.all(rates > 0) and np.all(rates < 1)
    assert not np.array_equal(out_cfg, cfgs)
    ctx = _ctx(st, 16, capi.F64, len(cfgs))
    fresh = fermion.evaluate_amplitude(ctx, st, out_cfg)
    assert np.max(np.abs(amps / fresh - 1)) < 1e-7           # SIGNED: the stored amplitude keeps the graded sign through every accepted move
    # same seeds, same chain
    out2, amps2, _ = hostapi.fermion_mc_sweeps(st, cfgs, seeds, 16, 2, 1)
    assert np.array_equal(out2, out_cfg)


def test_fermion_exact_sum_gradient_vs_finite_differences(fixtures_dir):
    """Energy gradient of a fermionic state (ExactSumEnergyEvaluator over the extended components on the device,
    folded back to the stored components): grad = <E_loc O*> - E <O*> (exact_summation_energy_evaluator.h:286-295),
    i.e. half the derivative of E for real parameters -- checked against central differences of the ORACLE energy."""
    from peps_amd import fermion, hostapi
    d = os.path.join(fixtures_dir, "spinless_fermion_tps_t2_0.000000_double_from_simple_update")
    st = fermion.FermionState.load(d)
    cfgs = _half_filling_configs()
    t, V = 1.0, 0.3
    e, grad = hostapi.fermion_exact_sum(st, cfgs, 8, t, V, batch=6, dtype=1)
    gts = ofermion.load_fermion_sitps(d)
    tp = BMPSTruncateParams.SVD(8, 8, 0.0)
    model = ofermion.SquareSpinlessFermionOBC(t, 0.0, V)
    e0 = ofermion.exact_sum_energy(ofermion.FermionSITPS(gts), list(cfgs), tp, model)
    assert abs(e - e0) < 1e-10
    rng = np.random.default_rng(2)
    checked = 0
    for r, c, s in [(0, 0, 0), (0, 1, 1), (1, 0, 0), (1, 1, 1), (0, 0, 1), (1, 1, 0)]:
        a = gts[r][c][s].arr
        nzs = np.argwhere(np.abs(a) > 1e-12)
        ix = tuple(nzs[rng.integers(len(nzs))])
        h = 1e-5
        vals = []
        for sgn in (+1, -1):
            a[ix] += sgn * h
            vals.append(ofermion.exact_sum_energy(ofermion.FermionSITPS(gts), list(cfgs), tp, model))
            a[ix] -= sgn * h
        fd = (vals[0] - vals[1]) / (2 * h)
        g = grad[(r, c, s) + ix[:4]]
        assert abs(2 * g - fd) < 1e-6 * max(1.0, abs(fd)), (r, c, s, ix, g, fd)
        checked += 1
    assert checked == 6
    # parity-forbidden entries carry no gradient
    assert np.all(grad[0, 0, 0][np.abs(st.extended_flat()[0, 0, 0]) == 0] == 0)


@pytest.mark.parametrize("name,e_ref", [("tj_model_tps_doublelowest", -2.9431635706137875),
                                        ("tj_model_tps_double_from_simple_update", -2.78008187385)])
def test_k4_tj_model_exact_sum_on_device(fixtures_dir, name, e_ref):
    """2x2 t-J known answers (test_exact_summation_evaluator.cpp:795-990; three physical states per site, two odd)
    with amplitudes and exchange / hop ratios from the device, Python path and C++ host layer."""
    from peps_amd import capi, fermion, hostapi
    st = fermion.FermionState.load(os.path.join(fixtures_dir, name))
    cfgs = np.array([np.array(p).reshape(2, 2) for p in sorted(set(itertools.permutations([2, 2, 0, 1])))])
    ctx = _ctx(st, 4, capi.F64, len(cfgs))
    amp = fermion.evaluate_amplitude(ctx, st, cfgs)
    e_loc, _ = fermion.tj_energy(ctx, st, cfgs, 1.0, 0.3, 0.075, 0.0)
    w = amp ** 2
    assert abs(np.sum(w * e_loc) / np.sum(w) - e_ref) < 1e-9
    amps2, en2, _ = hostapi.fermion_energy(st, cfgs, 4, 1.0, 0.075, 1, "tj", 0.3, 0.0)
    assert np.max(np.abs(amps2 - amp)) < 1e-12 * np.max(np.abs(amp)) and np.max(np.abs(en2 - e_loc)) < 1e-9


def test_k7_tj_network_route_consistency_on_device(fixtures_dir):
    """K7 on the device: the reference's 20 x 24 projected t-J network (test_bmps_contractor.cpp:688-865, tests/k7_tj.py),
    BMPSTruncateParams(16, 50, 1e-15), all 21 routes of Contract2DTNUsingBMPSContractor -- row passes on the row-major
    decorated components, column passes on the column-major ones -- agree in magnitude to 1e-7 (:855-858) and with the oracle."""
    import k1_routes
    import k7_tj
    from oracle.contractor import BMPSContractor
    from peps_amd import capi, fermion
    from test_oracle_k7 import decorated_tn
    st, relabel = k7_tj.build_state(fixtures_dir)
    cfg = relabel(k7_tj.CONFIG)
    flat = st.extended_flat()
    hor, ver = k1_routes.routes_by_pass(st.rows)
    amps = []
    for order, ops in ((fermion.ROW, hor), (fermion.COL, ver)):
        ctx = capi.Context(st.rows, st.cols, st.D, fermion.NVAR * st.d, k7_tj.DB_MAX, dtype=capi.F64, max_walkers=1,
                           chi_min=k7_tj.DB_MIN, trunc_err=1e-15)
        ctx.state_upload(flat)
        ctx.set_configs(st.ext_config(cfg, order)[None])
        amps += [float(a[0]) for a in k1_routes._walk(ops, ctx, None, device=True)]
        ctx.close()
    mag = np.abs(np.array(amps))
    # 3e-7, not the oracle's 1e-7: the device builds the carry from a float64 Gram matrix, which resolves it down to
    # sqrt(n eps64) ~ 2e-7 of its largest direction (an explicit QR, as in the reference and the oracle, goes to eps64);
    # with D_max = 50 on this state the routes then differ by 1.5e-7 instead of 0.9e-7
    assert len(amps) == k1_routes.N_AMPS and np.max(np.abs(mag / mag[0] - 1)) < 3e-7, mag / mag[0] - 1
    tn = decorated_tn(st, flat, st.ext_config(cfg, fermion.ROW))
    c = BMPSContractor(st.rows, st.cols)
    c.Init(tn)
    c.SetTruncateParams(BMPSTruncateParams.SVD(k7_tj.DB_MIN, k7_tj.DB_MAX, 1e-15))
    ref = float(k1_routes._walk(hor[:4], c, tn, device=False)[0])
    assert abs(amps[0] / ref - 1) < 3e-7


def test_reference_fermion_gradient_signatures_on_device(fixtures_dir):
    """The reference's golden signatures of the exact-summation GRADIENT of a fermionic state (NormSquare and
    WeightedProbeInnerProduct, test_exact_summation_evaluator.cpp:50-71, :379-400: spinless fermions, 'lowest' state at
    t2 = 0, SVD(8, 8, 1e-16), six half-filling configurations): 2.184991439005157e-17 and 7.407222090395872e-18, asserted there
    to an absolute 1e-8.  Both are sums of squares: blind to the element-wise sign convention of CalGTenForFermionicTensors
    (which stays unpinned), but they pin the MAGNITUDE of every gradient component of the device path (holes resident in HBM,
    pepsgpu_grad_accumulate_states, folded back to the stored components)."""
    from peps_amd import fermion, hostapi
    st = fermion.FermionState.load(os.path.join(fixtures_dir, "spinless_fermion_tps_t2_0.000000_doublelowest"))
    cfgs = _half_filling_configs()
    e, grad = hostapi.fermion_exact_sum(st, cfgs, 8, 1.0, 0.0, batch=6, dtype=1)
    assert abs(e - (-2.0)) < 1e-7
    ns = wp = 0.0
    for r in range(2):
        for c in range(2):
            for i in range(grad.shape[2]):
                n2 = float(np.sum(grad[r, c, i] ** 2))
                ns += n2
                wp += 0.012 * ((r + 1) * 11 + (c + 1) * 5 + (i + 1) * 2) * n2
    print("fermion gradient signatures: NormSquare %.15e (ref 2.184991439005157e-17)  probe %.15e (ref 7.407222090395872e-18)" % (ns, wp))
    assert abs(ns / 2.184991439005157e-17 - 1) < 1e-6 and abs(wp / 7.407222090395872e-18 - 1) < 1e-6    # (the reference asserts an absolute 1e-8; measured here: 1e-9 relative)
    assert ns < 1e-12                                                                               # ... and the scale it implies


def test_k9_reference_tj_measurer_regression_energy_on_the_device(fixtures_dir):
    """K9 through the HIP path (f64 mode): the REFERENCE's deterministic regression value of MCPEPSMeasurer on a fermionic state --
    tests/test_model_solvers/test_tJ_model_solver.cpp:72-75, 233-275: -14.74320489110316 +- 1e-8 -- 6x6 t-J, two holes, fU1 tensors,
    D = 8, configuration0, MCUpdateSquareNNExchange(42), 10 warm-up sweeps, 10 samples, SVD(8, 16, 1e-15), SquaretJNNModel(1, 0.3, 0).
    pepshost_fermion_measure_energy: ONE std::mt19937 stream over warm-up, rebuild and samples.  (Staged at the end of round 4, first
    run -- green -- in round 5; the oracle reproduces the value to 2e-15, tests/test_oracle_fermion.py.)"""
    from peps_amd import fermion, hostapi
    d = os.path.join(fixtures_dir, "tps_tJ_6x6Hole2_J0.3_D8_fU1")
    st = fermion.FermionState.load(d)
    cfg = np.loadtxt(os.path.join(d, "configuration0"), dtype=int).reshape(1, 6, 6)
    hostapi.set_truncate_params(8, 1e-15, 0)                       # BMPSTruncateParams::SVD(8, 16, 1e-15)
    try:
        en, _, _ = hostapi.fermion_measure_energy(st, cfg, [42], 16, 10, 10, 1, "tj", t=1.0, J=0.3, V=0.0, mu=0.0, dtype=1)
    finally:
        hostapi.set_truncate_params()
    e = float(np.mean(en[:, 0]))
    print("K9 on the device: energy %.14f (reference -14.74320489110316, diff %.1e)" % (e, abs(e + 14.74320489110316)))
    assert abs(e - (-14.74320489110316)) < 1e-8


@pytest.mark.parametrize("shape,D,chi", [((4, 4), 4, 16), ((3, 5), 3, 9)])
def test_nnn_hop_with_twisted_environments_equals_fresh_amplitudes_and_oracle(shape, D, chi):
    """Round 5: the diagonal (t2) hop of a fermionic state with the environments of the row pass -- the
    reference's flow (square_spinless_fermion.h:161-213, square_nnn_energy_solver.h:203-265) -- instead of one fresh contraction
    per hop.  In the decorated form the hop flips the variant of a whole stretch of two rows, so the hopped amplitude is a
    plaquette replacement against parity-twisted BTen2 environments (peps_amd/fermion.py::nnn_hop_energy_local).  Checked per
    diagonal bond against the fresh-amplitude form of rounds 2-4 and, on the total energy, against the graded oracle (f64)."""
    from peps_amd import capi, fermion
    rows, cols = shape
    st = fermion.random_even_state(rows, cols, D, seed=31)
    _, fs = _oracle_view(st)
    rng = np.random.default_rng(17)
    cfgs = rng.integers(0, 2, size=(6, rows, cols))
    cfgs[(rows * cols - cfgs.sum(axis=(1, 2))) % 2 == 1, 0, 0] ^= 1
    ctx = _ctx(st, chi, capi.F64, len(cfgs))
    b_local, b_fresh = {}, {}
    e_local = fermion.nnn_hop_energy_local(ctx, st, cfgs, 0.7, b_local)
    e_fresh = fermion.nnn_hop_energy(ctx, st, cfgs, 0.7, b_fresh)
    assert np.count_nonzero(b_fresh["dr"]) + np.count_nonzero(b_fresh["ur"]) > 10
    for key in ("dr", "ur"):
        assert np.max(np.abs(b_local[key] - b_fresh[key])) < 1e-9 * max(1.0, np.max(np.abs(b_fresh[key]))), key
    assert np.max(np.abs(e_local - e_fresh)) < 1e-9 * max(1.0, np.max(np.abs(e_fresh)))
    tp = BMPSTruncateParams.SVD(chi, chi, 0.0)
    m0, m2 = ofermion.SquareSpinlessFermionOBC(1.0, 0.0, 0.5), ofermion.SquareSpinlessFermionOBC(1.0, 0.7, 0.5)
    e_tot, _ = fermion.spinless_fermion_energy(ctx, st, cfgs, 1.0, 0.5, 0.7)          # nnn = "local" is the default
    for k, cfg in enumerate(cfgs):
        want = m2.CalEnergy(fs, cfg, tp)[0]
        assert abs(e_tot[k] - want) < 1e-8 * max(1.0, abs(want)), (k, e_tot[k], want)
        assert abs((want - m0.CalEnergy(fs, cfg, tp)[0]) - e_local[k]) < 1e-8 * max(1.0, abs(want))
    assert np.all(ctx.walker_flags() == 0)
