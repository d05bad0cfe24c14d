"""Fermionic (fZ2-graded) path of the oracle, pinned on the reference's own known answers
(tests/test_algorithm/test_exact_summation_evaluator.cpp:130-150,268-470): 2x2 spinless fermions at half
filling, t2 in {2.1, 0, -2.5}, 'lowest' states (exact free-fermion energies) and simple-update states
(golden energies :432-436)."""
import itertools
import json
import os

import numpy as np
import pytest

from oracle import fermion, graded
from oracle.bmps import BMPSTruncateParams
from oracle.graded import GT

CASES = [(2.1, "2.100000", -4.2, -4.1879072654), (0.0, "0.000000", -2.0, -1.98218053854),
         (-2.5, "-2.500000", -5.0, -4.98966397657)]


def _half_filling_configs():
    return [np.array(p).reshape(2, 2) for p in sorted(set(itertools.permutations([0, 0, 1, 1])))]


def _dense_energy(gts, t, t2):
    """<Psi|H|Psi>/<Psi|Psi> in the Fock basis ordered like the parity legs (row-major), amplitudes from the
    full graded contraction -- no boundary MPS, no decoration."""
    sites = [(0, 0), (0, 1), (1, 0), (1, 1)]
    cfgs = []
    for occ in itertools.combinations(range(4), 2):
        c = np.ones((2, 2), dtype=int)
        for o in occ:
            c[sites[o]] = 0
        cfgs.append((occ, c))
    psi = np.array([graded.graded_amplitude_exact(gts, c) for _, c in cfgs])
    idx = {occ: i for i, (occ, _) in enumerate(cfgs)}
    H = np.zeros((6, 6))
    for i, (occ, _) in enumerate(cfgs):
        for a, b, amp in [(0, 1, t), (2, 3, t), (0, 2, t), (1, 3, t), (0, 3, t2), (1, 2, t2)]:
            for src, dst in ((a, b), (b, a)):
                if src in occ and dst not in occ:
                    lst = list(occ)
                    pos = lst.index(src)
                    s = (-1) ** pos
                    lst.pop(pos)
                    s *= (-1) ** sum(1 for x in lst if x < dst)
                    H[idx[tuple(sorted(lst + [dst]))], i] += -amp * s
    return psi @ H @ psi / (psi @ psi)


@pytest.mark.parametrize("t2,name,e_lowest,e_su", CASES)
def test_graded_conventions_reproduce_reference_energies(fixtures_dir, t2, name, e_lowest, e_su):
    for suffix, ref, tol in (("lowest", e_lowest, 1e-9), ("_from_simple_update", e_su, 1e-9)):
        gts = fermion.load_fermion_sitps(os.path.join(fixtures_dir, "spinless_fermion_tps_t2_%s_double%s" % (name, suffix)))
        assert abs(_dense_energy(gts, 1.0, t2) - ref) < tol


@pytest.mark.parametrize("t2,name,e_lowest,e_su", CASES)
def test_decorated_bmps_path_reproduces_reference_energies(fixtures_dir, t2, name, e_lowest, e_su):
    """the product formulation: sign-decorated dense tensors through the bosonic BMPS contractor, NN hops from
    ReplaceNNSiteTrace in the row / column pass, SVD(8, 8) as the reference test"""
    tp = BMPSTruncateParams.SVD(8, 8, 0.0)
    for suffix, ref in (("lowest", e_lowest), ("_from_simple_update", e_su)):
        gts = fermion.load_fermion_sitps(os.path.join(fixtures_dir, "spinless_fermion_tps_t2_%s_double%s" % (name, suffix)))
        fs = fermion.FermionSITPS(gts)
        e = fermion.exact_sum_energy(fs, _half_filling_configs(), tp, fermion.SquareSpinlessFermionOBC(1.0, t2, 0.0))
        assert abs(e - ref) < 1e-9


def _random_even_network(Ly, Lx, D, rng):
    hp = [[np.array([0]) if c in (0, Lx) else np.r_[0, rng.integers(0, 2, D - 2), 1] for c in range(Lx + 1)] for r in range(Ly)]
    vp = [[np.array([0]) if r in (0, Ly) else np.r_[0, rng.integers(0, 2, D - 2), 1] for c in range(Lx)] for r in range(Ly + 1)]
    sit = [[None] * Lx for _ in range(Ly)]
    for r in range(Ly):
        for c in range(Lx):
            comp = []
            for n in (1, 0):          # state 0 occupied (odd), state 1 empty (even)
                par = [hp[r][c], vp[r + 1][c], hp[r][c + 1], vp[r][c], np.array([n])]
                tot = (par[0][:, None, None, None, None] + par[1][None, :, None, None, None] + par[2][None, None, :, None, None]
                       + par[3][None, None, None, :, None] + n) % 2
                comp.append(GT(rng.standard_normal(tot.shape) * (tot == 0), par, [-1, 1, 1, -1, -1]))
            sit[r][c] = comp
    return sit


@pytest.mark.parametrize("Ly,Lx,D", [(2, 3, 3), (3, 3, 3), (3, 4, 2)])
def test_decorated_dense_contraction_equals_graded_contraction(Ly, Lx, D):
    """<S|Psi> of the graded network (oracle/graded.py) == sigma(N_f) x ordinary contraction of the decorated
    tensors, for EVERY configuration, parity legs in row-major and in column-major order."""
    rng = np.random.default_rng(100 * Ly + 10 * Lx + D)
    gts = _random_even_network(Ly, Lx, D, rng)
    fs = fermion.FermionSITPS(gts)
    tp = BMPSTruncateParams.SVD(64, 64, 0.0)          # exact at these sizes
    n_nonzero = 0
    for bits in itertools.product((0, 1), repeat=Ly * Lx):
        cfg = np.array(bits).reshape(Ly, Lx)
        g = graded.graded_amplitude_exact(gts, cfg)
        if sum(1 - b for b in bits) % 2 == 1:
            assert abs(g) < 1e-12                      # odd total parity: the network vanishes
            continue
        n_nonzero += 1
        scale = max(1.0, abs(g))
        assert abs(fs.amplitude(cfg, tp, fermion.ROW) - g) < 1e-9 * scale
        assert abs(fs.amplitude(cfg, tp, fermion.COL) - fs.kappa(cfg) * g) < 1e-9 * scale
    assert n_nonzero == 2 ** (Ly * Lx - 1)


@pytest.mark.parametrize("name,e_ref", [("tj_model_tps_doublelowest", -2.9431635706137875),
                                        ("tj_model_tps_double_from_simple_update", -2.78008187385)])
def test_tj_model_reference_energies(fixtures_dir, name, e_ref):
    """2x2 t-J model (t = 1, J = 0.3, V = J/4, mu = 0; one up, one down, two holes): known answers of
    test_exact_summation_evaluator.cpp:795-990 through the decorated BMPS path, SVD(4, 4)."""
    gts = fermion.load_fermion_sitps(os.path.join(fixtures_dir, name))
    fs = fermion.FermionSITPS(gts)
    assert fs.d == 3 and fs.nf == [1, 1, 0]
    cfgs = [np.array(p).reshape(2, 2) for p in sorted(set(itertools.permutations([2, 2, 0, 1])))]
    e = fermion.exact_sum_energy(fs, cfgs, BMPSTruncateParams.SVD(4, 4, 0.0), fermion.SquaretJVModelOBC(1.0, 0.0, 0.3, 0.075, 0.0))
    assert abs(e - e_ref) < 1e-10


# ExactSumMeasurerMPI known answers (tests/test_algorithm/test_exact_summation_measurer.cpp:205-240, real tensors):
# 2x2 spinless fermions, t = 1, t2 = V = 0, the simple-update state, SVD(8, 8, 1e-16), half filling
MEASURER_GOLDEN = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "k4_exact_sum_measurer.json")))["observables"]


def test_exact_sum_measurer_reproduces_reference_registry(fixtures_dir):
    tp = BMPSTruncateParams.SVD(8, 8, 1e-16)
    gts = fermion.load_fermion_sitps(os.path.join(fixtures_dir, "spinless_fermion_tps_t2_0.000000_double_from_simple_update"))
    fs = fermion.FermionSITPS(gts)
    model = fermion.SquareSpinlessFermionOBC(1.0, 0.0, 0.0)
    cfgs = _half_filling_configs()
    obs = fermion.exact_sum_measure(fs, cfgs, tp, model)
    assert set(obs) == set(MEASURER_GOLDEN)
    for key, want in MEASURER_GOLDEN.items():
        assert obs[key].shape == (len(want),), key
        assert np.max(np.abs(obs[key] - np.array(want))) < 1e-10, key
    # the reference test's own structural checks (:221-229): energy = sum of bond energies, total charge = 2
    assert abs(obs["energy"][0] - sum(obs[k].sum() for k in ("bond_energy_h", "bond_energy_v", "bond_energy_dr", "bond_energy_ur"))) < 1e-10
    assert abs(obs["charge"].sum() - 2.0) < 1e-10
    # rank decomposition (:276-290: serial == 4 ranks)
    parts = [fermion.exact_sum_measure(fs, cfgs, tp, model, r, 4) for r in range(4)]
    w = sum(p[1] for p in parts)
    for key in obs:
        assert np.max(np.abs(sum(p[0][key] for p in parts if key in p[0]) / w - obs[key])) < 1e-13
    with pytest.raises(RuntimeError):
        fermion.exact_sum_measure(fs, [], tp, model)


def test_k9_reference_tj_measurer_regression_energy(fixtures_dir):
    """K9 -- the reference's deterministic regression value of MCPEPSMeasurer on a FERMIONIC state
    (tests/test_model_solvers/test_tJ_model_solver.cpp:72-75, :233-275): 6x6 t-J state with two holes (fU1-symmetric tensors, D = 8,
    J / t = 0.3), start configuration `configuration0`, MCUpdateSquareNNExchange(42), 10 warm-up sweeps, NormalizeStateOrder1, 10 samples one
    sweep apart, BMPSTruncateParams::SVD(8, 16, 1e-15), SquaretJNNModel(t = 1, J = 0.3, mu = 0): energy -14.74320489110316, tolerance 1e-8.
    The oracle reproduces it to 1e-14: the decoding of the U(1)-block-sparse fermionic tensors, the decorated-dense form of the graded
    network at this size, the fermionic Monte-Carlo chain deviate for deviate over twenty sweeps of a three-state exchange updater, the
    truncated boundary MPS of a fermionic network, and every sign of the t-J local energy (hole hops along rows and columns, spin exchange)."""
    import os
    from oracle.graded import load_qlten_z2
    d = os.path.join(fixtures_dir, "tps_tJ_6x6Hole2_J0.3_D8_fU1")
    L = 6
    gts = [[[load_qlten_z2(os.path.join(d, "tps_ten%d_%d_%d.qlten" % (r, c, s))) for s in range(3)] for c in range(L)] for r in range(L)]
    F = fermion
    fs = F.FermionSITPS(gts)
    assert fs.nf == [1, 1, 0]                                        # up, down: odd; empty: even (tj_single_site_state.h:19-23)
    cfg = np.loadtxt(os.path.join(d, "configuration0"), dtype=int).reshape(L, L)
    assert sorted(np.bincount(cfg.ravel(), minlength=3).tolist()) == [2, 17, 17]
    tp = BMPSTruncateParams.SVD(8, 16, 1e-15)
    amp = abs(fs.component(cfg, F.ROW, tp).amplitude)                # TPSWaveFunctionComponent constructor
    upd = F.MCUpdateSquareNNExchangeOBC(42)
    for _ in range(10):                                              # MonteCarloEngine::WarmUp
        _, amp = upd(fs, cfg, amp, tp)
    amp = abs(fs.component(cfg, F.ROW, tp).amplitude)                # NormalizeStateOrder1 rebuilds the component (the scale drops out)
    model = F.SquaretJVModelOBC(1.0, 0.0, 0.3, 0.0, 0.0)             # SquaretJNNModel(t, J, mu) = mixin(t, 0, J, 0, mu) (square_tJ_model.h:619-623)
    es = []
    for _ in range(10):                                              # MCPEPSMeasurer::Measure_
        _, amp = upd(fs, cfg, amp, tp)
        es.append(model.CalEnergy(fs, cfg, tp)[0])
    assert abs(np.mean(es) - (-14.74320489110316)) < 1e-8            # EXPECTED_ENERGY, ENERGY_TOLERANCE (:74-75)
    assert abs(np.mean(es) - (-14.74320489110316)) < 1e-12


def test_reference_tj_exact_sum_measurer_registry(fixtures_dir):
    """ExactSumMeasurerMPI with SquaretJVModel on the reference's 2x2 t-J states (tests/test_algorithm/test_exact_summation_measurer.cpp:
    652-795): the whole registry of the simple-update state -- energy, spin_z, charge, the four bond-energy maps -- at the reference's
    1e-10, total charge 2, energy == sum of the bond energies, and the 'lowest' state within the tolerances stated there."""
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "k4_tj_exact_sum_measurer.json")))
    cfgs = [np.array(p).reshape(2, 2) for p in sorted(set(itertools.permutations([0, 1, 2, 2])))]      # :683-687
    tp = BMPSTruncateParams.SVD(4, 4, 0.0)
    model = fermion.SquaretJVModelOBC(1.0, 0.0, 0.3, 0.075, 0.0)
    fs = fermion.FermionSITPS(fermion.load_fermion_sitps(os.path.join(fixtures_dir, "tj_model_tps_double_from_simple_update")))
    obs = fermion.exact_sum_measure(fs, cfgs, tp, model)
    assert set(obs) == set(gold["observables"])
    assert abs(np.sum(obs["charge"]) - 2.0) < 1e-10
    assert abs(np.sum(obs["energy"]) - sum(np.sum(obs[k]) for k in obs if k.startswith("bond_energy"))) < 1e-10
    for key, want in gold["observables"].items():
        assert np.max(np.abs(np.asarray(obs[key]) - np.asarray(want))) < 1e-10, key
    fs = fermion.FermionSITPS(fermion.load_fermion_sitps(os.path.join(fixtures_dir, "tj_model_tps_doublelowest")))
    obs = fermion.exact_sum_measure(fs, cfgs, tp, model)
    tol = {"energy": 6e-8, "spin_z": 5e-4, "charge": 5e-4}
    for key, want in gold["lowest"].items():
        assert np.max(np.abs(np.asarray(obs[key]) - np.asarray(want))) < tol.get(key, 1e-5), key


def test_reference_spinless_fermion_measurer_t2_sweep_and_lowest_state_observables(fixtures_dir):
    """SpinlessFermionMeasurerTest.SimpleUpdateStateT2Sweep / LowestStateT2SweepEnergy / LowestStateObservables
    (tests/test_algorithm/test_exact_summation_measurer.cpp:292-407): for t2 in {2.1, 0, -2.5} and both states the registry has the six
    keys, energy == sum of the four bond maps (1e-8) and total charge 2; the t2 = 2.1 'lowest' state reproduces the QuSpin ED
    observables -- energy -4.2, uniform charge 1/2, vanishing nearest-neighbour bond energies, BOTH diagonal bond energies -2.1 (the
    sign of every next-nearest-neighbour hop)."""
    tp = BMPSTruncateParams.SVD(8, 8, 1e-16)
    cfgs = _half_filling_configs()
    keys = {"energy", "charge", "bond_energy_h", "bond_energy_v", "bond_energy_dr", "bond_energy_ur"}
    for t2 in (2.1, 0.0, -2.5):
        for kind in ("_double_from_simple_update", "_doublelowest"):
            fs = fermion.FermionSITPS(fermion.load_fermion_sitps(os.path.join(fixtures_dir, "spinless_fermion_tps_t2_%.6f%s" % (t2, kind))))
            obs = fermion.exact_sum_measure(fs, cfgs, tp, fermion.SquareSpinlessFermionOBC(1.0, t2, 0.0))
            assert set(obs) == keys
            assert abs(np.sum(obs["energy"]) - sum(np.sum(obs[k]) for k in keys if k.startswith("bond"))) < 1e-8
            assert abs(np.sum(obs["charge"]) - 2.0) < 1e-8
            if t2 == 2.1 and kind == "_doublelowest":
                assert abs(obs["energy"][0] + 4.2) < 6e-8
                assert np.max(np.abs(obs["charge"] - 0.5)) < 1e-5
                assert np.max(np.abs(obs["bond_energy_h"])) < 1e-5 and np.max(np.abs(obs["bond_energy_v"])) < 1e-5
                assert abs(obs["bond_energy_dr"][0] + 2.1) < 1e-5 and abs(obs["bond_energy_ur"][0] + 2.1) < 1e-5


@pytest.mark.parametrize("t2,name,e_lowest,e_su", CASES)
def test_complex_fixtures_reproduce_reference_energies(fixtures_dir, t2, name, e_lowest, e_su):
    """The QLTEN_Complex build of the same reference test (tests/CMakeLists.txt:369-383 compiles test_exact_summation_evaluator.cpp for both
    element types; `_complexlowest` / `_complex_from_simple_update` fixtures, :306-330): the same six energies from complex tensors, imaginary
    part of the energy zero (the reference asserts |Im| < 1e-10 on the measurer side, test_exact_summation_measurer.cpp:219-257)."""
    tp = BMPSTruncateParams.SVD(8, 8, 0.0)
    for suffix, ref in (("lowest", e_lowest), ("_from_simple_update", e_su)):
        gts = fermion.load_fermion_sitps(os.path.join(fixtures_dir, "spinless_fermion_tps_t2_%s_complex%s" % (name, suffix)), complex_data=True)
        fs = fermion.FermionSITPS(gts)
        e = fermion.exact_sum_energy(fs, _half_filling_configs(), tp, fermion.SquareSpinlessFermionOBC(1.0, t2, 0.0))
        assert abs(e - ref) < 1e-9 and abs(np.imag(e)) < 1e-10
    assert np.max(np.abs(gts[0][0][0].arr.imag)) > 0.1            # the simple-update fixture really is complex


def test_exact_sum_measurer_registry_on_the_complex_fixture(fixtures_dir):
    """test_exact_summation_measurer.cpp:243-257 (QLTEN_Complex branch): the complex simple-update state gives the real build's registry
    (the two lists of that test agree to 1e-15) with zero imaginary parts."""
    tp = BMPSTruncateParams.SVD(8, 8, 1e-16)
    gts = fermion.load_fermion_sitps(os.path.join(fixtures_dir, "spinless_fermion_tps_t2_0.000000_complex_from_simple_update"), complex_data=True)
    obs = fermion.exact_sum_measure(fermion.FermionSITPS(gts), _half_filling_configs(), tp, fermion.SquareSpinlessFermionOBC(1.0, 0.0, 0.0))
    assert set(obs) == set(MEASURER_GOLDEN)
    for key, want in MEASURER_GOLDEN.items():
        assert np.max(np.abs(obs[key] - np.array(want))) < 1e-10, key                 # kTol
        assert np.max(np.abs(np.imag(obs[key]))) < 1e-10, key                         # kImagTol


@pytest.mark.parametrize("complex_data,name", [(False, "spinless_fermion_tps_t2_2.100000_double_from_simple_update"),
                                               (True, "spinless_fermion_tps_t2_2.100000_complex_from_simple_update"),
                                               (True, "spinless_fermion_tps_t2_-2.500000_complexlowest")])
def test_product_decoration_equals_the_oracle_decoration(fixtures_dir, complex_data, name):
    """peps_amd/fermion.py (the product's loader and sign decoration: FermionState.load / extended_flat / ext_config / sigma / kappa; NumPy
    only, never imports the oracle) against oracle/fermion.py on the reference's fixtures, real and QLTEN_Complex: the 4 d extended
    components the device is given are, element for element, the oracle's decorated tensors, and the extended configurations and the two
    signs agree on all 16 configurations.  (CPU: what reaches the GPU is checked before it gets there.)"""
    from peps_amd import fermion as pfermion
    d = os.path.join(fixtures_dir, name)
    st = pfermion.FermionState.load(d, complex_data=complex_data)
    fs = fermion.FermionSITPS(fermion.load_fermion_sitps(d, complex_data=complex_data))
    assert st.is_complex == complex_data and list(st.nf) == list(fs.nf)
    flat = st.extended_flat()
    assert flat.dtype == (np.complex128 if complex_data else np.float64) and flat.shape[2] == 4 * st.d
    for r in range(st.rows):
        for c in range(st.cols):
            for s in range(4 * st.d):
                want = fs.ext[r][c][s]
                got = flat[(r, c, s) + tuple(slice(0, k) for k in want.shape)]
                assert np.array_equal(got, want), (r, c, s)
                pad = flat[r, c, s].copy()
                pad[tuple(slice(0, k) for k in want.shape)] = 0
                assert not np.any(pad)                                   # zero padding beyond the bond dimensions of the site
    cfgs = np.array(list(itertools.product([0, 1], repeat=4))).reshape(-1, 2, 2)
    for order in (fermion.ROW, fermion.COL):
        ext = st.ext_config(cfgs, order)
        for k, cfg in enumerate(cfgs):
            assert np.array_equal(ext[k], fs.ext_config(cfg, order))
    assert np.array_equal(st.sigma(cfgs), [fs.sigma(c) for c in cfgs]) and np.array_equal(st.kappa(cfgs), [fs.kappa(c) for c in cfgs])
