"""Measurement: registry observables of the C++ measurement solver and the MCPEPSMeasurer loop on the
device against the oracle restatement (square_nnn_model_measurement_solver.h, square_spin_onehalf_xxz_obc.h:22-60,
205-330, monte_carlo_peps_measurer_impl.h, psi_consistency.h)."""
import os

import numpy as np
import pytest

from oracle import qlten_io, vmc
from oracle.bmps import BMPSTruncateParams
from peps_amd import synthetic

pytestmark = pytest.mark.gpu
F32, F64 = 0, 1


def _host():
    from peps_amd import hostapi
    return hostapi


def _oracle_obs(s, cfgs, chi, model):
    tp = BMPSTruncateParams.SVD(chi, chi, 0.0)
    out = []
    for c in cfgs:
        comp = vmc.TPSWaveFunctionComponent(s, c, tp)
        ms = vmc.SquareNNNModelMeasurementSolver(model)
        out.append((ms.EvaluateObservables(s, comp), ms.last_psi_summary))
    return out


@pytest.mark.parametrize("dt,tol", [(F64, 1e-9), (F32, 5e-5)])
@pytest.mark.parametrize("model,params", [("xxz", (1.0, 0.8, 0.3)), ("j1j2", (1.0, 1.0, 0.5, 0.4, 0.0))])
def test_registry_observables_fixed_configs(model, params, dt, tol):
    """every registry key of EvaluateObservables on identical configurations: energy, spin_z, bond_energy_h/v(/dr/ur),
    SzSz_all2all, the S+S- / S-S+ channel along the middle row, and the psi summary of the sample (8x6 lattice: x0 = 1)"""
    host = _host()
    Ly = Lx = 6
    D, chi = 3, 9
    s = synthetic.make_sitps(Ly, D)
    cfgs = synthetic.make_configs(Ly, 5, "heisenberg")
    om = vmc.SquareSpinOneHalfXXZModelOBC(*params) if model == "xxz" else vmc.SquareSpinOneHalfJ1J2XXZModelOBC(*params)
    ref = _oracle_obs(s, cfgs, chi, om)
    got, psi = host.measure(synthetic.sitps_to_flat(s, D), cfgs, chi, model, params, dtype=dt)
    assert set(got) == set(ref[0][0])
    for w, (obs, (pm, prel)) in enumerate(ref):
        for key, want in obs.items():
            want = np.asarray(want, dtype=np.float64)
            assert got[key][w].shape == want.shape, key
            assert np.max(np.abs(got[key][w] - want)) < tol * 10 * max(1.0, np.max(np.abs(want))), (key, w)
        assert abs(psi[0][w] / pm - 1) < tol * 10
        assert abs(psi[1][w] - prel) < tol * 10
    sp = np.array([np.asarray(o["SpSm_row"]) + np.asarray(o["SmSp_row"]) for o, _ in ref])
    assert np.count_nonzero(sp) > 0                                # the off-diagonal channel is exercised


@pytest.mark.parametrize("dt,tol", [(F64, 1e-9), (F32, 5e-5)])
def test_triangle_j1j2_registry_observables_fixed_configs(dt, tol):
    """SpinOneHalfTriJ1J2HeisenbergSqrPEPS::EvaluateObservables (spin_onehalf_triangle_heisenbergJ1J2_sqrpeps.h:65-297): energy, spin_z,
    the three J1 bond maps, SzSz_row / SmSp_row / SpSm_row of the middle row, SzSz_all2all and the psi summary against the oracle on
    identical configurations (6x6)."""
    host = _host()
    L, D, chi, j2 = 6, 3, 9, 0.2
    s = synthetic.make_sitps(L, D)
    cfgs = synthetic.make_configs(L, 4, "heisenberg")
    tp = BMPSTruncateParams.SVD(chi, chi, 0.0)
    got, psi = host.measure(synthetic.sitps_to_flat(s, D), cfgs, chi, "trij1j2", (j2,), dtype=dt)
    nz = 0
    for w, c in enumerate(cfgs):
        model = vmc.SpinOneHalfTriJ1J2HeisenbergSqrPEPS(j2)
        obs = model.EvaluateObservables(s, vmc.TPSWaveFunctionComponent(s, c, tp))
        assert set(got) == set(obs)
        for key, want in obs.items():
            want = np.asarray(want, dtype=np.float64)
            assert got[key][w].shape == want.shape, key
            assert np.max(np.abs(got[key][w] - want)) < tol * 10 * max(1.0, np.max(np.abs(want))), (key, w)
        assert abs(psi[0][w] / model.last_psi_summary[0] - 1) < tol * 10
        assert abs(psi[1][w] - model.last_psi_summary[1]) < tol * 10
        nz += np.count_nonzero(obs["SpSm_row"]) + np.count_nonzero(obs["SmSp_row"])
    assert nz > 0


def test_registry_observables_k5_fixture(fixtures_dir):
    """the reference's 4x4 D=8 Heisenberg state (tests/slow_tests/test_data/tps_square_heisenberg4x4D8Double)"""
    host = _host()
    s = qlten_io.load_sitps(os.path.join(fixtures_dir, "tps_square_heisenberg4x4D8Double"))
    cfgs = np.stack([synthetic.checkerboard(4)] + list(synthetic.make_configs(4, 3, "heisenberg")))
    ref = _oracle_obs(s, cfgs, 16, vmc.SquareSpinOneHalfXXZModelOBC())
    got, psi = host.measure(synthetic.sitps_to_flat(s, 8), cfgs, 16, "xxz", (1.0, 1.0, 0.0), dtype=F64)
    for w, (obs, (pm, prel)) in enumerate(ref):
        for key, want in obs.items():
            assert np.max(np.abs(got[key][w] - np.asarray(want))) < 1e-8 * max(1.0, np.max(np.abs(want))), (key, w)


@pytest.mark.parametrize("updater,cls", [("exchange", "MCUpdateSquareNNExchangeOBC"), ("fullspace", "MCUpdateSquareNNFullSpaceUpdateOBC")])
def test_mc_measurer_identical_chain_statistics(updater, cls, tmp_path):
    """MCPEPSMeasurer: warm-up + samples with the same std::mt19937 streams as the oracle chains (f64): the per-walker
    sample means, and mean / standard error across the walkers (one walker = one MPI rank of the reference,
    GatherStatisticListOfData), agree; DumpData writes stats/*.csv and samples/psi.csv."""
    host = _host()
    L, D, chi = 4, 3, 9
    s = synthetic.make_sitps(L, D)
    flat = synthetic.sitps_to_flat(s, D)
    cfgs = synthetic.make_configs(L, 4, "heisenberg")
    seeds = np.array([21, 22, 23, 24], dtype=np.uint64)
    warm, nsamp, between = 2, 3, 2
    tp = BMPSTruncateParams.SVD(chi, chi, 0.0)
    model = vmc.SquareSpinOneHalfXXZModelOBC(1.0, 1.0, 0.0)
    means = []
    for w in range(len(cfgs)):
        comp = vmc.TPSWaveFunctionComponent(s, cfgs[w], tp)
        upd = getattr(vmc, cls)(seed=int(seeds[w]))
        for _ in range(warm):
            upd(s, comp)
        acc = {}
        for _ in range(nsamp):
            for _ in range(between):
                upd(s, comp)
            obs = vmc.SquareNNNModelMeasurementSolver(model).EvaluateObservables(s, comp)
            for k, v in obs.items():
                acc[k] = acc.get(k, 0.0) + np.asarray(v, dtype=np.float64)
        means.append({k: v / nsamp for k, v in acc.items()})
    run_cfgs = cfgs.copy()
    got, psi = host.measure(flat, run_cfgs, chi, "xxz", (1.0, 1.0, 0.0), seeds=seeds, updater=updater, warmup_sweeps=warm,
                            n_samples=nsamp, sweeps_between_samples=between, dump_dir=str(tmp_path), dtype=F64)
    from oracle import statistics
    n = len(cfgs)
    for key in means[0]:
        stack = np.stack([m[key] for m in means])
        mean, err = statistics.gather_statistic_list_of_data(stack)          # oracle/statistics.py (statistics.h:288-340, pinned on test_statistics.cpp)
        assert np.max(np.abs(got[key][0] - mean)) < 1e-8 * max(1.0, np.max(np.abs(mean))), key
        assert np.max(np.abs(got[key][1] - err)) < 1e-8 * max(1.0, np.max(np.abs(err))), key
    assert os.path.exists(tmp_path / "stats" / "energy.csv")
    assert os.path.exists(tmp_path / "stats" / "bond_energy_h_mean.csv") and os.path.exists(tmp_path / "stats" / "spin_z_stderr.csv")
    rows = open(tmp_path / "stats" / "energy.csv").read().strip().split("\n")
    assert rows[0] == "index,mean,stderr" and abs(float(rows[1].split(",")[1]) - got["energy"][0][0]) < 1e-12
    mat = np.loadtxt(tmp_path / "stats" / "bond_energy_h_mean.csv", delimiter=",")
    assert mat.shape == (L, L - 1) and np.max(np.abs(mat.ravel() - got["bond_energy_h"][0])) < 1e-12
    psi_rows = open(tmp_path / "samples" / "psi.csv").read().strip().split("\n")
    assert len(psi_rows) == 1 + nsamp * n


@pytest.mark.parametrize("dt,tol,chi", [(F64, 1e-9, 9), (F64, 1e-9, 27), (F32, 5e-5, 9)])
def test_structure_factor_cross_row_spsm(dt, tol, chi):
    """StructureFactorMeasurementMixin::MeasureStructureFactor through the BMPSWalker-equivalent stack operations
    (park / unpark of the DOWN levels, UP stack as the walker): every tuple of SpSm_cross against the oracle restatement,
    at a truncating and at an exact chi; the registry keys measured before it are unchanged."""
    host = _host()
    L, D = 4, 3
    s = synthetic.make_sitps(L, D)
    cfgs = synthetic.make_configs(L, 4, "heisenberg")
    tp = BMPSTruncateParams.SVD(chi, chi, 0.0)
    params = (1.0, 1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0)          # params[7] = structure factor on
    got, _ = host.measure(synthetic.sitps_to_flat(s, D), cfgs, chi, "xxz", params, dtype=dt)
    plain, _ = host.measure(synthetic.sitps_to_flat(s, D), cfgs, chi, "xxz", params[:3], dtype=dt)
    assert "SpSm_cross" in got and "SpSm_cross" not in plain
    for key in plain:
        assert np.array_equal(plain[key], got[key]), key
    for w, cfg in enumerate(cfgs):
        comp = vmc.TPSWaveFunctionComponent(s, cfg, tp)
        want = np.array(vmc.measure_structure_factor(s, comp)).reshape(-1, 5)
        have = got["SpSm_cross"][w].reshape(-1, 5)
        assert have.shape == want.shape
        assert np.array_equal(have[:, :4], want[:, :4])
        scale = np.max(np.abs(want[:, 4]))
        assert scale > 0 and np.max(np.abs(have[:, 4] - want[:, 4])) < tol * 10 * scale


@pytest.mark.parametrize("dt,tol", [(F64, 1e-9), (F32, 5e-5)])
def test_exact_sum_measurer_all_binary_configs(dt, tol):
    """ExactSumMeasurerMPI + GenerateAllBinaryConfigs (exact_summation_measurer.h:53-72, 103-257) through the C++ solver:
    every registry key, weighted by |psi|^2 over all 64 configurations of a 2x3 lattice, against the oracle; the rank
    decomposition adds up to the serial result (test_exact_summation_measurer.cpp:276-290); empty list rejected."""
    host = _host()
    Ly, Lx, D, chi = 2, 3, 3, 9
    rng = np.random.default_rng(5)
    s = [[[rng.standard_normal((1 if c == 0 else D, 1 if r == Ly - 1 else D, 1 if c == Lx - 1 else D, 1 if r == 0 else D)) + 0.3
           for _ in range(2)] for c in range(Lx)] for r in range(Ly)]
    flat = np.zeros((Ly, Lx, 2, D, D, D, D))
    for r in range(Ly):
        for c in range(Lx):
            for k in range(2):
                t = s[r][c][k]
                flat[r, c, k, :t.shape[0], :t.shape[1], :t.shape[2], :t.shape[3]] = t
    params = (1.0, 0.7, 0.2)
    tp = BMPSTruncateParams.SVD(chi, chi, 0.0)
    all_cfgs = vmc.generate_all_binary_configs(Lx, Ly)
    assert len(all_cfgs) == 64 and all_cfgs[5][0].tolist() == [1, 0, 1] and all_cfgs[8][1].tolist() == [1, 0, 0]
    want = vmc.exact_sum_measure(s, all_cfgs, tp, lambda: vmc.SquareNNNModelMeasurementSolver(vmc.SquareSpinOneHalfXXZModelOBC(*params)))
    acc, w = host.exact_sum_measure_partial(flat, None, chi, "xxz", params, 0, 1, 64, dt)
    assert set(acc) == set(want)
    for key, v in want.items():
        assert acc[key].shape == v.shape, key
        assert np.max(np.abs(acc[key] / w - v)) < tol * 10 * max(1.0, np.max(np.abs(v))), key
    parts = [host.exact_sum_measure_partial(flat, np.array(all_cfgs), chi, "xxz", params, r, 3, 7, dt) for r in range(3)]
    wp = sum(p[1] for p in parts)
    assert abs(wp / w - 1) < tol
    for key in acc:
        assert np.max(np.abs(sum(p[0][key] for p in parts) / wp - acc[key] / w)) < tol * 10
    with pytest.raises(Exception):
        host.exact_sum_measure_partial(flat, np.zeros((0, Ly, Lx), dtype=np.int32), chi, "xxz", params, 0, 1, 4, dt)
    assert host.exact_sum_measure_partial(flat, np.array(all_cfgs[:2]), chi, "xxz", params, 3, 4, 4, dt) == ({}, 0.0)


def _bond(sitps):
    return max(max(t.shape) for row in sitps for site in row for t in site)


def test_k8_reference_structure_factor_regression_on_the_device(fixtures_dir):
    """K8 through the HIP path (f64 mode): the REFERENCE's regression vector of MCPEPSMeasurer -- tests/test_model_solvers/
    test_square_xxz_measurer.cpp:204-381: 96 SpSm_cross values at 1e-10, energy -9.22 +- 0.01 -- 4x4 D = 8 Heisenberg fixture,
    checkerboard start, MCUpdateSquareNNExchange(42), 5 warm-up sweeps, NormalizeStateOrder1, 5 samples, SVD(8, 16, 1e-15).
    MCPEPSMeasurer of the host layer (its Execute runs the engine's WarmUp incl. the order-1 rescale, round 5: the first run on a GPU
    found it missing -- every value off by the common factor 41.7); params[6] = 1: the structure factor in the stack state the
    reference measures it in.  (Staged at the end of round 4, first run in round 5.)"""
    import json
    host = _host()
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "xxz_spsm_cross_reference_golden.json")))
    s = qlten_io.load_sitps(os.path.join(fixtures_dir, "tps_square_heisenberg4x4D8Double"))
    cfg = np.array([[[(r + c) % 2 for c in range(4)] for r in range(4)]], dtype=np.int32)
    host.set_truncate_params(8, 1e-15, 0)
    try:
        out, _ = host.measure(synthetic.sitps_to_flat(s, _bond(s)), cfg, 16, "xxz", (1.0, 1.0, 0.0, 0.0, 0.0, 0.0, 1.0, 1.0), seeds=[42],
                              updater="exchange", warmup_sweeps=5, n_samples=5, sweeps_between_samples=1, dtype=F64)
    finally:
        host.set_truncate_params()
    vals = np.asarray(out["SpSm_cross"][0]).reshape(-1, 5)[:, 4]
    want = np.array(gold["spsm_cross_values"])
    print("K8 on the device: max |SpSm_cross - reference| = %.2e over %d values" % (np.max(np.abs(vals - want)), len(want)))
    assert np.max(np.abs(vals - want)) < 1e-10
    assert np.count_nonzero(vals) == np.count_nonzero(want) == 44
    assert abs(out["energy"][0][0] - gold["energy"]) < gold["energy_tol"]


def test_tfim_registry_on_the_device(fixtures_dir):
    """TransverseFieldIsingSquareOBC::EvaluateObservables of the host layer on all 16 configurations of the reference's 2x2 state against
    the oracle's registry (itself pinned on tests/test_algorithm/test_exact_summation_measurer.cpp:548-651 at 1e-10)"""
    host = _host()
    s = qlten_io.load_sitps(os.path.join(fixtures_dir, "transverse_ising_tps_double_from_simple_update"))
    cfgs = np.array(vmc.generate_all_binary_configs(2, 2), dtype=np.int32)
    got, _ = host.measure(synthetic.sitps_to_flat(s, _bond(s)), cfgs, 8, "tfim", (1.0,), dtype=F64)
    tp = BMPSTruncateParams.SVD(8, 8, 0.0)
    for w, c in enumerate(cfgs):
        want = vmc.TransverseFieldIsingSquareOBC(1.0).EvaluateObservables(s, vmc.TPSWaveFunctionComponent(s, c, tp))
        assert set(got) == set(want)
        for key, v in want.items():
            assert np.max(np.abs(got[key][w] - np.asarray(v, dtype=np.float64))) < 1e-9, (key, w)
