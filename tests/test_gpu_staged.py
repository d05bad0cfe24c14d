"""STAGED device tests (written at the end of round 4 after the GPU budget of the round was spent): they have NOT been run on a GPU
yet and are therefore opt-in -- `PEPS_STAGED_TESTS=1 python -m pytest tests/test_gpu_staged.py -m gpu`.  Each one puts on the device
(f64 mode, through `libpepshost.so`) a number the REFERENCE produced and the oracle already reproduces on the CPU (DESIGN 2, K8 / K9 and the
TFIM registry); once green they move into the regular files."""
import json
import os

import numpy as np
import pytest

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(os.environ.get("PEPS_STAGED_TESTS") != "1", reason="staged in round 4 without GPU time: set PEPS_STAGED_TESTS=1")]
GOLD = os.path.join(os.path.dirname(__file__), "golden")
F64 = 1


def _host():
    from peps_amd import hostapi
    return hostapi


def _bond(sitps):
    return max(max(t.shape) for row in sitps for site in row for t in site)


def test_k9_tj_measurer_regression_energy_on_the_device(fixtures_dir):
    """tests/test_model_solvers/test_tJ_model_solver.cpp:72-75, 233-275: -14.74320489110316 +- 1e-8 (the oracle: 2e-15,
    tests/test_oracle_fermion.py::test_k9_reference_tj_measurer_regression_energy)"""
    from peps_amd import fermion
    host = _host()
    d = os.path.join(fixtures_dir, "tps_tJ_6x6Hole2_J0.3_D8_fU1")
    st = fermion.FermionState.load(d)
    cfg = np.loadtxt(os.path.join(d, "configuration0"), dtype=int).reshape(1, 6, 6)
    host.set_truncate_params(8, 1e-15, 0)                       # BMPSTruncateParams::SVD(8, 16, 1e-15)
    try:
        en, _, _ = host.fermion_measure_energy(st, cfg, [42], 16, 10, 10, 1, "tj", t=1.0, J=0.3, V=0.0, mu=0.0, dtype=F64)
    finally:
        host.set_truncate_params()
    assert abs(np.mean(en[:, 0]) - (-14.74320489110316)) < 1e-8


def test_k8_xxz_structure_factor_regression_on_the_device(fixtures_dir):
    """tests/test_model_solvers/test_square_xxz_measurer.cpp:204-381: 96 SpSm_cross values at 1e-10, energy -9.22 +- 0.01 (the oracle:
    6e-16, tests/test_oracle_measure.py::test_reference_structure_factor_regression_golden).  MCPEPSMeasurer of the host layer: warm-up,
    order-1 rescale, samples; params[6] = 1: the structure factor in the stack state the reference measures it in."""
    from oracle import qlten_io
    from peps_amd import synthetic
    host = _host()
    gold = json.load(open(os.path.join(GOLD, "xxz_spsm_cross_reference_golden.json")))
    s = qlten_io.load_sitps(os.path.join(fixtures_dir, "tps_square_heisenberg4x4D8Double"))
    cfg = np.array([[[(r + c) % 2 for c in range(4)] for r in range(4)]], dtype=np.int32)
    host.set_truncate_params(8, 1e-15, 0)
    try:
        out, _ = host.measure(synthetic.sitps_to_flat(s, _bond(s)), cfg, 16, "xxz", (1.0, 1.0, 0.0, 0.0, 0.0, 0.0, 1.0, 1.0), seeds=[42],
                              updater="exchange", warmup_sweeps=5, n_samples=5, sweeps_between_samples=1, dtype=F64)
    finally:
        host.set_truncate_params()
    vals = np.asarray(out["SpSm_cross"][0]).reshape(-1, 5)[:, 4]
    assert np.max(np.abs(vals - np.array(gold["spsm_cross_values"]))) < 1e-10
    assert abs(out["energy"][0][0] - gold["energy"]) < gold["energy_tol"]


def test_tfim_registry_on_the_device(fixtures_dir):
    """TransverseFieldIsingSquareOBC::EvaluateObservables of the host layer on all 16 configurations of the reference's 2x2 state against
    the oracle's registry (itself pinned on tests/test_algorithm/test_exact_summation_measurer.cpp:548-651 at 1e-10)"""
    from oracle import qlten_io, vmc
    from oracle.bmps import BMPSTruncateParams
    from peps_amd import synthetic
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
