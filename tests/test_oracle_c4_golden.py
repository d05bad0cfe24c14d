"""The committed oracle fixture of the C4 real-state energy test (tests/golden/c4_real_energy_golden.json, checked against the device in
tests/test_gpu_realrank.py::test_real_rank_c4_energy_vs_oracle_golden) is what the oracle computes: two of its sixteen configurations
are recomputed here (oracle/epool.py, two worker processes, ~25 s)."""
import json
import os

import numpy as np

from conftest import FIXTURES, GOLDEN


def test_c4_real_energy_golden_is_reproducible():
    from oracle import epool
    from peps_amd import hostapi, synthetic
    g = json.load(open(os.path.join(GOLDEN, "c4_real_energy_golden.json")))
    L, chi = g["L"], g["chi"]
    cfgs = np.array(g["configs"], dtype=np.int32)
    assert np.array_equal(cfgs, synthetic.make_configs_near_neel(L, len(cfgs), seed0=g["seed0"]))
    flat = synthetic.tile_flat_state(hostapi.load_sitps(os.path.join(FIXTURES, synthetic.REAL_FIXTURE), 8), L)
    pick = [0, 7]
    h = epool.start(flat, cfgs[pick], chi, (1.0, 1.0, 0.0), nprocs=2, blas_threads=2)
    e, a, _ = epool.collect(h, timeout=900)
    for k, i in enumerate(pick):
        assert abs(e[k] / g["energy"][i] - 1) < 1e-12
    assert abs((a[1] / a[0]) / (g["psi_over_psi0"][7] / g["psi_over_psi0"][0]) - 1) < 1e-10
