"""CPU suite: the plain-C / LAPACK restatement (oracle/cbmps.c, the cpu_baseline of bench.py) against the NumPy oracle that
is pinned on the reference's known answers, and against the reference's own 4x4 D=8 fixture (K5)."""
import os

import numpy as np
import pytest

from oracle import cbmps, qlten_io, vmc
from oracle.bmps import BMPSTruncateParams
from peps_amd import synthetic


@pytest.mark.parametrize("L,D,chi,noise", [(4, 2, 4, 0.1), (5, 3, 6, 1.0), (6, 4, 8, 0.3), (6, 3, 9, 1.0)])
def test_c_restatement_matches_numpy_oracle(L, D, chi, noise):
    """same op sequence (bmps_impl.h:756-862, :225-263), two independent codes: amplitudes agree to rounding"""
    sitps = synthetic.make_sitps(L, D, noise=noise)
    cfgs = synthetic.make_configs(L, 6, "heisenberg")
    flat = synthetic.sitps_to_flat(sitps, D)
    got, sec = cbmps.amplitudes(flat, cfgs, chi, nthreads=1)
    tp = BMPSTruncateParams.SVD(chi, chi, 0.0)
    ref = np.array([vmc.TPSWaveFunctionComponent(sitps, c, tp).amplitude for c in cfgs])
    assert np.max(np.abs(got / ref - 1)) < 1e-11
    assert sec > 0.0


def test_c_restatement_threads_are_independent_walkers():
    """one walker per thread (the reference's one-walker-per-rank model): same numbers whatever the thread count"""
    L, D, chi = 6, 4, 12
    flat = synthetic.sitps_to_flat(synthetic.make_sitps(L, D, noise=0.5), D)
    cfgs = synthetic.make_configs(L, 9, "heisenberg")
    a1, _ = cbmps.amplitudes(flat, cfgs, chi, nthreads=1)
    a4, _ = cbmps.amplitudes(flat, cfgs, chi, nthreads=4)
    assert np.array_equal(a1, a4)


def test_c_restatement_one_walker_per_process():
    """the all-cores cpu_baseline of bench.py: configurations sharded over single-threaded worker processes of a child interpreter"""
    L, D, chi = 5, 3, 6
    flat = synthetic.sitps_to_flat(synthetic.make_sitps(L, D, noise=0.5), D)
    cfgs = synthetic.make_configs(L, 7, "heisenberg")
    a1, _ = cbmps.amplitudes(flat, cfgs, chi, nthreads=1)
    a3, sec, nproc = cbmps.amplitudes_multiprocess(flat, cfgs, chi, 3)
    assert np.array_equal(a1, a3) and nproc == 3 and sec > 0.0


def test_c_restatement_on_reference_fixture_k5(fixtures_dir):
    """K5: checkerboard amplitude of the reference's 4x4 D=8 state, 1.441641034201432e+02 (brute force), BMPS chi=64 exact"""
    s = qlten_io.load_sitps(os.path.join(fixtures_dir, "tps_square_heisenberg4x4D8Double"))
    flat = synthetic.sitps_to_flat(s, 8)
    got, _ = cbmps.amplitudes(flat, synthetic.checkerboard(4)[None], 64, nthreads=1)
    assert abs(got[0] / 1.441641034201432e+02 - 1) < 1e-12
