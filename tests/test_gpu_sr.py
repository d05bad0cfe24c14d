"""Stochastic reconfiguration on the device: the O* samples stay in HBM; S-matrix product and the
CG solve against the oracle restatement of SRSMatrix / ConjugateGradientSolver built from host-side holes."""
import numpy as np
import pytest

from oracle import sr as osr
from peps_amd import synthetic

pytestmark = pytest.mark.gpu


def _collect(ctx, cfgs, L, D):
    """punch every hole of every walker: once to the host (reference O* samples) and once into the device store"""
    from peps_amd.capi import LEFT, RIGHT, UP, DOWN, HORIZONTAL
    n = len(cfgs)
    ctx.set_configs(cfgs)
    psi = ctx.evaluate_amplitude()
    ctx.set_configs(cfgs)
    ctx.generate_bmps_approach(UP)
    ostar = np.zeros((n, L, L, 2, D, D, D, D))
    for row in range(L):
        ctx.init_bten(LEFT, row)
        ctx.grow_full_bten(RIGHT, row, 1, True)
        for col in range(L):
            h = ctx.punch_hole(row, col, HORIZONTAL)
            ctx.punch_hole_store(row, col, HORIZONTAL)
            for w in range(n):
                ostar[w, row, col, cfgs[w, row, col]] = h[w] / psi[w]
            if col < L - 1:
                ctx.shift_bten_window(RIGHT)
        if row < L - 1:
            ctx.shift_bmps_window(DOWN)
    return psi, ostar


@pytest.mark.parametrize("dt,tol", [("f64", 1e-10), ("f32", 2e-5)])
def test_sr_matvec_and_cg(dt, tol):
    from peps_amd import capi, sr
    L, D, chi = 4, 3, 9
    sitps = synthetic.make_sitps(L, D)
    ctx = capi.Context(L, L, D, 2, chi, dtype=capi.F32 if dt == "f32" else capi.F64, max_walkers=24)
    ctx.state_upload(synthetic.sitps_to_flat(sitps, D))
    ctx.sr_begin(48)
    samples = []
    for batch in range(2):                       # two batches of walkers appended to the store
        cfgs = synthetic.make_configs(L, 24, "heisenberg", seed0=100 + 50 * batch)
        psi, ostar = _collect(ctx, cfgs, L, D)
        ctx.sr_append(psi)
        samples += list(ostar)
    assert ctx.sr_count() == 48
    mean = np.mean(samples, axis=0)
    S = sr.DeviceSRSMatrix(ctx, diag_shift=1e-3)
    assert np.max(np.abs(S.mean - mean)) < tol * np.max(np.abs(mean)) * 10
    ref = osr.SRSMatrix(samples, mean, 1, 1e-3)
    rng = np.random.default_rng(0)
    for _ in range(3):
        v = rng.standard_normal(mean.shape)
        got, want = S * v, (ref * v).reshape(mean.shape)
        assert np.max(np.abs(got - want)) < tol * 10 * np.max(np.abs(want))
    # natural gradient: (S + shift) x = g
    g = (ref * rng.standard_normal(mean.shape)).reshape(mean.shape)     # a right-hand side in the range of S
    x, res, it = sr.conjugate_gradient(S, g, max_iter=200, relative_tolerance=1e-8)
    xo, reso, ito = osr.conjugate_gradient(lambda y: ref * y, g, np.zeros(g.size), 200, 1e-8)
    assert res <= 1e-8 * np.linalg.norm(g) * 1.01
    assert np.linalg.norm(x.ravel() - xo) < max(tol * 1e3, 1e-6) * np.linalg.norm(xo)
