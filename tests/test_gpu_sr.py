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


def _store(dt, L=4, D=3, chi=9, nb=24, batches=2):
    from peps_amd import capi
    sitps = synthetic.make_sitps(L, D)
    ctx = capi.Context(L, L, D, 2, chi, dtype=capi.F32 if dt == "f32" else capi.F64, max_walkers=nb)
    ctx.state_upload(synthetic.sitps_to_flat(sitps, D))
    ctx.sr_begin(nb * batches)
    samples = []
    for batch in range(batches):
        cfgs = synthetic.make_configs(L, nb, "heisenberg", seed0=300 + 50 * batch)
        psi, ostar = _collect(ctx, cfgs, L, D)
        ctx.sr_append(psi)
        samples += list(ostar)
    return ctx, samples


def _structural_mask(L=4, D=3):
    """1 where the zero padded state layout holds a tensor element (boundary legs have dimension 1)"""
    return (synthetic.sitps_to_flat(synthetic.make_sitps(L, D), D).reshape(L, L, 2, D, D, D, D) != 0).astype(np.float64)


@pytest.mark.parametrize("dt,tol", [("f64", 1e-5), ("f32", 2e-3)])
def test_device_resident_cg_follows_the_reference_solver(dt, tol):
    """pepsgpu_sr_cg_solve (every vector in HBM) against the oracle restatement of ConjugateGradientSolver
    (utility/conjugate_gradient_solver.h:181-276): same iterate, iteration count and termination reason, including the
    periodic residual recomputation and the max-iteration / best-iterate exit."""
    ctx, samples = _store(dt)
    mean = np.mean(samples, axis=0)
    shp = mean.shape
    rng = np.random.default_rng(1)
    mask = _structural_mask()          # the device vectors live on the tensor elements only, not on the padding
    for shift, interval in ((1e-3, 20), (1e-2, 3)):
        ref = osr.SRSMatrix(samples, mean, 1, shift)
        g = (ref * (mask * rng.standard_normal(shp))).reshape(shp)
        x, res, it, why = ctx.sr_cg_solve(g, None, shift, 200, 1e-8, 0.0, interval, 0.5)
        xo, reso, ito, whyo = osr.conjugate_gradient_full(lambda y: ref * y, g, np.zeros(g.size), 200, 1e-8, 0.0, interval, 0.5)
        assert why == whyo == osr.K_CONVERGED
        # 50-100 iterations on a rank-48 system with restarts decided by thresholds: the summation order of the dots
        # moves the exit by a few steps; the iterate itself is pinned below and, step for step, by the 3-iteration run
        assert abs(it - ito) <= max(3, ito // 8)
        assert np.linalg.norm(x.ravel() - xo) < tol * np.linalg.norm(xo)
        assert res <= 1e-8 * np.linalg.norm(g) * 1.01
        # warm start from the solution: converged at once
        x2, res2, it2, why2 = ctx.sr_cg_solve(g, x, shift, 200, 1e-6, 0.0, interval, 0.5)
        assert it2 == 0 and why2 == osr.K_CONVERGED
    ref = osr.SRSMatrix(samples, mean, 1, 1e-3)
    g = (ref * (mask * rng.standard_normal(shp))).reshape(shp)
    x, res, it, why = ctx.sr_cg_solve(g, None, 1e-3, 3, 1e-14, 0.0, 20, 0.5)
    xo, reso, ito, whyo = osr.conjugate_gradient_full(lambda y: ref * y, g, np.zeros(g.size), 3, 1e-14, 0.0, 20, 0.5)
    assert (it, why) == (ito, whyo) == (3, osr.K_MAX_ITERATIONS)
    assert abs(res / reso - 1) < max(tol, 1e-6) and np.linalg.norm(x.ravel() - xo) < tol * np.linalg.norm(xo)
    with pytest.raises(ValueError):
        ctx.sr_cg_solve(g, None, 0.0, -1, 1e-8)


@pytest.mark.parametrize("dt,tol", [("f64", 1e-10), ("f32", 2e-5)])
def test_minsr_direction_on_device(dt, tol):
    """MinSR (optimizer/minsr_tmatrix.h, minsr_eigensolve.h, optimizer_impl.h:1126-1215): Gram of the resident samples as
    one GEMM, T matrix, pseudo-inverse, back-substitution -- against the oracle; and the reference's own check
    (tests/test_algorithm/test_sr_vs_minsr_equivalence.cpp): MinSR with zero cutoff == SR solved by CG without shift."""
    from peps_amd import sr
    ctx, samples = _store(dt)
    n = len(samples)
    flat = np.stack([s.ravel() for s in samples])
    g = ctx.sr_gram()
    want = flat @ flat.T
    assert np.max(np.abs(g - want)) < tol * np.max(np.abs(want))
    rng = np.random.default_rng(2)
    y = rng.standard_normal(n)
    ws = ctx.sr_weighted_sum(y)
    assert np.max(np.abs(ws.ravel() - y @ flat)) < tol * 10 * np.max(np.abs(y @ flat))
    e_loc = rng.standard_normal(n)
    e_mean = float(e_loc.mean())
    mean = np.mean(samples, axis=0)
    batch = sr.DeviceSampleBatch(ctx)
    for kw in ({"r_pinv": 1e-12, "a_pinv": 0.0, "soft_cutoff": True}, {"r_pinv": 1e-6, "a_pinv": 1e-9, "soft_cutoff": False}):
        if dt == "f32" and kw["soft_cutoff"]:
            kw = dict(kw, r_pinv=1e-5)          # keep the cutoff above the f32 rounding of the Gram entries
        d, nrm = sr.minsr_direction(batch, e_loc, e_mean, **kw)
        do, nrmo = osr.minsr_direction(samples, mean, e_loc, e_mean, **kw)
        assert np.linalg.norm(d - do) < max(tol * 1e3, 1e-7) * nrmo, kw
        assert abs(nrm / nrmo - 1) < max(tol * 1e3, 1e-7)
    if dt == "f64":
        grad = ((e_loc - e_mean)[:, None] * flat).mean(axis=0).reshape(mean.shape)
        x, res, it, why = sr.natural_gradient(ctx, grad, 0.0, max_iter=500, relative_tolerance=1e-13)
        d, _ = sr.minsr_direction(batch, e_loc, e_mean, r_pinv=0.0, a_pinv=1e-11, soft_cutoff=False)
        assert why == sr.K_CONVERGED
        assert np.linalg.norm(x.ravel() - d) < 1e-7 * np.linalg.norm(d)


def test_sr_matvec_complex_vs_oracle():
    """TenElemT = QLTEN_Complex (SRSMatrix is templated over it, stochastic_reconfiguration_smatrix.h:36-99): O* samples of a complex
    state resident in HBM (O*_i = conj(1 / psi_i) Dag(hole_i), mc_energy_grad_evaluator.h:245-270), their sum and the S-matrix product
    with the positive-definite pairing <a, b> = sum conj(a) b of SplitIndexTPS::operator* (split_index_tps.h:370-377), against the
    oracle's SRSMatrix built from host-side holes: 1e-9."""
    from peps_amd import capi, sr
    from peps_amd.capi import LEFT, RIGHT, UP, DOWN, HORIZONTAL
    L, D, chi, n = 4, 3, 9, 16
    rng = np.random.default_rng(12)
    sitps = synthetic.make_sitps(L, D)
    flat = synthetic.sitps_to_flat(sitps, D).astype(np.complex128)
    flat = flat * np.exp(2j * np.pi * rng.random(flat.shape)) * (np.abs(flat) > 0)
    ctx = capi.Context(L, L, D, 2, chi, dtype=capi.C128, max_walkers=n)
    ctx.state_upload(flat)
    ctx.sr_begin(2 * n)
    samples = []
    for batch in range(2):
        cfgs = synthetic.make_configs(L, n, "heisenberg", seed0=300 + 40 * batch)
        ctx.set_configs(cfgs)
        psi = ctx.evaluate_amplitude()
        ctx.set_configs(cfgs)
        ctx.generate_bmps_approach(UP)
        ostar = np.zeros((n, L, L, 2, D, D, D, D), dtype=np.complex128)
        for row in range(L):
            ctx.init_bten(LEFT, row)
            ctx.grow_full_bten(RIGHT, row, 1, True)
            for col in range(L):
                h = ctx.punch_hole(row, col, HORIZONTAL)
                ctx.punch_hole_store(row, col, HORIZONTAL)
                for w in range(n):
                    ostar[w, row, col, cfgs[w, row, col]] = np.conj(h[w]) / np.conj(psi[w])      # conj(1 / psi) Dag(hole)
                if col < L - 1:
                    ctx.shift_bten_window(RIGHT)
            if row < L - 1:
                ctx.shift_bmps_window(DOWN)
        ctx.sr_append(psi)
        samples += list(ostar)
    assert ctx.sr_count() == 2 * n
    total = ctx.sr_sum()
    want_sum = np.sum(samples, axis=0)
    assert np.max(np.abs(total - want_sum)) < 1e-9 * np.max(np.abs(want_sum))
    mean = want_sum / (2 * n)
    ref = osr.SRSMatrix(samples, mean, 1, 0.0)
    for _ in range(3):
        v = rng.standard_normal(mean.shape) + 1j * rng.standard_normal(mean.shape)
        got = ctx.sr_matvec(v, np.vdot(mean.ravel(), v.ravel()), 1.0 / (2 * n))
        want = (ref * v).reshape(mean.shape)
        assert np.max(np.abs(got - want)) < 1e-9 * np.max(np.abs(want))
    # S is Hermitian positive semi-definite under that pairing
    v = rng.standard_normal(mean.shape) + 1j * rng.standard_normal(mean.shape)
    sv = ctx.sr_matvec(v, np.vdot(mean.ravel(), v.ravel()), 1.0 / (2 * n))
    q = np.vdot(v.ravel(), sv.ravel())
    assert q.real > 0 and abs(q.imag) < 1e-9 * q.real
    # the conjugate-gradient solve on the device (round 4: complex contexts too) against the oracle's solver on the oracle's S:
    # same iteration count, same termination reason, same solution
    b = (ref * (rng.standard_normal(mean.shape) + 1j * rng.standard_normal(mean.shape))).reshape(mean.shape) \
        + 0.01 * (rng.standard_normal(mean.shape) + 1j * rng.standard_normal(mean.shape)) * (np.abs(mean) > 0)
    # (detail::pap_is_valid asks |Im p*(A p)| < 1e-10 ABSOLUTE: with O* of this size the first product already fails it by rounding --
    # oracle and device agree on that exit; the solves below run on a right-hand side scaled so that p*(A p) = O(1))
    full = osr.SRSMatrix(samples, mean, 1, 1e-3)
    _, _, ir, why_r = osr.conjugate_gradient_full(lambda x: full * x, b.ravel(), np.zeros(b.size, dtype=np.complex128), 50, 1e-8, 0.0, 20, 0.5)
    _, _, idv, why_d = ctx.sr_cg_solve(b, None, 1e-3, 50, 1e-8, 0.0, 20, 0.5)
    assert (idv, why_d) == (ir, why_r)
    b = b / np.sqrt(abs(np.vdot(b.ravel(), full * b.ravel())))
    for shift, rtol, recompute in ((1e-3, 1e-8, 20), (1e-2, 1e-5, 3)):
        full = osr.SRSMatrix(samples, mean, 1, shift)
        xr, rr, ir, why_r = osr.conjugate_gradient_full(lambda x: full * x, b.ravel(), np.zeros(b.size, dtype=np.complex128), 200, rtol, 0.0,
                                                        recompute, 0.5)
        xd, rd, idv, why_d = ctx.sr_cg_solve(b, None, shift, 200, rtol, 0.0, recompute, 0.5)
        assert why_d == why_r == osr.K_CONVERGED and abs(idv - ir) <= 1, (idv, ir, why_d, why_r)
        # (both iterates satisfy the termination criterion; between them stands the conditioning of S + shift times the rounding of
        # two different summation orders)
        # (the iteration counts may differ by one: the residual of the later exit is smaller by a CG contraction factor -- the device's
        # must not be WORSE than the oracle's by more than that)
        assert np.max(np.abs(xd.ravel() - xr)) < 1e-4 * np.max(np.abs(xr)) and rd < 1e-6 * np.linalg.norm(b.ravel()) + 1.5 * rr
        assert np.linalg.norm((full * xd.ravel()) - b.ravel()) <= 1.5 * rtol * np.linalg.norm(b.ravel())
    # MinSR on the complex store (MinSRTMatrix / ReplicatedEigenSolveComplex are templated over TenElemT): Gram with the conjugated
    # pairing on the device, epsilon_bar = conj(E_loc - E) / Ns, back-substitution with complex weights -- against the oracle
    ip = ctx.sr_gram()
    o2 = np.stack([x.ravel() for x in samples])
    want_ip = o2.conj() @ o2.T
    assert np.max(np.abs(ip - want_ip)) < 1e-9 * np.max(np.abs(want_ip)) and np.max(np.abs(ip - ip.conj().T)) == 0.0
    yw = rng.standard_normal(2 * n) + 1j * rng.standard_normal(2 * n)
    ws = ctx.sr_weighted_sum(yw)
    assert np.max(np.abs(ws.ravel() - yw @ o2)) < 1e-9 * np.max(np.abs(yw @ o2))
    e_loc = rng.standard_normal(2 * n) + 1j * rng.standard_normal(2 * n)
    for kw in ({"r_pinv": 1e-12, "a_pinv": 0.0, "soft_cutoff": True}, {"r_pinv": 1e-6, "a_pinv": 0.0, "soft_cutoff": False}):
        dm, nm = sr.minsr_direction(sr.DeviceSampleBatch(ctx), e_loc, complex(e_loc.mean()), **kw)
        do, no = osr.minsr_direction(samples, mean, e_loc, complex(e_loc.mean()), **kw)
        assert np.linalg.norm(dm.ravel() - do) < 1e-7 * no and abs(nm - no) < 1e-7 * no, (kw, np.linalg.norm(dm.ravel() - do), no)
    # the host-vector solver of the multi-rank path (peps_amd/sr.py: one all-reduce per product) on the same complex store
    S = sr.DeviceSRSMatrix(ctx, diag_shift=1e-2)
    xh, rh, ih, why_h = sr.conjugate_gradient(S, b, max_iter=200, relative_tolerance=1e-5, residual_recompute_interval=3, full_output=True)
    assert why_h == osr.K_CONVERGED and abs(ih - idv) <= 1 and np.max(np.abs(xh - xd)) < 1e-4 * np.max(np.abs(xd))
    assert np.linalg.norm((osr.SRSMatrix(samples, mean, 1, 1e-2) * xh.ravel()) - b.ravel()) <= 1.5e-5 * np.linalg.norm(b.ravel())
    ctx.close()
