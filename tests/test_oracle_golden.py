"""CPU suite: the oracle against the reference's own known answers (SURVEY.md section 8c K1,K3,K4,K5)."""
import os

import numpy as np
import pytest

from oracle import ising, qlten_io
from oracle.bmps import BMPSTruncateParams, LEFT, DOWN, RIGHT, UP, HORIZONTAL, VERTICAL
from oracle.contractor import BMPSContractor, TensorNetwork2D
from oracle import vmc


def test_k1_ising_free_energy_all_routes():
    """tests/test_2d_tn/test_bmps_contractor.cpp:273-405,472-493 (NN routes, SVD(10,30,1e-15)), tol 1e-8."""
    tn, lognorm, beta = ising.build_ising_tn(12, 12)
    f_ex = ising.exact_free_energy(12, 12, 1.0 / beta)
    c = BMPSContractor(12, 12)
    c.Init(tn)
    c.SetTruncateParams(BMPSTruncateParams.SVD(10, 30, 1e-15))
    amps = []
    c.GrowBMPSForRow(tn, 2)
    c.InitBTen(tn, LEFT, 2)
    c.GrowFullBTen(tn, RIGHT, 2, 2, True)
    amps.append(c.Trace(tn, (2, 0), HORIZONTAL))
    c.ShiftBTenWindow(tn, RIGHT)
    amps.append(c.Trace(tn, (2, 1), HORIZONTAL))
    c.GrowBMPSForCol(tn, 1)
    c.InitBTen(tn, DOWN, 1)
    c.GrowFullBTen(tn, UP, 1, 2, True)
    amps.append(c.Trace(tn, (10, 1), VERTICAL))
    c.ShiftBTenWindow(tn, UP)
    amps.append(c.Trace(tn, (9, 1), VERTICAL))
    # one-site trace route (trace.h:30-88) on row 2
    c.GrowBMPSForRow(tn, 2)
    c.GrowFullBTen(tn, LEFT, 2, 1, True)
    c.GrowFullBTen(tn, RIGHT, 2, 1, True)
    c.TruncateBTen(LEFT, 4)
    c.TruncateBTen(RIGHT, 12 - 3)
    amps.append(c.ReplaceOneSiteTrace(tn, (2, 3), tn((2, 3)), HORIZONTAL))
    for a in amps:
        assert abs(-(np.log(a) + lognorm) / 144 / beta - f_ex) < 1e-8


def test_k1_ising_all_21_routes_incl_bten2():
    """The complete route list of Contract2DTNUsingBMPSContractor (test_bmps_contractor.cpp:273-405):
    NN / TNN traces, BTen2 growth and window shifts, NNN and sqrt(5) replacement traces in both
    orientations and both diagonal directions; tolerance 1e-8 as the reference (:472-493)."""
    import k1_routes
    tn, lognorm, beta = ising.build_ising_tn(12, 12)
    f_ex = ising.exact_free_energy(12, 12, 1.0 / beta)
    c = BMPSContractor(12, 12)
    c.Init(tn)
    c.SetTruncateParams(BMPSTruncateParams.SVD(10, 30, 1e-15))
    amps = k1_routes.run_oracle(c, tn)
    assert len(amps) == k1_routes.N_AMPS
    for a in amps:
        assert abs(-(np.log(a) + lognorm) / 144 / beta - f_ex) < 1e-8


def test_k3_punch_hole_and_invalidate():
    """tests/test_2d_tn/test_bmps_contractor.cpp:407-470"""
    tn, _, _ = ising.build_ising_tn(12, 12)
    c = BMPSContractor(12, 12)
    c.Init(tn)
    c.SetTruncateParams(BMPSTruncateParams.SVD(4, 10, 1e-10))
    c.GrowBMPSForRow(tn, 2)
    c.GrowFullBTen(tn, LEFT, 2, 2, True)
    c.GrowFullBTen(tn, RIGHT, 2, 2, True)
    val1 = c.Trace(tn, (2, 0), HORIZONTAL)
    hole = c.PunchHole(tn, (2, 1), HORIZONTAL)
    tr = c.Trace(tn, (2, 1), HORIZONTAL)
    assert abs(np.tensordot(hole, tn((2, 1)), axes=4) - tr) < 1e-10
    tn.set((2, 1), tn((2, 1)) * 0.5)
    c.EraseEnvsAfterUpdate((2, 1))
    c.GrowBMPSForRow(tn, 2)
    c.GrowFullBTen(tn, LEFT, 2, 2, True)
    c.GrowFullBTen(tn, RIGHT, 2, 2, True)
    val2 = c.Trace(tn, (2, 0), HORIZONTAL)
    assert abs(val2 - 0.5 * val1) < 1e-10


K4 = [
    # directory, model, configs, reference energy, tolerance, reference citation
    ("heisenberg_tps_double_from_simple_update", "xxz", "perm22", -1.99521278793, 1e-10,
     "test_exact_summation_evaluator.cpp:606"),
    ("heisenberg_tps_doublelowest", "xxz", "perm22", -2.0, 6e-8, "test_exact_summation_evaluator.cpp:139-174"),
    ("transverse_ising_tps_double_from_simple_update", "tfim", "all", -5.19991995228, 1e-10,
     "test_exact_summation_evaluator.cpp:775"),
    ("transverse_ising_tps_doublelowest", "tfim", "all",
     -2.0 * (np.sqrt(2 - 2 * np.cos(np.pi / 4)) + np.sqrt(2 - 2 * np.cos(3 * np.pi / 4))), 6e-8,
     "test_exact_summation_evaluator.cpp:250-259"),
]


@pytest.mark.parametrize("name,model,cfgs,e_ref,tol,cite", K4)
def test_k4_2x2_exact_sum_energy(fixtures_dir, name, model, cfgs, e_ref, tol, cite):
    s = qlten_io.load_sitps(os.path.join(fixtures_dir, name))
    tp = BMPSTruncateParams.SVD(1, 8, 1e-16)   # truncation used by the reference test (:355,:870)
    if model == "xxz":
        m = vmc.SquareSpinOneHalfXXZModelOBC()
        configs = vmc.generate_all_permutation_configs([2, 2], 2, 2)
    else:
        m = vmc.TransverseFieldIsingSquareOBC(1.0)
        configs = vmc.all_product_configs(2, 2, 2)
    e, grad, w = vmc.exact_sum_energy_evaluator(s, configs, tp, m)
    assert abs(e - e_ref) < tol, cite


def test_k4_complex_matches_real(fixtures_dir):
    s = qlten_io.load_sitps(os.path.join(fixtures_dir, "heisenberg_tps_complex_from_simple_update"),
                            complex_data=True)
    tp = BMPSTruncateParams.SVD(1, 8, 1e-16)
    e, _, _ = vmc.exact_sum_energy_evaluator(s, vmc.generate_all_permutation_configs([2, 2], 2, 2), tp,
                                             vmc.SquareSpinOneHalfXXZModelOBC())
    assert abs(e - (-1.99521278793)) < 1e-10 and abs(e.imag) < 1e-12


def test_exact_sum_rank_partition(fixtures_dir):
    """exact_summation_energy_evaluator.h:201: round-robin partition over ranks sums to the same result
    (the reference runs this test under mpirun -n 4)."""
    s = qlten_io.load_sitps(os.path.join(fixtures_dir, "heisenberg_tps_double_from_simple_update"))
    tp = BMPSTruncateParams.SVD(1, 8, 1e-16)
    cfgs = vmc.generate_all_permutation_configs([2, 2], 2, 2)
    m = vmc.SquareSpinOneHalfXXZModelOBC()
    e1, g1, _ = vmc.exact_sum_energy_evaluator(s, cfgs, tp, m)
    parts = [vmc.exact_sum_partials(s, cfgs, tp, m, r, 4) for r in range(4)]
    so, seo, w, we = parts[0]
    for p in parts[1:]:
        for r in range(2):
            for c in range(2):
                for k in range(2):
                    so[r][c][k] = so[r][c][k] + p[0][r][c][k]
                    seo[r][c][k] = seo[r][c][k] + p[1][r][c][k]
        w += p[2]
        we += p[3]
    e4, g4, _ = vmc.finish_exact_sum(so, seo, w, we)
    assert abs(e1 - e4) < 1e-13
    assert np.allclose(g1[1][0][1], g4[1][0][1], atol=1e-13)


def test_k5_4x4_d8_amplitude(fixtures_dir):
    """tests/slow_tests/test_data/tps_square_heisenberg4x4D8Double: BMPS amplitude of the checkerboard
    configuration converges to the brute-force contraction (1.441641034201432e+02)."""
    s = qlten_io.load_sitps(os.path.join(fixtures_dir, "tps_square_heisenberg4x4D8Double"))
    cb = np.array([[(r + c) % 2 for c in range(4)] for r in range(4)])
    exact = ising.exact_contract(TensorNetwork2D.from_sitps(s, cb))
    assert abs(exact - 1.441641034201432e+02) < 1e-9
    comp = vmc.TPSWaveFunctionComponent(s, cb, BMPSTruncateParams.SVD(64, 64, 0.0))
    assert abs(comp.amplitude - exact) < 1e-9 * abs(exact)
    comp16 = vmc.TPSWaveFunctionComponent(s, cb, BMPSTruncateParams.SVD(16, 16, 0.0))
    assert abs(comp16.amplitude - exact) < 1e-5 * abs(exact)


def test_sweep_keeps_amplitude_consistent(fixtures_dir):
    """After a full MC sweep (square_nn_updater.h:30-81) the carried amplitude equals a fresh
    EvaluateAmplitude of the final configuration; Sz is conserved by the exchange updater."""
    s = qlten_io.load_sitps(os.path.join(fixtures_dir, "tps_square_heisenberg4x4D8Double"))
    cb = np.array([[(r + c) % 2 for c in range(4)] for r in range(4)])
    tp = BMPSTruncateParams.SVD(64, 64, 0.0)
    comp = vmc.TPSWaveFunctionComponent(s, cb, tp)
    upd = vmc.MCUpdateSquareNNExchangeOBC(seed=3)
    rates = upd(s, comp)
    assert 0.0 <= rates[0] <= 1.0
    assert comp.config.sum() == 8
    fresh = vmc.TPSWaveFunctionComponent(s, comp.config, tp)
    assert abs(fresh.amplitude - comp.amplitude) < 1e-9 * abs(fresh.amplitude)
    upd2 = vmc.MCUpdateSquareNNFullSpaceUpdateOBC(seed=5)
    upd2(s, comp)
    fresh = vmc.TPSWaveFunctionComponent(s, comp.config, tp)
    assert abs(fresh.amplitude - comp.amplitude) < 1e-9 * abs(fresh.amplitude)


def test_j1j2_nnn_solver_matches_dense_hamiltonian():
    """SquareSpinOneHalfJ1J2XXZModelOBC (NNN pass of square_nnn_energy_solver.h:203-265 through
    BTen2 / ReplaceNNNSiteTrace) against <psi|H|psi>/<psi|psi> with a dense 512x512 J1-J2 Hamiltonian
    built independently, 3x3 OBC, exact chi."""
    from peps_amd import synthetic
    sitps = synthetic.make_sitps(3, 2)
    j2 = 0.5
    model = vmc.SquareSpinOneHalfJ1J2XXZModelOBC(1.0, 1.0, j2, j2, 0.0)
    cfgs = vmc.all_product_configs(2, 3, 3)
    tp = BMPSTruncateParams.SVD(16, 16, 0.0)
    e, _, _ = vmc.exact_sum_energy_evaluator(sitps, cfgs, tp, model)
    psi = np.array([vmc.TPSWaveFunctionComponent(sitps, c, tp).amplitude for c in cfgs])
    idx = {tuple(c.ravel()): i for i, c in enumerate(cfgs)}
    bonds = []
    for r in range(3):
        for c in range(3):
            if c < 2: bonds.append(((r, c), (r, c + 1), 1.0))
            if r < 2: bonds.append(((r, c), (r + 1, c), 1.0))
            if r < 2 and c < 2:
                bonds.append(((r, c), (r + 1, c + 1), j2))
                bonds.append(((r + 1, c), (r, c + 1), j2))
    H = np.zeros((len(cfgs), len(cfgs)))
    for i, cf in enumerate(cfgs):
        for a, b, J in bonds:
            if cf[a] == cf[b]:
                H[i, i] += 0.25 * J
            else:
                H[i, i] -= 0.25 * J
                c2 = cf.copy()
                c2[a], c2[b] = cf[b], cf[a]
                H[idx[tuple(c2.ravel())], i] += 0.5 * J
    assert abs(e - psi @ H @ psi / (psi @ psi)) < 1e-12


def test_sr_oracle_matches_dense_algebra():
    """oracle/sr.py: SRSMatrix product == explicit covariance matrix, CG == dense solve (restatement of
    stochastic_reconfiguration_smatrix.h:37-99 and conjugate_gradient_solver.h)"""
    from oracle import sr
    rng = np.random.default_rng(4)
    n, m = 40, 25
    O = rng.standard_normal((n, m))
    mean = O.mean(axis=0)
    S = sr.SRSMatrix(list(O), mean, 1, 0.05)
    dense = (O - mean).T @ (O - mean) / n + 0.05 * np.eye(m)
    v = rng.standard_normal(m)
    assert np.max(np.abs(S * v - dense @ v)) < 1e-12
    g = rng.standard_normal(m)
    x, res, it = sr.conjugate_gradient(lambda y: S * y, g, np.zeros(m), 200, 1e-12)
    assert np.max(np.abs(x - np.linalg.solve(dense, g))) < 1e-8


@pytest.mark.parametrize("scheme", ["Variational2Site", "Variational1Site"])
def test_k1_ising_all_21_routes_variational(scheme):
    """test_bmps_contractor.cpp:472-486: the same 21 routes with Variational2Site(10,30,1e-15,1e-14,10)
    and Variational1Site(10,30,1e-15,1e-14,10) (bmps_impl.h:864-1172), tolerance 1e-8."""
    import k1_routes
    tn, lognorm, beta = ising.build_ising_tn(12, 12)
    f_ex = ising.exact_free_energy(12, 12, 1.0 / beta)
    c = BMPSContractor(12, 12)
    c.Init(tn)
    c.SetTruncateParams(getattr(BMPSTruncateParams, scheme)(10, 30, 1e-15, 1e-14, 10))
    amps = k1_routes.run_oracle(c, tn)
    assert len(amps) == k1_routes.N_AMPS
    for a in amps:
        assert abs(-(np.log(a) + lognorm) / 144 / beta - f_ex) < 1e-8


def _mps_to_dense(tensors):
    v = tensors[0]
    for t in tensors[1:]:
        v = np.tensordot(v, t, axes=([v.ndim - 1], [0]))
    return v.reshape(-1)


@pytest.mark.parametrize("pos", [0, 1, 2, 3])
def test_variational_compression_is_at_least_as_close_as_its_initial_guess(pos):
    """Property of MultiplyMPO{2,1}SiteVariationalCompress_: the alternating sweeps can only lower the distance
    to the exact product BMPS x MPO, starting from MakeVariationalInitGuess_ (bmps_impl.h:1174-1212); at
    D_max >= the exact bond dimension all three schemes return the exact product."""
    from oracle import bmps as ob
    from peps_amd import synthetic
    L, D = 5, 3
    sitps = synthetic.make_sitps(L, D, noise=0.5)
    cfg = synthetic.make_configs(L, 1, "heisenberg")[0]
    from oracle.contractor import TensorNetwork2D
    tn = TensorNetwork2D.from_sitps(sitps, cfg)
    c = BMPSContractor(L, L)
    c.Init(tn)
    c.SetTruncateParams(BMPSTruncateParams.SVD(D * D, D * D, 0.0))
    c.GrowBMPSStep(tn, pos)                                   # exact first row: bond D
    mps = c.bmps_set[pos][-1]
    if pos in (1, 3):
        num = 1 if pos == 3 else L - 2
        mpo = [tn((num, k)) for k in range(L)]
    else:
        num = 1 if pos == 0 else L - 2
        mpo = [tn((k, num)) for k in range(L)]
    exact = mps.multiply_mpo(mpo, ob.SVD_COMPRESS, D * D, D * D, 0.0)
    ex = _mps_to_dense(exact.tensors)
    chi = 4
    rev = list(reversed(mpo)) if pos > 1 else list(mpo)
    init = mps._variational_init_guess(rev, chi, chi, 0.0)
    d_init = np.linalg.norm(_mps_to_dense(init.tensors) - ex)
    svd = mps.multiply_mpo(mpo, ob.SVD_COMPRESS, chi, chi, 0.0)
    d_svd = np.linalg.norm(_mps_to_dense(svd.tensors) - ex)
    for scheme in (ob.VARIATION2Site, ob.VARIATION1Site):
        var = mps.multiply_mpo(mpo, scheme, chi, chi, 0.0, 1e-12, 20)
        d_var = np.linalg.norm(_mps_to_dense(var.tensors) - ex)
        assert d_var <= d_init * (1 + 1e-9)
        assert d_var <= d_svd * (1 + 1e-6)                    # the variational optimum is no worse than the SVD sweep
        full = mps.multiply_mpo(mpo, scheme, D * D, D * D, 0.0, 1e-12, 5)
        assert np.linalg.norm(_mps_to_dense(full.tensors) - ex) < 1e-10 * np.linalg.norm(ex)


# golden signatures of the exact-summation GRADIENT that the reference's tests hold (deterministic ones: NormSquare and
# WeightedProbeInnerProduct, test_exact_summation_evaluator.cpp:50-71; the "random probe" ones need TensorToolkit's RNG)
GRAD_SIGNATURES = [
    ("heisenberg_tps_doublelowest", "xxz", 2.69115141087757e-08, 8.719330571244627e-09, "test_exact_summation_evaluator.cpp:575-576"),
    ("transverse_ising_tps_doublelowest", "tfim", 1.290630314256308e-10, 4.081475798300479e-11, "test_exact_summation_evaluator.cpp:744-745"),
]


def gradient_signatures(grad):
    """(NormSquare, WeightedProbeInnerProduct) of a SITPS-shaped gradient grad[r][c][i]: the probe multiplies component i of
    site (r, c) by 0.012 ((r + 1) 11 + (c + 1) 5 + (i + 1) 2), so x * probe = sum of base times the component's norm^2."""
    ns = wp = 0.0
    for r, row in enumerate(grad):
        for c, comps in enumerate(row):
            for i, t in enumerate(comps):
                n2 = float(np.sum(np.abs(np.asarray(t)) ** 2))
                ns += n2
                wp += 0.012 * ((r + 1) * 11 + (c + 1) * 5 + (i + 1) * 2) * n2
    return ns, wp


@pytest.mark.parametrize("name,model,norm2,probe,cite", GRAD_SIGNATURES)
def test_reference_gradient_signatures(fixtures_dir, name, model, norm2, probe, cite):
    """Row a17: the gradient <E* O*> - E* <O*> of the exact-sum evaluator reproduces the 16-digit golden values the
    reference prints for its own fixtures (its assertion tolerance is an absolute 1e-8; here 1e-9 relative)."""
    s = qlten_io.load_sitps(os.path.join(fixtures_dir, name))
    tp = BMPSTruncateParams.SVD(1, 8, 1e-16)
    if model == "xxz":
        m, configs = vmc.SquareSpinOneHalfXXZModelOBC(), vmc.generate_all_permutation_configs([2, 2], 2, 2)
    else:
        m, configs = vmc.TransverseFieldIsingSquareOBC(1.0), vmc.all_product_configs(2, 2, 2)
    e, grad, w = vmc.exact_sum_energy_evaluator(s, configs, tp, m)
    ns, wp = gradient_signatures(grad)
    assert abs(ns / norm2 - 1) < 1e-9, cite
    assert abs(wp / probe - 1) < 1e-9, cite


def test_complex_conjugate_gradient_host_and_oracle_agree():
    """ConjugateGradientSolver for TenElemT = QLTEN_Complex (utility/conjugate_gradient_solver.h:142-156, 181-276): a * b = sum conj(a) b,
    pap valid when Re > 0 and |Im| < 1e-10, the restart test on the real part.  The host-vector solver of the multi-rank path
    (peps_amd/sr.py) and the oracle's restatement walk the same iterates on a Hermitian positive-definite system; both solve it."""
    from oracle import sr as osr
    from peps_amd import sr
    rng = np.random.default_rng(5)
    m = 40
    a = rng.standard_normal((m, m)) + 1j * rng.standard_normal((m, m))
    h = a.conj().T @ a / m + 0.05 * np.eye(m)

    class Mat:
        def __mul__(self, v):
            return h @ v

    b = rng.standard_normal(m) + 1j * rng.standard_normal(m)
    xo, ro, io, wo = osr.conjugate_gradient_full(lambda v: h @ v, b, np.zeros(m, dtype=np.complex128), 300, 1e-10, 0.0, 7, 0.5)
    xh, rh, ih, wh = sr.conjugate_gradient(Mat(), b, None, 300, 1e-10, 0.0, 7, 0.5, full_output=True)
    assert wo == wh == osr.K_CONVERGED and io == ih and np.allclose(xo, xh, rtol=0, atol=1e-12)
    assert np.linalg.norm(h @ xh - b) <= 1e-10 * np.linalg.norm(b) * 1.01
    assert np.max(np.abs(xh - np.linalg.solve(h, b))) < 1e-8
    # a matrix that is not self-adjoint: p * (A p) gets an imaginary part -> the indefinite-matrix exit, in both
    g = h + 0.3j * np.triu(np.ones((m, m)), 1)
    _, _, io2, wo2 = osr.conjugate_gradient_full(lambda v: g @ v, b, np.zeros(m, dtype=np.complex128), 300, 1e-10, 0.0, 7, 0.5)

    class Mat2:
        def __mul__(self, v):
            return g @ v

    _, _, ih2, wh2 = sr.conjugate_gradient(Mat2(), b, None, 300, 1e-10, 0.0, 7, 0.5, full_output=True)
    assert wo2 == wh2 == osr.K_INDEFINITE and io2 == ih2


def test_k5_oracle_against_a_dense_contraction(fixtures_dir):
    """The oracle's boundary-MPS path at an exact chi (64 = D^2 on the 4x4 lattice) against an INDEPENDENT dense contraction of the
    reference's 4x4 D = 8 Heisenberg fixture (tests/golden/k5_complex_dense.json, scripts/make_k5_complex_golden.py: all 65 536 amplitudes by
    plain tensordot, no boundary MPS): amplitude and local energy of seeded Sz = 0 configurations, for the double fixture and for its
    QLTEN_Complex twin (the complex-typed file holds the same state: Im psi = 0 to the last bit); the exact energy of the Sz = 0 sector
    is the number the survey quotes, -9.1891559611."""
    import json
    import os
    from oracle import qlten_io, vmc
    from oracle.bmps import BMPSTruncateParams
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "k5_complex_dense.json")))
    assert abs(gold["energy_sz0_sector"][0] - (-9.1891559611)) < 1e-9 and gold["energy_sz0_sector"][1] == 0.0
    tp = BMPSTruncateParams.SVD(64, 64, 0.0)
    model = vmc.SquareSpinOneHalfXXZModelOBC(1.0, 1.0, 0.0)
    for name, cplx in (("tps_square_heisenberg4x4D8Double", False), ("tps_square_heisenberg4x4D8Complex", True)):
        s = qlten_io.load_sitps(os.path.join(fixtures_dir, name), complex_data=cplx)
        ratios = []
        for k in (0, 7, 19):
            cfg = np.array(gold["configs"][k])
            comp = vmc.TPSWaveFunctionComponent(s, cfg, tp)
            ratios.append(comp.amplitude / complex(*gold["amplitude"][k]))
            e = model.CalEnergyAndHoles(s, comp, False)[0]
            assert abs(e - complex(*gold["e_loc"][k])) < 1e-8
        # the golden was taken from the complex-typed file; the double file holds the same state at another overall scale (1.7716...)
        assert np.max(np.abs(np.array(ratios) / ratios[0] - 1)) < 1e-10
        assert abs(ratios[0] - 1) < 1e-10 if cplx else abs(ratios[0] - 1.77163865188) < 1e-9


def test_k2_complex_variational_one_site_oracle():
    """The reference's complex variational run (test_bmps_contractor.cpp:663-673) through the ORACLE: the 24 x 10 Ising network in the Z2
    (Hadamard) basis of the bonds with a random phase on every tensor, Variational1Site(1, 10, 1e-15, 1e-14, 10) -- every route of the
    reference's Contract2DTNUsingBMPSContractor gives the exact free energy to 1e-8 with a vanishing imaginary part.  This pins the
    conjugations of the oracle's variational restatement (oracle/bmps.py: Dag() of the environments) that the device test of the truncating
    complex runs (tests/test_gpu_complex.py) leans on."""
    import k1_routes
    rows, cols = 24, 10
    tn, lognorm, beta = ising.build_ising_tn(cols, rows)
    f_ex = ising.exact_free_energy(cols, rows, 1.0 / beta)
    H = np.array([[1.0, 1.0], [1.0, -1.0]]) / np.sqrt(2.0)
    rng = np.random.default_rng(11)
    ph = rng.uniform(size=(rows, cols))

    def site(rc):
        t = tn(rc)
        for ax in range(4):
            if t.shape[ax] == 2:
                t = np.moveaxis(np.tensordot(H, t, axes=([1], [ax])), 0, ax)
        t = t.astype(np.complex128) * np.exp(2j * np.pi * ph[rc])
        return t * np.exp(-2j * np.pi * ph.sum()) if rc == (0, 0) else t

    class Tn:
        def __init__(self):
            self.rows, self.cols = rows, cols

        def __call__(self, rc):
            return site(rc)
    ztn = TensorNetwork2D.from_sitps([[[site((r, c))] for c in range(cols)] for r in range(rows)], np.zeros((rows, cols), dtype=int))
    c = BMPSContractor(rows, cols)
    c.Init(ztn)
    c.SetTruncateParams(BMPSTruncateParams.Variational1Site(1, 10, 1e-15, 1e-14, 10))
    amps = k1_routes.run_oracle(c, ztn, rows)
    assert len(amps) > 10
    for a in amps:
        z = complex(a)
        assert abs(-(np.log(z.real) + lognorm) / (rows * cols) / beta - f_ex) < 1e-8
        assert abs(z.imag) < 1e-10 * abs(z.real)
