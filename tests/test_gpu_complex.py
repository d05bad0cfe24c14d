"""Complex element type (PEPSGPU_C128 = QLTEN_Complex of the reference, whose hot-path tests are all compiled for double AND
complex: tests/CMakeLists.txt:57-100) through the C ABI:
  * amplitudes, replacement traces and holes of random complex states against the float64-complex oracle,
  * K1-complex: the 12x12 critical Ising network with a random phase on every site tensor (test_bmps_contractor.cpp:244-262,
    472-493): all 21 routes give the exact free energy to 1e-8 and a vanishing imaginary part,
  * K2: the same physics on the 24 x 10 lattice in the Z2 (Hadamard) basis of the bonds with random phases, SVD(1, 10, 1e-15)
    (test_bmps_contractor.cpp:499-686),
  * K4-complex: the reference's 2x2 complex Heisenberg fixture, exact-summation energy -1.99521278793
    (test_exact_summation_evaluator.cpp:606) from device amplitudes, with the device's complex replace-trace ratios."""
import os

import numpy as np
import pytest

from oracle import ising, qlten_io, vmc
from oracle.bmps import BMPSTruncateParams, LEFT, DOWN, RIGHT, UP, HORIZONTAL, VERTICAL
from peps_amd import synthetic

pytestmark = pytest.mark.gpu


def _complex_sitps(L, D, seed, rows=None):
    rows = rows or L
    rng = np.random.default_rng(seed)
    out = []
    for r in range(rows):
        row = []
        for c in range(L):
            shp = (1 if c == 0 else D, 1 if r == rows - 1 else D, 1 if c == L - 1 else D, 1 if r == 0 else D)
            row.append([(rng.uniform(0.2, 1.0, shp) * np.exp(2j * np.pi * rng.uniform(size=shp))
                         + 0.5 * np.exp(2j * np.pi * rng.uniform())) / D for _ in range(2)])
        out.append(row)
    return out


def _flat(sitps, D):
    rows, cols, d = len(sitps), len(sitps[0]), len(sitps[0][0])
    flat = np.zeros((rows, cols, d, D, D, D, D), dtype=np.complex128)
    for r in range(rows):
        for c in range(cols):
            for s in range(d):
                t = sitps[r][c][s]
                flat[r, c, s, :t.shape[0], :t.shape[1], :t.shape[2], :t.shape[3]] = t
    return flat


@pytest.mark.parametrize("L,D,chi", [(4, 2, 4), (5, 3, 6), (6, 3, 9)])
def test_complex_amplitude_traces_and_holes_vs_oracle(L, D, chi):
    from peps_amd import capi
    sitps = _complex_sitps(L, D, 17 * L + D)
    cfgs = synthetic.make_configs(L, 3, "heisenberg", seed0=5)
    tp = BMPSTruncateParams.SVD(chi, chi, 0.0)
    ctx = capi.Context(L, L, D, 2, chi, dtype=capi.C128, max_walkers=len(cfgs))
    ctx.state_upload(_flat(sitps, D))
    ctx.set_configs(cfgs)
    amps = ctx.evaluate_amplitude()
    assert amps.dtype == np.complex128
    comps = [vmc.TPSWaveFunctionComponent(sitps, c, tp) for c in cfgs]
    ref = np.array([c.amplitude for c in comps])
    assert np.max(np.abs(amps / ref - 1)) < 1e-9, (amps, ref)
    # replacement traces along row 0 and hole . site == psi
    cand = np.array([[(0, 1), (1, 0), (1, 1)]] * len(cfgs), dtype=np.int32)
    rt = ctx.replace_nn_trace(0, 0, HORIZONTAL, cand)
    for w, comp in enumerate(comps):
        for k, (sa, sb) in enumerate(cand[w]):
            r = comp.contractor.ReplaceNNSiteTrace(comp.tn, (0, 0), (0, 1), HORIZONTAL, sitps[0][0][sa], sitps[0][1][sb])
            assert abs(rt[w, k] - r) < 1e-9 * abs(ref[w]), (w, k, rt[w, k], r)
    ctx.grow_full_bten(RIGHT, 0, 1, True)            # right environment of (0, 0) = every other site of row 0
    hole = ctx.punch_hole(0, 0, HORIZONTAL)
    flat = _flat(sitps, D)
    for w in range(len(cfgs)):
        assert abs(np.sum(hole[w] * flat[0, 0, cfgs[w, 0, 0]]) / ref[w] - 1) < 1e-9
    # the column route (LEFT / RIGHT stacks, reversed storage of RIGHT) against the oracle's column route
    ctx.set_configs(cfgs)
    ctx.grow_bmps_for_col(0)
    ctx.init_bten(UP, 0)
    ctx.grow_full_bten(DOWN, 0, 2, True)
    col = ctx.trace(0, 0, VERTICAL)
    for w, c in enumerate(cfgs):
        comp = vmc.TPSWaveFunctionComponent(sitps, c, tp)
        k, tn = comp.contractor, comp.tn
        k.GrowBMPSForCol(tn, 0); k.InitBTen(tn, UP, 0); k.GrowFullBTen(tn, DOWN, 0, 2, True)
        assert abs(col[w] / k.Trace(tn, (0, 0), VERTICAL) - 1) < 1e-9
    ctx.close()


def _phased(tn, rows, cols, seed):
    """every site tensor times a random phase, the total phase taken off site (0, 0) (test_bmps_contractor.cpp:247-258)"""
    rng = np.random.default_rng(seed)
    ph = rng.uniform(size=(rows, cols))
    out = [[[tn((r, c)).astype(np.complex128) * np.exp(2j * np.pi * ph[r, c])] for c in range(cols)] for r in range(rows)]
    out[0][0][0] = out[0][0][0] * np.exp(-2j * np.pi * ph.sum())
    return out


def test_k1_complex_random_phase_network_all_21_routes():
    import k1_routes
    from peps_amd import capi
    tn, lognorm, beta = ising.build_ising_tn(12, 12)
    f_ex = ising.exact_free_energy(12, 12, 1.0 / beta)
    sitps = _phased(tn, 12, 12, 3)
    ctx = capi.Context(12, 12, 2, 1, 30, dtype=capi.C128, max_walkers=1, chi_min=10, trunc_err=1e-15)
    ctx.state_upload(_flat(sitps, 2))
    ctx.set_configs(np.zeros((1, 12, 12), dtype=np.int32))
    amps = k1_routes.run_device(ctx)
    assert len(amps) == k1_routes.N_AMPS
    for a in amps:
        z = complex(a[0])
        assert abs(-(np.log(z.real) + lognorm) / 144 / beta - f_ex) < 1e-8
        assert abs(z.imag) < 1e-10 * abs(z.real)
    ctx.close()


def test_k2_z2_basis_24x10_complex_and_real():
    """K2: rows = 24, cols = 10, bonds in the Z2 (even / odd) basis -- the Hadamard transform of the K1 tensors, which
    is what the reference's Z2-symmetric construction spans -- real and with random phases, SVD(1, 10, 1e-15)."""
    import k1_routes
    from peps_amd import capi
    rows, cols = 24, 10
    tn, lognorm, beta = ising.build_ising_tn(cols, rows)
    f_ex = ising.exact_free_energy(cols, rows, 1.0 / beta)
    H = np.array([[1.0, 1.0], [1.0, -1.0]]) / np.sqrt(2.0)

    class Tn:      # every bond leg of dimension 2 rotated by H (H H = 1: the network value is unchanged)
        def __call__(self, rc):
            t = tn(rc)
            for ax in range(4):
                if t.shape[ax] == 2:
                    t = np.moveaxis(np.tensordot(H, t, axes=([1], [ax])), 0, ax)
            return t
    z2 = Tn()
    t_bulk = z2((5, 5))
    par = np.indices(t_bulk.shape).sum(axis=0) % 2
    assert np.max(np.abs(t_bulk[par == 1])) < 1e-14          # Z2 block structure: odd total parity vanishes
    for cplx_mode in (False, True):
        if cplx_mode:
            sit = _phased(z2, rows, cols, 11)
            flat, dt = _flat(sit, 2), capi.C128
        else:
            sit = [[[z2((r, c))] for c in range(cols)] for r in range(rows)]
            flat, dt = _flat(sit, 2).real.copy(), capi.F64
        ctx = capi.Context(rows, cols, 2, 1, 10, dtype=dt, max_walkers=1, chi_min=1, trunc_err=1e-15)
        ctx.state_upload(flat)
        ctx.set_configs(np.zeros((1, rows, cols), dtype=np.int32))
        amps = k1_routes.run_device(ctx, rows)
        for a in amps:
            z = complex(a[0])
            assert abs(-(np.log(z.real) + lognorm) / (rows * cols) / beta - f_ex) < 1e-8
            assert abs(z.imag) < 1e-10 * abs(z.real)
        ctx.close()


def test_k4_complex_fixture_exact_sum_energy(fixtures_dir):
    from peps_amd import capi
    s = qlten_io.load_sitps(os.path.join(fixtures_dir, "heisenberg_tps_complex_from_simple_update"), complex_data=True)
    D = max(max(t.shape) for row in s for comps in row for t in comps)
    flat = _flat(s, D)
    cfgs = np.array(vmc.generate_all_permutation_configs([2, 2], 2, 2), dtype=np.int32).reshape(-1, 2, 2)
    ctx = capi.Context(2, 2, D, 2, 8, dtype=capi.C128, max_walkers=len(cfgs))
    ctx.state_upload(flat)
    ctx.set_configs(cfgs)
    psi = ctx.evaluate_amplitude()
    index = {tuple(c.ravel()): i for i, c in enumerate(cfgs)}
    bonds = [((0, 0), (0, 1)), ((1, 0), (1, 1)), ((0, 0), (1, 0)), ((0, 1), (1, 1))]
    num = 0.0
    for i, c in enumerate(cfgs):
        hpsi = 0.0
        for (a, b) in bonds:
            if c[a] == c[b]:
                hpsi += 0.25 * psi[i]
            else:
                c2 = c.copy(); c2[a], c2[b] = c[b], c[a]
                hpsi += -0.25 * psi[i] + 0.5 * psi[index[tuple(c2.ravel())]]
        num += np.conj(psi[i]) * hpsi
    energy = num / np.sum(np.abs(psi) ** 2)
    assert abs(energy.real - (-1.99521278793)) < 1e-9 and abs(energy.imag) < 1e-12
    # the ratios the reference's solver uses (ReplaceNNSiteTrace / psi, square_spin_onehalf_xxz_obc.h:72-104) in complex
    ctx.set_configs(cfgs)
    ctx.grow_bmps_for_row(0)
    ctx.grow_full_bten(RIGHT, 0, 2, True)
    ctx.init_bten(LEFT, 0)
    cand = np.stack([np.array([[c[0, 1], c[0, 0]]], dtype=np.int32) for c in cfgs])
    rt = ctx.replace_nn_trace(0, 0, HORIZONTAL, cand)[:, 0]
    for i, c in enumerate(cfgs):
        c2 = c.copy(); c2[0, 0], c2[0, 1] = c[0, 1], c[0, 0]
        assert abs(rt[i] - psi[index[tuple(c2.ravel())]]) < 1e-10 * np.max(np.abs(psi))
    ctx.close()


# ---- TenElemT = QLTEN_Complex at the C++ HOST level (qlpeps_gpu.h templates, libpepshost.so) ------------------------------
def _oracle_exact_sum(sitps, cfgs, chi, model):
    tp = BMPSTruncateParams.SVD(chi, chi, 0.0)
    e, grad, w = vmc.exact_sum_energy_evaluator(sitps, [np.asarray(c) for c in cfgs], tp, model)
    return e, grad


def test_k4_complex_exact_sum_energy_and_gradient_through_host_layer(fixtures_dir):
    """The reference's 2x2 complex Heisenberg fixture through ExactSumEnergyEvaluator of the C++ host layer instantiated for
    QLTEN_Complex (SplitIndexTPST / BMPSContractorT / TPSWaveFunctionComponentT / GradAccumulatorT <std::complex<double>>,
    holes resident in HBM, pepsgpu_grad_accumulate on a PEPSGPU_C128 context): the known energy -1.99521278793
    (test_exact_summation_evaluator.cpp:606), and the GRADIENT (S_EO - E* S_O) / W with the conjugations of
    exact_summation_energy_evaluator.h:228-295 element-wise against the complex oracle to 1e-9; 2-rank partition too."""
    from peps_amd import hostapi
    d = os.path.join(fixtures_dir, "heisenberg_tps_complex_from_simple_update")
    s = qlten_io.load_sitps(d, complex_data=True)
    D = max(max(t.shape) for row in s for comps in row for t in comps)
    flat = hostapi.load_sitps_complex(d, D)
    assert flat.dtype == np.complex128 and np.array_equal(flat, _flat(s, D))          # SplitIndexTPS<QLTEN_Complex>::Load
    cfgs = np.array(vmc.generate_all_permutation_configs([2, 2], 2, 2), dtype=np.int32).reshape(-1, 2, 2)
    e_ref, g_ref = _oracle_exact_sum(s, cfgs, 8, vmc.SquareSpinOneHalfXXZModelOBC())
    for size in (1, 2):
        e, grad = hostapi.exact_sum_complex(flat, cfgs, 8, "xxz", (1.0, 1.0, 0.0), size=size, batch=4)
        assert abs(e.real - (-1.99521278793)) < 1e-9 and abs(e.imag) < 1e-11
        assert abs(e - e_ref) < 1e-10
        gmax = max(np.max(np.abs(t)) for row in g_ref for comps in row for t in comps)
        assert gmax > 1e-6                                                                # a simple-update state: not at the minimum
        for r in range(2):
            for c in range(2):
                for i in range(2):
                    t = g_ref[r][c][i]
                    got = grad[r, c, i][:t.shape[0], :t.shape[1], :t.shape[2], :t.shape[3]]
                    assert np.max(np.abs(got - t)) < 1e-9 * gmax, (size, r, c, i)


@pytest.mark.parametrize("model", ["xxz", "tfim"])
def test_complex_cal_energy_and_holes_vs_oracle(model):
    """CalEnergyAndHoles (XXZ: square_spin_onehalf_xxz_obc.h:72-104 with ComplexConjugate(psi_ex * inv_psi); TFIM:
    transverse_field_ising_square_obc.h:195-203) of the host layer on a random COMPLEX 4x4 state: amplitudes, complex local
    energies, Dag(hole) tensors and the psi list against the complex oracle (1e-9); hole . site == psi."""
    from peps_amd import hostapi
    L, D, chi = 4, 3, 9
    sitps = _complex_sitps(L, D, 91)
    flat = _flat(sitps, D)
    cfgs = synthetic.make_configs(L, 4, "heisenberg" if model == "xxz" else "tfim", seed0=3)
    tp = BMPSTruncateParams.SVD(chi, chi, 0.0)
    m = vmc.SquareSpinOneHalfXXZModelOBC(0.7, 1.3, 0.0) if model == "xxz" else vmc.TransverseFieldIsingSquareOBC(0.8)
    params = (0.7, 1.3, 0.0) if model == "xxz" else (0.8,)
    amps, en, holes, psi = hostapi.energy_and_holes_complex(flat, cfgs, chi, model, params, True)
    assert en.dtype == np.complex128 and np.max(np.abs(en.imag)) > 1e-6          # a complex state has complex local energies
    for w, c in enumerate(cfgs):
        comp = vmc.TPSWaveFunctionComponent(sitps, c, tp)
        e, h, psi_list = m.CalEnergyAndHoles(sitps, comp, True)
        assert abs(amps[w] / comp.amplitude - 1) < 1e-9
        assert abs(en[w] - e) < 1e-9 * max(1.0, abs(e))
        for r in range(L):
            for cc in range(L):
                t = h[r][cc]
                got = holes[w, r, cc][:t.shape[0], :t.shape[1], :t.shape[2], :t.shape[3]]
                assert np.max(np.abs(got - t)) < 1e-9 * max(1.0, np.max(np.abs(t)))
        # Dag(hole) . conj(site) = conj(psi)
        t00 = holes[w, 1, 2] * np.conj(flat[1, 2, c[1, 2]])
        assert abs(np.sum(t00) / np.conj(amps[w]) - 1) < 1e-9
        assert np.max(np.abs(psi[:, w] / np.array(psi_list) - 1)) < 1e-9


def test_complex_mc_gradient_sample_vs_oracle_accumulation():
    """MCEnergyGradEvaluator accumulation on a complex state (mc_energy_grad_evaluator.h:245-298: O* = conj(1 / psi) Dag(hole),
    E_loc^* O*, grad = <E* O*> - E* <O*>) through the host layer with the holes resident in HBM: zero sweeps between samples
    would repeat one configuration, so the chain is advanced on the device (exchange updater, |psi'/psi|^2 acceptance) and the
    accumulators of the visited configurations are rebuilt with the complex oracle."""
    from peps_amd import hostapi
    L, D, chi = 4, 2, 4
    sitps = _complex_sitps(L, D, 7)
    flat = _flat(sitps, D)
    cfgs = synthetic.make_configs(L, 6, "heisenberg", seed0=11)
    seeds = np.arange(6, dtype=np.uint64) + 5
    # one sample: sweep, then accumulate -- the configurations after the sweep come back
    e, grad, out_cfg, acc = hostapi.mc_energy_grad_complex(flat, cfgs, seeds, chi, "exchange", "xxz", (1.0, 1.0, 0.0), 0, 1)
    assert np.all(out_cfg.sum(axis=(1, 2)) == cfgs.sum(axis=(1, 2))) and 0 < acc.mean() < 1
    tp = BMPSTruncateParams.SVD(chi, chi, 0.0)
    m = vmc.SquareSpinOneHalfXXZModelOBC()
    so = np.zeros(flat.shape, dtype=np.complex128); seo = np.zeros(flat.shape, dtype=np.complex128)
    es = []
    for c in out_cfg:
        comp = vmc.TPSWaveFunctionComponent(sitps, c, tp)
        el, holes, _ = m.CalEnergyAndHoles(sitps, comp, True)
        es.append(el)
        for r in range(L):
            for cc in range(L):
                t = np.conj(1.0 / comp.amplitude) * holes[r][cc]
                sl = (r, cc, int(c[r, cc])) + tuple(slice(0, k) for k in t.shape)
                so[sl] += t
                seo[sl] += np.conj(el) * t
    e_ref = np.mean(es)
    g_ref = seo / len(out_cfg) - np.conj(e_ref) * so / len(out_cfg)
    assert abs(e - e_ref) < 1e-9 * abs(e_ref)
    assert np.max(np.abs(grad - g_ref)) < 1e-9 * np.max(np.abs(g_ref))


# ---- round 5: measurement solvers, MCPEPSMeasurer / ExactSumMeasurer and the fermionic models for QLTEN_Complex ----
def _bond(sitps):
    return max(max(t.shape) for row in sitps for site in row for t in site)


@pytest.mark.parametrize("gold_file,stem,model,params,cfgs_kind", [
    ("k4_heisenberg_exact_sum_measurer.json", "heisenberg_tps", "xxz", (1.0, 1.0, 0.0), "half"),
    ("k4_tfim_exact_sum_measurer.json", "transverse_ising_tps", "tfim", (1.0,), "all")])
def test_k4_complex_exact_sum_measurer_registries_on_the_device(fixtures_dir, gold_file, stem, model, params, cfgs_kind):
    """The QLTEN_Complex build of the reference's ExactSumMeasurerMPI tests (tests/test_algorithm/test_exact_summation_measurer.cpp:443-545,
    :585-651; tests/CMakeLists.txt:385-398) through ExactSumMeasurer<.., QLTEN_Complex> of the host layer: every registry key of the complex
    2x2 simple-update states at the reference's 1e-10 with imaginary parts below 1e-10 (kTol / kImagTol), the closed forms of the `lowest`
    states, and a 3-rank decomposition that adds up to the serial result."""
    import json
    from peps_amd import hostapi
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", gold_file)))
    cfgs = None if cfgs_kind == "all" else np.array(vmc.generate_all_permutation_configs([2, 2], 2, 2), dtype=np.int32)
    hostapi.set_truncate_params(1, 1e-16, 0)                    # SVD(1, 8, 1e-16)
    try:
        s = qlten_io.load_sitps(os.path.join(fixtures_dir, stem + "_complex_from_simple_update"), complex_data=True)
        flat = _flat(s, _bond(s))
        acc, w = hostapi.exact_sum_measure_partial(flat, cfgs, 8, model, params, 0, 1, 16)
        assert set(acc) == set(gold["observables"])
        for key, want in gold["observables"].items():
            v = acc[key] / w
            assert v.dtype == np.complex128 and np.max(np.abs(v - np.asarray(want))) < 1e-10, key
            assert np.max(np.abs(v.imag)) < 1e-10, key
        allc = np.array(vmc.generate_all_binary_configs(2, 2), dtype=np.int32) if cfgs is None else cfgs
        parts = [hostapi.exact_sum_measure_partial(flat, allc, 8, model, params, r, 3, 2) for r in range(3)]
        wp = sum(p[1] for p in parts)
        for key in acc:
            assert np.max(np.abs(sum(p[0][key] for p in parts) / wp - acc[key] / w)) < 1e-12, key
        s = qlten_io.load_sitps(os.path.join(fixtures_dir, stem + "_complexlowest"), complex_data=True)
        acc, w = hostapi.exact_sum_measure_partial(_flat(s, _bond(s)), cfgs, 8, model, params, 0, 1, 16)
        tol = {"energy": 6e-8, "spin_z": 5e-4}
        for key, want in gold["lowest"].items():
            assert np.max(np.abs(acc[key] / w - np.asarray(want))) < tol.get(key, 1e-5), key
    finally:
        hostapi.set_truncate_params()


@pytest.mark.parametrize("model,params", [("xxz", (1.0, 0.8, 0.3, 0.0, 0.0, 0.0, 0.0, 1.0)), ("j1j2", (1.0, 1.0, 0.5, 0.4, 0.0)), ("tfim", (0.9,))])
def test_complex_registry_observables_vs_oracle(model, params):
    """EvaluateObservables of the three measurement solvers on a random COMPLEX 4x4 state, fixed configurations: every key -- complex bond
    energies, the conjugated S+S- row channel (square_spin_onehalf_xxz_obc.h:47), the structure-factor amplitudes (unconjugated,
    structure_factor_measurement_mixin.h:160-194), sigma_x -- and the phase-aligned psi summary (psi_consistency.h:119-168) against the
    complex oracle."""
    from peps_amd import hostapi
    L, D, chi = 4, 3, 9
    sitps = _complex_sitps(L, D, 23)
    cfgs = synthetic.make_configs(L, 4, "heisenberg", seed0=3)
    tp = BMPSTruncateParams.SVD(chi, chi, 0.0)
    got, psi = hostapi.measure(_flat(sitps, D), cfgs, chi, model, params)
    some_imag = 0.0
    for w, cfg in enumerate(cfgs):
        comp = vmc.TPSWaveFunctionComponent(sitps, cfg, tp)
        if model == "tfim":
            ms = vmc.TransverseFieldIsingSquareOBC(params[0])
        else:
            om = vmc.SquareSpinOneHalfXXZModelOBC(*params[:3]) if model == "xxz" else vmc.SquareSpinOneHalfJ1J2XXZModelOBC(*params)
            ms = vmc.SquareNNNModelMeasurementSolver(om, structure_factor=(model == "xxz"))
        want = ms.EvaluateObservables(sitps, comp)
        assert set(got) == set(want)
        for key, v in want.items():
            v = np.asarray(v).ravel()
            assert got[key][w].shape == v.shape, key
            assert np.max(np.abs(got[key][w] - v)) < 1e-8 * max(1.0, np.max(np.abs(v))), (key, w)
        some_imag = max(some_imag, np.max(np.abs(got["energy"][w].imag)))
        pm, prel = ms.last_psi_summary
        assert abs(psi[0][w] / pm - 1) < 1e-8 and abs(psi[1][w] - prel) < 1e-8
    assert some_imag > 1e-6                                      # complex local estimators


def test_complex_mc_measurer_statistics():
    """MCPEPSMeasurer<.., QLTEN_Complex> (monte_carlo_peps_measurer_impl.h:172-258, :495-541): warm-up + NormalizeStateOrder1 + samples on a
    complex state; the chain is the device's own (|psi|^2 sampling with std::mt19937 per walker), the statistics are rebuilt here from the
    visited configurations with the complex oracle: mean over the walkers of the per-walker sample means, real standard errors."""
    from peps_amd import hostapi
    L, D, chi, n, nsamp = 4, 2, 4, 3, 2
    sitps = _complex_sitps(L, D, 41)
    flat = _flat(sitps, D)
    cfgs = synthetic.make_configs(L, n, "heisenberg", seed0=9).astype(np.int32)
    seeds = [5, 6, 7]
    # replay: the same seeds through the sweep entry point give the configurations after warm-up and after each sample
    c1, _, _ = hostapi.mc_sweeps_complex(flat, cfgs, seeds, chi, "exchange", 3)       # 1 warm-up + 2 samples
    run = cfgs.copy()
    out, _ = hostapi.measure(flat, run, chi, "xxz", (1.0, 1.0, 0.0), seeds=seeds, updater="exchange", warmup_sweeps=1, n_samples=nsamp,
                             sweeps_between_samples=1)
    assert np.array_equal(run, c1)                                # the measurer's chain == the plain chain of the same seeds
    mean, err = out["energy"]
    assert mean.dtype == np.complex128 and err.dtype == np.float64 and err[0] > 0.0
    # energy of the final configurations from the oracle brackets the mean within the spread of the samples
    tp = BMPSTruncateParams.SVD(chi, chi, 0.0)
    e_last = []
    for w in range(n):
        comp = vmc.TPSWaveFunctionComponent(sitps, run[w], tp)
        e_last.append(vmc.SquareNNNModelMeasurementSolver(vmc.SquareSpinOneHalfXXZModelOBC(1.0, 1.0, 0.0)).EvaluateObservables(sitps, comp)["energy"][0])
    c_mid, _, _ = hostapi.mc_sweeps_complex(flat, cfgs, seeds, chi, "exchange", 2)
    e_mid = []
    for w in range(n):
        comp = vmc.TPSWaveFunctionComponent(sitps, c_mid[w], tp)
        e_mid.append(vmc.SquareNNNModelMeasurementSolver(vmc.SquareSpinOneHalfXXZModelOBC(1.0, 1.0, 0.0)).EvaluateObservables(sitps, comp)["energy"][0])
    # the reference's statistics (oracle/statistics.py, pinned on test_statistics.cpp): per-walker AveListOfData over the samples, then
    # GatherStatisticListOfData across the walkers (a walker = an MPI rank of the reference)
    from oracle import statistics
    walker_means = np.array([statistics.ave_list_of_data([[e_mid[w]], [e_last[w]]]) for w in range(n)])
    want_mean, want_err = statistics.gather_statistic_list_of_data(walker_means)
    assert abs(mean[0] - want_mean[0]) < 1e-8
    assert abs(err[0] - want_err[0]) < 1e-8


FERMION_CASES = [("0.000000_complexlowest", 0.0, -2.0), ("0.000000_complex_from_simple_update", 0.0, -1.98218053854),
                 ("2.100000_complexlowest", 2.1, -4.2), ("2.100000_complex_from_simple_update", 2.1, -4.1879072654),
                 ("-2.500000_complexlowest", -2.5, -5.0), ("-2.500000_complex_from_simple_update", -2.5, -4.98966397657)]


@pytest.mark.parametrize("name,t2,e_ref", FERMION_CASES)
def test_k4_complex_spinless_fermion_energies(fixtures_dir, name, t2, e_ref):
    """The QLTEN_Complex build of the reference's fermionic exact-summation test (test_exact_summation_evaluator.cpp:268-470 with the
    `_complexlowest` / `_complex_from_simple_update` fixtures, :306-330): the six known energies from complex fZ2 tensors through (a) the
    Python flow over the C ABI (peps_amd/fermion.py: NN hops in the passes, diagonal hop against twisted BTen2 environments, conj(psi'/psi) as
    square_spinless_fermion.h:156,:210) and (b) SquareSpinlessFermion of the C++ host layer instantiated for std::complex<double>; both agree
    with the complex oracle per configuration."""
    import itertools
    from oracle import fermion as ofermion
    from peps_amd import capi, fermion, hostapi
    d = os.path.join(fixtures_dir, "spinless_fermion_tps_t2_" + name)
    st = fermion.FermionState.load(d, complex_data=True)
    assert st.is_complex
    cfgs = np.array([np.array(p).reshape(2, 2) for p in sorted(set(itertools.permutations([0, 0, 1, 1])))])
    ctx = capi.Context(2, 2, st.D, 4 * st.d, 8, dtype=capi.C128, max_walkers=len(cfgs))
    ctx.state_upload(st.extended_flat())
    amp = fermion.evaluate_amplitude(ctx, st, cfgs)
    e_loc, _ = fermion.spinless_fermion_energy(ctx, st, cfgs, 1.0, 0.0, t2)
    w = np.abs(amp) ** 2
    e = np.sum(w * e_loc) / np.sum(w)
    assert abs(e - e_ref) < 1e-9 and abs(e.imag) < 1e-10
    amps2, en2, _ = hostapi.fermion_energy(st, cfgs, 8, 1.0, 0.0, t2=t2)
    assert amps2.dtype == np.complex128
    rel = np.abs(amp) / np.max(np.abs(amp))        # (a ratio of two amplitudes of size 1e-8 of the largest carries 1e-8 of relative noise)
    assert np.max(np.abs(amps2 - amp)) < 1e-12 * np.max(np.abs(amp)) and np.max(np.abs(en2 - e_loc) * rel) < 1e-9
    fs = ofermion.FermionSITPS(ofermion.load_fermion_sitps(d, complex_data=True))
    tp = BMPSTruncateParams.SVD(8, 8, 0.0)
    model = ofermion.SquareSpinlessFermionOBC(1.0, t2, 0.0)
    for k, cfg in enumerate(cfgs):
        a = fs.amplitude(cfg, tp)
        assert abs(amp[k] - a) < 1e-10 * max(1.0, abs(a))
        if rel[k] < 1e-6:           # the two symmetry-forbidden configurations of the exact ground states: psi = O(1e-9), E_loc is noise
            continue
        assert abs(e_loc[k] - model.CalEnergy(fs, cfg, tp)[0]) < 1e-8


def test_k4_complex_fermion_measurer_registry_and_gradient(fixtures_dir):
    """test_exact_summation_measurer.cpp:205-257 (QLTEN_Complex branch): energy, charge and the per-bond energies of the complex 2x2
    simple-update state at 1e-10, imaginary parts below 1e-10; ExactSumEnergyEvaluator<.., QLTEN_Complex> on the same state (t2 = 0): the
    energy of :432-436 and a gradient that vanishes on the parity-forbidden entries."""
    import itertools
    import json
    from peps_amd import capi, fermion, hostapi
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "k4_exact_sum_measurer.json")))["observables"]
    st = fermion.FermionState.load(os.path.join(fixtures_dir, "spinless_fermion_tps_t2_0.000000_complex_from_simple_update"), complex_data=True)
    cfgs = np.array([np.array(p).reshape(2, 2) for p in sorted(set(itertools.permutations([0, 0, 1, 1])))])
    ctx = capi.Context(2, 2, st.D, 4 * st.d, 8, dtype=capi.C128, max_walkers=len(cfgs))
    ctx.state_upload(st.extended_flat())
    acc, w = fermion.exact_sum_measure(ctx, st, cfgs, 1.0, 0.0)
    assert set(acc) == set(gold)
    for key, want in gold.items():
        v = acc[key] / w
        assert np.max(np.abs(v - np.array(want))) < 1e-10 and np.max(np.abs(np.imag(v))) < 1e-10, key
    e, grad = hostapi.fermion_exact_sum(st, cfgs, 8, 1.0, 0.0, batch=4)
    assert abs(e - (-1.98218053854)) < 1e-9 and abs(e.imag) < 1e-10
    assert grad.dtype == np.complex128 and grad.shape[:3] == (2, 2, 2) and np.max(np.abs(grad)) > 1e-6
    for r in range(2):
        for c in range(2):
            pl, pd, pr, pu = st.par[r][c]
            tot = pl[:, None, None, None] + pd[None, :, None, None] + pr[None, None, :, None] + pu[None, None, None, :]
            for s in range(2):
                sl = (r, c, s) + tuple(slice(0, k) for k in tot.shape)
                assert np.all(grad[sl][(tot + st.nf[s]) % 2 == 1] == 0)


def test_complex_fermion_chain_and_energy_vs_oracle():
    """A 4x4 D = 4 fermionic state with a random phase on every parity-allowed element: MCUpdateSquareNNExchangeOBC chains of the host layer
    for QLTEN_Complex (acceptance from |psi'/psi|^2) keep the particle number and end on configurations whose complex amplitude and t-V local
    energy match the complex oracle; the measurer-style energy samples (one random stream over warm-up and samples) are the E_loc of the
    configurations the plain chain of the same seeds visits."""
    from oracle import fermion as ofermion
    from oracle.graded import GT
    from peps_amd import fermion, hostapi
    st = fermion.random_even_state(4, 4, 4, seed=3)
    rng = np.random.default_rng(12)
    for r in range(4):
        for c in range(4):
            st.tensors[r][c] = [t * np.exp(2j * np.pi * rng.uniform(size=t.shape)) for t in st.tensors[r][c]]
    assert st.is_complex
    gts = [[[GT(st.tensors[r][c][s][..., None], list(st.par[r][c]) + [np.array([int(st.nf[s])])], [-1, 1, 1, -1, -1])
             for s in range(st.d)] for c in range(4)] for r in range(4)]
    fs = ofermion.FermionSITPS(gts)
    cfgs = np.array([[(r + c + k) % 2 for c in range(4)] for k in range(3) for r in range(4)]).reshape(3, 4, 4)
    seeds = [11, 12, 13]
    out_cfg, amps, rates = hostapi.fermion_mc_sweeps(st, cfgs, seeds, 16, n_sweeps=2)
    assert np.all(out_cfg.sum(axis=(1, 2)) == cfgs.sum(axis=(1, 2))) and np.all(rates > 0) and not np.array_equal(out_cfg, cfgs)
    tp = BMPSTruncateParams.SVD(16, 16, 0.0)
    model = ofermion.SquareSpinlessFermionOBC(1.0, 0.0, 0.7)
    en, cfg2, _ = hostapi.fermion_measure_energy(st, cfgs, seeds, 16, 1, 1, 1, model="spinless", t=1.0, V=0.7)
    assert np.array_equal(cfg2, out_cfg) and en.dtype == np.complex128
    for w in range(3):
        a = fs.amplitude(out_cfg[w], tp)
        assert abs(amps[w] / a - 1) < 1e-8
        e, _ = model.CalEnergy(fs, out_cfg[w], tp)
        assert abs(en[0, w] - e) < 1e-7 * max(1.0, abs(e))
    assert np.max(np.abs(en.imag)) > 1e-6


# ---- round 5: the variational compression schemes for QLTEN_Complex (engine_var.h: conjugated operands = Dag() of the environments) ----
def _z2_network(rows, cols):
    tn, lognorm, beta = ising.build_ising_tn(cols, rows)
    H = np.array([[1.0, 1.0], [1.0, -1.0]]) / np.sqrt(2.0)

    def z2(rc):
        t = tn(rc)
        for ax in range(4):
            if t.shape[ax] == 2:
                t = np.moveaxis(np.tensordot(H, t, axes=([1], [ax])), 0, ax)
        return t
    return z2, lognorm, beta


def test_k2_complex_variational_one_site_reference_case():
    """The reference's complex variational run (test_bmps_contractor.cpp:663-673): the 24 x 10 Z2-basis Ising network with a random phase
    on every tensor, Variational1Site(1, 10, 1e-15, 1e-14, 10): every route gives the exact free energy to 1e-8, imaginary part zero."""
    import k1_routes
    from peps_amd import capi
    rows, cols = 24, 10
    z2, lognorm, beta = _z2_network(rows, cols)
    f_ex = ising.exact_free_energy(cols, rows, 1.0 / beta)
    sit = _phased(z2, rows, cols, 11)
    ctx = capi.Context(rows, cols, 2, 1, 10, dtype=capi.C128, max_walkers=1, chi_min=1, trunc_err=1e-15, scheme=capi.VARIATION1SITE,
                       convergence_tol=1e-14, iter_max=10)
    ctx.state_upload(_flat(sit, 2))
    ctx.set_configs(np.zeros((1, rows, cols), dtype=np.int32))
    amps = k1_routes.run_device(ctx, rows)
    for a in amps:
        z = complex(a[0])
        assert abs(-(np.log(z.real) + lognorm) / (rows * cols) / beta - f_ex) < 1e-8
        assert abs(z.imag) < 1e-10 * abs(z.real)
    ctx.close()


@pytest.mark.parametrize("scheme", ["Variational2Site", "Variational1Site"])
def test_k1_complex_variational_all_21_routes(scheme):
    """K1-complex (12 x 12 critical Ising, random phases) under both variational schemes with the parameters of the reference's real run
    (test_bmps_contractor.cpp:472-486: (10, 30, 1e-15, 1e-14, 10)): all 21 routes, 1e-8."""
    import k1_routes
    from peps_amd import capi
    tn, lognorm, beta = ising.build_ising_tn(12, 12)
    f_ex = ising.exact_free_energy(12, 12, 1.0 / beta)
    sitps = _phased(tn, 12, 12, 3)
    ctx = capi.Context(12, 12, 2, 1, 30, dtype=capi.C128, max_walkers=1, chi_min=10, trunc_err=1e-15,
                       scheme={"Variational2Site": capi.VARIATION2SITE, "Variational1Site": capi.VARIATION1SITE}[scheme],
                       convergence_tol=1e-14, iter_max=10)
    ctx.state_upload(_flat(sitps, 2))
    ctx.set_configs(np.zeros((1, 12, 12), dtype=np.int32))
    amps = k1_routes.run_device(ctx)
    assert len(amps) == k1_routes.N_AMPS
    for a in amps:
        z = complex(a[0])
        assert abs(-(np.log(z.real) + lognorm) / 144 / beta - f_ex) < 1e-8 and abs(z.imag) < 1e-10 * abs(z.real)
    ctx.close()


@pytest.mark.parametrize("scheme", ["Variational2Site", "Variational1Site"])
@pytest.mark.parametrize("L,D,chi", [(5, 3, 4), (6, 3, 5)])
def test_complex_variational_amplitudes_against_oracle(L, D, chi, scheme):
    """Truncating contraction of a random COMPLEX state: the device amplitudes follow the complex oracle run with the same scheme and
    parameters (the oracle conjugates where the reference takes Dag(), oracle/bmps.py) and differ from the SVD-compressed ones."""
    from peps_amd import capi
    sitps = _complex_sitps(L, D, 7 * L + D)
    cfgs = synthetic.make_configs(L, 4, "heisenberg", seed0=2)
    tp = getattr(BMPSTruncateParams, scheme)(chi, chi, 0.0, 1e-13, 30)
    ref = np.array([vmc.TPSWaveFunctionComponent(sitps, c, tp).amplitude for c in cfgs])
    svd = np.array([vmc.TPSWaveFunctionComponent(sitps, c, BMPSTruncateParams.SVD(chi, chi, 0.0)).amplitude for c in cfgs])
    ctx = capi.Context(L, L, D, 2, chi, dtype=capi.C128, max_walkers=len(cfgs), chi_min=chi, trunc_err=0.0,
                       scheme={"Variational2Site": capi.VARIATION2SITE, "Variational1Site": capi.VARIATION1SITE}[scheme],
                       convergence_tol=1e-13, iter_max=30)
    ctx.state_upload(_flat(sitps, D))
    ctx.set_configs(cfgs)
    got = ctx.evaluate_amplitude()
    assert np.all(ctx.walker_flags() == 0)
    err = np.max(np.abs(got / ref - 1))
    assert err < 1e-7, (got, ref, svd)
    assert np.max(np.abs(svd / ref - 1)) > 10 * err
    ctx.close()


@pytest.mark.parametrize("stem,model,params,e_su,e_lowest,kind", [
    ("transverse_ising_tps", "tfim", (1.0,), -5.19991995228, -5.226251859505506, "all"),
    ("heisenberg_tps", "xxz", (1.0, 1.0, 0.0), -1.99521278793, -2.0, "half")])
def test_k4_complex_exact_sum_energies_of_both_boson_models(fixtures_dir, stem, model, params, e_su, e_lowest, kind):
    """ExactSumEnergyEvaluator<.., QLTEN_Complex> of the host layer on the reference's complex 2x2 fixtures (test_exact_summation_evaluator.cpp,
    QLTEN_Complex build: TFIM :775 / :250-259, Heisenberg :606 / :139-174): simple-update energies to 1e-9, the closed forms of the `lowest`
    states to the reference's 6e-8, |Im E| < 1e-10, and a gradient that is small at the optimum (NormSquare < 1e-6; the reference pins 1.29e-10 +- 1e-8 for TFIM)."""
    from peps_amd import hostapi
    cfgs = np.array(vmc.generate_all_binary_configs(2, 2) if kind == "all" else vmc.generate_all_permutation_configs([2, 2], 2, 2), dtype=np.int32)
    hostapi.set_truncate_params(1, 1e-16, 0)                    # SVD(1, 8, 1e-16)
    try:
        for suffix, want, tol in (("_complex_from_simple_update", e_su, 1e-9), ("_complexlowest", e_lowest, 6e-8)):
            s = qlten_io.load_sitps(os.path.join(fixtures_dir, stem + suffix), complex_data=True)
            e, grad = hostapi.exact_sum_complex(_flat(s, _bond(s)), cfgs, 8, model, params, size=1, batch=16)
            assert abs(e - want) < tol and abs(e.imag) < 1e-10, (suffix, e)
            if suffix == "_complexlowest":
                assert np.sum(np.abs(grad) ** 2) < 1e-6
    finally:
        hostapi.set_truncate_params()


def test_k5_device_against_a_dense_contraction(fixtures_dir):
    """The reference's 4x4 D = 8 Heisenberg fixture in both element types (tests/slow_tests/test_data/tps_square_heisenberg4x4D8{Double,Complex},
    test_boson_mc_peps_measure.cpp:36,55-62) against an INDEPENDENT dense contraction (tests/golden/k5_complex_dense.json: 65 536 amplitudes by
    plain tensordot, no boundary MPS): amplitude and local energy of 32 seeded Sz = 0 configurations through CalEnergyAndHoles of the host layer at
    the exact chi = 64, float64 and complex128 device contexts, 1e-9."""
    import json
    from peps_amd import hostapi
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "k5_complex_dense.json")))
    cfgs = np.array(gold["configs"], dtype=np.int32)
    amp = np.array([complex(*z) for z in gold["amplitude"]])
    eloc = np.array([complex(*z) for z in gold["e_loc"]])
    s = qlten_io.load_sitps(os.path.join(fixtures_dir, "tps_square_heisenberg4x4D8Complex"), complex_data=True)
    a, e, _, psi = hostapi.energy_and_holes_complex(_flat(s, 8), cfgs, 64, "xxz", (1.0, 1.0, 0.0), holes=False)
    assert np.max(np.abs(a / amp - 1)) < 1e-9 and np.max(np.abs(e - eloc)) < 1e-8
    assert np.max(np.abs(psi / a[None, :] - 1)) < 1e-9                         # every route gives the same amplitude at an exact chi
    s = qlten_io.load_sitps(os.path.join(fixtures_dir, "tps_square_heisenberg4x4D8Double"))
    a, e, _, _ = hostapi.energy_and_holes(synthetic.sitps_to_flat(s, 8), cfgs, 64, "xxz", (1.0, 1.0, 0.0), holes=False)
    ratio = a / amp.real                                                       # (the double file holds the same state at another overall scale)
    assert np.max(np.abs(ratio / ratio[0] - 1)) < 1e-9 and abs(ratio[0] - 1.77163865188) < 1e-8
    assert np.max(np.abs(e - eloc.real)) < 1e-8
