"""GPU parity tests of the C++ host layer (peps_amd/host/qlpeps_gpu.h: updaters, solvers, evaluators)
driven through libpepshost.so, against the oracle on identical configuration lists / seeds."""
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


def _oracle_energy(sitps, cfgs, chi, model):
    tp = BMPSTruncateParams.SVD(chi, chi, 0.0)
    out = []
    for c in cfgs:
        comp = vmc.TPSWaveFunctionComponent(sitps, c, tp)
        e, holes, psis = model.CalEnergyAndHoles(sitps, comp, True)
        out.append((comp.amplitude, e, holes, psis))
    return out


@pytest.mark.parametrize("dt,tol", [(F64, 1e-9), (F32, 2e-5)])
def test_xxz_energy_and_holes_fixed_configs(fixtures_dir, dt, tol):
    """E_loc(S) and hole tensors on identical configurations: 4x4 D=8 reference fixture (K5), chi=16."""
    host = _host()
    s = qlten_io.load_sitps(os.path.join(fixtures_dir, "tps_square_heisenberg4x4D8Double"))
    cfgs = np.stack([synthetic.checkerboard(4)] + list(synthetic.make_configs(4, 5, "heisenberg")))
    ref = _oracle_energy(s, cfgs, 16, vmc.SquareSpinOneHalfXXZModelOBC())
    amps, en, holes, psi = host.energy_and_holes(synthetic.sitps_to_flat(s, 8), cfgs, 16, "xxz", (1.0, 1.0, 0.0), True, dt)
    for w, (a, e, h, ps) in enumerate(ref):
        assert abs(amps[w] / a - 1) < tol
        assert abs(en[w] - e) < tol * max(1.0, abs(e)) * 10
        for r in range(4):
            for c in range(4):
                hr = h[r][c]
                hd = holes[w, r, c][:hr.shape[0], :hr.shape[1], :hr.shape[2], :hr.shape[3]]
                assert np.max(np.abs(hd - hr)) < tol * 10 * np.max(np.abs(hr))
        assert np.max(np.abs(psi[:, w] / np.array(ps) - 1)) < tol * 10      # psi along every row and column


@pytest.mark.parametrize("dt,tol", [(F64, 1e-9), (F32, 2e-5)])
def test_tfim_energy_fixed_configs(dt, tol):
    host = _host()
    L, D, chi = 5, 3, 9
    s = synthetic.make_sitps(L, D)
    cfgs = synthetic.make_configs(L, 6, "tfim")
    ref = _oracle_energy(s, cfgs, chi, vmc.TransverseFieldIsingSquareOBC(0.7))
    amps, en, holes, psi = host.energy_and_holes(synthetic.sitps_to_flat(s, D), cfgs, chi, "tfim", (0.7,), True, dt)
    for w, (a, e, h, ps) in enumerate(ref):
        assert abs(amps[w] / a - 1) < tol
        assert abs(en[w] - e) < tol * max(1.0, abs(e)) * 10


K4 = [("heisenberg_tps_double_from_simple_update", "xxz", -1.99521278793, 1e-10),
      ("heisenberg_tps_doublelowest", "xxz", -2.0, 6e-8),
      ("transverse_ising_tps_double_from_simple_update", "tfim", -5.19991995228, 1e-10),
      ("transverse_ising_tps_doublelowest", "tfim",
       -2.0 * (np.sqrt(2 - 2 * np.cos(np.pi / 4)) + np.sqrt(2 - 2 * np.cos(3 * np.pi / 4))), 6e-8)]


@pytest.mark.parametrize("name,model,norm2,probe", [
    ("heisenberg_tps_doublelowest", "xxz", 2.69115141087757e-08, 8.719330571244627e-09),
    ("transverse_ising_tps_doublelowest", "tfim", 1.290630314256308e-10, 4.081475798300479e-11)])
def test_reference_gradient_signatures_on_device(fixtures_dir, name, model, norm2, probe):
    """The reference's golden gradient signatures (NormSquare, WeightedProbeInnerProduct: test_exact_summation_evaluator.cpp:
    50-71, 575-576, 744-745) from the DEVICE gradient (C++ ExactSumEnergyEvaluator, holes resident in HBM, f64)."""
    host = _host()
    s = qlten_io.load_sitps(os.path.join(fixtures_dir, name))
    flat = synthetic.sitps_to_flat(s, 4)
    if model == "xxz":
        cfgs, params = np.array(vmc.generate_all_permutation_configs([2, 2], 2, 2)), (1.0, 1.0, 0.0)
    else:
        cfgs, params = np.array(vmc.all_product_configs(2, 2, 2)), (1.0,)
    e, grad = host.exact_sum_finish(host.exact_sum_partial(flat, cfgs, 8, model, params, 0, 1, 16, F64), flat.shape)
    ns = wp = 0.0
    for r in range(2):
        for c in range(2):
            for i in range(2):
                n2 = float(np.sum(grad[r, c, i] ** 2))
                ns += n2
                wp += 0.012 * ((r + 1) * 11 + (c + 1) * 5 + (i + 1) * 2) * n2
    assert abs(ns / norm2 - 1) < 1e-7 and abs(wp / probe - 1) < 1e-7


@pytest.mark.parametrize("name,model,e_ref,tol", K4)
def test_k4_exact_sum_on_device(fixtures_dir, name, model, e_ref, tol):
    """The reference's own exact-summation known answers (test_exact_summation_evaluator.cpp:139-174,
    :250-259, :606, :775) through the C++ ExactSumEnergyEvaluator on the device (f64), incl. the
    4-rank round-robin partition of exact_summation_energy_evaluator.h:201; gradient vs the oracle."""
    host = _host()
    s = qlten_io.load_sitps(os.path.join(fixtures_dir, name))
    flat = synthetic.sitps_to_flat(s, 4)
    if model == "xxz":
        cfgs = np.array(vmc.generate_all_permutation_configs([2, 2], 2, 2))
        params, m = (1.0, 1.0, 0.0), vmc.SquareSpinOneHalfXXZModelOBC()
    else:
        cfgs = np.array(vmc.all_product_configs(2, 2, 2))
        params, m = (1.0,), vmc.TransverseFieldIsingSquareOBC(1.0)
    packed = sum(host.exact_sum_partial(flat, cfgs, 8, model, params, r, 4, 3, F64) for r in range(4))
    e, grad = host.exact_sum_finish(packed, flat.shape)
    assert abs(e - e_ref) < tol
    e_o, g_o, _ = vmc.exact_sum_energy_evaluator(s, list(cfgs), BMPSTruncateParams.SVD(8, 8, 0.0), m)
    assert abs(e - e_o) < 1e-10
    for r in range(2):
        for c in range(2):
            for k in range(2):
                go = g_o[r][c][k]
                gd = grad[r, c, k][:go.shape[0], :go.shape[1], :go.shape[2], :go.shape[3]]
                assert np.max(np.abs(gd - go)) < 1e-9


def test_k5_exact_sum_4x4_d8(fixtures_dir):
    """All 12 870 Sz=0 configurations of the 4x4 D=8 reference state on the device (f32, chi=16):
    E = -9.1891559611 (dense contraction), north-star tolerance 1e-6 relative."""
    host = _host()
    s = qlten_io.load_sitps(os.path.join(fixtures_dir, "tps_square_heisenberg4x4D8Double"))
    flat = synthetic.sitps_to_flat(s, 8)
    cfgs = np.array(vmc.generate_all_permutation_configs([8, 8], 4, 4)).astype(np.int32)
    assert len(cfgs) == 12870
    packed = host.exact_sum_partial(flat, cfgs, 64, "xxz", (1.0, 1.0, 0.0), 0, 1, 1024, F32)
    e, grad = host.exact_sum_finish(packed, flat.shape)
    assert abs(e / -9.1891559611 - 1) < 1e-6


def test_mc_chain_identical_to_oracle_f64():
    """Same std::mt19937 stream, same sweep schedule, f64 device path: the Markov chain is the
    oracle's chain -- identical configurations after 2 sweeps for both updaters."""
    host = _host()
    L, D, chi = 4, 3, 9
    s = synthetic.make_sitps(L, D)
    flat = synthetic.sitps_to_flat(s, D)
    cfgs = synthetic.make_configs(L, 5, "heisenberg")
    seeds = np.array([11, 12, 13, 14, 15], dtype=np.uint64)
    tp = BMPSTruncateParams.SVD(chi, chi, 0.0)
    for name, cls in (("exchange", vmc.MCUpdateSquareNNExchangeOBC), ("fullspace", vmc.MCUpdateSquareNNFullSpaceUpdateOBC)):
        out_cfg, amps, rates = host.mc_sweeps(flat, cfgs, seeds, chi, name, 2, F64)
        for w in range(len(cfgs)):
            comp = vmc.TPSWaveFunctionComponent(s, cfgs[w], tp)
            upd = cls(seed=int(seeds[w]))
            r = [upd(s, comp)[0] for _ in range(2)]
            assert np.array_equal(comp.config, out_cfg[w]), (name, w)
            assert abs(amps[w] / comp.amplitude - 1) < 1e-8
            assert abs(rates[w] - np.mean(r)) < 1e-12


def test_mc_sweep_consistency_f32(fixtures_dir):
    """f32 path: after sweeps the carried amplitude equals a fresh EvaluateAmplitude of the final
    configuration (oracle), and the exchange updater conserves Sz."""
    host = _host()
    s = qlten_io.load_sitps(os.path.join(fixtures_dir, "tps_square_heisenberg4x4D8Double"))
    flat = synthetic.sitps_to_flat(s, 8)
    cfgs = synthetic.make_configs(4, 8, "heisenberg")
    out_cfg, amps, rates = host.mc_sweeps(flat, cfgs, np.arange(8, dtype=np.uint64) + 3, 16, "exchange", 3, F32)
    tp = BMPSTruncateParams.SVD(16, 16, 0.0)
    assert np.all(out_cfg.reshape(8, -1).sum(1) == 8)
    assert np.all((rates >= 0) & (rates <= 1)) and rates.max() > 0
    for w in range(8):
        fresh = vmc.TPSWaveFunctionComponent(s, out_cfg[w], tp).amplitude
        assert abs(amps[w] / fresh - 1) < 1e-4


def test_device_slice_sweep_is_the_per_bond_chain():
    """pepsgpu_sweep_slice_exchange (a whole row / column of exchange moves on the device: traces, Metropolis tests with the
    deviates the host drew ahead, exchanges in the device's configuration table) against the per-bond hook path of the same
    updater (PEPSHOST_NO_DEVICE_SWEEP=1): identical configurations, accept rates and amplitudes after three sweeps, f64 and f32 --
    the device path consumes exactly the deviates the reference's TwoSiteNNUpdateLocalImpl draws, in the same order
    (square_nn_updater.h:142-189).  (The f64 device path is also the one test_mc_chain_identical_to_oracle_f64 holds against
    the oracle's chain.)"""
    import json
    import subprocess
    import sys
    code = r"""
import json, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from peps_amd import hostapi, synthetic
L, D, chi = 6, 4, 12
flat = synthetic.sitps_to_flat(synthetic.make_sitps(L, D, noise=0.5), D)
cfgs = synthetic.make_configs(L, 24, "heisenberg", seed0=13)
out = {}
for name, dt in (("f64", 1), ("f32", 0)):
    c, a, r = hostapi.mc_sweeps(flat, cfgs, np.arange(24, dtype=np.uint64) + 90, chi, "exchange", 3, dt)
    out[name] = {"cfg": c.tolist(), "amp": [float(x) for x in a], "rate": [float(x) for x in r]}
    # the energy evaluation: a row / column per device call (pepsgpu_nn_exchange_slice) against the per-bond hook path
    _, en, _, psi = hostapi.energy_and_holes(flat, cfgs, chi, "xxz", (1.0, 0.8, 0.1), False, dt)
    packed, _, _ = hostapi.mc_energy_grad_partial(flat, cfgs, np.arange(24, dtype=np.uint64) + 90, chi, "exchange", "xxz", (1.0, 0.8, 0.1), 1, 2, dt)
    out[name].update(energy=[float(x) for x in en], psi=np.asarray(psi).tolist(), packed_sum=float(np.sum(packed)), packed_abs=float(np.sum(np.abs(packed))))
print(json.dumps(out))
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for name, env in (("device", {}), ("hook", {"PEPSHOST_NO_DEVICE_SWEEP": "1"})):
        r = subprocess.run([sys.executable, "-c", code, root], env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        res[name] = json.loads(r.stdout.strip().splitlines()[-1])
    for dt, tol in (("f64", 1e-12), ("f32", 1e-5)):
        a, b = res["device"][dt], res["hook"][dt]
        assert a["cfg"] == b["cfg"], dt                      # the same chain
        assert a["rate"] == b["rate"] and max(a["rate"]) > 0
        assert np.max(np.abs(np.array(a["amp"]) / np.array(b["amp"]) - 1)) < tol
        assert np.max(np.abs(np.array(a["energy"]) / np.array(b["energy"]) - 1)) < tol
        assert np.max(np.abs(np.array(a["psi"]) / np.array(b["psi"]) - 1)) < tol
        assert abs(a["packed_sum"] - b["packed_sum"]) < tol * a["packed_abs"]      # energy + gradient sums of the VMC samples
    assert res["device"]["f64"]["cfg"] != synthetic.make_configs(6, 24, "heisenberg", seed0=13).tolist()      # ... and it moved


def test_device_slice_sweeps_of_the_other_updaters_and_types():
    """Round 6: the device-side slice sweeps beyond the real bosonic exchange updater, each against the per-bond
    hook path of the same updater (PEPSHOST_NO_DEVICE_SWEEP=1) -- identical configurations and accept rates after two sweeps:
      * MCUpdateSquareNNFullSpaceUpdateOBC (square_nn_updater.h:253-293; pepsgpu_sweep_slice_fullspace: Suwa-Todo over the d^2 states
        of a bond on the device, the walker's mt19937 consumed two words per bond as the reference's long double draw does), real
        f64 / f32 and complex;
      * MCUpdateSquareNNExchangeOBC on a COMPLEX state (interleaved amplitudes, |psi'| / |psi|);
      * MCUpdateSquareNNExchangeOBC on a FERMIONIC state (pepsgpu_sweep_slice_exchange_tab: the exchange of the decorated extended
        states as a table), real f64 / f32 -- the chain of the hook path is the one K9 pins on the reference's regression energy."""
    import json
    import subprocess
    import sys
    code = r"""
import json, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from peps_amd import hostapi, synthetic, fermion
L, D, chi = 6, 4, 12
flat = synthetic.sitps_to_flat(synthetic.make_sitps(L, D, noise=0.5), D)
rng = np.random.default_rng(3)
cflat = flat * np.exp(2j * np.pi * rng.uniform(size=flat.shape))
cfgs = synthetic.make_configs(L, 16, "heisenberg", seed0=13)
seeds = np.arange(16, dtype=np.uint64) + 90
out = {}
for name, dt in (("fullspace_f64", 1), ("fullspace_f32", 0)):
    c, a, r = hostapi.mc_sweeps(flat, cfgs, seeds, chi, "fullspace", 2, dt)
    out[name] = {"cfg": c.tolist(), "amp": [float(x) for x in a], "rate": [float(x) for x in r]}
for name, upd in (("exchange_c128", "exchange"), ("fullspace_c128", "fullspace")):
    c, a, r = hostapi.mc_sweeps_complex(cflat, cfgs, seeds, chi, upd, 2)
    out[name] = {"cfg": c.tolist(), "amp": [[float(x.real), float(x.imag)] for x in a], "rate": [float(x) for x in r]}
st = fermion.random_even_state(5, 4, 3, seed=11)
fc = np.stack([np.random.default_rng(100 + k).permutation(np.r_[np.zeros(10, dtype=int), np.ones(10, dtype=int)]).reshape(5, 4) for k in range(12)])
for name, dt in (("fermion_f64", 1), ("fermion_f32", 0)):
    c, a, r = hostapi.fermion_mc_sweeps(st, fc, np.arange(12, dtype=np.uint64) + 7, 9, 2, dt)
    out[name] = {"cfg": c.tolist(), "amp": [float(x) for x in a], "rate": [float(x) for x in r], "start": fc.tolist()}
print(json.dumps(out))
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for name, env in (("device", {}), ("hook", {"PEPSHOST_NO_DEVICE_SWEEP": "1"})):
        r = subprocess.run([sys.executable, "-c", code, root], env=dict(os.environ, **env), capture_output=True, text=True, timeout=1800)
        assert r.returncode == 0, r.stderr[-2000:]
        res[name] = json.loads(r.stdout.strip().splitlines()[-1])
    for key, tol in (("fullspace_f64", 1e-12), ("fullspace_f32", 1e-5), ("exchange_c128", 1e-12), ("fullspace_c128", 1e-12),
                     ("fermion_f64", 1e-12), ("fermion_f32", 1e-5)):
        a, b = res["device"][key], res["hook"][key]
        assert a["cfg"] == b["cfg"], key                     # the same chain
        assert a["rate"] == b["rate"] and max(a["rate"]) > 0, key
        x, y = np.array(a["amp"]), np.array(b["amp"])
        if x.ndim == 2:
            x, y = x[:, 0] + 1j * x[:, 1], y[:, 0] + 1j * y[:, 1]
        assert np.max(np.abs(x / y - 1)) < tol, (key, np.max(np.abs(x / y - 1)))
    assert res["device"]["fermion_f64"]["cfg"] != res["device"]["fermion_f64"]["start"]


def test_monte_carlo_engine_rescue_and_normalize_order1(fixtures_dir):
    """MonteCarloEngine of the host layer (monte_carlo_engine.h:146-240 WarmUp / StepSweep / NormalizeStateOrder1, :340-414
    EnsureConfigurationValidity): walkers whose amplitude falls outside the rescue window take the configuration of the first
    valid walker and the batch counts as not warmed up; after the warm-up sweeps every site tensor is scaled by
    (1 / max |psi|)^(1 / (Lx Ly)), so the largest amplitude of the batch is 1 and every amplitude is the oracle's amplitude of the
    SCALED state on the final configuration."""
    host = _host()
    s = qlten_io.load_sitps(os.path.join(fixtures_dir, "tps_square_heisenberg4x4D8Double"))
    flat = synthetic.sitps_to_flat(s, 8)
    cfgs = synthetic.make_configs(4, 12, "heisenberg", seed0=5)
    tp = BMPSTruncateParams.SVD(16, 16, 0.0)
    a0 = np.array([abs(vmc.TPSWaveFunctionComponent(s, c, tp).amplitude) for c in cfgs])
    thr = float(np.sort(a0)[3] * 1.0001)             # the four smallest amplitudes are "invalid"
    bad = a0 < thr
    assert bad.sum() == 4 and not bad[int(np.argmax(~bad))]
    # no sweeps: rescue + normalisation only
    st, out_cfg, amps, scale, rescued = host.mc_engine_warmup(flat, cfgs, np.arange(12, dtype=np.uint64) + 11, 16, 0, True, thr, 0.0, F64)
    src = int(np.argmax(~bad))
    assert rescued == 4
    for w in range(12):
        assert np.array_equal(out_cfg[w], cfgs[src] if bad[w] else cfgs[w])
    assert abs(np.max(np.abs(amps)) - 1.0) < 1e-9
    assert abs(scale * np.max(a0[~bad]) - 1.0) < 1e-9
    assert np.allclose(st, flat * scale ** (1.0 / 16), rtol=1e-14)
    s2 = synthetic.flat_to_sitps(st)
    for w in (0, src, int(np.argmax(bad))):
        assert abs(amps[w] / vmc.TPSWaveFunctionComponent(s2, out_cfg[w], tp).amplitude - 1) < 1e-9
    # rescue disabled: the reference aborts, the host layer throws
    with pytest.raises(RuntimeError):
        host.mc_engine_warmup(flat, cfgs, np.arange(12, dtype=np.uint64) + 11, 16, 0, False, thr, 0.0, F64)
    # several ranks (monte_carlo_engine.h:344-387, :214-222): the two collectives are callbacks.  A stand-in for a second rank: the
    # local batch = the eleven walkers below the largest amplitude, the window admits only that largest one -- EVERY local walker is
    # invalid, the "other rank" holds the valid configuration and a larger maximum: all eleven take its configuration, the state is
    # scaled by the other rank's maximum.
    top = int(np.argmax(a0))
    rest = np.array([w for w in range(12) if w != top])
    thr_top = float(np.sort(a0)[-2] * 1.0001)
    assert a0[top] > thr_top
    calls = []

    def exchange(n_invalid, have_valid, cfg):
        calls.append((n_invalid, have_valid))
        return n_invalid, cfgs[top].ravel().astype(np.int32)        # (the other rank: no invalid walker, it is the first valid rank)

    st, out_cfg, amps, scale, rescued = host.mc_engine_warmup(flat, cfgs[rest], np.arange(11, dtype=np.uint64) + 11, 16, 0, True, thr_top, 0.0, F64,
                                                              lambda x: max(x, 3.0 * a0[top]), exchange)
    assert rescued == 11 and calls == [(11, False)] and all(np.array_equal(out_cfg[w], cfgs[top]) for w in range(11))
    assert abs(scale * 3.0 * a0[top] - 1.0) < 1e-9 and np.max(np.abs(np.abs(amps) - 1.0 / 3.0)) < 1e-9
    # a rank with valid and invalid walkers among others: the donor is what the collective returns (the first valid RANK's
    # configuration), not the local first valid walker
    other = int(np.argmax(~bad & (np.arange(12) != src)))
    st, out_cfg, amps, scale, rescued = host.mc_engine_warmup(flat, cfgs, np.arange(12, dtype=np.uint64) + 11, 16, 0, True, thr, 0.0, F64, None,
                                                              lambda ni, hv, c: (ni + 2, cfgs[other].ravel()))
    assert rescued == 4 and all(np.array_equal(out_cfg[w], cfgs[other] if bad[w] else cfgs[w]) for w in range(12))
    # no rank holds a valid walker: the reference aborts
    with pytest.raises(RuntimeError):
        host.mc_engine_warmup(flat, cfgs[rest], np.arange(11, dtype=np.uint64) + 11, 16, 0, True, thr_top, 0.0, F64, None, lambda ni, hv, c: (-1, c))
    # with warm-up sweeps (f32): amplitudes stay consistent with the scaled state, max |psi| = 1
    st, out_cfg, amps, scale, rescued = host.mc_engine_warmup(flat, cfgs, np.arange(12, dtype=np.uint64) + 11, 16, 2, True, 0.0, 0.0, F32)
    assert rescued == 0 and abs(np.max(np.abs(amps)) - 1.0) < 1e-5
    s2 = synthetic.flat_to_sitps(st)
    for w in (0, 5, 11):
        assert abs(amps[w] / vmc.TPSWaveFunctionComponent(s2, out_cfg[w], tp).amplitude - 1) < 1e-4


def test_device_gradient_accumulation_matches_host_path(fixtures_dir):
    """pepsgpu_grad_accumulate (holes resident in HBM) == host-side accumulation of PunchHole outputs."""
    from peps_amd import capi
    s = qlten_io.load_sitps(os.path.join(fixtures_dir, "tps_square_heisenberg4x4D8Double"))
    flat = synthetic.sitps_to_flat(s, 8)
    cfgs = synthetic.make_configs(4, 6, "heisenberg")
    ctx = capi.Context(4, 4, 8, 2, 16, dtype=capi.F64, max_walkers=6)
    ctx.state_upload(flat)
    ctx.set_configs(cfgs)
    psi = ctx.evaluate_amplitude()
    eloc = np.linspace(-1.0, 1.0, 6)
    ctx.grad_reset()
    ref_so = np.zeros_like(flat)
    ref_seo = np.zeros_like(flat)
    ctx.generate_bmps_approach(3)
    for row in range(4):
        ctx.init_bten(0, row)
        ctx.grow_full_bten(2, row, 1, True)
        for col in range(4):
            h = ctx.punch_hole(row, col, 0)
            ctx.punch_hole_store(row, col, 0)
            for w in range(6):
                ref_so[row, col, cfgs[w, row, col]] += h[w] / psi[w]
                ref_seo[row, col, cfgs[w, row, col]] += eloc[w] * h[w] / psi[w]
            if col < 3:
                ctx.shift_bten_window(2)
        if row < 3:
            ctx.shift_bmps_window(1)
    ctx.grad_accumulate(psi, eloc, False)
    so, seo = ctx.grad_read()
    assert np.max(np.abs(so - ref_so)) < 1e-10 * np.max(np.abs(ref_so))
    assert np.max(np.abs(seo - ref_seo)) < 1e-10 * np.max(np.abs(ref_seo))
    # pepsgpu_grad_accumulate_states: the component of every site named by the caller (fermionic states: the extended
    # state of the decoration the holes were punched in) -- here the flipped configuration, after the walkers moved on
    other = 1 - cfgs
    ctx.set_configs(np.ascontiguousarray(cfgs[::-1]))
    ctx.grad_reset()
    ctx.grad_accumulate(psi, eloc, False, states=other)
    so2, seo2 = ctx.grad_read()
    assert np.max(np.abs(so2 - ref_so[:, :, ::-1])) < 1e-10 * np.max(np.abs(ref_so))
    assert np.max(np.abs(seo2 - ref_seo[:, :, ::-1])) < 1e-10 * np.max(np.abs(ref_seo))
    with pytest.raises(ValueError):
        ctx.grad_accumulate(psi, eloc, False, states=other + 2)


def test_mc_energy_and_gradient_vs_exact_sum(fixtures_dir):
    """MC evaluator loop (sweep + CalEnergyAndHoles + O* accumulation, all walkers) against the
    exact summation on the 2x2 Heisenberg fixture: energy within the statistical error, gradient
    direction error < 0.3 (the reference's own criterion, test_mc_energy_grad_evaluator.cpp:304-336)."""
    host = _host()
    s = qlten_io.load_sitps(os.path.join(fixtures_dir, "heisenberg_tps_double_from_simple_update"))
    flat = synthetic.sitps_to_flat(s, 4)
    all_cfgs = np.array(vmc.generate_all_permutation_configs([2, 2], 2, 2))
    e_ex, g_ex = host.exact_sum_finish(host.exact_sum_partial(flat, all_cfgs, 8, "xxz", (1.0, 1.0, 0.0), 0, 1, 8, F64), flat.shape)
    n = 64
    cfgs = np.stack([all_cfgs[i % len(all_cfgs)] for i in range(n)])
    packed, _, acc = host.mc_energy_grad_partial(flat, cfgs, np.arange(n, dtype=np.uint64) + 1, 8, "exchange", "xxz",
                                                 (1.0, 1.0, 0.0), 5, 40, F64)
    e_mc, g_mc = host.exact_sum_finish(packed, flat.shape)
    nsamp = packed[-1]
    var = packed[-2] / nsamp - (packed[-3] / nsamp) ** 2
    err = np.sqrt(max(var, 0) / nsamp)
    assert nsamp == n * 40
    assert abs(e_mc - e_ex) < 6 * err + 1e-3
    cosang = np.sum(g_mc * g_ex) / np.linalg.norm(g_mc) / np.linalg.norm(g_ex) if np.linalg.norm(g_ex) > 1e-8 else 1.0
    assert np.linalg.norm(g_mc - g_ex) < 0.3 * max(np.linalg.norm(g_ex), 0.05) or cosang > 0.9


@pytest.mark.parametrize("dt,tol", [(F64, 1e-9), (F32, 2e-5)])
def test_j1j2_energy_fixed_configs_and_exact_sum(dt, tol):
    """J1-J2 XXZ (SquareSpinOneHalfJ1J2XXZModelOBC): the NNN pass of square_nnn_energy_solver.h:203-265
    on the device (BTen2 growth, ShiftBTen2Window, ReplaceNNNSiteTrace per diagonal) against the oracle
    on identical configurations, and the 3x3 exact-sum energy against the oracle's."""
    host = _host()
    L, D, chi = 5, 3, 9
    s = synthetic.make_sitps(L, D)
    cfgs = synthetic.make_configs(L, 5, "heisenberg")
    params = (1.0, 0.9, 0.4, 0.55, 0.1)
    ref = _oracle_energy(s, cfgs, chi, vmc.SquareSpinOneHalfJ1J2XXZModelOBC(*params))
    amps, en, holes, psi = host.energy_and_holes(synthetic.sitps_to_flat(s, D), cfgs, chi, "j1j2", params, True, dt)
    for w, (a, e, h, ps) in enumerate(ref):
        assert abs(amps[w] / a - 1) < tol
        assert abs(en[w] - e) < tol * max(1.0, abs(e)) * 10
    s3 = synthetic.make_sitps(3, 2)
    all_cfg = np.array(vmc.all_product_configs(2, 3, 3)).astype(np.int32)
    packed = host.exact_sum_partial(synthetic.sitps_to_flat(s3, 2), all_cfg, 16, "j1j2", params, 0, 1, 128, dt)
    e, grad = host.exact_sum_finish(packed, (3, 3, 2, 2))
    e_o, g_o, _ = vmc.exact_sum_energy_evaluator(s3, list(all_cfg), BMPSTruncateParams.SVD(16, 16, 0.0),
                                                 vmc.SquareSpinOneHalfJ1J2XXZModelOBC(*params))
    assert abs(e / e_o - 1) < tol


@pytest.mark.parametrize("dt,tol", [(F64, 1e-9), (F32, 2e-5)])
def test_triangle_heisenberg_energy_and_exact_sum(dt, tol):
    """SpinOneHalfTriHeisenbergSqrPEPS (spin_onehalf_triangle_heisenberg_sqrpeps.h:39-112): the Heisenberg model of the triangular
    lattice on a square PEPS -- nearest-neighbour bonds + the left-down -> right-up diagonal of every plaquette, the other diagonal
    contributing nothing -- local energies against the oracle on identical configurations, the 3x3 exact-sum energy against the
    oracle's, and the difference to the J1-J2 model with both diagonals (the dropped diagonal is not zero on this state)."""
    host = _host()
    L, D, chi = 5, 3, 9
    s = synthetic.make_sitps(L, D)
    cfgs = synthetic.make_configs(L, 5, "heisenberg")
    ref = _oracle_energy(s, cfgs, chi, vmc.SpinOneHalfTriHeisenbergSqrPEPS())
    amps, en, holes, psi = host.energy_and_holes(synthetic.sitps_to_flat(s, D), cfgs, chi, "triangle", (), True, dt)
    _, en_j1j2, _, _ = host.energy_and_holes(synthetic.sitps_to_flat(s, D), cfgs, chi, "j1j2", (1.0, 1.0, 1.0, 1.0, 0.0), False, dt)
    for w, (a, e, h, ps) in enumerate(ref):
        assert abs(amps[w] / a - 1) < tol
        assert abs(en[w] - e) < tol * max(1.0, abs(e)) * 10
    assert np.max(np.abs(en - en_j1j2)) > 1e-2
    s3 = synthetic.make_sitps(3, 2)
    all_cfg = np.array(vmc.all_product_configs(2, 3, 3)).astype(np.int32)
    packed = host.exact_sum_partial(synthetic.sitps_to_flat(s3, 2), all_cfg, 16, "triangle", (), 0, 1, 128, dt)
    e, grad = host.exact_sum_finish(packed, (3, 3, 2, 2))
    e_o, g_o, _ = vmc.exact_sum_energy_evaluator(s3, list(all_cfg), BMPSTruncateParams.SVD(16, 16, 0.0), vmc.SpinOneHalfTriHeisenbergSqrPEPS())
    assert abs(e / e_o - 1) < tol


@pytest.mark.parametrize("dt,tol", [(F64, 1e-9), (F32, 2e-5)])
def test_triangle_j1j2_heisenberg_energy_holes_and_exact_sum(dt, tol):
    """SpinOneHalfTriJ1J2HeisenbergSqrPEPS (spin_onehalf_triangle_heisenbergJ1J2_sqrpeps.h:304-446): the model's own traversal on the
    device -- h / v bonds on BTen, both plaquette diagonals and the flat sqrt5 link on BTen2 in the row pass, the steep sqrt5 link on
    BTen2(UP / DOWN, remain_sites 3) in the column pass -- local energies, holes and the psi list against the oracle on identical
    configurations (5x5: every ShiftBTen2Window of both passes runs), j2 = 0 against j2 != 0 (the J2 links are not zero on this state)
    and the 3x3 exact-sum energy against the oracle's."""
    host = _host()
    L, D, chi, j2 = 5, 3, 9, 0.35
    s = synthetic.make_sitps(L, D)
    cfgs = synthetic.make_configs(L, 5, "heisenberg")
    ref = _oracle_energy(s, cfgs, chi, vmc.SpinOneHalfTriJ1J2HeisenbergSqrPEPS(j2))
    amps, en, holes, psi = host.energy_and_holes(synthetic.sitps_to_flat(s, D), cfgs, chi, "trij1j2", (j2,), True, dt)
    _, en_j1, _, _ = host.energy_and_holes(synthetic.sitps_to_flat(s, D), cfgs, chi, "trij1j2", (0.0,), False, dt)
    _, en_tri, _, _ = host.energy_and_holes(synthetic.sitps_to_flat(s, D), cfgs, chi, "triangle", (), False, dt)
    for w, (a, e, h, ps) in enumerate(ref):
        assert abs(amps[w] / a - 1) < tol
        assert abs(en[w] - e) < tol * max(1.0, abs(e)) * 10
        for r in range(L):
            for c in range(L):
                hr = h[r][c]
                hd = holes[w, r, c][:hr.shape[0], :hr.shape[1], :hr.shape[2], :hr.shape[3]]
                assert np.max(np.abs(hd - hr)) < tol * 10 * np.max(np.abs(hr))
        assert np.max(np.abs(psi[:, w] / np.array(ps) - 1)) < tol * 10
    assert np.max(np.abs(en - en_j1)) > 1e-2
    assert np.max(np.abs(en_j1 - en_tri)) < tol * 10 * max(1.0, np.max(np.abs(en_tri)))      # j2 = 0: the J1 model of the triangular lattice
    s3 = synthetic.make_sitps(3, 2)
    all_cfg = np.array(vmc.all_product_configs(2, 3, 3)).astype(np.int32)
    packed = host.exact_sum_partial(synthetic.sitps_to_flat(s3, 2), all_cfg, 16, "trij1j2", (j2,), 0, 1, 128, dt)
    e, grad = host.exact_sum_finish(packed, (3, 3, 2, 2))
    e_o, g_o, _ = vmc.exact_sum_energy_evaluator(s3, list(all_cfg), BMPSTruncateParams.SVD(16, 16, 0.0), vmc.SpinOneHalfTriJ1J2HeisenbergSqrPEPS(j2))
    assert abs(e / e_o - 1) < tol
    gmax = max(np.max(np.abs(g_o[r][c][k])) for r in range(3) for c in range(3) for k in range(2))
    for r in range(3):                                       # the gradient runs through this model's own hole traversal
        for c in range(3):
            for k in range(2):
                go = g_o[r][c][k]
                gd = grad[r, c, k][:go.shape[0], :go.shape[1], :go.shape[2], :go.shape[3]]
                assert np.max(np.abs(gd - go)) < tol * 10 * max(1.0, gmax)


@pytest.mark.parametrize("scheme,name", [(1, "Variational2Site"), (2, "Variational1Site")])
def test_xxz_energy_with_variational_truncate_params(scheme, name):
    """The C++ solver with BMPSTruncateParams::Variational2Site / 1Site (bmps.h:81-97): local energies and
    amplitudes follow the oracle run with the same parameters on a truncating contraction (5x5, D=3, chi=4)."""
    host = _host()
    L, D, chi = 5, 3, 4
    s = synthetic.make_sitps(L, D, noise=1.0)
    cfgs = synthetic.make_configs(L, 4, "heisenberg")
    tp = getattr(BMPSTruncateParams, name)(chi, chi, 0.0, 1e-13, 30)
    model = vmc.SquareSpinOneHalfXXZModelOBC()
    ref = []
    for c in cfgs:
        comp = vmc.TPSWaveFunctionComponent(s, c, tp)
        e, _, _ = model.CalEnergyAndHoles(s, comp, True)
        ref.append((comp.amplitude, e))
    svd = _oracle_energy(s, cfgs, chi, model)
    host.set_truncate_params(chi, 0.0, scheme, 1e-13, 30)
    try:
        amps, en, _, _ = host.energy_and_holes(synthetic.sitps_to_flat(s, D), cfgs, chi, "xxz", (1.0, 1.0, 0.0), True, F64)
    finally:
        host.set_truncate_params()
    for w, (a, e) in enumerate(ref):
        assert abs(amps[w] / a - 1) < 1e-7
        assert abs(en[w] - e) < 1e-6 * max(1.0, abs(e))
    assert max(abs(svd[w][1] - ref[w][1]) for w in range(len(ref))) > 1e-5      # the scheme is in effect


@pytest.mark.parametrize("weights,init", [([1, 0.5, 0.3, 0.01, 0.06, 2], 0), ([1, 0.3, 1e-30], 0), ([9.6, 9.6, 1], 1), ([0.0, 1.0, 0.0], 1)])
def test_device_suwa_todo_is_the_reference_chain(weights, init):
    """The decision of pepsgpu_sweep_slice_fullspace (sw_suwa_todo_decide, float64) on the unit cases of the reference's
    test_suwa_todo_update.cpp:100-103 (+ an absorbing state, :52-58) against the host layer's SuwaTodoStateUpdate (long double, the
    reference's arithmetic; tests/test_cpu_suwa_todo.py pins it on the reference's cases and on the oracle): the same std::mt19937
    stream -> the same chain, 4000 steps, two seeds."""
    from peps_amd import capi, hostapi
    from oracle import vmc
    for seed in (0, 20240115):
        rng = vmc.StdMT19937(seed)
        words = np.array([rng.raw() for _ in range(2 * 4000)], dtype=np.uint32)
        dev = capi.diag_suwa_todo(weights, init, words)
        host = hostapi.suwa_todo_chain(init, weights, seed, 4000)
        assert np.array_equal(dev, host), int(np.argmax(dev != host))
