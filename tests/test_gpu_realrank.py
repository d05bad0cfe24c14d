"""End-to-end parity of the absorption at the numerical rank of a REAL PEPS: the reference's own optimised D = 8 Heisenberg
state (tests/slow_tests/test_data/tps_square_heisenberg4x4D8Double, used at test_boson_mc_peps_measure.cpp:55-62) tiled by
position class to larger lattices (peps_amd.synthetic.tile_flat_state).  Its carry is ~0.9 dense within its shapes, so at bulk
shapes (carry D * chi = 256 rows) the f32 route above 128 live rows runs: device amplitude AND XXZ local energy against the
float64 oracle, with ctx.stats() proving the route (carry_live_max > 128).  Configurations: checkerboard + random
nearest-neighbour exchanges (what a Monte-Carlo chain on an antiferromagnetic state visits)."""
import os

import numpy as np
import pytest

from oracle import vmc
from oracle.bmps import BMPSTruncateParams
from peps_amd import synthetic

from conftest import FIXTURES

pytestmark = pytest.mark.gpu
F32, F64 = 0, 1
REAL = os.path.join(FIXTURES, synthetic.REAL_FIXTURE)


def _state(L, D=8):
    """tiled state (flat upload layout), rescaled so that psi(checkerboard) = O(1)"""
    from peps_amd import capi, hostapi
    f4 = hostapi.load_sitps(REAL, 8)
    if D < 8:        # a D < 8 variant for the C3 shapes: the leading D x D x D x D corner of every tensor
        f4 = np.ascontiguousarray(f4[:, :, :, :D, :D, :D, :D])
    flat = synthetic.tile_flat_state(f4, L)
    ctx = capi.Context(L, L, D, 2, 4 * D, dtype=capi.F64, max_walkers=1)
    ctx.state_upload(flat)
    ctx.set_configs(synthetic.checkerboard(L)[None])
    psi = float(ctx.evaluate_amplitude()[0])
    ctx.close()
    return flat * abs(psi) ** (-1.0 / (L * L))


def _oracle(flat, cfgs, chi):
    sitps = synthetic.flat_to_sitps(flat)
    tp = BMPSTruncateParams.SVD(chi, chi, 0.0)
    model = vmc.SquareSpinOneHalfXXZModelOBC(1.0, 1.0, 0.0)
    ref_a, ref_e = [], []
    for c in cfgs:
        comp = vmc.TPSWaveFunctionComponent(sitps, c, tp)
        ref_a.append(comp.amplitude)
        ref_e.append(model.CalEnergyAndHoles(sitps, comp, False)[0])
    return np.array(ref_a), np.array(ref_e)


CASES = [("6x6 D=8 chi=32", 6, 8, 32, 128), ("8x8 D=8 chi=32", 8, 8, 32, 128), ("C3 shapes 10x10 D=6 chi=24", 10, 6, 24, 72)]


@pytest.mark.parametrize("name,L,D,chi,live_min", CASES)
@pytest.mark.parametrize("dt,tol_a,tol_e", [(F32, 1e-5, 1e-6), (F64, 1e-9, 1e-9)])
def test_real_rank_amplitude_and_energy_vs_oracle(name, L, D, chi, live_min, dt, tol_a, tol_e, monkeypatch):
    from peps_amd import capi, hostapi
    flat = _state(L, D)
    cfgs = synthetic.make_configs_near_neel(L, 3 if L >= 10 else 4, seed0=211)
    ref_a, ref_e = _oracle(flat, cfgs, chi)
    monkeypatch.setenv("PEPSGPU_DEBUG_SWEEPS", "1")
    ctx = capi.Context(L, L, D, 2, chi, dtype=dt, max_walkers=len(cfgs))
    monkeypatch.delenv("PEPSGPU_DEBUG_SWEEPS")
    ctx.state_upload(flat)
    ctx.set_configs(cfgs)
    amps = ctx.evaluate_amplitude()
    st = ctx.stats()
    assert np.all(ctx.walker_flags() == 0)
    ctx.close()
    assert st["carry_live_max"] > live_min, st       # the dense route: more than 128 live carry rows (of D * chi = 256)
    assert np.max(np.abs(amps / ref_a - 1)) < tol_a, (amps, ref_a)
    a2, en, _, psi = hostapi.energy_and_holes(flat, cfgs, chi, "xxz", (1.0, 1.0, 0.0), False, dt)
    assert np.max(np.abs(a2 / ref_a - 1)) < tol_a
    assert np.max(np.abs(en / ref_e - 1)) < tol_e, (en, ref_e)


def test_real_rank_c4_batch_f32_vs_f64_and_routes():
    """C4 at scale on the tiled state (no oracle sample can afford it): 128 walkers, f32 against the f64 device mode (pinned to
    the oracle below) on the row route AND on the column route, live carry > 128 rows, no walker flagged.  Round 5: with the
    backward pair and Y = Tt V^T of the precise sites accumulated in float64 the f32 amplitude sits at max 8e-6 / median 1.8e-6
    (n = 256, row route); round 6 (pivoted first compression, float64 Cholesky-QR of the projected rows): max 7.9e-6 / 5.6e-6, median
    1.5e-6 / 1.7e-6 on the two routes -- asserted here at max < 1e-5 and median < 4e-6 on both (round 5: 2e-5 / 5e-6; rounds 3-4 a 3e-4 distribution).
    The row and the column contraction are DIFFERENT truncations of the same network: they agree to the truncation error
    (chi = 32 against chi = 48 changes psi by ~1e-5 on this state), asserted at 1e-3."""
    from peps_amd import capi
    L, D, chi, _ = synthetic.CONFIGS["C4"]
    flat = _state(L)
    cfgs = synthetic.make_configs_near_neel(L, 128, seed0=307)
    row, col = {}, {}
    for dt in (capi.F32, capi.F64):
        os.environ["PEPSGPU_DEBUG_SWEEPS"] = "1"
        try:
            ctx = capi.Context(L, L, D, 2, chi, dtype=dt, max_walkers=len(cfgs))
        finally:
            os.environ.pop("PEPSGPU_DEBUG_SWEEPS", None)
        ctx.state_upload(flat)
        ctx.set_configs(cfgs)
        row[dt] = ctx.evaluate_amplitude()
        assert np.all(ctx.walker_flags() == 0)
        st = ctx.stats()
        assert st["carry_live_max"] > 128, st
        ctx.grow_bmps_for_col(0)
        ctx.init_bten(capi.UP, 0)
        ctx.grow_full_bten(capi.DOWN, 0, 2, True)
        col[dt] = ctx.trace(0, 0, capi.VERTICAL)
        ctx.close()
    for a in (row, col):
        rel = np.abs(a[capi.F32] / a[capi.F64] - 1)
        print("C4 real state f32 vs f64 mode, %s route: max %.2e median %.2e (n = %d)" % ("row" if a is row else "column", rel.max(), np.median(rel), len(rel)))
        assert np.median(rel) < 4e-6 and np.max(rel) < 1e-5, (int(np.argmax(rel)), float(np.max(rel)), float(np.median(rel)))
    assert np.max(np.abs(col[capi.F64] / row[capi.F64] - 1)) < 1e-3


def test_real_rank_c4_amplitudes_vs_oracle():
    """The tiled real state at C4 (12x12, D=8, chi=32) against the ORACLE itself (oracle/cbmps.c, the float64 plain-C restatement
    of bmps_impl.h:756-862 / :225-263 on LAPACK; one configuration per process), 32 near-Neel configurations: the f64 device mode
    to 1e-8, the f32 mode to SURVEY 8(d)'s 1e-5 on EVERY configuration (round 3 asserted a 1e-4 distribution here: the f32
    accumulation of Y = Tt V^T carried a common-mode 1.5e-5; DESIGN 3e has the budget by stage)."""
    from oracle import cbmps
    from peps_amd import capi
    L, D, chi, _ = synthetic.CONFIGS["C4"]
    flat = _state(L)
    cfgs = synthetic.make_configs_near_neel(L, 32, seed0=307)
    ref, _, _ = cbmps.amplitudes_multiprocess(flat, cfgs, chi, min(16, os.cpu_count() or 1))
    for dt, tol in ((capi.F64, 1e-8), (capi.F32, 1e-5)):
        ctx = capi.Context(L, L, D, 2, chi, dtype=dt, max_walkers=len(cfgs))
        ctx.state_upload(flat)
        ctx.set_configs(cfgs)
        a = ctx.evaluate_amplitude()
        assert np.all(ctx.walker_flags() == 0)
        ctx.close()
        rel = np.abs(a / ref - 1)
        print("C4 real state %s vs oracle/cbmps.c: max %.2e median %.2e (n = %d)" % ("f64" if dt == capi.F64 else "f32", rel.max(), np.median(rel), len(cfgs)))
        assert rel.max() < tol, rel


def test_f64_dense_truncation_route_against_the_general_kernels():
    """Round 5: the dense float64 truncation route (two Gram + Cholesky compressions, oversampled subspace, Rayleigh-Ritz Jacobi in
    LDS, guard, second-chance factorisation; engine_impl.h) against the general one-sided Jacobi it replaces on the dense sites
    (PEPSGPU_NO_F64_DENSE_ROUTE=1, read once per process: hence the subprocess) -- 8x8 tiled real state, D = 8, chi = 32 (carry of
    256 rows): the two f64 amplitudes agree to 2e-9 on every configuration, and the profile shows the route's Gram category running."""
    import json
    import subprocess
    import sys
    code = r'''
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from peps_amd import capi, synthetic
import test_gpu_realrank as t
L, D, chi = 8, 8, 32
flat = t._state(L)
cfgs = synthetic.make_configs_near_neel(L, 24, seed0=11)
ctx = capi.Context(L, L, D, 2, chi, dtype=capi.F64, max_walkers=len(cfgs))
ctx.state_upload(flat)
ctx.set_configs(cfgs)
ctx.evaluate_amplitude()
ctx.profile_enable(1)
ctx.set_configs(cfgs)
a = ctx.evaluate_amplitude()
prof = ctx.profile_read()
assert np.all(ctx.walker_flags() == 0)
print("RESULT " + json.dumps({"a": a.tolist(), "trunc_gram_launches": prof["trunc_gram"]["launches"]}))
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    # (round 6: "route" = the pivoted row selection + Gram-Schmidt + subspace iteration form; "round5" = the two-Cholesky form it replaced)
    for name, env in (("route", {}), ("round5", {"PEPSGPU_F64_PIVOT": "0"}), ("general", {"PEPSGPU_NO_F64_DENSE_ROUTE": "1"})):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=root, env=dict(os.environ, PYTHONPATH=root, **env),
                           timeout=900)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        res[name] = json.loads([l for l in r.stdout.split("\n") if l.startswith("RESULT ")][0][7:])
    b = np.array(res["general"]["a"])
    assert res["general"]["trunc_gram_launches"] == 0
    for name in ("route", "round5"):
        a = np.array(res[name]["a"])
        print("f64 dense %s vs general kernels: max rel diff %.2e (n = %d)" % (name, np.max(np.abs(a / b - 1)), len(a)))
        assert res[name]["trunc_gram_launches"] > 0
        assert np.max(np.abs(a / b - 1)) < 2e-9


def test_c128_dense_truncation_route_against_the_general_kernels():
    """The same route for the COMPLEX element type (engine_cplx.h, round 5: Hermitian Gram + Cholesky compressions, complex Jacobi on
    the small factor and on the 2 chi rows of Z = U^H M) against the general complex Jacobi (PEPSGPU_NO_C128_DENSE_ROUTE=1, subprocess):
    8x8 tiled real state with a random phase on every tensor element, D = 8, chi = 32 -- amplitudes (complex) agree to 2e-9."""
    import json
    import subprocess
    import sys
    code = r'''
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from peps_amd import capi, synthetic
import test_gpu_realrank as t
L, D, chi = 8, 8, 32
flat = t._state(L)
flat = flat * np.exp(2j * np.pi * np.random.default_rng(5).uniform(size=flat.shape))
cfgs = synthetic.make_configs_near_neel(L, 12, seed0=11)
ctx = capi.Context(L, L, D, 2, chi, dtype=capi.C128, max_walkers=len(cfgs))
ctx.state_upload(flat)
ctx.set_configs(cfgs)
a = ctx.evaluate_amplitude()
assert np.all(ctx.walker_flags() == 0)
print("RESULT " + json.dumps({"re": a.real.tolist(), "im": a.imag.tolist()}))
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    # (round 6: "route" = the randomised range finder + subspace iteration form; "round5" = the Hermitian two-Cholesky form it replaced)
    for name, env in (("route", {}), ("round5", {"PEPSGPU_F64_PIVOT": "0"}), ("general", {"PEPSGPU_NO_C128_DENSE_ROUTE": "1"})):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=root, env=dict(os.environ, PYTHONPATH=root, **env),
                           timeout=1500)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        d = json.loads([l for l in r.stdout.split("\n") if l.startswith("RESULT ")][0][7:])
        res[name] = np.array(d["re"]) + 1j * np.array(d["im"])
    for name in ("route", "round5"):
        rel = np.abs(res[name] / res["general"] - 1)
        print("c128 dense %s vs general kernels: max rel diff %.2e (n = %d)" % (name, rel.max(), len(rel)))
        assert rel.max() < 2e-9


def test_real_rank_c4_energy_vs_oracle_golden():
    """the XXZ local energy of the tiled real state at C4 (12x12, D = 8, chi = 32) on the device against the
    float64 oracle at n = 16 (round 5 had this in the bench line only, at n = 4).  The oracle values are a committed fixture
    (tests/golden/c4_real_energy_golden.json, made by scripts/make_c4_energy_golden.py: oracle/epool.py, 40 s on 8 cores); energies
    and amplitude RATIOS do not depend on the overall scale of the state.  Tolerances: north_star's 1e-6 on the f32 energy (measured
    2.2e-7), 5e-8 on the f64 energy (5.4e-9); SURVEY 8(d)'s 1e-5 on the f32 amplitudes (ratios: 2e-5; measured 4.7e-6) and 1e-6 on the
    f64 amplitude ratios (measured 2.3e-7: a configuration whose truncation boundary sigma_chi ~ sigma_chi+1 is nearly degenerate
    amplifies the 1e-9 perturbation of the f64 mode's Gram-based factors a hundredfold, as it amplifies the 1e-7 of the f32 engine to
    1e-5 -- the same walkers carry the tail of both)."""
    import json
    from peps_amd import capi, hostapi
    g = json.load(open(os.path.join(os.path.dirname(FIXTURES), "c4_real_energy_golden.json")))
    L, D, chi = g["L"], g["D"], g["chi"]
    cfgs = np.array(g["configs"], dtype=np.int32)
    assert np.array_equal(cfgs, synthetic.make_configs_near_neel(L, len(cfgs), seed0=g["seed0"]))
    e_ref, r_ref = np.array(g["energy"]), np.array(g["psi_over_psi0"])
    flat = _state(L)
    for dt, tol_e, tol_a in ((capi.F32, 1e-6, 1e-5), (capi.F64, 5e-8, 5e-7)):
        amps, en, _, _ = hostapi.energy_and_holes(flat, cfgs, chi, "xxz", (1.0, 1.0, 0.0), False, dt)
        err_e = np.max(np.abs(en / e_ref - 1))
        err_a = np.max(np.abs((amps / amps[0]) / r_ref - 1))
        print("C4 real state, n = %d, %s: E_loc max rel err %.2e, psi ratio max rel err %.2e" % (len(cfgs), "f32" if dt == capi.F32 else "f64", err_e, err_a))
        assert err_e < tol_e, (dt, err_e)
        assert err_a < 2 * tol_a, (dt, err_a)      # (a ratio of two amplitudes: twice the single-amplitude tolerance)


def test_round6_routes_against_their_predecessors_and_the_oracle():
    """The route switches the library keeps (read once per process: subprocesses).  The three route changes of round 6: PEPSGPU_PIVOT_CHOL=0 (full
    factorisation of the truncation Gram in the natural order instead of the pivoted one capped at 56 rows), PEPSGPU_ROWS_QR=0 (polishing
    Jacobi + select + Newton-Schulz instead of the float64 Cholesky-QR of the projected rows), PEPSGPU_TRI=0 (the chained contraction
    and M = R Tt multiply the zero blocks of the triangular carry).  8x8 at C4's bond dimensions (carry of 256 rows, dense route):
    every variant within SURVEY 8(d)'s 1e-5 of the float64 oracle; the triangular form skips exact zeros, so it is BIT-identical."""
    import json
    import subprocess
    import sys
    L, D, chi = 8, 8, 32
    flat = _state(L, D)
    cfgs = synthetic.make_configs_near_neel(L, 4, seed0=211)
    ref_a, _ = _oracle(flat, cfgs, chi)
    code = r"""
import json, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from peps_amd import capi
flat, cfgs = np.load(sys.argv[2])["flat"], np.load(sys.argv[2])["cfgs"]
ctx = capi.Context(8, 8, 8, 2, 32, dtype=capi.F32, max_walkers=len(cfgs))
ctx.state_upload(flat); ctx.set_configs(cfgs)
a = ctx.evaluate_amplitude()
assert np.all(ctx.walker_flags() == 0)
print(json.dumps([float(x) for x in a]))
"""
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as td:
        f = os.path.join(td, "in.npz")
        np.savez(f, flat=flat, cfgs=cfgs)
        res = {}
        variants = (("default", {}), ("no_pivot", {"PEPSGPU_PIVOT_CHOL": "0"}), ("no_rows_qr", {"PEPSGPU_ROWS_QR": "0"}), ("no_tri", {"PEPSGPU_TRI": "0"}),
                    # the other route switches the library keeps: each selects a working path
                    ("no_mfma", {"PEPSGPU_NO_MFMA": "1"}), ("static_shapes", {"PEPSGPU_NO_RANK_ADAPT": "1"}), ("precise_never", {"PEPSGPU_PRECISE": "0"}),
                    ("precise_always", {"PEPSGPU_PRECISE": "2"}), ("no_midroute", {"PEPSGPU_NO_MIDROUTE": "1"}), ("f64_grams", {"PEPSGPU_NO_I8_GRAM": "1"}))
        for name, env in variants:
            r = subprocess.run([sys.executable, "-c", code, root, f], env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
            assert r.returncode == 0, r.stderr[-2000:]
            res[name] = np.array(json.loads(r.stdout.strip().splitlines()[-1]))
    for name, a in res.items():
        err = np.max(np.abs(a / ref_a - 1))
        print("8x8 D = 8 chi = 32 real state, f32, %s: max rel err vs oracle %.2e" % (name, err))
        assert err < (3e-5 if name == "precise_never" else 1e-5), (name, err)      # (without the float64-accumulating forms: round 3's error level)
    assert np.array_equal(res["default"], res["no_tri"])
    assert not np.array_equal(res["default"], res["no_pivot"]) and not np.array_equal(res["default"], res["no_rows_qr"])     # ... and the switches switch
