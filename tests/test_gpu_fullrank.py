"""Parity of the DENSE route of the absorption (carry rank > 32: f64 Gram GEMM, blocked Cholesky, full-size Jacobi /
eigen truncation) at bulk shapes, end to end: device amplitude AND XXZ local energy against the float64 oracle on states of
full rank (i.i.d. random site tensors, make_sitps(noise=1.0)), where every C3 / C4 parity test on the SURVEY 8(d) synthetic
state only ever runs the <= 16-row kernels.  ctx.stats() proves which route ran (carry_live_max > 32)."""
import os

import numpy as np
import pytest

from oracle import vmc
from oracle.bmps import BMPSTruncateParams
from peps_amd import synthetic

pytestmark = pytest.mark.gpu
F32, F64 = 0, 1

CASES = [("C3 10x10 D=6 chi=24", 10, 6, 24), ("6x6 D=8 chi=32", 6, 8, 32)]


def _state(L, D):
    from peps_amd import capi
    sitps = synthetic.make_sitps(L, D, noise=1.0)
    ctx = capi.Context(L, L, D, 2, 8, dtype=capi.F64, max_walkers=1)
    ctx.state_upload(synthetic.sitps_to_flat(sitps, D))
    ctx.set_configs(synthetic.checkerboard(L)[None])
    psi = float(ctx.evaluate_amplitude()[0])
    ctx.close()
    return synthetic.rescale_sitps(sitps, psi)      # amplitudes O(1): keeps f32 ranges comfortable, nothing else


@pytest.mark.parametrize("name,L,D,chi", CASES)
@pytest.mark.parametrize("dt,tol_a,tol_e", [(F32, 1e-5, 1e-6), (F64, 1e-9, 1e-9)])
def test_full_rank_amplitude_and_energy_vs_oracle(name, L, D, chi, dt, tol_a, tol_e, monkeypatch):
    from peps_amd import capi, hostapi
    sitps = _state(L, D)
    flat = synthetic.sitps_to_flat(sitps, D)
    cfgs = synthetic.make_configs(L, 4, "heisenberg", seed0=101)
    tp = BMPSTruncateParams.SVD(chi, chi, 0.0)
    model = vmc.SquareSpinOneHalfXXZModelOBC(1.0, 1.0, 0.0)
    ref_a, ref_e = [], []
    for c in cfgs:
        comp = vmc.TPSWaveFunctionComponent(sitps, c, tp)
        ref_a.append(comp.amplitude)
        ref_e.append(model.CalEnergyAndHoles(sitps, comp, False)[0])
    ref_a, ref_e = np.array(ref_a), np.array(ref_e)
    # which route: largest live carry of any walker (diagnostics are read back only with PEPSGPU_DEBUG_SWEEPS=1)
    monkeypatch.setenv("PEPSGPU_DEBUG_SWEEPS", "1")
    ctx = capi.Context(L, L, D, 2, chi, dtype=dt, max_walkers=len(cfgs))
    monkeypatch.delenv("PEPSGPU_DEBUG_SWEEPS")
    ctx.state_upload(flat)
    ctx.set_configs(cfgs)
    amps = ctx.evaluate_amplitude()
    st = ctx.stats()
    ctx.close()
    assert st["carry_live_max"] > 32, st            # the dense (rank > 32) route was taken
    assert np.max(np.abs(amps / ref_a - 1)) < tol_a, (amps, ref_a)
    a2, en, _, psi = hostapi.energy_and_holes(flat, cfgs, chi, "xxz", (1.0, 1.0, 0.0), False, dt)
    assert np.max(np.abs(a2 / ref_a - 1)) < tol_a
    assert np.max(np.abs(en / ref_e - 1)) < tol_e, (en, ref_e)


def test_full_rank_route_consistency_c4_batch():
    """full-size property no oracle sample can afford: 256 C4 walkers on the full-rank state, f32 against the f64 device mode
    (pinned to the oracle above at 1e-9), every walker within 1e-5; and no walker flagged."""
    from peps_amd import capi
    L, D, chi, _ = synthetic.CONFIGS["C4"]
    sitps = _state(L, D)
    flat = synthetic.sitps_to_flat(sitps, D)
    cfgs = synthetic.make_configs(L, 256, "heisenberg", seed0=303)
    amps = {}
    for dt in (capi.F32, capi.F64):
        ctx = capi.Context(L, L, D, 2, chi, dtype=dt, max_walkers=len(cfgs))
        ctx.state_upload(flat)
        ctx.set_configs(cfgs)
        amps[dt] = ctx.evaluate_amplitude()
        assert np.all(ctx.walker_flags() == 0)
        ctx.close()
    rel = np.abs(amps[capi.F32] / amps[capi.F64] - 1)
    assert np.max(rel) < 1e-5, (int(np.argmax(rel)), float(np.max(rel)))


def test_missed_rank_hint_redoes_the_absorption():
    """Performance hints never decide results: with a WRONG hint forced (PEPSGPU_FORCE_ROWS_CAP=16: 'no walker has more than
    16 live carry rows', on a state whose carry has many more) the Jacobi size classes above it are not launched, the live
    counts read back at the end of the absorption expose the miss, and the absorption is redone without hints: amplitudes as
    in the reference run, and the redo counter says it happened.  (The toggle is read once per process: child interpreter.)"""
    import json
    import subprocess
    import sys
    code = r'''
import json, numpy as np
from peps_amd import capi, synthetic
L, D, chi = 6, 6, 24
sitps = synthetic.make_sitps(L, D, noise=1.0)
cfgs = synthetic.make_configs(L, 6, "heisenberg", seed0=5)
ctx = capi.Context(L, L, D, 2, chi, dtype=capi.F32, max_walkers=len(cfgs))
ctx.state_upload(synthetic.sitps_to_flat(sitps, D))
ctx.set_configs(cfgs)
a = ctx.evaluate_amplitude()
st = ctx.stats()
print(json.dumps({"amps": [float(x) for x in a], "redone": st["absorptions_redone"], "live_max": st["carry_live_max"]}))
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for name, env in (("plain", {}), ("forced", {"PEPSGPU_FORCE_ROWS_CAP": "16"})):
        e = dict(os.environ, PEPSGPU_DEBUG_SWEEPS="1", PEPSGPU_NO_MIDROUTE="1", **env)   # (the mid route has its own size classes)
        r = subprocess.run([sys.executable, "-c", code], cwd=root, env=e, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[name] = json.loads(r.stdout.strip().splitlines()[-1])
    assert outs["forced"]["live_max"] > 16                       # the hint WAS wrong
    assert outs["forced"]["redone"] > outs["plain"]["redone"]    # ... and the absorptions were redone
    a, b = np.array(outs["plain"]["amps"]), np.array(outs["forced"]["amps"])
    assert np.max(np.abs(b / a - 1)) < 2e-5


def test_missed_fallback_hint_redoes_the_absorption():
    """The same for the hint of round 3 that skips the Gram + Cholesky launches behind the fused factor (a row whose carry stayed
    at <= 24 rows): forced on a full-rank state (PEPSGPU_FORCE_SKIP_FALLBACK=1), the walkers the fused factor flags are left
    without a factor, the flags read back at the end of the absorption expose it and the absorption is redone with every launch."""
    import json
    import subprocess
    import sys
    code = r'''
import json, numpy as np
from peps_amd import capi, synthetic
L, D, chi = 8, 6, 24
sitps = synthetic.make_sitps(L, D, noise=1.0)
cfgs = synthetic.make_configs(L, 8, "heisenberg", seed0=9)
ctx = capi.Context(L, L, D, 2, chi, dtype=capi.F32, max_walkers=len(cfgs))
ctx.state_upload(synthetic.sitps_to_flat(sitps, D))
ctx.set_configs(cfgs)
a = ctx.evaluate_amplitude()
ctx.set_configs(cfgs[::-1].copy())
b = ctx.evaluate_amplitude()
st = ctx.stats()
flags = ctx.walker_flags()
# the C++ host layer on the same state (TPSWaveFunctionComponent::EvaluateAmplitude throws 'Empty tensor' on a flagged walker)
from peps_amd import hostapi
_, e, _, _ = hostapi.energy_and_holes(synthetic.sitps_to_flat(sitps, D), cfgs[:4], chi, model="xxz", params=(1.0, 1.0, 0.0), holes=False, dtype=0)
print(json.dumps({"amps": [float(x) for x in a] + [float(x) for x in b[::-1]], "redone": st["absorptions_redone"], "live_max": st["carry_live_max"],
                  "flags": [int(x) for x in flags], "host_e": [float(x) for x in np.atleast_1d(e)]}))
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for name, env in (("plain", {}), ("forced", {"PEPSGPU_FORCE_SKIP_FALLBACK": "1"})):
        e = dict(os.environ, PEPSGPU_DEBUG_SWEEPS="1", **env)
        r = subprocess.run([sys.executable, "-c", code], cwd=root, env=e, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[name] = json.loads(r.stdout.strip().splitlines()[-1])
    assert outs["forced"]["live_max"] > 32                       # walkers beyond what the fused factor covers: the hint WAS wrong
    assert outs["forced"]["redone"] > outs["plain"]["redone"]    # ... and the absorptions were redone
    a, b = np.array(outs["plain"]["amps"]), np.array(outs["forced"]["amps"])
    assert np.all(np.isfinite(b)) and np.max(np.abs(b / a - 1)) < 2e-5
    # the failed attempt must not leave its sticky walker flags behind (a valid configuration was reported as 'Empty tensor')
    assert not any(outs["forced"]["flags"]) and not any(outs["plain"]["flags"])
    assert np.allclose(outs["forced"]["host_e"], outs["plain"]["host_e"], rtol=1e-4)


@pytest.mark.parametrize("L,D,chi", [(8, 7, 42), (8, 5, 60), (8, 8, 36)])
def test_carry_of_more_than_256_columns(L, D, chi):
    """D chi > 256 on a state of full rank (the carry really fills its 294 / 300 / 288 columns): amplitudes against the float64 oracle,
    f32 and f64.  Round 6 found this combination broken since round 3 -- launch_chol_upper handed every order >= 48 to the blocked
    Cholesky, whose four waves cover 256 columns, so the columns beyond were never updated: zeros, flags, or a silently wrong f64
    amplitude (2.7e-3 at D = 8, chi = 36).  No BASELINE config goes there (C4 is D chi = 256 exactly), the C ABI accepts it."""
    from peps_amd import capi
    sitps = synthetic.make_sitps(L, D, noise=1.0)
    flat = synthetic.sitps_to_flat(sitps, D, np.float64)
    cfgs = synthetic.make_configs(L, 3, "heisenberg", seed0=5)
    tp = BMPSTruncateParams.SVD(chi, chi, 0.0)
    ref = np.array([vmc.TPSWaveFunctionComponent(sitps, c, tp).amplitude for c in cfgs])
    for dt, tol in ((capi.F32, 1e-5), (capi.F64, 1e-9)):
        # (sites of more than 256 columns run with static shapes: the rank statistics of the adaptive kernels do not see them)
        ctx = capi.Context(L, L, D, 2, chi, dtype=dt, max_walkers=len(cfgs))
        ctx.state_upload(flat)
        ctx.set_configs(cfgs)
        a = ctx.evaluate_amplitude()
        assert np.all(ctx.walker_flags() == 0)
        ctx.close()
        assert np.max(np.abs(a / ref - 1)) < tol, (dt, a, ref)
