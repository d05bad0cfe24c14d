"""Where a Monte-Carlo sweep spends its time: marginal seconds per sweep of the C++ host layer's exchange
updater at C4 on the headline state (or --state real), device slice path against the per-bond hook path (PEPSHOST_NO_DEVICE_SWEEP),
at several walker counts; the amplitude rate of the same batch beside it (a sweep = 4 (L - 1) absorptions + the bond traces)."""
import argparse, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = r"""
import json, sys, time
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
from peps_amd import capi, hostapi, synthetic
nw, real, nsw = int(sys.argv[2]), sys.argv[3] == "real", int(sys.argv[4])
L, D, chi, _ = synthetic.CONFIGS["C4"]
if real:
    from conftest import FIXTURES
    import os
    f4 = hostapi.load_sitps(os.path.join(FIXTURES, synthetic.REAL_FIXTURE), 8)
    flat = synthetic.tile_flat_state(f4, L)
    cfgs = synthetic.make_configs_near_neel(L, nw, seed0=307)
else:
    flat = synthetic.sitps_to_flat(synthetic.make_sitps(L, D), D)
    cfgs = synthetic.make_configs(L, nw, "heisenberg")
ctx = capi.Context(L, L, D, 2, chi, dtype=capi.F32, max_walkers=1)
ctx.state_upload(flat); ctx.set_configs(synthetic.checkerboard(L)[None])
flat = flat * abs(float(ctx.evaluate_amplitude()[0])) ** (-1.0 / (L * L)); ctx.close()
seeds = np.arange(nw, dtype=np.uint64) + 100
ctx = capi.Context(L, L, D, 2, chi, dtype=capi.F32, max_walkers=nw)
ctx.state_upload(flat); ctx.set_configs(cfgs); ctx.evaluate_amplitude(); ctx.sync()
t0 = time.perf_counter(); ctx.set_configs(cfgs); ctx.evaluate_amplitude(); ctx.sync(); t_amp = time.perf_counter() - t0
ctx.close()
hostapi.mc_sweeps(flat, cfgs, seeds, chi, "exchange", 1, 0)
t0 = time.perf_counter(); hostapi.mc_sweeps(flat, cfgs, seeds, chi, "exchange", 1, 0); t1 = time.perf_counter() - t0
t0 = time.perf_counter(); _, _, r = hostapi.mc_sweeps(flat, cfgs, seeds, chi, "exchange", 1 + nsw, 0); tn = time.perf_counter() - t0
t0 = time.perf_counter(); hostapi.mc_energy_grad_partial(flat, cfgs, seeds, chi, "exchange", "xxz", (1.0, 1.0, 0.0), 0, 1, 0); v1 = time.perf_counter() - t0
t0 = time.perf_counter(); hostapi.mc_energy_grad_partial(flat, cfgs, seeds, chi, "exchange", "xxz", (1.0, 1.0, 0.0), 0, 1 + nsw, 0); vn = time.perf_counter() - t0
print(json.dumps({"walkers": nw, "amp_per_s": nw / t_amp, "s_per_amplitude_batch": t_amp, "s_per_sweep": (tn - t1) / nsw,
                  "sweeps_per_s": nsw * nw / (tn - t1), "sweep_over_4_amplitudes": (tn - t1) / nsw / (4 * t_amp),
                  "s_per_vmc_sample": (vn - v1) / nsw, "vmc_samples_per_s": nsw * nw / (vn - v1), "accept": float(np.mean(r))}))
"""
ap = argparse.ArgumentParser()
ap.add_argument("--walkers", default="8192,24576")
ap.add_argument("--state", default="synthetic")
ap.add_argument("--sweeps", type=int, default=2)
ap.add_argument("--paths", default="device,hook")
a = ap.parse_args()
for nw in a.walkers.split(","):
    for path in a.paths.split(","):
        env = dict(os.environ)
        if path == "hook":
            env["PEPSHOST_NO_DEVICE_SWEEP"] = "1"
        r = subprocess.run([sys.executable, "-c", WORKER, ROOT, nw, a.state, str(a.sweeps)], env=env, capture_output=True, text=True, timeout=3000)
        line = r.stdout.strip().splitlines()[-1] if r.returncode == 0 and r.stdout.strip() else json.dumps({"error": r.stderr[-800:]})
        print(json.dumps({"path": path, "state": a.state, **json.loads(line)}), flush=True)
