"""MC sweeps/s and VMC samples/s on the tiled optimised state of the reference (the real_rank workload), host layer."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from peps_amd import capi, hostapi, synthetic
nw = int(sys.argv[1]) if len(sys.argv) > 1 else 512
L, D, chi = 12, 8, 32
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
flat = synthetic.tile_flat_state(hostapi.load_sitps(os.path.join(ROOT, "tests/golden/ref_fixtures", synthetic.REAL_FIXTURE), 8), L)
ctx = capi.Context(L, L, D, 2, chi, dtype=capi.F32, max_walkers=1)
ctx.state_upload(flat); ctx.set_configs(synthetic.checkerboard(L)[None])
flat = flat * abs(float(ctx.evaluate_amplitude()[0])) ** (-1.0 / (L * L)); ctx.close()
cfgs = synthetic.make_configs_near_neel(L, nw)
seeds = np.arange(nw, dtype=np.uint64) + 100
res = {"walkers": nw}
hostapi.mc_sweeps(flat, cfgs, seeds, chi, "exchange", 1, 0)
t0 = time.time(); out_cfg, amps, rates = hostapi.mc_sweeps(flat, cfgs, seeds, chi, "exchange", 2, 0); dt = time.time() - t0
res["sweeps_per_s_exchange"] = 2 * nw / dt; res["accept"] = float(rates.mean())
t0 = time.time(); a, e, h, psi = hostapi.energy_and_holes(flat, cfgs, chi, "xxz", (1.0, 1.0, 0.0), False, 0); dt = time.time() - t0
res["energy_samples_per_s"] = nw / dt
res["psi_consistency_max_rel_spread"] = float(np.max(np.abs(psi / psi[0] - 1)))
print(json.dumps(res))
