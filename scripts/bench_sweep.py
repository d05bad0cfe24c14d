"""Secondary throughput figures of SURVEY 8(d): MC sweeps/s and energy-samples/s at C4 (host layer)."""
import sys, time, json
import numpy as np
sys.path.insert(0, '.')
from peps_amd import capi, hostapi, synthetic
name = sys.argv[1] if len(sys.argv) > 1 else "C4"
nw = int(sys.argv[2]) if len(sys.argv) > 2 else 256
L, D, chi, model = synthetic.CONFIGS[name]
sitps = synthetic.make_sitps(L, D)
ctx = capi.Context(L, L, D, 2, chi, dtype=capi.F32, max_walkers=1)
ctx.state_upload(synthetic.sitps_to_flat(sitps, D)); ctx.set_configs(synthetic.checkerboard(L)[None])
sitps = synthetic.rescale_sitps(sitps, float(ctx.evaluate_amplitude()[0])); del ctx
flat = synthetic.sitps_to_flat(sitps, D)
cfgs = synthetic.make_configs(L, nw, "heisenberg")
seeds = np.arange(nw, dtype=np.uint64) + 100
res = {}
for upd in ("exchange", "fullspace"):
    hostapi.mc_sweeps(flat, cfgs, seeds, chi, upd, 1, 0)
    t0 = time.time(); out_cfg, amps, rates = hostapi.mc_sweeps(flat, cfgs, seeds, chi, upd, 2, 0); dt = time.time() - t0
    res["sweeps_per_s_" + upd] = 2 * nw / dt
    res["accept_" + upd] = float(rates.mean())
hostapi.energy_and_holes(flat, cfgs, chi, "xxz", (1.0, 1.0, 0.0), True, 0)
t0 = time.time(); a, e, h, psi = hostapi.energy_and_holes(flat, cfgs, chi, "xxz", (1.0, 1.0, 0.0), True, 0); dt = time.time() - t0
res["energy_and_holes_samples_per_s"] = nw / dt
res["psi_consistency_max_rel_spread"] = float(np.max(np.abs(psi / psi[0] - 1)))
t0 = time.time(); packed, _, acc = hostapi.mc_energy_grad_partial(flat, cfgs, seeds, chi, "exchange", "xxz", (1.0, 1.0, 0.0), 0, 2, 0); dt = time.time() - t0
res["vmc_samples_per_s_sweep_plus_energy_grad_device_holes"] = 2 * nw / dt
e, g = hostapi.exact_sum_finish(packed, flat.shape)
res["mc_energy_per_site"] = e / (L * L)
res["workload"] = name; res["walkers"] = nw
print(json.dumps(res))
