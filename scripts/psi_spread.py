"""psi along every row / column route of CalEnergyAndHoles for the first nw synthetic configurations: worst walker,
its per-route values in f32 and f64"""
import sys, json
import numpy as np
sys.path.insert(0, '.')
from peps_amd import capi, hostapi, synthetic
nw = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
wl = sys.argv[2] if len(sys.argv) > 2 else "C4"
noise = float(sys.argv[3]) if len(sys.argv) > 3 else 0.1
L, D, chi, model = synthetic.CONFIGS[wl]
sitps = synthetic.make_sitps(L, D, noise=noise)
ctx = capi.Context(L, L, D, 2, chi, dtype=capi.F32, max_walkers=1)
ctx.state_upload(synthetic.sitps_to_flat(sitps, D)); ctx.set_configs(synthetic.checkerboard(L)[None])
sitps = synthetic.rescale_sitps(sitps, float(ctx.evaluate_amplitude()[0])); del ctx
flat = synthetic.sitps_to_flat(sitps, D)
cfgs = synthetic.make_configs(L, nw, "heisenberg")
print("workload", wl, "noise", noise)
a, e, h, psi = hostapi.energy_and_holes(flat, cfgs, chi, "xxz", (1.0, 1.0, 0.0), False, 0)
spread = np.max(np.abs(psi / psi[0] - 1), axis=0)
w = int(np.argmax(spread))
print("worst walker", w, "spread", spread[w], "second", np.sort(spread)[-2], "n>3e-5:", int(np.sum(spread > 3e-5)), "of", nw, "outliers", [(int(i), float(spread[i]), int(np.argmax(np.abs(psi[:, i] / np.median(psi[:, i]) - 1)))) for i in np.where(spread > 3e-5)[0][:12]])
print("f32 routes", psi[:, w])
sub = cfgs[w:w + 1]
a1, e1, h1, psi1 = hostapi.energy_and_holes(flat, sub, chi, "xxz", (1.0, 1.0, 0.0), False, 0)
print("alone f32 ", psi1[:, 0])
a2, e2, h2, psi2 = hostapi.energy_and_holes(flat, sub, chi, "xxz", (1.0, 1.0, 0.0), False, 1)
print("alone f64 ", psi2[:, 0])
np.save("gpurun_out/worst_cfg.npy", sub)
