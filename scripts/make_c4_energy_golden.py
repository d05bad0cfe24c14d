"""Generates tests/golden/c4_real_energy_golden.json (round 6): XXZ local energies of n = 16 near-Neel
configurations of the tiled real state at C4 (12x12, D = 8, chi = 32, SVD(chi, chi, 0)) from the float64 NumPy oracle
(oracle/epool.py -> oracle/vmc.py: SquareNNNModelEnergySolver::CalEnergyAndHolesImpl restated), plus the amplitude ratios
psi(cfg) / psi(cfg_0).  Both are invariant under the overall scale of the state, so the GPU test may rescale it as it likes.
Runs on the CPU only (about 20 minutes on 8 cores): python scripts/make_c4_energy_golden.py"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from peps_amd import hostapi, synthetic
from oracle import epool
L, D, chi, n, seed = 12, 8, 32, 16, 4242
flat = synthetic.tile_flat_state(hostapi.load_sitps(os.path.join(ROOT, "tests/golden/ref_fixtures", synthetic.REAL_FIXTURE), 8), L)
cfgs = synthetic.make_configs_near_neel(L, n, seed0=seed)
t0 = time.time()
h = epool.start(flat, cfgs, chi, (1.0, 1.0, 0.0), nprocs=8, blas_threads=1)
e, a, sec = epool.collect(h, timeout=7200)
out = {"what": "tiled real state (tps_square_heisenberg4x4D8Double by position class) at 12x12, chi = 32, XXZ (1, 1, 0): oracle E_loc and psi ratios",
       "L": L, "D": D, "chi": chi, "seed0": seed, "configs": cfgs.tolist(), "energy": [float(x) for x in e],
       "psi_over_psi0": [float(x / a[0]) for x in a], "oracle_seconds": sec}
json.dump(out, open(os.path.join(ROOT, "tests/golden/c4_real_energy_golden.json"), "w"))
print("done in", round(time.time() - t0), "s", e[:3], a[:3])
