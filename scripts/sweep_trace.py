"""the program rocprofv3 traces for the sweep profile: C4 headline state (or argv[2] == real), argv[1] walkers, 1 + 2 sweeps of the
exchange updater through the C++ host layer (device slice path)"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from peps_amd import hostapi, synthetic
nw = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
real = len(sys.argv) > 2 and sys.argv[2] == "real"
L, D, chi, _ = synthetic.CONFIGS["C4"]
if real:
    from conftest import FIXTURES
    flat = synthetic.tile_flat_state(hostapi.load_sitps(os.path.join(FIXTURES, synthetic.REAL_FIXTURE), 8), L) * 0.5
    cfgs = synthetic.make_configs_near_neel(L, nw, seed0=307)
else:
    flat = synthetic.sitps_to_flat(synthetic.make_sitps(L, D), D)
    cfgs = synthetic.make_configs(L, nw, "heisenberg")
hostapi.mc_sweeps(flat, cfgs, np.arange(nw, dtype=np.uint64) + 100, chi, "exchange", 3, 0)
