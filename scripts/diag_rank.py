"""Per-launch live-row statistics of the chi-truncation blocks (PEPSGPU_DEBUG_VERBOSE) for a synthetic C4 state of given
noise, and for the reference's 4x4 D=8 fixture.  usage: python scripts/diag_rank.py <noise> [walkers]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PEPSGPU_DEBUG_SWEEPS"] = "1"
os.environ["PEPSGPU_DEBUG_VERBOSE"] = "1"
import numpy as np
from peps_amd import capi, synthetic, hostapi
noise = float(sys.argv[1]); nw = int(sys.argv[2]) if len(sys.argv) > 2 else 64
if noise < 0:
    flat = hostapi.load_sitps("tests/golden/ref_fixtures/tps_square_heisenberg4x4D8Double", 8)
    L, D, chi = 4, 8, 32
else:
    L, D, chi, _ = synthetic.CONFIGS["C4"]
    flat = synthetic.sitps_to_flat(synthetic.make_sitps(L, D, noise=noise), D)
ctx = capi.Context(L, L, D, 2, chi, dtype=capi.F32, max_walkers=nw)
ctx.state_upload(flat)
ctx.set_configs(synthetic.make_configs(L, nw, "heisenberg"))
a = ctx.evaluate_amplitude()
print(ctx.stats())
