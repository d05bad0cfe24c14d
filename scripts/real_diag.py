"""Diagnostics of the real_rank leg: rows kept by the compressions of the truncation route at every site (16 walkers)."""
import os, sys
os.environ["PEPSGPU_DEBUG_SWEEPS"] = "1"
os.environ["PEPSGPU_DEBUG_VERBOSE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from peps_amd import capi, hostapi, synthetic
L, D, chi = 12, 8, 32
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
flat = synthetic.tile_flat_state(hostapi.load_sitps(os.path.join(ROOT, "tests/golden/ref_fixtures", synthetic.REAL_FIXTURE), 8), L)
ctx = capi.Context(L, L, D, 2, chi, dtype=capi.F32, max_walkers=16)
ctx.state_upload(flat)
ctx.set_configs(synthetic.checkerboard(L)[None].repeat(16, 0))
psi = ctx.evaluate_amplitude()[0]
flat = flat * abs(psi) ** (-1.0 / (L * L))
ctx.state_upload(flat)
ctx.set_configs(synthetic.make_configs_near_neel(L, 16))
print(ctx.evaluate_amplitude())
print(ctx.stats())
