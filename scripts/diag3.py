"""Per-launch Jacobi diagnostics at C4 (16 walkers): sweeps and live rows per absorbed site."""
import os, sys
os.environ["PEPSGPU_DEBUG_SWEEPS"] = "1"; os.environ["PEPSGPU_DEBUG_VERBOSE"] = "1"
sys.path.insert(0, '.')
import numpy as np
from peps_amd import capi, synthetic
L, D, chi, _ = synthetic.CONFIGS["C4"]
noise = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
sitps = synthetic.make_sitps(L, D, noise=noise)
ctx = capi.Context(L, L, D, 2, chi, dtype=capi.F32, max_walkers=16)
ctx.state_upload(synthetic.sitps_to_flat(sitps, D)); ctx.set_configs(synthetic.make_configs(L, 16, "heisenberg"))
for k in range(6):
    ctx.grow_bmps_step(capi.UP)
print(ctx.stats())
