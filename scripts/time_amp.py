"""Kernel-category timing of EvaluateAmplitude at C4 regardless of the result (timing experiments)."""
import sys, json
sys.path.insert(0, '.')
import numpy as np
from peps_amd import capi, synthetic
nw = int(sys.argv[1]) if len(sys.argv) > 1 else 512
L, D, chi, _ = synthetic.CONFIGS["C4"]
sitps = synthetic.make_sitps(L, D)
ctx = capi.Context(L, L, D, 2, chi, dtype=capi.F32, max_walkers=nw)
ctx.state_upload(synthetic.sitps_to_flat(sitps, D))
cfg = synthetic.make_configs(L, nw, "heisenberg")
ctx.set_configs(cfg); ctx.evaluate_amplitude()
ctx.profile_enable(True); ctx.profile_read()
for _ in range(2):
    ctx.set_configs(cfg); ctx.evaluate_amplitude()
p = ctx.profile_read()
print({k: round(v["ms"], 2) for k, v in p.items() if v["launches"]})
