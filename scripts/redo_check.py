import sys; sys.path.insert(0,'.')
import numpy as np
from peps_amd import capi, synthetic
L, D, chi, kind = synthetic.CONFIGS["C4"]
for noise, nw in ((0.1, 16384), (1.0, 2048)):
    sitps = synthetic.make_sitps(L, D, noise=noise)
    ctx = capi.Context(L, L, D, 2, chi, dtype=capi.F32, max_walkers=nw)
    ctx.state_upload(synthetic.sitps_to_flat(sitps, D))
    for it in range(3):
        ctx.set_configs(synthetic.make_configs(L, nw, kind, seed0=7 + 100000 * it)); ctx.evaluate_amplitude()
    st = ctx.stats(); print(noise, nw, "absorptions", st["absorptions"], "redone", st["absorptions_redone"])
    ctx.close()
