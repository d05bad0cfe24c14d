"""Arena growth over repeated evaluations (all three compression schemes, both rank regimes): device bytes after warm-up must
stay flat."""
import sys, json
sys.path.insert(0, '.')
import numpy as np
from peps_amd import capi, synthetic
L, D, chi, kind = synthetic.CONFIGS["C3"]
out = {}
for noise in (0.1, 1.0):
    sitps = synthetic.make_sitps(L, D, noise=noise)
    flat = synthetic.sitps_to_flat(sitps, D)
    for name, scheme in (("svd", 0), ("var2", 1), ("var1", 2)):
        nw = 512
        ctx = capi.Context(L, L, D, 2, chi, dtype=capi.F32, max_walkers=nw, scheme=scheme, convergence_tol=1e-5, iter_max=2)
        ctx.state_upload(flat)
        sizes = []
        for it in range(8):
            cfg = synthetic.make_configs(L, nw, kind, seed0=7 + 1000 * it)
            ctx.set_configs(cfg); ctx.evaluate_amplitude(); ctx.sync()
            sizes.append(ctx.stats().get("device_bytes", 0))
        out["%s_noise%g" % (name, noise)] = [round(s / 1e6, 1) for s in sizes]
        ctx.close()
print(json.dumps(out, indent=0))
