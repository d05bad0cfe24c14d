"""Amplitude throughput of the complex element type (PEPSGPU_C128): fresh EvaluateAmplitude of random complex states.
PEPSGPU_NO_CPLX_MFMA=1 selects the VALU GEMMs of round 2 for an A/B."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from peps_amd import capi, synthetic
res = {}
for (L, D, chi, nw) in ((8, 4, 16, 512), (10, 6, 24, 128), (12, 8, 32, 64)):
    rng = np.random.default_rng(5)
    flat = np.zeros((L, L, 2, D, D, D, D), dtype=np.complex128)
    for r in range(L):
        for c in range(L):
            shp = synthetic.bond_dims(L, D, r, c)
            for s in range(2):
                t = (rng.uniform(0.2, 1.0, shp) * np.exp(2j * np.pi * rng.uniform(size=shp)) + 0.5 * np.exp(2j * np.pi * rng.uniform())) / D
                flat[r, c, s, :shp[0], :shp[1], :shp[2], :shp[3]] = t
    ctx = capi.Context(L, L, D, 2, chi, dtype=capi.C128, max_walkers=nw)
    ctx.state_upload(flat)
    cf = synthetic.make_configs(L, nw, "heisenberg")
    ctx.set_configs(cf); a0 = ctx.evaluate_amplitude()
    t0 = time.perf_counter()
    for k in range(2):
        ctx.set_configs(synthetic.make_configs(L, nw, "heisenberg", seed0=1000 * (k + 1))); a = ctx.evaluate_amplitude()
    dt = (time.perf_counter() - t0) / 2
    res["%dx%d_D%d_chi%d" % (L, L, D, chi)] = {"amp_per_s": nw / dt, "walkers": nw, "checksum": float(np.abs(a0).sum())}
    ctx.close()
print(json.dumps(res))
