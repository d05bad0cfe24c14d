"""What the profiling brackets (two HIP events per category launch) cost inside the timed region of bench.py: the headline
workload with ctx.profile_enable(True) against (False)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from peps_amd import capi, synthetic
L, D, chi, _ = synthetic.CONFIGS["C4"]
nw = int(sys.argv[1]) if len(sys.argv) > 1 else 49152
sitps = synthetic.make_sitps(L, D, noise=0.1)
ctx = capi.Context(L, L, D, 2, chi, dtype=capi.F32, max_walkers=nw)
ctx.state_upload(synthetic.sitps_to_flat(sitps, D, np.float64))
ctx.set_configs(synthetic.checkerboard(L)[None])
psi_ref = float(ctx.evaluate_amplitude()[0])
ctx.state_upload(synthetic.sitps_to_flat(synthetic.rescale_sitps(sitps, psi_ref), D, np.float64))
batches = [synthetic.make_configs(L, nw, "heisenberg", seed0=100 + k) for k in range(4)]
ctx.set_configs(batches[0]); ctx.evaluate_amplitude()
for rep in range(2):
    for on in (True, False):
        ctx.profile_enable(on)
        if on:
            ctx.profile_read()
        t0 = time.perf_counter()
        for k in range(1, 4):
            ctx.set_configs(batches[k]); ctx.evaluate_amplitude()
        dt = (time.perf_counter() - t0) / 3
        if on:
            ctx.profile_read()
        print("profiling", on, "ms per step %.2f" % (dt * 1e3), "amp/s %.0f" % (nw / dt))
ctx.profile_enable(False)
