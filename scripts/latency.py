"""Latency of one fresh EvaluateAmplitude of the headline state at small batches (ms per call, mean of the timed calls)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from peps_amd import capi, synthetic
L, D, chi, _ = synthetic.CONFIGS["C4"]
sitps = synthetic.make_sitps(L, D, noise=0.1)
ctx = capi.Context(L, L, D, 2, chi, dtype=capi.F32, max_walkers=1)
ctx.state_upload(synthetic.sitps_to_flat(sitps, D, np.float64))
ctx.set_configs(synthetic.checkerboard(L)[None])
psi_ref = float(ctx.evaluate_amplitude()[0]); ctx.close()
flat = synthetic.sitps_to_flat(synthetic.rescale_sitps(sitps, psi_ref), D, np.float64)
out = []
for nw in [int(x) for x in (sys.argv[1:] or ["1", "64", "1024", "2048", "4096"])]:
    ctx = capi.Context(L, L, D, 2, chi, dtype=capi.F32, max_walkers=nw)
    ctx.state_upload(flat)
    batches = [synthetic.make_configs(L, nw, "heisenberg", seed0=100 + k) for k in range(8)]
    for k in range(3):
        ctx.set_configs(batches[k]); a0 = ctx.evaluate_amplitude()
    t0 = time.perf_counter()
    for k in range(3, 8):
        ctx.set_configs(batches[k]); a = ctx.evaluate_amplitude()
    dt = (time.perf_counter() - t0) / 5
    out.append("%d: %.2f ms" % (nw, dt * 1e3))
    ctx.close()
print("PEPSGPU_SMALL_BATCH=%s" % os.environ.get("PEPSGPU_SMALL_BATCH", "default"), "  ".join(out), "checksum %.6e" % float(np.sum(np.abs(a))))
