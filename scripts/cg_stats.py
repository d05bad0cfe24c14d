"""Where colgram_dense_kernel spends its time on the full_rank state (PEPSGPU_CG_STATS=1): K, live columns, ticks per phase."""
import os, sys, ctypes as C
os.environ["PEPSGPU_CG_STATS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from peps_amd import capi, synthetic
L, D, chi, _ = synthetic.CONFIGS["C4"]
nw = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
sitps = synthetic.make_sitps(L, D, noise=1.0)
ctx = capi.Context(L, L, D, 2, chi, dtype=capi.F32, max_walkers=nw)
ctx.state_upload(synthetic.sitps_to_flat(sitps, D, np.float64))
ctx.set_configs(synthetic.checkerboard(L)[None])
psi_ref = float(ctx.evaluate_amplitude()[0])
ctx.state_upload(synthetic.sitps_to_flat(synthetic.rescale_sitps(sitps, psi_ref), D, np.float64))
out = (C.c_double * 16)()
for k in range(3):
    ctx.set_configs(synthetic.make_configs(L, nw, "heisenberg", seed0=100 + k))
    ctx.evaluate_amplitude()
    capi.lib().pepsgpu_diag_cg_stats(out)
    v = list(out)
    nblk = max(v[0], 1)
    print("step", k, "blocks", int(v[0]), "K %.1f ncols %.1f live_out %.1f" % (v[1] / nblk, v[2] / nblk, v[6] / nblk),
          "us per block: gram %.1f chol %.1f out %.1f" % (v[3] / nblk / 100, v[4] / nblk / 100, v[5] / nblk / 100),
          "chol phases: diag %.1f subst %.1f update %.1f" % (v[7] / nblk / 100, v[8] / nblk / 100, v[9] / nblk / 100),
          "live columns <=64 / <=80 / <=96 / <=128: %s" % [round(x / nblk, 3) for x in v[12:16]])
