"""probe: carries beyond 256 rows (D chi > 256) against the oracle on a small lattice; usage: bigbond_probe.py L D chi [noise]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from peps_amd import capi, synthetic
from oracle import vmc
from oracle.bmps import BMPSTruncateParams
L, D, chi = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
noise = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
sitps = synthetic.make_sitps(L, D, noise=noise)
flat = synthetic.sitps_to_flat(sitps, D, np.float64)
cfgs = synthetic.make_configs(L, 3, "heisenberg", seed0=5)
tp = BMPSTruncateParams.SVD(chi, chi, 0.0)
ref = np.array([vmc.TPSWaveFunctionComponent(sitps, c, tp).amplitude for c in cfgs])
for dt in (capi.F32, capi.F64):
    try:
        ctx = capi.Context(L, L, D, 2, chi, dtype=dt, max_walkers=len(cfgs))
        ctx.state_upload(flat); ctx.set_configs(cfgs)
        a = ctx.evaluate_amplitude()
        print("%dx%d D=%d chi=%d (D chi = %d) dtype %d: max rel err %.2e flags %d" % (L, L, D, chi, D * chi, dt, np.max(np.abs(a / ref - 1)), int(np.sum(ctx.walker_flags() != 0))), a, ref)
        ctx.close()
    except Exception as e:
        print("dtype", dt, "raised", repr(e)[:300])
