"""Round 6 prototype helper (CPU, checker side only): captures the truncation inputs (the m x (u k2) blocks whose SVD the reference takes,
bmps_impl.h:235-238) of bulk rows of the tiled real state at C4 from the NumPy oracle -> /tmp/proto/M.npz.  The device's M = R Tt is this
block up to an orthogonal change of the row basis; the prototype of the truncation numerics (scripts/proto_subspace.py) works on these."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from peps_amd import hostapi, synthetic
from oracle import vmc, tensor as T
from oracle.bmps import BMPSTruncateParams
L, D, chi = 12, 8, 32
flat = synthetic.tile_flat_state(hostapi.load_sitps(os.path.join(ROOT, "tests/golden/ref_fixtures", synthetic.REAL_FIXTURE), 8), L)
sitps = synthetic.flat_to_sitps(flat)
cfgs = synthetic.make_configs_near_neel(L, 2, seed0=307)
caught = []
orig = T.svd_trunc
def hook(a, ldims, trunc_err, dmin, dmax):
    m = a.reshape(int(np.prod(a.shape[:ldims])), -1)
    if m.shape[0] == 256 and m.shape[1] == 256:
        caught.append(m.copy())
    return orig(a, ldims, trunc_err, dmin, dmax)
T.svd_trunc = hook
tp = BMPSTruncateParams.SVD(chi, chi, 0.0)
for c in cfgs:
    comp = vmc.TPSWaveFunctionComponent(sitps, c, tp)
    print("amp", comp.amplitude, "blocks", len(caught))
np.savez_compressed("/tmp/proto/M.npz", M=np.stack(caught))
