"""Round 6 probe: random small shapes (rows x cols, D, chi, noise) through evaluate_amplitude against the NumPy oracle, f32 and f64 --
looks for silently wrong corners of the shape space (found: D chi > 256, see tests/test_gpu_fullrank.py).  usage: fuzz_probe.py [n] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from peps_amd import capi, synthetic
from oracle import vmc
from oracle.bmps import BMPSTruncateParams
n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for case in range(n):
    L = int(rng.integers(4, 8))
    D = int(rng.integers(2, 8))
    chi = int(rng.integers(D, min(D * D + 6, 70)))
    noise = float(rng.choice([0.1, 0.5, 1.0]))
    sitps = synthetic.make_sitps(L, D, noise=noise)
    flat = synthetic.sitps_to_flat(sitps, D, np.float64)
    cfgs = synthetic.make_configs(L, 3, "heisenberg", seed0=int(rng.integers(1, 1000)))
    tp = BMPSTruncateParams.SVD(chi, chi, 0.0)
    t0 = time.time()
    ref = np.array([vmc.TPSWaveFunctionComponent(sitps, c, tp).amplitude for c in cfgs])
    line = "L=%d D=%d chi=%d noise=%.1f (oracle %.0f s):" % (L, D, chi, noise, time.time() - t0)
    for dt, tol in ((capi.F32, 2e-5), (capi.F64, 1e-8)):
        try:
            ctx = capi.Context(L, L, D, 2, chi, dtype=dt, max_walkers=len(cfgs))
            ctx.state_upload(flat); ctx.set_configs(cfgs)
            a = ctx.evaluate_amplitude()
            err = float(np.max(np.abs(a / ref - 1)))
            fl = int(np.sum(ctx.walker_flags() != 0))
            ctx.close()
            ok = err < tol and fl == 0
        except Exception as e:
            err, fl, ok = -1.0, -1, False
            line += " EXC %s" % repr(e)[:120]
        bad += not ok
        line += "  %s err %.1e flags %d%s" % ("f32" if dt == capi.F32 else "f64", err, fl, "" if ok else "  <== FAIL")
    print(line, flush=True)
print("failures:", bad)
