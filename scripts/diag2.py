import numpy as np, sys, time, os
sys.path.insert(0, '.')
from peps_amd import capi, synthetic
from oracle import ising, vmc
from oracle.bmps import *
for L, chi in [(12, 30), (12, 10)]:
    tn, lognorm, beta = ising.build_ising_tn(L, L)
    sitps = [[[tn((r, c))] for c in range(L)] for r in range(L)]
    comp = vmc.TPSWaveFunctionComponent(sitps, np.zeros((L, L), int), BMPSTruncateParams.SVD(chi, chi, 0.0))
    for dt in (capi.F64, capi.F32):
        ctx = capi.Context(L, L, 2, 1, chi, dtype=dt, max_walkers=1)
        ctx.state_upload(synthetic.sitps_to_flat(sitps, 2, np.float64))
        ctx.set_configs(np.zeros((1, L, L), dtype=np.int32))
        a = ctx.evaluate_amplitude()[0]
        print("ising", L, chi, dt, "dev/oracle-1", a / comp.amplitude - 1, ctx.stats())
for name in ("C2", "C3", "C4"):
    L, D, chi, model = synthetic.CONFIGS[name]
    sitps = synthetic.make_sitps(L, D)
    for dt in (capi.F32, capi.F64):
        nw = 16
        cfgs = synthetic.make_configs(L, nw, "heisenberg")
        ctx = capi.Context(L, L, D, 2, chi, dtype=dt, max_walkers=nw)
        ctx.state_upload(synthetic.sitps_to_flat(sitps, D, np.float64))
        ctx.set_configs(cfgs)
        t0 = time.time(); a = ctx.evaluate_amplitude(); t1 = time.time()
        ctx.set_configs(cfgs)
        t2 = time.time(); a2 = ctx.evaluate_amplitude(); t3 = time.time()
        print(name, "f32" if dt == 0 else "f64", "nw", nw, "first %.3fs second %.3fs -> %.1f amp/s" % (t1 - t0, t3 - t2, nw / (t3 - t2)), ctx.stats(), a[:3])
        if name != "C4":
            tp = BMPSTruncateParams.SVD(chi, chi, 0.0)
            t0 = time.time()
            ref = np.array([vmc.TPSWaveFunctionComponent(sitps, c, tp).amplitude for c in cfgs[:4]])
            print("   oracle %.2fs/amp" % ((time.time() - t0) / 4), "max rel err", np.max(np.abs(a[:4] / ref - 1)))
        del ctx
