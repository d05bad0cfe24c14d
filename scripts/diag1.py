import numpy as np, sys
sys.path.insert(0, '.')
from peps_amd import capi, synthetic
from oracle import ising, vmc
from oracle.bmps import *
from oracle.contractor import *
rng = np.random.default_rng(0)
for (m, ln) in [(64, 64), (100, 100), (128,128), (144, 144), (256, 256)]:
    r = min(m, ln)
    U, _ = np.linalg.qr(rng.standard_normal((m, r)))
    V, _ = np.linalg.qr(rng.standard_normal((ln, r)))
    for spec in (np.logspace(0, -5, r), np.linspace(1, 0.1, r)):
        M = ((U * spec) @ V.T)[None]
        for dt in (capi.F32, capi.F64):
            for fg in (False, True):
                Mo, Vt, S, sw = capi.diag_jacobi(dt, M, r, fg)
                sref = np.linalg.svd(M[0], compute_uv=False)
                print(m, ln, "graded" if spec[-1] < 1e-3 else "flat", "f32" if dt == 0 else "f64", "glob" if fg else "auto",
                      "sweeps", sw[0], "max rel err S", np.max(np.abs(S[0] - sref) / sref[0]), "orth", np.max(np.abs(Vt[0].astype(float) @ Vt[0].astype(float).T - np.eye(r))))
# Ising
for L, chi in [(4, 4), (6, 8), (8, 16), (12, 30), (12, 10)]:
    tn, lognorm, beta = ising.build_ising_tn(L, L)
    sitps = [[[tn((r, c))] for c in range(L)] for r in range(L)]
    comp = vmc.TPSWaveFunctionComponent(sitps, np.zeros((L, L), int), BMPSTruncateParams.SVD(chi, chi, 0.0))
    for dt in (capi.F64, capi.F32):
        ctx = capi.Context(L, L, 2, 1, chi, dtype=dt, max_walkers=1)
        ctx.state_upload(synthetic.sitps_to_flat(sitps, 2, np.float64))
        ctx.set_configs(np.zeros((1, L, L), dtype=np.int32))
        a = ctx.evaluate_amplitude()[0]
        print("ising", L, chi, dt, "dev", a, "oracle", comp.amplitude, "exact", ising.exact_contract(tn), "flags", ctx.walker_flags())
