"""Throughput of BASELINE config C5 (8x8 spinless t-V, fZ2-graded, D=6, chi=24) on one GPU: fresh amplitudes/s
through the sign-decorated path (peps_amd/fermion.py), f32 and f64, and E_loc samples/s (C++ host layer)."""
import sys, time, json
sys.path.insert(0, '.')
import numpy as np
from peps_amd import capi, fermion, hostapi
nw = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
L, D, chi = 8, 6, 24
st = fermion.random_even_state(L, L, D, seed=11)
rng = np.random.default_rng(1)
cfgs = np.stack([rng.permutation(np.r_[np.zeros(32, dtype=int), np.ones(32, dtype=int)]).reshape(L, L) for _ in range(nw)])
res = {"workload": "C5", "walkers": nw}
for name, dt in (("f32", capi.F32), ("f64", capi.F64)):
    n = nw if dt == capi.F32 else nw // 4
    ctx = capi.Context(L, L, D, 4 * st.d, chi, dtype=dt, max_walkers=n)
    ctx.state_upload(st.extended_flat(D))
    fermion.evaluate_amplitude(ctx, st, cfgs[:n])
    t0 = time.time()
    for _ in range(2):
        a = fermion.evaluate_amplitude(ctx, st, cfgs[:n])
    res["amplitudes_per_s_" + name] = 2 * n / (time.time() - t0)
    res["zero_flags_" + name] = int(np.count_nonzero(ctx.walker_flags()))
    del ctx
n = min(nw, 1024)
hostapi.fermion_energy(st, cfgs[:n], chi, 1.0, 1.0, capi.F32)
t0 = time.time(); hostapi.fermion_energy(st, cfgs[:n], chi, 1.0, 1.0, capi.F32)
res["energy_samples_per_s_f32"] = n / (time.time() - t0)
print(json.dumps(res))
