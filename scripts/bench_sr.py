"""S-matrix product with the O* samples resident in HBM at C4: time per product and effective bandwidth."""
import sys, time, json
sys.path.insert(0, '.')
import numpy as np
from peps_amd import capi, synthetic, sr
from peps_amd.capi import LEFT, RIGHT, UP, DOWN, HORIZONTAL
nw = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
L, D, chi, _ = synthetic.CONFIGS["C4"]
sitps = synthetic.make_sitps(L, D)
ctx = capi.Context(L, L, D, 2, chi, dtype=capi.F32, max_walkers=nw)
ctx.state_upload(synthetic.sitps_to_flat(sitps, D))
cfgs = synthetic.make_configs(L, nw, "heisenberg")
ctx.set_configs(cfgs); psi = ctx.evaluate_amplitude(); ctx.set_configs(cfgs)
t0 = time.time()
ctx.generate_bmps_approach(UP)
for row in range(L):
    ctx.init_bten(LEFT, row); ctx.grow_full_bten(RIGHT, row, 1, True)
    for col in range(L):
        ctx.punch_hole_store(row, col, HORIZONTAL)
        if col < L - 1: ctx.shift_bten_window(RIGHT)
    if row < L - 1: ctx.shift_bmps_window(DOWN)
t_holes = time.time() - t0
ctx.sr_begin(nw); ctx.sr_append(psi)
S = sr.DeviceSRSMatrix(ctx, 1e-3)
v = np.random.default_rng(0).standard_normal(S.mean.shape)
S * v
t0 = time.time()
for _ in range(5): S * v
dt = (time.time() - t0) / 5
bytes_ = 2.0 * nw * L * L * D ** 4 * 4
print(json.dumps({"samples": nw, "holes_all_sites_s": t_holes, "matvec_ms": dt * 1e3, "sample_sweep_GBps": bytes_ / dt / 1e9}))
