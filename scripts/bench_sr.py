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
out = {"samples": nw, "holes_all_sites_s": t_holes, "matvec_host_vectors_ms": dt * 1e3}
# device-resident CG: fixed number of iterations (tolerance 0), time per iteration
g = S * v
ctx.sr_cg_solve(g, None, 1e-3, 2, 0.0, 0.0, 0, 0.5)
t0 = time.time()
x, res, it, why = ctx.sr_cg_solve(g, None, 1e-3, 20, 0.0, 0.0, 0, 0.5)
dt_cg = (time.time() - t0) / max(it, 1)
out.update({"cg_device_ms_per_iteration": dt_cg * 1e3, "cg_iterations": it, "sample_sweep_GBps_in_cg": bytes_ / dt_cg / 1e9})
# MinSR: raw Gram of the resident samples (one GEMM, K = 2 * 144 * 4096), eigensolve on the host, back-substitution
t0 = time.time(); gm = ctx.sr_gram(); t_gram = time.time() - t0
flops = 2.0 * nw * nw / 2 * (L * L * 2 * D ** 4)
e = np.random.default_rng(1).standard_normal(nw)
t0 = time.time(); d, nrm = sr.minsr_direction(sr.DeviceSampleBatch(ctx), e, float(e.mean())); t_minsr = time.time() - t0
out.update({"minsr_gram_s": t_gram, "minsr_gram_TFLOPs": flops / t_gram / 1e12, "minsr_direction_total_s": t_minsr})
print(json.dumps(out))
