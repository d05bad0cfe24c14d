"""EvaluateAmplitude throughput with the variational compression schemes (bmps_impl.h:864-1172) beside
SVD_COMPRESS, same synthetic workload as bench.py.  usage: bench_var.py [workload] [walkers] [noise] [iter_max]"""
import sys, json, time
sys.path.insert(0, '.')
import numpy as np
from peps_amd import capi, synthetic
wl = sys.argv[1] if len(sys.argv) > 1 else "C4"
nw = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
noise = float(sys.argv[3]) if len(sys.argv) > 3 else 0.1
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 3
L, D, chi, kind = synthetic.CONFIGS[wl]
sitps = synthetic.make_sitps(L, D, noise=noise)
cfg = synthetic.make_configs(L, nw, kind)
flat = synthetic.sitps_to_flat(sitps, D)
out = {"workload": wl, "walkers": nw, "noise": noise, "iter_max": iters, "convergence_tol": 1e-5}
ref = None
for name, scheme in (("svd", 0), ("variational_2site", 1), ("variational_1site", 2)):
    ctx = capi.Context(L, L, D, 2, chi, dtype=capi.F32, max_walkers=nw, scheme=scheme, convergence_tol=1e-5, iter_max=iters)
    ctx.state_upload(flat)
    ctx.set_configs(cfg); a = ctx.evaluate_amplitude()
    ctx.profile_enable(True); ctx.profile_read()
    ctx.sync(); t = time.time()
    ctx.set_configs(cfg); a = ctx.evaluate_amplitude()
    ctx.sync(); dt = time.time() - t
    prof = ctx.profile_read(); ctx.profile_enable(False)
    if ref is None:
        ref = a
    out[name] = {"amp_per_s": round(nw / dt, 1), "max_rel_dev_from_svd": float(np.max(np.abs(a / ref - 1))),
                 "zero_flags": int(np.sum(ctx.walker_flags() != 0)), "kernel_ms": {k: round(v["ms"], 1) for k, v in prof.items() if v["launches"]}, "contract_tflops": round(prof["contract"]["exec_flops"] / max(prof["contract"]["ms"], 1e-9) / 1e9, 1), "device_GB": round(ctx.stats().get("device_bytes", 0) / 1e9, 2)}
    ctx.close()
print(json.dumps(out))
