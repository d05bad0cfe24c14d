"""Who are the walkers in the tail of the f32-vs-f64 amplitude error on the tiled real state at C4?  Prints, for the worst ones, the size of
their amplitude relative to the batch median and how many sites they differ from the Neel pattern at."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from peps_amd import capi, hostapi, synthetic
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
L, D, chi = 12, 8, 32
flat = synthetic.tile_flat_state(hostapi.load_sitps(os.path.join(ROOT, "tests/golden/ref_fixtures", synthetic.REAL_FIXTURE), 8), L)
ctx = capi.Context(L, L, D, 2, 4 * D, dtype=capi.F64, max_walkers=1)
ctx.state_upload(flat); ctx.set_configs(synthetic.checkerboard(L)[None])
flat = flat * abs(float(ctx.evaluate_amplitude()[0])) ** (-1.0 / (L * L)); ctx.close()
cfgs = synthetic.make_configs_near_neel(L, n, seed0=100000)
amp = {}
for name, dt in (("f32", capi.F32), ("f64", capi.F64)):
    c = capi.Context(L, L, D, 2, chi, dtype=dt, max_walkers=n)
    c.state_upload(flat); c.set_configs(cfgs); amp[name] = c.evaluate_amplitude(); c.close()
rel = np.abs(amp["f32"] / amp["f64"] - 1)
mag = np.abs(amp["f64"]); med = np.median(mag)
neel = synthetic.checkerboard(L)
diff = np.minimum(np.sum(cfgs != neel[None], axis=(1, 2)), np.sum(cfgs != (1 - neel)[None], axis=(1, 2)))
order = np.argsort(-rel)
out = {"n": n, "median_rel": float(np.median(rel)), "p99_rel": float(np.percentile(rel, 99)), "max_rel": float(rel.max()),
       "corr_log_rel_log_mag": float(np.corrcoef(np.log(rel + 1e-12), np.log(mag))[0, 1]),
       "worst": [{"rel": float(rel[i]), "abs_psi_over_median": float(mag[i] / med), "sites_off_neel": int(diff[i])} for i in order[:6]],
       "sites_off_neel_median": float(np.median(diff))}
print(json.dumps(out))
