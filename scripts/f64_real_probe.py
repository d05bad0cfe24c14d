"""Amplitudes/s of one dtype on the tiled real state at C4 (or the C5 fermionic state) at a given batch size: the probe behind the
f64-mode figures of DESIGN 6.   usage: python scripts/f64_real_probe.py [f64|f32|c128] [walkers] [real|c5]"""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from peps_amd import capi, hostapi, synthetic
dt = {"f64": capi.F64, "f32": capi.F32, "c128": capi.C128}[sys.argv[1] if len(sys.argv) > 1 else "f64"]
nw = int(sys.argv[2]) if len(sys.argv) > 2 else 512
what = sys.argv[3] if len(sys.argv) > 3 else "real"
if what == "real":
    L, D, chi, dphys = 12, 8, 32, 2
    flat = synthetic.tile_flat_state(hostapi.load_sitps(os.path.join(ROOT, "tests/golden/ref_fixtures", synthetic.REAL_FIXTURE), 8), L)
    ctx = capi.Context(L, L, D, 2, chi, dtype=capi.F32, max_walkers=1)
    ctx.state_upload(flat); ctx.set_configs(synthetic.checkerboard(L)[None])
    flat = flat * abs(float(ctx.evaluate_amplitude()[0])) ** (-1.0 / (L * L)); ctx.close()
    if dt == capi.C128:
        flat = flat * np.exp(2j * np.pi * np.random.default_rng(5).uniform(size=flat.shape))
    batches = [synthetic.make_configs_near_neel(L, nw, seed0=307 + 1000 * k) for k in range(2)]
elif what.startswith("noise"):      # the synthetic state of bench.py: noise0.1 = headline, noise1 = full rank
    L, D, chi, dphys = 12, 8, 32, 2
    sitps = synthetic.make_sitps(L, D, noise=float(what[5:]))
    ctx = capi.Context(L, L, D, 2, chi, dtype=capi.F32, max_walkers=1)
    ctx.state_upload(synthetic.sitps_to_flat(sitps, D, np.float64)); ctx.set_configs(synthetic.checkerboard(L)[None])
    flat = synthetic.sitps_to_flat(synthetic.rescale_sitps(sitps, float(ctx.evaluate_amplitude()[0])), D, np.float64); ctx.close()
    batches = [synthetic.make_configs(L, nw, "heisenberg", seed0=7 + 1000 * k) for k in range(2)]
else:
    from peps_amd import fermion
    L, D, chi = 8, 6, 24
    st = fermion.random_even_state(L, L, D, seed=11)
    flat, dphys = st.extended_flat(D), fermion.NVAR * st.d
    batches = [st.ext_config(synthetic.make_configs(L, nw, "heisenberg", seed0=80000 + 7 * k), fermion.ROW) for k in range(2)]
ctx = capi.Context(L, L, D, dphys, chi, dtype=dt, max_walkers=nw)
ctx.state_upload(flat)
ctx.set_configs(batches[0][:min(nw, 64)]); ctx.evaluate_amplitude(); ctx.sync()       # hints, allocations
ctx.set_configs(batches[0]); ctx.evaluate_amplitude(); ctx.sync()
ctx.profile_enable(1)
t0 = time.perf_counter()
ctx.set_configs(batches[1]); a = ctx.evaluate_amplitude(); ctx.sync()
dtm = time.perf_counter() - t0
prof = ctx.profile_read()
print(json.dumps({"dtype": sys.argv[1] if len(sys.argv) > 1 else "f64", "what": what, "walkers": nw, "amp_per_s": nw / dtm, "seconds": dtm,
                  "flags": int(np.sum(ctx.walker_flags() != 0)),
                  "kernel_ms": {k: round(v["ms"], 1) for k, v in prof.items() if v.get("launches")} if isinstance(prof, dict) else None}))
