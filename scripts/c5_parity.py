"""C5 amplitude parity: device f32 / f64 against the float64 C restatement on the decorated network; max / median / weighted."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from peps_amd import capi, fermion, synthetic
from oracle import cbmps
L, D, chi, n = 8, 6, 24, 256
st = fermion.random_even_state(L, L, D, seed=11)
flat = st.extended_flat(D)
phys = synthetic.make_configs(L, n, "heisenberg", seed0=7)
ext = st.ext_config(phys, fermion.ROW)
ref, _, _ = cbmps.amplitudes_multiprocess(flat, ext, chi, 64)
res = {}
for name, dt in (("f32", capi.F32), ("f64", capi.F64)):
    ctx = capi.Context(L, L, D, 4 * st.d, chi, dtype=dt, max_walkers=n)
    ctx.state_upload(flat); ctx.set_configs(ext)
    a = ctx.evaluate_amplitude()
    rel = np.abs(a / ref - 1)
    res[name] = {"max": float(rel.max()), "median": float(np.median(rel)), "rms_weighted": float(np.sqrt(np.sum((a - ref) ** 2) / np.sum(ref ** 2))),
                 "argmax_amp_over_median": float(abs(ref[np.argmax(rel)]) / np.median(np.abs(ref)))}
    ctx.close()
print(json.dumps(res))
