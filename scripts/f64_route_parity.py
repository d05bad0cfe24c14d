"""f64 device mode against oracle/cbmps.c on the tiled real state at C4, with the dense f64 truncation route and with the general
kernels (PEPSGPU_NO_F64_DENSE_ROUTE=1), same configurations.   usage: python scripts/f64_route_parity.py [n] [seed0]"""
import json, os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from peps_amd import capi, hostapi, synthetic
from oracle import cbmps
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
L, D, chi = 12, 8, 32
flat = synthetic.tile_flat_state(hostapi.load_sitps(os.path.join(ROOT, "tests/golden/ref_fixtures", synthetic.REAL_FIXTURE), 8), L)
ctx = capi.Context(L, L, D, 2, 4 * D, dtype=capi.F64, max_walkers=1)
ctx.state_upload(flat); ctx.set_configs(synthetic.checkerboard(L)[None])
flat = flat * abs(float(ctx.evaluate_amplitude()[0])) ** (-1.0 / (L * L)); ctx.close()
cfgs = synthetic.make_configs_near_neel(L, n, seed0=seed0)
ref, _, _ = cbmps.amplitudes_multiprocess(flat, cfgs, chi, min(n, 16))
WORKER = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
from peps_amd import capi
d = np.load(sys.argv[2])
ctx = capi.Context(12, 12, 8, 2, 32, dtype=capi.F64, max_walkers=len(d["cfgs"]))
ctx.state_upload(d["flat"]); ctx.set_configs(d["cfgs"])
np.save(sys.argv[3], ctx.evaluate_amplitude())
"""
out = {"n": n, "seed0": seed0}
with tempfile.TemporaryDirectory() as td:
    np.savez(os.path.join(td, "job.npz"), flat=flat, cfgs=cfgs)
    for name, env in (("route", {}), ("general", {"PEPSGPU_NO_F64_DENSE_ROUTE": "1"})):
        res = os.path.join(td, name + ".npy")
        r = subprocess.run([sys.executable, "-c", WORKER, ROOT, os.path.join(td, "job.npz"), res], env=dict(os.environ, **env), capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-1500:]
        rel = np.abs(np.load(res) / ref - 1)
        out[name] = {"max": float(rel.max()), "median": float(np.median(rel)), "argmax": int(np.argmax(rel))}
print(json.dumps(out))
