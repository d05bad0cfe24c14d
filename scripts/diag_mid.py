"""A/B of the mid route: amplitudes and BMPS bonds of a small full-rank case with PEPSGPU_NO_MIDROUTE on / off (child processes)."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, json
sys.path.insert(0, %r)
import numpy as np
from peps_amd import capi, synthetic
L, D, chi = 6, 8, 32
sitps = synthetic.make_sitps(L, D, noise=1.0)
flat = synthetic.sitps_to_flat(sitps, D)
cfgs = synthetic.make_configs(L, 4, "heisenberg", seed0=101)
ctx = capi.Context(L, L, D, 2, chi, dtype=capi.F32, max_walkers=4)
ctx.state_upload(flat); ctx.set_configs(cfgs)
ctx.grow_bmps_step(capi.DOWN)
ctx.grow_bmps_step(capi.DOWN)
out = {}
for idx in range(L):
    t, ls = ctx.get_bmps_tensor(capi.DOWN, 2, idx)
    out["t%%d" %% idx] = [list(t.shape), float(np.abs(t).sum()), ls.tolist()]
ctx.set_configs(cfgs)
out["amp"] = ctx.evaluate_amplitude().tolist()
print("RESULT" + json.dumps(out))
''' % ROOT
res = {}
for mode in ("mid", "nomid"):
    env = dict(os.environ)
    if mode == "nomid":
        env["PEPSGPU_NO_MIDROUTE"] = "1"
    env.update({k: v for k, v in [a.split("=") for a in sys.argv[1:]]})
    r = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, env=env)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
    if not line:
        print(mode, "FAILED", r.stdout[-2000:], r.stderr[-3000:]); continue
    res[mode] = json.loads(line[0][6:])
    print(mode, "amp", res[mode]["amp"])
    for k in sorted(res[mode]):
        if k != "amp": print("   ", k, res[mode][k])
