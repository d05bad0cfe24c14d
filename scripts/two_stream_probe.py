"""Round 6 probe: amplitudes/s of ONE context against TWO (or more) contexts driven from host threads (each its own stream and
walker batch; ctypes releases the GIL during a call) -- do kernels of different phases of the absorption overlap across streams?
usage: python scripts/two_stream_probe.py [real|synthetic|full] [total walkers] [contexts ...]"""
import json, os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from peps_amd import capi, hostapi, synthetic
what = sys.argv[1] if len(sys.argv) > 1 else "real"
total = int(sys.argv[2]) if len(sys.argv) > 2 else 12288
splits = [int(x) for x in sys.argv[3:]] or [1, 2]
L, D, chi = 12, 8, 32
if what == "real":
    flat = synthetic.tile_flat_state(hostapi.load_sitps(os.path.join(ROOT, "tests/golden/ref_fixtures", synthetic.REAL_FIXTURE), 8), L)
    c = capi.Context(L, L, D, 2, chi, dtype=capi.F32, max_walkers=1)
    c.state_upload(flat); c.set_configs(synthetic.checkerboard(L)[None])
    flat = flat * abs(float(c.evaluate_amplitude()[0])) ** (-1.0 / (L * L)); c.close()
    gen = lambda n, seed: synthetic.make_configs_near_neel(L, n, seed0=seed)
else:
    flat = synthetic.sitps_to_flat(synthetic.make_sitps(L, D, noise=0.1 if what == "synthetic" else 1.0), D, np.float64)
    gen = lambda n, seed: synthetic.make_configs(L, n, "heisenberg", seed0=seed)
res = {}
for k in splits:
    n = total // k
    ctxs = [capi.Context(L, L, D, 2, chi, dtype=capi.F32, max_walkers=n) for _ in range(k)]
    batches = [[gen(n, 1000 * (3 * j + s) + 7) for s in range(3)] for j in range(k)]
    out = [None] * k
    def work(j, steps):
        c = ctxs[j]
        for s in steps:
            c.set_configs(batches[j][s]); out[j] = c.evaluate_amplitude()
        c.sync()
    for j in range(k):
        ctxs[j].state_upload(flat)
    ts = [threading.Thread(target=work, args=(j, [0])) for j in range(k)]
    [t.start() for t in ts]; [t.join() for t in ts]
    t0 = time.perf_counter()
    ts = [threading.Thread(target=work, args=(j, [1, 2])) for j in range(k)]
    [t.start() for t in ts]; [t.join() for t in ts]
    dt = time.perf_counter() - t0
    res["%d_contexts" % k] = {"walkers_each": n, "amp_per_s": 2 * n * k / dt, "seconds": dt, "flags": int(sum(np.sum(c.walker_flags() != 0) for c in ctxs))}
    for c in ctxs: c.close()
print(json.dumps({"state": what, "total": total, **res}))
