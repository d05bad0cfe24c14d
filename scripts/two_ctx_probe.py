"""Does the GPU run two half-batches faster than one batch?  Two contexts (two HIP streams), each driven by its own host thread
(ctypes releases the GIL inside the C ABI), against one context with all walkers.  argv: state (real | synthetic | full), walkers."""
import sys, os, time, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from peps_amd import capi, hostapi, synthetic
state = sys.argv[1] if len(sys.argv) > 1 else "real"
nw = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
nctx = int(sys.argv[3]) if len(sys.argv) > 3 else 2
L, D, chi, _ = synthetic.CONFIGS["C4"]
if state == "real":
    from conftest import FIXTURES
    flat = synthetic.tile_flat_state(hostapi.load_sitps(os.path.join(FIXTURES, synthetic.REAL_FIXTURE), 8), L) * 0.5
    cfgs = synthetic.make_configs_near_neel(L, nw, seed0=307)
else:
    flat = synthetic.sitps_to_flat(synthetic.make_sitps(L, D, noise=1.0 if state == "full" else 0.1), D)
    cfgs = synthetic.make_configs(L, nw, "heisenberg")
import json
def run(parts, reps=3):
    ctxs = []
    for p in parts:
        c = capi.Context(L, L, D, 2, chi, dtype=capi.F32, max_walkers=len(p))
        c.state_upload(flat)
        ctxs.append(c)
    outs = [None] * len(parts)
    def work(i):
        for _ in range(reps):
            ctxs[i].set_configs(parts[i])
            outs[i] = ctxs[i].evaluate_amplitude()
    for i in range(len(parts)):       # warm-up
        ctxs[i].set_configs(parts[i]); ctxs[i].evaluate_amplitude()
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(i,)) for i in range(len(parts))]
    for t in th: t.start()
    for t in th: t.join()
    dt = (time.perf_counter() - t0) / reps
    for c in ctxs: c.close()
    return dt, np.concatenate(outs)
t1, a1 = run([cfgs])
t2, a2 = run(np.array_split(cfgs, nctx))
print(json.dumps({"state": state, "walkers": nw, "one_context_amp_per_s": nw / t1, "contexts": nctx, "split_amp_per_s": nw / t2,
                  "ratio": t1 / t2, "max_rel_diff": float(np.max(np.abs(a2 / a1 - 1)))}))
