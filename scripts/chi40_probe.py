"""probe: 8x8 tiled real state at D = 8, chi = 40 (carry D chi = 320 > 256) against the oracle, f32 and f64"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from peps_amd import capi, synthetic
import test_gpu_realrank as t
L, D, chi = 8, 8, int(sys.argv[1]) if len(sys.argv) > 1 else 40
flat = t._state(L, D)
cfgs = synthetic.make_configs_near_neel(L, 4, seed0=211)
ref_a, _ = t._oracle(flat, cfgs, chi)
for dt in (capi.F32, capi.F64):
    ctx = capi.Context(L, L, D, 2, chi, dtype=dt, max_walkers=len(cfgs))
    ctx.state_upload(flat); ctx.set_configs(cfgs)
    a = ctx.evaluate_amplitude()
    print("8x8 D=8 chi=%d dtype %d max rel err vs oracle %.2e flags %d" % (chi, dt, np.max(np.abs(a / ref_a - 1)), int(np.sum(ctx.walker_flags() != 0))), a, ref_a)
    ctx.close()
