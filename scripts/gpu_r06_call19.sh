#!/bin/bash
# round 6, call 19: device Suwa-Todo unit cases, identical-chain tests again, chi = 40 pivot cap (64-row kernel) smoke via the 8x8 case at chi = 40
cd /root/repo; mkdir -p gpurun_out/r06
timeout 1800 python -m pytest tests/test_gpu_host.py -m gpu -q -x --tb=short -k "suwa_todo or slice_sweep" 2>&1 | grep -vE "^RCCL|^HIP|^ROCm|^Hostname|^Librccl" | tail -5
python - <<'PY'
import os, sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from peps_amd import capi, synthetic
import test_gpu_realrank as t
L, D, chi = 8, 8, 40
flat = t._state(L, D)
cfgs = synthetic.make_configs_near_neel(L, 4, seed0=211)
ref_a, _ = t._oracle(flat, cfgs, chi)
for dt, tol in ((capi.F32, 1e-5), (capi.F64, 1e-8)):
    ctx = capi.Context(L, L, D, 2, chi, dtype=dt, max_walkers=len(cfgs))
    ctx.state_upload(flat); ctx.set_configs(cfgs)
    a = ctx.evaluate_amplitude()
    print("8x8 D=8 chi=40 dtype", dt, "max rel err vs oracle %.2e" % np.max(np.abs(a / ref_a - 1)), "flags", int(np.sum(ctx.walker_flags() != 0)))
    ctx.close()
PY
