#!/bin/bash
# round 5, call 18: dense f64 route with the second-chance factorisation of the walkers above 128 rows: rate, who is left, parity
cd /root/repo
mkdir -p gpurun_out/r05
timeout 900 python scripts/f64_real_probe.py f64 2048 real 2>&1 | tail -1
PEPSGPU_DEBUG_SWEEPS=1 PEPSGPU_DEBUG_VERBOSE=1 timeout 600 python scripts/f64_real_probe.py f64 1024 real 2> gpurun_out/r05/f64_route_diag5.err | tail -1
grep "f64 dense route" gpurun_out/r05/f64_route_diag5.err | tail -9 | cut -c1-330
timeout 900 python scripts/error_budget.py --walkers 64 --state real --oracle 32 --only "f32" > gpurun_out/r05/budget18_c4_real.json 2> gpurun_out/r05/budget18_c4_real.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05/budget18_c4_real.json"))
print("f64 route vs oracle (n = 32):", d["runs"]["f64"])
PY
timeout 1500 python -m pytest tests/test_gpu_realrank.py tests/test_gpu_kernels.py -m gpu -q -x --tb=short 2>&1 | tail -3
