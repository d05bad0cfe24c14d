#!/bin/bash
# round 5, call 16: dense f64 route entered from chi + 4 kept directions (the guard prices what the factors dropped): rates, parity
cd /root/repo
mkdir -p gpurun_out/r05
timeout 900 python scripts/f64_real_probe.py f64 2048 real 2>&1 | tail -1
timeout 600 python scripts/f64_real_probe.py f64 4096 c5 2>&1 | tail -1
PEPSGPU_DEBUG_SWEEPS=1 PEPSGPU_DEBUG_VERBOSE=1 timeout 600 python scripts/f64_real_probe.py f64 512 c5 2> gpurun_out/r05/f64_route_diag4.err | tail -1
grep "f64 dense route" gpurun_out/r05/f64_route_diag4.err | tail -8 | cut -c1-330
timeout 900 python scripts/error_budget.py --walkers 64 --state real --oracle 32 --only "f32" > gpurun_out/r05/budget16_c4_real.json 2> gpurun_out/r05/budget16_c4_real.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05/budget16_c4_real.json"))
print("f64 route vs oracle (n = 32):", d["runs"]["f64"])
PY
timeout 2400 python -m pytest tests/test_gpu_realrank.py tests/test_gpu_parity.py tests/test_gpu_fermion.py tests/test_gpu_configs.py tests/test_gpu_host.py -m gpu -q -x --tb=short 2>&1 | tail -4
