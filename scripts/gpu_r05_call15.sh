#!/bin/bash
# round 5, call 15: dense f64 route with its stragglers on the side stream: rate, parity
cd /root/repo
mkdir -p gpurun_out/r05
for nw in 1024 2048; do timeout 900 python scripts/f64_real_probe.py f64 $nw real 2>&1 | tail -1; done
PEPSGPU_NO_F64_ROUTE_SIDE=1 timeout 900 python scripts/f64_real_probe.py f64 2048 real 2>&1 | tail -1
timeout 600 python scripts/f64_real_probe.py f64 4096 c5 2>&1 | tail -1
timeout 900 python scripts/error_budget.py --walkers 64 --state real --oracle 32 --only "f32" > gpurun_out/r05/budget15_c4_real.json 2> gpurun_out/r05/budget15_c4_real.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05/budget15_c4_real.json"))
print("f64 route vs oracle (n = 32):", d["runs"]["f64"])
PY
timeout 1500 python -m pytest tests/test_gpu_realrank.py tests/test_gpu_parity.py -m gpu -q -x --tb=short 2>&1 | tail -3
