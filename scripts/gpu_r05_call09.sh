#!/bin/bash
# round 5, call 9: the dense float64 truncation route (two-level preconditioning with oversampling + Rayleigh-Ritz Jacobi in LDS):
# parity of the f64 mode (real-state tests at 1e-8 / 1e-9 against the oracle, the f64 parity suite), then its rate
cd /root/repo
mkdir -p gpurun_out/r05
timeout 1500 python -m pytest tests/test_gpu_realrank.py -m gpu -q -x --tb=short > gpurun_out/r05/call09_tests.log 2>&1
echo "realrank rc=$?"; tail -6 gpurun_out/r05/call09_tests.log
timeout 900 python scripts/error_budget.py --walkers 64 --state real --oracle 32 --only "f32" > gpurun_out/r05/budget9_c4_real.json 2> gpurun_out/r05/budget9_c4_real.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05/budget9_c4_real.json"))
print("f64 route vs oracle (n = 32):", d["runs"]["f64"])
PY
for nw in 512 2048; do timeout 900 python scripts/f64_real_probe.py f64 $nw real 2>&1 | tail -1; done
PEPSGPU_NO_F64_DENSE_ROUTE=1 timeout 900 python scripts/f64_real_probe.py f64 512 real 2>&1 | tail -1
timeout 600 python scripts/f64_real_probe.py f64 4096 c5 2>&1 | tail -1
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_host.py -m gpu -q -x --tb=short > gpurun_out/r05/call09_tests2.log 2>&1
echo "parity rc=$?"; tail -4 gpurun_out/r05/call09_tests2.log
