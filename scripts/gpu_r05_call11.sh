#!/bin/bash
# round 5, call 11: dense f64 route with the guard and the 3/4-rank subspace at the edge sites; dynamic-LDS general Jacobi; rates + parity
cd /root/repo
mkdir -p gpurun_out/r05
for nw in 512 2048; do timeout 900 python scripts/f64_real_probe.py f64 $nw real 2>&1 | tail -1; done
timeout 600 python scripts/f64_real_probe.py f64 4096 c5 2>&1 | tail -1
PEPSGPU_NO_JACOBI_DYN_LDS=1 timeout 600 python scripts/f64_real_probe.py f64 4096 c5 2>&1 | tail -1
timeout 900 python scripts/error_budget.py --walkers 64 --state real --oracle 32 --only "f32" > gpurun_out/r05/budget11_c4_real.json 2> gpurun_out/r05/budget11_c4_real.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05/budget11_c4_real.json"))
print("f64 route vs oracle (n = 32):", d["runs"]["f64"])
PY
timeout 2400 python -m pytest tests/test_gpu_realrank.py tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_fermion.py tests/test_gpu_kernels.py -m gpu -q -x --tb=short > gpurun_out/r05/call11_tests.log 2>&1
echo "tests rc=$?"; tail -6 gpurun_out/r05/call11_tests.log
