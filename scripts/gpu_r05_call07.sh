#!/bin/bash
# round 5, call 7: the generic Jacobi with four pairs per wave at a time (f64 bulk): kernel tests, f64 parity, f64 rates on the dense real state
cd /root/repo
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --tb=short -k "jacobi" > gpurun_out/r05/call07_tests.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r05/call07_tests.log
timeout 1500 python -m pytest tests/test_gpu_realrank.py tests/test_gpu_parity.py -m gpu -q -x --tb=short > gpurun_out/r05/call07_tests2.log 2>&1
echo "tests2 rc=$?"; tail -3 gpurun_out/r05/call07_tests2.log
for nw in 512 2048; do timeout 900 python scripts/f64_real_probe.py f64 $nw real 2>&1 | tail -1; done
timeout 600 python scripts/f64_real_probe.py f64 4096 c5 2>&1 | tail -1
