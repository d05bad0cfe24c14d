#!/bin/bash
# LDS / register Jacobi for every row length class up to 256: kernel tests, rates, f64 / fermion / complex / measure tests
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short 2>&1 | tail -2
for a in "f64 4096 c5" "f64 2048 real" "f64 2048 noise0.1" "c128 512 real"; do python3 scripts/f64_real_probe.py $a 2>&1 | grep "^{" | tail -1 | cut -c1-330; done
timeout 2000 python -m pytest tests/test_gpu_fermion.py tests/test_gpu_measure.py tests/test_gpu_realrank.py -m gpu -q --tb=short 2>&1 | grep -E "FAILED|passed|failed" | tail -3
