#!/bin/bash
# LDS / register Jacobi for rows of 129-192 and 193-256 elements: jacobi kernel tests, rates, fermion tests
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -k "jacobi" 2>&1 | tail -1
for a in "f64 4096 c5" "f64 2048 real" "f64 2048 noise0.1" "f64 4096 c5" "f64 2048 real"; do python3 scripts/f64_real_probe.py $a 2>&1 | grep "^{" | tail -1 | cut -c1-330; done
timeout 1200 python -m pytest tests/test_gpu_fermion.py -m gpu -q --tb=short 2>&1 | grep -E "FAILED|passed|failed" | tail -2
