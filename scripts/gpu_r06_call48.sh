#!/bin/bash
# gram_cols_f64_kernel with a straight-line k loop: kernel tests, rates, f64-route tests
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short 2>&1 | tail -2
for a in "f64 2048 real" "f64 4096 c5" "f32 4096 c5" "f32 8192 noise1" "f32 8192 real"; do python3 scripts/f64_real_probe.py $a 2>&1 | grep "^{" | tail -1 | cut -c1-330; done
timeout 3000 python -m pytest tests/test_gpu_realrank.py tests/test_gpu_fermion.py tests/test_gpu_fullrank.py -m gpu -q --tb=short 2>&1 | grep -E "FAILED|passed|failed" | tail -4
