#!/bin/bash
# LDS Jacobi: register rows by length class, two interleaved pairs per wave pass: kernel tests, f64 / C5 / complex tests and rates
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --tb=short -k "jacobi" 2>&1 | tail -2
for a in "f64 2048 real" "f64 4096 c5" "f64 2048 noise0.1"; do python3 scripts/f64_real_probe.py $a 2>&1 | grep "^{" | tail -1 | cut -c1-420; done
timeout 2000 python -m pytest tests/test_gpu_realrank.py tests/test_gpu_fermion.py tests/test_gpu_parity.py tests/test_gpu_measure.py -m gpu -q -x --tb=short 2>&1 | tail -3
