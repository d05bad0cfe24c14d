#!/bin/bash
# three Newton steps in the reciprocal / rsqrt helpers: kernel tests, rates, then the parts of the suite that run the factor kernels and the f64 / complex modes
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short 2>&1 | tail -3
for a in "f32 49152 noise0.1" "f32 8192 real" "f64 2048 real" "c128 512 real" "f64 4096 c5"; do python3 scripts/f64_real_probe.py $a 2>&1 | grep "^{" | tail -1 | cut -c1-300; done
timeout 3000 python -m pytest tests -m gpu -q --tb=short --deselect tests/test_gpu_kernels.py 2>&1 | grep -E "FAILED|passed|failed" | tail -6 | tee gpurun_out/r06/suite_call44.txt
