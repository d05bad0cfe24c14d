#!/bin/bash
# tournament pairing without modulo; then the final profiles of the secondary legs
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -k "jacobi" 2>&1 | tail -2
for a in "f64 2048 real" "f64 4096 c5" "c128 512 real"; do python3 scripts/f64_real_probe.py $a 2>&1 | grep "^{" | tail -1 | cut -c1-330; done
timeout 1500 python -m pytest tests/test_gpu_complex.py tests/test_gpu_fermion.py -m gpu -q --tb=short 2>&1 | grep -E "FAILED|passed|failed" | tail -3
bash scripts/gpu_r06_call28.sh > gpurun_out/r06/call28_final.log 2>&1
