#!/bin/bash
# round 5 call 43: resident Cholesky with look-ahead (every step under its own timeout)
mkdir -p gpurun_out/r05
PEPSGPU_CHOL_RESIDENT=1 timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "cholesky" 2>&1 | tail -5
NW=8192 VAR=PEPSGPU_CHOL_RESIDENT VALS="- 1" timeout 400 bash scripts/ab_real.sh 2>&1 | tail -4
