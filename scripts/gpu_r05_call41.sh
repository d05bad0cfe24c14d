#!/bin/bash
# round 5 call 41: register-resident Cholesky (chol_resident_kernel, PEPSGPU_CHOL_RESIDENT=1): kernel tests, then A/B on the real leg
mkdir -p gpurun_out/r05
PEPSGPU_CHOL_RESIDENT=1 timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "cholesky" 2>&1 | tail -15
NW=8192 VAR=PEPSGPU_CHOL_RESIDENT VALS="- 1" timeout 900 bash scripts/ab_real.sh 2>&1 | tail -4
