#!/bin/bash
# round 6, call 02: pivoted first compression (chol_pivot.h): kernel test, real-leg A/B over the cap (0 = round-5 route), parity of the real leg
cd /root/repo; mkdir -p gpurun_out/r06
export GRAFT_REPO_ROOT=/root/repo
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --tb=short -k "pivoted_cholesky" 2>&1 | tail -15
VAR=PEPSGPU_PIVOT_CHOL VALS="0 64 56 48" NW=8192 bash scripts/ab_real.sh
timeout 1500 python -m pytest tests/test_gpu_realrank.py -m gpu -q -x --tb=short -s -k "c4_amplitudes_vs_oracle or c4_batch or amplitude_and_energy" 2>&1 | grep -E "max|passed|failed|Error|error" | tail -15
