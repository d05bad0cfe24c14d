#!/bin/bash
# round 6, call 05: triangular carry (chain kernel + mgemm_dense skip the zero blocks of the Cholesky factor): kernel tests, real-leg A/B, parity
cd /root/repo; mkdir -p gpurun_out/r06
export GRAFT_REPO_ROOT=/root/repo
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --tb=short -k "chained_contraction or mgemm_dense" 2>&1 | tail -15
VAR=PEPSGPU_TRI VALS="0 1" NW=8192 bash scripts/ab_real.sh
timeout 1500 python -m pytest tests/test_gpu_realrank.py -m gpu -q -x --tb=short -s -k "c4_amplitudes_vs_oracle or c4_batch or amplitude_and_energy" 2>&1 | grep -E "max|passed|failed|Error|error" | tail -15
