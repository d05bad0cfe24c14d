#!/bin/bash
# round 6, call 04: rows_qr (Cholesky-QR of the projected rows instead of polish Jacobi + select + Newton-Schulz): kernel tests, real-leg A/B, parity
cd /root/repo; mkdir -p gpurun_out/r06
export GRAFT_REPO_ROOT=/root/repo
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --tb=short -k "pivoted_cholesky or rows_qr" 2>&1 | tail -15
export PEPSGPU_PIVOT_CHOL=56
VAR=PEPSGPU_ROWS_QR VALS="0 1" NW=8192 bash scripts/ab_real.sh
unset PEPSGPU_PIVOT_CHOL
VAR=PEPSGPU_PIVOT_CHOL VALS="64 56" NW=8192 bash scripts/ab_real.sh
export PEPSGPU_PIVOT_CHOL=56
timeout 1500 python -m pytest tests/test_gpu_realrank.py -m gpu -q -x --tb=short -s -k "c4_amplitudes_vs_oracle or c4_batch or amplitude_and_energy" 2>&1 | grep -E "max|passed|failed|Error|error" | tail -15
