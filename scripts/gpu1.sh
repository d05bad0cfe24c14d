set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -q 2>&1 | tail -15 > gpurun_out/kernels.log; tail -15 gpurun_out/kernels.log
timeout 1500 python -m pytest tests/test_gpu_parity.py -q 2>&1 | tail -40 > gpurun_out/parity.log; tail -40 gpurun_out/parity.log
PEPSGPU_DEBUG_SWEEPS=1 timeout 1200 python scripts/diag2.py > gpurun_out/diag2.log 2>&1; tail -40 gpurun_out/diag2.log
