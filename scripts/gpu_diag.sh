cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export PEPSGPU_DEBUG_SWEEPS=1
timeout 1200 python scripts/diag2.py > gpurun_out/diag2.log 2>&1; tail -40 gpurun_out/diag2.log
