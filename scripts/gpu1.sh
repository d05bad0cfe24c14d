cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_gpu_host.py -q -x 2>&1 | tail -30
