cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2d
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_gpu_kernels.py -q -m gpu -k "mid_route or streaming_gram" > gpurun_out/r2d/pytest_k.log 2>&1; echo "pytest rc=$?"; tail -30 gpurun_out/r2d/pytest_k.log
