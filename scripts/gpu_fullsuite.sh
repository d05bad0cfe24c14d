cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/suite
export TMPDIR=/tmp
timeout 3000 python3 -m pytest tests -x -q -m gpu > gpurun_out/suite/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/suite/pytest_gpu.log
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
