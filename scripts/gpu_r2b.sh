cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2b
export TMPDIR=/tmp
timeout 1500 python3 -m pytest tests/test_gpu_comm.py tests/test_gpu_fullrank.py -x -q -m gpu > gpurun_out/r2b/pytest_new.log 2>&1; echo "pytest new rc=$?"; tail -15 gpurun_out/r2b/pytest_new.log
timeout 900 python3 bench.py > gpurun_out/r2b/bench_default.json 2> gpurun_out/r2b/bench_default.err; echo "bench rc=$?"; tail -c 3000 gpurun_out/r2b/bench_default.json; tail -5 gpurun_out/r2b/bench_default.err
