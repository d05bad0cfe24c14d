cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2c
export TMPDIR=/tmp
timeout 1200 python3 -m pytest tests/test_gpu_fullrank.py tests/test_gpu_kernels.py -x -q -m gpu > gpurun_out/r2c/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/r2c/pytest.log
for mode in mid nomid; do
  if [ $mode = nomid ]; then export PEPSGPU_NO_MIDROUTE=1; else unset PEPSGPU_NO_MIDROUTE; fi
  timeout 600 python3 bench.py --noise 1.0 --walkers 2048 --steps 2 --warmup 1 --no-cpu-baseline --no-route-check > gpurun_out/r2c/bench_n1.0_$mode.json 2> gpurun_out/r2c/bench_n1.0_$mode.err; echo "bench $mode rc=$?"
  python3 -c "
import json; d=json.load(open('gpurun_out/r2c/bench_n1.0_$mode.json')); print('$mode', d['value'], d['ms_per_step'], d['kernel_ms'])"
done
unset PEPSGPU_NO_MIDROUTE
timeout 600 python3 bench.py --noise 0.3 --walkers 2048 --steps 2 --warmup 1 --no-cpu-baseline --no-route-check > gpurun_out/r2c/bench_n0.3.json 2>/dev/null
python3 -c "
import json; d=json.load(open('gpurun_out/r2c/bench_n0.3.json')); print('n0.3', d['value'], d['ms_per_step'], d['kernel_ms'])"
timeout 600 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-route-check --no-full-rank > gpurun_out/r2c/bench_head.json 2>/dev/null
python3 -c "
import json; d=json.load(open('gpurun_out/r2c/bench_head.json')); print('head', d['value'], d['ms_per_step'], d['kernel_ms'])"
