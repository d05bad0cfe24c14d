cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2j
export TMPDIR=/tmp
timeout 1500 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullrank.py -q -m gpu > gpurun_out/r2j/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r2j/pytest.log | cut -c1-300
run() { tag=$1; shift
  timeout 600 python3 bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-route-check --no-full-rank > gpurun_out/r2j/bench_$tag.json 2> gpurun_out/r2j/bench_$tag.err
  python3 -c "
import json; d=json.load(open('gpurun_out/r2j/bench_$tag.json')); print('$tag', round(d['value'],1), round(d['ms_per_step'],1), d['kernel_ms'])"
}
run head
PEPSGPU_NO_COLGRAM=1 run head_nocolgram
run n1.0_4096 --noise 1.0 --walkers 4096
run n0.3 --noise 0.3 --walkers 4096
