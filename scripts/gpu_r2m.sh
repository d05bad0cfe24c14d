cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2m
export TMPDIR=/tmp
run() { tag=$1; shift
  timeout 600 python3 bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-route-check --no-full-rank > gpurun_out/r2m/bench_$tag.json 2> gpurun_out/r2m/bench_$tag.err
  python3 -c "
import json; d=json.load(open('gpurun_out/r2m/bench_$tag.json')); print('$tag', round(d['value'],1), round(d['ms_per_step'],1), d['kernel_ms'])"
}
run head
PEPSGPU_CG_PHASE=1 run head_phase1
PEPSGPU_NO_COLGRAM=1 run head_nocolgram
run n1.0_4096 --noise 1.0 --walkers 4096
PEPSGPU_NO_COLGRAM=1 run n1.0_4096_nocolgram --noise 1.0 --walkers 4096
timeout 1500 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullrank.py -q -m gpu > gpurun_out/r2m/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r2m/pytest.log | cut -c1-300
