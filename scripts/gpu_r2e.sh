cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2e
export TMPDIR=/tmp
run() { tag=$1; shift
  timeout 600 python3 bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-route-check --no-full-rank > gpurun_out/r2e/bench_$tag.json 2> gpurun_out/r2e/bench_$tag.err
  python3 -c "
import json; d=json.load(open('gpurun_out/r2e/bench_$tag.json')); print('$tag', round(d['value'],1), round(d['ms_per_step'],1), d['kernel_ms'])"
}
run n1.0 --noise 1.0 --walkers 2048
PEPSGPU_NO_GRAMDIRECT=1 run n1.0_nogramdirect --noise 1.0 --walkers 2048
PEPSGPU_NO_MIDROUTE=1 run n1.0_nomid --noise 1.0 --walkers 2048
run n0.3 --noise 0.3 --walkers 2048
PEPSGPU_NO_MIDROUTE=1 run n0.3_nomid --noise 0.3 --walkers 2048
run head
PEPSGPU_NO_MIDROUTE=1 PEPSGPU_NO_GRAMDIRECT=1 run head_old
