cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2k
export TMPDIR=/tmp
run() { tag=$1; shift
  timeout 600 python3 bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-route-check --no-full-rank > gpurun_out/r2k/bench_$tag.json 2> gpurun_out/r2k/bench_$tag.err
  python3 -c "
import json; d=json.load(open('gpurun_out/r2k/bench_$tag.json')); print('$tag', round(d['value'],1), round(d['ms_per_step'],1), d['kernel_ms'])"
}
run head
run n1.0_4096 --noise 1.0 --walkers 4096
PEPSGPU_NO_COLGRAM=1 run n1.0_4096_nocolgram --noise 1.0 --walkers 4096
