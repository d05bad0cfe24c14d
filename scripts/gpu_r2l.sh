cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2l
export TMPDIR=/tmp
run() { tag=$1; shift
  timeout 600 python3 bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-route-check --no-full-rank > gpurun_out/r2l/bench_$tag.json 2> gpurun_out/r2l/bench_$tag.err
  python3 -c "
import json; d=json.load(open('gpurun_out/r2l/bench_$tag.json')); print('$tag', round(d['value'],1), round(d['ms_per_step'],1), d['kernel_ms']['cholesky'])"
}
run head
PEPSGPU_CG_PHASE=1 run head_phase1
PEPSGPU_CG_PHASE=2 run head_phase2
