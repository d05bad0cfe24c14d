cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2i
export TMPDIR=/tmp
python3 scripts/diag_mid.py 2>&1 | grep amp
run() { tag=$1; shift
  timeout 600 python3 bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-route-check --no-full-rank > gpurun_out/r2i/bench_$tag.json 2> gpurun_out/r2i/bench_$tag.err
  python3 -c "
import json; d=json.load(open('gpurun_out/r2i/bench_$tag.json')); print('$tag', round(d['value'],1), round(d['ms_per_step'],1), d['kernel_ms']); print(json.dumps({k:round(v['tflops'],1) for k,v in d['mfma']['categories'].items()}))"
}
run n1.0_4096 --noise 1.0 --walkers 4096
PEPSGPU_NO_FUSED_MIDGRAM=1 run n1.0_4096_nofusedmid --noise 1.0 --walkers 4096
run n0.3 --noise 0.3 --walkers 4096
timeout 1500 python3 -m pytest tests/test_gpu_fullrank.py -x -q -m gpu > gpurun_out/r2i/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r2i/pytest.log
