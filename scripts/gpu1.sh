cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x 2>&1 | tail -3
timeout 1500 python -m pytest tests/test_gpu_parity.py -q 2>&1 | tail -3
timeout 900 python bench.py --walkers 512 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_reg.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['kernel_ms'], d['roofline'])"
