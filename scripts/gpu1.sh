cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for nz in 0.1 0.3 1.0; do
timeout 900 python bench.py --walkers 512 --steps 2 --warmup 1 --no-cpu-baseline --noise $nz 2>&1 | tail -1 | tee gpurun_out/bench_noise$nz.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['synthetic_noise'], d['value'], d['kernel_ms'], d['workload_rank'], d['roofline']['kernel'], d['roofline']['frac'])"
done
