cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
for nw in 64 256 512; do
  timeout 900 python bench.py --walkers $nw --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_nw$nw.json
done
timeout 900 python bench.py --walkers 256 --steps 1 --warmup 1 --dtype f64 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_f64.json
