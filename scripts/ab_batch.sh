# batch-size scan of the two secondary legs: amplitudes/s against walkers per step
cd $GRAFT_REPO_ROOT
for nw in 4096 8192; do
  python bench.py --noise 1.0 --walkers $nw --steps 2 --warmup 1 --no-route-check --no-energy-check --no-sweeps --no-latency --no-other-modes --no-full-rank --no-real-rank --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('full_rank state walkers $nw', round(d['value'],1), round(d['ms_per_step'],1))"
done
for nw in 8192 12288; do
  python bench.py --state real --walkers $nw --steps 2 --warmup 1 --no-route-check --no-energy-check --no-sweeps --no-latency --no-other-modes --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('real state walkers $nw', round(d['value'],1), round(d['ms_per_step'],1))"
done
