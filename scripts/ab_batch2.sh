cd $GRAFT_REPO_ROOT
for nw in 12288 16384; do
  python bench.py --noise 1.0 --walkers $nw --steps 2 --warmup 1 --no-route-check --no-energy-check --no-sweeps --no-latency --no-other-modes --no-full-rank --no-real-rank --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('full_rank state walkers $nw', round(d['value'],1), round(d['ms_per_step'],1))"
  rocm-smi --showmemuse 2>/dev/null | grep -i "GPU\[0\]" | head -2
done
