cd $GRAFT_REPO_ROOT
for st in "20 5" "3 1"; do
  set -- $st
  python bench.py --steps $1 --warmup $2 --no-full-rank --no-real-rank --no-other-modes --no-energy-check --no-latency --no-route-check --cpu-seconds 4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); v=d['vmc']; print('steps $1', round(d['value']), round(v['mc_sweeps_per_s']), round(v['vmc_samples_per_s']), {k: round(x,2) for k,x in v['call_seconds'].items()})"
done
