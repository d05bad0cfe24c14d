#!/bin/bash
# round 4, call 18: chol_blocked_kernel with 3 / 5 / 8 k-steps of the update in flight (real leg, 8192 walkers)
cd /root/repo; mkdir -p gpurun_out/r04
COMMON="--steps 2 --warmup 1 --no-cpu-baseline --no-route-check --no-full-rank --no-real-rank --no-sweeps --no-latency --no-energy-check --no-other-modes --state real --walkers 8192"
for pf in 3 5 8; do
  PEPSGPU_CHB_PF=$pf python3 bench.py $COMMON > gpurun_out/r04/bench18_pf$pf.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open('gpurun_out/r04/bench18_pf$pf.json').read().strip().splitlines()[-1])
print("pf $pf value", d["value"], {k: round(v,1) for k,v in d["kernel_ms"].items() if k in ("cholesky","trunc_gram","gram_f64")})
PY
done
