#!/bin/bash
# round 4, call 20: the whole GPU suite on the final tree; chol_blocked_kernel at two blocks per CU / two k-steps in flight (real leg)
cd /root/repo; mkdir -p gpurun_out/r04
python -m pytest tests -m gpu -x -q > gpurun_out/r04/full_suite_final.log 2>&1
echo "pytest rc $?" >> gpurun_out/r04/full_suite_final.log
tail -3 gpurun_out/r04/full_suite_final.log
COMMON="--steps 2 --warmup 1 --no-cpu-baseline --no-route-check --no-full-rank --no-real-rank --no-sweeps --no-latency --no-energy-check --no-other-modes --state real --walkers 8192"
for v in "default:" "minb2:PEPSGPU_CHB_MINB=2" "pf2:PEPSGPU_CHB_PF=2"; do
  name=${v%%:*}; E=${v#*:}
  env $E python3 bench.py $COMMON > gpurun_out/r04/bench20_$name.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open('gpurun_out/r04/bench20_$name.json').read().strip().splitlines()[-1])
print("$name value", d["value"], {k: round(v,1) for k,v in d["kernel_ms"].items() if k in ("cholesky","trunc_gram")})
PY
done
