#!/bin/bash
# round 4, call 25: A/B of the rotation without the "no lane rotates" branch (JRX_BRANCHFREE): kernel + parity tests, the three legs
cd /root/repo; mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py tests/test_gpu_realrank.py -x -q -m gpu > gpurun_out/r04/t25.log 2>&1
echo "pytest rc $?" >> gpurun_out/r04/t25.log
grep -E "passed|failed|rc |Error" gpurun_out/r04/t25.log | tail -4
timeout 300 python bench.py --steps 5 --warmup 2 --no-other-modes --no-sweeps --no-cpu-baseline --no-energy-check --no-latency > gpurun_out/r04/bench25.json 2> gpurun_out/r04/bench25.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04/bench25.json').read().strip().splitlines()[-1])
print("value", d["value"], {k: round(v,1) for k,v in d.get("kernel_ms",{}).items() if "jacobi" in k})
for leg in ("full_rank","real_rank"):
    x=d.get(leg,{})
    print(leg, x.get("value"), (x.get("parity_on_sample") or {}).get("max_rel_err_amplitude"), {k: round(v,1) for k,v in x.get("kernel_ms",{}).items() if "jacobi" in k})
PY
