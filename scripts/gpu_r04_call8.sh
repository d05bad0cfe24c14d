#!/bin/bash
# round 4, call 8: the exact-integer Gram (gram_i8.h) under the real-rank / full-rank / fermion tests; the three legs
cd /root/repo
mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_realrank.py tests/test_gpu_fullrank.py tests/test_gpu_fermion.py tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -m gpu > gpurun_out/r04/t8.log 2>&1
echo "pytest rc $?" >> gpurun_out/r04/t8.log
grep -E "passed|failed|rc |Error|C4 real" gpurun_out/r04/t8.log | tail -8
timeout 900 python bench.py --steps 5 --warmup 2 --no-other-modes --no-sweeps > gpurun_out/r04/bench8.json 2> gpurun_out/r04/bench8.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04/bench8.json').read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "parity", d.get("parity_on_sample", {}).get("max_rel_err_amplitude"))
for leg in ("full_rank","real_rank"):
    x=d.get(leg,{})
    print(leg, x.get("value"), x.get("parity_on_sample", {}).get("max_rel_err_amplitude"), x.get("kernel_ms"))
    print("   roofline", x.get("roofline"))
PY
