#!/bin/bash
# round 4, call 15: 12-wave integer Gram in the library: real-rank / full-rank / kernel tests, the three legs; row form with 12 waves in the microbenchmark
cd /root/repo
mkdir -p gpurun_out/r04
(cd scripts && timeout 300 ./build/gram_i8_bench 2048 1536 5) 2>&1 | grep -E "ROWS|rows_f64" | cut -c1-200
python -m pytest tests/test_gpu_realrank.py tests/test_gpu_fullrank.py tests/test_gpu_kernels.py -x -q -m gpu > gpurun_out/r04/t15.log 2>&1
echo "pytest rc $?" >> gpurun_out/r04/t15.log
grep -E "passed|failed|rc |Error|C4 real" gpurun_out/r04/t15.log | tail -8
timeout 900 python bench.py --steps 5 --warmup 2 --no-other-modes --no-sweeps > gpurun_out/r04/bench15.json 2> gpurun_out/r04/bench15.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04/bench15.json').read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "parity", d.get("parity_on_sample", {}).get("max_rel_err_amplitude"))
for leg in ("full_rank","real_rank"):
    x=d.get(leg,{})
    print(leg, x.get("value"), x.get("parity_on_sample", {}).get("max_rel_err_amplitude"), x.get("kernel_ms"))
    print("   mfma", x.get("mfma", {}).get("categories", {}).get("gram_f64"))
PY
