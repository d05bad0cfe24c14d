#!/bin/bash
# round 4, call 24: gram_chol_wave_kernel with the conversion itself as the opaque statement (no copies): kernel + parity tests, headline
cd /root/repo; mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r04/t24.log 2>&1
echo "pytest rc $?" >> gpurun_out/r04/t24.log
grep -E "passed|failed|rc |Error" gpurun_out/r04/t24.log | tail -4
timeout 300 python bench.py --steps 5 --warmup 2 --no-other-modes --no-sweeps --no-cpu-baseline --no-energy-check --no-real-rank --no-latency > gpurun_out/r04/bench24.json 2> gpurun_out/r04/bench24.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04/bench24.json').read().strip().splitlines()[-1])
print("value", d["value"], d.get("parity_on_sample"), {k: round(v,1) for k,v in d.get("kernel_ms",{}).items()})
x=d.get("full_rank",{})
print("full", x.get("value"), x.get("parity_on_sample"))
PY
