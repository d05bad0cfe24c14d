#!/bin/bash
# round 4, call 23: Jacobi rows as register pairs + folded DPP adds (all Jacobi kernels), the triangular J1-J2 model: tests, the three legs
cd /root/repo; mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_host.py tests/test_gpu_measure.py -x -q -m gpu -k "triangle or j1j2" > gpurun_out/r04/t23a.log 2>&1
echo "pytest rc $?" >> gpurun_out/r04/t23a.log
grep -E "passed|failed|rc |Error|assert" gpurun_out/r04/t23a.log | tail -6
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullrank.py tests/test_gpu_realrank.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r04/t23.log 2>&1
echo "pytest rc $?" >> gpurun_out/r04/t23.log
grep -E "passed|failed|rc |Error" gpurun_out/r04/t23.log | tail -4
timeout 900 python bench.py --steps 5 --warmup 2 --no-other-modes --no-sweeps --no-cpu-baseline --no-energy-check > gpurun_out/r04/bench23.json 2> gpurun_out/r04/bench23.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04/bench23.json').read().strip().splitlines()[-1])
print("value", d["value"], "parity", d.get("parity_on_sample", {}).get("max_rel_err_amplitude"), {k: round(v,1) for k,v in d.get("kernel_ms",{}).items()})
for leg in ("full_rank","real_rank"):
    x=d.get(leg,{})
    print(leg, x.get("value"), x.get("parity_on_sample", {}).get("max_rel_err_amplitude"), {k: round(v,1) for k,v in x.get("kernel_ms",{}).items()})
PY
