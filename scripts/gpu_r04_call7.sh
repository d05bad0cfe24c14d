#!/bin/bash
# round 4, call 7: skinny f64 GEMM + chained BTen step under the tests; sweep probe; the three legs
mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_parity.py tests/test_gpu_host.py tests/test_gpu_walker.py tests/test_gpu_measure.py tests/test_gpu_realrank.py tests/test_gpu_kernels.py tests/test_gpu_sr.py tests/test_gpu_configs.py -x -q -m gpu > gpurun_out/r04/t7.log 2>&1
echo "pytest rc $?" >> gpurun_out/r04/t7.log
grep -E "passed|failed|rc |Error" gpurun_out/r04/t7.log | tail -5
timeout 1500 python scripts/sweep_probe.py --walkers 8192 --state synthetic --paths device > gpurun_out/r04/sweep_probe3_synth.jsonl 2> gpurun_out/r04/sweep_probe3_synth.err
cat gpurun_out/r04/sweep_probe3_synth.jsonl
PEPSGPU_NO_BTEN_CHAIN=1 timeout 1500 python scripts/sweep_probe.py --walkers 8192 --state synthetic --paths device > gpurun_out/r04/sweep_probe3_synth_nochain.jsonl 2>/dev/null
cat gpurun_out/r04/sweep_probe3_synth_nochain.jsonl
timeout 1500 python scripts/sweep_probe.py --walkers 2048 --state real --sweeps 2 --paths device > gpurun_out/r04/sweep_probe3_real.jsonl 2> gpurun_out/r04/sweep_probe3_real.err
cat gpurun_out/r04/sweep_probe3_real.jsonl
timeout 900 python bench.py --steps 5 --warmup 2 --no-other-modes --no-sweeps > gpurun_out/r04/bench7.json 2> gpurun_out/r04/bench7.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04/bench7.json').read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "parity", d.get("parity_on_sample", {}).get("max_rel_err_amplitude"))
for leg in ("full_rank","real_rank"):
    x=d.get(leg,{})
    print(leg, x.get("value"), x.get("parity_on_sample", {}).get("max_rel_err_amplitude"), x.get("f64_mode"), x.get("kernel_ms"))
PY
