#!/bin/bash
# round 6, call 18: headline sweeps/s with 3 and with 10 timed sweeps (r05 quoted 18.0 k on 3), plus the extended switch-coverage tests
cd /root/repo; mkdir -p gpurun_out/r06
for n in 3 10; do
python bench.py --steps 3 --warmup 1 --no-full-rank --no-real-rank --no-latency --no-other-modes --no-route-check --no-energy-check --cpu-seconds 2 --sweep-count $n 2>/dev/null > gpurun_out/r06/headline_sweeps_$n.json
python - <<PY
import json
d = json.load(open("gpurun_out/r06/headline_sweeps_$n.json"))
print("sweep-count $n:", round(d["vmc"]["mc_sweeps_per_s"]), round(d["vmc"]["vmc_samples_per_s"]), d["vmc"]["call_seconds"])
PY
done
timeout 2400 python -m pytest tests/test_gpu_realrank.py -m gpu -q -x --tb=short -s -k "round6_routes or dense_truncation_route" 2>&1 | grep -E "8x8|dense|passed|failed|Error|assert" | tail -20
