#!/bin/bash
# round 4, call 4: tightened parity tests (C2, C5, C4 real state vs oracle), slice sweep, precise-site policy on the three states,
# bf16 split compute-only pricing, the default bench line
mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_host.py tests/test_gpu_configs.py tests/test_gpu_fermion.py tests/test_gpu_realrank.py tests/test_gpu_fullrank.py -x -q -m gpu -s > gpurun_out/r04/t4.log 2>&1
echo "pytest rc $?" >> gpurun_out/r04/t4.log
grep -E "^C[245]|passed|failed|Error|error|rc " gpurun_out/r04/t4.log | tail -30
ONLY="f32;f32 round 3 (no ortho polish, Y f32 chain);f32 acc64 all contractions"
for st in real synthetic full; do
  timeout 600 python scripts/error_budget.py --walkers 64 --state $st --only "$ONLY" > gpurun_out/r04/budget4_c4_$st.json 2> gpurun_out/r04/budget4_c4_$st.err
  echo "== $st"; grep "^f" gpurun_out/r04/budget4_c4_$st.err
done
(cd scripts && ./build/bf16_split_bench 1024 10) > gpurun_out/r04/bf16_split2.jsonl 2> gpurun_out/r04/bf16_split2.err
cat gpurun_out/r04/bf16_split2.jsonl
timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/r04/bench4.json 2> gpurun_out/r04/bench4.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04/bench4.json').read().strip().splitlines()[-1])
def g(x,*k):
    for q in k:
        x = x.get(q) if isinstance(x, dict) else None
    return x
print("value", d["value"], "ms", d["ms_per_step"], "sweeps", d.get("mc_sweeps_per_s"), "vmc", d.get("vmc_samples_per_s"), "n1", d.get("n1_ms"))
print("parity", d.get("parity_on_sample"))
for leg in ("full_rank","real_rank"):
    print(leg, g(d,leg,"value"), g(d,leg,"parity_on_sample"), g(d,leg,"vmc"), g(d,leg,"kernel_ms"))
print("kernel_ms", d.get("kernel_ms"))
PY
