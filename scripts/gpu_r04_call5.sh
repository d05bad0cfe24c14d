#!/bin/bash
# round 4, call 5: new host-layer tests, sweep probe (device slice path vs per-bond hook path, by batch size), sweep kernel trace,
# headline / full / real legs with the persistent carry hint
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
python -m pytest tests/test_gpu_host.py "tests/test_gpu_fermion.py::test_c5_spinless_tV_8x8_d6_chi24" -x -q -m gpu -s > gpurun_out/r04/t5.log 2>&1
echo "pytest rc $?" >> gpurun_out/r04/t5.log
grep -E "passed|failed|rc |Error" gpurun_out/r04/t5.log | tail -5
timeout 1500 python scripts/sweep_probe.py --walkers 8192,24576 --state synthetic > gpurun_out/r04/sweep_probe_synth.jsonl 2> gpurun_out/r04/sweep_probe_synth.err
cat gpurun_out/r04/sweep_probe_synth.jsonl
timeout 1500 python scripts/sweep_probe.py --walkers 2048 --state real --sweeps 2 > gpurun_out/r04/sweep_probe_real.jsonl 2> gpurun_out/r04/sweep_probe_real.err
cat gpurun_out/r04/sweep_probe_real.jsonl
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04/prof_sweep -o sw -- python3 scripts/sweep_trace.py 8192 > gpurun_out/r04/prof_sweep.log 2>&1
python scripts/trace_summary.py $(find gpurun_out/r04/prof_sweep -name "*kernel_trace.csv" | head -1) > gpurun_out/r04/sweep_trace_summary.txt 2>&1
head -40 gpurun_out/r04/sweep_trace_summary.txt
find gpurun_out/r04/prof_sweep -name "*kernel_trace.csv" -delete
timeout 900 python bench.py --steps 5 --warmup 2 --no-other-modes --no-sweeps > gpurun_out/r04/bench5.json 2> gpurun_out/r04/bench5.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04/bench5.json').read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "parity", d.get("parity_on_sample"))
for leg in ("full_rank","real_rank"):
    x=d.get(leg,{})
    print(leg, x.get("value"), x.get("parity_on_sample"), x.get("f64_mode"), x.get("kernel_ms"))
print("kernel_ms", d.get("kernel_ms"))
print("roofline", {k:v for k,v in d["roofline"].items() if k not in ("note","counted","largest_priced_kernel")})
PY
