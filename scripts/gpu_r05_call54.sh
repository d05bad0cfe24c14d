#!/bin/bash
# round 5, call 54: final tree -- smoke, the whole -m gpu suite, then the driver's bench command (each step under a timeout)
cd /root/repo; mkdir -p gpurun_out/r05
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1500 python -m pytest tests -m gpu -q --tb=short > gpurun_out/r05/gpu_suite_final4.log 2>&1
echo "suite rc=$?"; tail -4 gpurun_out/r05/gpu_suite_final4.log
( time timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05/bench_driver_run7.json 2> gpurun_out/r05/bench_driver_run7.err ) 2>&1 | tail -3
timeout 60 python - <<'PY'
import json
d = json.load(open("gpurun_out/r05/bench_driver_run7.json"))
print("value", d["value"], "ms_per_step", d["ms_per_step"], "frac", d["roofline"]["frac"])
for leg in ("full_rank", "real_rank"):
    l = d.get(leg, {})
    print(leg, l.get("value"), l.get("parity_on_sample", {}).get("max_rel_err_amplitude"), "f64_mode", l.get("f64_mode"))
print("C5", json.dumps(d["other_modes"]["C5_spinless_tV_8x8_D6_chi24"])[:400])
PY
