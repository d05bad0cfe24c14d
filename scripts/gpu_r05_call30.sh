#!/bin/bash
# round 5, call 30: the driver's bench command on the final tree (f64 figure of the real leg at 2 048 walkers)
cd /root/repo
mkdir -p gpurun_out/r05
( time timeout 1500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05/bench_driver_run4.json 2> gpurun_out/r05/bench_driver_run4.err ) 2>&1 | tail -3
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05/bench_driver_run4.json"))
print("value", d["value"], "ms_per_step", d["ms_per_step"], "frac", d["roofline"]["frac"])
for leg in ("full_rank", "real_rank"):
    l = d.get(leg, {})
    print(leg, l.get("value"), l.get("parity_on_sample", {}).get("max_rel_err_amplitude"), "f64_mode", l.get("f64_mode"))
print("C5", json.dumps(d["other_modes"]["C5_spinless_tV_8x8_D6_chi24"])[:400])
PY
