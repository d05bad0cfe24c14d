#!/bin/bash
# round 6, call 09: the driver's bench command on the current tree (new legs: baseline_configs; real_rank at 12288 walkers x 5 steps; 10 timed sweeps)
cd /root/repo; mkdir -p gpurun_out/r06
( time timeout 2400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06/bench_driver_run1.json 2> gpurun_out/r06/bench_driver_run1.err ) 2>&1 | tail -3
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06/bench_driver_run1.json"))
print("value", d["value"], "ms_per_step", d["ms_per_step"], "roofline", {k: d["roofline"].get(k) for k in ("bound", "achieved", "frac", "traffic")})
for leg in ("full_rank", "real_rank"):
    l = d.get(leg, {})
    print(leg, l.get("value"), l.get("walkers_per_gpu"), l.get("parity_on_sample"), "f64_mode", l.get("f64_mode"), "vmc", {k: l.get("vmc", {}).get(k) for k in ("mc_sweeps_per_s", "vmc_samples_per_s")} if isinstance(l.get("vmc"), dict) else None)
    print("   energy", l.get("energy_parity"))
print("baseline_configs", json.dumps(d.get("baseline_configs"))[:1500])
print("vmc", json.dumps(d.get("vmc"))[:700], "n1_ms", d.get("n1_ms"))
print("C5", json.dumps(d.get("other_modes", {}).get("C5_spinless_tV_8x8_D6_chi24"))[:600])
print("real complex", d.get("real_rank", {}).get("complex128"))
PY
tail -5 gpurun_out/r06/bench_driver_run1.err
