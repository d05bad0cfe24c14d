#!/bin/bash
# round 5, call 27: the driver's commands on the final tree: smoke, bench (--gpus 1 --steps 20 --warmup 5), the whole -m gpu suite
cd /root/repo
mkdir -p gpurun_out/r05
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
( time timeout 1500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05/bench_driver_run3.json 2> gpurun_out/r05/bench_driver_run3.err ) 2>&1 | tail -3
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05/bench_driver_run3.json"))
print("value", d["value"], "ms_per_step", d["ms_per_step"], "roofline", {k: d["roofline"].get(k) for k in ("bound", "achieved", "frac", "traffic")})
for leg in ("full_rank", "real_rank"):
    l = d.get(leg, {})
    print(leg, l.get("value"), l.get("parity_on_sample"), "f64_mode", l.get("f64_mode"), "roofline.traffic", (l.get("roofline") or {}).get("traffic"), "vmc", l.get("vmc", {}).get("sweeps_per_s") if isinstance(l.get("vmc"), dict) else None)
print("parity", d.get("parity_on_sample"))
print("other_modes", json.dumps(d.get("other_modes"))[:1500])
print("vmc", json.dumps(d.get("vmc"))[:600], "n1_ms", d.get("n1_ms"))
print("real complex", d.get("real_rank", {}).get("complex128"))
PY
timeout 3000 python -m pytest tests -m gpu -q --tb=short > gpurun_out/r05/gpu_suite3.log 2>&1
echo "suite rc=$?"; tail -6 gpurun_out/r05/gpu_suite3.log
