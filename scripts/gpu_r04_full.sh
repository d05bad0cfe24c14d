#!/bin/bash
# the driver's sequence: the whole GPU suite, smoke, the default bench line
cd /root/repo
mkdir -p gpurun_out/r04
python -m pytest tests -m gpu -x -q > gpurun_out/r04/full_suite.log 2>&1
echo "pytest rc $?" >> gpurun_out/r04/full_suite.log
tail -4 gpurun_out/r04/full_suite.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04/smoke.log 2>&1; tail -1 gpurun_out/r04/smoke.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r04/bench_driver.json 2> gpurun_out/r04/bench_driver.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04/bench_driver.json').read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "sweeps", d.get("mc_sweeps_per_s"), "vmc", d.get("vmc_samples_per_s"), "n1", d.get("n1_ms"))
print("parity", d.get("parity_on_sample")); print("energy", {k: d.get("energy_parity", {}).get(k) for k in ("max_rel_err_energy", "n")})
for leg in ("full_rank","real_rank"):
    x=d.get(leg,{})
    print(leg, x.get("value"), x.get("parity_on_sample"), x.get("f64_mode"), x.get("vmc", {}).get("mc_sweeps_per_s") if isinstance(x.get("vmc"), dict) else x.get("vmc"), x.get("energy_parity", {}).get("max_rel_err_energy"))
    r=x.get("roofline",{}); print("   roofline", r.get("bound"), r.get("kernel"), r.get("frac"), r.get("share_of_kernel_time"))
print("roofline", {k:v for k,v in d["roofline"].items() if k in ("bound","kernel","frac","achieved","traffic","avg_launch_us","frac_priced_with")})
print("other", d.get("other_modes"))
PY
python bench.py > gpurun_out/r04/bench_default.json 2> gpurun_out/r04/bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04/bench_default.json').read().strip().splitlines()[-1])
print("default: value", d["value"], "sweeps", d.get("mc_sweeps_per_s"), "vmc", d.get("vmc_samples_per_s"), "full", d["full_rank"]["value"], "real", d["real_rank"]["value"], d["real_rank"].get("vmc", {}).get("mc_sweeps_per_s"))
print("roofline", {k: d["roofline"].get(k) for k in ("bound", "kernel", "frac", "traffic", "frac_priced_with")})
print("real roofline", {k: d["real_rank"]["roofline"].get(k) for k in ("bound", "kernel", "share_of_kernel_time")}, {k: d["real_rank"]["roofline"].get("largest_priced_kernel", {}).get(k) for k in ("kernel", "frac", "traffic")})
PY
