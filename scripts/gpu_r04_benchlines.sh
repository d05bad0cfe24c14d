#!/bin/bash
# the two bench lines of the final tree: the driver's command and the defaults
cd /root/repo; mkdir -p gpurun_out/r04
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04/bench_driver.json 2> gpurun_out/r04/bench_driver.err
python bench.py > gpurun_out/r04/bench_default.json 2> gpurun_out/r04/bench_default.err
python - <<'PY'
import json
for f in ("driver", "default"):
    d=json.loads(open('gpurun_out/r04/bench_%s.json' % f).read().strip().splitlines()[-1])
    print(f, "value", d["value"], "sweeps", d.get("mc_sweeps_per_s"), "vmc", d.get("vmc_samples_per_s"), "full", d["full_rank"]["value"], "real", d["real_rank"]["value"],
          d["real_rank"].get("vmc", {}).get("mc_sweeps_per_s"), d["real_rank"].get("vmc", {}).get("vmc_samples_per_s"), "parity", d["parity_on_sample"]["max_rel_err_amplitude"],
          d["real_rank"]["parity_on_sample"]["max_rel_err_amplitude"], "traffic", d["roofline"].get("traffic"))
PY
