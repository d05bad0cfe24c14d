#!/bin/bash
# round 4, the driver's bench line on the last tree (after call 24) + smoke + the kernel trace of the headline leg
cd /root/repo; mkdir -p gpurun_out/r04 gpurun_out/r04prof
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04/smoke.log 2>&1; tail -1 gpurun_out/r04/smoke.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r04/bench_driver.json 2> gpurun_out/r04/bench_driver.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04/bench_driver.json').read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "sweeps", d.get("mc_sweeps_per_s"), "vmc", d.get("vmc_samples_per_s"), "n1", d.get("n1_ms"))
print("parity", d.get("parity_on_sample", {}).get("max_rel_err_amplitude"), "energy", d.get("energy_parity", {}).get("max_rel_err_energy"))
for leg in ("full_rank","real_rank"):
    x=d.get(leg,{})
    print(leg, x.get("value"), x.get("parity_on_sample", {}).get("max_rel_err_amplitude"), (x.get("vmc") or {}).get("mc_sweeps_per_s"), (x.get("vmc") or {}).get("vmc_samples_per_s"))
print("roofline", {k:v for k,v in d["roofline"].items() if k in ("bound","kernel","frac","achieved","traffic","avg_launch_us")})
print("other", {k: v.get("amp_per_s") for k, v in d.get("other_modes", {}).items()})
PY
O=gpurun_out/r04prof; export TMPDIR=/tmp
COMMON="--steps 1 --warmup 1 --no-cpu-baseline --no-route-check --no-full-rank --no-real-rank --no-sweeps --no-latency --no-energy-check --no-other-modes"
tag=c4_f32_noise0.1_nw49152
python3 bench.py $COMMON > $O/r04_bench_profiled_config_$tag.json 2> $O/bench_$tag.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o kt_$tag -- python3 bench.py $COMMON > $O/kt_$tag.log 2>&1
python3 scripts/trace_summary.py $O/kt_${tag}_kernel_trace.csv > $O/r04_kernel_trace_by_grid_$tag.txt
cp $O/kt_${tag}_kernel_stats.csv $O/r04_kernel_stats_$tag.csv
rm -f $O/kt_${tag}_kernel_trace.csv
head -6 $O/r04_kernel_trace_by_grid_$tag.txt | cut -c1-160
find $O -name "*.csv" -size +3M -delete
