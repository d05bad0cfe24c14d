# A/B helper: one short real_rank run per value of an environment switch ("-" = unset).  usage: VAR=NAME VALS="- 1" bash scripts/ab_real.sh
cd $GRAFT_REPO_ROOT
for v in $VALS; do
  if [ "$v" = "-" ]; then unset $VAR; else export $VAR=$v; fi
  python bench.py --state real --walkers ${NW:-4096} --steps 2 --warmup 1 --no-route-check --no-energy-check --no-sweeps --no-latency --no-other-modes --cpu-seconds 4 2>/dev/null > gpurun_out/ab_tmp.json
  python - <<PY
import json
d = json.load(open("gpurun_out/ab_tmp.json"))
print("$VAR=$v", round(d["value"], 1), round(d["ms_per_step"], 1), {k: round(x / 2) for k, x in d["kernel_ms"].items()}, d.get("parity_on_sample", {}).get("max_rel_err_amplitude"))
PY
done
