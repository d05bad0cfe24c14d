#!/bin/bash
# round 6, call 13: TRI as a template parameter of the chained kernel: kernel test, headline and real leg; C5 test with its measured errors
cd /root/repo; mkdir -p gpurun_out/r06
export GRAFT_REPO_ROOT=/root/repo
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fermion.py tests/test_gpu_realrank.py -m gpu -q -x --tb=short -s -k "chained_contraction or c5_spinless or round6_routes" 2>&1 | grep -E "C5|8x8|passed|failed|Error" | tail -12
python bench.py --steps 10 --warmup 3 --no-full-rank --no-real-rank --no-sweeps --no-latency --no-other-modes --no-route-check --no-energy-check --cpu-seconds 2 2>/dev/null > gpurun_out/r06/headline_tri_tmpl.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06/headline_tri_tmpl.json"))
print("headline", round(d["value"]), d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["frac"], {k: round(v) for k, v in d["kernel_ms"].items()})
PY
VAR=PEPSGPU_TRI VALS="1" NW=12288 bash scripts/ab_real.sh
