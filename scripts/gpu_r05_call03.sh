#!/bin/bash
# round 5, call 3: the float64-accumulating wave-per-tile body (kernel tests, real-state tests), ADVICE fixes (walker finaliser, non-finite
# Gram input), the error budget with the backward pair in f64 (n = 256), and what it costs on the real_rank leg
cd /root/repo
mkdir -p gpurun_out/r05
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_walker.py tests/test_gpu_realrank.py -m gpu -q -x --tb=short -k "wave_per_tile or chained or non_finite or walker or real_rank or tgemm" > gpurun_out/r05/call03_tests.log 2>&1
echo "tests rc=$?"; tail -15 gpurun_out/r05/call03_tests.log
ONLY="f32;f32 backward pair in f32 (round 4);f32 Y on the wave-per-tile f64 body;f32 acc64 M;f32 acc64 all contractions"
timeout 1500 python scripts/error_budget.py --walkers 256 --state real --oracle 32 --only "$ONLY" > gpurun_out/r05/budget3_c4_real.json 2> gpurun_out/r05/budget3_c4_real.err
grep "^f32" gpurun_out/r05/budget3_c4_real.err | cut -c1-260
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05/budget3_c4_real.json"))
print("f64 vs oracle", d["runs"]["f64"])
PY
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-/root/repo}
VAR=PEPSGPU_TT_ACC64 VALS="0 1" NW=4096 bash scripts/ab_real.sh
VAR=PEPSGPU_Y_ACC64 VALS="1 2" NW=4096 bash scripts/ab_real.sh
