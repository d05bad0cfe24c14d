#!/bin/bash
# round 5, call 4: the fused norm with the log-scale taken from the APPLIED f32 scale; Y on the wave-per-tile f64 body against the LDS-tiled one
cd /root/repo
mkdir -p gpurun_out/r05
timeout 1200 python -m pytest tests/test_gpu_walker.py tests/test_gpu_realrank.py -m gpu -q -x --tb=short > gpurun_out/r05/call04_tests.log 2>&1
echo "tests rc=$?"; tail -5 gpurun_out/r05/call04_tests.log
ONLY="f32;f32 Y on the wave-per-tile f64 body;f32 acc64 all contractions"
timeout 1500 python scripts/error_budget.py --walkers 256 --state real --only "$ONLY" > gpurun_out/r05/budget4_c4_real.json 2> gpurun_out/r05/budget4_c4_real.err
grep "^f32" gpurun_out/r05/budget4_c4_real.err | cut -c1-260
for st in synthetic full; do
  timeout 900 python scripts/error_budget.py --walkers 128 --state $st --only "f32;f32 no fused norm" > gpurun_out/r05/budget4_c4_$st.json 2> gpurun_out/r05/budget4_c4_$st.err
  echo "== $st"; grep "^f32" gpurun_out/r05/budget4_c4_$st.err | cut -c1-260
done
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-/root/repo}
VAR=PEPSGPU_Y_ACC64 VALS="1 2" NW=4096 bash scripts/ab_real.sh
