#!/bin/bash
# round 5, call 6: stage 2 of the chained kernel with two J tiles per wave (headline A/B: off / five blocks per CU / four blocks), kernel + parity tests
cd /root/repo
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -m gpu -q -x --tb=short -k "chained or tgemm or amplitude or stack or k1" > gpurun_out/r05/call06_tests.log 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/r05/call06_tests.log
export GRAFT_REPO_ROOT=/root/repo
ARGS="--steps 3 --warmup 1 --no-real-rank --no-sweeps --no-other-modes --no-latency" VARIANTS="s2p0:PEPSGPU_CHAIN_S2PAIR=0 s2p1:PEPSGPU_CHAIN_S2PAIR=1 s2p2:PEPSGPU_CHAIN_S2PAIR=2" bash scripts/gpu_ab.sh
