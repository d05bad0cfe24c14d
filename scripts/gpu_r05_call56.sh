#!/bin/bash
# round 5, call 56: per-launch duration of the final form of the resident Cholesky on the real leg (kernel trace, under a timeout)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
export TMPDIR=/tmp PEPSGPU_CHOL_RESIDENT=1
rm -rf /tmp/crp1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/crp1 -o x -- python3 bench.py --state real --walkers 8192 --steps 1 --warmup 1 --no-route-check --no-energy-check --no-sweeps --no-latency --no-other-modes --cpu-seconds 2 > /tmp/crp1.log 2>&1
f=$(find /tmp/crp1 -name "*kernel_trace.csv" | head -1)
if [ -n "$f" ]; then timeout 120 python3 scripts/trace_summary.py "$f" | grep -E "total ms|chol_" | head -6 > gpurun_out/r05/call56_resident1.txt; cut -c1-200 gpurun_out/r05/call56_resident1.txt; else echo "no trace"; tail -3 /tmp/crp1.log; fi
