#!/bin/bash
# round 5 call 45: per-launch durations of the left-looking and the register-resident Cholesky on the real leg (kernel trace; every step under a timeout)
mkdir -p gpurun_out/r05
export TMPDIR=/tmp
for v in 0 1; do
  export PEPSGPU_CHOL_RESIDENT=$v
  rm -rf /tmp/crp$v
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/crp$v -o x -- python3 bench.py --state real --walkers 8192 --steps 1 --warmup 1 --no-route-check --no-energy-check --no-sweeps --no-latency --no-other-modes --cpu-seconds 2 > /tmp/crp$v.log 2>&1
  f=$(find /tmp/crp$v -name "*kernel_trace.csv" | head -1)
  if [ -n "$f" ]; then timeout 120 python3 scripts/trace_summary.py "$f" | grep -E "total ms|chol_" | head -8 > gpurun_out/r05/call45_resident$v.txt; cat gpurun_out/r05/call45_resident$v.txt | cut -c1-220; else echo "no trace for $v"; tail -5 /tmp/crp$v.log; fi
done
