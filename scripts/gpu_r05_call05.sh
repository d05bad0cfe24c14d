#!/bin/bash
# round 5, call 5: Y on the wave-per-tile f64 body (prefer_tiled fixed); how the f64 / complex modes scale with the batch on the dense
# real state, and where the f64 time goes (kernel trace)
cd /root/repo
mkdir -p gpurun_out/r05
ONLY="f32;f32 Y on the wave-per-tile f64 body"
timeout 900 python scripts/error_budget.py --walkers 256 --state real --only "$ONLY" > gpurun_out/r05/budget5_c4_real.json 2> gpurun_out/r05/budget5_c4_real.err
grep "^f32" gpurun_out/r05/budget5_c4_real.err | cut -c1-260
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-/root/repo}
VAR=PEPSGPU_Y_ACC64 VALS="1 2" NW=4096 bash scripts/ab_real.sh
for nw in 256 1024 2048; do timeout 900 python scripts/f64_real_probe.py f64 $nw real 2>&1 | tail -1; done
timeout 600 python scripts/f64_real_probe.py c128 256 real 2>&1 | tail -1
timeout 600 python scripts/f64_real_probe.py c128 1024 real 2>&1 | tail -1
timeout 600 python scripts/f64_real_probe.py f64 1024 c5 2>&1 | tail -1
timeout 600 python scripts/f64_real_probe.py f64 4096 c5 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_f64 -o f64real -- python3 /root/repo/scripts/f64_real_probe.py f64 512 real > /root/repo/gpurun_out/r05/prof_f64.log 2>&1
f=$(find /tmp/prof_f64 -name "*kernel_stats.csv" | head -1)
cp "$f" /root/repo/gpurun_out/r05/f64real_nw512_kernel_stats.csv 2>/dev/null
head -25 "$f" | cut -c1-200
