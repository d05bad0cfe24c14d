#!/bin/bash
# round 4, call 17: kernel trace of the real_rank leg with the current tree
cd /root/repo; O=gpurun_out/r04; mkdir -p $O
export TMPDIR=/tmp
COMMON="--steps 1 --warmup 1 --no-cpu-baseline --no-route-check --no-full-rank --no-real-rank --no-sweeps --no-latency --no-energy-check --no-other-modes"
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o kt17 -- python3 bench.py $COMMON --state real --walkers 8192 > $O/kt17.log 2>&1
python3 scripts/trace_summary.py $O/kt17_kernel_trace.csv > $O/kernel_trace_by_grid_real17.txt
rm -f $O/kt17_kernel_trace.csv
head -45 $O/kernel_trace_by_grid_real17.txt | cut -c1-200
