#!/bin/bash
# round 4, call 21: kernel trace of the Monte-Carlo sweeps on the real state (2048 walkers)
cd /root/repo; O=gpurun_out/r04prof; mkdir -p $O
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o kt_sweep_real -- python3 scripts/sweep_trace.py 2048 real > $O/kt_sweep_real.log 2>&1
python3 scripts/trace_summary.py $O/kt_sweep_real_kernel_trace.csv > $O/r04_kernel_trace_by_grid_sweep_c4_f32_real_nw2048.txt
rm -f $O/kt_sweep_real_kernel_trace.csv
head -16 $O/r04_kernel_trace_by_grid_sweep_c4_f32_real_nw2048.txt | cut -c1-170
