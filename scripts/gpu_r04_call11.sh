#!/bin/bash
# round 4, call 11: kernel tests (integer Gram references), kernel trace of the sweeps with the current tree
cd /root/repo
mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu > gpurun_out/r04/t11.log 2>&1
echo "pytest rc $?" >> gpurun_out/r04/t11.log
grep -E "passed|failed|rc |Error" gpurun_out/r04/t11.log | tail -5
cd /tmp && export TMPDIR=/tmp && cd /root/repo
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04/prof_sweep3 -o sw -- python3 scripts/sweep_trace.py 8192 > gpurun_out/r04/prof_sweep3.log 2>&1
python scripts/trace_summary.py $(find gpurun_out/r04/prof_sweep3 -name "*kernel_trace.csv" | head -1) > gpurun_out/r04/sweep_trace_summary3.txt 2>&1
head -24 gpurun_out/r04/sweep_trace_summary3.txt | cut -c1-170
rm -rf gpurun_out/r04/prof_sweep3
