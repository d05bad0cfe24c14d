#!/bin/bash
# round 6, call 03: kernel trace of the real leg with the pivoted first compression (cap 56), by grid; kernel test again
cd /root/repo; mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --tb=short -k "pivoted_cholesky" 2>&1 | tail -5
cd /tmp && export TMPDIR=/tmp
export PEPSGPU_PIVOT_CHOL=${CAP:-56}
rm -rf /tmp/prof_real; mkdir -p /tmp/prof_real
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_real -o r1 -- python3 /root/repo/bench.py --state real --walkers 8192 --steps 2 --warmup 1 --no-cpu-baseline --no-route-check --no-energy-check --no-sweeps --no-latency --no-other-modes > /tmp/prof_real/bench.log 2>&1
t=$(find /tmp/prof_real -name "r1_kernel_trace.csv" | head -1)
python3 /root/repo/scripts/trace_summary.py "$t" > /root/repo/gpurun_out/r06/trace_real_pivot${PEPSGPU_PIVOT_CHOL}.txt
head -40 /root/repo/gpurun_out/r06/trace_real_pivot${PEPSGPU_PIVOT_CHOL}.txt | cut -c1-200
