#!/bin/bash
# round 6, call 06: kernel tests of the touched kernels, batch-size probe of the real leg, kernel trace by grid of the new default
cd /root/repo; mkdir -p gpurun_out/r06
export GRAFT_REPO_ROOT=/root/repo
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --tb=short -k "chained_contraction or mgemm_dense or pivoted or rows_qr" 2>&1 | tail -5
for nw in 12288 16384; do VAR=PEPSGPU_TRI VALS="1" NW=$nw bash scripts/ab_real.sh; done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_real; mkdir -p /tmp/prof_real
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_real -o r1 -- python3 /root/repo/bench.py --state real --walkers 8192 --steps 2 --warmup 1 --no-cpu-baseline --no-route-check --no-energy-check --no-sweeps --no-latency --no-other-modes > /tmp/prof_real/bench.log 2>&1
t=$(find /tmp/prof_real -name "r1_kernel_trace.csv" | head -1)
python3 /root/repo/scripts/trace_summary.py "$t" > /root/repo/gpurun_out/r06/trace_real_call06.txt
head -36 /root/repo/gpurun_out/r06/trace_real_call06.txt | cut -c1-200
