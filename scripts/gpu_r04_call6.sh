#!/bin/bash
# round 4, call 6: trace dot kernel under the parity tests, sweep probe again
mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_parity.py tests/test_gpu_host.py tests/test_gpu_walker.py tests/test_gpu_complex.py tests/test_gpu_measure.py -x -q -m gpu > gpurun_out/r04/t6.log 2>&1
echo "pytest rc $?" >> gpurun_out/r04/t6.log
grep -E "passed|failed|rc |Error" gpurun_out/r04/t6.log | tail -5
timeout 1500 python scripts/sweep_probe.py --walkers 8192,24576 --state synthetic --paths device > gpurun_out/r04/sweep_probe2_synth.jsonl 2> gpurun_out/r04/sweep_probe2_synth.err
cat gpurun_out/r04/sweep_probe2_synth.jsonl
timeout 1500 python scripts/sweep_probe.py --walkers 2048 --state real --sweeps 2 --paths device > gpurun_out/r04/sweep_probe2_real.jsonl 2> gpurun_out/r04/sweep_probe2_real.err
cat gpurun_out/r04/sweep_probe2_real.jsonl
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04/prof_sweep2 -o sw -- python3 scripts/sweep_trace.py 8192 > gpurun_out/r04/prof_sweep2.log 2>&1
python scripts/trace_summary.py $(find gpurun_out/r04/prof_sweep2 -name "*kernel_trace.csv" | head -1) > gpurun_out/r04/sweep_trace_summary2.txt 2>&1
head -14 gpurun_out/r04/sweep_trace_summary2.txt
find gpurun_out/r04/prof_sweep2 -name "*kernel_trace.csv" -delete
