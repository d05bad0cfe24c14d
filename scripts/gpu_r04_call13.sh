#!/bin/bash
# round 4, call 13: kernel / host / measure / parity tests; sweep probe at 16384 walkers
cd /root/repo
mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_host.py tests/test_gpu_measure.py tests/test_gpu_parity.py tests/test_gpu_walker.py -x -q -m gpu > gpurun_out/r04/t13.log 2>&1
echo "pytest rc $?" >> gpurun_out/r04/t13.log
grep -E "passed|failed|rc |Error" gpurun_out/r04/t13.log | tail -5
timeout 900 python scripts/sweep_probe.py --walkers 16384 --state synthetic --paths device > gpurun_out/r04/sweep_probe13_synth16k.jsonl 2> gpurun_out/r04/sweep_probe13_synth16k.err
cut -c1-420 gpurun_out/r04/sweep_probe13_synth16k.jsonl; tail -3 gpurun_out/r04/sweep_probe13_synth16k.err
(cd scripts && ./build/gram_i8_bench 2048 1536 5) > gpurun_out/r04/gram_i8_microbench.jsonl 2>&1
