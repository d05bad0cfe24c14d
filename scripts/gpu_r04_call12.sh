#!/bin/bash
# round 4, call 12: identity exchanges skipped in the slice sweep / energy slice; kernel tests; sweep probes
cd /root/repo
mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_host.py tests/test_gpu_measure.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r04/t12.log 2>&1
echo "pytest rc $?" >> gpurun_out/r04/t12.log
grep -E "passed|failed|rc |Error" gpurun_out/r04/t12.log | tail -5
timeout 900 python scripts/sweep_probe.py --walkers 8192 --state synthetic --paths device > gpurun_out/r04/sweep_probe12_synth.jsonl 2> gpurun_out/r04/sweep_probe12_synth.err
cut -c1-420 gpurun_out/r04/sweep_probe12_synth.jsonl
timeout 1500 python scripts/sweep_probe.py --walkers 2048 --state real --sweeps 2 --paths device > gpurun_out/r04/sweep_probe12_real.jsonl 2> gpurun_out/r04/sweep_probe12_real.err
cut -c1-420 gpurun_out/r04/sweep_probe12_real.jsonl
(cd scripts && ./build/gram_i8_bench 2048 1536 5) > gpurun_out/r04/gram_i8_microbench.jsonl 2>&1
