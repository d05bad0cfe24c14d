#!/bin/bash
# round 4, call 14: 12-wave form of the integer Gram (microbenchmark); sweep probe at 32768 walkers
cd /root/repo
mkdir -p gpurun_out/r04
(cd scripts && timeout 300 ./build/gram_i8_bench 2048 1536 5) 2>&1 | cut -c1-200
timeout 900 python scripts/sweep_probe.py --walkers 32768 --state synthetic --paths device > gpurun_out/r04/sweep_probe14_synth32k.jsonl 2> gpurun_out/r04/sweep_probe14_synth32k.err
cut -c1-420 gpurun_out/r04/sweep_probe14_synth32k.jsonl; tail -3 gpurun_out/r04/sweep_probe14_synth32k.err
