#!/bin/bash
# round 4, call 16: the trace's half step reused as the next environment of accepted walkers: host / measure / walker tests, sweep probes
cd /root/repo
mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_host.py tests/test_gpu_measure.py tests/test_gpu_walker.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r04/t16.log 2>&1
echo "pytest rc $?" >> gpurun_out/r04/t16.log
grep -E "passed|failed|rc |Error" gpurun_out/r04/t16.log | tail -5
timeout 900 python scripts/sweep_probe.py --walkers 16384 --state synthetic --paths device > gpurun_out/r04/sweep_probe16_synth16k.jsonl 2> gpurun_out/r04/sweep_probe16_synth16k.err
cut -c1-420 gpurun_out/r04/sweep_probe16_synth16k.jsonl; tail -3 gpurun_out/r04/sweep_probe16_synth16k.err
PEPSGPU_NO_SWEEP_REUSE=1 timeout 900 python scripts/sweep_probe.py --walkers 16384 --state synthetic --paths device > gpurun_out/r04/sweep_probe16_synth16k_noreuse.jsonl 2>/dev/null
cut -c1-420 gpurun_out/r04/sweep_probe16_synth16k_noreuse.jsonl
timeout 1500 python scripts/sweep_probe.py --walkers 2048 --state real --sweeps 2 --paths device > gpurun_out/r04/sweep_probe16_real.jsonl 2> gpurun_out/r04/sweep_probe16_real.err
cut -c1-420 gpurun_out/r04/sweep_probe16_real.jsonl
