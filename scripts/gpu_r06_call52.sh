#!/bin/bash
# tile index split by the float reciprocal in the wave-per-tile bodies: kernel + parity tests, rates of the f32 legs
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -m gpu -q --tb=short 2>&1 | tail -2
python3 scripts/floor_probe.py run f64 2048 2>&1 | tail -1
python3 scripts/floor_probe.py run f32_fdiv 8192 2>&1 | tail -1
python3 scripts/floor_probe.py analyse | grep -E "fdiv"
for a in "f32 49152 noise0.1" "f32 49152 noise0.1" "f32 8192 noise1" "f32 8192 real"; do python3 scripts/f64_real_probe.py $a 2>&1 | grep "^{" | tail -1 | cut -c1-200; done
