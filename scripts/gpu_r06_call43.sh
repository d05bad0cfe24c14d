#!/bin/bash
# hardware-estimate reciprocal / rsqrt (+ Newton) in the pivot steps of every factor kernel and in the complex / tiny-f64 Jacobi: full kernel tests,
# rates of every leg, then the full GPU suite
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --tb=short 2>&1 | tail -2
python3 scripts/floor_probe.py run f64 2048 2>&1 | tail -1
python3 scripts/floor_probe.py run f32_rsq 8192 2>&1 | tail -1
python3 scripts/floor_probe.py analyse | grep -E "rsq"
for a in "f32 49152 noise0.1" "f32 8192 noise1" "f32 4096 c5" "f64 2048 real" "c128 512 real" "f64 4096 c5"; do python3 scripts/f64_real_probe.py $a 2>&1 | grep "^{" | tail -1 | cut -c1-420; done
timeout 3000 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -6 | tee gpurun_out/r06/suite_call43.txt
