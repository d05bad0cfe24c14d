#!/bin/bash
# unconditional row loads in the Gram-free factor kernels; predicate-at-store in the LDS-tiled GEMMs: tests + rates of every leg
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -m gpu -q -x --tb=short 2>&1 | tail -3
python3 scripts/floor_probe.py run f64 2048 2>&1 | tail -1
python3 scripts/floor_probe.py run f32_uncond 8192 2>&1 | tail -1
python3 scripts/floor_probe.py analyse | grep -E "uncond"
for a in "f32 49152 noise0.1" "f32 8192 noise1" "f32 4096 c5" "f64 2048 real" "c128 512 real" "f64 4096 c5"; do python3 scripts/f64_real_probe.py $a 2>&1 | grep "^{" | tail -1 | cut -c1-420; done
