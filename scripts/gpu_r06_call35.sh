#!/bin/bash
# ring of k-rounds in the wave-per-tile bodies (chained + direct kernels): kernel tests, parity of the f32 legs, rates
mkdir -p gpurun_out/r06
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -m gpu -q -x --tb=short 2>&1 | tail -3
python3 scripts/floor_probe.py run f64 2048 2>&1 | tail -1
python3 scripts/floor_probe.py run f32_ring 8192 2>&1 | tail -1
python3 scripts/floor_probe.py analyse | grep -E "ring"
for a in "f32 8192 real" "f32 49152 noise0.1" "f32 8192 noise1" "f32 4096 c5"; do python3 scripts/f64_real_probe.py $a 2>&1 | grep "^{" | tail -1 | cut -c1-420; done
