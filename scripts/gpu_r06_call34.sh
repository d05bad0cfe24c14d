#!/bin/bash
# branch-free operand loads of the LDS-tiled GEMMs + chol_blocked v2: kernel tests, rates of every mode
mkdir -p gpurun_out/r06
timeout 1200 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --tb=short 2>&1 | tail -3
python3 scripts/floor_probe.py run f64 2048 2>&1 | tail -1
python3 scripts/floor_probe.py run f32_new 8192 2>&1 | tail -1
python3 scripts/floor_probe.py analyse | grep -E "new"
for a in "f64 2048 real" "c128 512 real" "f64 4096 c5" "f32 4096 c5" "f32 8192 real"; do python3 scripts/f64_real_probe.py $a 2>&1 | grep "^{" | tail -1 | cut -c1-330; done
