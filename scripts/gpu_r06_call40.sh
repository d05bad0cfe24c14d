#!/bin/bash
# general Jacobi kernel with LDS-typed / register-resident rows: kernel tests, f64 + complex + C5 rates; then the rest of the GPU suite
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --tb=short -k "jacobi" 2>&1 | tail -3
for a in "f64 2048 real" "f64 4096 c5" "c128 512 real" "f32 8192 real"; do python3 scripts/f64_real_probe.py $a 2>&1 | grep "^{" | tail -1 | cut -c1-420; done
timeout 3000 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -8 | tee gpurun_out/r06/suite_call40.txt
