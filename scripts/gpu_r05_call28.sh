#!/bin/bash
# round 5, call 28: complex Cholesky on 1024 threads: complex suite, route test, rates
cd /root/repo; mkdir -p gpurun_out/r05
timeout 1500 python -m pytest tests/test_gpu_complex.py tests/test_gpu_realrank.py -m gpu -q -x --tb=short -k "complex or c128" 2>&1 | tail -3
timeout 900 python scripts/f64_real_probe.py c128 512 real 2>&1 | tail -1
timeout 900 python scripts/f64_real_probe.py c128 2048 real 2>&1 | tail -1
