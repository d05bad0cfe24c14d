#!/bin/bash
# round 5, call 23: the dense route for the complex element type: rates, agreement with the general kernel, the complex suite
cd /root/repo
mkdir -p gpurun_out/r05
timeout 900 python scripts/f64_real_probe.py c128 128 real 2>&1 | tail -1
timeout 900 python scripts/f64_real_probe.py c128 512 real 2>&1 | tail -1
timeout 2400 python -m pytest tests/test_gpu_realrank.py tests/test_gpu_complex.py -m gpu -q -x --tb=short -s -k "c128 or complex" 2>&1 | tail -8
