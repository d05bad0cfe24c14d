#!/bin/bash
# round 5, call 21: the two new tests (three new ABI calls on a bosonic state; dense f64 route against the general kernels)
cd /root/repo
mkdir -p gpurun_out/r05
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_realrank.py -m gpu -q -x --tb=short -s -k "plaquette_trace or dense_truncation_route" 2>&1 | tail -15
