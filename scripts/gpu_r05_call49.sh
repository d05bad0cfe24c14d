#!/bin/bash
# round 5, call 49: after the removal of the timing switches from the opt-in Cholesky kernel: smoke, kernel tests, complex suite, real-rank tests
cd /root/repo; mkdir -p gpurun_out/r05
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_complex.py tests/test_gpu_realrank.py -m gpu -q --tb=short 2>&1 | tail -8 | cut -c1-300
