#!/bin/bash
# round 5, call 53: complex row Jacobi with both rows of a pair in registers and two pairs per wave in flight: complex suite, the route test, the rate
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
timeout 600 python -m pytest tests/test_gpu_complex.py tests/test_gpu_sr.py -m gpu -q 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_realrank.py -m gpu -q -k "c128" 2>&1 | tail -3
timeout 300 python scripts/f64_real_probe.py c128 512 real 2>&1 | grep -v "^\[pepsgpu\]" | tail -2 | cut -c1-400
