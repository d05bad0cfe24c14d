#!/bin/bash
# round 5 call 40: complex variational compression + the complex host layer again (after the rebuild), variational real tests as regression
mkdir -p gpurun_out/r05
timeout 1500 python -m pytest tests/test_gpu_complex.py tests/test_gpu_variational.py -q -m gpu 2>&1 | tail -60 > gpurun_out/r05/call40_tests.log
echo "tests rc=$?"; tail -60 gpurun_out/r05/call40_tests.log | cut -c1-300
