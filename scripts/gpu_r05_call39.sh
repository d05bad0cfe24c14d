#!/bin/bash
# round 5 call 38: QLTEN_Complex measurement solvers / MCPEPSMeasurer / ExactSumMeasurer / fermionic models of the host layer
mkdir -p gpurun_out/r05
timeout 1500 python -m pytest tests/test_gpu_complex.py tests/test_gpu_measure.py tests/test_gpu_fermion.py -q -m gpu 2>&1 | tail -40 > gpurun_out/r05/call39_tests.log
echo "tests rc=$?"; tail -40 gpurun_out/r05/call39_tests.log
