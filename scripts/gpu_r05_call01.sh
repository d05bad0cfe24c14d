#!/bin/bash
# round 5, call 1: the three staged device tests of round 4 (K8, K9, TFIM registry through the HIP path)
cd /root/repo
mkdir -p gpurun_out/r05
PEPS_STAGED_TESTS=1 timeout 1500 python -m pytest tests/test_gpu_staged.py -m gpu -q -x --tb=long > gpurun_out/r05/staged1.log 2>&1
echo "rc=$?"; tail -60 gpurun_out/r05/staged1.log
