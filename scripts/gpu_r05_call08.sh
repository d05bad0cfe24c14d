#!/bin/bash
# round 5, call 8: (a) the diagonal hop of fermionic states with twisted BTen2 environments (Python product path + C++ host layer);
# (b) the dense float64 truncation route: parity of the f64 mode, then its rate
cd /root/repo
mkdir -p gpurun_out/r05
timeout 1500 python -m pytest tests/test_gpu_fermion.py -m gpu -q -x --tb=short > gpurun_out/r05/call08_tests.log 2>&1
echo "fermion tests rc=$?"; tail -25 gpurun_out/r05/call08_tests.log
bash scripts/gpu_r05_call09.sh
