#!/bin/bash
# round 5, call 10: fermion suite after the selector fix (a four-column candidate table on a 2 x 2 lattice was read as a configuration
# table); who leaves the dense f64 route on the real state
cd /root/repo
mkdir -p gpurun_out/r05
timeout 1500 python -m pytest tests/test_gpu_fermion.py -m gpu -q -x --tb=short > gpurun_out/r05/call10_tests.log 2>&1
echo "fermion tests rc=$?"; tail -8 gpurun_out/r05/call10_tests.log
PEPSGPU_DEBUG_SWEEPS=1 PEPSGPU_DEBUG_VERBOSE=1 timeout 600 python scripts/f64_real_probe.py f64 128 real 2> gpurun_out/r05/f64_route_diag.err | tail -1
grep "f64 dense route" gpurun_out/r05/f64_route_diag.err | tail -40 | cut -c1-200
