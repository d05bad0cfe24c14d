#!/bin/bash
# round 5, call 12: f64 rates after the 136 KB LDS threshold; E_loc rates of the fermionic model with t2 = 0 / t2 != 0 (local, fresh)
cd /root/repo
mkdir -p gpurun_out/r05
timeout 900 python scripts/f64_real_probe.py f64 2048 real 2>&1 | tail -1
PEPSGPU_DEBUG_SWEEPS=1 PEPSGPU_DEBUG_VERBOSE=1 timeout 600 python scripts/f64_real_probe.py f64 256 real 2> gpurun_out/r05/f64_route_diag2.err | tail -1
grep "f64 dense route" gpurun_out/r05/f64_route_diag2.err | tail -11 | cut -c1-200
timeout 600 python scripts/f64_real_probe.py f64 4096 c5 2>&1 | tail -1
timeout 900 python scripts/bench_fermion_nnn.py 512 f32 2>&1 | tail -1
PEPSHOST_NNN_FRESH=1 timeout 1200 python scripts/bench_fermion_nnn.py 512 f32 2>&1 | tail -1
timeout 900 python scripts/bench_fermion_nnn.py 512 f64 2>&1 | tail -1
timeout 900 python -m pytest tests/test_gpu_realrank.py -m gpu -q -x --tb=short -k "f64 or oracle" 2>&1 | tail -3
