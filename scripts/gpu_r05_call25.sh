#!/bin/bash
# round 5, call 25: who leaves the complex dense route
cd /root/repo; mkdir -p gpurun_out/r05
PEPSGPU_DEBUG_SWEEPS=1 PEPSGPU_DEBUG_VERBOSE=1 timeout 900 python scripts/f64_real_probe.py c128 128 real 2> gpurun_out/r05/c128_route_diag.err | tail -1
grep "c128 dense route" gpurun_out/r05/c128_route_diag.err | tail -10 | cut -c1-260
grep "jacobi m=" gpurun_out/r05/c128_route_diag.err | sort | uniq -c | sort -rn | head -5
