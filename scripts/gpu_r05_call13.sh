#!/bin/bash
# round 5, call 13: why walkers leave the dense f64 route (stage counters), C5 f64 with the 136 KB dynamic-LDS cap
cd /root/repo
mkdir -p gpurun_out/r05
PEPSGPU_DEBUG_SWEEPS=1 PEPSGPU_DEBUG_VERBOSE=1 timeout 600 python scripts/f64_real_probe.py f64 1024 real 2> gpurun_out/r05/f64_route_diag3.err | tail -1
grep "f64 dense route" gpurun_out/r05/f64_route_diag3.err | tail -16 | cut -c1-330
timeout 600 python scripts/f64_real_probe.py f64 4096 c5 2>&1 | tail -1
