#!/bin/bash
# round 5, call 29: C5 f64 under a looser guard of the dense route (knob PEPSGPU_F64_ROUTE_TOL; default 1e-10): rate and parity
cd /root/repo; mkdir -p gpurun_out/r05
for tol in 1e-10 1e-9 1e-8; do
  echo "== PEPSGPU_F64_ROUTE_TOL=$tol"
  PEPSGPU_F64_ROUTE_TOL=$tol timeout 600 python scripts/f64_real_probe.py f64 4096 c5 2>&1 | tail -1 | cut -c1-200
  PEPSGPU_F64_ROUTE_TOL=$tol timeout 900 python -m pytest tests/test_gpu_fermion.py -m gpu -q -x --tb=short -s -k "c5_spinless and f64" 2>&1 | grep -E "C5 f64|passed|failed" | cut -c1-200
done
