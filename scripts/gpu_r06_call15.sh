#!/bin/bash
# round 6, call 15: the pivoted f64 route on smaller blocks too (m > MLO): rates and parity
cd /root/repo; mkdir -p gpurun_out/r06
for v in 128 63 47; do
  echo "== PEPSGPU_F64_PIVOT_MLO=$v"
  PEPSGPU_F64_PIVOT_MLO=$v timeout 900 python scripts/f64_real_probe.py f64 2048 real 2>&1 | tail -1 | cut -c1-330
  PEPSGPU_F64_PIVOT_MLO=$v timeout 900 python scripts/f64_real_probe.py f64 4096 c5 2>&1 | tail -1 | cut -c1-330
  PEPSGPU_F64_PIVOT_MLO=$v timeout 1500 python -m pytest tests/test_gpu_realrank.py tests/test_gpu_fermion.py -m gpu -q -x --tb=short -s -k "c4_amplitudes_vs_oracle or c5_spinless" 2>&1 | grep -E "C5 f64|C4 real state f64|passed|failed" | tail -4
done
