#!/bin/bash
# round 6, call 14: float64 dense route from the pivoted factor: parity tests of the f64 mode, rates (real state, C5), diagnostics
cd /root/repo; mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_gpu_realrank.py tests/test_gpu_fermion.py -m gpu -q -x --tb=short -s -k "c4_amplitudes_vs_oracle or f64_dense_truncation_route or c5_spinless or c4_energy_vs_oracle" 2>&1 | grep -E "C5|C4|max|passed|failed|Error|error|assert" | tail -14
for v in 0 1; do
  echo "== PEPSGPU_F64_PIVOT=$v"
  PEPSGPU_F64_PIVOT=$v timeout 900 python scripts/f64_real_probe.py f64 2048 real 2>&1 | tail -1 | cut -c1-330
  PEPSGPU_F64_PIVOT=$v timeout 900 python scripts/f64_real_probe.py f64 4096 c5 2>&1 | tail -1 | cut -c1-330
done
PEPSGPU_DEBUG_SWEEPS=1 PEPSGPU_DEBUG_VERBOSE=1 timeout 900 python scripts/f64_real_probe.py f64 256 c5 2> gpurun_out/r06/f64_pivot_diag_c5.err | tail -1 | cut -c1-200
grep "f64 pivoted route" gpurun_out/r06/f64_pivot_diag_c5.err | tail -6 | cut -c1-220
PEPSGPU_DEBUG_SWEEPS=1 PEPSGPU_DEBUG_VERBOSE=1 timeout 900 python scripts/f64_real_probe.py f64 256 real 2> gpurun_out/r06/f64_pivot_diag_real.err | tail -1 | cut -c1-200
grep "f64 pivoted route" gpurun_out/r06/f64_pivot_diag_real.err | tail -6 | cut -c1-220
