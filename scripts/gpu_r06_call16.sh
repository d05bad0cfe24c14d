#!/bin/bash
# round 6, call 16: complex dense route from the randomised range finder: complex suite, route test, rates
cd /root/repo; mkdir -p gpurun_out/r06
timeout 2400 python -m pytest tests/test_gpu_complex.py tests/test_gpu_realrank.py -m gpu -q -x --tb=short -s -k "complex or c128" 2>&1 | grep -E "c128|complex|max|passed|failed|Error|assert" | tail -12
for v in 0 1; do
  echo "== PEPSGPU_F64_PIVOT=$v"
  PEPSGPU_F64_PIVOT=$v timeout 900 python scripts/f64_real_probe.py c128 512 real 2>&1 | tail -1 | cut -c1-300
done
timeout 900 python scripts/f64_real_probe.py c128 2048 real 2>&1 | tail -1 | cut -c1-300
