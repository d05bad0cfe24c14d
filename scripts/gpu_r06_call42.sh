#!/bin/bash
# LDS Jacobi, rows shorter than 193: LDS-typed plain loops (A) against the generic pointer of rounds 1-5 (B); rows of 256: registers
for v in jacC jacD jacC jacD; do
  export PEPSGPU_LIB=$GRAFT_REPO_ROOT/peps_amd/lib/ab/$v.so
  echo "== $v"
  for a in "f64 2048 real" "f64 4096 c5"; do python3 scripts/f64_real_probe.py $a 2>&1 | grep "^{" | tail -1 | cut -c1-420; done
done
unset PEPSGPU_LIB
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --tb=short -k "jacobi" 2>&1 | tail -2
timeout 2000 python -m pytest tests/test_gpu_realrank.py tests/test_gpu_fermion.py tests/test_gpu_measure.py tests/test_gpu_complex.py -m gpu -q -x --tb=short 2>&1 | tail -2
