#!/bin/bash
# chol_blocked_kernel v2 (thread = column, LDS-DMA panel, deeper update prefetch): phases and end-to-end A/B against HEAD's
mkdir -p gpurun_out/r06
python3 scripts/floor_probe.py run f64 2048 2>&1 | tail -1
for v in ph_pfd3 ph_pfd4 ph_pfd6; do
  export PEPSGPU_LIB=$GRAFT_REPO_ROOT/peps_amd/lib/ab/$v.so
  for nb in 256 4096; do echo "== $v nb $nb"; python3 scripts/chol_micro.py $nb graded 2>&1 | grep -E "chb|^ok" | tail -2; done
done
for v in orig pfd3 pfd4 pfd6; do
  export PEPSGPU_LIB=$GRAFT_REPO_ROOT/peps_amd/lib/ab/$v.so
  echo "== $v"
  timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --tb=short -k "chol" 2>&1 | tail -2
  python3 scripts/floor_probe.py run f32_$v 8192 2>&1 | tail -1
done
unset PEPSGPU_LIB
python3 scripts/floor_probe.py analyse | grep -E "orig|pfd"
