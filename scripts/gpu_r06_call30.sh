#!/bin/bash
# chol_blocked_kernel with panels in pairs: A/B of three builds (orig = HEAD, pair_b3 = pairs at 3 blocks per CU, pair_b2 = pairs at 2)
mkdir -p gpurun_out/r06
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 scripts/floor_probe.py run f64 2048 2>&1 | tail -1
for v in orig pair_b3 pair_b2; do
  export PEPSGPU_LIB=$GRAFT_REPO_ROOT/peps_amd/lib/ab/$v.so
  echo "== $v"
  timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --tb=short -k "chol" 2>&1 | tail -3
  rm -rf /tmp/prof_$v; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -- python3 scripts/chol_micro.py 4096 graded 2>&1 | grep "^ok"
  f=$(find /tmp/prof_$v -name "*kernel_stats.csv" | head -1); grep chol_blocked $f | cut -c1-60,200-400
  python3 scripts/floor_probe.py run f32_$v 8192 2>&1 | tail -1
done
unset PEPSGPU_LIB
python3 scripts/floor_probe.py analyse | grep -v "_c"
