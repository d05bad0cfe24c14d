#!/bin/bash
# round 5 call 47: resident Cholesky after the load fix: kernel tests (forced on), per-block latency, and the default dispatch rule
# (launches of <= 256 walkers) against "never" on the real leg.  Every step under a timeout.
mkdir -p gpurun_out/r05
export TMPDIR=/tmp
PEPSGPU_CHOL_RESIDENT=1 timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "cholesky" 2>&1 | tail -3
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "cholesky" 2>&1 | tail -3
for cfg in "0 0" "1 0"; do
  set -- $cfg
  export PEPSGPU_CHOL_RESIDENT=$1 PEPSGPU_CR_DBG=$2
  rm -rf /tmp/crp
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/crp -o x -- python3 scripts/chol_micro.py 256 > /tmp/crp.log 2>&1
  f=$(find /tmp/crp -name "*kernel_stats.csv" 2>/dev/null | head -1)
  if [ -n "$f" ]; then echo "resident=$1 dbg=$2: $(grep -E 'chol_(resident|blocked)' "$f" | head -1 | sed 's/.*)",//' )"; else echo "resident=$1 dbg=$2: no stats"; fi
done
unset PEPSGPU_CHOL_RESIDENT PEPSGPU_CR_DBG
NW=8192 VAR=PEPSGPU_CHOL_RESIDENT VALS="0 -" timeout 400 bash scripts/ab_real.sh 2>&1 | tail -4
