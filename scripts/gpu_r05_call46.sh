#!/bin/bash
# round 5 call 46: where the resident Cholesky spends its time: 256 order-256 walkers (one block per CU, one round), phases switched off one at
# a time (PEPSGPU_CR_DBG; the results of those runs are wrong by design).  Every step under a timeout; no grep on an empty file name.
# (historical: the PEPSGPU_CR_DBG switch was removed from the kernel after this measurement; results in profiles/r05_chol_resident_phases.txt)
mkdir -p gpurun_out/r05
export TMPDIR=/tmp
for cfg in "0 0" "1 0" "1 1" "1 2" "1 4" "1 8" "1 16" "1 31"; do
  set -- $cfg
  export PEPSGPU_CHOL_RESIDENT=$1 PEPSGPU_CR_DBG=$2
  rm -rf /tmp/crp
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/crp -o x -- python3 scripts/chol_micro.py 256 > /tmp/crp.log 2>&1
  f=$(find /tmp/crp -name "*kernel_stats.csv" 2>/dev/null | head -1)
  if [ -n "$f" ]; then echo "resident=$1 dbg=$2: $(grep -E 'chol_(resident|blocked)' "$f" | head -1 | sed 's/.*)",//' )"; else echo "resident=$1 dbg=$2: no stats"; fi
  if [ "$2" = "0" ]; then grep -v "^W2026" /tmp/crp.log | tail -4; head -3 "$f" | cut -c1-300; fi
done | tee gpurun_out/r05/call46_phases.txt
