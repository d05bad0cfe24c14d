#!/bin/bash
# round 5 call 42: where the resident Cholesky spends its time (phases switched off one at a time; results of those runs are wrong by design)
# (historical: PEPSGPU_CR_DBG was a timing-only switch of chol_resident_kernel, removed after this measurement; this call also lost 29 GPU-minutes
# to a grep on an empty file name -- rocprofv3 needs --output-format csv for *_kernel_stats.csv)
mkdir -p gpurun_out/r05/cr
export TMPDIR=/tmp
for cfg in "0 0" "1 0" "1 1" "1 2" "1 4" "1 8" "1 7" "1 15"; do
  set -- $cfg
  export PEPSGPU_CHOL_RESIDENT=$1 PEPSGPU_CR_DBG=$2
  rm -rf /tmp/crp; rocprofv3 --kernel-trace --stats -d /tmp/crp -o x -- python3 scripts/chol_micro.py 2048 > /tmp/crp.log 2>&1
  f=$(find /tmp/crp -name "*kernel_stats.csv" | head -1)
  echo "resident=$1 dbg=$2: $(grep -E 'chol_(resident|blocked)' $f | head -1 | cut -d, -f1-4 | cut -c1-40,200-)"
  grep -E 'chol_(resident|blocked)' $f | head -1 | awk -F, '{print "   calls", $(NF-6), "avg_ns", $(NF-4)}'
done
