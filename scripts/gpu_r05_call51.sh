#!/bin/bash
# round 5, call 51: PMC passes of the real leg with the resident Cholesky forced on (PEPSGPU_CHOL_RESIDENT=1): HBM bytes and SQ counters of
# chol_resident_kernel, for the account of VERDICT r04 item 2b.  One counter group per pass, no tracing domains; every step under a timeout.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05prof; mkdir -p $O
export TMPDIR=/tmp PEPSGPU_CHOL_RESIDENT=1
COMMON="--steps 1 --warmup 1 --no-cpu-baseline --no-route-check --no-full-rank --no-real-rank --no-sweeps --no-latency --no-energy-check --no-other-modes --state real --walkers 8192"
tag=chol_resident
for cnt in FETCH_SIZE WRITE_SIZE; do
  timeout 240 rocprofv3 --pmc $cnt --output-format csv -d $O -o pmc_${cnt}_$tag -- python3 bench.py $COMMON > $O/pmc_${cnt}_$tag.log 2>&1
  f=$O/pmc_${cnt}_${tag}_counter_collection.csv
  if [ -s "$f" ]; then timeout 120 python3 scripts/pmc_summary.py $f > $O/r05_pmc_${cnt}_$tag.txt; rm -f $f; grep -E "chol_" $O/r05_pmc_${cnt}_$tag.txt | head -3 | cut -c1-260; else echo "no $cnt file"; tail -3 $O/pmc_${cnt}_$tag.log; fi
done
timeout 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O -o pmc_SQ_$tag -- python3 bench.py $COMMON > $O/pmc_SQ_$tag.log 2>&1
f=$O/pmc_SQ_${tag}_counter_collection.csv
if [ -s "$f" ]; then timeout 120 python3 scripts/pmc_summary.py $f > $O/r05_pmc_SQ_$tag.txt; rm -f $f; grep -E "chol_" $O/r05_pmc_SQ_$tag.txt | head -3 | cut -c1-400; else echo "no SQ file"; tail -3 $O/pmc_SQ_$tag.log; fi
find $O -name "*.csv" -size +3M -delete
