#!/bin/bash
# round 4, after the register-pair form of the Jacobi kernels: the SQ counter pass of the real-state and headline legs again
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04prof; mkdir -p $O
export TMPDIR=/tmp
COMMON="--steps 1 --warmup 1 --no-cpu-baseline --no-route-check --no-full-rank --no-real-rank --no-sweeps --no-latency --no-energy-check --no-other-modes"
sq() { tag=$1; shift
  timeout 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O -o pmc_SQ_$tag -- python3 bench.py $COMMON "$@" > $O/pmc_SQ_$tag.log 2>&1
  python3 scripts/pmc_summary.py $O/pmc_SQ_${tag}_counter_collection.csv > $O/r04_pmc_SQ_$tag.txt
  rm -f $O/pmc_SQ_${tag}_counter_collection.csv
  head -12 $O/r04_pmc_SQ_$tag.txt | cut -c1-220
}
sq c4_f32_real_nw8192 --state real --walkers 8192
sq c4_f32_noise0.1_nw49152
find $O -name "*.csv" -size +3M -delete
