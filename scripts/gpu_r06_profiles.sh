# round 6 evidence: kernel traces and PMC passes of the three bench legs
#   headline  : SURVEY 8(d) state, C4, 49152 walkers (the default of bench.py)
#   full_rank : i.i.d. random site tensors (noise 1.0), 8192 walkers
#   real_rank : the reference's optimised 4x4 D=8 state tiled to 12x12, 12288 walkers
# One counter group per pass (FETCH_SIZE takes 3 of the 4 TCC slots, WRITE_SIZE 2: MI355X_MICROARCH.md); kernel trace in its
# own run; no tracing domains beside --pmc.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06prof; mkdir -p $O
export TMPDIR=/tmp
COMMON="--steps 1 --warmup 1 --no-cpu-baseline --no-route-check --no-full-rank --no-real-rank --no-sweeps --no-latency --no-energy-check --no-other-modes"
leg() { # tag, extra args
  tag=$1; shift
  python3 bench.py $COMMON "$@" > $O/r06_bench_profiled_config_$tag.json 2> $O/bench_$tag.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $O -o kt_$tag -- python3 bench.py $COMMON "$@" > $O/kt_$tag.log 2>&1
  python3 scripts/trace_summary.py $O/kt_${tag}_kernel_trace.csv > $O/r06_kernel_trace_by_grid_$tag.txt
  cp $O/kt_${tag}_kernel_stats.csv $O/r06_kernel_stats_$tag.csv
  rm -f $O/kt_${tag}_kernel_trace.csv
  if [ -n "$KT_ONLY" ]; then head -8 $O/r06_kernel_trace_by_grid_$tag.txt; return; fi    # (kernel traces only: the PMC passes of an earlier call stay)
  for cnt in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $cnt --output-format csv -d $O -o pmc_${cnt}_$tag -- python3 bench.py $COMMON "$@" > $O/pmc_${cnt}_$tag.log 2>&1
    python3 scripts/pmc_summary.py $O/pmc_${cnt}_${tag}_counter_collection.csv > $O/r06_pmc_${cnt}_$tag.txt
    rm -f $O/pmc_${cnt}_${tag}_counter_collection.csv
  done
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O -o pmc_SQ_$tag -- python3 bench.py $COMMON "$@" > $O/pmc_SQ_$tag.log 2>&1
  python3 scripts/pmc_summary.py $O/pmc_SQ_${tag}_counter_collection.csv > $O/r06_pmc_SQ_$tag.txt
  rm -f $O/pmc_SQ_${tag}_counter_collection.csv
  head -8 $O/r06_kernel_trace_by_grid_$tag.txt
}
leg c4_f32_noise0.1_nw49152
leg c4_f32_noise1_nw8192 --noise 1.0 --walkers 8192
leg c4_f32_real_nw12288 --state real --walkers 12288
if [ -n "$KT_ONLY" ]; then find $O -name "*.csv" -size +3M -delete; exit 0; fi
# kernel trace of the Monte-Carlo sweeps (1 + 2 sweeps of the exchange updater through the C++ host layer, 8192 walkers of the headline state)
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o kt_sweep -- python3 scripts/sweep_trace.py 8192 > $O/kt_sweep.log 2>&1
python3 scripts/trace_summary.py $O/kt_sweep_kernel_trace.csv > $O/r06_kernel_trace_by_grid_sweep_c4_f32_noise0.1_nw8192.txt
rm -f $O/kt_sweep_kernel_trace.csv
head -8 $O/r06_kernel_trace_by_grid_sweep_c4_f32_noise0.1_nw8192.txt
find $O -name "*.csv" -size +3M -delete
ls $O | head -60
