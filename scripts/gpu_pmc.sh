# PMC passes for the bench command (separate from the kernel trace, one counter group per pass:
# FETCH_SIZE takes 3 of the 4 TCC slots, WRITE_SIZE 2 -- MI355X_MICROARCH.md, rocprofv3 PMC slots).
# NW = walkers per step (default 512); PMC_TRAFFIC_ONLY=1 skips the SQ pass.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
run_pass() {
  tag=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc -o $tag -- python3 bench.py --walkers ${NW:-512} --steps 1 --warmup 1 --no-cpu-baseline --no-route-check > gpurun_out/pmc/bench_$tag.log 2>&1
  echo "pass $tag rc=$?"
  f=gpurun_out/pmc/${tag}_counter_collection.csv
  if [ -f "$f" ]; then python scripts/pmc_summary.py $f > gpurun_out/pmc/${tag}_summary.txt; head -14 gpurun_out/pmc/${tag}_summary.txt; rm -f $f; fi
}
run_pass FETCH_SIZE FETCH_SIZE
run_pass WRITE_SIZE WRITE_SIZE
if [ -z "$PMC_TRAFFIC_ONLY" ]; then
  run_pass SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY
fi
find gpurun_out/pmc -name "*.csv" -size +2M -delete
