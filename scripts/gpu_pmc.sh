# PMC passes for the bench command (separate from the kernel trace, one counter group per pass:
# FETCH_SIZE takes 3 of the 4 TCC slots, WRITE_SIZE 2 -- MI355X_MICROARCH.md, rocprofv3 PMC slots).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
rocprofv3 -L > gpurun_out/pmc/counters_list.txt 2>&1
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --output-format csv -d gpurun_out/pmc -o $tag -- python3 bench.py --walkers 512 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/pmc/bench_$tag.log 2>&1
  echo "pass $tag rc=$?"
  ls gpurun_out/pmc | head
  f=$(ls gpurun_out/pmc/${tag}_counter_collection.csv 2>/dev/null)
  if [ -n "$f" ]; then python scripts/pmc_summary.py $f > gpurun_out/pmc/${tag}_summary.txt; head -30 gpurun_out/pmc/${tag}_summary.txt; rm -f $f; fi
done
find gpurun_out/pmc -name "*.csv" -size +2M -delete
