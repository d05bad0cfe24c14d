# raw write-request counters of the L2 -> fabric path: how many of the writes are full 64-byte requests
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/pmcwr
export TMPDIR=/tmp
rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d gpurun_out/pmcwr -o wr -- python3 bench.py --walkers ${NW:-32768} --steps 1 --warmup 1 --no-cpu-baseline --no-route-check --no-full-rank --no-energy-check > gpurun_out/pmcwr/bench.log 2>&1
echo rc=$?
f=gpurun_out/pmcwr/wr_counter_collection.csv
ls -la gpurun_out/pmcwr/
if [ -f "$f" ]; then python scripts/pmc_summary.py $f > gpurun_out/pmcwr/wr_summary.txt; head -30 gpurun_out/pmcwr/wr_summary.txt; rm -f $f; fi
tail -3 gpurun_out/pmcwr/bench.log | cut -c1-300
