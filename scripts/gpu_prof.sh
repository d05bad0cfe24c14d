cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/prof
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r1 -- python3 bench.py --walkers ${NW:-512} --steps 1 --warmup 1 --no-cpu-baseline --no-route-check > gpurun_out/prof/bench.log 2>&1
python scripts/trace_summary.py gpurun_out/prof/r1_kernel_trace.csv > gpurun_out/prof/trace_summary.txt
head -45 gpurun_out/prof/trace_summary.txt
head -3 gpurun_out/prof/r1_kernel_trace.csv | cut -c1-600
find gpurun_out/prof -name "*kernel_trace.csv" -delete
