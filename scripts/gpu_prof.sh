cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/prof
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r1 -- python3 bench.py --walkers 512 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/prof/bench.log 2>&1
tail -1 gpurun_out/prof/bench.log | cut -c1-300
find gpurun_out/prof -type f | head -20
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); cat "$f" | cut -c1-220 | head -30
find gpurun_out/prof -name "*kernel_trace.csv" -delete
find gpurun_out/prof -name "*.db" -delete
