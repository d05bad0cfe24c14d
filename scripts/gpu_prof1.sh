cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/prof_r1
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r1 -o r1 -- python3 bench.py --walkers 256 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/prof_r1/bench.log 2>&1
tail -2 gpurun_out/prof_r1/bench.log
find gpurun_out/prof_r1 -name "*stats*" | head; 
f=$(find gpurun_out/prof_r1 -name "*kernel_stats.csv" | head -1); head -30 "$f"
# keep only the small summaries
find gpurun_out/prof_r1 -name "*kernel_trace.csv" -size +20M -delete
timeout 1200 python bench.py --walkers 512 --steps 2 --warmup 1 2>&1 | tail -1 | tee gpurun_out/bench_full.json
