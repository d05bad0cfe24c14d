cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2f
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2f -o n10 -- python3 bench.py --noise 1.0 --walkers 2048 --steps 1 --warmup 1 --no-cpu-baseline --no-route-check --no-full-rank > gpurun_out/r2f/prof_bench.log 2>&1
python3 scripts/trace_summary.py gpurun_out/r2f/n10_kernel_trace.csv > gpurun_out/r2f/trace_summary_noise1.0.txt
head -50 gpurun_out/r2f/trace_summary_noise1.0.txt
find gpurun_out/r2f -name "*kernel_trace.csv" -delete
