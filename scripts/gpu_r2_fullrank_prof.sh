# round 2: where does the time go on a state of realistic / full rank (noise 1.0, 0.3)?
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2a
export TMPDIR=/tmp
for nz in 1.0 0.3; do
  python3 bench.py --noise $nz --walkers ${NW:-2048} --steps 1 --warmup 1 --no-cpu-baseline --no-route-check > gpurun_out/r2a/bench_noise$nz.json 2> gpurun_out/r2a/bench_noise$nz.err
  tail -c 1500 gpurun_out/r2a/bench_noise$nz.json
done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2a -o n10 -- python3 bench.py --noise 1.0 --walkers ${NW:-2048} --steps 1 --warmup 1 --no-cpu-baseline --no-route-check > gpurun_out/r2a/prof_bench.log 2>&1
python3 scripts/trace_summary.py gpurun_out/r2a/n10_kernel_trace.csv > gpurun_out/r2a/trace_summary_noise1.0.txt
head -60 gpurun_out/r2a/trace_summary_noise1.0.txt
find gpurun_out/r2a -name "*kernel_trace.csv" -delete
