# kernel trace of the real_rank leg (tiled optimised state of the reference, C4, 2048 walkers): per-kernel times by grid
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03prof; mkdir -p $O
export TMPDIR=/tmp
TAG=${TAG:-c4_f32_real_nw2048}
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o kt_$TAG -- python3 bench.py --state real --walkers ${NW:-2048} --steps 1 --warmup 1 --no-cpu-baseline --no-route-check --no-energy-check --no-sweeps --no-latency --no-other-modes > $O/kt_$TAG.log 2>&1
python3 scripts/trace_summary.py $O/kt_${TAG}_kernel_trace.csv > $O/r03_kernel_trace_by_grid_$TAG.txt
cp $O/kt_${TAG}_kernel_stats.csv $O/r03_kernel_stats_$TAG.csv
rm -f $O/kt_${TAG}_kernel_trace.csv
head -40 $O/r03_kernel_trace_by_grid_$TAG.txt
