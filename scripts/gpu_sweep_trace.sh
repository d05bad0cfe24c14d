cd $GRAFT_REPO_ROOT; O=gpurun_out/r03prof; mkdir -p $O
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o kt_sweep_real -- python3 scripts/bench_sweep_real.py ${NW:-512} > $O/kt_sweep_real.log 2>&1
python3 scripts/trace_summary.py $O/kt_sweep_real_kernel_trace.csv > $O/r03_kernel_trace_by_grid_sweep_real.txt
rm -f $O/kt_sweep_real_kernel_trace.csv
tail -2 $O/kt_sweep_real.log
head -30 $O/r03_kernel_trace_by_grid_sweep_real.txt
