cd $GRAFT_REPO_ROOT; O=gpurun_out/hist; mkdir -p $O; export TMPDIR=/tmp
COMMON="--steps 1 --warmup 1 --no-cpu-baseline --no-route-check --no-full-rank --no-real-rank --no-sweeps --no-latency --no-energy-check --no-other-modes"
rocprofv3 --kernel-trace --output-format csv -d $O -o kt -- python3 bench.py $COMMON > $O/kt.log 2>&1
python3 scripts/trace_hist.py $O/kt_kernel_trace.csv "$1" 100000
rm -f $O/kt_kernel_trace.csv
