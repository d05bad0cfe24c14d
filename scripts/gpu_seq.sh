cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/seq
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/seq -o s1 -- python3 bench.py --walkers 512 --steps 1 --warmup 0 --no-cpu-baseline --noise ${NOISE:-0.1} > gpurun_out/seq/bench.log 2>&1
n=$(wc -l < gpurun_out/seq/s1_kernel_trace.csv); echo rows $n
python scripts/trace_seq.py gpurun_out/seq/s1_kernel_trace.csv ${START:-1600} ${COUNT:-260} > gpurun_out/seq/seq.txt
rm -f gpurun_out/seq/s1_kernel_trace.csv
