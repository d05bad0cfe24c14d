# one-walker latency: where the host time of an amplitude goes (HIP API trace + kernel trace, no counters)
cd $GRAFT_REPO_ROOT; O=gpurun_out/n1; mkdir -p $O; export TMPDIR=/tmp
ARGS="--walkers 1 --steps 40 --warmup 5 --no-cpu-baseline --no-route-check --no-full-rank --no-real-rank --no-sweeps --no-latency --no-energy-check --no-other-modes"
python3 bench.py $ARGS 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('n1 ms_per_step', d['ms_per_step'], 'launches_per_step', sum(d['launches_per_step'].values()) if isinstance(d.get('launches_per_step'), dict) else d.get('launches_per_step'))"
rocprofv3 --hip-trace --kernel-trace --stats --output-format csv -d $O -o n1 -- python3 bench.py $ARGS > $O/n1.log 2>&1
ls $O | head -20
python3 - <<'PY'
import csv, glob
for f in sorted(glob.glob("gpurun_out/n1/n1_hip_api_stats.csv")) + sorted(glob.glob("gpurun_out/n1/n1_hip_stats.csv")):
    print(f)
    rows = list(csv.DictReader(open(f)))
    for r in rows[:14]:
        print("  ", {k: r[k] for k in list(r)[:6]})
for f in sorted(glob.glob("gpurun_out/n1/n1_kernel_stats.csv")):
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows); n = sum(int(r["Calls"]) for r in rows)
    print("kernels: calls", n, "total ms", tot / 1e6)
    for r in rows[:8]:
        print("  ", r["Name"][:60], r["Calls"], r["TotalDurationNs"], r["AverageNs"])
PY
rm -f $O/*_trace.csv
