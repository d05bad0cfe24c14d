cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/prof2
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof2 -o r2 -- python3 bench.py --walkers 32768 --steps 1 --warmup 1 --no-cpu-baseline --no-route-check > gpurun_out/prof2/bench.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/prof2/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
h = collections.Counter()
for r in rows:
    n = r["Kernel_Name"]
    if "tgemm_direct_kernel<false, false>" in n and r["Grid_Size_Z"] == "32768" or ("tgemm_direct_kernel<false, false>" in n and r.get("Grid_Size", "") == ""):
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        h[min(int(d // 5) * 5, 100)] += 1
print(sorted(h.items()))
print(rows[0].keys())
PY
rm -rf gpurun_out/prof2
