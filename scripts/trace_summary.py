"""Aggregate a rocprofv3 kernel_trace.csv by (kernel, grid) to separate the GEMM shapes of one template."""
import csv, sys, collections
f = sys.argv[1]
agg = collections.defaultdict(lambda: [0, 0.0])
with open(f) as fh:
    rd = csv.DictReader(fh)
    for r in rd:
        name = r["Kernel_Name"].split("(")[0][-96:]      # (60 until round 5: the template lists grew)
        key = (name, r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""), r.get("Workgroup_Size_X", ""))
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        agg[key][0] += 1
        agg[key][1] += dur
tot = sum(v[1] for v in agg.values())
print("total ms", tot)
# every group that carries >= 0.02 % of the time or >= 16 launches (round 4: the top-40 cut dropped the small template variants of the
# chained contraction, so make_pmc_meta.py under-counted its launches per step on the real-state leg and bench.py quoted no traffic)
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
for k, v in [kv for i, kv in enumerate(rows) if i < 40 or kv[1][1] >= 2e-4 * tot or kv[1][0] >= 16][:160]:
    print("%8.2f ms %5d calls %7.1f us/call  %s" % (v[1], v[0], 1e3 * v[1] / v[0], k))
