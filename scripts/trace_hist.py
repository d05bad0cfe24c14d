"""Duration distribution of the launches of the kernels whose name contains a pattern (rocprofv3 kernel_trace.csv).
usage: trace_hist.py trace.csv pattern [min_grid]"""
import csv, sys
f, pat = sys.argv[1], sys.argv[2]
min_grid = int(sys.argv[3]) if len(sys.argv) > 3 else 0
by = {}
with open(f) as fh:
    for r in csv.DictReader(fh):
        if pat not in r["Kernel_Name"]:
            continue
        g = int(r.get("Grid_Size_X", r.get("Grid_Size", "0")))
        if g < min_grid:
            continue
        name = r["Kernel_Name"].split("(")[0][-50:]
        by.setdefault((name, g), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, d in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    d.sort()
    n = len(d)
    print(k, "n", n, "sum ms %.2f" % (sum(d) / 1e3), "us: min %.1f p25 %.1f p50 %.1f p75 %.1f p90 %.1f max %.1f" %
          (d[0], d[n // 4], d[n // 2], d[3 * n // 4], d[9 * n // 10], d[-1]))
