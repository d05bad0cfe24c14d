#!/bin/bash
# round 5, call 26: complex / f64 dense routes with the third-chance factorisation
cd /root/repo; mkdir -p gpurun_out/r05
PEPSGPU_DEBUG_SWEEPS=1 PEPSGPU_DEBUG_VERBOSE=1 timeout 900 python scripts/f64_real_probe.py c128 128 real 2> gpurun_out/r05/c128_route_diag2.err | tail -1
grep "c128 dense route" gpurun_out/r05/c128_route_diag2.err | tail -9 | cut -c1-100
timeout 900 python scripts/f64_real_probe.py c128 512 real 2>&1 | tail -1
timeout 900 python scripts/f64_real_probe.py f64 2048 real 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c128 -o c128 -- python3 /root/repo/scripts/f64_real_probe.py c128 512 real > /tmp/prof_c128.log 2>&1
f=$(find /tmp/prof_c128 -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:5]:
    print(r["Name"][:60], r["Calls"], "%.1f ms total" % (int(r["TotalDurationNs"]) / 1e6), "%.2f ms avg" % (float(r["AverageNs"]) / 1e6), "max %.1f ms" % (int(r["MaxNs"]) / 1e6), r["Percentage"])
PY
