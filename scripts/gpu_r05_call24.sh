#!/bin/bash
# round 5, call 24: kernel stats of the complex dense route (512 walkers)
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c128 -o c128 -- python3 /root/repo/scripts/f64_real_probe.py c128 512 real > /tmp/prof_c128.log 2>&1
tail -1 /tmp/prof_c128.log
f=$(find /tmp/prof_c128 -name "*kernel_stats.csv" | head -1)
head -9 "$f" | cut -c1-110,300-420
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:8]:
    print(r["Name"][:60], r["Calls"], "%.1f ms total" % (int(r["TotalDurationNs"]) / 1e6), "%.2f ms avg" % (float(r["AverageNs"]) / 1e6), r["Percentage"])
PY
