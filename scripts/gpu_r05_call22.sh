#!/bin/bash
# round 5, call 22: where the complex element type spends its time on the dense real state (kernel stats, 128 walkers)
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c128 -o c128 -- python3 /root/repo/scripts/f64_real_probe.py c128 128 real > /tmp/prof_c128.log 2>&1
tail -1 /tmp/prof_c128.log
f=$(find /tmp/prof_c128 -name "*kernel_stats.csv" | head -1)
head -12 "$f" | cut -c1-220
