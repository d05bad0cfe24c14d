#!/bin/bash
cd /root/repo
echo "== default chi 36"; timeout 600 python scripts/chi40_probe.py 36 2>&1 | tail -2 | cut -c1-420
echo "== no f64 route"; PEPSGPU_NO_F64_DENSE_ROUTE=1 timeout 900 python scripts/chi40_probe.py 36 2>&1 | tail -2 | cut -c1-420
echo "== static shapes"; PEPSGPU_NO_RANK_ADAPT=1 timeout 900 python scripts/chi40_probe.py 36 2>&1 | tail -2 | cut -c1-420
echo "== no midroute"; PEPSGPU_NO_MIDROUTE=1 timeout 900 python scripts/chi40_probe.py 36 2>&1 | tail -2 | cut -c1-420
echo "== no i8"; PEPSGPU_NO_I8_GRAM=1 timeout 900 python scripts/chi40_probe.py 36 2>&1 | tail -2 | cut -c1-420
