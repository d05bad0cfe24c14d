#!/bin/bash
cd /root/repo
timeout 900 python scripts/bigbond_probe.py 8 6 42 2>&1 | tail -2 | cut -c1-400
timeout 900 python scripts/bigbond_probe.py 8 6 48 2>&1 | tail -2 | cut -c1-400
echo "== static shapes"; PEPSGPU_NO_RANK_ADAPT=1 timeout 900 python scripts/bigbond_probe.py 8 6 48 2>&1 | tail -2 | cut -c1-400
echo "== no midroute"; PEPSGPU_NO_MIDROUTE=1 timeout 900 python scripts/bigbond_probe.py 8 6 48 2>&1 | tail -2 | cut -c1-400
echo "== no i8 gram, no f64 route"; PEPSGPU_NO_I8_GRAM=1 PEPSGPU_NO_F64_DENSE_ROUTE=1 timeout 900 python scripts/bigbond_probe.py 8 6 48 2>&1 | tail -2 | cut -c1-400
