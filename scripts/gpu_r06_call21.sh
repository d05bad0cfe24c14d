#!/bin/bash
cd /root/repo
timeout 600 python scripts/bigbond_probe.py 6 4 64 2>&1 | tail -2
timeout 600 python scripts/bigbond_probe.py 6 4 70 2>&1 | tail -2
timeout 600 python scripts/bigbond_probe.py 6 4 100 2>&1 | tail -2
echo "== static shapes"; PEPSGPU_NO_RANK_ADAPT=1 timeout 600 python scripts/bigbond_probe.py 6 4 70 2>&1 | tail -2
echo "== no dense f64 route"; PEPSGPU_NO_F64_DENSE_ROUTE=1 timeout 600 python scripts/bigbond_probe.py 6 4 70 2>&1 | tail -2
echo "== no midroute"; PEPSGPU_NO_MIDROUTE=1 timeout 600 python scripts/bigbond_probe.py 6 4 70 2>&1 | tail -2
