#!/bin/bash
cd /root/repo
echo "== default"; timeout 600 python scripts/chi40_probe.py 40 2>&1 | tail -2
echo "== round-5 routes"; PEPSGPU_PIVOT_CHOL=0 PEPSGPU_ROWS_QR=0 PEPSGPU_TRI=0 PEPSGPU_F64_PIVOT=0 timeout 600 python scripts/chi40_probe.py 40 2>&1 | tail -2
echo "== chi 36 default"; timeout 600 python scripts/chi40_probe.py 36 2>&1 | tail -2
