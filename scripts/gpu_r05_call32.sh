#!/bin/bash
# round 5, call 32: pivot threshold of the first factorisation of the dense routes (knob): rate and parity against the oracle
cd /root/repo
for ts in 1 16 64; do
  echo "== PEPSGPU_ROUTE_THRESH_SCALE=$ts"
  PEPSGPU_ROUTE_THRESH_SCALE=$ts timeout 900 python scripts/f64_real_probe.py f64 2048 real 2>&1 | tail -1 | cut -c1-330
  PEPSGPU_ROUTE_THRESH_SCALE=$ts timeout 900 python scripts/f64_real_probe.py c128 512 real 2>&1 | tail -1 | cut -c1-120
  PEPSGPU_ROUTE_THRESH_SCALE=$ts timeout 1200 python scripts/f64_route_parity.py 32 307 2>&1 | tail -1
done
