#!/bin/bash
# round 5, call 31: f64 mode against the oracle with and without the dense route, same configurations (two seeds)
cd /root/repo
timeout 1200 python scripts/f64_route_parity.py 64 100000 2>&1 | tail -1
timeout 1200 python scripts/f64_route_parity.py 64 307 2>&1 | tail -1
