#!/bin/bash
cd /root/repo
timeout 900 python scripts/bigbond_probe.py 8 8 36 1.0 2>&1 | tail -2 | cut -c1-420
timeout 900 python scripts/bigbond_probe.py 8 8 36 0.1 2>&1 | tail -2 | cut -c1-420
timeout 900 python scripts/bigbond_probe.py 8 8 33 1.0 2>&1 | tail -2 | cut -c1-420
timeout 900 python scripts/chi40_probe.py 32 2>&1 | tail -2 | cut -c1-420
timeout 900 python scripts/chi40_probe.py 33 2>&1 | tail -2 | cut -c1-420
