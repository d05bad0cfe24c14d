#!/bin/bash
cd /root/repo
timeout 900 python scripts/bigbond_probe.py 8 7 37 1.0 2>&1 | tail -2 | cut -c1-300
timeout 900 python scripts/bigbond_probe.py 8 7 42 1.0 2>&1 | tail -2 | cut -c1-300
timeout 900 python scripts/bigbond_probe.py 8 5 60 1.0 2>&1 | tail -2 | cut -c1-300
timeout 900 python scripts/bigbond_probe.py 8 8 36 1.0 2>&1 | tail -2 | cut -c1-300
timeout 900 python scripts/chi40_probe.py 36 2>&1 | tail -2 | cut -c1-300
