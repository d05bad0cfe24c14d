#!/bin/bash
cd /root/repo
timeout 2400 python scripts/fuzz_probe.py 30 1 2>&1 | tail -32 | cut -c1-200
