#!/bin/bash
# round 5, call 37: the tail of the f32-vs-f64 error distribution on the real state (2 048 configurations)
cd /root/repo
timeout 900 python scripts/f32_tail_probe.py 2048 2>&1 | tail -1
