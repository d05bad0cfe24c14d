#!/bin/bash
# round 6, call 11: one context against two / three contexts on host threads (own streams): real state, headline state, full-rank state
cd /root/repo; mkdir -p gpurun_out/r06
timeout 900 python scripts/two_stream_probe.py real 12288 1 2 3 2>&1 | tail -1
timeout 900 python scripts/two_stream_probe.py synthetic 49152 1 2 4 2>&1 | tail -1
timeout 900 python scripts/two_stream_probe.py full 8192 1 2 2>&1 | tail -1
