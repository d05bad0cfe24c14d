#!/bin/bash
# round 5, call 36: smoke + the whole -m gpu suite on the final tree
cd /root/repo; mkdir -p gpurun_out/r05
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 3000 python -m pytest tests -m gpu -q --tb=short > gpurun_out/r05/gpu_suite_final.log 2>&1
echo "suite rc=$?"; tail -4 gpurun_out/r05/gpu_suite_final.log
