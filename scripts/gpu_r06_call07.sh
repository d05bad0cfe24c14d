#!/bin/bash
# round 6, call 07: the whole -m gpu suite on the new defaults (pivoted first compression, rows_qr, triangular carry)
cd /root/repo; mkdir -p gpurun_out/r06
timeout 3300 python -m pytest tests -m gpu -q --tb=short -x > gpurun_out/r06/gpu_suite1.log 2>&1
echo "suite rc=$?"; tail -8 gpurun_out/r06/gpu_suite1.log
