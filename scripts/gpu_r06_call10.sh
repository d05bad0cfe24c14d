#!/bin/bash
# round 6, call 10: the whole -m gpu suite after the prune (90 -> 21 environment switches, dead kernels removed)
cd /root/repo; mkdir -p gpurun_out/r06
timeout 3300 python -m pytest tests -m gpu -q --tb=short -x > gpurun_out/r06/gpu_suite2.log 2>&1
echo "suite rc=$?"; grep -vE "^RCCL|^HIP|^ROCm|^Hostname|^Librccl" gpurun_out/r06/gpu_suite2.log | tail -12
