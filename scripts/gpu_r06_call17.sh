#!/bin/bash
# round 6, call 17: the whole -m gpu suite, smoke, then the committed profiles again (TRI template, new f64 / complex routes)
cd /root/repo; mkdir -p gpurun_out/r06
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 3300 python -m pytest tests -m gpu -q --tb=short > gpurun_out/r06/gpu_suite3.log 2>&1
echo "suite rc=$?"; grep -vE "^RCCL|^HIP|^ROCm|^Hostname|^Librccl" gpurun_out/r06/gpu_suite3.log | tail -8
export GRAFT_REPO_ROOT=/root/repo
bash scripts/gpu_r06_profiles.sh 2>&1 | tail -12
