#!/bin/bash
# round 5, call 57: the whole -m gpu suite with the opt-in resident Cholesky forced on (PEPSGPU_CHOL_RESIDENT=1) -- does anything downstream notice?
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
export PEPSGPU_CHOL_RESIDENT=1
timeout 840 python -m pytest tests -m gpu -q --tb=line -x -k "not test_register_resident" > gpurun_out/r05/gpu_suite_resident.log 2>&1
echo "suite rc=$?"; tail -5 gpurun_out/r05/gpu_suite_resident.log | cut -c1-300
