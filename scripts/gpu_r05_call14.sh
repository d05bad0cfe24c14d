#!/bin/bash
# round 5, call 14: rocprofv3 kernel traces + PMC passes (FETCH_SIZE, WRITE_SIZE, SQ) of the three bench legs -> profiles/r05_*
cd /root/repo
export GRAFT_REPO_ROOT=/root/repo
bash scripts/gpu_r05_profiles.sh > gpurun_out/r05prof_run.log 2>&1
tail -30 gpurun_out/r05prof_run.log
ls gpurun_out/r05prof | head -50
