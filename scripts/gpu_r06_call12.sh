#!/bin/bash
# round 6, call 12: the committed profiles of the round (kernel traces + FETCH / WRITE / SQ passes of the three bench legs)
export GRAFT_REPO_ROOT=/root/repo
bash scripts/gpu_r06_profiles.sh 2>&1 | tail -40
