#!/bin/bash
# round 5, call 52: do the numeric launch knobs of rounds 3-4 still sit at their optimum on the final build?  (real leg, 8192 walkers; each run under a timeout)
cd $GRAFT_REPO_ROOT
NW=8192 VAR=PEPSGPU_CHAIN_DENSE_LDS VALS="- 16384" timeout 300 bash scripts/ab_real.sh 2>&1 | tail -2 | cut -c1-330
NW=8192 VAR=PEPSGPU_CHB_PF VALS="3" timeout 200 bash scripts/ab_real.sh 2>&1 | tail -1 | cut -c1-330
NW=8192 VAR=PEPSGPU_MID_MINB VALS="3" timeout 200 bash scripts/ab_real.sh 2>&1 | tail -1 | cut -c1-330
NW=8192 VAR=PEPSGPU_COLGRAM_RCAP VALS="96" timeout 200 bash scripts/ab_real.sh 2>&1 | tail -1 | cut -c1-330
