#!/bin/bash
# round 5, call 59: the complex measurer test with the oracle's statistics module (test-only change)
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_measure.py -m gpu -q -k "mc_measurer_identical" 2>&1 | tail -4 | cut -c1-300
