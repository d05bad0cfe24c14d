#!/bin/bash
# round 5, call 55: the device (f64 and c128) against the dense contraction of the reference's 4x4 D=8 fixture (test only; no product change)
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_complex.py -m gpu -q -k "k5_device" 2>&1 | tail -6 | cut -c1-300
