#!/bin/bash
# round 5, call 2: staged device tests again (K8 after the order-1 rescale of MCPEPSMeasurer, TFIM registry), the measurer tests, and
# which f32 contraction stage carries what is left of the real-state amplitude error (128 configurations, vs the f64 mode)
cd /root/repo
mkdir -p gpurun_out/r05
PEPS_STAGED_TESTS=1 timeout 900 python -m pytest tests/test_gpu_staged.py tests/test_gpu_measure.py -m gpu -q --tb=short > gpurun_out/r05/staged2.log 2>&1
echo "rc=$?"; tail -40 gpurun_out/r05/staged2.log
ONLY="f32;f32 acc64 X,P;f32 acc64 Z,Tt;f32 acc64 M;f32 acc64 all contractions;f32 acc64 X,P,Z,Tt;f32 acc64 Z,Tt,M;f32 acc64 X,P,Z,Tt,M;f32 no chain"
timeout 1500 python scripts/error_budget.py --walkers 128 --state real --oracle 32 --only "$ONLY" > gpurun_out/r05/budget2_c4_real.json 2> gpurun_out/r05/budget2_c4_real.err
grep "^f32" gpurun_out/r05/budget2_c4_real.err | cut -c1-260
tail -c 600 gpurun_out/r05/budget2_c4_real.json
