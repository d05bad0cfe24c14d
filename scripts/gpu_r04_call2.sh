#!/bin/bash
# round 4, call 2: BMPSWalker object tests (+ the measurer on it) and the f32 route variants of the error budget
mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_walker.py tests/test_gpu_measure.py -x -q -m gpu > gpurun_out/r04/t2.log 2>&1
echo "pytest rc $?" >> gpurun_out/r04/t2.log
tail -15 gpurun_out/r04/t2.log
ONLY="f32,f32 no ortho polish (round 3),f32 acc64 X,P,f32 acc64 Z,Tt,f32 acc64 M,f32 acc64 Y,f32 acc64 all contractions,f32 acc64 all, no ortho polish,f32 no two-level,f32 no mid route (Jacobi on M),f32 no mid route, no ortho polish,f32 no chain,f32 no rank adapt"
timeout 1500 python scripts/error_budget.py --walkers 64 > gpurun_out/r04/budget2_c4_real.json 2> gpurun_out/r04/budget2_c4_real.err
grep "^f32" gpurun_out/r04/budget2_c4_real.err
