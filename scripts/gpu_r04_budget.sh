#!/bin/bash
# round 4, call 1: ADVICE fixes under test + the f32 error budget by stage on the tiled real state at C4
mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_fullrank.py tests/test_gpu_realrank.py -x -q -m gpu > gpurun_out/r04/t1.log 2>&1
echo "pytest rc $?" >> gpurun_out/r04/t1.log
tail -3 gpurun_out/r04/t1.log
timeout 1500 python scripts/error_budget.py --walkers 64 --oracle 16 > gpurun_out/r04/budget_c4_real.json 2> gpurun_out/r04/budget_c4_real.err
tail -30 gpurun_out/r04/budget_c4_real.err
