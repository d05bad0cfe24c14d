#!/bin/bash
# final tree: driver-style bench run, then the whole GPU suite
mkdir -p gpurun_out/r06
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06/bench_driver_final2.json 2> gpurun_out/r06/bench_driver_final2.err
tail -c 200 gpurun_out/r06/bench_driver_final2.json
timeout 3000 python -m pytest tests -m gpu -q --tb=short 2>&1 | grep -E "FAILED|passed|failed" | tail -6 | tee gpurun_out/r06/suite_final.txt
