#!/bin/bash
# round 4, call 3: walker tests again; which property of the Y = Tt V^T launch carries the f32 bias; the drained-accumulator kernel;
# the same on the other two bench states; bf16 split microbenchmark
mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_walker.py tests/test_gpu_measure.py -x -q -m gpu > gpurun_out/r04/t3.log 2>&1
echo "pytest rc $?" >> gpurun_out/r04/t3.log
tail -12 gpurun_out/r04/t3.log
ONLY="f32;f32 Y f32 chain (round 3);f32 round 3 (no ortho polish, Y f32 chain);f32 no ortho polish;f32 acc64 Y;f32 acc64 all contractions;f32 Y on the LDS-tiled f32 kernel;f32 no fused norm;f32 no tt swap;f32 no vector loads"
timeout 900 python scripts/error_budget.py --walkers 64 --only "$ONLY" > gpurun_out/r04/budget3_c4_real.json 2> gpurun_out/r04/budget3_c4_real.err
grep "^f32" gpurun_out/r04/budget3_c4_real.err
ONLY2="f32;f32 Y f32 chain (round 3);f32 round 3 (no ortho polish, Y f32 chain);f32 acc64 all contractions;f64+floors_f32"
for st in synthetic full; do
  timeout 600 python scripts/error_budget.py --walkers 64 --state $st --only "$ONLY2" > gpurun_out/r04/budget3_c4_$st.json 2> gpurun_out/r04/budget3_c4_$st.err
  echo "== $st"; grep "^f" gpurun_out/r04/budget3_c4_$st.err
done
(cd scripts && /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -o build/bf16_split_bench bf16_split_bench.hip 2>/dev/null; ./build/bf16_split_bench 1024 10) > gpurun_out/r04/bf16_split.jsonl 2> gpurun_out/r04/bf16_split.err
cat gpurun_out/r04/bf16_split.jsonl
