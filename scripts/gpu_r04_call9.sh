#!/bin/bash
# round 4, call 9: what the exact-integer Grams cost in amplitude error (128 configurations, three states)
cd /root/repo
mkdir -p gpurun_out/r04
ONLY="f32;f32 Grams on the f64 matrix cores (no i8);f32 i8 column Gram only"
for st in real full synthetic; do
  timeout 900 python scripts/error_budget.py --walkers 128 --state $st --only "$ONLY" > gpurun_out/r04/budget9_c4_$st.json 2> gpurun_out/r04/budget9_c4_$st.err
  echo "== $st"; grep "^f32" gpurun_out/r04/budget9_c4_$st.err | cut -c1-260
done
