#!/bin/bash
# bisect of test_carry_of_more_than_256_columns over today's library commits
for c in 43d2b61 75ff549 ef5d4ed; do
  echo "== $c"
  PEPSGPU_LIB=$GRAFT_REPO_ROOT/peps_amd/lib/ab/bis_$c.so timeout 600 python -m pytest tests/test_gpu_fullrank.py -m gpu -q -x --tb=line -k "carry_of_more" 2>&1 | tail -4
done
