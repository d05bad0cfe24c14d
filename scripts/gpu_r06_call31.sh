#!/bin/bash
# phase timing of chol_blocked_kernel (instrumented builds outside the product tree): single panels vs pairs, alone and at full occupancy
for v in ph_single ph_pair; do
  export PEPSGPU_LIB=$GRAFT_REPO_ROOT/peps_amd/lib/ab/$v.so
  for nb in 256 768 4096; do echo "== $v nb $nb"; python3 scripts/chol_micro.py $nb graded 2>&1 | grep -E "chb|^ok" | tail -2; done
done
