#!/bin/bash
# A/B of the Householder factor's noise-floor multiplier (PEPSGPU_QR_NOISE) on the headline workload
for nz in "$@"; do
  PEPSGPU_QR_FACTOR=1 PEPSGPU_QR_NOISE=$nz python bench.py --steps 2 --warmup 1 --cpu-seconds 1 2>/dev/null | NZ=$nz python -c "
import sys,json,os; d=json.loads(sys.stdin.readlines()[-1]); print(os.environ['NZ'], round(d['value']), d['kernel_ms']['cholesky'], d['kernel_ms']['contract'], d['workload_rank']['carry_live_fraction'], d.get('parity_on_sample'))"
done
