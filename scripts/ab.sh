#!/bin/bash
# A/B of one environment switch on the default bench: ab.sh VAR v1 v2 ... ; prints value and kernel_ms per setting
var=$1; shift
for v in "$@"; do
  env $var=$v python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-route-check 2>/dev/null | V=$v python -c "
import sys,json,os; d=json.loads(sys.stdin.readlines()[-1]); print(os.environ['V'], round(d['value']), d['kernel_ms'])"
done
