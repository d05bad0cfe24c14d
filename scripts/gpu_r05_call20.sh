#!/bin/bash
# round 5, call 20: the 16 x 16 x 4 tile body of the chained kernel: kernel test, then headline A/B (off / on) with SQ counters of both
cd /root/repo
mkdir -p gpurun_out/r05
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --tb=short -k "chained" 2>&1 | tail -3
export GRAFT_REPO_ROOT=/root/repo
ARGS="--steps 3 --warmup 1 --no-real-rank --no-sweeps --no-other-modes --no-latency" VARIANTS="t16off:PEPSGPU_TILE16=0 t16on:PEPSGPU_TILE16=1" bash scripts/gpu_ab.sh
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  PEPSGPU_TILE16=$v rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d /tmp/t16_$v -o sq -- python3 /root/repo/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-route-check --no-full-rank --no-real-rank --no-sweeps --no-latency --no-energy-check --no-other-modes > /tmp/t16_$v.log 2>&1
  f=$(find /tmp/t16_$v -name "*counter_collection.csv" | head -1)
  python3 /root/repo/scripts/pmc_summary.py "$f" | grep "tgemm_chain_kernel" | head -12 > /root/repo/gpurun_out/r05/t16_sq_$v.txt
  echo "== PEPSGPU_TILE16=$v"; cat /root/repo/gpurun_out/r05/t16_sq_$v.txt | cut -c1-200
done
