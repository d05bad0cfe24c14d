#!/bin/bash
# round 6, call 28: kernel-level evidence of the secondary modes at their FINAL routes:
# rocprofv3 kernel stats and the three PMC passes of the f64 real leg, the complex real leg and C5 (f64, f32)
cd /root/repo; mkdir -p gpurun_out/r06
export GRAFT_REPO_ROOT=/root/repo
cd /tmp && export TMPDIR=/tmp
prof_leg() {   # tag dtype walkers state
  tag=$1; dt=$2; nw=$3; st=$4
  rm -rf /tmp/p_$tag; mkdir -p /tmp/p_$tag
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$tag -o kt -- python3 /root/repo/scripts/f64_real_probe.py $dt $nw $st > /tmp/p_$tag/kt.log 2>&1
  tail -1 /tmp/p_$tag/kt.log | cut -c1-400
  f=$(find /tmp/p_$tag -name "kt_kernel_stats.csv" | head -1)
  [ -f "$f" ] && cp "$f" /root/repo/gpurun_out/r06/kernel_stats_${tag}.csv
  t=$(find /tmp/p_$tag -name "kt_kernel_trace.csv" | head -1)
  [ -f "$t" ] && python3 /root/repo/scripts/trace_summary.py "$t" > /root/repo/gpurun_out/r06/kernel_trace_by_grid_${tag}.txt
  grep "^{" /tmp/p_$tag/kt.log | tail -1 > /root/repo/gpurun_out/r06/probe_${tag}.json
  for grp in "FETCH_SIZE:FETCH_SIZE" "WRITE_SIZE:WRITE_SIZE" "SQ:SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
    g=${grp%%:*}; ctr=${grp#*:}
    rm -rf /tmp/p_$tag/$g
    timeout 900 rocprofv3 --pmc $ctr --output-format csv -d /tmp/p_$tag/$g -o pmc -- python3 /root/repo/scripts/f64_real_probe.py $dt $nw $st > /tmp/p_$tag/$g.log 2>&1
    c=$(find /tmp/p_$tag/$g -name "*counter_collection.csv" | head -1)
    [ -f "$c" ] && python3 /root/repo/scripts/pmc_summary.py "$c" > /root/repo/gpurun_out/r06/pmc_${g}_${tag}.txt
    rm -rf /tmp/p_$tag/$g
  done
  head -8 /root/repo/gpurun_out/r06/kernel_stats_${tag}.csv | cut -c1-90,200-330
}
prof_leg c4_f64_real_nw2048 f64 2048 real
prof_leg c4_c128_real_nw512 c128 512 real
prof_leg c5_f64_nw4096 f64 4096 c5
prof_leg c5_f32_nw4096 f32 4096 c5
