#!/bin/bash
# round 4, call 10: the three-stage chained BTen step under the tests; sweep probe with / without it, two LDS sizes
cd /root/repo
mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_parity.py tests/test_gpu_host.py tests/test_gpu_walker.py tests/test_gpu_measure.py tests/test_gpu_kernels.py tests/test_gpu_configs.py -x -q -m gpu > gpurun_out/r04/t10.log 2>&1
echo "pytest rc $?" >> gpurun_out/r04/t10.log
grep -E "passed|failed|rc |Error" gpurun_out/r04/t10.log | tail -5
for v in default lds4096 nochain3; do
  case $v in
    default) E="";;
    lds4096) E="PEPSGPU_CHAIN3_LDS=4096";;
    nochain3) E="PEPSGPU_NO_BTEN_CHAIN3=1";;
  esac
  env $E timeout 900 python scripts/sweep_probe.py --walkers 8192 --state synthetic --paths device > gpurun_out/r04/sweep_probe10_synth_$v.jsonl 2> gpurun_out/r04/sweep_probe10_synth_$v.err
  echo "$v: $(cut -c1-400 gpurun_out/r04/sweep_probe10_synth_$v.jsonl)"
done
timeout 1500 python scripts/sweep_probe.py --walkers 2048 --state real --sweeps 2 --paths device > gpurun_out/r04/sweep_probe10_real.jsonl 2> gpurun_out/r04/sweep_probe10_real.err
cut -c1-400 gpurun_out/r04/sweep_probe10_real.jsonl
