#!/bin/bash
# round 6, call 08: device-side sweeps of the other updaters / types (identical chains), C4 energy golden test, walker lifetime test,
# and the gate probe (f32 / f32 with f64 accumulation everywhere / f64 on 2048 real-state walkers)
cd /root/repo; mkdir -p gpurun_out/r06
timeout 1800 python -m pytest tests/test_gpu_host.py tests/test_gpu_walker.py tests/test_gpu_realrank.py tests/test_gpu_fermion.py -m gpu -q -x --tb=short -s -k "slice_sweep or releases_its_device or c4_energy_vs_oracle or k9" 2>&1 | grep -vE "^RCCL|^HIP|^ROCm|^Hostname|^Librccl" | tail -15
timeout 900 python scripts/gate_probe.py run f32 2048
PEPSGPU_ACC64=15 timeout 900 python scripts/gate_probe.py run f32acc 2048
timeout 1200 python scripts/gate_probe.py run f64 2048
python scripts/gate_probe.py analyse
