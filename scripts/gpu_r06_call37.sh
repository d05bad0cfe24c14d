#!/bin/bash
# Y = Tt V^T of precise sites: wave-per-tile f64 body (mode 1) against the LDS-tiled skinny f64 kernel + normalisation (mode 2)
mkdir -p gpurun_out/r06
python3 scripts/floor_probe.py run f64 2048 2>&1 | tail -1
python3 scripts/floor_probe.py run f32_ym1 8192 2>&1 | tail -1
PEPSGPU_LIB=$GRAFT_REPO_ROOT/peps_amd/lib/ab/ym2.so python3 scripts/floor_probe.py run f32_ym2 8192 2>&1 | tail -1
python3 scripts/floor_probe.py analyse | grep -E "ym"
python3 scripts/f64_real_probe.py f32 8192 real 2>&1 | grep "^{" | tail -1 | cut -c1-420
PEPSGPU_LIB=$GRAFT_REPO_ROOT/peps_amd/lib/ab/ym2.so python3 scripts/f64_real_probe.py f32 8192 real 2>&1 | grep "^{" | tail -1 | cut -c1-420
