#!/bin/bash
# floors of the f32 engine: parity against the f64 mode and rate on the real state, by PEPSGPU_F32_EPS
mkdir -p gpurun_out/r06
python3 scripts/floor_probe.py run f64 2048 2>&1 | tail -1
python3 scripts/floor_probe.py run f32_c8 8192 2>&1 | tail -1
PEPSGPU_F32_EPS=2.98e-8 python3 scripts/floor_probe.py run f32_c4 8192 2>&1 | tail -1
PEPSGPU_F32_EPS=1.49e-8 python3 scripts/floor_probe.py run f32_c2 8192 2>&1 | tail -1
PEPSGPU_F32_EPS=1.19e-7 python3 scripts/floor_probe.py run f32_c16 8192 2>&1 | tail -1
for e in 5.96e-8 2.98e-8 1.49e-8 1.19e-7; do PEPSGPU_DEBUG_SWEEPS=1 PEPSGPU_F32_EPS=$e python3 scripts/floor_probe.py run st_$e 1024 2>&1 | tail -1; cat gpurun_out/r06/floor_st_$e.json; echo; done
python3 scripts/floor_probe.py analyse
