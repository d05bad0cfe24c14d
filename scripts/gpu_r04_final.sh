#!/bin/bash
# round 4, last call: the driver's sequence on the final tree (whole GPU suite, smoke, bench lines) + the kernel traces of the three legs
cd /root/repo
bash scripts/gpu_r04_full.sh
KT_ONLY=1 bash scripts/gpu_r04_profiles.sh > gpurun_out/r04/profiles_final.log 2>&1
tail -30 gpurun_out/r04/profiles_final.log | cut -c1-180
