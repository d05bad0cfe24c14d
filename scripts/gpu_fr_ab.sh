# A/B of the full-rank leg (noise 1.0, 4096 walkers) under environment toggles: VARIANTS="name:ENV=1,ENV2=3 name2:..."
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/frab
for v in ${VARIANTS:-base:}; do
  name=${v%%:*}; envs=${v#*:}
  ( for e in ${envs//,/ }; do [ -n "$e" ] && export "$e"; done
    python3 bench.py --noise 1.0 --walkers ${NW:-4096} --steps 2 --warmup 1 --no-cpu-baseline --no-route-check --no-full-rank --no-energy-check > gpurun_out/frab/$name.json 2>gpurun_out/frab/$name.err )
  python3 - <<PY
import json
try:
    d=json.load(open('gpurun_out/frab/$name.json'))
    print('$name', round(d['value'],1), round(d['ms_per_step'],1), {k:round(v,1) for k,v in d['kernel_ms'].items()}, 'par', d.get('parity_on_sample',{}).get('max_rel_err'))
except Exception as e:
    print('$name', 'FAILED', e); print(open('gpurun_out/frab/$name.err').read()[-1500:])
PY
done
