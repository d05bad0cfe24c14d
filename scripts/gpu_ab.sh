# A/B of a bench leg under environment toggles.  VARIANTS="name:ENV=1,ENV2=3 ..."; ARGS = extra bench.py flags
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ab
for v in ${VARIANTS:-base:}; do
  name=${v%%:*}; envs=${v#*:}
  ( for e in ${envs//,/ }; do [ -n "$e" ] && export "$e"; done
    python3 bench.py ${ARGS} --no-cpu-baseline --no-route-check --no-full-rank --no-energy-check > gpurun_out/ab/$name.json 2>gpurun_out/ab/$name.err )
  python3 - <<PY
import json
try:
    d=json.load(open('gpurun_out/ab/$name.json'))
    print('$name', round(d['value'],1), round(d['ms_per_step'],1), {k:round(v,1) for k,v in d['kernel_ms'].items()}, 'par', d.get('parity_on_sample',{}))
except Exception as e:
    print('$name', 'FAILED', e); print(open('gpurun_out/ab/$name.err').read()[-1500:])
PY
done
