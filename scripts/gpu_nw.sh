cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/nw
for nw in 16384 32768 49152 65535; do
  python3 bench.py --walkers $nw --steps 2 --warmup 1 --no-cpu-baseline --no-route-check --no-full-rank --no-energy-check > gpurun_out/nw/b_$nw.json 2>gpurun_out/nw/b_$nw.err
  python3 -c "
import json
try:
    d=json.load(open('gpurun_out/nw/b_$nw.json')); print($nw, round(d['value'],1), round(d['ms_per_step'],2))
except Exception as e:
    print($nw, 'FAILED'); print(open('gpurun_out/nw/b_$nw.err').read()[-800:])
"
done
