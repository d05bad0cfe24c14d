cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/n1
for nw in 1 16 256 4096; do
  python3 bench.py --walkers $nw --steps 5 --warmup 2 --no-cpu-baseline --no-route-check --no-full-rank > gpurun_out/n1/b_$nw.json 2>/dev/null
  python3 -c "
import json; d=json.load(open('gpurun_out/n1/b_$nw.json')); print($nw, round(d['value'],2), round(d['ms_per_step'],3))"
done
