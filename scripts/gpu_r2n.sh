cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2n
export TMPDIR=/tmp
( time python3 bench.py > gpurun_out/r2n/bench_default.json 2> gpurun_out/r2n/bench_default.err ) 2> gpurun_out/r2n/bench_default.time; echo "default rc=$?"; cat gpurun_out/r2n/bench_default.time | tail -3
python3 -c "
import json; d=json.load(open('gpurun_out/r2n/bench_default.json')); print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac']); print(d['cpu_baseline']['value'], d['parity_on_sample']['max_rel_err_amplitude'], d.get('energy_rel_err')); print(d['full_rank']['value'], d['full_rank']['kernel_ms'], d['full_rank']['parity_on_sample'], d['full_rank']['real_state_rank'])"
( time python3 bench.py --workload C5 --walkers 8192 > gpurun_out/r2n/bench_c5.json 2> gpurun_out/r2n/bench_c5.err ) 2> gpurun_out/r2n/bench_c5.time; echo "c5 rc=$?"; tail -3 gpurun_out/r2n/bench_c5.err
python3 -c "
import json; d=json.load(open('gpurun_out/r2n/bench_c5.json')); print('C5', d['value'], d['ms_per_step'], d['kernel_ms'], d['route_consistency'], d.get('parity_on_sample'), d['cpu_baseline']['value'])"
