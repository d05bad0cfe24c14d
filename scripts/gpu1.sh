cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
PEPS_BENCH_BACKEND=gloo PEPS_BENCH_NDEV=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 1 --warmup 1 --walkers 128 2>&1 | tail -1 | cut -c1-400
