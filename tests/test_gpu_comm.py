"""The exchange step on the device: HBM accumulators aliased without a copy, RCCL through the library's own
communicator (one rank here: the box has one GPU), bench.py --gpus 2 end to end on one device through the gloo hook."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from peps_amd import synthetic

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ctx(n=4):
    from peps_amd import capi
    L, D, chi = 4, 3, 6
    ctx = capi.Context(L, L, D, 2, chi, dtype=capi.F64, max_walkers=n)
    sitps = synthetic.make_sitps(L, D)
    ctx.state_upload(synthetic.sitps_to_flat(sitps, D))
    ctx.set_configs(synthetic.make_configs(L, n, "heisenberg"))
    return ctx


def test_grad_accumulators_alias_in_hbm_and_library_allreduce():
    import torch
    from peps_amd import capi, dist as pdist
    ctx = _ctx()
    ctx.grad_reset()
    so, seo, n = ctx.grad_device_ptr()
    assert so and seo and n == 4 * 4 * 2 * 3 ** 4
    ctx.sync()
    t = pdist.device_tensor(so, n)
    assert t.is_cuda and t.dtype == torch.float64 and t.data_ptr() == so        # zero-copy alias
    t.fill_(1.0)                        # written through torch, read back through the library
    torch.cuda.synchronize()
    got, _ = ctx.grad_read()            # state-upload layout; a slot holds the site's true (un-padded) elements only
    true_elems = sum(2 * int(np.prod(synthetic.bond_dims(4, 3, r, c))) for r in range(4) for c in range(4))
    assert got.sum() == true_elems and set(np.unique(got)) <= {0.0, 1.0}
    # a context without a communicator is one rank: reductions are the identity
    assert ctx.comm_size() == 1
    v = np.arange(5, dtype=np.float64)
    assert np.array_equal(ctx.allreduce(v.copy()), v)
    # a real RCCL communicator of one rank: same results through ncclAllReduce on the context's stream
    ctx.comm_init(1, 0, capi.comm_unique_id())
    assert ctx.comm_size() == 1 and ctx.comm_rank() == 0
    assert np.array_equal(ctx.allreduce(v.copy()), v)
    assert np.array_equal(ctx.allreduce(v.copy(), op="max"), v)
    ctx.grad_allreduce()
    got2, _ = ctx.grad_read()
    assert np.array_equal(got, got2)
    ctx.allreduce_device(seo, n)
    ctx.comm_destroy()
    ctx.close()


def test_state_broadcast_one_rank_communicator():
    """pepsgpu_bcast_state through a real RCCL communicator of one rank (ncclBroadcast on the context's stream): the state
    stays what was uploaded, a context that only ever received a broadcast counts as having a state; without a communicator
    on a multi-rank setup the call is refused (here: one rank without communicator = the identity)."""
    from peps_amd import capi
    ctx = _ctx()
    a0 = ctx.evaluate_amplitude()
    ctx.bcast_state(0)                                   # no communicator, one rank: identity
    assert np.array_equal(ctx.evaluate_amplitude(), a0)
    ctx.comm_init(1, 0, capi.comm_unique_id())
    ctx.bcast_state(0)                                   # ncclBroadcast, root = the only rank
    assert np.array_equal(ctx.evaluate_amplitude(), a0)
    with pytest.raises(ValueError):
        ctx.bcast_state(1)                               # root outside the communicator
    ctx.comm_destroy()
    ctx.close()


TWO_GPU_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np
import torch, torch.distributed as dist
from peps_amd import capi, dist as pdist, synthetic
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
pdist.init("nccl")
L, D, chi = 4, 3, 6
ctx = capi.Context(L, L, D, 2, chi, dtype=capi.F64, device=int(os.environ["LOCAL_RANK"]), max_walkers=4)
pdist.comm_init(ctx)
assert ctx.comm_size() == world and ctx.comm_rank() == rank
new = synthetic.sitps_to_flat(synthetic.make_sitps(L, D, noise=0.3), D)
pdist.broadcast_state(ctx, new if rank == 0 else None, src=0)      # device path: upload on rank 0 + ncclBroadcast
ctx.set_configs(synthetic.make_configs(L, 4, "heisenberg"))
a = ctx.evaluate_amplitude()
ref = capi.Context(L, L, D, 2, chi, dtype=capi.F64, device=int(os.environ["LOCAL_RANK"]), max_walkers=4)
ref.state_upload(new); ref.set_configs(synthetic.make_configs(L, 4, "heisenberg"))
assert np.array_equal(a, ref.evaluate_amplitude())
ctx.grad_reset()
so, seo, n = ctx.grad_device_ptr(); ctx.sync()
t = pdist.device_tensor(so, n); t.fill_(1.0); torch.cuda.synchronize()
ctx.grad_allreduce()                                                # library communicator, in place in HBM
got, _ = ctx.grad_read()
assert set(np.unique(got)) <= {0.0, float(world)}, np.unique(got)
v = ctx.allreduce(np.array([1.0, float(rank)]))
assert v[0] == world and v[1] == sum(range(world))
if rank == 0:
    print("OK")
dist.barrier(); dist.destroy_process_group()
'''


def test_two_gpu_comm_init_bcast_and_grad_allreduce():
    """N = 2 on hardware (skipped on a one-GPU box): pepsgpu_comm_init with nranks = 2, the state broadcast and the in-place
    gradient all-reduce across two GPUs; the reduced accumulators equal `world` where every rank wrote 1."""
    import tempfile
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    with tempfile.TemporaryDirectory() as td:
        wp = os.path.join(td, "worker.py")
        open(wp, "w").write(TWO_GPU_WORKER)
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
        for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
            env.pop(k, None)
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                            "--master-addr", "127.0.0.1", "--master-port", "29533", wp, ROOT],
                           capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "OK" in r.stdout


def test_max_walkers_beyond_grid_limit_is_refused_at_creation():
    from peps_amd import capi
    with pytest.raises(ValueError):
        capi.Context(4, 4, 2, 2, 4, max_walkers=70000)


def test_failed_call_returns_its_temporaries():
    """an operation that throws half way (here: a trace without its environments) must not leak arena blocks"""
    ctx = _ctx()
    ctx.evaluate_amplitude()
    before = ctx.stats()["device_bytes"]
    for _ in range(20):
        with pytest.raises(RuntimeError):
            ctx.replace_nn_trace(2, 1, 0, np.zeros((4, 1, 2), dtype=np.int32))     # row 2 has no UP/DOWN pair yet
    assert ctx.stats()["device_bytes"] <= before + (1 << 20)
    ctx.close()


def test_bench_gpus_2_on_one_device_gloo():
    """`python bench.py --gpus 2` starts two ranks itself; with the gloo hook both share this box's GPU.  Real device work
    (small batch), one JSON line, n_gpus == 2, value = both ranks' walkers / max-over-ranks time."""
    env = dict(os.environ, PEPS_BENCH_BACKEND="gloo", PEPS_BENCH_NDEV="1", MASTER_ADDR="127.0.0.1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--workload", "C2",
                        "--walkers", "256", "--no-cpu-baseline", "--no-route-check", "--no-full-rank"],
                       capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_reported_by_backend"] == 2
    assert out["value"] > 0 and abs(out["value"] - 2 * 256 / (out["ms_per_step"] * 1e-3)) < 1e-6 * out["value"]
