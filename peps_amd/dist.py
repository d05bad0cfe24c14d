"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on
ROCm, "gloo" in CPU tests).  The hot path shards walkers / configurations with NO data-path
collective; the only exchange is one all-reduce(sum) of the packed energy / gradient accumulators
per evaluation, which replaces the reference's MPI calls:
  MPIMeanTensor per (site, component)       monte_carlo_tools/statistics_tensor.h:37-79
  MPI_Gather/Gatherv of energy bins         monte_carlo_tools/statistics.h:185-207
  MPI_Send/Recv of S_O, S_EO + MPI_Reduce   exact_summation_energy_evaluator.h:252-280
  MPI_Allreduce(MAX) of acceptance rates    mc_energy_grad_evaluator.h:405-410
"""
import os

import numpy as np


def init(backend=None):
    import torch
    import torch.distributed as dist
    if dist.is_initialized():
        return dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group(backend, rank=int(os.environ.get("RANK", "0")), world_size=int(os.environ.get("WORLD_SIZE", "1")))
    return dist


def shard_indices(n, rank, size):
    """Round-robin partition i = rank, rank + size, ... (exact_summation_energy_evaluator.h:201;
    walkers: monte_carlo_engine.h:97-98)."""
    return range(rank, n, size)


def walker_seed(base, rank, walker, walkers_per_rank):
    """Per-walker stream, distinct across ranks (pattern of tests/test_algorithm/test_boson_mc_sr_golden.cpp:85-87)."""
    return (int(base) ^ (0x9E3779B97F4A7C15 * (rank * walkers_per_rank + walker + 1))) & 0xFFFFFFFF


class _DevicePtr:
    """__cuda_array_interface__ view of a raw HBM pointer of the library (no copy, no ownership)."""

    def __init__(self, ptr, n, typestr="<f8"):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False), "version": 2,
                                         "strides": None}


def device_tensor(ptr, n, typestr="<f8"):
    """torch tensor aliasing n elements at device pointer ptr (the context's GPU must be torch's current device)."""
    import torch
    return torch.as_tensor(_DevicePtr(ptr, n, typestr), device="cuda")


def comm_init(ctx):
    """Give the context its own RCCL communicator (pepsgpu_comm_init) over the ranks of the initialised process group:
    rank 0 draws the unique id, torch.distributed carries the 128 bytes (the reference host would MPI_Bcast them)."""
    import torch.distributed as dist
    from . import capi
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    box = [capi.comm_unique_id() if dist.get_rank() == 0 else None]
    dist.broadcast_object_list(box, src=0)
    ctx.comm_init(dist.get_world_size(), dist.get_rank(), box[0])


def reduced_grad(ctx, use_library_comm=False):
    """S_O, S_EO of pepsgpu_grad_accumulate summed over the ranks (replaces MPIMeanTensor, statistics_tensor.h:37-79, and
    the Send/Recv + reduce of exact_summation_energy_evaluator.h:252-280), returned in the state-upload layout.
    backend "nccl": the all-reduce runs on the HBM accumulators themselves -- through torch.distributed on a zero-copy
    alias of the device pointers, or through the library's own communicator (comm_init) -- and only the result is read
    back.  Any other backend (gloo in the CPU / one-GPU tests): read back, reduce on the host."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return ctx.grad_read()
    if dist.get_backend() == "nccl":
        if use_library_comm:
            ctx.grad_allreduce()
        else:
            so, seo, n = ctx.grad_device_ptr()
            ctx.sync()                       # the accumulation kernels run on the context's stream
            for ptr in (so, seo):
                dist.all_reduce(device_tensor(ptr, n), op=dist.ReduceOp.SUM)
            torch.cuda.synchronize()
        return ctx.grad_read()
    so, seo = ctx.grad_read()
    return allreduce_sum(so.ravel()).reshape(so.shape), allreduce_sum(seo.ravel()).reshape(seo.shape)


def _reduce(vec, op):
    import torch
    import torch.distributed as dist
    v = np.ascontiguousarray(vec, dtype=np.float64)
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return v.copy()
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.from_numpy(v.copy()).to(dev)
    dist.all_reduce(t, op=op)
    return t.cpu().numpy()


def allreduce_sum(vec):
    import torch.distributed as dist
    return _reduce(vec, dist.ReduceOp.SUM)


def allreduce_max(vec):
    import torch.distributed as dist
    return _reduce(vec, dist.ReduceOp.MAX)


def allreduce_min(vec):
    import torch.distributed as dist
    return _reduce(vec, dist.ReduceOp.MIN)


def exchange_valid_configuration(n_invalid, have_valid, cfg):
    """The collective of MonteCarloEngine::EnsureConfigurationValidity (monte_carlo_engine.h:344-387: MPI_Allgather of the validity
    flags, MPI_BCast of the configuration of the FIRST valid rank) for ranks that hold many walkers each: every rank passes the number
    of its invalid walkers, whether it holds a valid one, and the configuration (flat int32, one lattice) of its first valid walker.
    Returns (total number of invalid walkers over the ranks, or -1 when no rank holds a valid one; the configuration of the lowest
    rank that holds one).  Without an initialised process group: the local answer."""
    cfg = np.ascontiguousarray(cfg, dtype=np.int32).ravel()
    try:
        import torch
        import torch.distributed as dist
        multi = dist.is_initialized() and dist.get_world_size() > 1
    except ImportError:
        multi = False
    if not multi:
        return (int(n_invalid) if (have_valid or n_invalid == 0) else -1), cfg
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    mine = torch.from_numpy(np.concatenate([[int(n_invalid), int(bool(have_valid))], cfg]).astype(np.int64)).to(dev)
    allv = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(allv, mine)
    allv = [t.cpu().numpy() for t in allv]
    total = int(sum(int(a[0]) for a in allv))
    src = next((r for r, a in enumerate(allv) if a[1]), None)
    if src is None:
        return (0 if total == 0 else -1), cfg
    return total, allv[src][2:].astype(np.int32)


def broadcast_state(ctx, flat, src=0):
    """The parameter broadcast after an optimizer update (replaces the per-tensor MPI_Bcast of
    split_index_tps_impl.h:778-880): rank `src` holds the new state `flat` (upload layout); every rank's context ends up with it.
    backend "nccl" with a library communicator (comm_init): rank src uploads, ONE ncclBroadcast moves the HBM buffer to the other
    GPUs over xGMI (pepsgpu_bcast_state) -- no host upload there.  Any other backend (gloo: CPU tests, one-GPU boxes): the host
    array is broadcast and every rank uploads it.  `ctx` may be anything with state_upload (and bcast_state / comm_size);
    returns the array the ranks other than src received (None on the device path)."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        ctx.state_upload(flat)
        return None
    # the ranks agree on the path first (all-reduce MIN of "my context has a library communicator of the full size"): a rank whose
    # comm_init failed must not wait in a host broadcast while the others sit in ncclBroadcast
    have_lib = 1.0 if (dist.get_backend() == "nccl" and getattr(ctx, "comm_size", lambda: 1)() == dist.get_world_size()) else 0.0
    if allreduce_min(np.array([have_lib]))[0] > 0.5:
        if dist.get_rank() == src:
            ctx.state_upload(flat)
        ctx.bcast_state(src)
        return None
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    # dtype and shape travel with the object list: a complex (PEPSGPU_C128) state keeps its imaginary part
    meta = [None]
    if dist.get_rank() == src:
        flat = np.ascontiguousarray(flat)
        if not np.iscomplexobj(flat):
            flat = flat.astype(np.float64, copy=False)
        else:
            flat = flat.astype(np.complex128, copy=False)
        meta[0] = (tuple(flat.shape), flat.dtype.str)
    dist.broadcast_object_list(meta, src=src)
    shape, dstr = meta[0]
    cplx = np.dtype(dstr).kind == "c"
    if dist.get_rank() == src:
        host = np.ascontiguousarray(flat).view(np.float64) if cplx else flat      # interleaved (re, im) pairs
        t = torch.from_numpy(host.copy()).to(dev)
    else:
        n = int(np.prod(shape)) * (2 if cplx else 1)
        t = torch.empty(n if cplx else shape, dtype=torch.float64, device=dev)
    if cplx:
        t = t.reshape(-1)
    dist.broadcast(t, src=src)
    got = t.cpu().numpy()
    if cplx:
        got = got.view(np.complex128).reshape(shape)
    ctx.state_upload(got)
    return got
