#!/usr/bin/env python3
"""Headline benchmark: configuration-amplitudes/sec of the boundary-MPS hot path on MI355X.

A "step" = one pass of the hot path over one batch of synthetic input: `walkers` FRESH
configurations per GPU, each through TPSWaveFunctionComponent::EvaluateAmplitude
(wave_function_component.h:187-212: L-1 row absorptions with SVD(chi,chi,0), L-2 BTen steps, one
trace).  Workload at N=1 = BASELINE.json's metric configuration: 12x12 spin-1/2 Heisenberg PEPS,
D=8, chi=32, synthetic state of SURVEY 8(d).  N>1: one rank per GPU, walkers
sharded, no data-path collective (weak scaling).

    python bench.py --gpus N --steps K --warmup W
        N > 1 without WORLD_SIZE in the environment: this process only starts
        `python -m torch.distributed.run --nproc-per-node N ... bench.py ...` as a CHILD (before anything here
        touches the GPU) and exits with its code.
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

The JSON line carries, besides the headline leg (the SURVEY 8(d) synthetic state, numerically of low rank),
a second leg `full_rank` on i.i.d. random site tensors (--noise 1.0) of the same shapes: the regime in which the
dense MFMA GEMMs and the chi-truncation of full 256 x 256 blocks dominate.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"f32": 157.3, "f64": 78.6}     # /opt/skills/guides/MI355X_MICROARCH.md, dense MFMA peaks
PEAK_HBM_GBPS = 8000.0                        # same guide: HBM3E, ~8 TB/s
REAL_STATE = os.path.join(ROOT, "tests", "golden", "ref_fixtures", "tps_square_heisenberg4x4D8Double")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--walkers", type=int, default=None,
                    help="walkers (fresh configurations) per GPU per step; default 32768 (81 GB at the low-rank "
                         "headline workload), 4096 for f64 or for states of higher rank (--noise > 0.15: up to 11.6 MB / walker)")
    ap.add_argument("--workload", default="C4", choices=["C2", "C3", "C4", "C5"],
                    help="BASELINE config; C5 = 8x8 spinless-fermion t-V, fZ2-graded PEPS, D=6 chi=24 through the sign-decorated path")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--cpu-seconds", type=float, default=25.0, help="time budget of the CPU baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-route-check", action="store_true", help="skip the row- vs column-contraction diagnostic (profiling runs)")
    ap.add_argument("--no-energy-check", action="store_true", help="skip the E_loc parity sample (device vs float64 oracle)")
    ap.add_argument("--noise", type=float, default=0.1,
                    help="relative noise of the synthetic site tensors (SURVEY 8d: 0.1; 1.0 = full-rank stress case)")
    ap.add_argument("--state", default="synthetic", choices=["synthetic", "real"],
                    help="state of the MAIN leg: the SURVEY 8(d) synthetic state (default) or the tiled optimised state of the "
                         "reference (profiling runs of the real_rank leg)")
    ap.add_argument("--no-full-rank", action="store_true", help="skip the second leg on a state of full rank")
    ap.add_argument("--full-rank-walkers", type=int, default=4096)
    ap.add_argument("--full-rank-steps", type=int, default=2)
    ap.add_argument("--no-real-rank", action="store_true", help="skip the third leg on the tiled optimised state of the reference")
    ap.add_argument("--real-rank-walkers", type=int, default=2048)
    ap.add_argument("--real-rank-steps", type=int, default=2)
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / rendezvous / reduction plumbing only, NO device work: value is null (CPU tests of --gpus N)")
    args = ap.parse_args()
    if args.walkers is None:
        args.walkers = 32768 if (args.dtype == "f32" and args.noise <= 0.15 and args.state == "synthetic") else (2048 if args.state == "real" else 4096)
    return args


def spawn_ranks(args):
    """--gpus N from a plain `python bench.py`: start the N ranks as a child torch.distributed.run.  Nothing in this
    process has touched the GPU yet (and never will: it only waits), so no GPU-initialised process is replaced."""
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus,
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def cpu_baseline(flat, cfgs, chi, budget_s):
    """The reference's CPU path restated in plain C on LAPACK (oracle/cbmps.c: bmps_impl.h:756-862, :225-263 op for op,
    float64), timed on this host in the reference's execution model (independent walkers, one per PROCESS as one per MPI
    rank, BLAS threads = 1): (i) one process, one walker; (ii) one walker per core on all cores.  The pool lives in a child
    interpreter (oracle/cbmps.py) that never touches the GPU.  Bounded sample; returns a dict."""
    from oracle import cbmps
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    a1, s1, _ = cbmps.amplitudes_multiprocess(flat, cfgs[:1], chi, 1)
    # all cores: as many rounds of `cores` walkers as the budget allows (at least one; a loaded socket runs each walker
    # slower than the single one), capped by the sample at hand
    rounds = max(1, int((budget_s - s1) / max(s1 * 2.0, 1e-3)))
    n_all = min(len(cfgs), cores * rounds)
    aN, sN, nproc = cbmps.amplitudes_multiprocess(flat, cfgs[:n_all], chi, min(cores, n_all))
    return {"single": {"value": 1.0 / s1, "n": 1, "threads": 1, "seconds": s1},
            "all_cores": {"value": n_all / sN, "n": int(n_all), "threads": int(nproc), "seconds": sN},
            "cores": int(cores), "amps": aN}


def pmc_traffic_bytes(category, args, nw, launches_per_step):
    """HBM bytes per launch of the dominant kernel category from the committed rocprofv3 --pmc passes (separate
    FETCH_SIZE / WRITE_SIZE passes, KiB, FETCH doubled as MI355X_MICROARCH.md 'HBM' prescribes for gfx950).  PMC counters
    cannot be read from inside this process, so the figure is quoted only when this run is the run the passes were
    collected on: same workload / dtype / walkers / noise AND the same number of launches per step in this category
    (profiles/r02_pmc_meta.json); any difference (a changed kernel mix) returns null instead of a stale number."""
    meta_path = os.path.join(ROOT, "profiles", "r02_pmc_meta.json")
    if not os.path.exists(meta_path):
        return None
    try:
        meta = json.load(open(meta_path))
    except Exception:
        return None
    if (meta.get("workload"), meta.get("dtype"), meta.get("walkers"), meta.get("noise")) != (args.workload, args.dtype, nw, args.noise):
        return None
    ent = meta.get("categories", {}).get(category)
    if not ent or int(ent.get("launches_per_step", -1)) != int(launches_per_step):
        return None
    return float(ent["hbm_bytes_per_launch"])


class Leg:
    """One timed pass: `steps` batches of `nw` fresh configurations through EvaluateAmplitude on this rank's GPU."""

    def __init__(self, capi, synthetic, L, D, chi, dt, device, nw, noise, fermionic=False, real=False):
        self.capi, self.synthetic = capi, synthetic
        self.L, self.D, self.chi, self.dt, self.device, self.nw, self.noise = L, D, chi, dt, device, nw, noise
        self.fermion = None
        self.real = real
        if fermionic:
            # C5: synthetic parity-even state (the reference ships no fermionic state beyond 2x2); a configuration enters the
            # device as the row-major extended states of the decorated components (peps_amd/fermion.py), the graded amplitude
            # is sigma(N_f) times the contraction -- the sign is applied on the host inside the timed region
            from peps_amd import fermion
            self.fermion = fermion
            self.fstate = fermion.random_even_state(L, L, D, seed=11)
            self.pdim = fermion.NVAR * self.fstate.d
            self.ctx = capi.Context(L, L, D, self.pdim, chi, dtype=dt, device=device, max_walkers=nw)
            self.flat = self.fstate.extended_flat(D)
            self.ctx.state_upload(self.flat)
            self.sitps = None
            return
        self.pdim = 2
        self.ctx = capi.Context(L, L, D, 2, chi, dtype=dt, device=device, max_walkers=nw)
        if real:
            # the reference's own optimised D = 8 state (4x4 fixture), tiled by position class to L x L: site tensors with
            # the singular spectra of a real PEPS (synthetic.tile_flat_state); configurations near the Neel state
            from peps_amd import hostapi
            assert D == 8, "the reference's optimised fixture has D = 8"
            sitps = synthetic.flat_to_sitps(synthetic.tile_flat_state(hostapi.load_sitps(REAL_STATE, 8), L))
        else:
            # synthetic state of SURVEY 8(d); psi(S_ref) normalisation evaluated with the device path itself
            sitps = synthetic.make_sitps(L, D, noise=noise)
        self.ctx.state_upload(synthetic.sitps_to_flat(sitps, D, np.float64))
        self.ctx.set_configs(synthetic.checkerboard(L)[None])
        psi_ref = float(self.ctx.evaluate_amplitude()[0])
        self.sitps = synthetic.rescale_sitps(sitps, psi_ref)
        self.flat = synthetic.sitps_to_flat(self.sitps, D, np.float64)
        self.ctx.state_upload(self.flat)

    def run(self, steps, warmup, rank, world, barrier):
        ctx, nw = self.ctx, self.nw
        total = warmup + steps
        gen = self.synthetic.make_configs_near_neel if self.real else (lambda L, n, seed0: self.synthetic.make_configs(L, n, "heisenberg", seed0=seed0))
        self.batches = [gen(self.L, nw, seed0=7 + (s * world + rank) * nw) for s in range(total)]
        sig = None
        if self.fermion is not None:      # half filling: state 0 = occupied; device labels + graded sign per configuration
            self.phys = self.batches
            sig = [self.fstate.sigma(b) for b in self.phys]
            self.batches = [self.fstate.ext_config(b, self.fermion.ROW) for b in self.phys]
        amps_first = None
        for s in range(warmup):
            ctx.set_configs(self.batches[s])
            a = ctx.evaluate_amplitude() * (sig[s] if sig else 1)
            if amps_first is None:
                amps_first = a
        ctx.profile_enable(True)
        ctx.profile_read()
        barrier(ctx)
        t0 = time.perf_counter()
        nz = 0
        for s in range(warmup, total):
            ctx.set_configs(self.batches[s])
            a = ctx.evaluate_amplitude() * (sig[s] if sig else 1)
            nz += int(np.count_nonzero(ctx.walker_flags()))
            if amps_first is None:
                amps_first = a
        barrier(ctx)
        elapsed = time.perf_counter() - t0
        prof = ctx.profile_read()
        ctx.profile_enable(False)
        self.amps_first = amps_first
        return elapsed, prof, nz

    def rank_diagnostics(self):
        """numerical rank of the carry on this state (the Jacobi / Gram / Cholesky cost follows it); outside the timed region"""
        os.environ["PEPSGPU_DEBUG_SWEEPS"] = "1"
        try:
            d = self.capi.Context(self.L, self.L, self.D, self.pdim, self.chi, dtype=self.dt, device=self.device, max_walkers=16)
            d.state_upload(self.flat)
            d.set_configs(self.batches[0][:16])
            d.evaluate_amplitude()
            st = d.stats()
            d.close()
        finally:
            os.environ.pop("PEPSGPU_DEBUG_SWEEPS", None)
        return {"carry_live_fraction": st["carry_live_fraction"], "carry_live_max": st["carry_live_max"], "noise": self.noise,
                "jacobi_sweeps_max": st["jacobi_sweeps_max"]}

    def close(self):
        self.ctx.close()


def roofline_of(prof, dtype, traffic=None):
    """Roofline object of the dominant kernel category of a leg (see the note in the JSON)."""
    peak = PEAK_TFLOPS[dtype]
    dom = max(prof, key=lambda k: prof[k]["ms"])
    dsec = prof[dom]["ms"] * 1e-3
    # flops the kernels of the dominant category contracted (2*I*J*K over the walkers' live extents, counted on the
    # device); categories without a tensor GEMM fall back to the reference-algorithm count
    counted = prof[dom]["exec_flops"] if prof[dom]["exec_flops"] > 0 else prof[dom]["alg_flops"]
    achieved = counted / dsec / 1e12 if dsec > 0 else 0.0
    ref_equiv = prof[dom]["alg_flops"] / dsec / 1e12 if dsec > 0 else 0.0
    # compulsory traffic of the same launches: bytes of the live operand and result elements, counted on the device next to
    # the flops.  Below the machine balance (peak flops / peak HBM bandwidth) the kernel is bound by HBM, not by MFMA issue.
    dbytes = prof[dom].get("bytes", 0.0)
    intensity = counted / dbytes if dbytes > 0 else float("inf")
    # the Gram kernels (streaming, and the ones fused with the Cholesky: categories cholesky / trunc_gram) run on
    # v_mfma_f64_16x16x4_f64; a category whose launches counted no MFMA flops on the device has no MFMA roofline
    kpeak = PEAK_TFLOPS["f64"] if dom in ("gram_f64", "cholesky", "trunc_gram", "trunc_apply") else peak
    balance = kpeak * 1e12 / (PEAK_HBM_GBPS * 1e9)
    hbm_bound = intensity < balance
    gbps = dbytes / dsec / 1e9 if dsec > 0 else 0.0
    return dom, {
        "bound": "hbm" if hbm_bound else "mfma",
        "kernel": dom,
        "achieved": gbps if hbm_bound else achieved,
        "peak": PEAK_HBM_GBPS if hbm_bound else kpeak,
        "unit": "GB/s" if hbm_bound else "TFLOP/s",
        "frac": gbps / PEAK_HBM_GBPS if hbm_bound else achieved / kpeak,
        "traffic": traffic,
        "algorithmic_bytes_per_launch": dbytes / max(prof[dom]["launches"], 1),
        "arithmetic_intensity_flop_per_byte": intensity if dbytes > 0 else None,
        "machine_balance_flop_per_byte": balance,
        "mfma_tflops": achieved,
        "mfma_frac": achieved / kpeak,
        "avg_launch_ms": prof[dom]["ms"] / max(prof[dom]["launches"], 1),
        "launches": prof[dom]["launches"],
        "reference_equivalent_tflops": ref_equiv,
    }


def mfma_summary(prof, dtype, step_seconds_total):
    """Executed MFMA flops of every category that runs on the matrix cores (device-counted 2*I*J*K over live extents) against
    their own time and against the whole timed region."""
    cats = {}
    tot_fl = tot_ms = 0.0
    for k in ("contract", "gram_f64", "cholesky", "env", "trunc_gram", "trunc_apply"):
        if k in prof and prof[k]["launches"] and prof[k]["exec_flops"] > 0:
            pk = PEAK_TFLOPS["f64"] if k in ("gram_f64", "cholesky", "trunc_gram", "trunc_apply") else PEAK_TFLOPS[dtype]
            tf = prof[k]["exec_flops"] / (prof[k]["ms"] * 1e-3) / 1e12 if prof[k]["ms"] > 0 else 0.0
            cats[k] = {"ms": round(prof[k]["ms"], 3), "tflops": tf, "peak": pk, "frac": tf / pk}
            tot_fl += prof[k]["exec_flops"]
            tot_ms += prof[k]["ms"]
    return {"categories": cats,
            "mfma_tflops_in_mfma_kernels": tot_fl / (tot_ms * 1e-3) / 1e12 if tot_ms > 0 else 0.0,
            "mfma_tflops_over_whole_step": tot_fl / step_seconds_total / 1e12 if step_seconds_total > 0 else 0.0,
            "mfma_kernel_time_share": tot_ms * 1e-3 / step_seconds_total if step_seconds_total > 0 else 0.0}


def energy_parity(leg, n, budget_s):
    """E_loc of `n` configurations of the timed batch on the device (C++ host layer, the reference's XXZ solver schedule) against
    the float64 oracle restatement of the same solver (oracle/vmc.py: square_nnn_energy_solver.h, square_spin_onehalf_xxz_obc.h)."""
    from peps_amd import hostapi
    from oracle import vmc
    from oracle.bmps import BMPSTruncateParams
    cfgs = leg.batches[0][:n]
    hostapi.set_device(leg.device)
    res = hostapi.energy_and_holes(leg.flat, cfgs, leg.chi, model="xxz", params=(1.0, 1.0, 0.0), holes=False,
                                   dtype=0 if leg.dt == leg.capi.F32 else 1)
    e_dev = np.asarray(res[1] if isinstance(res, (tuple, list)) else res["energy"])
    tp = BMPSTruncateParams.SVD(leg.chi, leg.chi, 0.0)
    model = vmc.SquareSpinOneHalfXXZModelOBC(1.0, 1.0, 0.0)
    e_ref = []
    t0 = time.perf_counter()
    for c in cfgs:
        comp = vmc.TPSWaveFunctionComponent(leg.sitps, c, tp)
        out = model.CalEnergyAndHoles(leg.sitps, comp, calchols=False)
        e_ref.append(float(out[0] if isinstance(out, (tuple, list)) else out))
        if time.perf_counter() - t0 > budget_s:
            break
    e_ref = np.array(e_ref)
    k = len(e_ref)
    return {"max_rel_err_energy": float(np.max(np.abs(e_dev[:k] - e_ref) / np.abs(e_ref))), "n": int(k),
            "tolerance": 1e-6 if leg.dt == leg.capi.F64 else 1e-5,
            "e_per_site_device": [float(x) / (leg.L * leg.L) for x in e_dev[:k]],
            "e_per_site_oracle": [float(x) / (leg.L * leg.L) for x in e_ref],
            "oracle_seconds": time.perf_counter() - t0}


def real_state_rank(capi, device, dt):
    """carry_live_fraction of the reference's own optimised D=8 state (4x4 Heisenberg, tests/slow_tests fixture) under the
    C4 truncation chi=32: which synthetic leg resembles a real PEPS."""
    from peps_amd import hostapi, synthetic
    if not os.path.isdir(REAL_STATE):
        return None
    flat = hostapi.load_sitps(REAL_STATE, 8)
    os.environ["PEPSGPU_DEBUG_SWEEPS"] = "1"
    try:
        c = capi.Context(4, 4, 8, 2, 32, dtype=dt, device=device, max_walkers=16)
        c.state_upload(flat)
        c.set_configs(synthetic.make_configs(4, 16, "heisenberg"))
        c.evaluate_amplitude()
        st = c.stats()
        c.close()
    finally:
        os.environ.pop("PEPSGPU_DEBUG_SWEEPS", None)
    return {"state": "tests/golden/ref_fixtures/tps_square_heisenberg4x4D8Double (reference fixture, 4x4, D=8), chi=32",
            "carry_live_fraction": st["carry_live_fraction"], "jacobi_sweeps_max": st["jacobi_sweeps_max"]}


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    torch = None
    backend = None
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # test hooks (CPU / one-GPU boxes): PEPS_BENCH_BACKEND=gloo PEPS_BENCH_NDEV=1 runs N ranks on one device
        backend = os.environ.get("PEPS_BENCH_BACKEND", "nccl")
        if not args.dry_run:
            ndev = int(os.environ.get("PEPS_BENCH_NDEV", "0")) or torch.cuda.device_count()
            local_rank = local_rank % max(ndev, 1)
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world)

    def barrier(ctx=None):
        if dist is not None:
            dist.barrier()
            if backend == "nccl":
                torch.cuda.synchronize()
        if ctx is not None:
            ctx.sync()

    def max_over_ranks(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    from peps_amd import synthetic
    from peps_amd.flops import reference_flops
    fermionic = args.workload == "C5"
    L, D, chi, model = (8, 6, 24, "spinless_tV") if fermionic else synthetic.CONFIGS[args.workload]
    nw = args.walkers
    fl = reference_flops(L, D, chi)
    workload = ("%s: %dx%d %s PEPS, D=%d, chi=%d, fresh EvaluateAmplitude per configuration (SVD(chi,chi,0) truncation)"
                % (args.workload, L, L, "spinless-fermion t-V (fZ2-graded, sign-decorated components)" if fermionic
                   else "spin-1/2 Heisenberg", D, chi))

    if args.dry_run:
        # plumbing only: rendezvous, barrier, max-over-ranks, one line from rank 0 -- no device, no measurement
        barrier()
        t0 = time.perf_counter()
        barrier()
        elapsed = max_over_ranks(time.perf_counter() - t0)
        if rank == 0:
            print(json.dumps({"metric": "configuration-amplitudes/sec", "value": None, "unit": "amplitudes/s", "dry_run": True,
                              "n_gpus": world, "ranks_reported_by_backend": dist.get_world_size() if dist else 1,
                              "backend": backend, "steps": args.steps, "warmup": args.warmup, "ms_per_step": None,
                              "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
                              "data": "synthetic", "config": {"workload": workload, "walkers_per_gpu": nw},
                              "rendezvous_seconds": elapsed}))
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    from peps_amd import capi
    dt = capi.F32 if args.dtype == "f32" else capi.F64
    leg = Leg(capi, synthetic, L, D, chi, dt, local_rank, nw, args.noise, fermionic, real=args.state == "real")
    elapsed, prof, nz = leg.run(args.steps, args.warmup, rank, world, barrier)
    elapsed = max_over_ranks(elapsed)

    out = None
    if rank == 0:
        n_amp = nw * args.steps * world
        value = n_amp / elapsed
        dom = max(prof, key=lambda k: prof[k]["ms"])
        traffic = pmc_traffic_bytes(dom, args, nw, prof[dom]["launches"] / max(args.steps, 1))
        dom, roof = roofline_of(prof, args.dtype, traffic)
        roof["note"] = ("bound: the launches of this category contract 2*I*J*K flops over each walker's live extents and "
                        "move (I*K + K*J + I*J) elements (both counted on the device); their ratio against the machine "
                        "balance decides whether the roofline is priced in bytes (hbm) or flops (mfma).  achieved = that "
                        "count / HIP-event time on the launch stream; mfma_tflops is the flop rate of the same launches.  "
                        "traffic = measured HBM bytes per full-size launch (PMC; null unless this run has the launch mix of "
                        "the committed passes).  reference_equivalent_tflops prices the same launches with the flops of the "
                        "reference ops they replace (SURVEY 8d): the rank-adaptive path needs far fewer flops than the "
                        "reference algorithm on this workload (workload_rank).  DESIGN.md section 3.")
        out = {
            "metric": "configuration-amplitudes/sec",
            "value": value,
            "unit": "amplitudes/s",
            "n_gpus": world,
            "ranks_reported_by_backend": dist.get_world_size() if dist else 1,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {
                "workload": workload,
                "walkers_per_gpu": nw,
                "parallelism": "walkers sharded over %d GPU(s), no data-path collective" % world,
                "flops_per_amplitude_reference_algorithm": fl["total"],
                "synthetic_noise": args.noise,
                "state": args.state,
            },
            "roofline": roof,
            "job_tflops_reference_count": value * fl["total"] / 1e12 / world,
            "job_frac_of_peak": value * fl["total"] / 1e12 / world / PEAK_TFLOPS[args.dtype],
            "kernel_ms": {k: round(v["ms"], 3) for k, v in prof.items() if v["launches"]},
            "mfma": mfma_summary(prof, args.dtype, elapsed),
            "walkers_with_vanishing_amplitude": nz,
        }
        out["workload_rank"] = leg.rank_diagnostics()
        # size-independent property at the full size, outside the timed region: the amplitude of the same configuration
        # contracted row-wise (DOWN stack, trace at row 0) and column-wise (RIGHT stack, trace at column 0) must agree
        nrc = 0 if args.no_route_check else min(nw, 2048)
        if nrc:
            rctx = capi.Context(L, L, D, leg.pdim, chi, dtype=dt, device=local_rank, max_walkers=nrc)
            rctx.state_upload(leg.flat)
            rctx.set_configs(leg.batches[0][:nrc])
            a_row = rctx.evaluate_amplitude()
            # (fermions: the column pass runs on the column-major decorated components; the two values then differ by the
            # reordering sign of the occupied modes, so magnitudes are compared)
            rctx.set_configs(leg.fstate.ext_config(leg.phys[0][:nrc], leg.fermion.COL) if leg.fermion else leg.batches[0][:nrc])
            rctx.grow_bmps_for_col(0)
            rctx.init_bten(capi.UP, 0)
            rctx.grow_full_bten(capi.DOWN, 0, 2, True)
            a_col = rctx.trace(0, 0, capi.VERTICAL)
            ratio = np.abs(a_col / a_row) if leg.fermion else a_col / a_row
            out["route_consistency"] = {"max_rel_spread_row_vs_column_contraction": float(np.max(np.abs(ratio - 1))),
                                        "n": int(nrc)}
            rctx.close()
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(leg.flat, leg.batches[0][:max(64, 2 * (os.cpu_count() or 1))], chi, args.cpu_seconds)
            amps = cb.pop("amps")
            out["cpu_baseline"] = {
                "value": cb["all_cores"]["value"], "unit": "amplitudes/s", "cores": cb["all_cores"]["threads"], "kind": "port",
                "sample": "%d configuration(s) of the same %s workload, one walker per process on %d core(s), BLAS threads = 1 "
                          "(the reference's execution model, monte_carlo_engine.h:97-98), through oracle/cbmps.c: plain C on "
                          "LAPACK dgelqf/dorglq/dgesdd/dgemm, float64, op-for-op restatement of bmps_impl.h:756-862,225-263; "
                          "the upstream binary cannot be built here"
                          % (cb["all_cores"]["n"], args.workload, cb["all_cores"]["threads"]),
                "single_thread": {"value": cb["single"]["value"], "cores": 1, "n": 1,
                                  "sample": "one walker on one thread (one reference MPI rank)"},
                "host_cores": cb["cores"],
            }
            n = len(amps)
            if leg.fermion:
                amps = amps * leg.fstate.sigma(leg.phys[0][:n])      # the C restatement contracts the decorated network
            rel = np.abs(leg.amps_first[:n] / amps - 1)
            out["parity_on_sample"] = {"max_rel_err_amplitude": float(np.max(rel)), "median_rel_err_amplitude": float(np.median(rel)),
                                       "rms_err_over_rms_amplitude": float(np.sqrt(np.sum((leg.amps_first[:n] - amps) ** 2) / np.sum(amps ** 2))),
                                       "n": int(n), "tolerance": 1e-5, "checker": "oracle/cbmps.c (float64)"}
            if leg.fermion:
                out["parity_on_sample"]["note"] = ("fermionic amplitudes are alternating sums: a configuration whose amplitude is "
                                                   "orders of magnitude below the typical one loses that many digits in f32 (max); the "
                                                   "tolerance 1e-4 applies to the rms error over the rms amplitude, the f64 mode is the "
                                                   "parity-grade path (DESIGN.md 3b)")
                out["parity_on_sample"]["tolerance"] = 1e-4
        if world == 1 and not args.no_cpu_baseline and not args.no_energy_check and not fermionic:
            try:
                out["energy_parity"] = energy_parity(leg, 1, 120.0)
                out["energy_rel_err"] = out["energy_parity"]["max_rel_err_energy"]
            except Exception as e:      # the main line must survive a failure of a diagnostic
                out["energy_parity"] = {"error": repr(e)}
    leg.close()
    del leg

    # ---- further legs on the same shapes: a state of full rank (i.i.d. random site tensors) and a state of the rank of a
    #      REAL PEPS (the reference's optimised 4x4 D=8 fixture tiled to L x L, configurations near the Neel state) ----
    def extra_leg(real, noise, walkers, steps):
        fr = None
        fleg = None
        try:
            fnw = min(walkers, nw)
            fleg = Leg(capi, synthetic, L, D, chi, dt, local_rank, fnw, noise, real=real)
            fel, fprof, fnz = fleg.run(steps, 1, rank, world, barrier)
        except Exception as e:
            fr = {"error": repr(e)}
            fel, fprof, fnz = 0.0, None, 0
        # every rank reaches the collectives below whatever happened on it: a rank-local failure must not leave the others
        # waiting in an all-reduce (the failure flag is reduced with MAX, then all ranks skip the leg together)
        failed = max_over_ranks(1.0 if fr is not None else 0.0) > 0
        fel = max_over_ranks(fel)
        if failed:
            if fleg is not None:
                fleg.close()
            return fr if fr is not None else {"error": "another rank failed on this leg"}
        try:
            if rank == 0:
                fdom, froof = roofline_of(fprof, args.dtype)
                fr = {"value": fnw * steps * world / fel, "unit": "amplitudes/s", "walkers_per_gpu": fnw,
                      "steps": steps, "warmup": 1, "ms_per_step": fel / steps * 1e3,
                      "state": ("reference fixture tps_square_heisenberg4x4D8Double tiled by position class to %dx%d, "
                                "configurations = checkerboard + %d random NN exchanges" % (L, L, L * L // 8)) if real
                               else "i.i.d. random site tensors (synthetic noise 1.0)",
                      "roofline": froof,
                      "kernel_ms": {k: round(v["ms"], 3) for k, v in fprof.items() if v["launches"]},
                      "mfma": mfma_summary(fprof, args.dtype, fel),
                      "walkers_with_vanishing_amplitude": fnz, "workload_rank": fleg.rank_diagnostics()}
                if not real:
                    fr["synthetic_noise"] = noise
                if world == 1 and not args.no_cpu_baseline:
                    from oracle import cbmps
                    k = min(8, os.cpu_count() or 1)
                    ra, _, _ = cbmps.amplitudes_multiprocess(fleg.flat, fleg.batches[0][:k], chi, k)
                    rel = np.abs(fleg.amps_first[:k] / ra - 1)
                    fr["parity_on_sample"] = {"max_rel_err_amplitude": float(np.max(rel)), "median_rel_err_amplitude": float(np.median(rel)),
                                              "n": int(k), "tolerance": 1e-5, "checker": "oracle/cbmps.c (float64)"}
        except Exception as e:
            fr = {"error": repr(e)}
        fleg.close()
        return fr

    if not args.no_full_rank and args.noise < 0.5 and not fermionic and args.state == "synthetic":
        fr = extra_leg(False, 1.0, args.full_rank_walkers, args.full_rank_steps)
        if rank == 0:
            if fr is not None and "error" not in fr:
                fr["real_state_rank"] = real_state_rank(capi, local_rank, dt)
            out["full_rank"] = fr
    if not args.no_real_rank and args.noise < 0.5 and not fermionic and args.state == "synthetic" and D == 8 and os.path.isdir(REAL_STATE):
        rr = extra_leg(True, 0.0, args.real_rank_walkers, args.real_rank_steps)
        if rank == 0:
            out["real_rank"] = rr

    # ---- the exchange step, outside the timed region: all-reduce of the HBM-resident accumulators over RCCL ----
    if dist is not None and backend == "nccl":
        coll = None
        try:
            from peps_amd import dist as pdist
            cctx = capi.Context(L, L, D, 2, chi, dtype=dt, device=local_rank, max_walkers=16)
            cctx.grad_reset()
            so, seo, n = cctx.grad_device_ptr()
            cctx.sync()
            ts = [pdist.device_tensor(p, n) for p in (so, seo)]
            for t in ts:
                t.fill_(1.0)
            torch.cuda.synchronize()
            dist.barrier()
            t0 = time.perf_counter()
            for t in ts:
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
            torch.cuda.synchronize()
            t_torch = time.perf_counter() - t0
            ok = bool(abs(float(ts[0][0].item()) - world) < 1e-12 and abs(float(ts[1][-1].item()) - world) < 1e-12)
            pdist.comm_init(cctx)
            cctx.grad_allreduce()          # first call builds the rings
            dist.barrier()
            t0 = time.perf_counter()
            cctx.grad_allreduce()
            t_lib = time.perf_counter() - t0
            coll = {"what": "all-reduce(sum) of the device-resident gradient accumulators S_O, S_EO (float64), in place in HBM",
                    "bytes": int(2 * n * 8), "ranks": dist.get_world_size(), "result_correct": ok,
                    "ms_torch_distributed_nccl": t_torch * 1e3, "ms_pepsgpu_grad_allreduce": t_lib * 1e3,
                    "library_comm_size": cctx.comm_size()}
            cctx.close()
        except Exception as e:
            coll = {"error": repr(e)}
        if rank == 0:
            out["collective"] = coll

    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
