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
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"f32": 157.3, "f64": 78.6, "i8": 5000.0}     # /opt/skills/guides/MI355X_MICROARCH.md, dense MFMA peaks (i8: 2 x the bf16 rate)
I8_PRODUCTS = 9      # byte products per float64-grade product of the exact-integer Gram kernels (peps_amd/csrc/gram_i8.h)
PEAK_HBM_GBPS = 8000.0                        # same guide: HBM3E, ~8 TB/s
REAL_STATE = os.path.join(ROOT, "tests", "golden", "ref_fixtures", "tps_square_heisenberg4x4D8Double")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--walkers", type=int, default=None,
                    help="walkers (fresh configurations) per GPU per step; default 49152 (121 GB at the low-rank "
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
    ap.add_argument("--full-rank-walkers", type=int, default=8192)
    ap.add_argument("--full-rank-steps", type=int, default=5)
    ap.add_argument("--no-real-rank", action="store_true", help="skip the third leg on the tiled optimised state of the reference")
    ap.add_argument("--real-rank-walkers", type=int, default=12288)
    ap.add_argument("--real-rank-steps", type=int, default=5)
    ap.add_argument("--no-sweeps", action="store_true", help="skip the MC sweeps/s and VMC samples/s measurement")
    ap.add_argument("--sweep-walkers", type=int, default=16384)
    ap.add_argument("--real-sweep-walkers", type=int, default=2048, help="walkers of the sweep / VMC-sample rates on the real_rank leg")
    ap.add_argument("--real-sweep-count", type=int, default=3, help="timed sweeps / samples there")
    ap.add_argument("--sweep-count", type=int, default=10)
    ap.add_argument("--no-other-modes", action="store_true", help="skip the short runs of the f64 / variational / complex / C5 modes")
    ap.add_argument("--no-latency", action="store_true", help="skip the one-walker latency measurement (n1_ms)")
    ap.add_argument("--energy-n", type=int, default=8, help="configurations of the E_loc parity sample of the main leg (half of it on the extra legs)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / rendezvous / reduction plumbing only, NO device work: value is null (CPU tests of --gpus N)")
    args = ap.parse_args()
    if args.walkers is None:
        args.walkers = 49152 if (args.dtype == "f32" and args.noise <= 0.15 and args.state == "synthetic") else (2048 if args.state == "real" else 4096)
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


def host_cpu_info():
    """What 'cores' means on this box: logical CPUs, affinity mask, cgroup CPU quota (cpu.max: quota / period), physical cores
    and SMT threads per core from /proc/cpuinfo."""
    info = {"logical_cpus": os.cpu_count()}
    try:
        info["affinity_cpus"] = len(os.sched_getaffinity(0))
    except Exception:
        pass
    for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(p).read().split()
            if p.endswith("cpu.max"):
                info["cgroup_cpu_max"] = " ".join(txt)
                info["cgroup_effective_cpus"] = None if txt[0] == "max" else float(txt[0]) / float(txt[1])
            else:
                per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                q = float(txt[0])
                info["cgroup_effective_cpus"] = None if q < 0 else q / per
            break
        except Exception:
            continue
    try:
        cores, sib = set(), None
        phys = core = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
                cores.add((phys, core))
            elif line.startswith("siblings") and sib is None:
                sib = int(line.split(":")[1])
            elif line.startswith("cpu cores") and "cores_per_socket" not in info:
                info["cores_per_socket"] = int(line.split(":")[1])
        info["physical_cores"] = len(cores) or None
        if sib and info.get("cores_per_socket"):
            info["smt_threads_per_core"] = sib // info["cores_per_socket"]
    except Exception:
        pass
    try:
        info["loadavg_1min_before"] = os.getloadavg()[0]
    except Exception:
        pass
    return info


def cpu_baseline(flat, cfgs, chi, budget_s):
    """The reference's CPU path restated in plain C on LAPACK (oracle/cbmps.c: bmps_impl.h:756-862, :225-263 op for op,
    float64), timed on this host in the reference's execution model (independent walkers, one per PROCESS as one per MPI
    rank, BLAS threads = 1): (i) one process, one walker; (ii) an intermediate point (32 processes); (iii) one walker per
    core on all cores.  Each leg reports the busy seconds of its worker processes (min / median / max): a socket whose
    memory system is saturated by 256 concurrent LAPACK processes runs each of them many times slower than alone.  The
    pool lives in a child interpreter (oracle/cbmps.py) that never touches the GPU.  Bounded sample; returns a dict."""
    from oracle import cbmps
    info = host_cpu_info()
    logical = info.get("affinity_cpus") or info.get("logical_cpus") or 1
    # CPUs this process may actually use: the cgroup CPU quota when there is one (a container on a 256-thread host with
    # cpu.max = "1600000 100000" gets 16 CPUs' worth of time however many processes it starts)
    eff = info.get("cgroup_effective_cpus")
    cores = int(max(1, min(logical, round(eff)))) if eff else int(logical)

    def leg(n, nproc):
        d = cbmps.amplitudes_multiprocess_detail(flat, cfgs[:n], chi, nproc)
        ps = np.sort(np.asarray(d["proc_seconds"], dtype=np.float64))
        return {"value": n / d["seconds"], "n": int(n), "threads": int(d["nprocs"]), "seconds": d["seconds"],
                "per_process_seconds": {"min": float(ps[0]), "median": float(np.median(ps)), "max": float(ps[-1])},
                "walkers_per_process": float(n) / d["nprocs"]}, d["amps"]

    single, _ = leg(1, 1)
    s1 = single["seconds"]
    out = {"single": single, "cores": int(cores), "logical_cpus": int(logical), "host": info}
    # all usable cores: as many rounds of `cores` walkers as the budget allows (at least two), capped by the sample at hand
    rounds = max(2, int((budget_s - s1) / max(s1 * 2.5, 1e-3)))
    n_all = min(len(cfgs), cores * rounds)
    allc, amps = leg(n_all, min(cores, n_all))
    out["all_cores"] = allc
    out["amps"] = amps
    if logical >= 2 * cores and len(cfgs) >= 2 * cores:      # oversubscription probe: twice as many processes as usable CPUs
        over, _ = leg(2 * cores, 2 * cores)
        out["oversubscribed_2x"] = over
    return out


# the newest committed PMC summary (scripts/make_pmc_meta.py N): traffic and SQ shares are quoted only for a run of the profiled shape
PMC_META = next((p for p in (os.path.join(ROOT, "profiles", "r%02d_pmc_meta.json" % r) for r in (6, 5, 4, 3)) if os.path.exists(p)),
                os.path.join(ROOT, "profiles", "r04_pmc_meta.json"))


def pmc_traffic_bytes(kernel, leg_tag, nw, launches_per_step):
    """HBM bytes per launch of one kernel from the committed rocprofv3 --pmc passes (separate FETCH_SIZE / WRITE_SIZE passes,
    KiB, FETCH DOUBLED as MI355X_MICROARCH.md 'HBM' prescribes for gfx950; scripts/make_pmc_meta.py).  PMC counters cannot be
    read from inside this process, so the figure is quoted only when this run is the run the passes were collected on: same
    leg, same walkers AND the same number of launches of that kernel per step (profiles/r03_pmc_meta.json); any difference
    (a changed kernel mix) returns null instead of a stale number."""
    if not os.path.exists(PMC_META):
        return None
    try:
        meta = json.load(open(PMC_META)).get(leg_tag)
    except Exception:
        return None
    if not meta or int(meta.get("walkers", -1)) != int(nw):
        return None
    ent = meta.get("kernels", {}).get(kernel)
    if not ent:
        return None
    # bytes PER LAUNCH of the dominant kernel depend on the state, the batch and the kernel, not on how many launches a step holds:
    # the extra legs of the driver run see a few launches more or fewer than the profiled single-leg command (different warm-up /
    # hint history: 253 against 220.5 in round 4, which left real_rank.roofline.traffic null).  Quoted when the counts agree within
    # 20 %, with both counts beside the figure; a changed kernel mix (another kernel, another batch, a count further off) is null.
    prof_lps = float(ent.get("launches_per_step", -1))
    if prof_lps <= 0 or abs(prof_lps - float(launches_per_step)) > 0.2 * prof_lps:
        return None
    return {"bytes_per_launch": float(ent["hbm_bytes_per_launch"]), "avg_us_rocprof": ent.get("avg_us"),
            "source": meta.get("source"), "launches_per_step_profiled": prof_lps}


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
        self.local_elapsed = elapsed
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


# profiling categories that are ONE kernel (a bracket = a launch of that kernel) or one family of tensor-GEMM kernels, with
# device-counted flops and bytes: the candidates of the roofline object.  The other categories (Gram-free factor, Jacobi,
# select ...) are VALU / latency bound and listed with their share of the step in `kernel_ms`.
ROOF_CATS = {"contract_chain": ("tgemm_chain_kernel", "f32"), "contract": ("tgemm_direct_kernel", "f32"),
             "gram_f64": ("gram_cols_i8_kernel (193..256 columns: exact integer Gram, 9 i8 products) / gram_cols_f64_kernel", "i8x9"),
             "trunc_gram": ("gram_cols_i8_kernel<ROWS> + chol_upper_kernel / mid_gram_chol_kernel", "i8x9"),
             "trunc_apply": ("tgemm_kernel<f32,f32,f32,f64>", "f64"), "env": ("tgemm_kernel (BTen / trace)", "f32")}


# categories without a flop or byte roofline (VALU / exchange latency): named dominant when they are the largest (bound "valu")
VALU_CATS = {"jacobi": "jacobi_rows_grp_kernel / jacobi_rows_tiny4_kernel (one-sided Jacobi, register tournament)",
             "jacobi_edge": "jacobi_rows_kernel", "cholesky": "gram_chol_wave_kernel (Gram-free factor)",
             "select": "select_rows_kernel / ortho_rows_kernel", "normalize": "normalize_kernel"}


def pmc_sq_shares(kernel_names, leg_tag):
    """SQ_WAIT_ANY / SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_VALU as shares of SQ_WAVE_CYCLES of the kernels of a category, from the
    committed SQ pass of the same leg (profiles/r0N_pmc_meta.json, key "sq"); null when the pass is not there"""
    if not os.path.exists(PMC_META):
        return None
    try:
        sq = json.load(open(PMC_META)).get(leg_tag, {}).get("sq", {})
    except Exception:
        return None
    out = {}
    for k, v in sq.items():
        if any(k in part or part.split()[0] in k for part in kernel_names.split(" / ")):
            out[k] = v
    return out or None


def roofline_of(prof, dtype, steps, leg_tag=None, nw=None):
    """Roofline object of the dominant kernel of a leg: the single-kernel category with the largest HIP-event time."""
    cats = dict(ROOF_CATS)
    # the factor category is one MFMA kernel family only on dense states (colgram_dense_kernel: Gram in accumulators + Cholesky in
    # LDS, flops and bytes counted on the device); on the headline it is the VALU Gram-free factor (no counted flops): not a candidate
    if "cholesky" in prof and prof["cholesky"]["exec_flops"] > 0 and prof["cholesky"].get("bytes", 0) > 0:
        cats["cholesky"] = ("colgram_dense_kernel (+ chol_blocked_kernel, chol_lowrank_kernel)", "f64")
    cands = [k for k in cats if k in prof and prof[k]["launches"] and prof[k]["ms"] > 0]
    dom = max(cands, key=lambda k: prof[k]["ms"])
    # A category that runs no MFMA and streams little (one-sided Jacobi sweeps, the Gram-free factor, select / normalise) can be
    # the largest of a leg: it is then NAMED as the dominant one (bound "valu": no flop or byte roofline applies, frac null) with
    # its wave-cycle wait shares from the committed SQ pass, and the largest priced kernel rides along.
    valu = {k: v for k, v in VALU_CATS.items() if k in prof and prof[k]["launches"] and prof[k]["ms"] > 0 and k not in cats}
    vdom = max(valu, key=lambda k: prof[k]["ms"]) if valu else None
    valu_roof = None
    if vdom and prof[vdom]["ms"] > prof[dom]["ms"]:
        tot = sum(v["ms"] for v in prof.values())
        vl = max(prof[vdom]["launches"], 1)
        valu_roof = {"bound": "valu", "kernel": VALU_CATS[vdom], "category": vdom, "achieved": None, "peak": None, "unit": None, "frac": None,
                     "traffic": None, "avg_launch_us": prof[vdom]["ms"] / vl * 1e3, "launches_per_step": vl / max(steps, 1),
                     "share_of_kernel_time": prof[vdom]["ms"] / tot if tot > 0 else None,
                     "sq_wait": pmc_sq_shares(VALU_CATS[vdom], leg_tag) if leg_tag else None}
    kname, kdt = cats[dom]
    dsec = prof[dom]["ms"] * 1e-3
    launches = max(prof[dom]["launches"], 1)
    total_ms = sum(v["ms"] for v in prof.values())
    counted = prof[dom]["exec_flops"] if prof[dom]["exec_flops"] > 0 else prof[dom]["alg_flops"]
    # exact-integer Gram: the device counts the float64-grade flops of the Gram; the matrix cores execute nine i8 products for each
    i8 = kdt == "i8x9" and os.environ.get("PEPSGPU_NO_I8_GRAM") is None
    f64_equiv = counted / dsec / 1e12
    tflops = f64_equiv * (I8_PRODUCTS if i8 else 1)
    dbytes = prof[dom].get("bytes", 0.0)
    intensity = counted / dbytes if dbytes > 0 else float("inf")
    kpeak = PEAK_TFLOPS["i8"] if i8 else PEAK_TFLOPS["f64"] if kdt in ("f64", "i8x9") else PEAK_TFLOPS[dtype]
    balance = kpeak * 1e12 / (PEAK_HBM_GBPS * 1e9)
    hbm_bound = intensity < balance
    alg_gbps = dbytes / dsec / 1e9
    avg_ms = prof[dom]["ms"] / launches
    pmc = pmc_traffic_bytes(kname, leg_tag, nw, launches / max(steps, 1)) if leg_tag else None
    traffic = pmc["bytes_per_launch"] if pmc else None
    if hbm_bound:
        # priced with the MEASURED HBM traffic of the kernel when the committed PMC passes are of this very run shape, else
        # with the device-counted operand + result bytes (which charge the L2-resident site tensor to every walker)
        achieved = (traffic / (avg_ms * 1e-3) / 1e9) if traffic else alg_gbps
        frac = achieved / PEAK_HBM_GBPS
    else:
        achieved, frac = tflops, tflops / kpeak
    roof = {
        "bound": "hbm" if hbm_bound else "mfma",
        "kernel": kname,
        "category": dom,
        "achieved": achieved,
        "peak": PEAK_HBM_GBPS if hbm_bound else kpeak,
        "unit": "GB/s" if hbm_bound else "TFLOP/s",
        "frac": frac,
        "frac_priced_with": ("pmc_traffic" if traffic else "device_counted_bytes") if hbm_bound else "device_counted_flops",
        "traffic": traffic,
        "traffic_source": pmc["source"] if pmc else None,
        "traffic_launches_per_step_profiled": pmc["launches_per_step_profiled"] if pmc else None,
        "avg_launch_us": avg_ms * 1e3,
        "avg_launch_us_rocprof": pmc["avg_us_rocprof"] if pmc else None,
        "launches_per_step": launches / max(steps, 1),
        "share_of_kernel_time": prof[dom]["ms"] / total_ms if total_ms > 0 else None,
        "counted": {"bytes_per_launch": dbytes / launches, "GBps": alg_gbps, "frac_of_hbm_peak": alg_gbps / PEAK_HBM_GBPS,
                    "flops_per_launch": counted / launches},
        "arithmetic_intensity_flop_per_byte": intensity if dbytes > 0 else None,
        "machine_balance_flop_per_byte": balance,
        "mfma_tflops": tflops,
        "mfma_frac": tflops / kpeak,
    }
    if i8:
        roof["mfma_dtype"] = "i8 (v_mfma_i32_16x16x64_i8, %d byte products per float64-grade product)" % I8_PRODUCTS
        roof["float64_equivalent_tflops"] = f64_equiv
        roof["float64_mfma_peak_it_replaces"] = PEAK_TFLOPS["f64"]
    if valu_roof is not None:
        valu_roof["largest_priced_kernel"] = roof
        return vdom, valu_roof
    return dom, roof


def mfma_summary(prof, dtype, step_seconds_total):
    """Executed MFMA flops of every category that runs on the matrix cores (device-counted 2*I*J*K over live extents) against
    their own time and against the whole timed region."""
    cats = {}
    tot_fl = tot_ms = 0.0
    for k in ("contract", "contract_chain", "gram_f64", "cholesky", "env", "trunc_gram", "trunc_apply"):
        if k in prof and prof[k]["launches"] and prof[k]["exec_flops"] > 0:
            pk = PEAK_TFLOPS["f64"] if k in ("gram_f64", "cholesky", "trunc_gram", "trunc_apply") else PEAK_TFLOPS[dtype]
            tf = prof[k]["exec_flops"] / (prof[k]["ms"] * 1e-3) / 1e12 if prof[k]["ms"] > 0 else 0.0
            cats[k] = {"ms": round(prof[k]["ms"], 3), "tflops": tf, "peak": pk, "frac": tf / pk}
            if k in ("gram_f64", "trunc_gram") and os.environ.get("PEPSGPU_NO_I8_GRAM") is None:
                # (float64-grade flops counted on the device; the dense walkers run them as 9 i8 products each, gram_i8.h: the
                # fraction is of the float64 matrix peak the integer kernel replaces and may exceed 1)
                cats[k]["note"] = "float64-equivalent flops; dense launches execute on the i8 matrix cores"
            tot_fl += prof[k]["exec_flops"]
            tot_ms += prof[k]["ms"]
    return {"categories": cats,
            "mfma_tflops_in_mfma_kernels": tot_fl / (tot_ms * 1e-3) / 1e12 if tot_ms > 0 else 0.0,
            "mfma_tflops_over_whole_step": tot_fl / step_seconds_total / 1e12 if step_seconds_total > 0 else 0.0,
            "mfma_kernel_time_share": tot_ms * 1e-3 / step_seconds_total if step_seconds_total > 0 else 0.0}


def energy_parity_start(flat, cfgs, chi):
    """Starts the float64 oracle (oracle/epool.py: the reference's XXZ solver schedule restated, one configuration per
    single-threaded worker process, in a child interpreter that never touches the GPU); energy_parity_finish() joins it and
    compares with the device (C++ host layer, pepshost_energy_and_holes).  Started only AFTER every timed GPU leg: the box gives a
    container 16 CPUs' worth of time, and a busy oracle pool throttles the host thread that launches the kernels."""
    from oracle import epool
    return {"h": epool.start(flat, cfgs, chi, (1.0, 1.0, 0.0), nprocs=len(cfgs), blas_threads=1), "cfgs": cfgs}


def energy_parity_finish(leg_info, pend, timeout_s):
    from peps_amd import hostapi
    from oracle import epool
    L, chi, dt, device, flat = leg_info
    hostapi.set_device(device)
    res = hostapi.energy_and_holes(flat, pend["cfgs"], chi, model="xxz", params=(1.0, 1.0, 0.0), holes=False, dtype=dt)
    e_dev = np.asarray(res[1])
    e_ref, a_ref, sec = epool.collect(pend["h"], timeout=timeout_s)
    return {"max_rel_err_energy": float(np.max(np.abs(e_dev - e_ref) / np.abs(e_ref))), "n": int(len(e_ref)),
            "tolerance": 1e-6,
            "max_rel_err_amplitude": float(np.max(np.abs(np.asarray(res[0]) / a_ref - 1))),
            "e_per_site_device": [float(x) / (L * L) for x in e_dev],
            "e_per_site_oracle": [float(x) / (L * L) for x in e_ref],
            "checker": "oracle/epool.py (float64 NumPy restatement of the reference's XXZ solver, one process per configuration)",
            "oracle_seconds": sec}


def vmc_rates(leg, nw, n_sweeps):
    """What VMC consumes: Monte-Carlo sweeps/s of the NN-exchange updater (square_nn_updater.h:25-83: 4(L-1) row /
    column absorptions + 2L(L-1) replacement traces per sweep and walker, Metropolis on the host per bond) and complete VMC
    samples/s (one sweep + CalEnergyAndHoles + O* accumulation with the holes resident in HBM, mc_energy_grad_evaluator.h:245-278)
    through the C++ host layer, on `nw` walkers of the leg's state; marginal rates (set-up cancels)."""
    from peps_amd import hostapi
    hostapi.set_device(leg.device)
    dtc = 0 if leg.dt == leg.capi.F32 else 1
    cfgs = leg.batches[0][:nw]
    seeds = np.arange(nw, dtype=np.uint64) + 100
    # marginal rates: wall time of a call with 1 + n sweeps minus that of a call with 1 sweep -- the context set-up, the state
    # upload and the first-pass allocations of the host-layer call (up to a second at 8192 walkers, and moving with the state of
    # the allocator: the figure including them varied by 20 % between otherwise identical runs) cancel
    hostapi.mc_sweeps(leg.flat, cfgs, seeds, leg.chi, "exchange", 1, dtc)                      # untimed: kernels loaded
    t0 = time.perf_counter()
    hostapi.mc_sweeps(leg.flat, cfgs, seeds, leg.chi, "exchange", 1, dtc)
    t_sw1 = time.perf_counter() - t0
    t0 = time.perf_counter()
    _, _, rates = hostapi.mc_sweeps(leg.flat, cfgs, seeds, leg.chi, "exchange", 1 + n_sweeps, dtc)
    t_swn = time.perf_counter() - t0
    # (the fixed part of a call moves by up to a second between identical calls -- allocator state, first touch of the hole store --
    # and only upwards: the one-sweep call is timed again BEHIND the long call and the smaller of the two is subtracted, which can only
    # under-report the marginal rate)
    t0 = time.perf_counter()
    hostapi.mc_sweeps(leg.flat, cfgs, seeds, leg.chi, "exchange", 1, dtc)
    t_sw1 = min(t_sw1, time.perf_counter() - t0)
    t_sw = max(t_swn - t_sw1, 1e-9)
    hostapi.mc_energy_grad_partial(leg.flat, cfgs, seeds, leg.chi, "exchange", "xxz", (1.0, 1.0, 0.0), 0, 1, dtc)   # untimed
    t0 = time.perf_counter()
    hostapi.mc_energy_grad_partial(leg.flat, cfgs, seeds, leg.chi, "exchange", "xxz", (1.0, 1.0, 0.0), 0, 1, dtc)
    t_v1 = time.perf_counter() - t0
    t0 = time.perf_counter()
    packed, _, acc = hostapi.mc_energy_grad_partial(leg.flat, cfgs, seeds, leg.chi, "exchange", "xxz", (1.0, 1.0, 0.0), 0, 1 + n_sweeps, dtc)
    t_vn = time.perf_counter() - t0
    t0 = time.perf_counter()
    hostapi.mc_energy_grad_partial(leg.flat, cfgs, seeds, leg.chi, "exchange", "xxz", (1.0, 1.0, 0.0), 0, 1, dtc)
    t_v1 = min(t_v1, time.perf_counter() - t0)
    t_vmc = max(t_vn - t_v1, 1e-9)
    e, _ = hostapi.exact_sum_finish(packed, leg.flat.shape)
    return {"mc_sweeps_per_s": n_sweeps * nw / t_sw, "vmc_samples_per_s": n_sweeps * nw / t_vmc, "walkers": int(nw),
            "sweeps_timed": int(n_sweeps), "updater": "MCUpdateSquareNNExchangeOBC", "accept_rate": float(np.mean(rates)),
            "mc_energy_per_site": float(e) / (leg.L * leg.L),
            "what": "sweep = 4(L-1) absorptions + 2L(L-1) replacement traces per walker; VMC sample = sweep + CalEnergyAndHoles + O* "
                    "accumulation (holes resident in HBM); marginal wall time per sweep / sample of the host-layer call (1 + n against 1)",
            "call_seconds": {"sweeps_1": t_sw1, "sweeps_1_plus_n": t_swn, "samples_1": t_v1, "samples_1_plus_n": t_vn}}


def c5_parity(capi, fermion, st, chi, device, cfgs):
    """f32 and f64 device amplitude + t-V local energy of a few C5 configurations against the graded float64 oracle (checker only)."""
    from oracle import fermion as ofermion
    from oracle.bmps import BMPSTruncateParams
    from oracle.graded import GT
    gts = [[[GT(st.tensors[r][c][s][..., None], list(st.par[r][c]) + [np.array([int(st.nf[s])])], [-1, 1, 1, -1, -1])
             for s in range(st.d)] for c in range(st.cols)] for r in range(st.rows)]
    fs = ofermion.FermionSITPS(gts)
    tp = BMPSTruncateParams.SVD(chi, chi, 0.0)
    model = ofermion.SquareSpinlessFermionOBC(1.0, 0.0, 1.0)
    # configurations as a Monte-Carlo run visits them (float64 device chains of the C++ host layer: three NN-exchange sweeps from the
    # half-filled starts handed in) -- random occupations have amplitudes at the rounding level of the alternating sum
    from peps_amd import hostapi
    cfgs, _, _ = hostapi.fermion_mc_sweeps(st, cfgs, np.arange(len(cfgs), dtype=np.uint64) + 700, chi, 3, 1)
    ref_a = np.array([fs.amplitude(c, tp) for c in cfgs])
    ref_e = np.array([model.CalEnergy(fs, c, tp)[0] for c in cfgs])
    res = {"n": int(len(cfgs)), "model": "t = 1, V = 1", "configurations": "end points of float64 device chains (3 sweeps, seeds 700..)"}
    for name, dt in (("f32", capi.F32), ("f64", capi.F64)):
        ctx = capi.Context(st.rows, st.cols, st.D, fermion.NVAR * st.d, chi, dtype=dt, device=device, max_walkers=len(cfgs))
        ctx.state_upload(st.extended_flat(st.D))
        amp = fermion.evaluate_amplitude(ctx, st, cfgs)
        e_loc, _ = fermion.spinless_fermion_energy(ctx, st, cfgs, 1.0, 1.0)
        ctx.close()
        res[name] = {"amplitude_max_rel": float(np.max(np.abs(amp / ref_a - 1))), "energy_max_rel": float(np.max(np.abs(e_loc / ref_e - 1)))}
    return res


def mode_profile(mode):
    """dominant kernels of a secondary mode from its committed rocprofv3 runs (profiles/r06_mode_meta.json, scripts/make_mode_meta.py):
    share of the kernel time, HBM bytes per launch and TB/s from the PMC passes, SQ shares; None when the file is not there"""
    p = os.path.join(ROOT, "profiles", "r06_mode_meta.json")
    try:
        e = json.load(open(p)).get(mode)
    except Exception:
        return None
    if not e:
        return None
    return {"source": e["source"], "profiled_amp_per_s": (e.get("probe") or {}).get("amp_per_s"), "profiled_walkers": (e.get("probe") or {}).get("walkers"),
            "dominant_kernels": [{k: v for k, v in kk.items() if k != "calls"} for kk in e["kernels"][:3]]}


def other_modes(capi, synthetic, device, L, D, chi):
    """Driver-measured figures of the secondary modes of the path (one short run each, outside the timed region; rounds 1-2 quoted
    them from builder scripts only): the f64 device mode, the two variational compression schemes (bmps_impl.h:864-1172), the
    complex element type, and BASELINE config C5 (8x8 spinless fermions, fZ2-graded, D = 6, chi = 24, sign-decorated path)."""
    out = {}

    def rate(ctx, batches, post=None):
        ctx.set_configs(batches[0]); ctx.evaluate_amplitude()
        ctx.sync(); t0 = time.perf_counter()
        for b in batches[1:]:
            ctx.set_configs(b); a = ctx.evaluate_amplitude()
        ctx.sync()
        return sum(len(b) for b in batches[1:]) / (time.perf_counter() - t0)

    sitps = synthetic.make_sitps(L, D)
    flat = synthetic.sitps_to_flat(sitps, D, np.float64)
    try:
        nw = 2048
        c = capi.Context(L, L, D, 2, chi, dtype=capi.F64, device=device, max_walkers=nw)
        c.state_upload(flat)
        out["f64_mode"] = {"amp_per_s": rate(c, [synthetic.make_configs(L, nw, "heisenberg", seed0=50000 + 7 * k) for k in range(2)]), "walkers": nw}
        c.close()
    except Exception as e:
        out["f64_mode"] = {"error": repr(e)}
    for name, scheme in (("variational_2site", 1), ("variational_1site", 2)):
        try:
            nw = 1024
            c = capi.Context(L, L, D, 2, chi, dtype=capi.F32, device=device, max_walkers=nw, scheme=scheme, convergence_tol=1e-5, iter_max=3)
            c.state_upload(flat)
            out[name] = {"amp_per_s": rate(c, [synthetic.make_configs(L, nw, "heisenberg", seed0=60000 + 7 * k) for k in range(2)]), "walkers": nw,
                         "iter_max": 3, "convergence_tol": 1e-5}
            c.close()
        except Exception as e:
            out[name] = {"error": repr(e)}
    try:
        nw = 128
        rng = np.random.default_rng(5)
        cflat = flat * np.exp(2j * np.pi * rng.uniform(size=flat.shape))
        c = capi.Context(L, L, D, 2, chi, dtype=capi.C128, device=device, max_walkers=nw)
        c.state_upload(cflat)
        out["complex128"] = {"amp_per_s": rate(c, [synthetic.make_configs(L, nw, "heisenberg", seed0=70000 + 7 * k) for k in range(2)]), "walkers": nw,
                             "note": "static shapes (no rank adaptivity), GEMMs on the f64 matrix cores; this synthetic state falls to the resolution of a "
                                     "Gram within a few directions -- its figure on a dense state is real_rank.complex128 (round 4: 6.6, round 5: 82, "
                                     "round 6: 170 amp/s)"}
        c.close()
    except Exception as e:
        out["complex128"] = {"error": repr(e)}
    try:
        from peps_amd import fermion
        l5, d5, chi5, nw = 8, 6, 24, 8192
        st = fermion.random_even_state(l5, l5, d5, seed=11)
        c = capi.Context(l5, l5, d5, fermion.NVAR * st.d, chi5, dtype=capi.F32, device=device, max_walkers=nw)
        c.state_upload(st.extended_flat(d5))
        phys = [synthetic.make_configs(l5, nw, "heisenberg", seed0=80000 + 7 * k) for k in range(3)]
        c5 = {"amp_per_s": rate(c, [st.ext_config(p, fermion.ROW) for p in phys]), "walkers": nw, "dtype": "f32"}
        c.close()
        # the float64 device mode (the parity-grade path for fermions: C5's amplitudes are alternating sums) and what both modes
        # measure against the oracle on a sample (checker only, outside every timed region): tests/test_gpu_fermion.py::C5_TOL
        nw64 = 1024
        c = capi.Context(l5, l5, d5, fermion.NVAR * st.d, chi5, dtype=capi.F64, device=device, max_walkers=nw64)
        c.state_upload(st.extended_flat(d5))
        c5["f64_mode"] = {"amp_per_s": rate(c, [st.ext_config(p[:nw64], fermion.ROW) for p in phys[:2]]), "walkers": nw64, "profile": mode_profile("c5_f64")}
        c5["profile"] = mode_profile("c5_f32")
        c.close()
        c5["tolerance"] = {"f32": {"amplitude": 2e-5, "energy": 2e-5}, "f64": {"amplitude": 1e-7, "energy": 1e-7},
                           "north_star_energy": 1e-6, "note": "relative, asserted in tests/test_gpu_fermion.py (C5_TOL) on eight chain-visited "
                           "configurations; the f64 mode is the one that holds north_star's 1e-6 on fermionic energies"}
        try:
            c5["parity_on_sample"] = c5_parity(capi, fermion, st, chi5, device, phys[0][:4])
        except Exception as e:
            c5["parity_on_sample"] = {"error": repr(e)}
        out["C5_spinless_tV_8x8_D6_chi24"] = c5
    except Exception as e:
        out["C5_spinless_tV_8x8_D6_chi24"] = {"error": repr(e)}
    return out


def baseline_config_rates(capi, synthetic, device):
    """BASELINE configs C2 (8x8 TFIM, D = 4, chi = 16, "local MC updater" = MCUpdateSquareNNFullSpaceUpdateOBC), C3 (10x10 Heisenberg,
    D = 6, chi = 24, exchange updater) and C5 (8x8 spinless fermions, D = 6, chi = 24, exchange updater on the decorated network):
    fresh amplitudes/s through the C ABI and Monte-Carlo sweeps/s through the C++ host layer with the device-side slice sweeps of
    round 6 (marginal rate: a call with 1 + n sweeps against a call with 1), f32, synthetic states."""
    from peps_amd import hostapi, fermion
    hostapi.set_device(device)
    out = {}

    def amp_rate(ctx, batches):
        ctx.set_configs(batches[0]); ctx.evaluate_amplitude()
        ctx.sync(); t0 = time.perf_counter()
        for b in batches[1:]:
            ctx.set_configs(b); ctx.evaluate_amplitude()
        ctx.sync()
        return sum(len(b) for b in batches[1:]) / (time.perf_counter() - t0)

    def sweep_rate(call, n_sweeps):
        call(1)                                   # untimed: kernels loaded, allocator warm
        t0 = time.perf_counter(); call(1); t1 = time.perf_counter() - t0
        t0 = time.perf_counter(); rates = call(1 + n_sweeps); tn = time.perf_counter() - t0
        t0 = time.perf_counter(); call(1); t1 = min(t1, time.perf_counter() - t0)
        return max(tn - t1, 1e-9), rates

    for name, upd, nw in (("C2", "fullspace", 8192), ("C3", "exchange", 8192)):   # (walkers x candidates <= 65535: the grid z limit)
        try:
            L, D, chi, model = synthetic.CONFIGS[name]
            flat = synthetic.sitps_to_flat(synthetic.make_sitps(L, D), D, np.float64)
            mk = lambda k: synthetic.make_configs(L, nw, "heisenberg" if model == "heisenberg" else "tfim", seed0=91000 + 7 * k)
            c = capi.Context(L, L, D, 2, chi, dtype=capi.F32, device=device, max_walkers=nw)
            c.state_upload(flat)
            r = {"amp_per_s": amp_rate(c, [mk(k) for k in range(3)]), "walkers": nw, "dtype": "f32",
                 "workload": "%dx%d %s, D = %d, chi = %d, synthetic state (noise 0.1)" % (L, L, model, D, chi)}
            c.close()
            cfgs, seeds, ns = mk(0), np.arange(nw, dtype=np.uint64) + 300, 3
            dt_s, rates = sweep_rate(lambda n: hostapi.mc_sweeps(flat, cfgs, seeds, chi, upd, n, 0)[2], ns)
            r.update(mc_sweeps_per_s=ns * nw / dt_s, sweeps_timed=ns, accept_rate=float(np.mean(rates)),
                     updater="MCUpdateSquareNNFullSpaceUpdateOBC (Suwa-Todo over 4 states per bond on the device)" if upd == "fullspace"
                     else "MCUpdateSquareNNExchangeOBC")
            out[name] = r
        except Exception as e:
            out[name] = {"error": repr(e)}
    try:
        l5, d5, chi5, nw = 8, 6, 24, 4096
        st = fermion.random_even_state(l5, l5, d5, seed=11)
        rng = np.random.default_rng(1)
        cfgs = np.stack([rng.permutation(np.r_[np.zeros(32, dtype=int), np.ones(32, dtype=int)]).reshape(l5, l5) for _ in range(nw)])
        seeds, ns = np.arange(nw, dtype=np.uint64) + 700, 3
        dt_s, rates = sweep_rate(lambda n: hostapi.fermion_mc_sweeps(st, cfgs, seeds, chi5, n, 0)[2], ns)
        out["C5"] = {"mc_sweeps_per_s": ns * nw / dt_s, "sweeps_timed": ns, "walkers": nw, "dtype": "f32", "accept_rate": float(np.mean(rates)),
                     "updater": "MCUpdateSquareNNExchangeOBC on the fermionic state (tabulated exchange of the extended states on the device)",
                     "workload": "8x8 spinless fermions (fZ2-graded, sign-decorated), D = 6, chi = 24; amplitudes/s: other_modes.C5_spinless_tV_8x8_D6_chi24"}
    except Exception as e:
        out["C5"] = {"error": repr(e)}
    return out


def n1_latency(leg, reps=3):
    """latency floor: one walker, one fresh EvaluateAmplitude (what a reference-style one-walker-per-call binding pays)"""
    c = leg.capi.Context(leg.L, leg.L, leg.D, leg.pdim, leg.chi, dtype=leg.dt, device=leg.device, max_walkers=1)
    c.state_upload(leg.flat)
    ts = []
    for r in range(reps + 1):
        c.set_configs(leg.batches[0][r:r + 1])
        c.sync()
        t0 = time.perf_counter()
        c.evaluate_amplitude()
        ts.append(time.perf_counter() - t0)
    c.close()
    return float(np.median(ts[1:]) * 1e3)


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


# exactly one JSON line per run: the main thread and the collective watchdog agree through these
emit_lock = threading.Lock()
emitted = [False]


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

    # per-rank step time (stragglers show in the first scaling record)
    def gather_ms(x):
        if dist is None:
            return [x]
        t = torch.zeros(world, dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        t[rank] = x
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return [float(v) for v in t.tolist()]
    rank_ms = gather_ms(leg.local_elapsed / args.steps * 1e3)

    NOTE = ("roofline = the dominant single kernel of the step (largest HIP-event time among the profiling categories that are one "
            "kernel; the event pair brackets exactly that kernel's launch on the launch stream).  bound: the launches contract "
            "2*I*J*K flops over each walker's live extents and move (I*K + K*J + I*J) live elements (both counted on the device); "
            "their ratio against the machine balance decides whether the roofline is priced in bytes (hbm) or flops (mfma).  "
            "hbm: achieved = MEASURED HBM bytes per launch (traffic: rocprofv3 --pmc FETCH_SIZE (doubled for gfx950) + WRITE_SIZE "
            "of the committed passes, quoted only when this run has the launch count of the profiled run) / average launch "
            "duration; without matching passes the device-counted bytes are used (frac_priced_with says which; the counted "
            "figure is always under `counted`).  DESIGN.md section 6.")
    out = None
    pend_energy = {}
    leg_tag = ("%s_%s_%s" % (args.workload, args.dtype, args.state if args.state == "real" else ("noise%g" % args.noise))).lower()
    if rank == 0:
        n_amp = nw * args.steps * world
        value = n_amp / elapsed
        dom, roof = roofline_of(prof, args.dtype, args.steps, leg_tag, nw)
        roof["note"] = NOTE
        out = {
            "metric": "configuration-amplitudes/sec",
            "value": value,
            "unit": "amplitudes/s",
            "n_gpus": world,
            "ranks_reported_by_backend": dist.get_world_size() if dist else 1,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "ms_per_step_by_rank": {"min": min(rank_ms), "max": max(rank_ms), "all": [round(x, 3) for x in rank_ms]},
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
                "note": ("the SURVEY 8(d) synthetic state (rank-1 background + 0.1 N(0,1)) is numerically of LOW rank (workload_rank: "
                         "~10 live carry rows of 256): the headline value is the throughput on that state; `real_rank` below is the "
                         "same shapes on the reference's optimised D = 8 state tiled to 12x12 (the regime a VMC user runs in), "
                         "`full_rank` on i.i.d. random site tensors") if args.state == "synthetic" and args.noise <= 0.15 else None,
            },
            "roofline": roof,
            "reference_algorithm_equivalent_tflops": value * fl["total"] / 1e12 / world,
            "reference_algorithm_equivalent_note": ("value x flops of the REFERENCE algorithm per amplitude (SURVEY 8d: explicit-Q QR + gesdd on the "
                                                    "dense shapes); a speed-up-equivalent, NOT a fraction of this machine's peak: the Q-less, "
                                                    "rank-adaptive path executes far fewer flops (mfma.mfma_tflops_over_whole_step)"),
            "kernel_ms": {k: round(v["ms"], 3) for k, v in prof.items() if v["launches"]},
            "launches_per_step": {k: v["launches"] / args.steps for k, v in prof.items() if v["launches"]},
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
        if world == 1 and not args.no_latency:
            try:
                out["n1_ms"] = n1_latency(leg)
            except Exception as e:
                out["n1_ms"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(leg.flat, leg.batches[0][:512], chi, args.cpu_seconds)
            amps = cb.pop("amps")
            ac, sg = cb["all_cores"], cb["single"]
            out["cpu_baseline"] = {
                "value": ac["value"], "unit": "amplitudes/s", "cores": ac["threads"], "kind": "port",
                "sample": "%d configuration(s) of the same %s workload, one walker per process on %d core(s), BLAS threads = 1 "
                          "(the reference's execution model, monte_carlo_engine.h:97-98), through oracle/cbmps.c: plain C on "
                          "LAPACK dgelqf/dorglq/dgesdd/dgemm, float64, op-for-op restatement of bmps_impl.h:756-862,225-263; "
                          "the upstream binary cannot be built here"
                          % (ac["n"], args.workload, ac["threads"]),
                "single_thread": {"value": sg["value"], "cores": 1, "n": 1, "seconds": sg["seconds"],
                                  "sample": "one walker on one thread (one reference MPI rank)"},
                "oversubscribed_2x": cb.get("oversubscribed_2x"),
                "all_cores_detail": {k: ac[k] for k in ("seconds", "per_process_seconds", "walkers_per_process")},
                "scaling_all_cores_over_single": ac["value"] / sg["value"],
                "host": cb["host"],
                "host_cores": cb["cores"],
                "host_logical_cpus": cb["logical_cpus"],
                "note": ("cores = CPUs this job may use = min(affinity mask, cgroup cpu.max quota / period) -- the GPU box exposes 256 "
                         "logical CPUs to a container whose quota is 16 CPUs' worth of time, so 256 processes run no faster than 16 "
                         "(round 2 quoted that oversubscribed figure as '256 cores').  per_process_seconds = busy time of each "
                         "single-threaded worker for its walkers_per_process walkers; against single_thread.seconds it shows how much "
                         "slower one LAPACK walker runs when every usable core runs one (shared L3 / DRAM bandwidth)"),
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
            pend_energy["main"] = ((L, chi, 0 if dt == capi.F32 else 1, local_rank, leg.flat), leg.batches[0][:args.energy_n])
        if world == 1 and not args.no_sweeps and not fermionic:
            try:
                out["vmc"] = vmc_rates(leg, min(args.sweep_walkers, nw), args.sweep_count)
                out["mc_sweeps_per_s"] = out["vmc"]["mc_sweeps_per_s"]
                out["vmc_samples_per_s"] = out["vmc"]["vmc_samples_per_s"]
            except Exception as e:
                out["vmc"] = {"error": repr(e)}
    leg.close()
    del leg
    if rank == 0 and world == 1 and not args.no_other_modes and args.workload == "C4" and args.state == "synthetic":
        try:
            out["other_modes"] = other_modes(capi, synthetic, local_rank, L, D, chi)
        except Exception as e:
            out["other_modes"] = {"error": repr(e)}
        try:
            out["baseline_configs"] = baseline_config_rates(capi, synthetic, local_rank)
        except Exception as e:
            out["baseline_configs"] = {"error": repr(e)}

    # ---- further legs on the same shapes: a state of full rank (i.i.d. random site tensors) and a state of the rank of a
    #      REAL PEPS (the reference's optimised 4x4 D=8 fixture tiled to L x L, configurations near the Neel state) ----
    def extra_leg(name, real, noise, walkers, steps):
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
                oracle_sample = None
                tag = ("%s_%s_%s" % (args.workload, args.dtype, "real" if real else ("noise%g" % noise))).lower()
                fdom, froof = roofline_of(fprof, args.dtype, steps, tag, fnw)
                fr = {"value": fnw * steps * world / fel, "unit": "amplitudes/s", "walkers_per_gpu": fnw,
                      "steps": steps, "warmup": 1, "ms_per_step": fel / steps * 1e3,
                      "state": ("reference fixture tps_square_heisenberg4x4D8Double tiled by position class to %dx%d, "
                                "configurations = checkerboard + %d random NN exchanges" % (L, L, L * L // 8)) if real
                               else "i.i.d. random site tensors (synthetic noise 1.0)",
                      "roofline": froof,
                      "kernel_ms": {k: round(v["ms"], 3) for k, v in fprof.items() if v["launches"]},
                      "launches_per_step": {k: v["launches"] / steps for k, v in fprof.items() if v["launches"]},
                      "mfma": mfma_summary(fprof, args.dtype, fel),
                      "walkers_with_vanishing_amplitude": fnz, "workload_rank": fleg.rank_diagnostics()}
                if not real:
                    fr["synthetic_noise"] = noise
                if world == 1 and not args.no_cpu_baseline:
                    from oracle import cbmps
                    k = min(32 if real else 8, len(fleg.batches[0]))
                    ra, _, _ = cbmps.amplitudes_multiprocess(fleg.flat, fleg.batches[0][:k], chi, min(k, 16))
                    oracle_sample = ra
                    rel = np.abs(fleg.amps_first[:k] / ra - 1)
                    fr["parity_on_sample"] = {"max_rel_err_amplitude": float(np.max(rel)), "median_rel_err_amplitude": float(np.median(rel)),
                                              "n": int(k), "tolerance": 1e-5, "checker": "oracle/cbmps.c (float64)"}
                    if real:
                        fr["parity_on_sample"]["note"] = ("SURVEY 8(d) gate 1e-5 (round 3 stated 1e-4 here: the f32 accumulation of Y = Tt V^T "
                                                          "carried a common-mode 1.5e-5 on this periodic state; DESIGN 3e has the budget by stage)")
                    if not args.no_energy_check:
                        pend_energy[name] = ((L, chi, 0 if dt == capi.F32 else 1, local_rank, fleg.flat), fleg.batches[0][:max(2, args.energy_n // 2)])
                if real and world == 1 and not args.no_other_modes and dt == capi.F32:
                    # the reference's own arithmetic on the realistic state: the f64 device mode (round 6: oversampled subspace from a pivoted
                    # row selection + float64 Gram-Schmidt + one step of subspace iteration, Jacobi on Z = U M in LDS; round 5: 297 amp/s), 2 048 walkers
                    try:
                        n64 = min(2048, fnw)
                        c64 = capi.Context(L, L, D, 2, chi, dtype=capi.F64, device=local_rank, max_walkers=n64)
                        c64.state_upload(fleg.flat)
                        c64.set_configs(fleg.batches[0][:64]); c64.evaluate_amplitude(); c64.sync()
                        t0 = time.perf_counter()
                        c64.set_configs(fleg.batches[0][:n64]); a64 = c64.evaluate_amplitude(); c64.sync()
                        t64 = time.perf_counter() - t0
                        c64.close()
                        rel = np.abs(fleg.amps_first[:n64] / a64 - 1)
                        fr["f64_mode"] = {"amp_per_s": n64 / t64, "walkers": n64,
                                          "parity_on_sample": ({"max_rel_err_amplitude": float(np.max(np.abs(a64[:len(oracle_sample)] / oracle_sample - 1))),
                                                                "n": int(len(oracle_sample)), "tolerance": 5e-8, "checker": "oracle/cbmps.c (float64)",
                                                                "note": "the f64 mode is bounded by its Gram-based FORWARD factor (carry resolved to ~2e-7 relative, "
                                                                        "HISTORY 8 'f64 mode'), not by the dense truncation route: with the route and with the general "
                                                                        "kernels the same configurations give the same error (scripts/f64_route_parity.py: 5.19e-9 / "
                                                                        "5.17e-9 and 7.601e-9 / 7.609e-9 max over 64); tests/test_gpu_realrank.py asserts 1e-8 on its sample"}
                                                               if oracle_sample is not None else None),
                                          "profile": mode_profile("f64_real"),
                                          "f32_vs_f64_amplitude": {"max_rel": float(np.max(rel)), "median_rel": float(np.median(rel)),
                                                                   "p99_rel": float(np.percentile(rel, 99)),
                                                                   "share_above_1e-5": float(np.mean(rel > 1e-5)), "n": int(n64)}}
                    except Exception as e:
                        fr["f64_mode"] = {"error": repr(e)}
                if real and world == 1 and not args.no_other_modes and dt == capi.F32:
                    # the complex element type on the same state (a random phase on every tensor element): the dense route of
                    # Engine<cplx<double>> (round 6: randomised range finder + subspace iteration; round 5: 82 amp/s; round 4: 6.6)
                    try:
                        nc = min(512, fnw)
                        cflat = fleg.flat * np.exp(2j * np.pi * np.random.default_rng(5).uniform(size=fleg.flat.shape))
                        cc = capi.Context(L, L, D, 2, chi, dtype=capi.C128, device=local_rank, max_walkers=nc)
                        cc.state_upload(cflat)
                        cc.set_configs(fleg.batches[0][:32]); cc.evaluate_amplitude(); cc.sync()
                        t0 = time.perf_counter()
                        cc.set_configs(fleg.batches[0][:nc]); cc.evaluate_amplitude(); cc.sync()
                        fr["complex128"] = {"amp_per_s": nc / (time.perf_counter() - t0), "walkers": nc,
                                            "flagged_walkers": int(np.sum(cc.walker_flags() != 0)), "profile": mode_profile("c128_real")}
                        cc.close()
                    except Exception as e:
                        fr["complex128"] = {"error": repr(e)}
                if real and world == 1 and not args.no_sweeps:
                    try:       # what VMC consumes, on the state a VMC user has (fewer walkers: a sweep costs ~4 amplitudes)
                        fr["vmc"] = vmc_rates(fleg, min(args.real_sweep_walkers, fnw), args.real_sweep_count)
                    except Exception as e:
                        fr["vmc"] = {"error": repr(e)}
        except Exception as e:
            fr = {"error": repr(e)}
        fleg.close()
        return fr

    if not args.no_full_rank and args.noise < 0.5 and not fermionic and args.state == "synthetic":
        fr = extra_leg("full_rank", False, 1.0, args.full_rank_walkers, args.full_rank_steps)
        if rank == 0:
            if fr is not None and "error" not in fr:
                fr["real_state_rank"] = real_state_rank(capi, local_rank, dt)
            out["full_rank"] = fr
    if not args.no_real_rank and args.noise < 0.5 and not fermionic and args.state == "synthetic" and D == 8 and os.path.isdir(REAL_STATE):
        rr = extra_leg("real_rank", True, 0.0, args.real_rank_walkers, args.real_rank_steps)
        if rank == 0:
            out["real_rank"] = rr

    # ---- the exchange step, outside the timed region: all-reduce of the HBM-resident accumulators and the state broadcast
    #      over RCCL.  Phased: after every phase the ranks agree (all-reduce MAX of a failure flag) whether to go on, so that a
    #      rank-local failure never leaves the others waiting inside a collective ----
    if dist is not None and backend == "nccl":
        coll = {}
        st = {}
        # Watchdog: the measured line must survive a collective that never returns (RCCL has not seen N > 1 ranks of this code
        # on hardware before the driver's scaling run): after 240 s rank 0 prints the line with the failure noted and every
        # rank leaves with os._exit (a blocked RCCL call cannot be interrupted from Python).
        # The run then ends with a NON-ZERO status on every rank (a job whose collective never completed is not a success;
        # nothing is re-executed from this GPU-initialised process), and exactly one of the watchdog and the main thread emits
        # the JSON line (emit_lock / emitted).
        coll_done = threading.Event()

        def watchdog():
            if not coll_done.wait(timeout=240.0):
                with emit_lock:
                    if rank == 0 and not emitted[0]:
                        emitted[0] = True
                        line = dict(out)
                        line["collective"] = {"error": "collective section did not finish within 240 s (watchdog); the timed legs above are unaffected"}
                        print(json.dumps(line), flush=True)
                    os._exit(3)

        threading.Thread(target=watchdog, daemon=True).start()

        def phase(fn):
            err = None
            try:
                fn()
            except Exception as e:
                err = repr(e)
            bad = max_over_ranks(1.0 if err else 0.0) > 0
            if bad:
                coll["error"] = err or coll.get("error") or "another rank failed"
            return not bad

        def p_setup():
            from peps_amd import dist as pdist
            st["pdist"] = pdist
            st["ctx"] = capi.Context(L, L, D, 2, chi, dtype=dt, device=local_rank, max_walkers=16)
            st["ctx"].grad_reset()
            so, seo, n = st["ctx"].grad_device_ptr()
            st["ctx"].sync()
            st["n"] = n
            st["ts"] = [pdist.device_tensor(p, n) for p in (so, seo)]
            for t in st["ts"]:
                t.fill_(1.0)
            torch.cuda.synchronize()

        def p_torch():
            dist.barrier()
            t0 = time.perf_counter()
            for t in st["ts"]:
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
            torch.cuda.synchronize()
            st["t_torch"] = time.perf_counter() - t0
            st["ok"] = bool(abs(float(st["ts"][0][0].item()) - world) < 1e-12 and abs(float(st["ts"][1][-1].item()) - world) < 1e-12)

        def p_comm():
            st["pdist"].comm_init(st["ctx"])

        def p_lib():
            cctx = st["ctx"]
            cctx.grad_allreduce()          # first call builds the rings
            dist.barrier()
            t0 = time.perf_counter()
            cctx.grad_allreduce()
            st["t_lib"] = time.perf_counter() - t0
            # parameter broadcast: rank 0 uploads a state, one ncclBroadcast puts it into every rank's HBM state buffer
            flat0 = synthetic.sitps_to_flat(synthetic.make_sitps(L, D, noise=args.noise), D, np.float64)
            if rank == 0:
                cctx.state_upload(flat0)
            cctx.bcast_state(0)            # warm
            dist.barrier()
            t0 = time.perf_counter()
            cctx.bcast_state(0)
            st["t_bcast"] = time.perf_counter() - t0
            cctx.set_configs(synthetic.checkerboard(L)[None])
            a_b = float(cctx.evaluate_amplitude()[0])
            cctx.state_upload(flat0)
            a_u = float(cctx.evaluate_amplitude()[0])
            st["bcast_ok"] = bool(a_b == a_u)

        if phase(p_setup) and phase(p_torch) and phase(p_comm) and phase(p_lib):
            ok_all = max_over_ranks(0.0 if (st["ok"] and st["bcast_ok"]) else 1.0) == 0.0
            coll = {"what": "all-reduce(sum) of the device-resident gradient accumulators S_O, S_EO (float64), in place in HBM; "
                            "ncclBroadcast of the flat SITPS from rank 0 (pepsgpu_bcast_state)",
                    "bytes": int(2 * st["n"] * 8), "ranks": dist.get_world_size(), "result_correct": bool(ok_all),
                    "ms_torch_distributed_nccl": st["t_torch"] * 1e3, "ms_pepsgpu_grad_allreduce": st["t_lib"] * 1e3,
                    "ms_pepsgpu_bcast_state": st["t_bcast"] * 1e3, "library_comm_size": st["ctx"].comm_size()}
        try:
            if "ctx" in st:
                st["ctx"].close()
        except Exception:
            pass
        coll_done.set()
        if rank == 0:
            out["collective"] = coll

    if rank == 0:
        # E_loc parity of every leg: the oracle pools of all legs run side by side now that no timed region is left
        started = {}
        for name, (info, cfgs) in pend_energy.items():
            try:
                started[name] = (info, energy_parity_start(info[4], cfgs, info[1]))
            except Exception as e:
                started[name] = (info, {"error": repr(e)})
        for name, (info, pend) in started.items():
            try:
                ep = pend if "error" in pend else energy_parity_finish(info, pend, 900.0)
            except Exception as e:
                ep = {"error": repr(e)}
            if name == "main":
                out["energy_parity"] = ep
                out["energy_rel_err"] = ep.get("max_rel_err_energy")
            elif isinstance(out.get(name), dict):
                out[name]["energy_parity"] = ep
        with emit_lock:
            if not emitted[0]:
                emitted[0] = True
                print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
