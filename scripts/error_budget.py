"""f32 error budget of the amplitude by stage, on the tiled real state at C4 (or --L / --D / --chi).

Two families of runs against the float64 device mode (itself pinned to the oracle at 1e-9; --oracle N also runs oracle/cbmps.c
on the first N configurations):

  * the FLOAT64 engine with exactly one stored intermediate rounded to float32 where the float32 engine stores it
    (PEPSGPU_INJECT_F32 letters: S state, P = R (A x W), R carry, T = Tt, M = R Tt, V = Vt, Y, E environments / BTen), and / or
    with the noise floors of the float32 engine (PEPSGPU_F64_EPS): what the float32 STORAGE of each stage costs;
  * the FLOAT32 engine on its default route and with one route switch at a time (what the route itself costs).

Prints one JSON object; per run: median / max of |psi / psi_f64 - 1|, the mean SIGNED relative difference (a common-mode bias
shows there) and its standard error.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


WORKER = r"""
import json, sys, time
import numpy as np
sys.path.insert(0, sys.argv[1])
from peps_amd import capi
d = np.load(sys.argv[2])
flat, cfgs = d["flat"], d["cfgs"]
L, D, chi, dt = int(d["L"]), int(d["D"]), int(d["chi"]), int(d["dt"])
ctx = capi.Context(L, L, D, 2, chi, dtype=dt, max_walkers=len(cfgs))
ctx.state_upload(flat)
ctx.set_configs(cfgs)
t0 = time.perf_counter()
a = ctx.evaluate_amplitude()
sec = time.perf_counter() - t0
flags = int(np.sum(ctx.walker_flags() != 0))
ctx.close()
np.savez(sys.argv[3], a=a, sec=sec, flags=flags)
"""


def run(capi, flat, cfgs, L, D, chi, dt, env):
    """one variant = one process: the library reads its environment switches once (static const)"""
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        job, res = os.path.join(td, "job.npz"), os.path.join(td, "res.npz")
        np.savez(job, flat=flat, cfgs=cfgs, L=L, D=D, chi=chi, dt=dt)
        r = subprocess.run([sys.executable, "-c", WORKER, ROOT, job, res], env=dict(os.environ, **env), capture_output=True, text=True,
                           timeout=1200)
        if r.returncode != 0:
            raise RuntimeError(r.stderr[-1500:])
        d = np.load(res)
        return d["a"], float(d["sec"]), int(d["flags"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--L", type=int, default=12)
    ap.add_argument("--D", type=int, default=8)
    ap.add_argument("--chi", type=int, default=32)
    ap.add_argument("--walkers", type=int, default=64)
    ap.add_argument("--oracle", type=int, default=0)
    ap.add_argument("--state", default="real", choices=["real", "synthetic", "full"])
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    from peps_amd import capi, hostapi, synthetic
    L, D, chi = args.L, args.D, args.chi
    if args.state == "real":
        from conftest import FIXTURES
        f4 = hostapi.load_sitps(os.path.join(FIXTURES, synthetic.REAL_FIXTURE), 8)
        if D < 8:
            f4 = np.ascontiguousarray(f4[:, :, :, :D, :D, :D, :D])
        flat = synthetic.tile_flat_state(f4, L)
        ctx = capi.Context(L, L, D, 2, 4 * D, dtype=capi.F64, max_walkers=1)
        ctx.state_upload(flat)
        ctx.set_configs(synthetic.checkerboard(L)[None])
        psi = float(ctx.evaluate_amplitude()[0])
        ctx.close()
        flat = flat * abs(psi) ** (-1.0 / (L * L))
        cfgs = synthetic.make_configs_near_neel(L, args.walkers, seed0=307)
    else:
        flat = synthetic.sitps_to_flat(synthetic.make_sitps(L, D, noise=1.0 if args.state == "full" else 0.1), D)
        cfgs = synthetic.make_configs(L, args.walkers, "heisenberg")
    EPS32 = "5.9604645e-8"
    runs = [("f64", capi.F64, {})]
    for c in "SPRTMVYE":
        runs.append(("f64+round(%s)" % c, capi.F64, {"PEPSGPU_INJECT_F32": c}))
    runs += [
        ("f64+round(all)", capi.F64, {"PEPSGPU_INJECT_F32": "SPRTMVYE"}),
        ("f64+floors_f32", capi.F64, {"PEPSGPU_F64_EPS": EPS32}),
        ("f64+round(all)+floors_f32", capi.F64, {"PEPSGPU_INJECT_F32": "SPRTMVYE", "PEPSGPU_F64_EPS": EPS32}),
        ("f32", capi.F32, {}),
        ("f32 no ortho polish", capi.F32, {"PEPSGPU_ORTHO_POLISH": "0"}),
        ("f32 Y f32 chain (round 3)", capi.F32, {"PEPSGPU_Y_ACC64": "0"}),
        ("f32 round 3 (no ortho polish, Y f32 chain)", capi.F32, {"PEPSGPU_ORTHO_POLISH": "0", "PEPSGPU_Y_ACC64": "0"}),
        ("f32 acc64 X,P", capi.F32, {"PEPSGPU_ACC64": "1"}),
        ("f32 acc64 Z,Tt", capi.F32, {"PEPSGPU_ACC64": "2"}),
        ("f32 acc64 M", capi.F32, {"PEPSGPU_ACC64": "4"}),
        ("f32 acc64 Y", capi.F32, {"PEPSGPU_ACC64": "8"}),
        ("f32 acc64 all contractions", capi.F32, {"PEPSGPU_ACC64": "15"}),
        ("f32 backward pair in f32 (round 4)", capi.F32, {"PEPSGPU_TT_ACC64": "0"}),
        ("f32 Y on the wave-per-tile f64 body", capi.F32, {"PEPSGPU_Y_ACC64": "1"}),
        ("f32 acc64 X,P,Z,Tt", capi.F32, {"PEPSGPU_ACC64": "3"}),
        ("f32 acc64 Z,Tt,M", capi.F32, {"PEPSGPU_ACC64": "6"}),
        ("f32 acc64 X,P,Z,Tt,M", capi.F32, {"PEPSGPU_ACC64": "7"}),
        ("f32 acc64 all, no ortho polish", capi.F32, {"PEPSGPU_ACC64": "15", "PEPSGPU_ORTHO_POLISH": "0"}),
        ("f32 Y on the LDS-tiled f32 kernel", capi.F32, {"PEPSGPU_Y_TILED": "1"}),
        ("f32 no fused norm", capi.F32, {"PEPSGPU_NO_FUSED_NORM": "1"}),
        ("f32 no tt swap", capi.F32, {"PEPSGPU_NO_TT_SWAP": "1"}),
        ("f32 no vector loads", capi.F32, {"PEPSGPU_TGEMM_NOVEC": "1"}),
        ("f32 no hints/shrink", capi.F32, {"PEPSGPU_NO_RANK_HINT_SKIP": "1", "PEPSGPU_NO_BOND_SHRINK": "1"}),
        ("f32 no two-level", capi.F32, {"PEPSGPU_NO_TWO_LEVEL": "1"}),
        ("f32 no mid route (Jacobi on M)", capi.F32, {"PEPSGPU_NO_MIDROUTE": "1"}),
        ("f32 no mid route, no ortho polish", capi.F32, {"PEPSGPU_NO_MIDROUTE": "1", "PEPSGPU_ORTHO_POLISH": "0"}),
        ("f32 no chain", capi.F32, {"PEPSGPU_NO_CHAIN": "1"}),
        ("f32 no rank adapt", capi.F32, {"PEPSGPU_NO_RANK_ADAPT": "1"}),
        ("f32 Grams on the f64 matrix cores (no i8)", capi.F32, {"PEPSGPU_NO_I8_GRAM": "1"}),
        ("f32 i8 column Gram only", capi.F32, {"PEPSGPU_NO_I8_ROWGRAM": "1"}),
    ]
    if args.only:
        want = set(args.only.split(";"))
        runs = [r for r in runs if r[0] == "f64" or r[0] in want]
    out = {"L": L, "D": D, "chi": chi, "walkers": len(cfgs), "state": args.state, "runs": {}}
    ref = None
    for name, dt, env in runs:
        try:
            a, sec, flags = run(capi, flat, cfgs, L, D, chi, dt, env)
        except Exception as e:
            out["runs"][name] = {"error": repr(e)}
            continue
        if ref is None:
            ref = a
            out["runs"][name] = {"seconds": sec, "flags": flags}
            if args.oracle:
                from oracle import cbmps
                k = min(args.oracle, len(cfgs))
                ra, _, _ = cbmps.amplitudes_multiprocess(flat, cfgs[:k], chi, k)
                out["runs"][name]["vs_oracle_cbmps_max_rel"] = float(np.max(np.abs(a[:k] / ra - 1)))
                out["oracle_n"] = k
            continue
        rel = a / ref - 1
        out["runs"][name] = {"median": float(np.median(np.abs(rel))), "max": float(np.max(np.abs(rel))),
                             "mean_signed": float(np.mean(rel)), "stderr_signed": float(np.std(rel) / np.sqrt(len(rel))),
                             "seconds": sec, "flags": flags}
        print(name, json.dumps(out["runs"][name]), file=sys.stderr, flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
