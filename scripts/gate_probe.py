"""Round 6, VERDICT r05 item 4: which walkers carry the tail of the f32 amplitude error on the tiled real state at C4, can |psi| / median
predict them, and what a re-evaluation with float64 accumulation in every contraction of the f32 engine (PEPSGPU_ACC64=15) gives.
usage: python scripts/gate_probe.py run <tag> [n] [seed]   (writes gpurun_out/r06/gate_<tag>.npy; dtype f64 when the tag starts with f64)
       python scripts/gate_probe.py analyse"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "gpurun_out", "r06")
if sys.argv[1] == "run":
    from peps_amd import capi, hostapi, synthetic
    tag = sys.argv[2]
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
    seed = int(sys.argv[4]) if len(sys.argv) > 4 else 100000
    L, D, chi = 12, 8, 32
    flat = synthetic.tile_flat_state(hostapi.load_sitps(os.path.join(ROOT, "tests/golden/ref_fixtures", synthetic.REAL_FIXTURE), 8), L)
    ctx = capi.Context(L, L, D, 2, 4 * D, dtype=capi.F64, max_walkers=1)
    ctx.state_upload(flat); ctx.set_configs(synthetic.checkerboard(L)[None])
    flat = flat * abs(float(ctx.evaluate_amplitude()[0])) ** (-1.0 / (L * L)); ctx.close()
    cfgs = synthetic.make_configs_near_neel(L, n, seed0=seed)
    c = capi.Context(L, L, D, 2, chi, dtype=capi.F64 if tag.startswith("f64") else capi.F32, max_walkers=n)
    c.state_upload(flat); c.set_configs(cfgs[:min(n, 64)]); c.evaluate_amplitude(); c.sync()
    t0 = time.perf_counter(); c.set_configs(cfgs); a = c.evaluate_amplitude(); c.sync(); dt = time.perf_counter() - t0
    np.save(os.path.join(OUT, "gate_%s.npy" % tag), a)
    print(tag, n, "walkers", round(dt, 3), "s")
else:
    ref = np.load(os.path.join(OUT, "gate_f64.npy"))
    med = np.median(np.abs(ref))
    res = {}
    for tag in ("f32", "f32acc"):
        p = os.path.join(OUT, "gate_%s.npy" % tag)
        if not os.path.exists(p): continue
        a = np.load(p)
        rel = np.abs(a / ref - 1)
        ratio = np.abs(a) / np.median(np.abs(a))
        r = {"median": float(np.median(rel)), "p99": float(np.percentile(rel, 99)), "max": float(rel.max()), "n_above_1e-5": int(np.sum(rel > 1e-5)),
             "n_above_5e-6": int(np.sum(rel > 5e-6))}
        for thr in (0.5, 0.25, 0.1, 0.05):
            fl = ratio < thr
            r["ratio<%g" % thr] = {"flagged": int(fl.sum()), "max_rel_unflagged": float(rel[~fl].max()), "n>5e-6 unflagged": int(np.sum(rel[~fl] > 5e-6))}
        order = np.argsort(-rel)[:8]
        r["worst"] = [{"rel": float(rel[i]), "ratio": float(ratio[i])} for i in order]
        res[tag] = r
    print(json.dumps(res, indent=1))
