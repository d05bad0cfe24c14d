"""Round 6: what the f32 noise floors cost in parity and buy in speed on the tiled real state at C4.  The f32 engine with its floors
scaled (PEPSGPU_F32_EPS, set by the caller's environment) against the f64 mode on the same near-Neel configurations.
usage: python scripts/floor_probe.py run <tag> [n]   (dtype f64 when the tag starts with f64; writes gpurun_out/r06/floor_<tag>.npy + .json)
       python scripts/floor_probe.py analyse"""
import glob, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "gpurun_out", "r06")
NREF = 2048
if sys.argv[1] == "run":
    from peps_amd import capi, hostapi, synthetic
    tag = sys.argv[2]
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
    L, D, chi = 12, 8, 32
    flat = synthetic.tile_flat_state(hostapi.load_sitps(os.path.join(ROOT, "tests/golden/ref_fixtures", synthetic.REAL_FIXTURE), 8), L)
    ctx = capi.Context(L, L, D, 2, 4 * D, dtype=capi.F64, max_walkers=1)
    ctx.state_upload(flat); ctx.set_configs(synthetic.checkerboard(L)[None])
    flat = flat * abs(float(ctx.evaluate_amplitude()[0])) ** (-1.0 / (L * L)); ctx.close()
    cfgs = synthetic.make_configs_near_neel(L, n, seed0=100000)
    c = capi.Context(L, L, D, 2, chi, dtype=capi.F64 if tag.startswith("f64") else capi.F32, max_walkers=n)
    c.state_upload(flat); c.set_configs(cfgs); c.evaluate_amplitude(); c.sync()
    t0 = time.perf_counter(); c.set_configs(cfgs); a = c.evaluate_amplitude(); c.sync(); dt = time.perf_counter() - t0
    np.save(os.path.join(OUT, "floor_%s.npy" % tag), a[:NREF])
    st = c.stats() if hasattr(c, "stats") else {}
    json.dump({"tag": tag, "walkers": n, "amp_per_s": n / dt, "eps": os.environ.get("PEPSGPU_F32_EPS"), "flags": int(np.sum(c.walker_flags() != 0)),
               "stats": {k: v for k, v in st.items() if "carry" in k or "live" in k}}, open(os.path.join(OUT, "floor_%s.json" % tag), "w"))
    print(tag, n, "walkers", round(n / dt, 1), "amp/s")
else:
    ref = np.load(os.path.join(OUT, "floor_f64.npy"))
    for p in sorted(glob.glob(os.path.join(OUT, "floor_f32*.npy"))):
        a = np.load(p); m = json.load(open(p[:-4] + ".json"))
        rel = np.abs(a / ref[:len(a)] - 1)
        print("%-14s eps %-8s %7.1f amp/s  median %.2e  p99 %.2e  max %.2e  n>1e-5: %d  n>5e-6: %d  %s" % (
            m["tag"], m["eps"], m["amp_per_s"], np.median(rel), np.percentile(rel, 99), rel.max(), np.sum(rel > 1e-5), np.sum(rel > 5e-6), m["stats"]))
