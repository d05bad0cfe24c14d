"""E_loc samples/s of the spinless-fermion model at C5 shapes (8x8, D = 6, chi = 24) through the C++ host layer: t2 = 0, t2 != 0 with the
environments of the row pass (twisted BTen2 sets, round 5) and -- PEPSHOST_NNN_FRESH=1 in the environment -- with one fresh batched
contraction per diagonal (rounds 2-4).   usage: python scripts/bench_fermion_nnn.py [walkers] [f32|f64]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from peps_amd import fermion, hostapi, synthetic
nw = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dt = 1 if (len(sys.argv) > 2 and sys.argv[2] == "f64") else 0
L, D, chi = 8, 6, 24
st = fermion.random_even_state(L, L, D, seed=11)
cfgs = synthetic.make_configs(L, nw, "heisenberg", seed0=80000)
res = {"walkers": nw, "dtype": "f64" if dt else "f32", "nnn": "fresh" if os.environ.get("PEPSHOST_NNN_FRESH") else "local"}
for name, t2 in (("t2=0", 0.0), ("t2=0.7", 0.7)):
    hostapi.fermion_energy(st, cfgs[:min(nw, 32)], chi, 1.0, 1.0, dt, t2=t2)
    t0 = time.perf_counter()
    a, e, psi = hostapi.fermion_energy(st, cfgs, chi, 1.0, 1.0, dt, t2=t2)
    sec = time.perf_counter() - t0
    res[name] = {"samples_per_s": nw / sec, "seconds": sec, "e_mean": float(np.mean(e))}
print(json.dumps(res))
