"""Golden vector for BASELINE config C1 (4x4 transverse-field Ising, D=2, chi=4, exact summation over
all 2^16 configurations): energy, |grad|, sum of weights and a few amplitudes from the float64 oracle
on the synthetic state of SURVEY 8(d).  Output: tests/golden/c1_exact_sum.json (committed).
Run: python scripts/make_c1_golden.py   (8 processes, ~1-2 minutes)"""
import json, os, sys
import numpy as np
from multiprocessing import Pool
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from oracle import vmc
from oracle.bmps import BMPSTruncateParams
from peps_amd import synthetic

L, D, CHI, H = 4, 2, 4, 3.0
NP = 8


def all_configs():
    n = L * L
    idx = np.arange(1 << n, dtype=np.int64)
    return ((idx[:, None] >> np.arange(n)[None, :]) & 1).astype(np.int32).reshape(-1, L, L)


def part(rank):
    os.environ["OMP_NUM_THREADS"] = "1"
    sitps = synthetic.make_sitps(L, D)
    so, seo, w, we = vmc.exact_sum_partials(sitps, all_configs(), BMPSTruncateParams.SVD(CHI, CHI, 0.0),
                                            vmc.TransverseFieldIsingSquareOBC(H), rank, NP)
    return so, seo, w, we


if __name__ == "__main__":
    with Pool(NP) as p:
        parts = p.map(part, range(NP))
    so, seo, w, we = parts[0]
    for q in parts[1:]:
        for r in range(L):
            for c in range(L):
                for s in range(2):
                    so[r][c][s] = so[r][c][s] + q[0][r][c][s]
                    seo[r][c][s] = seo[r][c][s] + q[1][r][c][s]
        w += q[2]; we += q[3]
    e, grad, wsum = vmc.finish_exact_sum(so, seo, w, we)
    g2 = sum(float(np.sum(np.abs(grad[r][c][s]) ** 2)) for r in range(L) for c in range(L) for s in range(2))
    sitps = synthetic.make_sitps(L, D)
    cfgs = all_configs()
    probe = [0, 1, 12345, 43690, 65535]
    tp = BMPSTruncateParams.SVD(CHI, CHI, 0.0)
    amps = [float(vmc.TPSWaveFunctionComponent(sitps, cfgs[i], tp).amplitude) for i in probe]
    out = {"workload": "C1: 4x4 TFIM h=%.1f, D=2, chi=4, synthetic state make_sitps(4,2), all 2^16 configurations" % H,
           "h": H, "energy": float(e), "grad_norm2": g2, "weight_sum": float(wsum),
           "grad_site00_s0": np.asarray(grad[0][0][0]).ravel().tolist(),
           "probe_config_index": probe, "probe_amplitudes": amps}
    path = os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "c1_exact_sum.json")
    json.dump(out, open(path, "w"), indent=1)
    print(out["energy"], g2, wsum)
