"""tiny2 / tiny Jacobi with mixed live row counts inside one launch (two walkers per wave): singular values and
orthogonality of every walker against numpy"""
import sys
import numpy as np
sys.path.insert(0, '.')
from peps_amd import capi
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
nb, m, ln = 4096, 16, int(sys.argv[2]) if len(sys.argv) > 2 else 96
M = np.zeros((nb, m, ln), dtype=np.float32)
mm = rng.integers(1, m + 1, size=nb)
for b in range(nb):
    k = mm[b]
    # graded spectrum, triangular-like rows as the carry produces
    a = rng.standard_normal((k, ln)) * (10.0 ** (-0.6 * np.arange(k)))[:, None]
    M[b, :k] = a
out, Vt, S, sw = capi.diag_jacobi(capi.F32, M, m, force_global=3)
worst = 0.0; bad = []
for b in range(nb):
    k = mm[b]
    sv = np.linalg.svd(M[b, :k].astype(np.float64), compute_uv=False)
    rows = out[b, :k].astype(np.float64)
    nr = np.sort(np.linalg.norm(rows, axis=1))[::-1]
    e1 = np.max(np.abs(nr - sv)) / sv[0]
    g = rows @ rows.T
    d = np.sqrt(np.diag(g)) + 1e-300
    off = np.max(np.abs(g / d[:, None] / d[None, :] - np.eye(k)) * (np.minimum(d[:, None], d[None, :]) > 1e-5 * sv[0]))
    e = max(e1, off)
    if e > worst: worst = e
    if e > 1e-4: bad.append((b, int(k), int(mm[b ^ 1]), float(e1), float(off), int(sw[b])))
print("ln", ln, "worst", worst, "bad", bad[:10], len(bad))
