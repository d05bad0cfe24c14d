"""rank reported by the Householder and the Gram form of the low-rank factor on a matrix with a decaying spectrum"""
import sys, os
sys.path.insert(0, '.')
import numpy as np
from peps_amd import capi
rng = np.random.default_rng(0)
K, n, nb = 80, 80, 4
out = {}
P = np.zeros((nb, K, n), dtype=np.float32)
for b in range(nb):
    u, _ = np.linalg.qr(rng.standard_normal((K, K)))
    v, _ = np.linalg.qr(rng.standard_normal((n, n)))
    s = 10.0 ** (-0.5 * np.arange(K))          # sigma_k = 10^(-k/2)
    P[b] = ((u * s) @ v.T).astype(np.float32)
R, ml = capi.diag_gram_chol(capi.F32, P)
sv = np.linalg.svd(P[0].astype(np.float64), compute_uv=False)
print("mode", os.environ.get("PEPSGPU_NO_QR_FACTOR"), "noise", os.environ.get("PEPSGPU_QR_NOISE"), "ml", ml,
      "numpy rank(>4.8e-7)", int(np.sum(sv / sv[0] > 4.8e-7)))
for b in range(1):
    Rb = R[b][:max(ml[b], 0)].astype(np.float64)
    G = P[b].astype(np.float64).T @ P[b].astype(np.float64)
    sc = np.max(np.diag(G))
    print("  |R^T R - G/maxd| / 1 =", np.max(np.abs(Rb.T @ Rb - G / sc)), "row norms", np.round(np.log10(np.linalg.norm(Rb, axis=1) + 1e-300), 1))
