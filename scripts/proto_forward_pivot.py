"""Round 6 prototype (CPU, checker side only): how many rows the forward factor R^T R = P^T P needs.  Captures the R factors of the
oracle's forward QR (bmps_impl.h:821) at the bulk sites of the tiled real state at C4 (R^T R is the gauge-invariant left norm matrix of
the untruncated product MPS = the device's Gram of P), rounds to the device's data (f32 storage of P: noise added at eps32 of the column
norms) and factors G three ways with the device's pivot threshold (8 eps32)^2 max diag: natural column order with dropped pivots
(chol_blocked_kernel), diagonal pivoting, diagonal pivoting in panels of 16 (the 16 largest remaining diagonals per panel)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from peps_amd import hostapi, synthetic
from oracle import vmc, tensor as T
from oracle.bmps import BMPSTruncateParams
L, D, chi = 12, 8, 32
cache = "/tmp/proto/Rfwd.npz"
if not os.path.exists(cache):
    flat = synthetic.tile_flat_state(hostapi.load_sitps(os.path.join(ROOT, "tests/golden/ref_fixtures", synthetic.REAL_FIXTURE), 8), L)
    sitps = synthetic.flat_to_sitps(flat)
    cfgs = synthetic.make_configs_near_neel(L, 1, seed0=307)
    caught = []
    orig = T.qr
    def hook(a, ldims):
        q, r = orig(a, ldims)
        r2 = r.reshape(r.shape[0], -1)
        if r2.shape == (256, 256):
            caught.append(r2.copy())
        return q, r
    T.qr = hook
    comp = vmc.TPSWaveFunctionComponent(sitps, cfgs[0], BMPSTruncateParams.SVD(chi, chi, 0.0))
    print("amp", comp.amplitude, "blocks", len(caught))
    np.savez_compressed(cache, R=np.stack(caught))
Rs = np.load(cache)["R"]
eps32 = 2.0 ** -24
TH = (8 * eps32) ** 2


def natural(G):
    n = G.shape[0]; G = G.copy(); th = TH * G.diagonal().max(); rows = []
    for j in range(n):
        if G[j, j] > th:
            r = G[j, :] / np.sqrt(G[j, j]); r[:j] = 0
            rows.append(r); G -= np.outer(r, r)
    return np.array(rows)


def pivoted(G, nb=1):
    n = G.shape[0]; G = G.copy(); th = TH * G.diagonal().max(); rows = []; done = np.zeros(n, bool)
    while True:
        d = np.where(done, -1, G.diagonal())
        sel = np.argsort(-d)[:nb]
        sel = [j for j in sel if d[j] > th]
        if not sel:
            break
        for j in sel:
            done[j] = True
            if G[j, j] > th:
                r = G[j, :] / np.sqrt(G[j, j]); r[done & (np.arange(n) != j)] *= 1  # residual columns of finished pivots are ~0 already
                rows.append(r); G -= np.outer(r, r)
    return np.array(rows)


rng = np.random.default_rng(1)
res = []
for R in Rs[::3]:
    # device data: P has f32 rounding -> G = (R + E)^T (R + E), |E_ij| ~ eps32/2 |R_ij| (a stand-in: the rounding lives on P, not on R)
    Rn = R * (1 + rng.uniform(-1, 1, R.shape) * 2.0 ** -24)
    G = Rn.T @ Rn
    s = np.linalg.svd(R, compute_uv=False)
    k_floor = int(np.sum(s > 4.8e-7 * s[0]))
    a, b, c = natural(G), pivoted(G, 1), pivoted(G, 16)
    err = [np.linalg.norm(x.T @ x - G) / np.linalg.norm(G) for x in (a, b, c)]
    res.append((len(a), len(b), len(c), k_floor))
    print("natural %3d  pivoted %3d  panels-of-16 %3d   sigma > 4.8e-7: %3d   |R^T R - G|/|G| %.1e %.1e %.1e" % (len(a), len(b), len(c), k_floor, *err))
res = np.array(res)
print("median rows: natural %d, pivoted %d, panels of 16 %d" % tuple(np.median(res[:, :3], axis=0)))

print("\nlost energy trace(G - R^T R) / trace(G) and rows, by threshold factor c in (c eps32)^2:")
for c in (8, 16, 32, 64):
    TH = (c * eps32) ** 2
    out = []
    for R in Rs[::6]:
        Rn = R * (1 + rng.uniform(-1, 1, R.shape) * 2.0 ** -24)
        G = Rn.T @ Rn
        a, b = natural(G), pivoted(G, 16)
        out.append((len(a), len(b), (np.trace(G) - np.sum(a * a)) / np.trace(G), (np.trace(G) - np.sum(b * b)) / np.trace(G)))
    out = np.array(out)
    print("c = %2d: natural rows %3d lost %.1e (max %.1e) | panels-of-16 pivoted rows %3d lost %.1e (max %.1e)" % (
        c, np.median(out[:, 0]), np.median(out[:, 2]), out[:, 2].max(), np.median(out[:, 1]), np.median(out[:, 3]), out[:, 3].max()))
