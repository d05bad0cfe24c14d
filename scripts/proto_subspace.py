"""Round 6 prototype (CPU, NumPy): numerics of the truncation on the dense route.  Emulates the device pipeline on the blocks of
scripts/proto_capture_M.py (f32 data, f64 Grams, Cholesky with dropped pivots, second compression) and compares, in the GRADED metric that
matters for the truncated MPS -- || (I - U U^T) U* S* || / s_1, the part of the kept M the subspace loses -- the full SVD of the small
factor B2 (what the one-sided Jacobi delivers) with a block subspace iteration + Rayleigh-Ritz on G2 = B B^T."""
import sys
import numpy as np
M_all = np.load("/tmp/proto/M.npz")["M"]
EPS32 = 2.0 ** -24 * 2   # 1.19e-7
chi = 32

def chol_drop(G, thr):
    """upper factor rows with dropped pivots: row j kept iff its pivot (Schur diagonal) >= thr * max diag"""
    n = G.shape[0]
    A = G.copy()
    dmax = np.max(np.diag(G))
    rows = []
    for j in range(n):
        piv = A[j, j]
        if piv < thr * dmax:
            continue
        r = A[j, :] / np.sqrt(piv)
        r[:j] = 0.0
        rows.append(r)
        A -= np.outer(r, r)
    return np.array(rows)

def graded_err(Usub, Ustar, s):
    """Usub: (k, m) orthonormal rows; Ustar (m, k) exact left vectors; s exact singular values (k,)"""
    Q, _ = np.linalg.qr(Usub.T)
    R = Ustar * s[None, :] - Q @ (Q.T @ (Ustar * s[None, :]))
    return np.linalg.norm(R, 2) / s[0], np.linalg.norm(R, axis=0) / s[0]

def pipeline(M, p=40, iters=2, start="unit", verbose=False, cholqr=False):
    M32 = (M / np.abs(M).max()).astype(np.float32).astype(np.float64)
    Us, s, _ = np.linalg.svd(M32)
    G = M32 @ M32.T
    thr = (8 * EPS32) ** 2
    B = chol_drop(G, thr).astype(np.float32).astype(np.float64)      # r x m
    r = B.shape[0]
    G2 = B @ B.T
    B2 = chol_drop(G2, thr).astype(np.float32).astype(np.float64)     # r2 x r
    r2 = B2.shape[0]
    # baseline: exact SVD of B2 (what the Jacobi gives): right singular vectors of B2 = left singular vectors of B
    _, s2, Wt = np.linalg.svd(B2)
    U_base = Wt[:chi] @ B
    U_base /= np.linalg.norm(U_base, axis=1)[:, None]
    e_base, _ = graded_err(U_base, Us[:, :chi], s[:chi])
    # candidate: block subspace iteration on G2 (f64)
    if start == "unit":
        Q = np.eye(r)[:, :p]
    else:
        Q = np.linalg.qr(np.random.default_rng(0).standard_normal((r, p)))[0]
    conds = []
    for it in range(iters):
        Z = G2 @ Q
        if cholqr:
            S = Z.T @ Z
            d = 1 / np.sqrt(np.diag(S))
            Ss = S * d[:, None] * d[None, :]
            conds.append(np.linalg.cond(Ss))
            Lc = np.linalg.cholesky(Ss)
            Q = (Z * d[None, :]) @ np.linalg.inv(Lc).T
        else:
            Q, _ = np.linalg.qr(Z)
    H = Q.T @ G2 @ Q
    lam, Y = np.linalg.eigh(H)
    order = np.argsort(-lam)
    W = (Q @ Y[:, order[:chi]]).T          # chi x r
    U_c = W @ B
    U_c /= np.linalg.norm(U_c, axis=1)[:, None]
    e_c, per = graded_err(U_c, Us[:, :chi], s[:chi])
    return dict(r=r, r2=r2, e_base=e_base, e_cand=e_c, s32=s[31] / s[0], s33=s[32] / s[0], s40=s[39] / s[0], conds=conds)

if __name__ == "__main__":
    p = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    start = sys.argv[3] if len(sys.argv) > 3 else "unit"
    cq = len(sys.argv) > 4 and sys.argv[4] == "cholqr"
    res = [pipeline(M, p, iters, start, cholqr=cq) for M in M_all[::3]]
    for k in ("r", "r2", "s32", "s33", "s40", "e_base", "e_cand"):
        v = np.array([x[k] for x in res])
        print("%-7s min %.3g median %.3g max %.3g" % (k, v.min(), np.median(v), v.max()))
    if cq:
        c = np.array([max(x["conds"]) for x in res]); print("cond of scaled Gram max", c.max(), "median", np.median(c))

def chol_pivoted(G, k, thr):
    """diagonally pivoted Cholesky stopped after k steps (or when the largest remaining pivot falls below thr * max diag):
    rows of the factor in pivot order (k x n), B^T B ~ G"""
    n = G.shape[0]
    d = np.diag(G).copy()
    dmax = d.max()
    L = np.zeros((k, n))
    piv = []
    for j in range(k):
        p = int(np.argmax(d))
        if d[p] < thr * dmax:
            break
        row = G[p, :] - L[:j, p] @ L[:j, :]
        row /= np.sqrt(d[p])
        L[j] = row
        d -= row * row
        d[p] = -1.0
        piv.append(p)
    return L[:len(piv)], piv

def pipeline_piv(M, k=64, second=True):
    M32 = (M / np.abs(M).max()).astype(np.float32).astype(np.float64)
    Us, s, _ = np.linalg.svd(M32)
    G = M32 @ M32.T
    thr = (8 * EPS32) ** 2
    B, piv = chol_pivoted(G, k, thr)
    B = B.astype(np.float32).astype(np.float64)
    r = B.shape[0]
    if second:
        G2 = B @ B.T
        B2 = chol_drop(G2, thr).astype(np.float32).astype(np.float64)
        _, s2, Wt = np.linalg.svd(B2)
        U = Wt[:chi] @ B
    else:
        _, s2, Vt = np.linalg.svd(B, full_matrices=False)
        U = Vt[:chi]
    U /= np.linalg.norm(U, axis=1)[:, None]
    e, per = graded_err(U, Us[:, :chi], s[:chi])
    return dict(r=r, e=e)

if __name__ == "__main__" and len(sys.argv) > 5:
    for k in (40, 48, 56, 64, 80):
        res = [pipeline_piv(M, k) for M in M_all[::3]]
        v = np.array([x["e"] for x in res]); rr = np.array([x["r"] for x in res])
        print("pivoted k=%d: r median %d max %d; graded err min %.3g median %.3g max %.3g" % (k, np.median(rr), rr.max(), v.min(), np.median(v), v.max()))

def chol_block_pivoted(G, k, thr, nb):
    """pivots chosen nb at a time by the current diagonal (one round of row fetches), processed in that order; a chosen pivot whose
    remaining diagonal fell below the threshold inside the round is skipped"""
    n = G.shape[0]
    d = np.diag(G).copy()
    dmax = d.max()
    L = np.zeros((k, n))
    j = 0
    done = np.zeros(n, bool)
    rounds = 0
    while j < k:
        cand = [p for p in np.argsort(-d)[:nb] if d[p] >= thr * dmax and not done[p]]
        if not cand:
            break
        rounds += 1
        for p in cand:
            if j >= k:
                break
            if d[p] < thr * dmax:
                continue
            row = G[p, :] - L[:j, p] @ L[:j, :]
            row[done] = 0.0
            row /= np.sqrt(d[p])
            L[j] = row
            d -= row * row
            d[p] = -1.0
            done[p] = True
            j += 1
    return L[:j], rounds

def pipeline_bpiv(M, k, nb):
    M32 = (M / np.abs(M).max()).astype(np.float32).astype(np.float64)
    Us, s, _ = np.linalg.svd(M32)
    G = M32 @ M32.T
    thr = (8 * EPS32) ** 2
    B, rounds = chol_block_pivoted(G, k, thr, nb)
    B = B.astype(np.float32).astype(np.float64)
    G2 = B @ B.T
    B2 = chol_drop(G2, thr).astype(np.float32).astype(np.float64)
    _, s2, Wt = np.linalg.svd(B2)
    U = Wt[:chi] @ B
    U /= np.linalg.norm(U, axis=1)[:, None]
    e, per = graded_err(U, Us[:, :chi], s[:chi])
    return dict(r=B.shape[0], e=e, rounds=rounds, r2=B2.shape[0])

if __name__ == "__main__" and len(sys.argv) > 5:
    for k, nb in ((64, 1), (64, 2), (64, 4), (64, 8), (56, 4), (72, 4), (80, 8)):
        res = [pipeline_bpiv(M, k, nb) for M in M_all[::3]]
        v = np.array([x["e"] for x in res]); rr = np.array([x["r"] for x in res]); ro = np.array([x["rounds"] for x in res]); r2 = np.array([x["r2"] for x in res])
        print("block-pivoted k=%d nb=%d: r median %d max %d r2 median %d; rounds median %d; graded err min %.3g median %.3g max %.3g" % (k, nb, np.median(rr), rr.max(), np.median(r2), np.median(ro), v.min(), np.median(v), v.max()))

def chol_wave_nominee(G, k, thr, sort=True, groups=4):
    """the kernel's rule: the columns are split among `groups` waves of 64; a round takes every wave's largest remaining diagonal
    as a candidate (if above the threshold), processes them (by decreasing diagonal when sort); a candidate whose diagonal fell below the
    threshold inside the round burns its slot (dead row)"""
    n = G.shape[0]
    d = np.diag(G).copy()
    dmax = d.max()
    L = np.zeros((k, n))
    live = np.zeros(k, bool)
    done = np.zeros(n, bool)
    j = 0
    rounds = 0
    while j + groups <= k:
        cand = []
        for g in range(groups):
            lo, hi = 64 * g, min(n, 64 * (g + 1))
            if lo >= hi: cand.append(-1); continue
            dd = np.where(done[lo:hi], -1.0, d[lo:hi])
            p = lo + int(np.argmax(dd))
            cand.append(p if dd[p - lo] >= thr * dmax else -1)
        if all(c < 0 for c in cand):
            break
        rounds += 1
        if sort:
            cand.sort(key=lambda p: -(d[p] if p >= 0 else -1))
        for p in cand:
            if p >= 0 and d[p] >= thr * dmax:
                row = G[p, :] - L[:j, p] @ L[:j, :]
                row[done] = 0.0
                row /= np.sqrt(d[p])
                L[j] = row
                live[j] = True
                d -= row * row
                d[p] = -1.0
                done[p] = True
            j += 1
    return L[live], rounds

def pipeline_nom(M, k, sort):
    M32 = (M / np.abs(M).max()).astype(np.float32).astype(np.float64)
    Us, s, _ = np.linalg.svd(M32)
    G = M32 @ M32.T
    thr = (8 * EPS32) ** 2
    B, rounds = chol_wave_nominee(G, k, thr, sort)
    B = B.astype(np.float32).astype(np.float64)
    G2 = B @ B.T
    B2 = chol_drop(G2, thr).astype(np.float32).astype(np.float64)
    _, s2, Wt = np.linalg.svd(B2)
    U = Wt[:chi] @ B
    U /= np.linalg.norm(U, axis=1)[:, None]
    e, per = graded_err(U, Us[:, :chi], s[:chi])
    return dict(r=B.shape[0], e=e, rounds=rounds, r2=B2.shape[0])

if __name__ == "__main__" and len(sys.argv) > 5:
    for k, srt in ((64, True), (64, False), (72, True), (80, True), (96, True)):
        res = [pipeline_nom(M, k, srt) for M in M_all[::2]]
        v = np.array([x["e"] for x in res]); rr = np.array([x["r"] for x in res]); ro = np.array([x["rounds"] for x in res]); r2 = np.array([x["r2"] for x in res])
        print("wave-nominee k=%d sort=%d: r median %d max %d r2 median %d; rounds median %d max %d; graded err min %.3g median %.3g max %.3g" % (k, srt, np.median(rr), rr.max(), np.median(r2), np.median(ro), ro.max(), v.min(), np.median(v), v.max()))

def chol_block_pivoted_burn(G, k, thr, nb):
    """as chol_block_pivoted, but with the kernel's static slots: a candidate rejected inside its round burns its slot"""
    n = G.shape[0]
    d = np.diag(G).copy()
    dmax = d.max()
    L = np.zeros((k, n)); live = np.zeros(k, bool); done = np.zeros(n, bool)
    j = 0; rounds = 0
    while j + nb <= k:
        order = np.argsort(-d)[:nb]
        cand = [int(p) if d[p] >= thr * dmax else -1 for p in order]
        if all(c < 0 for c in cand):
            break
        rounds += 1
        for p in cand:
            if p >= 0 and d[p] >= thr * dmax:
                row = G[p, :] - L[:j, p] @ L[:j, :]
                row[done] = 0.0
                row /= np.sqrt(d[p])
                L[j] = row; live[j] = True
                d -= row * row; d[p] = -1.0; done[p] = True
            j += 1
    return L[live], rounds

if __name__ == "__main__" and len(sys.argv) > 5:
    def run(k, nb):
        out = []
        for M in M_all[::2]:
            M32 = (M / np.abs(M).max()).astype(np.float32).astype(np.float64)
            Us, s, _ = np.linalg.svd(M32)
            G = M32 @ M32.T
            thr = (8 * EPS32) ** 2
            B, rounds = chol_block_pivoted_burn(G, k, thr, nb)
            B = B.astype(np.float32).astype(np.float64)
            B2 = chol_drop(B @ B.T, thr).astype(np.float32).astype(np.float64)
            _, s2, Wt = np.linalg.svd(B2)
            U = Wt[:chi] @ B
            U /= np.linalg.norm(U, axis=1)[:, None]
            out.append((B.shape[0], graded_err(U, Us[:, :chi], s[:chi])[0], rounds))
        a = np.array(out)
        print("burn k=%d nb=%d: r median %d max %d rounds median %d; graded err median %.3g max %.3g" % (k, nb, np.median(a[:, 0]), a[:, 0].max(), np.median(a[:, 2]), np.median(a[:, 1]), a[:, 1].max()))
    for k, nb in ((64, 4), (64, 8), (64, 2), (72, 4), (80, 4)):
        run(k, nb)
