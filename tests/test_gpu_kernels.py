"""GPU unit tests of the HIP kernels through the C ABI diagnostics (tgemm / Cholesky / Jacobi)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _capi():
    from peps_amd import capi
    return capi


def _ref_tgemm(I, J, K, sAi, sAk, sBk, sBj, sCi, sCj, A, B, C, nb, wA, wB, wC):
    C = C.astype(np.float64).copy()
    A = A.astype(np.float64)
    B = B.astype(np.float64)
    ii = np.indices(I).reshape(3, -1)
    jj = np.indices(J).reshape(3, -1)
    kk = np.indices(K).reshape(3, -1)
    oAi = (np.array(sAi)[:, None] * ii).sum(0)
    oCi = (np.array(sCi)[:, None] * ii).sum(0)
    oAk = (np.array(sAk)[:, None] * kk).sum(0)
    oBk = (np.array(sBk)[:, None] * kk).sum(0)
    oBj = (np.array(sBj)[:, None] * jj).sum(0)
    oCj = (np.array(sCj)[:, None] * jj).sum(0)
    for b in range(nb):
        Am = A[b * wA + oAi[:, None] + oAk[None, :]]
        Bm = B[b * wB + oBk[:, None] + oBj[None, :]]
        C[b * wC + oCi[:, None] + oCj[None, :]] = Am @ Bm
    return C


CASES = [
    # I, J, K sub-dims (outer..inner) -- plain, odd sizes, multi-index, K > table chunk
    ((1, 1, 64), (1, 1, 64), (1, 1, 32)),
    ((1, 1, 70), (1, 1, 33), (1, 1, 19)),
    ((1, 5, 7), (1, 3, 11), (1, 4, 9)),
    ((2, 3, 5), (3, 2, 4), (2, 5, 3)),
    ((1, 1, 130), (1, 1, 129), (1, 1, 2100)),
    ((1, 1, 1), (1, 1, 1), (3, 4, 5)),
    ((1, 1, 256), (1, 1, 8), (1, 1, 8)),
]


@pytest.mark.parametrize("case", range(len(CASES)))
@pytest.mark.parametrize("dt", ["f32", "f64", "f32f64"])
def test_tgemm_matches_numpy(case, dt):
    capi = _capi()
    I, J, K = CASES[case]
    rng = np.random.default_rng(100 + case)
    ni, nj, nk = int(np.prod(I)), int(np.prod(J)), int(np.prod(K))
    # A stored with a random permutation of the 6 (i,k) sub-indices, likewise B, C
    def strides(dims, perm):
        st = [0] * len(dims)
        acc = 1
        for ax in reversed(perm):
            st[ax] = acc
            acc *= dims[ax]
        return st
    pa = rng.permutation(6)
    stA = strides(list(I) + list(K), list(pa))
    pb = rng.permutation(6)
    stB = strides(list(K) + list(J), list(pb))
    pc = rng.permutation(6)
    stC = strides(list(I) + list(J), list(pc))
    nb = 3
    wA, wB, wC = ni * nk + 5, nk * nj + 3, ni * nj + 7
    A = rng.standard_normal(nb * wA)
    B = rng.standard_normal(nb * wB)
    C0 = np.full(nb * wC, 7.0)
    din = capi.F64 if dt == "f64" else capi.F32
    dout = capi.F32 if dt == "f32" else capi.F64
    if din == capi.F32:
        A = A.astype(np.float32)
        B = B.astype(np.float32)
    got = capi.diag_tgemm(din, dout, I, J, K, stA[:3], stA[3:], stB[:3], stB[3:], stC[:3], stC[3:], A, B, C0, nb, wA, wB, wC)
    ref = _ref_tgemm(I, J, K, stA[:3], stA[3:], stB[:3], stB[3:], stC[:3], stC[3:], A, B, C0, nb, wA, wB, wC)
    tol = 2e-5 * np.sqrt(nk) if dt == "f32" else 1e-12 * np.sqrt(nk)
    assert np.max(np.abs(got - ref)) < tol * max(1.0, np.max(np.abs(ref)))
    # untouched padding between batches keeps the initial value
    mask = np.ones(nb * wC, bool)
    for b in range(nb):
        mask[b * wC:b * wC + ni * nj] = False
    assert np.all(got[mask] == 7.0)


@pytest.mark.parametrize("case", range(len(CASES)))
@pytest.mark.parametrize("mode", [1, 2])
def test_tgemm_wave_per_tile_bodies(case, mode, monkeypatch):
    """The wave-per-tile kernel on the descriptor cases of the LDS-tiled one: mode 1 = tg_direct_body (f32 MFMA 32x32x2), mode 2 =
    tg_direct_body_f64 (round 5: f32 operands, v_mfma_f64_16x16x4_f64 accumulation -- the result is the float64 sum rounded once)."""
    capi = _capi()
    I, J, K = CASES[case]
    rng = np.random.default_rng(300 + case)
    ni, nj, nk = int(np.prod(I)), int(np.prod(J)), int(np.prod(K))
    def strides(dims, perm):
        st = [0] * len(dims)
        acc = 1
        for ax in reversed(perm):
            st[ax] = acc
            acc *= dims[ax]
        return st
    stA = strides(list(I) + list(K), list(rng.permutation(6)))
    stB = strides(list(K) + list(J), list(rng.permutation(6)))
    stC = strides(list(I) + list(J), list(rng.permutation(6)))
    nb = 3
    wA, wB, wC = ni * nk + 8, nk * nj + 4, ni * nj + 7
    A = rng.standard_normal(nb * wA).astype(np.float32)
    B = rng.standard_normal(nb * wB).astype(np.float32)
    C0 = np.full(nb * wC, 7.0)
    monkeypatch.setenv("PEPSGPU_DIAG_TGEMM_MODE", str(mode))
    got = capi.diag_tgemm(capi.F32, capi.F32, I, J, K, stA[:3], stA[3:], stB[:3], stB[3:], stC[:3], stC[3:], A, B, C0, nb, wA, wB, wC)
    ref = _ref_tgemm(I, J, K, stA[:3], stA[3:], stB[:3], stB[3:], stC[:3], stC[3:], A.astype(np.float64), B.astype(np.float64), C0, nb, wA, wB, wC)
    scale = max(1.0, np.max(np.abs(ref)))
    if mode == 2:      # one rounding to float32 of the exact float64 sum
        assert np.max(np.abs(got - ref.astype(np.float32))) <= 1.2e-7 * scale
    else:
        assert np.max(np.abs(got - ref)) < 2e-5 * np.sqrt(nk) * scale
    mask = np.ones(nb * wC, bool)
    for b in range(nb):
        mask[b * wC:b * wC + ni * nj] = False
    assert np.all(got[mask] == 7.0)


@pytest.mark.parametrize("n", [1, 7, 16, 40, 129, 256])
@pytest.mark.parametrize("dt", ["f32", "f64"])
def test_cholesky(n, dt):
    capi = _capi()
    rng = np.random.default_rng(n)
    nb = 3
    X = rng.standard_normal((nb, 2 * n + 3, n)) * np.logspace(0, -3, n)[None, None, :]
    G = np.einsum("bri,brj->bij", X, X)
    R = capi.diag_chol(capi.F32 if dt == "f32" else capi.F64, G).astype(np.float64)
    for b in range(nb):
        sc = np.max(np.diag(G[b]))
        assert np.allclose(np.tril(R[b], -1), 0)
        err = np.max(np.abs(R[b].T @ R[b] * sc - G[b])) / sc
        assert err < (3e-6 if dt == "f32" else 1e-12)


def test_cholesky_rank_deficient():
    capi = _capi()
    rng = np.random.default_rng(5)
    X = rng.standard_normal((2, 10, 32))        # rank 10 < 32
    G = np.einsum("bri,brj->bij", X, X)
    R = capi.diag_chol(capi.F64, G)
    for b in range(2):
        sc = np.max(np.diag(G[b]))
        assert np.all(np.isfinite(R[b]))
        assert np.max(np.abs(R[b].T @ R[b] * sc - G[b])) / sc < 1e-10


@pytest.mark.parametrize("shape", [(1, 5), (2, 2), (9, 33), (64, 64), (144, 144), (256, 256), (37, 200)])
@pytest.mark.parametrize("dt", ["f32", "f64"])
@pytest.mark.parametrize("force_global", [False, True])
def test_jacobi_rows(shape, dt, force_global):
    capi = _capi()
    m, ln = shape
    if force_global and m * ln > 64 * 64 and (m, ln) != (256, 256):
        pytest.skip("global-memory path covered by the small and the 256x256 cases")
    rng = np.random.default_rng(m * 1000 + ln)
    nb = 2
    r = min(m, ln)
    # graded spectrum like a boundary-MPS bond
    U, _ = np.linalg.qr(rng.standard_normal((m, r)))
    V, _ = np.linalg.qr(rng.standard_normal((ln, r)))
    s = np.logspace(0, -5, r)
    M = np.stack([(U * s) @ V.T, (U * s[::-1]) @ V.T])
    k = max(1, r // 2)
    dtc = capi.F32 if dt == "f32" else capi.F64
    Mo, Vt, S, sw = capi.diag_jacobi(dtc, M, k, force_global)
    eps = 3e-5 if dt == "f32" else 1e-12
    for b in range(nb):
        sref = np.linalg.svd(M[b], compute_uv=False)
        assert np.max(np.abs(S[b].astype(np.float64) - sref[:k])) < eps * sref[0]
        # rows of Vt orthonormal and spanning the dominant right singular subspace
        Vb = Vt[b].astype(np.float64)
        assert np.max(np.abs(Vb @ Vb.T - np.eye(k))) < 20 * eps
        Pref = np.linalg.svd(M[b])[2][:k]
        # projector difference only checked where the spectrum has a gap at k
        if k < r and sref[k] < 0.5 * sref[k - 1]:
            gap_tol = 50 * eps * sref[0] / (sref[k - 1] - sref[k])
            assert np.max(np.abs(Vb.T @ Vb - Pref.T @ Pref)) < max(gap_tol, 50 * eps)
        assert sw[b] < 40


@pytest.mark.parametrize("shape", [(256, 256), (200, 256), (256, 130), (129, 255), (17, 40)])
def test_jacobi_register_kernel_matches_generic(shape):
    """Register-resident 256x256 f32 Jacobi (jacobi_reg.h) == generic kernel: same singular values,
    orthonormal Vt, same dominant subspace; application-like input M = R T with R upper triangular."""
    capi = _capi()
    m, ln = shape
    rng = np.random.default_rng(m + 7 * ln)
    nb = 3
    Ms = []
    for b in range(nb):
        R = np.triu(rng.standard_normal((m, m))) * np.logspace(0, -6, m)[:, None]
        T = rng.standard_normal((m, ln))
        Ms.append(R @ T)
    M = np.stack(Ms)
    k = min(32, m, ln)
    Mo, Vt, S, sw = capi.diag_jacobi(capi.F32, M, k, 2)
    Mg, Vtg, Sg, swg = capi.diag_jacobi(capi.F32, M, k, 1)
    for b in range(nb):
        sref = np.linalg.svd(M[b], compute_uv=False)
        assert np.max(np.abs(S[b].astype(np.float64) - sref[:k])) < 3e-5 * sref[0]
        Vb = Vt[b].astype(np.float64)
        live = sref[:k] > 1e-5 * sref[0]          # rows under the noise floor come back as zero rows
        G = Vb @ Vb.T
        assert np.max(np.abs(G[np.ix_(live, live)] - np.eye(int(live.sum())))) < 1e-4
        # rotated rows stay a rotation of the input: Gram of the rows' span is preserved
        assert abs(np.linalg.norm(Mo[b]) / np.linalg.norm(M[b]) - 1) < 1e-5
        assert sw[b] < 40 and swg[b] < 40
        Pref = np.linalg.svd(M[b])[2][:k]
        if live.all() and (sref[k] < 0.5 * sref[k - 1] if k < len(sref) else True):
            assert np.max(np.abs(Vb.T @ Vb - Pref.T @ Pref)) < 2e-3


@pytest.mark.parametrize("shape", [(32, 256), (16, 256), (8, 256), (5, 200), (9, 256), (12, 256), (13, 77), (20, 131), (1, 64),
                                   (10, 96), (16, 128), (7, 50), (12, 127)])
def test_jacobi_small_rank_kernel(shape):
    """One-wave-per-walker Jacobi (jacobi_rows_small_kernel, carries with <= 32 existing rows):
    singular values and dominant subspace against LAPACK, several walkers per workgroup."""
    capi = _capi()
    m, ln = shape
    rng = np.random.default_rng(11 * m + ln)
    nb = 7                                        # not a multiple of the 4 walkers per workgroup
    M = np.stack([(np.triu(rng.standard_normal((m, m))) * np.logspace(0, -4, m)[:, None]) @ rng.standard_normal((m, ln))
                  for _ in range(nb)])
    k = min(m, ln)
    Mo, Vt, S, sw = capi.diag_jacobi(capi.F32, M, k, 3)
    for b in range(nb):
        sref = np.linalg.svd(M[b], compute_uv=False)
        assert np.max(np.abs(S[b].astype(np.float64) - sref[:k])) < 3e-5 * sref[0]
        Vb = Vt[b].astype(np.float64)
        live = sref[:k] > 1e-5 * sref[0]
        G = Vb @ Vb.T
        assert np.max(np.abs(G[np.ix_(live, live)] - np.eye(int(live.sum())))) < 1e-4
        assert abs(np.linalg.norm(Mo[b]) / np.linalg.norm(M[b]) - 1) < 1e-5
        assert sw[b] < 40


@pytest.mark.parametrize("dt", ["f32", "f64"])
@pytest.mark.parametrize("n,ranks", [(256, (3, 12, 31, 32, 33, 60, 256)), (144, (1, 20, 144)), (40, (5, 40))])
def test_cholesky_rank_adaptive_pair(n, ranks, dt):
    """chol_lowrank_kernel (rank <= 32, right-looking, LDS-resident) + chol_upper_kernel for the
    flagged walkers, as the absorption runs them: R^T R = G on the live rows, mlive = numerical rank,
    walkers of different rank in one launch.  Only the upper triangle of G is read."""
    capi = _capi()
    rng = np.random.default_rng(n + len(ranks))
    Gs = []
    for r in ranks:
        X = rng.standard_normal((r, n))
        G = X.T @ X
        G[np.tril_indices(n, -1)] = np.nan if False else 1e30     # lower triangle must never be read
        Gs.append(G)
    G = np.stack(Gs)
    R, ml = capi.diag_chol_adaptive(capi.F32 if dt == "f32" else capi.F64, G)
    tol = 3e-6 if dt == "f32" else 1e-9      # unpivoted factor of an exactly rank-deficient block
    for b, r in enumerate(ranks):
        Gu = np.triu(Gs[b]) + np.triu(Gs[b], 1).T
        sc = np.max(np.diag(Gu))
        slack = 0 if dt == "f32" else 2          # f64 threshold sits at rounding level: a rounding-noise pivot may pass
        assert 0 < ml[b] <= min(r + slack, n) and (r >= n or ml[b] >= r - 1), (r, ml[b])   # r = n: smallest directions may fall under the floor
        Rb = R[b, :ml[b]].astype(np.float64)
        assert np.max(np.abs(Rb.T @ Rb * sc - Gu)) / sc < tol, (r, ml[b])


@pytest.mark.parametrize("dt", ["f32", "f64"])
@pytest.mark.parametrize("K,n,rank", [(80, 256, 10), (96, 256, 32), (64, 256, 33), (16, 144, 16), (7, 40, 3), (64, 256, 1),
                                         (200, 256, 20), (288, 256, 28), (97, 200, 5), (300, 256, 10)])
def test_gram_free_lowrank_factor(K, n, rank, dt):
    """gram_chol_lowrank_kernel: R^T R = P^T P straight from the K live rows of P (no Gram matrix in
    memory); declines (mlive = -1) when the rank exceeds its cap of 32."""
    capi = _capi()
    cap = 96 if dt == "f32" else 48
    if K > cap + 3 * (cap - 32):                      # more rows than four passes fold in: declined
        R, ml = capi.diag_gram_chol(capi.F32 if dt == "f32" else capi.F64, np.zeros((2, K, n)) + 1.0)
        assert np.all(ml == -1)
        return
    rng = np.random.default_rng(K + n + rank)
    nb = 5
    P = np.stack([rng.standard_normal((K, rank)) @ rng.standard_normal((rank, n)) for _ in range(nb)])
    R, ml = capi.diag_gram_chol(capi.F32 if dt == "f32" else capi.F64, P)
    tol = 2e-5 if dt == "f32" else 1e-9
    for b in range(nb):
        if rank > 32 or (rank >= 31 and ml[b] == -1):     # at the cap the rounding noise of the input may tip it over
            assert ml[b] == -1
            continue
        Pb = P[b].astype(np.float32).astype(np.float64) if dt == "f32" else P[b]
        G = Pb.T @ Pb
        sc = np.max(np.diag(G))
        assert rank - 1 <= ml[b] <= min(rank + 2, K, n), (rank, ml[b])     # rounding noise of the input may pass as a pivot
        Rb = R[b, :ml[b]].astype(np.float64)
        assert np.max(np.abs(Rb.T @ Rb * sc - G)) / sc < tol


def test_rows_at_the_noise_floor_never_enter_vt_unorthogonalised():
    """A row whose norm sits at the Jacobi freeze threshold (NOISE_C eps |M|_F) is not rotated; select_rows must not
    keep it as a live direction (normalised it would overlap the dominant one).  4096 matrices with a residue-like row
    whose norm lies within a few 1e-6 (relative) of the threshold, i.e. inside the window where the two kernels'
    differently summed norms disagree; whatever rows come back non-zero must be orthonormal."""
    from peps_amd import capi
    rng = np.random.default_rng(11)
    ln, m, nb = 96, 16, 4096
    eps = 5.9604645e-8
    M = np.zeros((nb, m, ln), dtype=np.float32)
    for b in range(nb):
        big = rng.standard_normal(ln)
        M[b, 0] = big
        for r in range(1, 6):
            M[b, r] = 1e-4 * np.linalg.norm(big) * rng.standard_normal(ln) / np.sqrt(ln)
        noise = 0.6 * big / np.linalg.norm(big) + 0.8 * rng.standard_normal(ln) / np.sqrt(ln)   # mostly along the dominant row
        fro = np.sqrt(np.sum(M[b].astype(np.float64) ** 2))
        M[b, 6] = (1.0 + 6e-6 * (rng.random() - 0.5)) * 8 * eps * fro * noise / np.linalg.norm(noise)
    out, Vt, S, sw = capi.diag_jacobi(capi.F32, M, 8, force_global=3)
    worst = 0.0
    for b in range(nb):
        v = Vt[b].astype(np.float64)
        live = np.linalg.norm(v, axis=1) > 0
        g = v[live] @ v[live].T
        worst = max(worst, float(np.max(np.abs(g - np.eye(int(live.sum()))))))
    assert worst < 2e-5, worst


@pytest.mark.parametrize("dt", ["f32", "f64"])
def test_gram_free_factor_mixed_rows_and_ranks_in_one_launch(dt):
    """One launch with walkers of different live row counts and ranks: the short first pass (<= 64 rows, rank <= 16 per
    thread), the walkers it hands on to the full kernel (more rows or rank 17..32) and the multi-pass path (more rows than
    a pass holds) all take part; R^T R = P^T P / max diag and the rank for every walker."""
    from peps_amd import capi
    rng = np.random.default_rng(21)
    t = np.float32 if dt == "f32" else np.float64
    d = capi.F32 if dt == "f32" else capi.F64
    K, n = 160, 96
    cases = [(40, 6), (64, 12), (64, 20), (90, 10), (128, 14), (160, 9), (30, 30), (100, 28)] * 4
    P = np.zeros((len(cases), K, n), dtype=t)
    for b, (k, r) in enumerate(cases):
        P[b, :k] = (rng.standard_normal((k, r)) @ rng.standard_normal((r, n))).astype(t)
    R, ml = capi.diag_gram_chol(d, P)
    tol = 5e-6 if dt == "f32" else 1e-12
    for b, (k, r) in enumerate(cases):
        if ml[b] < 0:                      # declined to the Gram path: only beyond the capacity of the full kernel
            cap = 96 + 3 * 64 if dt == "f32" else 48 + 3 * 16        # KCAP + (passes - 1) (KCAP - 32)
            assert r > 32 or k > cap, (b, k, r, ml[b])
            continue
        assert r <= ml[b] <= r + 2, (b, k, r, ml[b])
        G = P[b].astype(np.float64).T @ P[b].astype(np.float64)
        Rb = R[b][:ml[b]].astype(np.float64)
        assert np.max(np.abs(Rb.T @ Rb - G / np.max(np.diag(G)))) < tol * 10, (b, k, r)
    if dt == "f32":
        assert np.all(ml >= 0)


@pytest.mark.parametrize("f64acc", [0, 1, 2])
def test_chained_contraction_pair_with_live_extents(f64acc, monkeypatch):
    """(f64acc = 1: the float64-accumulating form of both stages, round 5 -- the intermediate is still f32 in LDS.)
    tgemm_chain_kernel (X = R.A kept in LDS, P = W.X) with the descriptors of the absorption: per-walker live extents of
    the carry rows and of both bonds.  Entries whose live X exceeds the 24 KB LDS buffer are walked in chunks of carry rows
    (X still never leaves the chip); an entry is declined (flag -1, result untouched) only when one carry row's slice of X
    does not fit."""
    from peps_amd import capi
    # (f64acc = 2: the triangular form of round 6 on the 32 KB buffer of dense batches -- R a factor with row j zero before column j, its
    # zero l-blocks skipped in both stages)
    monkeypatch.setenv("PEPSGPU_DIAG_CHAIN_F64", "1" if f64acc == 1 else "0")
    monkeypatch.setenv("PEPSGPU_DIAG_TRI", "1" if f64acc == 2 else "0")
    rng = np.random.default_rng(5)
    for (nb, m, l, a, p, a2, l2, u) in [(48, 24, 8, 16, 8, 16, 8, 8), (6, 40, 8, 32, 8, 32, 8, 8), (4, 5, 8, 8, 8, 128, 4, 4)]:
        R = rng.standard_normal((nb, m, l, a)).astype(np.float32)
        tri = f64acc == 2
        if tri:       # a row-compacted triangular factor: row j zero before column j of the flattened (l, a) index
            Rf = R.reshape(nb, m, l * a)
            Rf[:, np.arange(m)[:, None] > np.arange(l * a)[None, :]] = 0.0
        cap = 8192 if tri else 6144
        A = rng.standard_normal((nb, a, p, a2)).astype(np.float32)
        W = rng.standard_normal((nb, l, p, l2, u)).astype(np.float32)
        live = np.stack([rng.integers(1, m + 1, nb), rng.integers(1, a + 1, nb), rng.integers(1, a2 + 1, nb)], axis=1)
        live[0] = (m, a, a2)                       # static extents: chunks of carry rows (or declined: third shape)
        live[1] = (min(m, 8), min(a, 10), 10)      # the headline's typical live sizes: 8*8*8*10 = 5120 -> one chunk
        P, fl = capi.diag_tgemm_chain(R, A, W, live)
        if tri and m > a:     # control: the skip is real -- a factor that is NOT triangular loses the blocks before its rows
            Rbad = R.copy(); Rbad[0, m - 1, 0, 0] = 1.0
            Pbad, _ = capi.diag_tgemm_chain(Rbad, A, W, live)
            assert np.array_equal(Pbad[0], P[0])
        n_chunked = 0
        for b in range(nb):
            ml, al, a2l = (int(x) for x in live[b])
            fits_row = l * p * a2l <= cap
            assert fl[b] == (0 if fits_row else -1), (b, live[b], fl[b])
            if not fits_row:
                assert not np.any(P[b])
                continue
            n_chunked += ml * l * p * a2l > cap
            X = np.einsum("mla,apc->mlpc", R[b, :ml, :, :al].astype(np.float64), A[b, :al, :, :a2l].astype(np.float64))
            want = np.einsum("mlpc,lpqu->muqc", X, W[b].astype(np.float64))
            got = P[b, :ml, :, :, :a2l]
            assert np.max(np.abs(got - want)) < (3e-7 if f64acc == 1 else 2e-5) * np.max(np.abs(want)), (b, live[b])
        if a2 <= 32:
            assert n_chunked >= 1
        else:
            assert fl[0] == (-1 if l * p * a2 > cap else 0)


@pytest.mark.parametrize("kernel", [5, 6, 7, 8])
@pytest.mark.parametrize("shape", [(128, 128), (70, 70), (96, 96), (33, 64), (100, 128), (17, 40), (64, 64), (5, 16), (61, 128),
                                   (128, 256), (90, 241), (64, 256), (40, 200)])
def test_jacobi_mid_route_kernel(shape, kernel):
    """Tournament kernels of the preconditioned mid route (up to 128 x 128) on triangular, graded input (the Cholesky factor
    they are given in the absorption): singular values, orthonormal Vt and dominant subspace against LAPACK.  5 / 6: jacobi_rows_grp_kernel<4,8> / <2,8> (16 lanes per row, four
    pairs per wave instruction; the two-wave form takes up to 64 rows); 7 / 8: <4,16> / <2,16>, rows up to 256 long (round 3: the
    route for walkers with up to 256 live carry rows, whose factor B keeps <= 128 live rows of length 256)."""
    capi = _capi()
    m, ln = shape
    if kernel in (6, 8) and m > 64:
        pytest.skip("two-wave tournament: up to 64 rows")
    if kernel < 7 and ln > 128:
        pytest.skip("rows up to 128 long")
    rng = np.random.default_rng(5 * m + ln)
    nb = 5
    M = np.stack([np.triu(rng.standard_normal((m, ln))) * np.logspace(0, -5, m)[:, None] for _ in range(nb)])
    k = min(32, m, ln)
    Mo, Vt, S, sw = capi.diag_jacobi(capi.F32, M, k, kernel)
    for b in range(nb):
        sref = np.linalg.svd(M[b], compute_uv=False)
        assert np.max(np.abs(S[b].astype(np.float64) - sref[:k])) < 3e-5 * sref[0]
        Vb = Vt[b].astype(np.float64)
        live = sref[:k] > 1e-5 * sref[0]
        G = Vb @ Vb.T
        assert np.max(np.abs(G[np.ix_(live, live)] - np.eye(int(live.sum())))) < 1e-4
        assert abs(np.linalg.norm(Mo[b]) / np.linalg.norm(M[b]) - 1) < 1e-5
        assert sw[b] < 40
        Pref = np.linalg.svd(M[b])[2][:k]
        if live.all() and (sref[k] < 0.5 * sref[k - 1] if k < len(sref) else True):
            assert np.max(np.abs(Vb.T @ Vb - Pref.T @ Pref)) < 2e-3


def _i8_image(X, axis):
    """float64 image of the float32 array X the exact-integer Gram kernels work on (peps_amd/csrc/gram_i8.h): per block of 64
    entries along `axis` (the contracted index) and per line, ONE power of two from the block maximum, entries rounded to integers
    |n| <= 2^22 (round half to even)."""
    X = np.moveaxis(np.asarray(X, dtype=np.float32), axis, -1)
    out = np.zeros(X.shape, dtype=np.float64)
    for k0 in range(0, X.shape[-1], 64):
        blk = X[..., k0:k0 + 64]
        m = np.max(np.abs(blk), axis=-1, keepdims=True).astype(np.float32)
        e = np.maximum(30, (m.view(np.uint32) >> 23).astype(np.int64))
        out[..., k0:k0 + 64] = np.ldexp(np.rint(np.ldexp(blk.astype(np.float64), 148 - e)), e - 148)
    return np.moveaxis(out, -1, axis)


@pytest.mark.parametrize("dt", ["f32", "f64"])
@pytest.mark.parametrize("K,n", [(560, 128), (2048, 256), (1500, 224), (333, 256), (37, 96), (5, 64), (301, 200), (64, 33), (1000, 72)])
def test_streaming_gram_kernel(K, n, dt):
    """gram_cols_f64_kernel (wave-per-block streaming Gram of the forward pass): G = P^T P with f64 accumulation of exact
    f32 products, per-walker live row counts, every 64 x 64 block on or above the diagonal."""
    capi = _capi()
    rng = np.random.default_rng(K + n)
    nb = 5
    t = np.float32 if dt == "f32" else np.float64
    P = rng.standard_normal((nb, K, n)).astype(t) * np.logspace(0, -6, n)[None, None, :].astype(t)
    klive = np.array([K, max(1, K // 2), 1, max(1, K - 3), min(K, 7)], dtype=np.int32)
    G = capi.diag_gram_cols(capi.F32 if dt == "f32" else capi.F64, P, klive)
    # round 4: f32 launches of dense shape (193..256 columns, >= 256 rows) run as exact integer arithmetic on the i8 matrix cores
    # (gram_i8.h): the result is the float64 Gram of the fixed-point image of P (asserted to 1e-13), which is within f32 rounding of
    # the Gram of P itself
    i8 = dt == "f32" and 192 < n <= 256 and K >= 256 and os.environ.get("PEPSGPU_NO_I8_GRAM") is None
    for b in range(nb):
        Pb = P[b, :klive[b]].astype(np.float64)
        exact = Pb.T @ Pb
        Pi = _i8_image(P[b, :klive[b]], 0) if i8 else Pb
        ref = Pi.T @ Pi
        for bi in range((n + 63) // 64):
            for bj in range(bi, (n + 63) // 64):
                sl = (slice(64 * bi, min(n, 64 * bi + 64)), slice(64 * bj, min(n, 64 * bj + 64)))
                scale = np.sqrt(np.outer(np.diag(exact)[sl[0]], np.diag(exact)[sl[1]])) + 1e-300
                err = np.abs(G[b][sl] - ref[sl]) / scale
                err_exact = np.abs(G[b][sl] - exact[sl]) / scale
                if bi == bj:      # a diagonal block holds its 16 x 16 tiles on or above the diagonal only (round 3)
                    ii, jj = np.indices(err.shape)
                    err = np.where(jj // 16 >= ii // 16, err, 0.0)
                    err_exact = np.where(jj // 16 >= ii // 16, err_exact, 0.0)
                assert np.max(err) < 1e-13 * max(1, klive[b]) ** 0.5 + 1e-15
                assert np.max(err_exact) < 1.5e-6        # (two factors, each entry within 2^-23 of its column's block maximum)
    G0 = capi.diag_gram_cols(capi.F32 if dt == "f32" else capi.F64, P, None)
    P0 = _i8_image(P[0], 0) if i8 else P[0].astype(np.float64)
    ref0 = P0.T @ P0
    assert np.max(np.abs(np.triu(G0[0]) - np.triu(ref0))[:64, :64]) < 1e-12 * np.max(np.abs(ref0))


@pytest.mark.parametrize("n,K", [(256, 256), (241, 256), (130, 64), (64, 16), (200, 128), (17, 32)])
def test_row_gram_kernel(n, K):
    """gram_rows_f64_kernel (round 3: G = M M^T of the truncation input of walkers with more than 128 live carry rows): exact f32
    products accumulated in f64 on the matrix cores, per-walker live row counts, every 64 x 64 block on or above the diagonal."""
    capi = _capi()
    rng = np.random.default_rng(3 * n + K)
    nb = 4
    M = (rng.standard_normal((nb, n, K)) * np.logspace(0, -6, n)[None, :, None]).astype(np.float32)
    nrows = np.array([n, max(1, n // 2), 1, max(1, n - 3)], dtype=np.int32)
    G = capi.diag_gram_rows(M, nrows)
    i8 = 128 < n <= 256 and os.environ.get("PEPSGPU_NO_I8_GRAM") is None     # round 4: exact integer arithmetic (gram_i8.h, ROWS form)
    for b in range(nb):
        Mb = M[b, :nrows[b]].astype(np.float64)
        exact = Mb @ Mb.T
        Mi = _i8_image(M[b, :nrows[b]], 1) if i8 else Mb
        ref = Mi @ Mi.T
        m = nrows[b]
        for bi in range((m + 63) // 64):
            for bj in range(bi, (m + 63) // 64):
                sl = (slice(64 * bi, min(m, 64 * bi + 64)), slice(64 * bj, min(m, 64 * bj + 64)))
                scale = np.sqrt(np.outer(np.diag(exact)[sl[0]], np.diag(exact)[sl[1]])) + 1e-300
                err = np.abs(G[b][sl] - ref[sl]) / scale
                err_exact = np.abs(G[b][sl] - exact[sl]) / scale
                if bi == bj and i8:      # (the Cholesky reads the upper triangle: the integer kernel stores the 16 x 16 tiles on or above the diagonal)
                    ii, jj = np.indices(err.shape)
                    err = np.where(jj // 16 >= ii // 16, err, 0.0)
                    err_exact = np.where(jj // 16 >= ii // 16, err_exact, 0.0)
                assert np.max(err) < 1e-13 * K ** 0.5 + 1e-15, (b, bi, bj)
                assert np.max(err_exact) < 1.5e-6, (b, bi, bj)


@pytest.mark.parametrize("bad", [np.nan, np.inf])
def test_integer_gram_kernels_propagate_non_finite_input(bad):
    """the exact-integer Grams (gram_i8.h) turn a NaN / Inf of P into finite digits; the kernel detects non-finite input
    itself and poisons the diagonal of G with NaN (what the float64 Gram it replaces propagated into the Cholesky pivots); a clean
    walker of the same launch is untouched."""
    capi = _capi()
    rng = np.random.default_rng(11)
    P = rng.standard_normal((2, 512, 256)).astype(np.float32)
    P[1, 300, 77] = bad
    G = capi.diag_gram_cols(capi.F32, P, None)
    assert np.all(np.isfinite(np.triu(G[0])[:16, :16])) and abs(G[0][3, 3] - np.sum(P[0, :, 3].astype(np.float64) ** 2)) < 1e-4
    assert np.all(np.isnan(np.diag(G[1])))
    M = rng.standard_normal((2, 256, 256)).astype(np.float32)
    M[0, 5, 200] = bad
    Gr = capi.diag_gram_rows(M, np.array([256, 256], dtype=np.int32))
    assert np.all(np.isnan(np.diag(Gr[0]))) and np.all(np.isfinite(np.diag(Gr[1])))


@pytest.mark.parametrize("n,rank", [(256, 256), (256, 97), (241, 180), (160, 33), (128, 128), (100, 7), (48, 48)])
def test_blocked_cholesky_rank_revealing_contract(n, rank):
    """chol_blocked_kernel (round 3, orders >= 48 through launch_chol_upper): R^T R = G on graded Gram matrices of deficient
    rank -- the rows whose pivot falls below the noise of the f32 data are dropped, the factor comes back compacted, and what it
    reproduces is G up to that floor."""
    capi = _capi()
    rng = np.random.default_rng(n + rank)
    nb = 3
    X = rng.standard_normal((nb, rank, n)) * np.logspace(0, -4, rank)[None, :, None]
    G = np.einsum("bri,brj->bij", X, X)
    R = capi.diag_chol(capi.F32, G).astype(np.float64)
    for b in range(nb):
        sc = np.max(np.diag(G[b]))
        err = np.max(np.abs(R[b].T @ R[b] * sc - G[b])) / sc
        assert err < 3e-6, (b, err)
        live = int(np.sum(np.any(R[b] != 0, axis=1)))
        # (a pivot at the rounding level of the Gram matrix may survive the threshold: at most a row or two of negligible norm)
        assert live <= min(rank + 2, n) and np.all(R[b][live:] == 0)


LDS_CHOL_CASES = [(128, 256, 128), (128, 200, 70), (100, 256, 100), (96, 300, 96), (97, 128, 60), (81, 128, 81), (80, 96, 80), (72, 256, 31), (49, 64, 49), (33, 64, 12), (16, 32, 16),
                  (9, 32, 9)]


@pytest.mark.parametrize("which", [0, 1])
@pytest.mark.parametrize("n,K,rank", LDS_CHOL_CASES)
def test_lds_gram_cholesky_kernels(which, n, K, rank):
    """mid_gram_chol_kernel (rows form, G = X X^T) and colgram_dense_kernel (column form, G = X^T X): Gram on the f64 matrix cores
    + the LDS-resident Cholesky -- round 3: blocked in panels of 16 (diagonal block in one wave's registers, forward substitution,
    MFMA trailing update).  Contract: R^T R = G / max diag up to the f32 floor, live rows compacted, a dependent direction
    dropped; mixed live counts in one launch."""
    _lds_gram_chol_case(which, n, K, rank)


def _lds_gram_chol_case(which, n, K, rank):
    capi = _capi()
    rng = np.random.default_rng(1000 * which + n + rank)
    nb = 5
    # rows of the n x K matrix M (rows form) with `rank` independent directions, graded over two decades
    Mx = np.zeros((nb, n, K))
    for b in range(nb):
        base = rng.standard_normal((rank, K)) * np.logspace(0, -2, rank)[:, None]
        Mx[b] = base if rank == n else rng.standard_normal((n, rank)) @ base
    nlive = np.array([n, n, max(1, n - 3), max(1, n // 2), n], dtype=np.int32)
    if which == 0:
        X = Mx.astype(np.float32)
        for b in range(nb):
            X[b, nlive[b]:] = 7.0        # rows beyond the live count must not be read into G
        nl = nlive
    else:
        # column form: P = [K rows][n columns], G = P^T P; live rows of P = K (or fewer: the rank is then capped by the live rows)
        X = np.ascontiguousarray(np.transpose(Mx, (0, 2, 1))).astype(np.float32)
        nl = np.array([K, K, max(1, K - 5), K, max(1, K // 2)], dtype=np.int32)
        for b in range(nb):
            X[b, nl[b]:] = 7.0
    R, ml = capi.diag_lds_gram_chol(which, X, nl)
    for b in range(nb):
        Xd = X[b].astype(np.float64)
        if which == 0:
            nn = int(nlive[b])
            G = Xd[:nn] @ Xd[:nn].T
        else:
            nn = n
            G = Xd[:nl[b]].T @ Xd[:nl[b]]
        sc = np.max(np.diag(G))
        m = int(ml[b])
        rk = min(rank, nn if which == 0 else int(nl[b]))
        assert 1 <= m <= min(nn, rk + 2), (b, m, nn, rk)
        Rb = R[b, :m, :nn].astype(np.float64)
        assert np.all(np.isfinite(Rb)), b
        err = np.max(np.abs(Rb.T @ Rb * sc - G)) / sc
        assert err < 5e-6, (which, b, err, m)
        # upper "staircase": row q starts at its pivot column, pivots in increasing order
        first = [int(np.flatnonzero(Rb[q])[0]) for q in range(m)]
        assert all(first[q] < first[q + 1] for q in range(m - 1)), (b, first)


@pytest.mark.parametrize("m,l,a,u,k2,tsw", [(256, 8, 32, 8, 32, 1), (241, 8, 32, 8, 32, 0), (200, 8, 32, 8, 28, 1), (160, 8, 24, 8, 32, 1),
                                             (256, 6, 24, 6, 32, 0)])
@pytest.mark.parametrize("tri", [0, 1])
def test_mgemm_dense_kernel(m, l, a, u, k2, tsw, tri, monkeypatch):
    """mgemm_dense_kernel (round 3): M = R Tt of dense walkers against NumPy -- live carry rows, live a (rows of Tt beyond it are
    garbage by contract: NaN here), live k2 (columns of M beyond it come back as zeros), both storage orders of Tt."""
    capi = _capi()
    la, uk = l * a, u * k2
    if la % 16 or uk % 32:
        pytest.skip("shape outside the kernel's contract")
    rng = np.random.default_rng(m + la + uk + tsw)
    nb = 4
    R = rng.standard_normal((nb, m, l, a)).astype(np.float32)
    monkeypatch.setenv("PEPSGPU_DIAG_TRI", str(tri))
    if tri:    # (round 6) R a row-compacted triangular factor: the row blocks below a k-chunk are not multiplied
        Rf = R.reshape(nb, m, la)
        Rf[:, np.arange(m)[:, None] > np.arange(la)[None, :]] = 0.0
    T = rng.standard_normal((nb, l, a, u, k2)).astype(np.float32)
    m_live = np.array([m, max(1, m - 17), 129, m], dtype=np.int32)
    a_live = np.array([a, a - 3, a, max(1, a // 2)], dtype=np.int32)
    k_live = np.array([k2, k2, max(1, k2 - 5), k2 // 2], dtype=np.int32)
    Tdev = T.copy()
    for b in range(nb):
        Tdev[b, :, a_live[b]:] = np.nan                      # never written by the producer: must not be read
    Tstore = np.transpose(Tdev, (0, 1, 2, 4, 3)) if tsw else Tdev
    M = capi.diag_mgemm_dense(R.reshape(nb, m, la), Tstore.reshape(nb, la, uk), a, u, k2, tsw, m_live, a_live, k_live)
    if tri:    # control: the skip is real -- an entry below the diagonal blocks is never read
        Rbad = R.copy().reshape(nb, m, la); Rbad[0, m - 1, 0] = 1.0
        Mbad = capi.diag_mgemm_dense(Rbad, Tstore.reshape(nb, la, uk), a, u, k2, tsw, m_live, a_live, k_live)
        assert np.array_equal(Mbad[0, :m_live[0]], M[0, :m_live[0]])
    for b in range(nb):
        ref = np.einsum("mla,lauk->muk", R[b, :, :, :a_live[b]].astype(np.float64), T[b, :, :a_live[b]].astype(np.float64))
        ref[:, :, k_live[b]:] = 0.0
        got = M[b].reshape(m, u, k2)
        assert np.all(np.isfinite(got[:m_live[b]]))
        assert np.max(np.abs(got[:m_live[b]] - ref[:m_live[b]])) < 2e-5 * np.max(np.abs(ref))
        assert np.all(got[:m_live[b], :, k_live[b]:] == 0)
        if m_live[b] < m:
            assert np.all(np.isnan(got[m_live[b]:]))         # rows beyond the live count: untouched


@pytest.mark.parametrize("kernel", [9, 10])
@pytest.mark.parametrize("shape", [(32, 128), (32, 256), (17, 80), (16, 40), (10, 80), (9, 256), (5, 16), (2, 64), (1, 32), (24, 200), (31, 97)])
def test_jacobi_one_wave_grouped_tournament(shape, kernel):
    """jacobi_rows_grp_kernel<1, 8> / <1, 16> (round 3): one wave per walker, four players of two blocks of four rows -- the polish
    of 17..32 rows and every walker of a small batch (<= 2048 walkers: latency).  Dense rows of graded norm (not triangular):
    singular values, orthonormal Vt, norm conservation against LAPACK."""
    capi = _capi()
    m, ln = shape
    if kernel == 9 and ln > 128:
        pytest.skip("rows up to 128 long")
    rng = np.random.default_rng(7 * m + ln + kernel)
    nb = 6
    M = np.stack([rng.standard_normal((m, ln)) * np.logspace(0, -3, m)[:, None] for _ in range(nb)])
    M[1] = M[1][::-1].copy()                      # rows in increasing norm
    if m > 2:
        M[2, m - 1] = M[2, 0]                     # a dependent row
    k = min(32, m, ln)
    Mo, Vt, S, sw = capi.diag_jacobi(capi.F32, M, k, kernel)
    for b in range(nb):
        sref = np.linalg.svd(M[b], compute_uv=False)
        assert np.max(np.abs(S[b].astype(np.float64) - sref[:k])) < 3e-5 * sref[0], (b, S[b], sref[:k])
        Vb = Vt[b].astype(np.float64)
        live = sref[:k] > 1e-5 * sref[0]
        G = Vb @ Vb.T
        assert np.max(np.abs(G[np.ix_(live, live)] - np.eye(int(live.sum())))) < 1e-4, b
        assert abs(np.linalg.norm(Mo[b]) / np.linalg.norm(M[b]) - 1) < 1e-5
        assert sw[b] < 40


@pytest.mark.parametrize("n,K,rank,kcap", [(256, 256, 256, 64), (224, 256, 40, 64), (200, 128, 70, 56), (256, 256, 256, 48), (130, 64, 20, 64)])
def test_pivoted_cholesky_first_compression(n, K, rank, kcap):
    """chol_pivot_kernel behind the i8 row Gram with both triangles (round 6, the first compression of the dense truncation route):
    rows in pivot order with decreasing diagonal; B^T B = X X^T exactly (f32 rounding) when the numerical rank fits the cap, and
    otherwise the residual is a Schur complement whose diagonal sits below the last pivot; the span of the rows holds the dominant
    left singular vectors of X in the graded sense || (I - Q Q^T) U_k s_k || <= 3e-6 s_1 for the first 32 directions (the spectrum of a
    truncation input: five decades over the first 32, another until 64)."""
    capi = _capi()
    rng = np.random.default_rng(n + rank)
    nb = 3
    U, _ = np.linalg.qr(rng.standard_normal((n, n)))
    V, _ = np.linalg.qr(rng.standard_normal((K, min(K, n))))
    r = min(rank, K, n)
    s = np.zeros(min(K, n))
    s[:r] = np.concatenate([np.logspace(0, -5, 32), np.logspace(-5, -6, 32)[1:], np.logspace(-6, -6.9, max(r - 63, 1))])[:r]
    X = np.stack([((U[:, :len(s)] * s) @ V.T) * (1 + 0.1 * b) for b in range(nb)]).astype(np.float32)
    nlive = np.array([n, n - 3, n], dtype=np.int32)
    R, ml = capi.diag_chol_pivot(X, nlive, kcap)
    for b in range(nb):
        nl = int(nlive[b])
        Xb = X[b, :nl].astype(np.float64)
        G = Xb @ Xb.T
        sc = np.max(np.diag(G))
        assert 0 < ml[b] <= kcap
        B = R[b, :ml[b]].astype(np.float64)
        assert np.all(np.isfinite(B))
        assert np.all(B[:, nl:] == 0)
        piv = np.max(np.abs(B), axis=1)      # a row's largest entry is its pivot (the diagonal of the permuted factor)
        B = B[:, :nl]
        res = G / sc - B.T @ B
        if r + 8 <= kcap:
            assert np.max(np.abs(res)) < 3e-6
            assert abs(int(ml[b]) - r) <= 4, (ml[b], r)
        else:
            assert ml[b] >= kcap - 4
            assert np.max(np.diag(res)) <= np.min(piv[max(0, ml[b] - 8):ml[b]]) ** 2 * 16 + 3e-7       # what is left sits at the level of the last pivots taken
        Us, ss, _ = np.linalg.svd(Xb, full_matrices=False)
        k = min(32, r)
        Q, _ = np.linalg.qr(B.T)
        lost = Us[:, :k] * ss[:k] - Q @ (Q.T @ (Us[:, :k] * ss[:k]))
        assert np.linalg.norm(lost, 2) / ss[0] < 3e-6, np.linalg.norm(lost, 2) / ss[0]


@pytest.mark.parametrize("k,ln", [(32, 256), (24, 144), (32, 64), (7, 36), (1, 256)])
def test_rows_qr_orthonormal_span(k, ln):
    """rows_qr_kernel (round 6): rows sigma_q v_q^T (five decades) contaminated by the dominant directions at 1e-7 sigma_1 (what
    V' = U^T M looks like) come out orthonormal to f32 storage accuracy and span the same space; a row below the liveness floor
    (2 * 8 eps32 |X|_F) or linearly dependent on the rows before it is dropped, the live rows come first, the rest of V is zero."""
    capi = _capi()
    rng = np.random.default_rng(k * 1000 + ln)
    nb = 4
    Q, _ = np.linalg.qr(rng.standard_normal((ln, min(ln, k + 8))))
    s = np.logspace(0, -5, k) if k > 1 else np.array([1.0])
    X = np.zeros((nb, k, ln))
    for b in range(nb):
        R = Q[:, :k].T * s[:, None]
        R = R + 1e-7 * rng.standard_normal((k, 1)) * Q[:, 0][None, :] + 1e-7 * rng.standard_normal((k, 1)) * Q[:, min(1, k - 1)][None, :]
        X[b] = R * (1.0 + b)
    klive = np.array([k, k, max(1, k - 2), k], dtype=np.int32)
    if k >= 4:
        X[1, k - 1] = X[1, 0] * 1e-3            # a dependent row: dropped
        X[3, k - 1] *= 1e-3 / 5                 # below the floor (2e-6 |X|_F): dropped
    V, kl = capi.diag_rows_qr(X, klive)
    X32 = X.astype(np.float32).astype(np.float64)
    for b in range(nb):
        n_in = int(klive[b])
        expect = n_in - (1 if (k >= 4 and b in (1, 3)) else 0)
        assert kl[b] == expect, (b, kl[b], expect)
        Vb = V[b, :kl[b]].astype(np.float64)
        assert np.all(V[b, kl[b]:] == 0)
        assert np.max(np.abs(Vb @ Vb.T - np.eye(kl[b]))) < 5e-7
        # same span: every kept input row is reproduced by its projection, relative to its own norm (the dropped ones: excluded)
        rows = [a for a in range(n_in) if not (k >= 4 and b in (1, 3) and a == k - 1)]
        for a in rows:
            x = X32[b, a]
            assert np.linalg.norm(x - Vb.T @ (Vb @ x)) < 2e-6 * np.linalg.norm(x) + 2e-7 * np.linalg.norm(X32[b, 0]), (b, a)
