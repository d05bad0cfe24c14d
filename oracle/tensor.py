"""Dense restatement of the TensorToolkit (qlten) primitives the hot path calls.

qlten is NOT vendored in /root/reference; semantics are inferred
from the call sites in include/qlpeps/one_dim_tn/boundary_mps/bmps_impl.h and
include/qlpeps/two_dim_tn/tensor_network_2d/bmps/impl/*.h.
Oracle = test infrastructure only (see oracle/__init__.py).
"""
import numpy as np


def contract(a, a_axes, b, b_axes):
    """qlten::Contract(&A, {a...}, &B, {b...}, &C): C = A's free legs in order, then B's.

    Call sites: bmps_impl.h:838, bmps_contractor_grow.h:178-180, bmps_contractor_trace.h:84-85.
    """
    return np.tensordot(a, b, axes=(list(a_axes), list(b_axes)))


def contract_cyclic(a, b, a0, b0, n):
    """qlten::Contract<T,QN,a_tail,b_head>(A, B, a0, b0, n, C).

    Contract the n cyclically consecutive legs of A starting at a0 with those of B starting
    at b0.  C = [A's remaining legs in cyclic order starting at a0+n] ++ [B's remaining legs
    in cyclic order starting at b0+n].  The bool template flags only pick a transposition
    strategy.  Call sites: bmps_impl.h:806-807, bmps_contractor_grow.h:577-578,
    bmps_contractor_trace.h:82-83.
    """
    ra, rb = a.ndim, b.ndim
    a_ctr = [(a0 + k) % ra for k in range(n)]
    b_ctr = [(b0 + k) % rb for k in range(n)]
    a_free = [(a0 + n + k) % ra for k in range(ra - n)]
    b_free = [(b0 + n + k) % rb for k in range(rb - n)]
    at = np.transpose(a, a_free + a_ctr)
    bt = np.transpose(b, b_ctr + b_free)
    sa = at.shape[:ra - n]
    sb = bt.shape[n:]
    k = int(np.prod(at.shape[ra - n:], dtype=np.int64))
    c = at.reshape(-1, k) @ bt.reshape(k, -1)
    return c.reshape(sa + sb)


def index_combine(d1, d2, dtype=np.float64):
    """qlten::IndexCombine(idx1, idx2, dir): rank-3 combiner (d1, d2, d1*d2), identity map.

    Call sites: bmps_impl.h:771, :831 (both legs have dim 1 on an OBC boundary).
    """
    c = np.zeros((d1, d2, d1 * d2), dtype=dtype)
    for i in range(d1):
        for j in range(d2):
            c[i, j, i * d2 + j] = 1.0
    return c


def qr(a, ldims):
    """qlten::QR(pA, ldims, div, pQ, pR): first `ldims` legs = rows; Q gets a new trailing
    bond, R a new leading bond (economy size).  Call site: bmps_impl.h:821."""
    lshape, rshape = a.shape[:ldims], a.shape[ldims:]
    m = a.reshape(int(np.prod(lshape, dtype=np.int64)), -1)
    q, r = np.linalg.qr(m, mode="reduced")
    k = q.shape[1]
    return q.reshape(lshape + (k,)), r.reshape((k,) + rshape)


def truncation_rank(s, trunc_err, dmin, dmax):
    """Kept dimension and actual truncation error of qlten::SVD(trunc_err, Dmin, Dmax).

    PARITY UNPINNED for trunc_err > 0 beyond one instance (the rule lives in TensorToolkit; K8 -- SVD(8, 16, 1e-15) inside the
    reference's MCPEPSMeasurer regression vector -- is reproduced to 1e-15 with it): singular values are
    discarded from the smallest while the kept count exceeds Dmax, or exceeds Dmin and the
    accumulated discarded weight / total weight stays strictly below trunc_err.  With
    trunc_err == 0 this keeps min(Dmax, len(s)) values, the only case throughput runs use
    (as test_exact_summation_evaluator.cpp:355 does).
    """
    n = len(s)
    total = float(np.sum(s * s))
    kept = n
    err = 0.0
    while kept > 0:
        if kept <= dmin and kept <= dmax:
            break
        w = float(s[kept - 1] ** 2) / total if total > 0 else 0.0
        if kept > dmax or (kept > dmin and err + w < trunc_err):
            err += w
            kept -= 1
        else:
            break
    return max(kept, 1), err


def svd_trunc(a, ldims, trunc_err, dmin, dmax):
    """qlten::SVD(pA, ldims, div, trunc_err, Dmin, Dmax, pU, pS, pVt, &err, &D).

    Returns u (lshape.., k), s (k,), vt (k, rshape..), actual_trunc_err, k.
    Call site: bmps_impl.h:235-238."""
    lshape, rshape = a.shape[:ldims], a.shape[ldims:]
    m = a.reshape(int(np.prod(lshape, dtype=np.int64)), -1)
    try:
        u, s, vt = np.linalg.svd(m, full_matrices=False)
    except np.linalg.LinAlgError:  # pragma: no cover - gesdd non-convergence fallback
        import scipy.linalg
        u, s, vt = scipy.linalg.svd(m, full_matrices=False, lapack_driver="gesvd")
    k, err = truncation_rank(s, trunc_err, dmin, dmax)
    return (u[:, :k].reshape(lshape + (k,)), s[:k].copy(),
            vt[:k].reshape((k,) + rshape), err, k)
