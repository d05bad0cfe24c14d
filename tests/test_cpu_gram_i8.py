"""The arithmetic of the exact-integer Gram kernels (peps_amd/csrc/gram_i8.h), restated in NumPy and checked for the invariants the
kernel relies on -- no GPU: a float32 block becomes integers |n| <= 2^22 by ONE power of two per line and block of 64 (round to
nearest even through the float add of 1.5 * 2^23), n = d0 2^16 + d1 2^8 + d2 with three signed bytes (`(n + 0x808080) ^ 0x808080`),
the nine byte products accumulate exactly in int32, the classes merge in int32 (U = 256 S0 + S1, W = 256 S2 + S3) and assemble
exactly in float64 (T = 2^24 U + 2^8 W + S4 < 2^53): the result is the float64 Gram of the fixed-point image.  The GPU tests
(tests/test_gpu_kernels.py) compare the kernels with the same image."""
import numpy as np
import pytest


def digits_of_block(blk):
    """blk [64, n] float32 -> (d [3, 64, n] int8, e [n] int): x ~ n 2^(e - 148), n = d0 2^16 + d1 2^8 + d2 (the kernel's `lay`)"""
    blk = np.asarray(blk, dtype=np.float32)
    m = np.max(np.abs(blk), axis=0).astype(np.float32)
    e = np.maximum(30, (m.view(np.uint32) >> 23).astype(np.int64))
    sc = ((275 - e).astype(np.uint32) << 23).view(np.float32)                # 2^(148 - e)
    y = (blk.astype(np.float64) * sc.astype(np.float64) + 12582912.0).astype(np.float32)      # v_fma_f32: one rounding, to nearest even
    yb = y.view(np.uint32)
    w = (yb + np.uint32((0x808080 - 0x4B400000) & 0xFFFFFFFF)) ^ np.uint32(0x808080)
    d = np.stack([((w >> 16) & 0xFF).astype(np.uint8).view(np.int8), ((w >> 8) & 0xFF).astype(np.uint8).view(np.int8),
                  (w & 0xFF).astype(np.uint8).view(np.int8)])
    assert np.all((w >> 24) == 0)
    return d, e


def gram_i8(P):
    """float64 Gram of the columns of P [K, n] float32 through the integer pipeline of the kernel"""
    K, n = P.shape
    G = np.zeros((n, n))
    for k0 in range(0, K, 64):
        blk = np.zeros((64, n), dtype=np.float32)
        blk[:min(64, K - k0)] = P[k0:k0 + 64]
        d, e = digits_of_block(blk)
        di = d.astype(np.int32)
        S = [np.zeros((n, n), dtype=np.int32) for _ in range(5)]
        for a in range(3):
            for b in range(3):
                S[a + b] += di[a].T @ di[b]                       # v_mfma_i32_16x16x64_i8: exact int32
        for c in range(5):
            assert np.max(np.abs(S[c].astype(np.int64))) < 2 ** 22 + 2 ** 21
        U = (S[0] << 8) + S[1]
        W = (S[2] << 8) + S[3]
        assert np.max(np.abs(U.astype(np.int64))) < 2 ** 31 and np.max(np.abs(W.astype(np.int64))) < 2 ** 31
        assert np.array_equal(U.astype(np.int64), S[0].astype(np.int64) * 256 + S[1]) and np.array_equal(W.astype(np.int64), S[2].astype(np.int64) * 256 + S[3])
        T = U.astype(np.float64) * 16777216.0 + (W.astype(np.float64) * 256.0 + S[4].astype(np.float64))
        assert np.max(np.abs(T)) < 2.0 ** 53
        G += np.ldexp(T, (e[:, None] + e[None, :] - 296).astype(np.int64))
    return G


def image(P):
    P = np.asarray(P, dtype=np.float32)
    out = np.zeros(P.shape)
    for k0 in range(0, P.shape[0], 64):
        blk = P[k0:k0 + 64]
        m = np.max(np.abs(blk), axis=0).astype(np.float32)
        e = np.maximum(30, (m.view(np.uint32) >> 23).astype(np.int64))
        out[k0:k0 + 64] = np.ldexp(np.rint(np.ldexp(blk.astype(np.float64), 148 - e)), e - 148)
    return out


@pytest.mark.parametrize("K,n,seed", [(64, 16, 0), (200, 48, 1), (333, 32, 2), (1, 16, 3)])
def test_digits_reproduce_the_integer_and_the_gram_is_exact(K, n, seed):
    rng = np.random.default_rng(seed)
    P = (rng.standard_normal((K, n)) * np.logspace(0, -6, n)[None, :] * np.exp(3 * rng.standard_normal((K, 1)))).astype(np.float32)
    P[:, 3] = 0.0                                            # a dead column
    if K > 70:
        P[64:128, 5] *= np.float32(1e-30)                    # a block far below the others: its own exponent (clamped at 2^-97)
    for k0 in range(0, K, 64):
        blk = np.zeros((64, n), dtype=np.float32)
        blk[:min(64, K - k0)] = P[k0:k0 + 64]
        d, e = digits_of_block(blk)
        nn = d[0].astype(np.int64) * 65536 + d[1].astype(np.int64) * 256 + d[2].astype(np.int64)
        assert np.max(np.abs(nn)) <= 2 ** 22
        want = np.rint(np.ldexp(blk.astype(np.float64), (148 - e)[None, :]))
        assert np.array_equal(nn, want.astype(np.int64))      # the float add rounds to nearest even, as rint does
    G = gram_i8(P)
    Pi = image(P)
    ref = Pi.T @ Pi
    scale = np.sqrt(np.outer(np.diag(ref), np.diag(ref))) + 1e-300
    assert np.max(np.abs(G - ref) / scale) < 1e-14
    exact = P.astype(np.float64).T @ P.astype(np.float64)
    live = np.diag(exact) > 0
    sc2 = np.sqrt(np.outer(np.diag(exact)[live], np.diag(exact)[live]))
    assert np.max(np.abs(G[np.ix_(live, live)] - exact[np.ix_(live, live)]) / sc2) < 1.5e-6
    assert np.all(G[3] == 0) and np.all(G[:, 3] == 0)
