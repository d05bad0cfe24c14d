"""Algorithmic flop model of one EvaluateAmplitude: the reference op sequence
of BMPS::MultiplyMPO with SVD compression (include/qlpeps/one_dim_tn/boundary_mps/bmps_impl.h:756-862,
:225-263), real arithmetic, truncation SVD(chi, chi, 0).  Used by bench.py for the roofline figures;
it is what `roofline.achieved` and the whole-job TFLOP/s are computed from, independent of the flops
the device algorithm actually executes."""


def reference_flops(L, D, chi):
    """Returns dict(gemm, qr, svd, total) flops per amplitude (L-1 row absorptions; the (L-2) BTen
    steps and the trace are ~1% and not counted)."""
    gemm = qr = svd = 0.0
    for k in range(L - 1):                     # k rows already absorbed
        def bond(i, kk):
            if i == 0 or i == L:
                return 1
            return min(chi, D ** kk if kk < 40 else chi, D ** i if i < 40 else chi, D ** (L - i) if L - i < 40 else chi)
        p = 1 if k == 0 else D                 # leg contracted with the BMPS physical leg
        u = D                                  # leg that becomes the new physical leg
        m = [1] * (L + 1)
        for i in range(L):
            dl = 1 if i == 0 else D
            dr = 1 if i == L - 1 else D
            bi, bi1 = bond(i, k), bond(i + 1, k)
            gemm += 2.0 * p * bi1 * bi * m[i] * dl                       # G1
            gemm += 2.0 * (bi1 * m[i]) * (dl * p) * (dr * u)             # G2
            if i < L - 1:
                R, C = m[i] * u, dr * bi1
                if R < C:
                    R, C = C, R
                qr += 2.0 * (2.0 * R * C * C - 2.0 / 3.0 * C ** 3)
                m[i + 1] = min(m[i] * u, dr * bi1)
        kk = [1] * (L + 1)
        for i in range(L - 1, 0, -1):
            r, c = m[i], u * kk[i + 1]
            kk[i] = min(chi, r, c)
            if r < c:
                r, c = c, r
            svd += 4.0 * r * c * c + 22.0 * c ** 3
            gemm += 2.0 * m[i - 1] * u * m[i] * kk[i]                    # G3
    return {"gemm": gemm, "qr": qr, "svd": svd, "total": gemm + qr + svd}
