"""Z2-graded (fermionic) tensors -- TEST INFRASTRUCTURE ONLY (oracle), never imported by peps_amd/.

The reference keeps every fermionic sign rule inside TensorToolkit (`qlten::Contract`, `Transpose`,
`FuseIndex` on `QLTensor<T, fZ2QN>`), which is absent here.  This module restates a Z2-graded tensor
algebra from first principles and PINS its two free conventions on the reference's own fixtures
(tests/test_oracle_fermion.py): with
  * Koszul sign (-1)^{p_a p_b} for every exchange of two legs (Transpose),
  * contraction of the legs a_1..a_n of A with b_1..b_n of B in NESTED order (a_1 .. a_n b_n .. b_1),
  * an adjacent pair [leg of A][leg of B] evaluates without sign when A's leg is OUT (+1) and B's is IN (-1),
    and with (-1)^p when A's leg is IN and B's is OUT,
the six 2x2 spinless-fermion known answers of tests/test_algorithm/test_exact_summation_evaluator.cpp:268-470
(exact ground-state energies -4.2, -2, -5 and simple-update energies -4.1879072654, -1.98218053854,
-4.98966397657) are reproduced to all printed digits; the other three combinations of the two binary
choices miss them by O(1).

Site tensors of a fermionic SplitIndexTPS are rank 5: legs (L, D, R, U, P) with directions (IN, OUT, OUT, IN, IN);
P is the 1-dimensional parity leg that replaces the projected physical leg (odd = occupied = state 0 of
square_spinless_fermion.h:35-37); every stored tensor is parity even.
"""
import numpy as np


class GT:
    """dense Z2-graded tensor: arr, par[k] = parity of each index value of leg k, dirs[k] = +1 OUT / -1 IN"""

    def __init__(self, arr, par, dirs):
        self.arr = arr
        self.par = [np.asarray(p, dtype=int) for p in par]
        self.dirs = list(dirs)

    @property
    def rank(self):
        return self.arr.ndim

    def is_even(self, tol=1e-14):
        tot = 0
        for k in range(self.rank):
            sh = [1] * self.rank
            sh[k] = -1
            tot = tot + self.par[k].reshape(sh)
        return not np.any((np.abs(self.arr) > tol) & (tot % 2 == 1))

    def _pvec(self, k):
        sh = [1] * self.rank
        sh[k] = -1
        return self.par[k].reshape(sh)

    def transpose(self, perm):
        perm = list(perm)
        posn = {ax: i for i, ax in enumerate(perm)}
        sign = np.ones([1] * self.rank)
        for a in range(self.rank):
            for b in range(a + 1, self.rank):
                if posn[a] > posn[b]:
                    sign = sign * (1 - 2 * ((self._pvec(a) * self._pvec(b)) % 2))
        return GT((self.arr * sign).transpose(perm), [self.par[p] for p in perm], [self.dirs[p] for p in perm])

    @staticmethod
    def contract(A, axA, B, axB):
        axA, axB = list(axA), list(axB)
        restA = [i for i in range(A.rank) if i not in axA]
        restB = [i for i in range(B.rank) if i not in axB]
        n = len(axA)
        A2 = A.transpose(restA + axA)
        B2 = B.transpose(axB[::-1] + restB)          # nested: a_1 .. a_n b_n .. b_1
        a_arr = A2.arr
        for k in range(n):
            la, lb = len(restA) + k, n - 1 - k
            assert A2.dirs[la] == -B2.dirs[lb], "contracted legs must have opposite directions"
            assert np.array_equal(A2.par[la], B2.par[lb]), "parity structure mismatch"
            if A2.dirs[la] == -1:                    # [IN][OUT] pair: supertrace sign
                sh = [1] * a_arr.ndim
                sh[la] = -1
                a_arr = a_arr * (1 - 2 * A2.par[la]).reshape(sh)
        res = np.tensordot(a_arr, B2.arr, axes=([len(restA) + k for k in range(n)], [n - 1 - k for k in range(n)]))
        return GT(res, A2.par[:len(restA)] + B2.par[n:], A2.dirs[:len(restA)] + B2.dirs[n:])


def load_qlten_z2(path, complex_data=False):
    """.qlten file with fZ2QN sectors -> GT (dense embedding, sectors in file order)."""
    with open(path, "rb") as f:
        buf = f.read()
    pos = 0

    def tok():
        nonlocal pos
        e = buf.index(b"\n", pos)
        t = buf[pos:e]
        pos = e + 1
        return t

    rank = int(tok())
    legs = []
    for _ in range(rank):
        nsec = int(tok())
        secs = []
        for _s in range(nsec):
            qn = int(tok()); tok(); deg = int(tok()); tok()
            secs.append((qn, deg))
        d = int(tok()); dim = int(tok()); tok()
        assert sum(x[1] for x in secs) == dim
        legs.append((secs, d))
    nblocks = int(tok())
    blocks = [[int(tok()) for _ in range(rank)] for _ in range(nblocks)]
    shape = tuple(sum(s[1] for s in secs) for secs, _ in legs)
    out = np.zeros(shape, dtype=np.complex128 if complex_data else np.float64)
    offs = [np.concatenate([[0], np.cumsum([s[1] for s in secs])]) for secs, _ in legs]
    item = 16 if complex_data else 8
    for c in blocks:
        bshape = tuple(legs[k][0][c[k]][1] for k in range(rank))
        n = int(np.prod(bshape))
        data = np.frombuffer(buf, dtype="<c16" if complex_data else "<f8", count=n, offset=pos)
        pos += n * item
        sl = tuple(slice(offs[k][c[k]], offs[k][c[k] + 1]) for k in range(rank))
        out[sl] = data.reshape(bshape)
    par = [np.concatenate([np.full(deg, qn % 2, dtype=int) for qn, deg in secs]) for secs, _ in legs]
    return GT(out, par, [d for _, d in legs])


def graded_amplitude_exact(sitps, cfg):
    """<S|Psi> by the graded contraction of the whole network (small lattices): rows are contracted
    left to right, then top to bottom; the parity legs end in row-major order.  sitps[r][c][s] = GT."""
    Ly, Lx = len(sitps), len(sitps[0])
    rows = []
    for r in range(Ly):
        cur = sitps[r][0][cfg[r][0]]
        labels = [("L", r, 0), ("D", r, 0), ("R", r, 0), ("U", r, 0), ("P", r, 0)]
        for c in range(1, Lx):
            t = sitps[r][c][cfg[r][c]]
            ia = labels.index(("R", r, c - 1))
            cur = GT.contract(cur, [ia], t, [0])
            labels = [x for x in labels if x != ("R", r, c - 1)] + [("D", r, c), ("R", r, c), ("U", r, c), ("P", r, c)]
        rows.append((cur, labels))
    cur, labels = rows[0]
    for r in range(1, Ly):
        nxt, nl = rows[r]
        axa = [labels.index(("D", r - 1, c)) for c in range(Lx)]
        axb = [nl.index(("U", r, c)) for c in range(Lx)]
        cur = GT.contract(cur, axa, nxt, axb)
        labels = [x for i, x in enumerate(labels) if i not in axa] + [x for i, x in enumerate(nl) if i not in axb]
    pl = [labels.index(("P", r, c)) for r in range(Ly) for c in range(Lx)]
    others = [i for i in range(cur.rank) if i not in pl]
    cur = cur.transpose(others + pl)
    assert cur.arr.size == 1
    return cur.arr.ravel()[0]
