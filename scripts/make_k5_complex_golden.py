"""Golden vectors for the reference's COMPLEX 4x4 D=8 Heisenberg fixture (tests/slow_tests/test_data/tps_square_heisenberg4x4D8Complex, the
QLTEN_Complex twin of K5; used by the reference's slow test tests/slow_tests/test_boson_mc_peps_measure.cpp:36,55-62,97-101): the state is
contracted DENSELY (no boundary MPS, no truncation) to its 65 536 amplitudes; written: the exact energy <psi|H|psi>/<psi|psi> of the OBC
Heisenberg model, and for 32 seeded Sz = 0 configurations the amplitude and the local energy with the reference's convention
E_loc(S) = sum_S' H_SS' conj(psi(S') / psi(S)) (square_spin_onehalf_xxz_obc.h:96-100).  NumPy only; reads the fixture from tests/golden.
usage: python scripts/make_k5_complex_golden.py  -> tests/golden/k5_complex_dense.json"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import qlten_io  # noqa: E402  (the loader of the .qlten format; test infrastructure, as this script is)

L = 4
s = qlten_io.load_sitps(os.path.join(ROOT, "tests", "golden", "ref_fixtures", "tps_square_heisenberg4x4D8Complex"), complex_data=True)


def dense_state():
    # X[P, (down bonds of the finished part of the current row), h, (up bonds still open from the row above)]
    X = np.ones((1, 1, 1, 1, 1, 1), dtype=np.complex128)          # P, b0, b1, b2, b3, h   (b_c: vertical bond at column c, h: horizontal)
    for r in range(L):
        for c in range(L):
            T = np.stack([s[r][c][p] for p in range(2)], axis=0)    # p, l, d, r, u
            # contract h with l and b_c with u
            X = np.moveaxis(X, 1 + c, -1)                            # P, (other b), h, b_c
            Xs = X.shape
            Y = np.tensordot(X.reshape(-1, Xs[-2] * Xs[-1]), T.transpose(1, 4, 0, 2, 3).reshape(Xs[-2] * Xs[-1], -1), axes=1)
            Y = Y.reshape(Xs[:-2] + (2, T.shape[2], T.shape[3]))     # P, (other b: 3 of them), p, d, r
            Y = np.moveaxis(Y, -3, 1)                                # P, p, other b.., d, r
            Y = Y.reshape((Y.shape[0] * 2,) + Y.shape[2:])           # (P p), other b (3), d, r
            Y = np.moveaxis(Y, -2, 1 + c)                            # (P p), b0..b3 with d in slot c, r(=h)
            X = Y
        assert X.shape[-1] == 1
    assert X.shape[1:] == (1, 1, 1, 1, 1)
    return X.reshape(-1)                                             # index = sum_k p_k 2^(15 - k), k = row-major site


psi = dense_state()
N = L * L
idx = np.arange(1 << N)
bits = ((idx[:, None] >> (N - 1 - np.arange(N))[None, :]) & 1)       # bits[:, k] = state of site k (row-major)
bonds = [(r * L + c, r * L + c + 1) for r in range(L) for c in range(L - 1)] + [(r * L + c, (r + 1) * L + c) for r in range(L - 1) for c in range(L)]
Hpsi = np.zeros_like(psi)
for a, b in bonds:
    same = bits[:, a] == bits[:, b]
    Hpsi += np.where(same, 0.25, -0.25) * psi
    flip = idx ^ ((1 << (N - 1 - a)) | (1 << (N - 1 - b)))
    Hpsi += np.where(same, 0.0, 0.5) * psi[flip]
norm = np.vdot(psi, psi).real
energy = np.vdot(psi, Hpsi) / norm
sz0 = bits.sum(axis=1) == N // 2
energy_sz0 = np.vdot(psi[sz0], Hpsi[sz0]) / np.vdot(psi[sz0], psi[sz0]).real      # the sector the exact-sum / MC runs live in
rng = np.random.default_rng(20260501)
pick = rng.choice(np.flatnonzero(sz0), size=32, replace=False)
out = {"source": "dense contraction of tests/golden/ref_fixtures/tps_square_heisenberg4x4D8Complex (scripts/make_k5_complex_golden.py)",
       "energy": [float(energy.real), float(energy.imag)], "energy_sz0_sector": [float(energy_sz0.real), float(energy_sz0.imag)],
       "weight_outside_sz0": float(1.0 - np.vdot(psi[sz0], psi[sz0]).real / norm),
       "configs": bits[pick].reshape(-1, L, L).tolist(),
       "amplitude": [[float(z.real), float(z.imag)] for z in psi[pick]],
       "e_loc": [[float(z.real), float(z.imag)] for z in np.conj(Hpsi[pick] / psi[pick])]}
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "k5_complex_dense.json"), "w"), indent=0)
print("E =", energy, " E(Sz=0 sector) =", energy_sz0, " weight outside Sz=0:", out["weight_outside_sz0"], " max |Im psi| / max |psi| =", np.abs(psi.imag).max() / np.abs(psi).max())
