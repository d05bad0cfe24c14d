"""K7: the projected fermionic t-J network of the reference's BMPS test (tests/test_2d_tn/test_bmps_contractor.cpp:688-865) as
DATA + a restatement of its fixture constructor CreateFiniteSizeOBCtJTPS (:770-846): a 20 x 24 OBC state tiled from the two
iPEPS tensors ipeps_tJ_t{a,b}_doping0.125.qlten (D = 4, fZ2-graded, physical (up, down | empty)), every boundary leg cut to
dimension one by its dominant singular vector, and the fixed configuration of :728-749.

Not reproduced: the element-wise fermionic signs of the graded Transpose calls around the boundary SVDs (they flip signs
of boundary-tensor components only); what the reference test asserts -- the magnitudes of the 21 routes agree to 1e-7 --
holds for any parity-even network and is what the tests built on this module check, on the oracle and on the device."""
import os

import numpy as np

from peps_amd import fermion

ROWS, COLS, DB_MIN, DB_MAX = 20, 24, 16, 50

CONFIG = np.array([
    [1, 0, 1, 2, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0],
    [0, 1, 1, 0, 0, 2, 1, 2, 0, 2, 0, 0, 2, 1, 0, 2, 0, 1, 0, 2, 0, 0, 2, 1],
    [1, 2, 0, 0, 1, 0, 1, 0, 1, 1, 0, 1, 2, 0, 2, 1, 1, 0, 1, 0, 1, 0, 1, 0],
    [1, 0, 0, 1, 1, 2, 0, 1, 0, 2, 0, 0, 1, 1, 0, 1, 0, 1, 0, 1, 1, 2, 0, 1],
    [1, 1, 2, 0, 1, 1, 1, 0, 2, 1, 2, 2, 0, 0, 1, 0, 1, 0, 1, 1, 0, 1, 1, 0],
    [2, 0, 0, 1, 2, 0, 0, 1, 1, 0, 2, 1, 1, 0, 0, 1, 0, 1, 0, 0, 1, 0, 1, 1],
    [1, 0, 1, 0, 1, 0, 1, 2, 0, 1, 0, 0, 0, 0, 1, 0, 1, 0, 1, 2, 1, 0, 1, 0],
    [0, 1, 0, 1, 1, 1, 0, 1, 0, 2, 2, 1, 1, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0, 1],
    [1, 0, 1, 0, 2, 0, 1, 0, 1, 0, 1, 0, 2, 0, 1, 0, 1, 0, 1, 0, 1, 1, 2, 0],
    [0, 1, 0, 1, 1, 1, 0, 1, 0, 0, 1, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0, 0, 0, 1],
    [1, 0, 1, 0, 0, 0, 1, 0, 0, 1, 2, 1, 0, 1, 0, 2, 1, 0, 1, 2, 1, 2, 0, 2],
    [0, 1, 0, 1, 2, 1, 0, 1, 0, 1, 1, 1, 0, 1, 0, 2, 1, 2, 0, 0, 0, 1, 2, 1],
    [1, 0, 1, 0, 1, 0, 1, 0, 1, 2, 0, 1, 1, 0, 1, 0, 1, 0, 0, 1, 0, 2, 2, 1],
    [0, 1, 0, 1, 0, 2, 0, 1, 0, 1, 0, 0, 0, 0, 2, 1, 0, 2, 1, 0, 2, 1, 0, 0],
    [1, 0, 1, 0, 1, 1, 1, 1, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0, 0, 1, 2, 1, 0, 2],
    [0, 1, 0, 1, 0, 2, 0, 0, 0, 1, 0, 1, 0, 1, 0, 0, 1, 2, 2, 1, 0, 1, 2, 1],
    [1, 0, 1, 0, 1, 0, 1, 0, 2, 0, 1, 0, 1, 0, 1, 0, 2, 1, 1, 2, 1, 0, 0, 0],
    [0, 1, 1, 2, 0, 1, 0, 1, 1, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0, 0, 0, 1, 1, 1],
    [1, 0, 2, 0, 1, 1, 0, 1, 0, 0, 1, 0, 1, 1, 1, 0, 1, 0, 1, 2, 1, 0, 0, 0],
    [0, 1, 0, 1, 2, 0, 0, 2, 1, 2, 0, 1, 0, 0, 0, 1, 0, 1, 0, 1, 0, 1, 1, 2]], dtype=np.int32)


def _trim(t, par, axis):
    """cut leg `axis` to dimension one with the dominant singular vector of (axis | the rest): SVD(.., Dmin = Dmax = 1)"""
    m = np.moveaxis(t, axis, 0)
    shp = m.shape
    u, s, vt = np.linalg.svd(m.reshape(shp[0], -1), full_matrices=False)
    v = vt[0]
    v = v * np.sign(v[np.argmax(np.abs(v))])
    # parity sector of the kept singular value = parity of the leg states carrying u[:, 0]
    sec = set(int(p) for p, w in zip(par[axis], np.abs(u[:, 0])) if w > 1e-12)
    assert len(sec) == 1, "dominant singular vector mixes parity sectors"
    out = np.moveaxis(v.reshape((1,) + shp[1:]), 0, axis)
    npar = list(par)
    npar[axis] = np.array([sec.pop()], dtype=np.int64)
    # the kept vector lives in one parity block; what the SVD leaves in the other block is rounding noise (1e-17): project
    tot = sum(np.asarray(npar[k]).reshape([-1 if j == k else 1 for j in range(out.ndim)]) for k in range(out.ndim))
    out = np.where(tot % 2 == 0, out, 0.0)
    return out, npar


def build_state(fixtures_dir, rows=ROWS, cols=COLS):
    """-> (FermionState with four state slots per site, nf = [1, 1, 0, 0]; relabel(config) -> device-side labels).

    Two of the four boundary cuts land in the ODD sector of their bond (the reference only prints a warning there,
    :789-791): the dim-1 boundary leg then carries a fermion.  A dangling odd mode sitting at a site merges with that site's
    physical mode (the relative order of two modes of one site is a sign per configuration, not per bond index), i.e. the
    site's states have the opposite parity: (up, down) even, empty odd.  FermionState labels parities per state slot, so such
    a site stores empty in slot 0 (odd) and up / down in slots 2, 3 (even); ordinary sites use slots 0, 1 (up, down) and 2
    (empty).  relabel() maps the reference's configuration values {0: up, 1: down, 2: empty} to the slots."""
    ta, pa, da = fermion._read_qlten_z2(os.path.join(fixtures_dir, "ipeps_tJ_ta_doping0.125.qlten"))
    tb, pb, db = fermion._read_qlten_z2(os.path.join(fixtures_dir, "ipeps_tJ_tb_doping0.125.qlten"))
    assert ta.shape == (4, 4, 4, 4, 3) and tuple(da) == (-1, 1, 1, -1, -1) and list(pa[4]) == [1, 1, 0]
    tensors = [[None] * cols for _ in range(rows)]
    pars = [[None] * cols for _ in range(rows)]
    slot = np.zeros((rows, cols, 3), dtype=np.int32)
    for r in range(rows):
        for c in range(cols):
            t, par = (ta, list(pa)) if (r + c) % 2 == 0 else (tb, list(pb))
            if r == 0: t, par = _trim(t, par, 3)                 # :786-794  UP leg
            elif r == rows - 1: t, par = _trim(t, par, 1)        # :795-803  DOWN leg
            if c == 0: t, par = _trim(t, par, 0)                 # :808-813  LEFT leg
            elif c == cols - 1: t, par = _trim(t, par, 2)        # :814-822  RIGHT leg
            t = t / np.linalg.norm(t) * 3.0                      # NormalizeAllSite, *= 3.0 (:828-829): a uniform scale per site
            flip = 0
            for k in range(4):
                if len(par[k]) == 1 and int(par[k][0]) == 1:      # odd dangling boundary leg -> into the site's state parity
                    flip ^= 1
                    par[k] = np.array([0], dtype=np.int64)
            zero = np.zeros(t.shape[:4])
            if flip: comps, slot[r, c] = [t[..., 2], zero, t[..., 0], t[..., 1]], (2, 3, 0)
            else: comps, slot[r, c] = [t[..., 0], t[..., 1], t[..., 2], zero], (0, 1, 2)
            tensors[r][c] = comps
            pars[r][c] = tuple(par[:4])
    state = fermion.FermionState(tensors, pars, [1, 1, 0, 0])

    def relabel(config):
        cfg = np.asarray(config)
        return np.take_along_axis(slot, cfg[..., None], axis=2)[..., 0].astype(np.int32)
    return state, relabel
