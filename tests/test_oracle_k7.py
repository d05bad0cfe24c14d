"""K7 in the oracle (CPU suite): the reference's 20 x 24 projected t-J network (test_bmps_contractor.cpp:688-865; tests/k7_tj.py)
through the decorated-tensor formulation of fermionic contractions.  The 21 routes of Contract2DTNUsingBMPSContractor must agree
in magnitude to 1e-7 with BMPSTruncateParams(16, 50, 1e-15) -- the reference's assertion (:855-858) -- where the row passes
run on the row-major decorated network and the column passes on the column-major one: the first pin of the decoration
identity beyond 2 x 2 / 3 x 4 lattices."""
import numpy as np

import k1_routes
import k7_tj
from oracle.bmps import BMPSTruncateParams
from oracle.contractor import BMPSContractor, TensorNetwork2D
from peps_amd import fermion


def decorated_tn(state, ext_flat, cfg_ext):
    tn = TensorNetwork2D(state.rows, state.cols)
    for r in range(state.rows):
        for c in range(state.cols):
            shp = state.tensors[r][c][0].shape
            tn.set((r, c), ext_flat[(r, c, int(cfg_ext[r, c])) + tuple(slice(0, k) for k in shp)].copy())
    return tn


def test_k7_tj_network_route_consistency_oracle(fixtures_dir):
    st, relabel = k7_tj.build_state(fixtures_dir)
    cfg = relabel(k7_tj.CONFIG)
    flat = st.extended_flat()
    hor, ver = k1_routes.routes_by_pass(st.rows)
    tp = BMPSTruncateParams.SVD(k7_tj.DB_MIN, k7_tj.DB_MAX, 1e-15)
    amps = []
    for order, ops in ((fermion.ROW, hor), (fermion.COL, ver)):
        tn = decorated_tn(st, flat, st.ext_config(cfg, order))
        c = BMPSContractor(st.rows, st.cols)
        c.Init(tn)
        c.SetTruncateParams(tp)
        amps += [float(a) for a in k1_routes._walk(ops, c, tn, device=False)]
    assert len(amps) == k1_routes.N_AMPS and abs(amps[0]) > 0
    mag = np.abs(np.array(amps))
    assert np.max(np.abs(mag / mag[0] - 1)) < 1e-7, mag / mag[0] - 1
    # the two mode orders differ by the reordering sign kappa of the occupied modes (and nothing else)
    kappa = int(st.kappa(cfg[None])[0])
    assert np.sign(amps[len([o for o in hor if o[0] in ("trace", "tnn", "nnn", "sqrt5")])]) == kappa * np.sign(amps[0])
