"""BMPSWalker in the oracle: the reference's own walker tests on the 12x12 Ising network
(tests/test_2d_tn/test_bmps_contractor.cpp:1071-1121 Basic, :1128-1187 ContractRow, :1201-1284 BTenCache,
:1294-1358 ShiftBTenWindow, :1368-1431 TraceWithTwoSiteBTen), SVD(10, 30, 1e-15) as there."""
import numpy as np
import pytest

from oracle import ising
from oracle.bmps import BMPSTruncateParams, LEFT, DOWN, RIGHT, UP, HORIZONTAL
from oracle.contractor import BMPSContractor


@pytest.fixture(scope="module")
def net():
    tn, lognorm, beta = ising.build_ising_tn(12, 12)
    return tn


def _setup(tn, row=2):
    c = BMPSContractor(tn.rows, tn.cols)
    c.Init(tn)
    c.SetTruncateParams(BMPSTruncateParams.SVD(10, 30, 1e-15))
    c.GrowBMPSForRow(tn, row)
    w = c.GetWalker(tn, UP)
    mpo = tn.get_row(row)
    bottom = c.bmps_set[DOWN][tn.rows - 1 - row]
    return c, w, mpo, bottom


def test_walker_basic(net):
    """:1071-1121: fork, EvolveStep, independence of the contractor's stack and of a second walker"""
    tn = net
    c = BMPSContractor(tn.rows, tn.cols)
    c.Init(tn)
    c.SetTruncateParams(BMPSTruncateParams.SVD(10, 30, 1e-15))
    c.GrowBMPSForRow(tn, 2)
    w = c.GetWalker(tn, UP)
    n0 = len(c.bmps_set[UP])
    assert n0 > 0 and w.GetStackSize() == n0 and w.GetPosition() == UP
    w.EvolveStep()
    assert w.GetStackSize() == n0 + 1 and len(c.bmps_set[UP]) == n0
    w.EvolveStep()
    assert w.GetStackSize() == n0 + 2 and len(c.bmps_set[UP]) == n0
    w2 = c.GetWalker(tn, UP)
    assert w2.GetStackSize() == n0
    w2.EvolveStep()
    assert w2.GetStackSize() == n0 + 1 and w.GetStackSize() == n0 + 2
    # an evolved walker equals the contractor's own growth (same MultiplyMPO)
    c.GrowBMPSStep(tn, UP)
    for a, b in zip(w2.GetBMPS().tensors, c.bmps_set[UP][-1].tensors):
        assert a.shape == b.shape and np.allclose(a, b, atol=1e-13)


def test_walker_contract_row(net):
    """:1128-1187: <walker | row 2 | bottom_env> == Trace of the contractor"""
    tn = net
    c, w, mpo, bottom = _setup(tn)
    assert w.GetStackSize() == 3 and len(c.bmps_set[DOWN]) >= tn.rows - 2
    val = w.ContractRow(mpo, bottom)
    c.InitBTen(tn, LEFT, 2)
    c.GrowFullBTen(tn, RIGHT, 2, 2, True)
    ref = c.Trace(tn, (2, 0), HORIZONTAL)
    assert ref != 0.0 and abs(val / ref - 1.0) < 1e-8


def test_walker_bten_cache(net):
    """:1201-1284"""
    tn = net
    c, w, mpo, bottom = _setup(tn)
    ref = w.ContractRow(mpo, bottom)
    assert ref != 0.0
    lx, mid = tn.cols, tn.cols // 2
    w.InitBTenLeft(mpo, bottom, mid)
    assert w.GetBTenLeftCol() == mid
    w.InitBTenRight(mpo, bottom, mid)
    assert w.GetBTenRightCol() == mid + 1
    w.ClearBTen()
    assert w.GetBTenLeftCol() == 0 and w.GetBTenRightCol() == 0
    w.InitBTenLeft(mpo, bottom, 0)
    for col in range(mid):
        w.GrowBTenLeftStep(mpo, bottom)
        assert w.GetBTenLeftCol() == col + 1
    w.InitBTenRight(mpo, bottom, mid)
    assert abs(w.TraceWithBTen(mpo[mid], mid, bottom) - ref) < 1e-8
    assert abs(c.GetWalker(tn, UP).ContractRow(mpo, bottom) - ref) < 1e-8
    with pytest.raises(RuntimeError):
        w.TraceWithBTen(mpo[mid + 2], mid + 2, bottom)        # left cache does not reach that far


def test_walker_shift_bten_window(net):
    """:1294-1358"""
    tn = net
    c, w, mpo, bottom = _setup(tn)
    ref = w.ContractRow(mpo, bottom)
    w.InitBTenLeft(mpo, bottom, 1)
    w.InitBTenRight(mpo, bottom, 1)
    assert (w.GetBTenLeftCol(), w.GetBTenRightCol()) == (1, 2)
    w.ShiftBTenWindow(mpo, bottom, RIGHT)
    assert (w.GetBTenLeftCol(), w.GetBTenRightCol()) == (2, 3)
    assert abs(w.TraceWithBTen(mpo[2], 2, bottom) - ref) < 1e-8
    w.ShiftBTenWindow(mpo, bottom, LEFT)
    assert (w.GetBTenLeftCol(), w.GetBTenRightCol()) == (1, 2)
    assert abs(w.TraceWithBTen(mpo[1], 1, bottom) - ref) < 1e-8


def test_walker_trace_with_two_site_bten(net):
    """:1368-1431"""
    tn = net
    c, w, mpo, bottom = _setup(tn)
    ref = w.ContractRow(mpo, bottom)
    assert ref != 0.0
    w.InitBTenLeft(mpo, bottom, 1)
    w.InitBTenRight(mpo, bottom, 2)
    assert (w.GetBTenLeftCol(), w.GetBTenRightCol()) == (1, 3)
    assert abs(w.TraceWithTwoSiteBTen(mpo[1], mpo[2], 1, mpo, bottom) - ref) < 1e-8
    w.ShiftBTenWindow(mpo, bottom, RIGHT)
    assert (w.GetBTenLeftCol(), w.GetBTenRightCol()) == (2, 4)
    assert abs(w.TraceWithTwoSiteBTen(mpo[2], mpo[3], 2, mpo, bottom) - ref) < 1e-8


def test_walker_evolve_with_a_foreign_mpo(net):
    """Evolve(mpo) with an MPO that is NOT a row of the network (bmps_walker.h:13-21; the structure-factor mixin evolves through
    an excited row): the closed value equals a direct contraction of the network with that row substituted."""
    tn = net
    rng = np.random.default_rng(4)
    c, w, mpo, bottom = _setup(tn)
    exc = [t * (1.0 + 0.3 * rng.standard_normal(t.shape)) for t in mpo]
    w.Evolve(exc)                                            # row 2 replaced by the foreign MPO
    c.GrowBMPSForRow(tn, 3)
    bottom3 = c.bmps_set[DOWN][tn.rows - 1 - 3]
    val = w.ContractRow(tn.get_row(3), bottom3)
    import copy
    tn2 = copy.copy(tn)
    tn2.t = [list(r) for r in tn.t]
    tn2.t[2] = exc
    c2 = BMPSContractor(tn.rows, tn.cols)
    c2.Init(tn2)
    c2.SetTruncateParams(BMPSTruncateParams.SVD(10, 30, 1e-15))
    c2.GrowBMPSForRow(tn2, 3)
    c2.InitBTen(tn2, LEFT, 3)
    c2.GrowFullBTen(tn2, RIGHT, 3, 2, True)
    ref = c2.Trace(tn2, (3, 0), HORIZONTAL)
    assert abs(val / ref - 1.0) < 1e-8
