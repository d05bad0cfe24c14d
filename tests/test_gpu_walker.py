"""BMPSWalker as an object behind the C ABI (pepsgpu_walker_*, peps_amd.capi.Walker): the reference's own walker tests on the device
(tests/test_2d_tn/test_bmps_contractor.cpp:1071-1121 Basic, :1128-1187 ContractRow, :1201-1284 BTenCache, :1294-1358 ShiftBTenWindow,
:1368-1431 TraceWithTwoSiteBTen; 12x12 Ising network, chi = 30, f64 at the reference's 1e-8), then what the walker is for: an
evolution through an MPO that is NOT a row of the network (explicit tensors, per-walker excited rows) against the oracle's
BMPSWalker on several configurations of a PEPS (f32 and f64), and the fermionic walker test (:878-985) on a decorated state."""
import numpy as np
import pytest

from oracle import ising
from oracle.bmps import BMPSTruncateParams, LEFT, DOWN, RIGHT, UP, HORIZONTAL
from oracle.contractor import BMPSContractor, TensorNetwork2D
from peps_amd import synthetic

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ising_ctx():
    from peps_amd import capi
    tn, lognorm, beta = ising.build_ising_tn(12, 12)
    sitps = [[[tn((r, c))] for c in range(12)] for r in range(12)]
    ctx = capi.Context(12, 12, 2, 1, 30, dtype=capi.F64, max_walkers=1)
    ctx.state_upload(synthetic.sitps_to_flat(sitps, 2, np.float64))
    yield ctx, tn
    ctx.close()


def _setup(ctx, row=2):
    ctx.set_configs(np.zeros((1, 12, 12), dtype=np.int32))
    ctx.grow_bmps_for_row(row)
    w = ctx.get_walker(UP)
    w.set_mpo(row)
    return w, 12 - 1 - row          # walker, index of the bottom environment in the DOWN stack


def _ref_trace(ctx, row=2):
    ctx.init_bten(LEFT, row)
    ctx.grow_full_bten(RIGHT, row, 2, True)
    return ctx.trace(row, 0, HORIZONTAL)[0]


def test_walker_basic(ising_ctx):
    ctx, tn = ising_ctx
    w, _ = _setup(ctx)
    n0 = ctx.bmps_stack_size(UP)
    assert n0 > 0 and w.GetStackSize() == n0 and w.GetPosition() == UP
    w.EvolveStep()
    assert w.GetStackSize() == n0 + 1 and ctx.bmps_stack_size(UP) == n0
    w.EvolveStep()
    assert w.GetStackSize() == n0 + 2 and ctx.bmps_stack_size(UP) == n0
    w2 = ctx.get_walker(UP)
    assert w2.GetStackSize() == n0
    w2.EvolveStep()
    assert w2.GetStackSize() == n0 + 1 and w.GetStackSize() == n0 + 2
    # the evolved fork is the BMPS the contractor's own growth produces (same MultiplyMPO)
    ctx.grow_bmps_step(UP)
    for i in (0, 5, 11):
        a, la = w2.GetBMPSTensor(i)
        b, lb = ctx.get_bmps_tensor(UP, ctx.bmps_stack_size(UP) - 1, i)
        assert a.shape == b.shape and np.allclose(a, b, atol=1e-12) and abs(la[0] - lb[0]) < 1e-12
    w.destroy(); w2.destroy()
    with pytest.raises(ValueError):
        ctx._ck(ctx._l.pepsgpu_walker_evolve_step(ctx._h, 12345))       # no such walker


def test_walker_contract_row(ising_ctx):
    ctx, tn = ising_ctx
    w, bot = _setup(ctx)
    assert w.GetStackSize() == 3 and ctx.bmps_stack_size(DOWN) >= 10
    val = w.ContractRow(bot)[0]
    ref = _ref_trace(ctx)
    assert ref != 0.0 and abs(val / ref - 1.0) < 1e-8


def test_walker_bten_cache(ising_ctx):
    ctx, tn = ising_ctx
    w, bot = _setup(ctx)
    ref = w.ContractRow(bot)[0]
    assert ref != 0.0
    mid = 6
    w.InitBTenLeft(bot, mid)
    assert w.GetBTenLeftCol() == mid
    w.InitBTenRight(bot, mid)
    assert w.GetBTenRightCol() == mid + 1
    w.ClearBTen()
    assert w.GetBTenLeftCol() == 0 and w.GetBTenRightCol() == 0
    w.InitBTenLeft(bot, 0)
    for col in range(mid):
        w.GrowBTenLeftStep(bot)
        assert w.GetBTenLeftCol() == col + 1
    w.InitBTenRight(bot, mid)
    assert abs(w.TraceWithBTen(bot, mid)[0] - ref) < 1e-8 * abs(ref)
    # the replacement named as a SITPS component and as an explicit tensor: the same tensor, the same value
    assert abs(w.TraceWithBTen(bot, mid, states=np.zeros(1, dtype=np.int32))[0] - ref) < 1e-8 * abs(ref)
    t = np.zeros((1, 2, 2, 2, 2))
    t[0] = tn((2, mid))
    assert abs(w.TraceWithBTen(bot, mid, tensors=t)[0] - ref) < 1e-8 * abs(ref)
    fresh = ctx.get_walker(UP)                      # a fresh fork: ContractRow as the fallback (:1276-1283)
    with pytest.raises(RuntimeError):
        fresh.ContractRow(bot)                      # (no TransferMPO named yet)
    fresh.set_mpo(2)
    assert abs(fresh.ContractRow(bot)[0] - ref) < 1e-8 * abs(ref)
    with pytest.raises(RuntimeError):
        w.TraceWithBTen(bot, mid + 2)               # the left cache does not reach that far
    w.ClearBTen()
    with pytest.raises(RuntimeError):
        w.GrowBTenRightStep(bot)                    # "Right BTen cache is empty. Call InitBTenRight first."


def test_walker_shift_bten_window(ising_ctx):
    ctx, tn = ising_ctx
    w, bot = _setup(ctx)
    ref = w.ContractRow(bot)[0]
    w.InitBTenLeft(bot, 1)
    w.InitBTenRight(bot, 1)
    assert (w.GetBTenLeftCol(), w.GetBTenRightCol()) == (1, 2)
    w.ShiftBTenWindow(bot, RIGHT)
    assert (w.GetBTenLeftCol(), w.GetBTenRightCol()) == (2, 3)
    assert abs(w.TraceWithBTen(bot, 2)[0] - ref) < 1e-8 * abs(ref)
    w.ShiftBTenWindow(bot, LEFT)
    assert (w.GetBTenLeftCol(), w.GetBTenRightCol()) == (1, 2)
    assert abs(w.TraceWithBTen(bot, 1)[0] - ref) < 1e-8 * abs(ref)


def test_walker_trace_with_two_site_bten(ising_ctx):
    ctx, tn = ising_ctx
    w, bot = _setup(ctx)
    ref = w.ContractRow(bot)[0]
    w.InitBTenLeft(bot, 1)
    w.InitBTenRight(bot, 2)
    assert (w.GetBTenLeftCol(), w.GetBTenRightCol()) == (1, 3)
    assert abs(w.TraceWithTwoSiteBTen(bot, 1)[0] - ref) < 1e-8 * abs(ref)
    w.ShiftBTenWindow(bot, RIGHT)
    assert (w.GetBTenLeftCol(), w.GetBTenRightCol()) == (2, 4)
    assert abs(w.TraceWithTwoSiteBTen(bot, 2, states=np.zeros((1, 2), dtype=np.int32))[0] - ref) < 1e-8 * abs(ref)
    t = np.zeros((1, 2, 2, 2, 2, 2))
    t[0, 0], t[0, 1] = tn((2, 2)), tn((2, 3))
    assert abs(w.TraceWithTwoSiteBTen(bot, 2, tensors=t)[0] - ref) < 1e-8 * abs(ref)


@pytest.mark.parametrize("dt,tol", [("f64", 1e-9), ("f32", 2e-5)])
def test_walker_foreign_mpo_and_excited_rows_vs_oracle(dt, tol):
    """What the object exists for (VERDICT r03 missing 1): Evolve through an MPO that is not a row of the context's network.
    6x6 D=3 chi=9 PEPS, several walkers: (i) an excited row named by per-walker states (structure-factor use), (ii) explicit
    per-walker tensors (the row's tensors scaled element-wise), (iii) one shared tensor set; each closed against the DOWN
    environment with the row below through ContractRow and TraceWithBTen, compared with the oracle's BMPSWalker."""
    from peps_amd import capi
    L, D, chi, row = 6, 3, 9, 2
    sitps = synthetic.make_sitps(L, D)
    cfgs = synthetic.make_configs(L, 4, "heisenberg", seed0=31)
    tp = BMPSTruncateParams.SVD(chi, chi, 0.0)
    rng = np.random.default_rng(8)
    exc_states = np.array([[1 - c[row, j] if j == (w % L) else c[row, j] for j in range(L)] for w, c in enumerate(cfgs)], dtype=np.int32)
    scale = [[1.0 + 0.3 * rng.standard_normal(synthetic.bond_dims(L, D, row, j)) for j in range(L)] for _ in cfgs]
    shared = [sitps[row][j][0] * (1.0 + 0.2 * rng.standard_normal(synthetic.bond_dims(L, D, row, j))) for j in range(L)]

    def pad(t):
        o = np.zeros((D, D, D, D))
        o[:t.shape[0], :t.shape[1], :t.shape[2], :t.shape[3]] = t
        return o

    ctx = capi.Context(L, L, D, 2, chi, dtype=capi.F32 if dt == "f32" else capi.F64, max_walkers=len(cfgs))
    ctx.state_upload(synthetic.sitps_to_flat(sitps, D, np.float64))
    ctx.set_configs(cfgs)
    ctx.grow_bmps_for_row(row + 1)              # UP has absorbed rows 0..row; DOWN the rows below row + 1
    bot = L - 1 - (row + 1)
    got = {}
    for name in ("states", "tensors", "shared"):
        w0 = ctx.get_walker(UP, level=row)       # fork of the UP level that has absorbed rows 0..row-1
        w = w0.clone()                           # (copy construction, as `auto excited_walker = main_walker;`)
        w0.destroy()
        assert w.GetStackSize() == row + 1
        if name == "states":
            w.set_mpo_states(row, exc_states)
        elif name == "tensors":
            w.set_mpo_tensors(row, np.array([[pad(sitps[row][j][cfgs[k, row, j]] * scale[k][j]) for j in range(L)] for k in range(len(cfgs))]))
        else:
            w.set_mpo_tensors(row, np.array([[pad(shared[j]) for j in range(L)]]))
        w.Evolve()
        assert w.GetStackSize() == row + 1       # Evolve(mpo) does not advance the counter (bmps_walker.h: note)
        w.set_mpo(row + 1)
        a = w.ContractRow(bot)
        w.InitBTenLeft(bot, 3)
        w.InitBTenRight(bot, 3)
        b = w.TraceWithBTen(bot, 3)
        # a replaced site at column 3 of the closing row: the other spin
        c = w.TraceWithBTen(bot, 3, states=(1 - cfgs[:, row + 1, 3]).astype(np.int32))
        got[name] = (a, b, c)
        w.destroy()
    assert np.all(ctx.walker_flags() == 0)
    ctx.close()
    for k, cfg in enumerate(cfgs):
        tn = TensorNetwork2D.from_sitps(sitps, cfg)
        c = BMPSContractor(L, L)
        c.Init(tn)
        c.SetTruncateParams(tp)
        c.GrowBMPSForRow(tn, row + 1)
        c.bmps_set[UP] = c.bmps_set[UP][:row + 1]
        bottom = c.bmps_set[DOWN][bot]
        for name in ("states", "tensors", "shared"):
            w = c.GetWalker(tn, UP)
            if name == "states":
                mpo = [sitps[row][j][exc_states[k, j]] for j in range(L)]
            elif name == "tensors":
                mpo = [sitps[row][j][cfg[row, j]] * scale[k][j] for j in range(L)]
            else:
                mpo = shared
            w.Evolve(mpo)
            nxt = tn.get_row(row + 1)
            ref_a = w.ContractRow(nxt, bottom)
            w.InitBTenLeft(nxt, bottom, 3)
            w.InitBTenRight(nxt, bottom, 3)
            ref_c = w.TraceWithBTen(sitps[row + 1][3][1 - cfg[row + 1, 3]], 3, bottom)
            a, b, cc = got[name]
            assert abs(a[k] / ref_a - 1) < tol, (name, k)
            assert abs(b[k] / ref_a - 1) < tol, (name, k)
            assert abs(cc[k] / ref_c - 1) < tol * 5, (name, k)


def test_walker_fermionic_bten(fixtures_dir):
    """BMPSWalkerFermionicBTenTest (test_bmps_contractor.cpp:878-985) on the reference's 20 x 24 projected t-J network (K7): the
    walker's TraceWithBTen agrees in magnitude with BMPSContractor::Trace (1e-8 relative) -- on the device a fermionic network is
    an ordinary network of sign-decorated components, so the same engine path runs."""
    import k7_tj
    from peps_amd import capi, fermion
    st, relabel = k7_tj.build_state(fixtures_dir)
    cfg = relabel(k7_tj.CONFIG)
    ctx = capi.Context(st.rows, st.cols, st.D, fermion.NVAR * st.d, k7_tj.DB_MAX, dtype=capi.F64, max_walkers=1,
                       chi_min=k7_tj.DB_MIN, trunc_err=1e-15)
    ctx.state_upload(st.extended_flat())
    ctx.set_configs(st.ext_config(cfg, fermion.ROW)[None])
    test_row, mid = st.rows // 2, st.cols // 2
    ctx.grow_bmps_for_row(test_row)
    w = ctx.get_walker(UP)
    assert w.GetStackSize() == test_row + 1
    w.set_mpo(test_row)
    bot = st.rows - 1 - test_row
    ctx.init_bten(LEFT, test_row)
    ctx.grow_full_bten(LEFT, test_row, st.cols - mid, False)
    ctx.grow_full_bten(RIGHT, test_row, mid + 1, True)
    own = np.array([[st.ext_config(cfg, fermion.ROW)[test_row, mid]]], dtype=np.int32)
    ref1 = ctx.replace_one_trace(test_row, mid, HORIZONTAL, own)[0, 0]       # BMPSContractor's own trace at (test_row, mid)
    assert ref1 != 0.0
    w.InitBTenLeft(bot, mid)
    assert w.GetBTenLeftCol() == mid
    w.InitBTenRight(bot, mid)
    assert w.GetBTenRightCol() == mid + 1
    v = w.TraceWithBTen(bot, mid)[0]
    assert abs(abs(v) / abs(ref1) - 1) < 1e-8
    # incremental left growth and a right cache grown step by step (:946-984)
    w.ClearBTen()
    w.InitBTenLeft(bot, 0)
    for col in range(mid):
        w.GrowBTenLeftStep(bot)
        assert w.GetBTenLeftCol() == col + 1
    w.InitBTenRight(bot, st.cols - 1)
    assert w.GetBTenRightCol() == st.cols
    for step in range(st.cols - 1 - mid):
        w.GrowBTenRightStep(bot)
        assert w.GetBTenRightCol() == st.cols - 1 - step
    assert w.GetBTenRightCol() == mid + 1
    assert abs(abs(w.TraceWithBTen(bot, mid)[0]) / abs(ref1) - 1) < 1e-8
    ctx.close()


def test_python_walker_releases_its_device_copy():
    """a Python Walker is a deep copy of one BMPS for all walkers of the context; it is released by destroy(), at the end
    of a `with` block and by the finalizer (the C++ BMPSWalker's destructor) -- a clone per source site must not pile up in the arena."""
    import gc
    from peps_amd import capi, synthetic
    L, D, chi = 4, 3, 9
    s = synthetic.make_sitps(L, D)
    ctx = capi.Context(L, L, D, 2, chi, dtype=capi.F64, max_walkers=2)
    ctx.state_upload(synthetic.sitps_to_flat(s, D))
    ctx.set_configs(synthetic.make_configs(L, 2, "heisenberg"))
    ctx.evaluate_amplitude()
    ctx.grow_bmps_for_row(2)
    with ctx.get_walker(capi.UP) as w:
        wid = w.wid
        c = w.clone()
        cid = c.wid
        del c
        gc.collect()
        with pytest.raises((RuntimeError, ValueError)):       # the clone is gone ("no such walker") ...
            capi.Walker(ctx, cid).GetStackSize()
        assert w.GetStackSize() >= 1            # ... the walker it was copied from is not
    with pytest.raises((RuntimeError, ValueError)):           # released at the end of the block
        capi.Walker(ctx, wid).GetStackSize()
    for _ in range(3):
        ctx.get_walker(capi.UP).clone()
    gc.collect()
    base = ctx.stats()["device_bytes"]
    for _ in range(40):                         # a clone per source site, dropped each time: the arena does not grow
        ctx.get_walker(capi.UP).clone()
    gc.collect()
    assert ctx.stats()["device_bytes"] == base
    ctx.close()
