"""GPU parity tests proper: the HIP path (through the C ABI) against the float64 oracle on the
same seeded inputs, against the reference's fixtures, and through size-independent properties."""
import os

import numpy as np
import pytest

from oracle import ising, qlten_io, vmc
from oracle.bmps import BMPSTruncateParams, LEFT, DOWN, RIGHT, UP, HORIZONTAL, VERTICAL
from oracle.contractor import BMPSContractor, TensorNetwork2D
from peps_amd import synthetic

pytestmark = pytest.mark.gpu

TOL = {"f32": 1e-5, "f64": 1e-9}     # relative amplitude tolerance device vs float64 oracle (1e-5)


def _ctx(L, D, d, chi, dt, n):
    from peps_amd import capi
    return capi.Context(L, L, D, d, chi, dtype=capi.F32 if dt == "f32" else capi.F64, max_walkers=max(n, 1))


def _upload(ctx, sitps, D):
    ctx.state_upload(synthetic.sitps_to_flat(sitps, D, np.float64))


def _oracle_amps(sitps, configs, chi):
    tp = BMPSTruncateParams.SVD(chi, chi, 0.0)
    return np.array([vmc.TPSWaveFunctionComponent(sitps, c, tp).amplitude for c in configs])


@pytest.mark.parametrize("dt", ["f32", "f64"])
@pytest.mark.parametrize("L,D,chi", [(4, 2, 4), (5, 3, 6), (6, 4, 8), (8, 4, 16)])
def test_evaluate_amplitude_synthetic(L, D, chi, dt):
    sitps = synthetic.make_sitps(L, D)
    cfgs = synthetic.make_configs(L, 6, "heisenberg")
    ref = _oracle_amps(sitps, cfgs, chi)
    ctx = _ctx(L, D, 2, chi, dt, len(cfgs))
    _upload(ctx, sitps, D)
    ctx.set_configs(cfgs)
    got = ctx.evaluate_amplitude()
    assert np.all(ctx.walker_flags() == 0)
    assert np.max(np.abs(got / ref - 1)) < TOL[dt], (got, ref)


@pytest.mark.parametrize("dt", ["f32", "f64"])
def test_k5_fixture_4x4_d8(fixtures_dir, dt):
    """Reference fixture tests/slow_tests/test_data/tps_square_heisenberg4x4D8Double."""
    s = qlten_io.load_sitps(os.path.join(fixtures_dir, "tps_square_heisenberg4x4D8Double"))
    cfgs = np.stack([synthetic.checkerboard(4)] + list(synthetic.make_configs(4, 5, "heisenberg")))
    for chi in (16, 64):
        ref = _oracle_amps(s, cfgs, chi)
        ctx = _ctx(4, 8, 2, chi, dt, len(cfgs))
        _upload(ctx, s, 8)
        ctx.set_configs(cfgs)
        got = ctx.evaluate_amplitude()
        assert np.max(np.abs(got / ref - 1)) < TOL[dt]
    # chi = 64 is exact: brute-force value of SURVEY 8c
    assert abs(got[0] / 1.441641034201432e+02 - 1) < TOL[dt]


@pytest.mark.parametrize("dt", ["f32", "f64"])
def test_all_trace_routes_and_positions(dt):
    """Every stack direction (UP/DOWN/LEFT/RIGHT), BTen direction and trace route gives the same
    amplitude (tests/test_2d_tn/test_bmps_contractor.cpp:273-405 does this for the reference)."""
    L, D, chi = 6, 3, 9
    sitps = synthetic.make_sitps(L, D)
    cfgs = synthetic.make_configs(L, 4, "heisenberg")
    tp = BMPSTruncateParams.SVD(chi, chi, 0.0)
    ctx = _ctx(L, D, 2, chi, dt, len(cfgs))
    _upload(ctx, sitps, D)
    ctx.set_configs(cfgs)
    dev = []
    ctx.grow_bmps_for_row(2)
    ctx.init_bten(LEFT, 2)
    ctx.grow_full_bten(RIGHT, 2, 2, True)
    dev.append(ctx.trace(2, 0, HORIZONTAL))
    ctx.shift_bten_window(RIGHT)
    dev.append(ctx.trace(2, 1, HORIZONTAL))
    ctx.grow_bmps_for_col(1)
    ctx.init_bten(DOWN, 1)
    ctx.grow_full_bten(UP, 1, 2, True)
    dev.append(ctx.trace(L - 2, 1, VERTICAL))
    ctx.shift_bten_window(UP)
    dev.append(ctx.trace(L - 3, 1, VERTICAL))
    ref = []
    for w, cfg in enumerate(cfgs):
        tn = TensorNetwork2D.from_sitps(sitps, cfg)
        c = BMPSContractor(L, L)
        c.Init(tn)
        c.SetTruncateParams(tp)
        r = []
        c.GrowBMPSForRow(tn, 2)
        c.InitBTen(tn, LEFT, 2)
        c.GrowFullBTen(tn, RIGHT, 2, 2, True)
        r.append(c.Trace(tn, (2, 0), HORIZONTAL))
        c.ShiftBTenWindow(tn, RIGHT)
        r.append(c.Trace(tn, (2, 1), HORIZONTAL))
        c.GrowBMPSForCol(tn, 1)
        c.InitBTen(tn, DOWN, 1)
        c.GrowFullBTen(tn, UP, 1, 2, True)
        r.append(c.Trace(tn, (L - 2, 1), VERTICAL))
        c.ShiftBTenWindow(tn, UP)
        r.append(c.Trace(tn, (L - 3, 1), VERTICAL))
        ref.append(r)
    ref = np.array(ref).T
    dev = np.array(dev)
    assert np.max(np.abs(dev / ref - 1)) < TOL[dt]


@pytest.mark.parametrize("dt", ["f32", "f64"])
def test_replace_traces_and_punch_hole(dt):
    """ReplaceNNSiteTrace / ReplaceOneSiteTrace ratios and PunchHole . site == Trace (K3,
    test_bmps_contractor.cpp:407-470), both orientations."""
    L, D, chi = 5, 3, 9
    sitps = synthetic.make_sitps(L, D)
    cfgs = synthetic.make_configs(L, 3, "heisenberg")
    tp = BMPSTruncateParams.SVD(chi, chi, 0.0)
    ctx = _ctx(L, D, 2, chi, dt, len(cfgs))
    _upload(ctx, sitps, D)
    ctx.set_configs(cfgs)
    row, col = 2, 1
    ctx.grow_bmps_for_row(row)
    ctx.grow_full_bten(LEFT, row, L - col, True)        # LEFT btens 0..col
    ctx.grow_full_bten(RIGHT, row, col + 2, True)       # RIGHT btens up to slice col+1
    cand = np.array([[[0, 0], [0, 1], [1, 0], [1, 1]]] * len(cfgs), dtype=np.int32)
    nn = ctx.replace_nn_trace(row, col, HORIZONTAL, cand)
    psi = ctx.trace(row, col, HORIZONTAL)
    # one more RIGHT step so that the one-site environment of (row, col) is available
    ctx.grow_bten_step(RIGHT)
    one = ctx.replace_one_trace(row, col, HORIZONTAL, np.array([[0, 1]] * len(cfgs), dtype=np.int32))
    hole = ctx.punch_hole(row, col, HORIZONTAL)
    for w, cfg in enumerate(cfgs):
        tn = TensorNetwork2D.from_sitps(sitps, cfg)
        c = BMPSContractor(L, L)
        c.Init(tn)
        c.SetTruncateParams(tp)
        c.GrowBMPSForRow(tn, row)
        c.GrowFullBTen(tn, LEFT, row, L - col, True)
        c.GrowFullBTen(tn, RIGHT, row, col + 2, True)
        for k, (s1, s2) in enumerate([(0, 0), (0, 1), (1, 0), (1, 1)]):
            ref = c.ReplaceNNSiteTrace(tn, (row, col), (row, col + 1), HORIZONTAL, sitps[row][col][s1], sitps[row][col + 1][s2])
            assert abs(nn[w, k] - ref) < TOL[dt] * abs(c.Trace(tn, (row, col), HORIZONTAL)) * 10
        ref_psi = c.Trace(tn, (row, col), HORIZONTAL)
        assert abs(psi[w] / ref_psi - 1) < TOL[dt]
        c.GrowBTenStep(tn, RIGHT)
        for k in range(2):
            ref = c.ReplaceOneSiteTrace(tn, (row, col), sitps[row][col][k], HORIZONTAL)
            assert abs(one[w, k] - ref) < TOL[dt] * abs(ref_psi) * 10
        h_ref = c.PunchHole(tn, (row, col), HORIZONTAL)
        dd = h_ref.shape
        h_dev = hole[w][:dd[0], :dd[1], :dd[2], :dd[3]]
        assert np.max(np.abs(h_dev - h_ref)) < TOL[dt] * 10 * np.max(np.abs(h_ref))
        assert abs(np.sum(h_dev * tn((row, col))) / ref_psi - 1) < TOL[dt] * 10


def test_k1_ising_device_f64():
    """K1 on the device: the 12x12 critical Ising network (uniform 'configuration') with fixed
    chi = 30 reproduces the exact free energy to 1e-8 (reference tolerance)."""
    tn, lognorm, beta = ising.build_ising_tn(12, 12)
    f_ex = ising.exact_free_energy(12, 12, 1.0 / beta)
    sitps = [[[tn((r, c))] for c in range(12)] for r in range(12)]
    ctx = _ctx(12, 2, 1, 30, "f64", 1)
    _upload(ctx, sitps, 2)
    ctx.set_configs(np.zeros((1, 12, 12), dtype=np.int32))
    amps = []
    ctx.grow_bmps_for_row(2)
    ctx.init_bten(LEFT, 2)
    ctx.grow_full_bten(RIGHT, 2, 2, True)
    amps.append(ctx.trace(2, 0, HORIZONTAL)[0])
    ctx.shift_bten_window(RIGHT)
    amps.append(ctx.trace(2, 1, HORIZONTAL)[0])
    ctx.grow_bmps_for_col(1)
    ctx.init_bten(DOWN, 1)
    ctx.grow_full_bten(UP, 1, 2, True)
    amps.append(ctx.trace(10, 1, VERTICAL)[0])
    for a in amps:
        assert abs(-(np.log(a) + lognorm) / 144 / beta - f_ex) < 1e-8


def test_k1_ising_device_all_21_routes_f64():
    """All 21 routes of Contract2DTNUsingBMPSContractor (test_bmps_contractor.cpp:273-405) on the
    device, BTen2 / NNN / TNN / sqrt(5) included, tolerance 1e-8 as the reference."""
    import k1_routes
    tn, lognorm, beta = ising.build_ising_tn(12, 12)
    f_ex = ising.exact_free_energy(12, 12, 1.0 / beta)
    sitps = [[[tn((r, c))] for c in range(12)] for r in range(12)]
    ctx = _ctx(12, 2, 1, 30, "f64", 1)
    _upload(ctx, sitps, 2)
    ctx.set_configs(np.zeros((1, 12, 12), dtype=np.int32))
    amps = k1_routes.run_device(ctx)
    assert len(amps) == k1_routes.N_AMPS
    for a in amps:
        assert abs(-(np.log(a[0]) + lognorm) / 144 / beta - f_ex) < 1e-8


@pytest.mark.parametrize("dt", ["f32", "f64"])
@pytest.mark.parametrize("orient", [HORIZONTAL, VERTICAL])
def test_nnn_tnn_sqrt5_replacement_traces(dt, orient):
    """ReplaceNNNSiteTrace / ReplaceTNNSiteTrace / ReplaceSqrt5DistTwoSiteTrace with candidate
    states (trace.h:207-536) against the oracle, several walkers, both diagonal directions."""
    import k1_routes
    from oracle.contractor import LEFTUP_TO_RIGHTDOWN, LEFTDOWN_TO_RIGHTUP
    L, D, chi = 6, 3, 9
    sitps = synthetic.make_sitps(L, D)
    cfgs = synthetic.make_configs(L, 3, "heisenberg")
    tp = BMPSTruncateParams.SVD(chi, chi, 0.0)
    ctx = _ctx(L, D, 2, chi, dt, len(cfgs))
    _upload(ctx, sitps, D)
    ctx.set_configs(cfgs)
    r0, c0 = (2, 1) if orient == HORIZONTAL else (1, 2)
    pairs = [(0, 0), (0, 1), (1, 0), (1, 1)]
    cand2 = np.array([pairs] * len(cfgs), dtype=np.int32)
    trip = [(0, 1, 0), (1, 0, 1), (1, 1, 0), (0, 0, 1), (1, 0, 0)]
    cand3 = np.array([trip] * len(cfgs), dtype=np.int32)

    def prepare(api, dev):
        if orient == HORIZONTAL:
            if dev:
                api.grow_bmps_for_row(r0); api.grow_full_bten(LEFT, r0, L - c0, True); api.grow_full_bten(RIGHT, r0, c0 + 2, True)
            else:
                api[0].GrowBMPSForRow(api[1], r0); api[0].GrowFullBTen(api[1], LEFT, r0, L - c0, True)
                api[0].GrowFullBTen(api[1], RIGHT, r0, c0 + 2, True)
        else:
            if dev:
                api.grow_bmps_for_col(c0); api.grow_full_bten(UP, c0, L - r0, True); api.grow_full_bten(DOWN, c0, r0 + 2, True)
            else:
                api[0].GrowBMPSForCol(api[1], c0); api[0].GrowFullBTen(api[1], UP, c0, L - r0, True)
                api[0].GrowFullBTen(api[1], DOWN, c0, r0 + 2, True)

    def prepare2(api, dev):
        # two-row environments around the 2x3 / 3x2 block with upper-left corner (r0, c0)
        if orient == HORIZONTAL:
            if dev:
                api.grow_bmps_for_row(r0); api.grow_full_bten2(LEFT, r0, L - c0, True); api.grow_full_bten2(RIGHT, r0, c0 + 2, True)
            else:
                api[0].GrowBMPSForRow(api[1], r0); api[0].GrowFullBTen2(api[1], LEFT, r0, L - c0, True)
                api[0].GrowFullBTen2(api[1], RIGHT, r0, c0 + 2, True)
        else:
            if dev:
                api.grow_bmps_for_col(c0); api.grow_full_bten2(UP, c0, L - r0, True); api.grow_full_bten2(DOWN, c0, r0 + 2, True)
            else:
                api[0].GrowBMPSForCol(api[1], c0); api[0].GrowFullBTen2(api[1], UP, c0, L - r0, True)
                api[0].GrowFullBTen2(api[1], DOWN, c0, r0 + 2, True)

    prepare(ctx, True)
    psi = ctx.trace(r0, c0, orient)
    tnn = ctx.replace_tnn_trace(r0, c0, orient, cand3)
    tnn0 = ctx.replace_tnn_trace(r0, c0, orient)
    prepare2(ctx, True)
    nnn = {d: ctx.replace_nnn_trace(r0, c0, d, orient, cand2) for d in (LEFTUP_TO_RIGHTDOWN, LEFTDOWN_TO_RIGHTUP)}
    nnn0 = ctx.replace_nnn_trace(r0, c0, LEFTUP_TO_RIGHTDOWN, orient)
    # the sqrt5 block needs the far BTen2 one site further out
    s5 = {}
    if orient == HORIZONTAL:
        ctx.grow_full_bten2(RIGHT, r0, c0 + 3, True)
    else:
        ctx.grow_full_bten2(DOWN, c0, r0 + 3, True)
    for d in (LEFTUP_TO_RIGHTDOWN, LEFTDOWN_TO_RIGHTUP):
        s5[d] = ctx.replace_sqrt5_trace(r0, c0, d, orient, cand2)
    assert np.all(ctx.walker_flags() == 0)
    tol = TOL[dt] * 20
    for w, cfg in enumerate(cfgs):
        tn = TensorNetwork2D.from_sitps(sitps, cfg)
        c = BMPSContractor(L, L)
        c.Init(tn)
        c.SetTruncateParams(tp)
        prepare((c, tn), False)
        ref_psi = c.Trace(tn, (r0, c0), orient)
        assert abs(psi[w] / ref_psi - 1) < TOL[dt]
        assert abs(tnn0[w] / ref_psi - 1) < tol
        sites = k1_routes.tnn_sites(r0, c0, orient)
        for k, st in enumerate(trip):
            ref = c.ReplaceTNNSiteTrace(tn, (r0, c0), orient, *[sitps[s[0]][s[1]][x] for s, x in zip(sites, st)])
            assert abs(tnn[w, k] - ref) < tol * abs(ref_psi), (k, tnn[w, k], ref)
        prepare2((c, tn), False)
        assert abs(nnn0[w] / ref_psi - 1) < tol
        for d in (LEFTUP_TO_RIGHTDOWN, LEFTDOWN_TO_RIGHTUP):
            sl, sr = k1_routes.nnn_sites(r0, c0, d)
            for k, (a, b) in enumerate(pairs):
                ref = c.ReplaceNNNSiteTrace(tn, (r0, c0), d, orient, sitps[sl[0]][sl[1]][a], sitps[sr[0]][sr[1]][b])
                assert abs(nnn[d][w, k] - ref) < tol * abs(ref_psi), (d, k, nnn[d][w, k], ref)
        if orient == HORIZONTAL:
            c.GrowFullBTen2(tn, RIGHT, r0, c0 + 3, True)
        else:
            c.GrowFullBTen2(tn, DOWN, c0, r0 + 3, True)
        for d in (LEFTUP_TO_RIGHTDOWN, LEFTDOWN_TO_RIGHTUP):
            sl, sr = k1_routes.sqrt5_sites(r0, c0, d, orient)
            for k, (a, b) in enumerate(pairs):
                ref = c.ReplaceSqrt5DistTwoSiteTrace(tn, (r0, c0), d, orient, sitps[sl[0]][sl[1]][a], sitps[sr[0]][sr[1]][b])
                assert abs(s5[d][w, k] - ref) < tol * abs(ref_psi), (d, k, s5[d][w, k], ref)


def test_error_codes():
    from peps_amd import capi
    ctx = capi.Context(4, 4, 2, 2, 4, dtype=capi.F32, max_walkers=2)
    with pytest.raises(RuntimeError):          # std::logic_error analogue: nothing uploaded
        ctx.n = 1
        ctx.evaluate_amplitude()
    sitps = synthetic.make_sitps(4, 2)
    _upload(ctx, sitps, 2)
    with pytest.raises(IndexError):            # std::out_of_range: config value >= physical dim
        ctx.set_configs(np.full((1, 4, 4), 2, dtype=np.int32))
    with pytest.raises(ValueError):            # too many walkers
        ctx.set_configs(np.zeros((3, 4, 4), dtype=np.int32))
    ctx.set_configs(np.zeros((2, 4, 4), dtype=np.int32))
    with pytest.raises(RuntimeError):          # trace without environments
        ctx.trace(1, 1, HORIZONTAL)


@pytest.mark.parametrize("dt", ["f32", "f64"])
@pytest.mark.parametrize("Ly,Lx", [(3, 6), (6, 3), (2, 5)])
def test_rectangular_lattices(Ly, Lx, dt):
    """rows != cols: amplitude, horizontal / vertical trace routes and hole . site == psi against the oracle"""
    from peps_amd import capi
    D, chi = 3, 9
    rng = np.random.default_rng(10 * Ly + Lx)
    sitps = []
    for r in range(Ly):
        row = []
        for c in range(Lx):
            shp = (1 if c == 0 else D, 1 if r == Ly - 1 else D, 1 if c == Lx - 1 else D, 1 if r == 0 else D)
            row.append([np.einsum("i,j,k,l->ijkl", *[rng.uniform(0.5, 1.5, size=n) for n in shp]) * 0.5
                        + 0.1 * rng.standard_normal(shp) for _ in range(2)])
        sitps.append(row)
    flat = np.zeros((Ly, Lx, 2, D, D, D, D))
    for r in range(Ly):
        for c in range(Lx):
            for s in range(2):
                t = sitps[r][c][s]
                flat[(r, c, s) + tuple(slice(0, k) for k in t.shape)] = t
    cfgs = rng.integers(0, 2, size=(4, Ly, Lx))
    tp = BMPSTruncateParams.SVD(chi, chi, 0.0)
    ctx = capi.Context(Ly, Lx, D, 2, chi, dtype=capi.F32 if dt == "f32" else capi.F64, max_walkers=4)
    ctx.state_upload(flat)
    ctx.set_configs(cfgs)
    amp = ctx.evaluate_amplitude()
    ctx.set_configs(cfgs)      # GrowBTenStep works on the tops of the BMPS stacks (grow.h:529-582): start from empty stacks
    r0, c0 = Ly // 2, Lx // 2 - 1
    ctx.grow_bmps_for_row(r0); ctx.grow_full_bten(LEFT, r0, Lx - c0, True); ctx.grow_full_bten(RIGHT, r0, c0 + 2, True)
    psi_h = ctx.trace(r0, c0, HORIZONTAL)
    ctx.grow_bten_step(RIGHT)
    hole = ctx.punch_hole(r0, c0, HORIZONTAL)
    c1 = Lx // 2
    ctx.set_configs(cfgs)
    ctx.grow_bmps_for_col(c1); ctx.grow_full_bten(UP, c1, Ly, True); ctx.grow_full_bten(DOWN, c1, 2, True)
    psi_v = ctx.trace(0, c1, VERTICAL)
    for w, cfg in enumerate(cfgs):
        ref = vmc.TPSWaveFunctionComponent(sitps, cfg, tp).amplitude
        assert abs(amp[w] / ref - 1) < TOL[dt]
        assert abs(psi_h[w] / ref - 1) < TOL[dt] * 10 and abs(psi_v[w] / ref - 1) < TOL[dt] * 10
        t = sitps[r0][c0][cfg[r0, c0]]
        h = hole[w][tuple(slice(0, k) for k in t.shape)]
        assert abs(np.sum(h * t) / ref - 1) < TOL[dt] * 10


@pytest.mark.parametrize("dt,terr,tol", [("f64", 1e-6, 1e-9), ("f64", 1e-3, 1e-9), ("f32", 1e-6, 2e-3), ("f32", 1e-3, 2e-3)])
def test_truncation_by_error_dmin_dmax(dt, terr, tol):
    """BMPSTruncateParams::SVD(D_min, D_max, trunc_err > 0) (bmps.h:47-98, qlten::SVD truncation at
    bmps_impl.h:235-238): the kept bond dimension is chosen per walker and per bond from the discarded
    weight; the device keeps the tensors zero padded to D_max.  Amplitudes and the kept dimensions of
    every bond of the DOWN stack agree with the oracle run with the same parameters."""
    from peps_amd import capi
    L, D, dmin, dmax = 6, 4, 2, 16
    sitps = synthetic.make_sitps(L, D, noise=0.3)
    cfgs = synthetic.make_configs(L, 5, "heisenberg")
    tp = BMPSTruncateParams.SVD(dmin, dmax, terr)
    comps = [vmc.TPSWaveFunctionComponent(sitps, c, tp) for c in cfgs]
    ref = np.array([c.amplitude for c in comps])
    fixed = _oracle_amps(sitps, cfgs, dmax)
    assert np.max(np.abs(ref / fixed - 1)) > 10 * tol or dt == "f32"   # the error rule does truncate more than D_max alone
    ctx = capi.Context(L, L, D, 2, dmax, dtype=capi.F32 if dt == "f32" else capi.F64, max_walkers=len(cfgs),
                       chi_min=dmin, trunc_err=terr)
    _upload(ctx, sitps, D)
    ctx.set_configs(cfgs)
    got = ctx.evaluate_amplitude()
    assert np.all(ctx.walker_flags() == 0)
    assert np.max(np.abs(got / ref - 1)) < tol, (got, ref)
    # kept dimensions: f64 always; f32 (where the Jacobi kernel of the walkers with few rows applies the rule itself, from its
    # registers) with the coarse error bound, whose cut is far from the rounding of the singular values
    if dt == "f64" or terr >= 1e-3:
        n_down = ctx.bmps_stack_size(DOWN)
        assert n_down == L
        for level in range(1, n_down):
            for idx in range(L):
                data, _ = ctx.get_bmps_tensor(DOWN, level, idx)
                for w, comp in enumerate(comps):
                    t_ref = comp.contractor.bmps_set[DOWN][level].tensors[idx]
                    kept_l = int(np.sum(np.any(data[w] != 0, axis=(1, 2))))
                    kept_r = int(np.sum(np.any(data[w] != 0, axis=(0, 1))))
                    assert (kept_l, kept_r) == (t_ref.shape[0], t_ref.shape[2]), (level, idx, w)
        assert any(t.shape[0] < min(dmax, D ** 3) for c in comps for t in c.contractor.bmps_set[DOWN][3].tensors[1:])


@pytest.mark.parametrize("dt", ["f64", "f32"])
def test_plaquette_trace_second_bten2_set_and_slice_override(dt):
    """The three calls behind the environment-reusing fermionic diagonal hop (round 5: pepsgpu_bten2_select_set,
    pepsgpu_cfg_override_slice, pepsgpu_replace_plaquette_trace), on a BOSONIC state where every piece has an independent answer:
    (a) the plaquette trace with the walkers' own states = psi; (b) with two replaced states on a diagonal = ReplaceNNNSiteTrace;
    (c) environments of the second set grown while rows r and r + 1 read other states, closed with four replaced tensors = the
    amplitude of the configuration with those two rows replaced (oracle); (d) the first set is untouched by all of it."""
    from peps_amd import capi
    L, D, chi = 5, 3, 27          # chi large enough that the two-row environments are exact
    sitps = synthetic.make_sitps(L, D)
    cfgs = synthetic.make_configs(L, 4, "heisenberg")
    rng = np.random.default_rng(3)
    ctx = _ctx(L, D, 2, chi, dt, len(cfgs))
    _upload(ctx, sitps, D)
    ctx.set_configs(cfgs)
    psi = ctx.evaluate_amplitude()
    tol = 50 * TOL[dt]
    row, col = 1, 2
    ctx.generate_bmps_approach(UP)
    ctx.shift_bmps_window(DOWN)                                   # UP has absorbed row 0, DOWN rows 3, 4: the pair (1, 2) is open
    ctx.grow_full_bten2(RIGHT, row, 2, True)
    ctx.init_bten2(LEFT, row)
    for _ in range(col):
        ctx.grow_bten2_step(LEFT, row)
    own = np.stack([cfgs[:, row, col], cfgs[:, row + 1, col], cfgs[:, row + 1, col + 1], cfgs[:, row, col + 1]], axis=-1)
    a = ctx.replace_plaquette_trace(row, col, None, 0, 0)
    assert np.max(np.abs(a / psi - 1)) < tol                                                   # (a)
    assert np.max(np.abs(ctx.replace_plaquette_trace(row, col, own[:, None, :], 0, 0)[:, 0] / psi - 1)) < tol
    swapped = own.copy()
    swapped[:, 0], swapped[:, 2] = own[:, 2], own[:, 0]                                        # the two ends of the dr diagonal exchanged
    b = ctx.replace_plaquette_trace(row, col, swapped[:, None, :], 0, 0)[:, 0]
    nnn = ctx.replace_nnn_trace(row, col, capi.LEFTUP_TO_RIGHTDOWN, HORIZONTAL, np.stack([own[:, 2], own[:, 0]], axis=-1)[:, None, :])[:, 0]
    assert np.max(np.abs(b - nnn)) < tol * np.max(np.abs(nnn))                                 # (b)
    # (c): rows 1 and 2 under other states
    new = cfgs.copy()
    new[:, row, :] = rng.integers(0, 2, size=(len(cfgs), L))
    new[:, row + 1, :] = rng.integers(0, 2, size=(len(cfgs), L))
    ctx.bten2_select_set(1)
    ctx.cfg_override_slice(HORIZONTAL, row, new[:, row, :])
    ctx.grow_full_bten2(RIGHT, row, 2, True)            # the RIGHT chain reads row 1 from the override, row 2 from the walkers' table ...
    ctx.cfg_override_slice(HORIZONTAL, row + 1, new[:, row + 1, :])
    ctx.init_bten2(LEFT, row)
    for _ in range(col):
        ctx.grow_bten2_step(LEFT, row)                  # ... and the LEFT chain row 2 from the override, row 1 from the walkers' table
    ctx.cfg_override_slice(HORIZONTAL, 0, None)
    # so the network that is closed is: row 1 = new right of the plaquette / old left of it, row 2 = new left / old right, plaquette = cand
    mixed = cfgs.copy()
    mixed[:, row, col + 2:] = new[:, row, col + 2:]
    mixed[:, row + 1, :col] = new[:, row + 1, :col]
    cand = np.stack([new[:, row, col], new[:, row + 1, col], new[:, row + 1, col + 1], new[:, row, col + 1]], axis=-1)
    mixed[:, row, col], mixed[:, row + 1, col], mixed[:, row + 1, col + 1], mixed[:, row, col + 1] = cand.T
    c = ctx.replace_plaquette_trace(row, col, cand[:, None, :], 1, 1)[:, 0]
    want = _oracle_amps(sitps, mixed, chi)
    assert np.max(np.abs(c / want - 1)) < tol, (c, want)                                       # (c)
    ctx.bten2_select_set(0)
    assert np.max(np.abs(ctx.replace_plaquette_trace(row, col, None, 0, 0) / psi - 1)) < tol   # (d)
    with pytest.raises((RuntimeError, ValueError)):
        ctx.bten2_select_set(2)
    ctx.close()
