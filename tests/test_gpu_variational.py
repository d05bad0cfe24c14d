"""Variational BMPS compression on the device (CompressMPSScheme::VARIATION2Site / VARIATION1Site,
bmps_impl.h:864-1212) against the oracle restatement and the K1 golden value."""
import numpy as np
import pytest

from oracle import ising, vmc
from oracle.bmps import BMPSTruncateParams
from peps_amd import synthetic

pytestmark = pytest.mark.gpu

SCHEMES = {"Variational2Site": 1, "Variational1Site": 2}


def _ctx(L, D, d, dmin, dmax, terr, scheme, tol, iters, dt, n):
    from peps_amd import capi
    return capi.Context(L, L, D, d, dmax, dtype=capi.F32 if dt == "f32" else capi.F64, max_walkers=n, chi_min=dmin,
                        trunc_err=terr, scheme=SCHEMES[scheme], convergence_tol=tol, iter_max=iters)


@pytest.mark.parametrize("scheme", list(SCHEMES))
def test_k1_ising_all_21_routes_variational_device(scheme):
    """test_bmps_contractor.cpp:472-486 on the device: Variational2Site / 1Site (10, 30, 1e-15, 1e-14, 10),
    all 21 routes, tolerance 1e-8 as the reference."""
    import k1_routes
    tn, lognorm, beta = ising.build_ising_tn(12, 12)
    f_ex = ising.exact_free_energy(12, 12, 1.0 / beta)
    sitps = [[[tn((r, c))] for c in range(12)] for r in range(12)]
    ctx = _ctx(12, 2, 1, 10, 30, 1e-15, scheme, 1e-14, 10, "f64", 1)
    ctx.state_upload(synthetic.sitps_to_flat(sitps, 2, np.float64))
    ctx.set_configs(np.zeros((1, 12, 12), dtype=np.int32))
    amps = k1_routes.run_device(ctx)
    assert len(amps) == k1_routes.N_AMPS
    for a in amps:
        assert abs(-(np.log(a[0]) + lognorm) / 144 / beta - f_ex) < 1e-8


@pytest.mark.parametrize("dt,tol", [("f64", 1e-7), ("f32", 2e-4)])
@pytest.mark.parametrize("scheme", list(SCHEMES))
@pytest.mark.parametrize("L,D,chi", [(5, 3, 4), (6, 4, 5)])
def test_variational_amplitudes_against_oracle(L, D, chi, scheme, dt, tol):
    """Truncating contraction (chi well below the exact bond): the device amplitudes follow the oracle run with the
    same scheme and parameters, and differ from the SVD-compressed ones (so the scheme is really in effect)."""
    sitps = synthetic.make_sitps(L, D, noise=1.0)
    cfgs = synthetic.make_configs(L, 4, "heisenberg")
    tp = getattr(BMPSTruncateParams, scheme)(chi, chi, 0.0, 1e-13, 30)
    ref = np.array([vmc.TPSWaveFunctionComponent(sitps, c, tp).amplitude for c in cfgs])
    svd = np.array([vmc.TPSWaveFunctionComponent(sitps, c, BMPSTruncateParams.SVD(chi, chi, 0.0)).amplitude for c in cfgs])
    ctx = _ctx(L, D, 2, chi, chi, 0.0, scheme, 1e-13, 30, dt, len(cfgs))
    ctx.state_upload(synthetic.sitps_to_flat(sitps, D, np.float64))
    ctx.set_configs(cfgs)
    got = ctx.evaluate_amplitude()
    assert np.all(ctx.walker_flags() == 0)
    err = np.max(np.abs(got / ref - 1))
    assert err < tol, (got, ref, svd)
    if dt == "f64":
        assert np.max(np.abs(svd / ref - 1)) > 10 * err


@pytest.mark.parametrize("dt,tol", [("f64", 1e-7), ("f32", 2e-4)])
@pytest.mark.parametrize("scheme", list(SCHEMES))
def test_variational_rank_adaptive_on_low_rank_state(scheme, dt, tol):
    """A state whose boundary bonds have numerical rank well below chi (noise 0.1): the sweeps contract over per-walker live
    extents that differ between walkers and bonds (engine_var.h: compact / masked legs of ein(), live rows of svd_rows);
    amplitudes against the oracle run with the same scheme."""
    L, D, chi = 6, 4, 12
    sitps = synthetic.make_sitps(L, D, noise=0.1)
    cfgs = synthetic.make_configs(L, 8, "heisenberg")
    tp = getattr(BMPSTruncateParams, scheme)(chi, chi, 0.0, 1e-13, 10)
    ref = np.array([vmc.TPSWaveFunctionComponent(sitps, c, tp).amplitude for c in cfgs])
    ctx = _ctx(L, D, 2, chi, chi, 0.0, scheme, 1e-13, 10, dt, len(cfgs))
    ctx.state_upload(synthetic.sitps_to_flat(sitps, D, np.float64))
    ctx.set_configs(cfgs)
    got = ctx.evaluate_amplitude()
    assert np.all(ctx.walker_flags() == 0)
    assert np.max(np.abs(got / ref - 1)) < tol, (got, ref)


def test_set_truncate_params_switches_scheme_and_validates():
    """SetTruncateParams (bmps_contractor.h:216): switching the scheme on a live context; bad parameters are refused."""
    from peps_amd import capi
    L, D, chi = 5, 3, 4
    sitps = synthetic.make_sitps(L, D, noise=1.0)
    cfgs = synthetic.make_configs(L, 3, "heisenberg")
    ctx = capi.Context(L, L, D, 2, chi, dtype=capi.F64, max_walkers=3)
    ctx.state_upload(synthetic.sitps_to_flat(sitps, D, np.float64))
    ctx.set_configs(cfgs)
    a_svd = ctx.evaluate_amplitude()
    ctx.set_truncate_params(chi, chi, 0.0, capi.VARIATION2SITE, 1e-13, 30)
    ctx.set_configs(cfgs)
    a_var = ctx.evaluate_amplitude()
    tp = BMPSTruncateParams.Variational2Site(chi, chi, 0.0, 1e-13, 30)
    ref = np.array([vmc.TPSWaveFunctionComponent(sitps, c, tp).amplitude for c in cfgs])
    assert np.max(np.abs(a_var / ref - 1)) < 1e-7
    assert np.max(np.abs(a_svd / ref - 1)) > 1e-6
    with pytest.raises(ValueError):
        ctx.set_truncate_params(chi, chi, 0.0, 7, 1e-13, 30)
    with pytest.raises(ValueError):
        ctx.set_truncate_params(chi, chi, 0.0, capi.VARIATION1SITE, 1e-13, 0)
    with pytest.raises(ValueError):
        ctx.set_truncate_params(chi + 1, chi, 0.0, capi.SVD_COMPRESS, 0.0, 0)
