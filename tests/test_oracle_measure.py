"""Oracle restatement of the measurement solver pinned by definition-level checks: the registry energy
equals the energy solver's, bond energies add up, the off-diagonal row channel equals brute-force amplitude ratios,
and the psi summary follows psi_consistency.h."""
import numpy as np

from oracle import vmc
from oracle.bmps import BMPSTruncateParams
from peps_amd import synthetic


def test_registry_observables_are_consistent_with_energy_solver_and_amplitude_ratios():
    L, D, chi = 4, 3, 9
    s = synthetic.make_sitps(L, D)
    tp = BMPSTruncateParams.SVD(chi, chi, 0.0)
    for cfg in synthetic.make_configs(L, 3, "heisenberg"):
        comp = vmc.TPSWaveFunctionComponent(s, cfg, tp)
        for model in (vmc.SquareSpinOneHalfXXZModelOBC(1.0, 0.7, 0.2), vmc.SquareSpinOneHalfJ1J2XXZModelOBC(1, 1, 0.5, 0.5)):
            ms = vmc.SquareNNNModelMeasurementSolver(model)
            obs = ms.EvaluateObservables(s, comp)
            e, _, psi_list = model.CalEnergyAndHoles(s, comp, False)
            assert abs(obs["energy"][0] - e) < 1e-12 * max(1.0, abs(e))
            bonds = sum(np.sum(obs[k]) for k in obs if k.startswith("bond_energy"))
            assert abs(bonds + model.EvaluateTotalOnsiteEnergy(comp.config) - e) < 1e-10 * max(1.0, abs(e))
            assert len(obs["SzSz_all2all"]) == L * L * (L * L + 1) // 2 and abs(obs["SzSz_all2all"][0] - 0.25) < 1e-15
            mean, rel = ms.last_psi_summary
            assert abs(mean - np.mean(psi_list)) < 1e-12 * abs(mean) and rel < 1e-10
            row, x0 = L // 2, L // 4
            chan = "SpSm_row" if cfg[row, x0] == 0 else "SmSp_row"
            other = "SmSp_row" if chan == "SpSm_row" else "SpSm_row"
            assert not np.any(obs[other])
            for i in range(1, L // 2 + 1):
                c2 = cfg.copy()
                if c2[row, x0] == c2[row, x0 + i]:
                    assert obs[chan][i - 1] == 0.0
                    continue
                c2[row, x0], c2[row, x0 + i] = c2[row, x0 + i], c2[row, x0]
                ratio = vmc.TPSWaveFunctionComponent(s, c2, tp).amplitude / comp.amplitude
                assert abs(obs[chan][i - 1] - ratio) < 1e-10 * max(1.0, abs(ratio))


def test_psi_consistency_summary_aligns_sign_branches():
    """psi_consistency.h:60-107: samples with negative overlap to the largest one are flipped before averaging"""
    mean, rel = vmc.compute_psi_consistency_summary_aligned([1.0, -1.02, 0.98])
    assert abs(abs(mean) - 1.0) < 1e-12 and abs(rel - 0.02) < 1e-12
    assert vmc.compute_psi_consistency_summary_aligned([]) == (0.0, 0.0)
    mean, rel = vmc.compute_psi_consistency_summary_aligned([2.0, 2.0])
    assert mean == 2.0 and rel == 0.0


def test_structure_factor_equals_amplitudes_of_doubly_flipped_configurations():
    """measure_structure_factor (BMPSWalker flow of structure_factor_measurement_mixin.h) at an exact chi: every open
    channel equals the amplitude of the configuration with (y1,x1) raised and (y2,x2) lowered, closed channels are 0."""
    L, D, chi = 4, 3, 27
    s = synthetic.make_sitps(L, D)
    tp = BMPSTruncateParams.SVD(chi, chi, 0.0)
    cfg = synthetic.make_configs(L, 2, "heisenberg")[1]
    comp = vmc.TPSWaveFunctionComponent(s, cfg, tp)
    t = np.array(vmc.measure_structure_factor(s, comp)).reshape(-1, 5)
    assert len(t) == sum(L * (L - 1 - y1) * L for y1 in range(L - 1))
    n_open = 0
    for y1, x1, y2, x2, val in t:
        y1, x1, y2, x2 = int(y1), int(x1), int(y2), int(x2)
        assert y2 > y1
        if cfg[y1, x1] == 0 and cfg[y2, x2] == 1:
            c2 = cfg.copy()
            c2[y1, x1], c2[y2, x2] = 1, 0
            ref = vmc.TPSWaveFunctionComponent(s, c2, tp).amplitude
            assert abs(val / ref - 1) < 1e-9
            n_open += 1
        else:
            assert val == 0.0
    assert n_open > 10
    # the measurement leaves the contractor usable: a fresh amplitude still comes out right
    assert abs(comp.EvaluateAmplitude() / vmc.TPSWaveFunctionComponent(s, cfg, tp).amplitude - 1) < 1e-12
