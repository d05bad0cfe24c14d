"""Wave-function component, MC sweep updaters, model energy solvers, exact-summation evaluator.

Restates (bosonic paths):
  include/qlpeps/vmc_basic/wave_function_component.h:136-379
  include/qlpeps/vmc_basic/configuration_update_strategies/square_nn_updater.h:25-189,253-293
  include/qlpeps/vmc_basic/monte_carlo_tools/suwa_todo_update.h:53-112
  include/qlpeps/algorithm/vmc_update/model_solvers/base/square_nnn_energy_solver.h:79-316
  include/qlpeps/algorithm/vmc_update/model_solvers/base/bond_traversal_mixin.h:113-144
  include/qlpeps/algorithm/vmc_update/model_solvers/square_spin_onehalf_xxz_obc.h:72-104
  include/qlpeps/algorithm/vmc_update/model_solvers/transverse_field_ising_square_obc.h:160-247
  include/qlpeps/algorithm/vmc_update/model_solvers/spin_onehalf_triangle_heisenberg_sqrpeps.h:39-112
  include/qlpeps/algorithm/vmc_update/model_solvers/spin_onehalf_triangle_heisenbergJ1J2_sqrpeps.h:48-463
  include/qlpeps/algorithm/vmc_update/exact_summation_energy_evaluator.h:74-95,173-302
Oracle = test infrastructure only.
"""
import itertools
import numpy as np

from .bmps import LEFT, DOWN, RIGHT, UP, HORIZONTAL, VERTICAL
from .contractor import TensorNetwork2D, BMPSContractor, LEFTUP_TO_RIGHTDOWN, LEFTDOWN_TO_RIGHTUP


class TPSWaveFunctionComponent:
    """wave_function_component.h:136-379.  `sitps[r][c]` = list (physical index) of rank-4 arrays."""

    def __init__(self, sitps, config, trun_para):
        self.config = np.array(config, dtype=np.int64)
        self.trun_para = trun_para
        rows, cols = self.config.shape
        self.tn = TensorNetwork2D.from_sitps(sitps, self.config)          # :159
        self.contractor = BMPSContractor(rows, cols)
        self.contractor.Init(self.tn)                                     # :160
        self.amplitude = None
        self.EvaluateAmplitude()                                          # :161

    def EvaluateAmplitude(self):
        """wave_function_component.h:187-212"""
        c, tn = self.contractor, self.tn
        c.SetTruncateParams(self.trun_para)
        c.GrowBMPSForRow(tn, 0)
        c.GrowFullBTen(tn, RIGHT, 0, 2, True)
        c.InitBTen(tn, LEFT, 0)
        self.amplitude = c.Trace(tn, (0, 0), HORIZONTAL)
        return self.amplitude

    def UpdateLocal(self, sitps, new_amplitude, *site_configs):
        """wave_function_component.h:345-378"""
        for site, cfg in site_configs:
            self.config[site[0], site[1]] = cfg
            self.tn.update_site_tensor(site, cfg, sitps)
            self.contractor.EraseEnvsAfterUpdate(site)
        self.amplitude = new_amplitude


# ---------------------------------------------------------------------------------------------
def suwa_todo_state_update(init_state, weights, uniform01):
    """suwa_todo_update.h:53-112.  `uniform01()` returns a deviate in [0,1); the reference draws
    uniform_real_distribution<long double>(start, start+w_i) from the engine."""
    w = [float(x) for x in weights]
    n = len(w)
    max_id = int(np.argmax(w))
    if max_id != 0:
        w[0], w[max_id] = w[max_id], w[0]
    if init_state == max_id:
        init_state = 0
    elif init_state == 0:
        init_state = max_id
    s = np.cumsum(np.array(w, dtype=np.longdouble))
    S = s[-1]
    s_im1 = np.longdouble(0) if init_state == 0 else s[init_state - 1]
    start = s_im1 + np.longdouble(w[0])
    if start >= S:
        start -= S
    hi = np.nextafter(start + np.longdouble(w[init_state]), start)
    x = np.longdouble(uniform01()) * (hi - start) + start
    if x >= S:
        x -= S
    final = int(np.searchsorted(s, x, side="right"))
    final = min(final, n - 1)
    if max_id != 0:
        if final == 0:
            final = max_id
        elif final == max_id:
            final = 0
    return final


class StdMT19937:
    """std::mt19937 + libstdc++ std::generate_canonical, restated so that the oracle draws the SAME
    deviates as the C++ updaters (monte_carlo_sweep_updater_base.h:18-47 uses std::mt19937 and
    std::uniform_real_distribution<double>(0,1); suwa_todo_update.h:96-101 draws a long double)."""

    def __init__(self, seed):
        self.bg = np.random.MT19937()
        self.bg._legacy_seeding(int(seed))        # init_genrand(seed) == std::mt19937(seed)

    def raw(self):
        return int(self.bg.random_raw())

    def u_double(self):
        s = float(self.raw()) * 1.0
        s = s + float(self.raw()) * 4294967296.0
        r = s / 18446744073709551616.0
        return r if r < 1.0 else float(np.nextafter(1.0, 0.0))

    def u_longdouble(self):
        s = np.longdouble(self.raw())
        s = s + np.longdouble(self.raw()) * np.longdouble(4294967296.0)
        r = s / (np.longdouble(4294967296.0) * np.longdouble(4294967296.0))
        return r if r < 1 else np.nextafter(np.longdouble(1), np.longdouble(0))


class MCUpdateSquareNNUpdateBaseOBC:
    """square_nn_updater.h:25-83: sweep schedule."""

    def __init__(self, seed=0):
        self.rng = StdMT19937(seed)

    def u_double(self):
        return self.rng.u_double()

    def u_longdouble(self):
        return self.rng.u_longdouble()

    def __call__(self, sitps, comp):
        tn, c = comp.tn, comp.contractor
        accept = 0
        c.SetTruncateParams(comp.trun_para)
        c.GenerateBMPSApproach(tn, UP)
        for row in range(tn.rows):
            c.InitBTen(tn, LEFT, row)
            c.GrowFullBTen(tn, RIGHT, row, 2, True)
            for col in range(tn.cols - 1):
                accept += self.two_site_update((row, col), (row, col + 1), HORIZONTAL, sitps, comp)
                if col < tn.cols - 2:
                    c.ShiftBTenWindow(tn, RIGHT)
            if row < tn.rows - 1:
                c.ShiftBMPSWindow(tn, DOWN)
        c.DeleteInnerBMPS(LEFT)
        c.DeleteInnerBMPS(RIGHT)
        c.GenerateBMPSApproach(tn, LEFT)
        for col in range(tn.cols):
            c.InitBTen(tn, UP, col)
            c.GrowFullBTen(tn, DOWN, col, 2, True)
            for row in range(tn.rows - 1):
                accept += self.two_site_update((row, col), (row + 1, col), VERTICAL, sitps, comp)
                if row < tn.rows - 2:
                    c.ShiftBTenWindow(tn, DOWN)
            if col < tn.cols - 1:
                c.ShiftBMPSWindow(tn, RIGHT)
        c.DeleteInnerBMPS(UP)
        bond_num = tn.cols * (tn.rows - 1) + tn.rows * (tn.cols - 1)
        return [accept / bond_num]


class MCUpdateSquareNNExchangeOBC(MCUpdateSquareNNUpdateBaseOBC):
    """square_nn_updater.h:142-189"""

    def two_site_update(self, s1, s2, bond_dir, sitps, comp):
        c1, c2 = int(comp.config[s1]), int(comp.config[s2])
        if c1 == c2:
            return False
        psi_b = comp.contractor.ReplaceNNSiteTrace(comp.tn, s1, s2, bond_dir,
                                                   sitps[s1[0]][s1[1]][c2], sitps[s2[0]][s2[1]][c1])
        psi_a = comp.amplitude
        if abs(psi_b) < abs(psi_a):
            div = abs(psi_b) / abs(psi_a)
            if not (self.u_double() < div * div):
                return False
        comp.UpdateLocal(sitps, psi_b, (s1, c2), (s2, c1))
        return True


class MCUpdateSquareNNFullSpaceUpdateOBC(MCUpdateSquareNNUpdateBaseOBC):
    """square_nn_updater.h:253-293"""

    def two_site_update(self, s1, s2, bond_dir, sitps, comp):
        dim = len(sitps[0][0])
        init = int(comp.config[s1]) * dim + int(comp.config[s2])
        alt = [None] * (dim * dim)
        alt[init] = comp.amplitude
        for k1 in range(dim):
            for k2 in range(dim):
                k = k1 * dim + k2
                if k != init:
                    alt[k] = comp.contractor.ReplaceNNSiteTrace(comp.tn, s1, s2, bond_dir,
                                                                sitps[s1[0]][s1[1]][k1], sitps[s2[0]][s2[1]][k2])
        weights = [abs(a / comp.amplitude) ** 2 for a in alt]
        final = suwa_todo_state_update(init, weights, self.u_longdouble)
        if final == init:
            return False
        comp.UpdateLocal(sitps, alt[final], (s1, final // dim), (s2, final % dim))
        return True


# ---------------------------------------------------------------------------------------------
class SquareNNModelEnergySolver:
    """square_nnn_energy_solver.h:37-316 (has_nnn_interaction = false unless a subclass sets it;
    the NNN pass is :203-265) + bond_traversal_mixin.h:113-144 for the vertical pass."""
    has_nnn_interaction = False

    def CalEnergyAndHoles(self, sitps, comp, calchols=True):
        tn, c = comp.tn, comp.contractor
        rows, cols = tn.rows, tn.cols
        holes = [[None] * cols for _ in range(rows)]
        bond_e, psi_list = [], []
        c.SetTruncateParams(comp.trun_para)
        c.GenerateBMPSApproach(tn, UP)                                    # :116
        for row in range(rows):
            c.InitBTen(tn, LEFT, row)                                     # :142
            c.GrowFullBTen(tn, RIGHT, row, 1, True)                       # :143
            psi = c.Trace(tn, (row, 0), HORIZONTAL)                       # :147
            if psi == 0:
                raise RuntimeError("Wavefunction amplitude is near zero, causing division by zero.")
            inv_psi = 1.0 / psi
            psi_list.append(psi)
            for col in range(cols):
                s1 = (row, col)
                if calchols:
                    holes[row][col] = np.conj(c.PunchHole(tn, s1, HORIZONTAL))   # :163 Dag(env)
                if col < cols - 1:
                    s2 = (row, col + 1)
                    bond_e.append(self.EvaluateBondEnergy(s1, s2, int(comp.config[s1]), int(comp.config[s2]),
                                                          HORIZONTAL, tn, c, sitps[row][col], sitps[row][col + 1],
                                                          inv_psi))
                    c.ShiftBTenWindow(tn, RIGHT)                          # :200
            if self.has_nnn_interaction and row < rows - 1:               # :203-265
                c.InitBTen2(tn, LEFT, row)
                c.GrowFullBTen2(tn, RIGHT, row, 2, True)
                for col in range(cols - 1):
                    s1, s2 = (row, col), (row + 1, col + 1)
                    e = self.EvaluateNNNEnergy(s1, s2, int(comp.config[s1]), int(comp.config[s2]), LEFTUP_TO_RIGHTDOWN,
                                               tn, c, sitps[s1[0]][s1[1]], sitps[s2[0]][s2[1]], inv_psi)
                    s1, s2 = (row + 1, col), (row, col + 1)
                    e = e + self.EvaluateNNNEnergy(s1, s2, int(comp.config[s1]), int(comp.config[s2]), LEFTDOWN_TO_RIGHTUP,
                                                   tn, c, sitps[s1[0]][s1[1]], sitps[s2[0]][s2[1]], inv_psi)
                    bond_e.append(e)
                    c.ShiftBTen2Window(tn, RIGHT, row)
            if row < rows - 1:
                c.ShiftBMPSWindow(tn, DOWN)                               # :126
        # vertical pass: bond_traversal_mixin.h:113-144
        c.GenerateBMPSApproach(tn, LEFT)
        for col in range(cols):
            c.InitBTen(tn, UP, col)
            c.GrowFullBTen(tn, DOWN, col, 2, True)
            psi = c.Trace(tn, (0, col), VERTICAL)
            if psi == 0:
                raise RuntimeError("Wavefunction amplitude is near zero, causing division by zero.")
            inv_psi = 1.0 / psi
            psi_list.append(psi)
            for row in range(rows - 1):
                s1, s2 = (row, col), (row + 1, col)
                bond_e.append(self.EvaluateBondEnergy(s1, s2, int(comp.config[s1]), int(comp.config[s2]),
                                                      VERTICAL, tn, c, sitps[row][col], sitps[row + 1][col],
                                                      inv_psi))
                if row < rows - 2:
                    c.ShiftBTenWindow(tn, DOWN)
            if col < cols - 1:
                c.ShiftBMPSWindow(tn, RIGHT)
        energy = sum(bond_e) + self.EvaluateTotalOnsiteEnergy(comp.config)   # :97-101
        return energy, holes, psi_list


class SquareSpinOneHalfXXZModelOBC(SquareNNModelEnergySolver):
    """square_spin_onehalf_xxz_obc.h:64-190"""

    def __init__(self, jz=1.0, jxy=1.0, pinning00=0.0):
        self.jz, self.jxy, self.pin = jz, jxy, pinning00

    def EvaluateBondEnergy(self, s1, s2, c1, c2, orient, tn, contractor, t1, t2, inv_psi):
        """:72-104"""
        if c1 == c2:
            return 0.25 * self.jz
        psi_ex = contractor.ReplaceNNSiteTrace(tn, s1, s2, orient, t1[c2], t2[c1])
        ratio = np.conj(psi_ex * inv_psi)
        return -0.25 * self.jz + ratio * 0.5 * self.jxy

    def EvaluateTotalOnsiteEnergy(self, config):
        """:139-141"""
        return -self.pin * (float(config[0, 0]) - 0.5)


class SquareSpinOneHalfJ1J2XXZModelOBC(SquareSpinOneHalfXXZModelOBC):
    """square_spin_onehalf_j1j2_xxz_obc.h:25-40 + SquareSpinOneHalfXXZModelMixIn::EvaluateNNNEnergy
    (square_spin_onehalf_xxz_obc.h:107-134)"""
    has_nnn_interaction = True

    def __init__(self, jz=1.0, jxy=1.0, jz2=0.0, jxy2=0.0, pinning00=0.0):
        super().__init__(jz, jxy, pinning00)
        self.jz2, self.jxy2 = jz2, jxy2

    def EvaluateNNNEnergy(self, s1, s2, c1, c2, diagonal_dir, tn, contractor, t1, t2, inv_psi):
        if c1 == c2:
            return 0.25 * self.jz2
        left_up = s1 if diagonal_dir == LEFTUP_TO_RIGHTDOWN else (s2[0], s1[1])
        psi_ex = contractor.ReplaceNNNSiteTrace(tn, left_up, diagonal_dir, HORIZONTAL, t1[c2], t2[c1])
        ratio = np.conj(psi_ex * inv_psi)
        return -0.25 * self.jz2 + ratio * 0.5 * self.jxy2


class SpinOneHalfTriHeisenbergSqrPEPS(SquareSpinOneHalfJ1J2XXZModelOBC):
    """spin_onehalf_triangle_heisenberg_sqrpeps.h:39-112: Heisenberg model of the triangular lattice on a square PEPS -- J = 1 on the
    nearest-neighbour bonds and on ONE diagonal of every plaquette (LEFTDOWN_TO_RIGHTUP); the other diagonal contributes 0 (:98-100)."""

    def __init__(self):
        super().__init__(1.0, 1.0, 1.0, 1.0, 0.0)

    def EvaluateNNNEnergy(self, s1, s2, c1, c2, diagonal_dir, tn, contractor, t1, t2, inv_psi):
        if diagonal_dir != LEFTDOWN_TO_RIGHTUP:
            return 0.0
        return super().EvaluateNNNEnergy(s1, s2, c1, c2, diagonal_dir, tn, contractor, t1, t2, inv_psi)


class SpinOneHalfTriJ1J2HeisenbergSqrPEPS:
    """spin_onehalf_triangle_heisenbergJ1J2_sqrpeps.h:48-446: J1-J2 Heisenberg model of the triangular lattice on a square PEPS.
    J1 = 1 bonds: horizontal, vertical and the left-down -> right-up diagonal of every plaquette; J2 bonds (distance sqrt 3 on the
    triangular lattice): the other plaquette diagonal (r, c)-(r+1, c+1), the flat sqrt5 link (r+1, c)-(r, c+2) of a 2 x 3 window and
    the steep sqrt5 link (r+2, c)-(r, c+1) of a 3 x 2 window.  Own traversal (CalEnergyAndHolesImpl :304-446): the row pass carries
    the horizontal bonds on BTen and both diagonals + the flat link on BTen2; the column pass carries the vertical bonds on BTen and
    the steep link on BTen2 (GrowFullBTen2(DOWN, col, 3))."""

    def __init__(self, j2):
        self.j2 = j2

    @staticmethod
    def _bond(c1, c2, psi_ex_fn, inv_psi):
        """0.25 for equal spins, else -0.25 + 0.5 conj(psi_ex / psi)   (:338-345 and the seven other bond blocks)"""
        if c1 == c2:
            return 0.25
        return -0.25 + np.conj(psi_ex_fn() * inv_psi) * 0.5

    def traverse(self, sitps, comp, calchols, on_bond):
        """The bond traversal of :317-444; on_bond(kind, s1, s2, value) for kind in h / v / ur (J1) and dr / flat / steep (J2)."""
        tn, c, cfg = comp.tn, comp.contractor, comp.config
        rows, cols = tn.rows, tn.cols
        holes = [[None] * cols for _ in range(rows)]
        psi_list = []
        T = lambda s: sitps[s[0]][s[1]]
        c.SetTruncateParams(comp.trun_para)
        c.GenerateBMPSApproach(tn, UP)                                    # :317
        for row in range(rows):
            c.InitBTen(tn, LEFT, row)                                     # :320
            c.GrowFullBTen(tn, RIGHT, row, 1, True)
            psi = c.Trace(tn, (row, 0), HORIZONTAL)                       # :322
            inv_psi = 1.0 / psi
            psi_list.append(psi)
            for col in range(cols):
                s1 = (row, col)
                if calchols:
                    holes[row][col] = np.conj(c.PunchHole(tn, s1, HORIZONTAL))          # :329
                if col < cols - 1:
                    s2 = (row, col + 1)
                    c1, c2 = int(cfg[s1]), int(cfg[s2])
                    on_bond("h", s1, s2, self._bond(c1, c2, lambda: c.ReplaceNNSiteTrace(tn, s1, s2, HORIZONTAL, T(s1)[c2], T(s2)[c1]), inv_psi))
                    c.ShiftBTenWindow(tn, RIGHT)                          # :346
            if row < rows - 1:
                c.InitBTen2(tn, LEFT, row)                                # :350
                c.GrowFullBTen2(tn, RIGHT, row, 2, True)
                for col in range(cols - 1):
                    s1, s2 = (row + 1, col), (row, col + 1)               # :355-367 J1 diagonal
                    c1, c2 = int(cfg[s1]), int(cfg[s2])
                    on_bond("ur", s1, s2, self._bond(c1, c2, lambda: c.ReplaceNNNSiteTrace(
                        tn, (row, col), LEFTDOWN_TO_RIGHTUP, HORIZONTAL, T(s1)[c2], T(s2)[c1]), inv_psi))
                    s1, s2 = (row, col), (row + 1, col + 1)               # :369-381 J2 diagonal
                    c1, c2 = int(cfg[s1]), int(cfg[s2])
                    on_bond("dr", s1, s2, self._bond(c1, c2, lambda: c.ReplaceNNNSiteTrace(
                        tn, (row, col), LEFTUP_TO_RIGHTDOWN, HORIZONTAL, T(s1)[c2], T(s2)[c1]), inv_psi))
                    if col < cols - 2:                                    # :383-397 flat sqrt5 link
                        s1, s2 = (row + 1, col), (row, col + 2)
                        c1, c2 = int(cfg[s1]), int(cfg[s2])
                        on_bond("flat", s1, s2, self._bond(c1, c2, lambda: c.ReplaceSqrt5DistTwoSiteTrace(
                            tn, (row, col), LEFTDOWN_TO_RIGHTUP, HORIZONTAL, T(s1)[c2], T(s2)[c1]), inv_psi))
                    c.ShiftBTen2Window(tn, RIGHT, row)                    # :398
                c.ShiftBMPSWindow(tn, DOWN)                               # :400
        c.GenerateBMPSApproach(tn, LEFT)                                  # :404
        for col in range(cols):
            c.InitBTen(tn, UP, col)
            c.GrowFullBTen(tn, DOWN, col, 2, True)
            psi = c.Trace(tn, (0, col), VERTICAL)
            inv_psi = 1.0 / psi
            psi_list.append(psi)
            for row in range(rows - 1):
                s1, s2 = (row, col), (row + 1, col)
                c1, c2 = int(cfg[s1]), int(cfg[s2])
                on_bond("v", s1, s2, self._bond(c1, c2, lambda: c.ReplaceNNSiteTrace(tn, s1, s2, VERTICAL, T(s1)[c2], T(s2)[c1]), inv_psi))
                if row < rows - 2:
                    c.ShiftBTenWindow(tn, DOWN)
            if col < cols - 1:
                c.InitBTen2(tn, UP, col)                                  # :425
                c.GrowFullBTen2(tn, DOWN, col, 3, True)
                for row in range(rows - 2):                               # :428-442 steep sqrt5 link
                    s1, s2 = (row + 2, col), (row, col + 1)
                    c1, c2 = int(cfg[s1]), int(cfg[s2])
                    on_bond("steep", s1, s2, self._bond(c1, c2, lambda: c.ReplaceSqrt5DistTwoSiteTrace(
                        tn, (row, col), LEFTDOWN_TO_RIGHTUP, VERTICAL, T(s1)[c2], T(s2)[c1]), inv_psi))
                    if row < rows - 3:
                        c.ShiftBTen2Window(tn, DOWN, col)
                c.ShiftBMPSWindow(tn, RIGHT)
        return holes, psi_list

    def CalEnergyAndHoles(self, sitps, comp, calchols=True):
        e = {"h": 0.0, "v": 0.0, "ur": 0.0, "dr": 0.0, "flat": 0.0, "steep": 0.0}

        def on_bond(kind, s1, s2, val):
            e[kind] = e[kind] + val
        holes, psi_list = self.traverse(sitps, comp, calchols, on_bond)
        return (e["h"] + e["v"] + e["ur"]) + self.j2 * (e["dr"] + e["flat"] + e["steep"]), holes, psi_list      # :445

    def EvaluateObservables(self, sitps, comp):
        """Registry of :65-277: energy, spin_z, bond_energy_h / v / ur (J1 bonds only; the J2 links enter the energy scalar),
        SzSz_row / SmSp_row / SpSm_row along the middle row, SzSz_all2all (packed upper triangle).  The reference takes the
        diagonal ratios against a second Trace of the row (`psi2`, :187) whose BTen window has been shifted away by then; the row's
        own psi -- the same amplitude -- is used here."""
        tn, cfg = comp.tn, comp.config
        ly, lx = tn.rows, tn.cols
        out = {"spin_z": [float(v) - 0.5 for v in cfg.ravel()]}
        e_h = np.zeros((ly, max(lx - 1, 0)), dtype=complex); e_v = np.zeros((max(ly - 1, 0), lx), dtype=complex)
        e_ur = np.zeros((max(ly - 1, 0), max(lx - 1, 0)), dtype=complex)
        tot = {"j1": 0.0, "j2": 0.0}

        def on_bond(kind, s1, s2, val):
            if kind == "h":
                e_h[s1] = val
            elif kind == "v":
                e_v[s1] = val
            elif kind == "ur":
                e_ur[s2[0], s1[1]] = val
            tot["j1" if kind in ("h", "v", "ur") else "j2"] += val
        _, psi_list = self.traverse(sitps, comp, False, on_bond)
        real = not np.iscomplexobj(psi_list[0])
        cast = (lambda a: [float(np.real(x)) for x in a.ravel()]) if real else (lambda a: list(a.ravel()))
        out["energy"] = [tot["j1"] + self.j2 * tot["j2"]]
        out["bond_energy_h"], out["bond_energy_v"], out["bond_energy_ur"] = cast(e_h), cast(e_v), cast(e_ur)
        row = ly // 2                                                      # :131-171
        sz1 = float(cfg[row, lx // 4]) - 0.5
        out["SzSz_row"] = [sz1 * (float(cfg[row, lx // 4 + i]) - 0.5) for i in range(1, lx // 2 + 1)]
        c = comp.contractor
        c.GenerateBMPSApproach(tn, UP)
        for _ in range(row):
            c.ShiftBMPSWindow(tn, DOWN)
        c.InitBTen(tn, LEFT, row)
        c.GrowFullBTen(tn, RIGHT, row, 1, True)
        inv_psi = 1.0 / c.Trace(tn, (row, 0), HORIZONTAL)
        for _ in range(lx - 1):
            c.ShiftBTenWindow(tn, RIGHT)
        corr = measure_spin_onehalf_off_diag_order_in_row(sitps, comp, inv_psi, row)
        zero = [0.0] * len(corr)
        if int(cfg[row, lx // 4]) == 0:
            out["SmSp_row"], out["SpSm_row"] = zero, corr
        else:
            out["SmSp_row"], out["SpSm_row"] = corr, zero
        flat = [int(v) for v in cfg.ravel()]                               # :260-275, :449-463: +-0.25, diagonal 0.25
        out["SzSz_all2all"] = [0.25 if flat[i] == flat[j] else -0.25 for i in range(len(flat)) for j in range(i, len(flat))]
        self.last_psi_summary = compute_psi_consistency_summary_aligned(psi_list)
        return out


def compute_psi_consistency_summary_aligned(psi_list):
    """ComputePsiConsistencySummaryAligned (algorithm/vmc_update/psi_consistency.h:60-107): (mean, max rel. deviation)
    after flipping the samples whose overlap with the largest-magnitude one is negative."""
    psi = np.asarray(psi_list)
    if psi.size == 0:
        return 0.0, 0.0
    ref = psi[int(np.argmax(np.abs(psi)))]
    aligned = psi.copy()
    if abs(ref) > 1e-14:
        flip = np.real(psi * np.conj(ref)) < 0.0
        aligned[flip] = -aligned[flip]
    mean = aligned.sum() / psi.size
    denom = max(abs(mean), np.finfo(np.float64).eps)
    return mean, float(np.max(np.abs(aligned - mean)) / denom)


def measure_spin_onehalf_off_diag_order_in_row(sitps, comp, inv_psi, row):
    """MeasureSpinOneHalfOffDiagOrderInRow (model_solvers/square_spin_onehalf_xxz_obc.h:22-60): the valid channel of
    <S+(x0) S-(x0+i)> / <S-(x0) S+(x0+i)> along `row`, x0 = lx/4, i = 1..lx/2; the BMPS of the row must exist."""
    tn, c, config = comp.tn, comp.contractor, comp.config
    lx = tn.cols
    site1 = (row, lx // 4)
    out = [0.0] * (lx // 2)
    tn.update_site_tensor(site1, 1 - int(config[site1]), sitps)
    c.EraseEnvsAfterUpdate(site1)
    c.CheckInvalidateEnvs(site1)
    c.GrowBTenStep(tn, LEFT)
    c.GrowFullBTen(tn, RIGHT, row, lx // 4 + 2, False)
    for i in range(1, lx // 2 + 1):
        site2 = (row, lx // 4 + i)
        c.CheckInvalidateEnvs(site2)
        if config[site2] != config[site1]:
            psi_ex = c.ReplaceOneSiteTrace(tn, site2, sitps[site2[0]][site2[1]][1 - int(config[site2])], HORIZONTAL)
            out[i - 1] = np.conj(psi_ex * inv_psi)
        c.CheckInvalidateEnvs(site2)
        c.ShiftBTenWindow(tn, RIGHT)
    tn.update_site_tensor(site1, int(config[site1]), sitps)
    c.EraseEnvsAfterUpdate(site1)
    return out


def measure_structure_factor(sitps, comp, reference_stack_state=False):
    """StructureFactorMeasurementMixin::MeasureStructureFactor (model_solvers/base/structure_factor_measurement_mixin.h:
    62-215) with BMPSWalker (two_dim_tn/tensor_network_2d/bmps/impl/bmps_walker.h): a walker forked from the UP vacuum is
    evolved through the excited row y1 (S+ at (y1, x1), source spin down) and the standard rows below; every row y2 > y1
    is closed against the DOWN environment of the rows below it with S- at (y2, x2) (target spin up).  Returns the flat
    tuples [y1, x1, y2, x2, value, ...]; value = amplitude of the doubly flipped configuration, 0 for a closed channel.
    Restated with the contractor's own primitives: the walker = an UP stack that is extended and cut back.

    reference_stack_state = False: the DOWN environments of all rows are grown first (every pair y1 < y2 is measured).
    reference_stack_state = True: the mixin exactly as the reference runs it -- it takes `contractor.GetBMPS(DOWN)` as the
    traversal of EvaluateObservables left it (ONE level after the row pass: the vacuum below the last row), and for a row y2
    whose environment is not in the stack it pushes zeros and `continue`s PAST the walker's Evolve (:139-149, :205-207): the
    values of y2 < Ly - 1 are zero, the walker of y1 < Ly - 2 reaches the last row without the rows between, only
    y1 = Ly - 2 is the amplitude the text above describes.  This form reproduces the 96 golden values of the reference's
    tests/test_model_solvers/test_square_xxz_measurer.cpp:246-345 (tests/test_oracle_measure.py)."""
    tn, c, config = comp.tn, comp.contractor, comp.config
    ly, lx = tn.rows, tn.cols
    out = []
    if not reference_stack_state:
        c.GenerateBMPSApproach(tn, UP)
    up_saved = list(c.bmps_set[UP])
    down_full = list(c.bmps_set[DOWN])
    n_down = len(down_full)
    main = [up_saved[0]]                                  # main_walker = BMPSWalker(tn, up_stack[0], UP, 1, trunc) (:121-122)
    for y1 in range(ly - 1):
        for x1 in range(lx):
            src_down = int(config[y1, x1]) == 0
            if src_down:
                tn.update_site_tensor((y1, x1), 1, sitps)
            c.bmps_set[UP] = list(main)
            c.GrowBMPSStep(tn, UP)                                            # excited_walker.Evolve(excited row y1)
            walker = c.bmps_set[UP][-1]
            if src_down:
                tn.update_site_tensor((y1, x1), int(config[y1, x1]), sitps)   # (the rows below are the standard ones)
            for y2 in range(y1 + 1, ly):
                row = [0.0] * lx
                if ly - 1 - y2 < n_down:
                    c.bmps_set[UP] = [walker] * (y2 + 1)                      # the walker is the UP boundary of row y2
                    c.bmps_set[DOWN] = down_full[:ly - y2]                    # bottom_env = down_stack[ly-1-y2]
                    c.InitBTen(tn, LEFT, y2)
                    c.GrowFullBTen(tn, RIGHT, y2, 1, True)
                    for x2 in range(lx):
                        if src_down and int(config[y2, x2]) == 1:
                            row[x2] = c.ReplaceOneSiteTrace(tn, (y2, x2), sitps[y2][x2][0], HORIZONTAL)
                        if x2 < lx - 1:
                            c.ShiftBTenWindow(tn, RIGHT)
                    if y2 < ly - 1:
                        c.GrowBMPSStep(tn, UP)                                # excited_walker.Evolve(standard row y2)
                        walker = c.bmps_set[UP][-1]
                for x2 in range(lx):
                    out += [float(y1), float(x1), float(y2), float(x2), row[x2]]
        c.bmps_set[UP] = list(main)
        c.bmps_set[DOWN] = list(down_full)
        c.GrowBMPSStep(tn, UP)                                                # main_walker.Evolve(standard row y1)
        main = list(c.bmps_set[UP])
    c.bmps_set[UP] = up_saved
    c.bmps_set[DOWN] = list(down_full)
    for pos in (LEFT, RIGHT, UP, DOWN):
        c.bten_set[pos] = []
    return out


class SquareNNNModelMeasurementSolver:
    """SquareNNNModelMeasurementSolver::EvaluateObservables (model_solvers/base/square_nnn_model_measurement_solver.h:
    30-254) over BondTraversalMixin::TraverseAllBonds (bond_traversal_mixin.h:22-145): registry keys energy, spin_z,
    bond_energy_h/v(/dr/ur) and the psi summary of the sample.  `model` supplies the bond terms (an energy-solver
    model of this module); XXZ adds SzSz_all2all and SmSp_row / SpSm_row (square_spin_onehalf_xxz_obc.h:215-288)."""

    def __init__(self, model, spin_onehalf_xxz=True, structure_factor=False, structure_factor_reference_stack_state=False):
        self.model, self.xxz, self.structure_factor = model, spin_onehalf_xxz, structure_factor
        self.sf_ref_state = structure_factor_reference_stack_state
        self.last_psi_summary = None

    def EvaluateObservables(self, sitps, comp):
        m, tn, c, config = self.model, comp.tn, comp.contractor, comp.config
        ly, lx = tn.rows, tn.cols
        out = {}
        c.SetTruncateParams(comp.trun_para)
        if self.xxz:
            out["spin_z"] = [float(v) - 0.5 for v in config.ravel()]
        dt = complex if np.iscomplexobj(sitps[0][0][0]) else float      # TenElemT of the registry (ObservableMap<TenElemT>)
        e_h = np.zeros((ly, lx - 1), dt); e_v = np.zeros((ly - 1, lx), dt)
        e_dr = np.zeros((ly - 1, lx - 1), dt); e_ur = np.zeros((ly - 1, lx - 1), dt)
        total, psi_list = 0.0, []
        c.GenerateBMPSApproach(tn, UP)
        for row in range(ly):
            c.InitBTen(tn, LEFT, row)
            c.GrowFullBTen(tn, RIGHT, row, 1, True)
            psi = c.Trace(tn, (row, 0), HORIZONTAL)
            inv_psi = 1.0 / psi
            psi_list.append(psi)
            for col in range(lx - 1):
                s1, s2 = (row, col), (row, col + 1)
                e_h[row, col] = m.EvaluateBondEnergy(s1, s2, int(config[s1]), int(config[s2]), HORIZONTAL, tn, c,
                                                     sitps[row][col], sitps[row][col + 1], inv_psi)
                total += e_h[row, col]
                c.ShiftBTenWindow(tn, RIGHT)
            if m.has_nnn_interaction and row < ly - 1:
                c.InitBTen2(tn, LEFT, row)
                c.GrowFullBTen2(tn, RIGHT, row, 2, True)
                for col in range(lx - 1):
                    s1, s2 = (row, col), (row + 1, col + 1)
                    e_dr[row, col] = m.EvaluateNNNEnergy(s1, s2, int(config[s1]), int(config[s2]), LEFTUP_TO_RIGHTDOWN, tn, c,
                                                         sitps[s1[0]][s1[1]], sitps[s2[0]][s2[1]], inv_psi)
                    s1, s2 = (row + 1, col), (row, col + 1)
                    e_ur[row, col] = m.EvaluateNNNEnergy(s1, s2, int(config[s1]), int(config[s2]), LEFTDOWN_TO_RIGHTUP, tn, c,
                                                         sitps[s1[0]][s1[1]], sitps[s2[0]][s2[1]], inv_psi)
                    total += e_dr[row, col] + e_ur[row, col]
                    c.ShiftBTen2Window(tn, RIGHT, row)
            if self.xxz and row == ly // 2:                            # EvaluateOffDiagOrderInRow row hook (:256-288)
                corr = measure_spin_onehalf_off_diag_order_in_row(sitps, comp, inv_psi, row)
                zero = [0.0] * len(corr)
                if int(config[row, lx // 4]) == 0:
                    out["SmSp_row"], out["SpSm_row"] = zero, corr
                else:
                    out["SmSp_row"], out["SpSm_row"] = corr, zero
            if row < ly - 1:
                c.ShiftBMPSWindow(tn, DOWN)
        c.GenerateBMPSApproach(tn, LEFT)
        for col in range(lx):
            c.InitBTen(tn, UP, col)
            c.GrowFullBTen(tn, DOWN, col, 2, True)
            psi = c.Trace(tn, (0, col), VERTICAL)
            inv_psi = 1.0 / psi
            psi_list.append(psi)
            for row in range(ly - 1):
                s1, s2 = (row, col), (row + 1, col)
                e_v[row, col] = m.EvaluateBondEnergy(s1, s2, int(config[s1]), int(config[s2]), VERTICAL, tn, c,
                                                     sitps[row][col], sitps[row + 1][col], inv_psi)
                total += e_v[row, col]
                if row < ly - 2:
                    c.ShiftBTenWindow(tn, DOWN)
            if col < lx - 1:
                c.ShiftBMPSWindow(tn, RIGHT)
        out["energy"] = [total + m.EvaluateTotalOnsiteEnergy(config)]
        out["bond_energy_h"] = list(e_h.ravel())
        out["bond_energy_v"] = list(e_v.ravel())
        if m.has_nnn_interaction:
            out["bond_energy_dr"] = list(e_dr.ravel())
            out["bond_energy_ur"] = list(e_ur.ravel())
        if self.xxz:
            sz = np.asarray(out["spin_z"])
            out["SzSz_all2all"] = [sz[i] * sz[j] for i in range(sz.size) for j in range(i, sz.size)]
        self.last_psi_summary = compute_psi_consistency_summary_aligned(psi_list)
        if self.structure_factor:                                     # square_spin_onehalf_xxz_obc.h:238-248
            out["SpSm_cross"] = measure_structure_factor(sitps, comp, self.sf_ref_state)
        return out


class TransverseFieldIsingSquareOBC:
    """transverse_field_ising_square_obc.h:28-247: H = -sum_<ij> sz sz - h sum_i sx"""

    def __init__(self, h):
        self.h = h

    def CalDiagTermEnergy(self, config):
        """:160-182"""
        cfg = np.asarray(config)
        e = 0.0
        e += np.sum(np.where(cfg[:, :-1] == cfg[:, 1:], -1.0, 1.0))
        e += np.sum(np.where(cfg[:-1, :] == cfg[1:, :], -1.0, 1.0))
        return float(e)

    def CalEnergyAndHoles(self, sitps, comp, calchols=True):
        """:211-247"""
        tn, c = comp.tn, comp.contractor
        rows, cols = tn.rows, tn.cols
        holes = [[None] * cols for _ in range(rows)]
        psi_list = []
        energy = 0.0
        c.SetTruncateParams(comp.trun_para)
        c.GenerateBMPSApproach(tn, UP)
        for row in range(rows):
            c.InitBTen(tn, LEFT, row)
            c.GrowFullBTen(tn, RIGHT, row, 1, True)
            psi = c.Trace(tn, (row, 0), HORIZONTAL)
            inv_psi = 1.0 / psi
            psi_list.append(psi)
            for col in range(cols):
                site = (row, col)
                if calchols:
                    holes[row][col] = np.conj(c.PunchHole(tn, site, HORIZONTAL))
                cfg = int(comp.config[site])
                psi_ex = c.ReplaceOneSiteTrace(tn, site, sitps[row][col][1 - cfg], HORIZONTAL)   # :195-203
                energy = energy + (-self.h) * np.conj(psi_ex * inv_psi)
                if col < cols - 1:
                    c.ShiftBTenWindow(tn, RIGHT)
            if row < rows - 1:
                c.ShiftBMPSWindow(tn, DOWN)
        energy = energy + self.CalDiagTermEnergy(comp.config)
        return energy, holes, psi_list

    def EvaluateObservables(self, sitps, comp):
        """Registry of the model (:60-140): energy, spin_z, sigma_x per site (= -off-diagonal term / h, 0 for h = 0), SzSz_row along the
        middle row (x0 = lx / 4, i = 1 .. lx / 2).  Same row pass as the energy; `last_psi_summary` as the other measurement solvers."""
        tn, c, config = comp.tn, comp.contractor, comp.config
        ly, lx = tn.rows, tn.cols
        out, psi_list, energy_ex = {}, [], 0.0
        cplx = False
        sigma_x = [[0.0] * lx for _ in range(ly)]
        two_point = []
        c.SetTruncateParams(comp.trun_para)
        c.GenerateBMPSApproach(tn, UP)
        for row in range(ly):
            c.InitBTen(tn, LEFT, row)
            c.GrowFullBTen(tn, RIGHT, row, 1, True)
            psi = c.Trace(tn, (row, 0), HORIZONTAL)
            psi_list.append(psi)
            cplx = cplx or np.iscomplexobj(psi)
            inv_psi = 1.0 / psi
            for col in range(lx):
                site = (row, col)
                psi_ex = c.ReplaceOneSiteTrace(tn, site, sitps[row][col][1 - int(config[site])], HORIZONTAL)   # :195-203
                ex = (-self.h) * np.conj(psi_ex * inv_psi)
                energy_ex = energy_ex + ex
                sigma_x[row][col] = (-ex) / self.h if self.h != 0.0 else 0.0                               # :96
                if col < lx - 1:
                    c.ShiftBTenWindow(tn, RIGHT)
            if row == ly // 2:                                                                              # :101-110
                sz1 = float(config[row, lx // 4]) - 0.5
                two_point += [sz1 * (float(config[row, lx // 4 + i]) - 0.5) for i in range(1, lx // 2 + 1)]
            if row < ly - 1:
                c.ShiftBMPSWindow(tn, DOWN)
        out["energy"] = [energy_ex + self.CalDiagTermEnergy(config)]
        out["spin_z"] = [float(v) - 0.5 for v in np.asarray(config).ravel()]
        out["sigma_x"] = [v for r in sigma_x for v in r]
        if two_point:
            out["SzSz_row"] = two_point
        self.last_psi_summary = compute_psi_consistency_summary_aligned(psi_list)
        return out


# ---------------------------------------------------------------------------------------------
def generate_all_permutation_configs(particle_counts, lx, ly):
    """exact_summation_energy_evaluator.h:74-95 (std::next_permutation order)."""
    base = []
    for i, n in enumerate(particle_counts):
        base += [i] * n
    seen = []
    # lexicographic distinct permutations == std::next_permutation sequence from sorted input
    for p in _distinct_perms(base):
        seen.append(np.array(p, dtype=np.int64).reshape(ly, lx))
    return seen


def _distinct_perms(seq):
    seq = sorted(seq)
    n = len(seq)
    while True:
        yield tuple(seq)
        i = n - 2
        while i >= 0 and seq[i] >= seq[i + 1]:
            i -= 1
        if i < 0:
            return
        j = n - 1
        while seq[j] <= seq[i]:
            j -= 1
        seq[i], seq[j] = seq[j], seq[i]
        seq[i + 1:] = reversed(seq[i + 1:])


def all_product_configs(d, lx, ly):
    """All d^(lx*ly) configurations (TFIM has no conserved quantum number)."""
    return [np.array(p, dtype=np.int64).reshape(ly, lx) for p in itertools.product(range(d), repeat=lx * ly)]


def exact_sum_energy_evaluator(sitps, all_configs, trun_para, model, rank=0, size=1):
    """exact_summation_energy_evaluator.h:173-302 (bosonic branch).  Returns
    (energy, gradient[r][c][s], weight_sum) for the configurations i = rank, rank+size, ...
    (the partial sums S_O, S_EO, sum w, sum wE of one rank are returned when size > 1 via
    `partials=True` semantics of exact_sum_partials)."""
    so, seo, wsum, wesum = exact_sum_partials(sitps, all_configs, trun_para, model, rank, size)
    return finish_exact_sum(so, seo, wsum, wesum)


def exact_sum_partials(sitps, all_configs, trun_para, model, rank=0, size=1):
    rows, cols = len(sitps), len(sitps[0])
    d = len(sitps[0][0])
    so = [[[np.zeros_like(sitps[r][c][s]) for s in range(d)] for c in range(cols)] for r in range(rows)]
    seo = [[[np.zeros_like(sitps[r][c][s]) for s in range(d)] for c in range(cols)] for r in range(rows)]
    wsum, wesum = 0.0, 0.0
    for i in range(rank, len(all_configs), size):                           # :201
        comp = TPSWaveFunctionComponent(sitps, all_configs[i], trun_para)
        w = abs(comp.amplitude) ** 2
        e_loc, holes, _ = model.CalEnergyAndHoles(sitps, comp, True)
        for r in range(rows):
            for c in range(cols):
                b = int(comp.config[r, c])
                inc = comp.amplitude * holes[r][c]                          # :231
                so[r][c][b] = so[r][c][b] + inc
                seo[r][c][b] = seo[r][c][b] + np.conj(e_loc) * inc
        wsum += w
        wesum = wesum + e_loc * w
    return so, seo, wsum, wesum


def finish_exact_sum(so, seo, wsum, wesum):
    """exact_summation_energy_evaluator.h:286-295"""
    energy = wesum / wsum
    grad = [[[(seo[r][c][s] - np.conj(energy) * so[r][c][s]) / wsum for s in range(len(so[r][c]))]
             for c in range(len(so[0]))] for r in range(len(so))]
    return energy, grad, wsum


def generate_all_binary_configs(lx, ly):
    """exact_summation_measurer.h:53-72: bit b of the counter i is site (b // lx, b % lx)"""
    n = lx * ly
    if n >= 64:
        raise ValueError("GenerateAllBinaryConfigs: Lx*Ly must be < size_t bit width")
    return [np.array([(i >> b) & 1 for b in range(n)], dtype=np.int64).reshape(ly, lx) for i in range(1 << n)]


def exact_sum_measure(sitps, all_configs, trun_para, make_solver, rank=0, size=1):
    """ExactSumMeasurerMPI (exact_summation_measurer.h:103-257): weight |psi(S)|^2 times the registry of
    solver.EvaluateObservables, summed over configurations rank, rank + size, ... (:130-150).  size == 1: the normalised
    registry (:243-250); else (weighted sums by key, weight sum) of this rank (what MPI_Reduce adds up, :209-235).
    make_solver() -> a fresh SquareNNNModelMeasurementSolver."""
    if len(all_configs) == 0:
        raise ValueError("ExactSumMeasurerMPI: all_configs must not be empty")                    # :113-115
    wsum, acc = 0.0, {}
    for i in range(rank, len(all_configs), size):
        comp = TPSWaveFunctionComponent(sitps, all_configs[i], trun_para)
        w = abs(comp.amplitude) ** 2
        wsum += w
        for key, vals in make_solver().EvaluateObservables(sitps, comp).items():
            acc[key] = acc.get(key, 0.0) + w * np.asarray(vals).ravel()
    if size > 1:
        return acc, wsum
    if not wsum > 0.0:
        raise RuntimeError("ExactSumMeasurerMPI: total weight must be positive")
    return {k: v / wsum for k, v in acc.items()}
