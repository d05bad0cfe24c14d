"""Fermionic (fZ2-graded) PEPS through the bosonic boundary-MPS machinery -- TEST INFRASTRUCTURE ONLY.

Result used by the product (peps_amd/fermion.py, host layer) and proven here against the graded
algebra of oracle/graded.py (tests/test_oracle_fermion.py):

  With the parity legs ordered row-major, the graded contraction of the projected network equals an
  ORDINARY contraction of sign-decorated site tensors times a sign that depends on the particle number only:

      <S|Psi>_row = sigma(N_f) * Contract_dense( T_v[s_v] * (-1)^{u * (number of fermions at sites <= v, row-major)} )
      sigma(N_f) = (-1)^{N_f + N_f (N_f - 1) / 2}

  (u = parity of the U-leg index).  Derivation: move every parity leg to the front (cost: sigma), order the
  legs of each tensor (L, U, D, R) so that horizontal bonds are adjacent [OUT][IN] pairs, contract rows
  top-down; the evenness of every tensor turns the Koszul signs of the vertical bonds into
  (-1)^{u (1 + l + J_v)} with J_v the Jordan-Wigner prefix parity, and the leg reordering (-1)^{u (d + r)}
  combines with it to (-1)^{u (J_v + n_v)}.  For the parity legs in COLUMN-major order the same argument
  (roles of rows and columns exchanged) gives the decoration
      (-1)^{u n + u + d r + l + l u + l J_v},   J_v = number of fermions before v in column-major order.

  A nearest-neighbour hop along a row (column) joins two modes that are adjacent in the row-major
  (column-major) order, so its Jordan-Wigner sign is +1 and it changes the decoration of its two sites only:
  every BMPS environment stays valid, and the reference's flow (horizontal bonds in the row pass, vertical
  bonds in the column pass, psi and psi' along the same path: docs/dev/design/math/
  fermion-sign-in-bmps-contraction.md) carries over with one ordinary (bosonic) contraction per pass.

Physical states: 0 = occupied (odd), 1 = empty (even) (square_spinless_fermion.h:35-37).
Extended state of a site = s + d * variant, variant 0/1 = row-major decoration with even/odd inclusive prefix,
2/3 = column-major decoration with J_v = 0/1.
"""
import numpy as np

from . import vmc
from .graded import GT, load_qlten_z2

ROW, COL = 0, 1
NVAR = 4


def load_fermion_sitps(directory, complex_data=False):
    import os
    with open(os.path.join(directory, "tps_meta.txt")) as f:
        toks = f.read().split()
    rows, cols, d = int(toks[0]), int(toks[1]), int(toks[2])
    return [[[load_qlten_z2(os.path.join(directory, "tps_ten%d_%d_%d.qlten" % (r, c, s)), complex_data)
              for s in range(d)] for c in range(cols)] for r in range(rows)]


class FermionSITPS:
    """dense, sign-decorated view of a graded SplitIndexTPS"""

    def __init__(self, gts):
        self.rows, self.cols, self.d = len(gts), len(gts[0]), len(gts[0][0])
        self.gts = gts
        self.nf = [int(gts[0][0][s].par[4][0]) for s in range(self.d)]     # fermion parity of each physical state
        self.par = [[tuple(gts[r][c][0].par[k] for k in range(4)) for c in range(self.cols)] for r in range(self.rows)]
        self.ext = [[self._decorate(r, c) for c in range(self.cols)] for r in range(self.rows)]

    def _decorate(self, r, c):
        pl, pd, pr, pu = self.par[r][c]
        l = pl[:, None, None, None]; dd = pd[None, :, None, None]; rr = pr[None, None, :, None]; u = pu[None, None, None, :]
        out = [None] * (NVAR * self.d)
        for s in range(self.d):
            t = self.gts[r][c][s]
            assert t.is_even() and tuple(t.dirs) == (-1, 1, 1, -1, -1)
            for k in range(4):
                assert np.array_equal(t.par[k], self.par[r][c][k])
            a = t.arr[..., 0]
            n = self.nf[s]
            out[s] = a
            out[s + self.d] = a * (1 - 2 * (u % 2))
            base = (u * n + u + dd * rr + l + l * u) % 2
            out[s + 2 * self.d] = a * (1 - 2 * base)
            out[s + 3 * self.d] = a * (1 - 2 * ((base + l) % 2))
        return out

    def ext_config(self, cfg, order):
        cfg = np.asarray(cfg)
        occ = np.array(self.nf)[cfg]                       # fermion number parity per site
        ext = np.zeros_like(cfg)
        if order == ROW:
            incl = np.cumsum(occ.ravel()).reshape(cfg.shape) % 2
            ext = cfg + self.d * incl
        else:
            flat = occ.T.ravel()                             # column-major
            before = (np.cumsum(flat) - flat) % 2
            ext = cfg + self.d * (2 + before.reshape(cfg.shape[::-1]).T)
        return ext

    def sigma(self, cfg):
        nf = int(np.sum(np.array(self.nf)[np.asarray(cfg)]))
        return (-1) ** (nf + nf * (nf - 1) // 2)

    def kappa(self, cfg):
        """sign of reordering the occupied modes from row-major to column-major order"""
        occ = np.array(self.nf)[np.asarray(cfg)]
        pos = [(c, r) for r in range(self.rows) for c in range(self.cols) if occ[r, c] % 2]   # row-major list, col-major keys
        inv = sum(1 for i in range(len(pos)) for j in range(i + 1, len(pos)) if pos[i] > pos[j])
        return (-1) ** inv

    def component(self, cfg, order, trun_para):
        return vmc.TPSWaveFunctionComponent(self.ext, self.ext_config(cfg, order), trun_para)

    def amplitude(self, cfg, trun_para, order=ROW):
        """<S|Psi> with the parity legs in row-major (ROW) or column-major (COL) order"""
        return self.sigma(cfg) * self.component(cfg, order, trun_para).amplitude


class SquareSpinlessFermionOBC:
    """square_spinless_fermion.h:51-200: H = -t sum_<ij> (c+_i c_j + h.c.) - t2 sum_<<ij>> (...) + V sum_<ij> n_i n_j.
    NN ratios come from ReplaceNNSiteTrace along the pass that keeps the hop local; the NNN hop is not local in
    the decorated form and is evaluated from a fresh amplitude (only the 2x2 known answers need t2 != 0)."""

    def __init__(self, t, t2=0.0, V=0.0):
        self.t, self.t2, self.V = t, t2, V

    def bond(self, c1, c2, n1, n2):
        """(diagonal energy, coefficient of conj(psi'/psi) for the exchanged configuration) of one NN bond"""
        return self.V * n1 * n2, (-self.t if c1 != c2 else 0.0)

    def onsite(self, cfg):
        return 0.0

    def CalEnergy(self, fs, cfg, trun_para, bonds=None):
        """bonds: optional dict filled with the per-bond energies {"h": (rows, cols-1), "v": (rows-1, cols),
        "dr"/"ur": (rows-1, cols-1)} that EvaluateObservables reports"""
        from .bmps import LEFT, DOWN, RIGHT, UP, HORIZONTAL, VERTICAL
        cfg = np.asarray(cfg)
        rows, cols, d = fs.rows, fs.cols, fs.d
        occ = np.array(fs.nf)[cfg]
        e = 0.0
        psis = []
        if bonds is not None:
            dt = np.result_type(fs.ext[0][0][0].dtype, np.float64)      # QLTEN_Complex states: complex local estimators
            bonds.update(h=np.zeros((rows, cols - 1), dt), v=np.zeros((rows - 1, cols), dt), dr=np.zeros((rows - 1, cols - 1), dt),
                         ur=np.zeros((rows - 1, cols - 1), dt))
        # ---- row pass: horizontal bonds (square_nnn_energy_solver.h:116-201)
        comp = fs.component(cfg, ROW, trun_para)
        tn, c = comp.tn, comp.contractor
        ext = comp.config
        c.GenerateBMPSApproach(tn, UP)
        for row in range(rows):
            c.InitBTen(tn, LEFT, row)
            c.GrowFullBTen(tn, RIGHT, row, 1, True)
            psis.append(c.Trace(tn, (row, 0), HORIZONTAL))
            for col in range(cols - 1):
                s1, s2 = (row, col), (row, col + 1)
                diag, off = self.bond(int(cfg[s1]), int(cfg[s2]), occ[s1], occ[s2])
                eb = diag
                if off != 0.0:
                    psi = c.Trace(tn, s1, HORIZONTAL)
                    new = cfg.copy(); new[s1], new[s2] = cfg[s2], cfg[s1]
                    ne = fs.ext_config(new, ROW)
                    psi_ex = c.ReplaceNNSiteTrace(tn, s1, s2, HORIZONTAL, fs.ext[row][col][ne[s1]], fs.ext[row][col + 1][ne[s2]])
                    eb = eb + off * np.conj(psi_ex / psi)
                e += eb
                if bonds is not None:
                    bonds["h"][row, col] = eb
                c.ShiftBTenWindow(tn, RIGHT)
            if row < rows - 1:
                c.ShiftBMPSWindow(tn, DOWN)
        # ---- column pass: vertical bonds (bond_traversal_mixin.h:113-144)
        comp = fs.component(cfg, COL, trun_para)
        tn, c = comp.tn, comp.contractor
        c.GenerateBMPSApproach(tn, LEFT)
        for col in range(cols):
            c.InitBTen(tn, UP, col)
            c.GrowFullBTen(tn, DOWN, col, 2, True)
            psis.append(c.Trace(tn, (0, col), VERTICAL))
            for row in range(rows - 1):
                s1, s2 = (row, col), (row + 1, col)
                diag, off = self.bond(int(cfg[s1]), int(cfg[s2]), occ[s1], occ[s2])
                eb = diag
                if off != 0.0:
                    psi = c.Trace(tn, s1, VERTICAL)
                    new = cfg.copy(); new[s1], new[s2] = cfg[s2], cfg[s1]
                    ne = fs.ext_config(new, COL)
                    psi_ex = c.ReplaceNNSiteTrace(tn, s1, s2, VERTICAL, fs.ext[row][col][ne[s1]], fs.ext[row + 1][col][ne[s2]])
                    eb = eb + off * np.conj(psi_ex / psi)
                e += eb
                if bonds is not None:
                    bonds["v"][row, col] = eb
                if row < rows - 2:
                    c.ShiftBTenWindow(tn, DOWN)
            if col < cols - 1:
                c.ShiftBMPSWindow(tn, RIGHT)
        # ---- NNN hops (t2): Jordan-Wigner string in row-major order, fresh amplitudes
        if self.t2 != 0.0:
            psi0 = fs.amplitude(cfg, trun_para)
            flat = occ.ravel()
            for row in range(rows - 1):
                for col in range(cols - 1):
                    for (a, b) in (((row, col), (row + 1, col + 1)), ((row + 1, col), (row, col + 1))):
                        if cfg[a] == cfg[b]:
                            continue
                        ia, ib = sorted((a[0] * cols + a[1], b[0] * cols + b[1]))
                        jw = (-1) ** int(np.sum(flat[ia + 1:ib]))
                        new = cfg.copy(); new[a], new[b] = cfg[b], cfg[a]
                        eb = -self.t2 * jw * np.conj(fs.amplitude(new, trun_para) / psi0)
                        e += eb
                        if bonds is not None:
                            bonds["dr" if a == (row, col) else "ur"][row, col] = eb
        return e + self.onsite(cfg), psis

    def EvaluateObservables(self, fs, cfg, trun_para):
        """registry of SquareNNNModelMeasurementSolver<SquareSpinlessFermion> (square_nnn_model_measurement_solver.h:33-210,
        square_spinless_fermion.h:87-115): energy, charge (1 - config: state 0 is occupied), per-bond energies"""
        bonds = {}
        e, _ = self.CalEnergy(fs, cfg, trun_para, bonds)
        return {"energy": np.array([e]), "charge": (1.0 - np.asarray(cfg)).ravel(), "bond_energy_h": bonds["h"].ravel(),
                "bond_energy_v": bonds["v"].ravel(), "bond_energy_dr": bonds["dr"].ravel(), "bond_energy_ur": bonds["ur"].ravel()}


class SquaretJVModelOBC(SquareSpinlessFermionOBC):
    """square_tJ_model.h:301-345 (SquaretJModelMixIn::EvaluateBondEnergy) + :215-228; states 0 up, 1 down, 2 empty
    (vmc_basic/tj_single_site_state.h:19-23); H = -t sum (c+ c + h.c.) + J sum (S.S - n n / 4) + V sum n n - mu N"""

    def __init__(self, t, t2, J, V, mu):
        super().__init__(t, t2, V)
        self.J, self.mu = J, mu

    def bond(self, c1, c2, n1, n2):
        if c1 == c2:
            return (0.0 if c1 == 2 else self.V), 0.0
        if c1 == 2 or c2 == 2:
            return 0.0, -self.t
        return -0.5 * self.J + self.V, 0.5 * self.J

    def onsite(self, cfg):
        return -self.mu * float(np.sum(np.asarray(cfg) != 2))

    def EvaluateObservables(self, fs, cfg, trun_para):
        """registry of SquareNNNModelMeasurementSolver<SquaretJVModel> (square_nnn_model_measurement_solver.h:33-210 with the t-J hooks
        square_tJ_model.h:215-228): energy, spin_z (+1/2 up, -1/2 down, 0 empty), charge (1 on an occupied site), per-bond energies"""
        bonds = {}
        e, _ = self.CalEnergy(fs, cfg, trun_para, bonds)
        c = np.asarray(cfg).ravel()
        return {"energy": np.array([e]), "spin_z": np.where(c == 0, 0.5, np.where(c == 1, -0.5, 0.0)), "charge": (c != 2).astype(np.float64),
                "bond_energy_h": bonds["h"].ravel(), "bond_energy_v": bonds["v"].ravel(), "bond_energy_dr": bonds["dr"].ravel(),
                "bond_energy_ur": bonds["ur"].ravel()}


class MCUpdateSquareNNExchangeOBC:
    """MCUpdateSquareNNExchangeOBC (vmc_basic/configuration_update_strategies/square_nn_updater.h:25-83, :142-189) on a fermionic state:
    the row pass runs on the row-major decorated network, the column pass on the column-major one (each bond exchange is then local:
    module docstring), both consume ONE std::mt19937 stream in the reference's order -- a deviate only when |psi_b| < |psi_a|.
    `amplitude` is |psi_a| as the reference carries it: the value of the constructor's EvaluateAmplitude, then every accepted psi_b
    (the sign convention of the two decorations differs, the Metropolis rule only sees magnitudes)."""

    def __init__(self, seed=0):
        self.rng = vmc.StdMT19937(seed)

    def _pass(self, fs, cfg, amp_abs, order, trun_para):
        from .bmps import LEFT, DOWN, RIGHT, UP, HORIZONTAL, VERTICAL
        comp = fs.component(cfg, order, trun_para)
        tn, c = comp.tn, comp.contractor
        rows, cols = fs.rows, fs.cols
        accepted = 0

        def update(s1, s2, bond_dir):                                           # TwoSiteNNUpdateLocalImpl (:146-188)
            nonlocal amp_abs, accepted
            c1, c2 = int(cfg[s1]), int(cfg[s2])
            if c1 == c2:
                return
            new = cfg.copy()
            new[s1], new[s2] = c2, c1
            ne = fs.ext_config(new, order)                                      # only the two sites of the bond change their decoration
            psi_b = c.ReplaceNNSiteTrace(tn, s1, s2, bond_dir, fs.ext[s1[0]][s1[1]][ne[s1]], fs.ext[s2[0]][s2[1]][ne[s2]])
            if abs(psi_b) < amp_abs:
                div = abs(psi_b) / amp_abs
                if not (self.rng.u_double() < div * div):
                    return
            cfg[s1], cfg[s2] = c2, c1
            comp.UpdateLocal(fs.ext, psi_b, (s1, int(ne[s1])), (s2, int(ne[s2])))
            amp_abs = abs(psi_b)
            accepted += 1

        if order == ROW:
            c.GenerateBMPSApproach(tn, UP)
            for row in range(rows):
                c.InitBTen(tn, LEFT, row)
                c.GrowFullBTen(tn, RIGHT, row, 2, True)
                for col in range(cols - 1):
                    update((row, col), (row, col + 1), HORIZONTAL)
                    if col < cols - 2:
                        c.ShiftBTenWindow(tn, RIGHT)
                if row < rows - 1:
                    c.ShiftBMPSWindow(tn, DOWN)
        else:
            c.GenerateBMPSApproach(tn, LEFT)
            for col in range(cols):
                c.InitBTen(tn, UP, col)
                c.GrowFullBTen(tn, DOWN, col, 2, True)
                for row in range(rows - 1):
                    update((row, col), (row + 1, col), VERTICAL)
                    if row < rows - 2:
                        c.ShiftBTenWindow(tn, DOWN)
                if col < cols - 1:
                    c.ShiftBMPSWindow(tn, RIGHT)
        return accepted, amp_abs

    def __call__(self, fs, cfg, amp_abs, trun_para):
        """one sweep; `cfg` (physical states) is updated in place.  Returns (accept rate, |psi| of the configuration reached)."""
        a1, amp_abs = self._pass(fs, cfg, amp_abs, ROW, trun_para)
        a2, amp_abs = self._pass(fs, cfg, amp_abs, COL, trun_para)
        return (a1 + a2) / float(fs.cols * (fs.rows - 1) + fs.rows * (fs.cols - 1)), amp_abs


def exact_sum_energy(fs, all_configs, trun_para, model):
    """ExactSumEnergyEvaluatorMPI (exact_summation_energy_evaluator.h:173-302), energy only"""
    wsum = wesum = 0.0
    for cfg in all_configs:
        amp = fs.amplitude(cfg, trun_para)
        if amp == 0:
            continue
        e, _ = model.CalEnergy(fs, cfg, trun_para)
        w = abs(amp) ** 2
        wsum += w
        wesum += w * e
    return wesum / wsum


def exact_sum_measure(fs, all_configs, trun_para, model, rank=0, size=1):
    """ExactSumMeasurerMPI (exact_summation_measurer.h:103-257) for a fermionic state: <O> = sum_S |psi(S)|^2 O_loc(S) /
    sum_S |psi(S)|^2 over configurations rank, rank + size, ...; returns (weighted sums by key, weight sum) for size > 1
    and the normalised registry for size == 1."""
    wsum, acc = 0.0, {}
    for i in range(rank, len(all_configs), size):
        cfg = all_configs[i]
        w = abs(fs.amplitude(cfg, trun_para)) ** 2
        wsum += w
        for key, vals in model.EvaluateObservables(fs, cfg, trun_para).items():
            acc[key] = acc.get(key, 0.0) + w * np.asarray(vals)
    if size > 1:
        return acc, wsum
    if not wsum > 0.0:
        raise RuntimeError("ExactSumMeasurerMPI: total weight must be positive")
    return {k: v / wsum for k, v in acc.items()}
