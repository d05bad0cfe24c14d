"""Fermionic (fZ2-graded) SplitIndexTPS on the bosonic device path.

The graded contraction of a projected fermionic PEPS equals an ordinary contraction of sign-decorated site
tensors times a sign that depends on the particle number only (statement and proof obligations:
oracle/fermion.py, tests/test_oracle_fermion.py -- this module restates the construction for the product
path and never imports the oracle):

    <S|Psi>_row = sigma(N_f) * Contract( T_v[s_v] * (-1)^{u * (fermions at sites <= v, row-major)} )
    column-major order:       decoration (-1)^{u n + u + d r + l + l u + l J_v}

so a fermionic state is uploaded as a SplitIndexTPS with 4*d "extended" components per site
(extended state = s + d * variant; variants 0/1 row-major even/odd, 2/3 column-major J = 0/1) and every
fermionic sign becomes a choice of component, i.e. an entry of the configuration table the device already
indexes the shared SITPS with.  Nearest-neighbour hops along a row (column) are adjacent in the row-major
(column-major) mode order: Jordan-Wigner sign +1, only the two sites' components change, all BMPS / BTen
environments stay valid -- the reference's flow (horizontal bonds in the row pass, vertical bonds in the
column pass, psi and psi' along the same path; square_nnn_energy_solver.h:116-201,
bond_traversal_mixin.h:113-144, square_spinless_fermion.h:134-159) runs unchanged.

Physical states: 0 = occupied (odd), 1 = empty (even) (square_spinless_fermion.h:35-37).
"""
import os

import numpy as np

ROW, COL = 0, 1
NVAR = 4


def _read_qlten_z2(path, complex_data=False):
    """one fZ2 .qlten file -> (dense array, [parity vector per leg], [direction per leg]); format: SURVEY 8c.
    complex_data: QLTEN_Complex payload (interleaved complex128)"""
    with open(path, "rb") as f:
        buf = f.read()
    pos = 0

    def tok():
        nonlocal pos
        e = buf.index(b"\n", pos)
        t = buf[pos:e]
        pos = e + 1
        return t

    rank = int(tok())
    legs = []
    for _ in range(rank):
        nsec = int(tok())
        secs = []
        for _s in range(nsec):
            qn = int(tok()); tok(); deg = int(tok()); tok()
            secs.append((qn, deg))
        d = int(tok()); dim = int(tok()); tok()
        if sum(x[1] for x in secs) != dim:
            raise ValueError("corrupt index in %s" % path)
        legs.append((secs, d))
    nblocks = int(tok())
    blocks = [[int(tok()) for _ in range(rank)] for _ in range(nblocks)]
    shape = tuple(sum(s[1] for s in secs) for secs, _ in legs)
    out = np.zeros(shape, dtype=np.complex128 if complex_data else np.float64)
    item = 16 if complex_data else 8
    offs = [np.concatenate([[0], np.cumsum([s[1] for s in secs])]) for secs, _ in legs]
    for c in blocks:
        bshape = tuple(legs[k][0][c[k]][1] for k in range(rank))
        n = int(np.prod(bshape))
        data = np.frombuffer(buf, dtype="<c16" if complex_data else "<f8", count=n, offset=pos)
        pos += n * item
        out[tuple(slice(offs[k][c[k]], offs[k][c[k] + 1]) for k in range(rank))] = data.reshape(bshape)
    par = [np.concatenate([np.full(deg, qn % 2, dtype=np.int64) for qn, deg in secs]) for secs, _ in legs]
    return out, par, [d for _, d in legs]


class FermionState:
    """tensors[r][c][s] = dense array (L, D, R, U); par[r][c] = 4 parity vectors; nf[s] = fermion parity of state s"""

    def __init__(self, tensors, par, nf):
        self.tensors, self.par, self.nf = tensors, par, np.asarray(nf, dtype=np.int64)
        self.rows, self.cols, self.d = len(tensors), len(tensors[0]), len(tensors[0][0])
        for r in range(self.rows):
            for c in range(self.cols):
                pl, pd, pr, pu = self.par[r][c]
                tot = pl[:, None, None, None] + pd[None, :, None, None] + pr[None, None, :, None] + pu[None, None, None, :]
                for s in range(self.d):
                    if np.any((np.abs(self.tensors[r][c][s]) > 0) & ((tot + self.nf[s]) % 2 == 1)):
                        raise ValueError("site tensor (%d, %d, state %d) is not parity even" % (r, c, s))

    @staticmethod
    def load(directory, complex_data=False):
        """SplitIndexTPS<.., fZ2QN>::Load layout: tps_meta.txt + tps_ten{r}_{c}_{s}.qlten, rank-5 tensors (L, D, R, U, parity);
        complex_data: a SplitIndexTPS<QLTEN_Complex, fZ2QN> dump"""
        with open(os.path.join(directory, "tps_meta.txt")) as f:
            toks = f.read().split()
        rows, cols, d = int(toks[0]), int(toks[1]), int(toks[2])
        tensors, par, nf = [[None] * cols for _ in range(rows)], [[None] * cols for _ in range(rows)], [None] * d
        for r in range(rows):
            for c in range(cols):
                comp = []
                for s in range(d):
                    a, p, dirs = _read_qlten_z2(os.path.join(directory, "tps_ten%d_%d_%d.qlten" % (r, c, s)), complex_data)
                    if a.ndim != 5 or tuple(dirs) != (-1, 1, 1, -1, -1) or a.shape[4] != 1:
                        raise ValueError("expected rank-5 fermionic site tensors (L, D, R, U, parity)")
                    if par[r][c] is None:
                        par[r][c] = tuple(p[:4])
                    elif any(not np.array_equal(par[r][c][k], p[k]) for k in range(4)):
                        raise ValueError("components of one site disagree on the parity structure of a bond")
                    if nf[s] is None:
                        nf[s] = int(p[4][0])
                    comp.append(a[..., 0])
                tensors[r][c] = comp
        return FermionState(tensors, par, nf)

    @property
    def D(self):
        return max(max(t[0].shape) for row in self.tensors for t in row)

    @property
    def is_complex(self):
        return np.iscomplexobj(self.tensors[0][0][0])

    def extended_flat(self, D=None, dtype=None):
        """upload buffer [row][col][4 d][D][D][D][D] of the decorated components (legs zero padded to D); element type of the state"""
        D = D or self.D
        dtype = dtype or (np.complex128 if self.is_complex else np.float64)
        out = np.zeros((self.rows, self.cols, NVAR * self.d, D, D, D, D), dtype=dtype)
        for r in range(self.rows):
            for c in range(self.cols):
                pl, pd, pr, pu = self.par[r][c]
                l = pl[:, None, None, None]; dd = pd[None, :, None, None]; rr = pr[None, None, :, None]; u = pu[None, None, None, :]
                for s in range(self.d):
                    a = self.tensors[r][c][s]
                    n = int(self.nf[s])
                    base = (u * n + u + dd * rr + l + l * u) % 2
                    sl = (r, c, slice(None)) + tuple(slice(0, k) for k in a.shape)
                    for var, sign in enumerate((np.ones_like(a), 1 - 2 * (u % 2) + 0 * a, 1 - 2 * base + 0 * a,
                                                1 - 2 * ((base + l) % 2) + 0 * a)):
                        out[(r, c, s + self.d * var) + tuple(slice(0, k) for k in a.shape)] = a * sign
        return out

    def ext_config(self, configs, order):
        """configs [..., rows, cols] of physical states -> extended states for the given mode order"""
        cfg = np.asarray(configs)
        occ = self.nf[cfg] % 2
        shp = cfg.shape
        if order == ROW:
            flat = occ.reshape(shp[:-2] + (-1,))
            incl = np.cumsum(flat, axis=-1) % 2
            return (cfg + self.d * incl.reshape(shp)).astype(np.int32)
        flat = np.swapaxes(occ, -1, -2).reshape(shp[:-2] + (-1,))
        before = (np.cumsum(flat, axis=-1) - flat) % 2
        before = np.swapaxes(before.reshape(shp[:-2] + (shp[-1], shp[-2])), -1, -2)
        return (cfg + self.d * (2 + before)).astype(np.int32)

    def sigma(self, configs):
        nf = np.sum(self.nf[np.asarray(configs)] % 2, axis=(-1, -2))
        return 1 - 2 * ((nf + nf * (nf - 1) // 2) % 2)

    def kappa(self, configs):
        """sign of reordering the occupied modes from row-major to column-major order (per configuration)"""
        cfg = np.asarray(configs)
        occ = self.nf[cfg] % 2
        lead = cfg.shape[:-2]
        out = np.ones(lead, dtype=np.int64)
        for ix in np.ndindex(*lead):
            o = occ[ix]
            keys = [(c, r) for r in range(self.rows) for c in range(self.cols) if o[r, c]]
            inv = sum(1 for i in range(len(keys)) for j in range(i + 1, len(keys)) if keys[i] > keys[j])
            out[ix] = 1 - 2 * (inv % 2)
        return out


def fold_gradient(state, grad_ext):
    """gradient with respect to the extended (decorated) components [rows][cols][4 d][D^4] -> gradient with respect
    to the stored components [rows][cols][d][D^4]: T''[s + d var] = sign_var * T[s], so dE/dT[s] = sum_var sign_var *
    dE/dT''[s + d var]; parity-forbidden entries (not parameters of a Z2-symmetric state) are zeroed."""
    rows, cols, d = state.rows, state.cols, state.d
    D = grad_ext.shape[3]
    out = np.zeros((rows, cols, d, D, D, D, D), dtype=grad_ext.dtype)
    for r in range(rows):
        for c in range(cols):
            pl, pd, pr, pu = state.par[r][c]
            l = pl[:, None, None, None]; dd = pd[None, :, None, None]; rr = pr[None, None, :, None]; u = pu[None, None, None, :]
            shp = (len(pl), len(pd), len(pr), len(pu))
            sl = tuple(slice(0, k) for k in shp)
            for s in range(d):
                n = int(state.nf[s])
                base = (u * n + u + dd * rr + l + l * u) % 2
                allowed = ((l + dd + rr + u + n) % 2 == 0)
                signs = (np.ones(shp), 1 - 2 * (u % 2) + np.zeros(shp), 1 - 2 * base + np.zeros(shp), 1 - 2 * ((base + l) % 2) + np.zeros(shp))
                acc = np.zeros(shp, dtype=grad_ext.dtype)
                for var in range(NVAR):
                    acc += signs[var] * grad_ext[(r, c, s + d * var) + sl]
                out[(r, c, s) + sl] = acc * allowed
    return out


def evaluate_amplitude(ctx, state, configs):
    """<S|Psi> (parity legs in row-major order) for a batch of physical configurations; ctx must hold
    state.extended_flat() (phys_dim = 4 d)."""
    ctx.set_configs(state.ext_config(configs, ROW))
    return state.sigma(configs) * ctx.evaluate_amplitude()


def tj_energy(ctx, state, configs, t, J, V=0.0, mu=0.0):
    """E_loc(S) of the t-J-V model (square_tJ_model.h:301-345 + :215-228; states 0 up, 1 down, 2 empty):
    H = -t sum (c+ c + h.c.) + J sum (S.S - n n / 4) + V sum n n - mu N, nearest neighbours only (t2 = 0)."""
    def bond(c1, c2):
        same = c1 == c2
        hole = (c1 == 2) | (c2 == 2)
        diag = np.where(same, np.where(c1 == 2, 0.0, V), np.where(hole, 0.0, -0.5 * J + V))
        off = np.where(same, 0.0, np.where(hole, -t, 0.5 * J))
        return diag, off
    e, psis = nn_energy(ctx, state, configs, bond)
    return e - mu * np.sum(np.asarray(configs) != 2, axis=(1, 2)), psis


def spinless_fermion_energy(ctx, state, configs, t, V=0.0, t2=0.0, bonds=None, nnn="local"):
    """E_loc(S) of H = -t sum_<ij> (c+_i c_j + h.c.) - t2 sum_<<ij>> (c+_i c_j + h.c.) + V sum_<ij> n_i n_j
    (square_spinless_fermion.h:134-200).  Returns (energy [n], psi_list [rows + cols][n]).  The NN hops are local
    replacements inside the row / column pass.  The diagonal hop (t2): nnn = "local" (round 5) reuses the environments of the
    row pass -- a plaquette replacement against parity-twisted BTen2 environments (nnn_hop_energy_local); nnn = "fresh" takes
    every hopped amplitude from a fresh contraction (nnn_hop_energy, rounds 2-4; kept as the independent check)."""
    nf = state.nf
    def bond(c1, c2):
        return V * (nf[c1] % 2) * (nf[c2] % 2), np.where(c1 != c2, -t, 0.0)
    e, psis = nn_energy(ctx, state, configs, bond, bonds)
    if t2 != 0.0:
        e = e + (nnn_hop_energy_local if nnn == "local" else nnn_hop_energy)(ctx, state, configs, t2, bonds)
    return e, psis


def nnn_hop_energy_local(ctx, state, configs, t2, bonds=None):
    """The diagonal hops of all plaquettes with the environments of ONE row pass (the reference's flow,
    square_nnn_energy_solver.h:203-265: BTen2 environments of the row pair, ReplaceNNNSiteTrace per diagonal).

    The reference's graded ReplaceNNNSiteTrace carries the signs in the tensor algebra.  In the sign-decorated form a hop between
    a = (r, c) / (r+1, c) and b = (r+1, c+1) / (r, c+1) flips the variant (parity of the fermion count up to and including the
    site, row-major) of every site between the two ends: row r right of the plaquette and row r+1 left of it.  A variant flip of
    a whole row is a second configuration table, so the hopped amplitude is a local replacement of the four plaquette tensors
    against TWISTED environments: the LEFT BTen2 grown with row r+1 flipped, the RIGHT BTen2 grown with row r flipped
    (pepsgpu_cfg_override_slice + the second BTen2 set, pepsgpu_replace_plaquette_trace).  psi of the same plaquette comes from
    the untwisted set along the same path.  Cost per row pair: two more BTen2 chains instead of 2 (cols - 1) fresh contractions."""
    from .capi import LEFT, RIGHT, UP, DOWN, HORIZONTAL
    cfg = np.asarray(configs)
    n, rows, cols = cfg.shape
    d = state.d
    occ = (np.asarray(state.nf)[cfg] % 2).reshape(n, -1)
    ext = state.ext_config(cfg, ROW)
    flip = np.where(ext // d == 0, ext + d, ext - d).astype(np.int32)       # variant 0 <-> 1 (row-major variants)
    dt = np.complex128 if state.is_complex else np.float64
    e = np.zeros(n, dt)
    if bonds is not None:
        bonds.update(dr=np.zeros((n, rows - 1, cols - 1), dt), ur=np.zeros((n, rows - 1, cols - 1), dt))
    if cols < 2 or rows < 2:
        return e
    ctx.set_configs(ext)
    ctx.generate_bmps_approach(UP)                      # DOWN stack complete, UP at the vacuum
    try:
        for row in range(rows - 1):
            # untwisted set 0 and twisted set 1: the whole RIGHT stack at once (its levels are read by index), LEFT step by step
            ctx.bten2_select_set(0)
            ctx.grow_full_bten2(RIGHT, row, 2, True)
            ctx.init_bten2(LEFT, row)
            ctx.bten2_select_set(1)
            ctx.cfg_override_slice(HORIZONTAL, row, flip[:, row, :])
            ctx.grow_full_bten2(RIGHT, row, 2, True)
            ctx.cfg_override_slice(HORIZONTAL, row + 1, flip[:, row + 1, :])
            ctx.init_bten2(LEFT, row)
            for col in range(cols - 1):
                # (the plaquette's own states are handed over explicitly: a site that reads a configuration table would read the
                # override that is set for the twisted chain)
                own = np.stack([ext[:, row, col], ext[:, row + 1, col], ext[:, row + 1, col + 1], ext[:, row, col + 1]], axis=-1)
                psi = ctx.replace_plaquette_trace(row, col, own[:, None, :], 0, 0)[:, 0]
                cands, terms = [], []
                for key, a, b in (("dr", (row, col), (row + 1, col + 1)), ("ur", (row + 1, col), (row, col + 1))):
                    differ = cfg[:, a[0], a[1]] != cfg[:, b[0], b[1]]
                    ia, ib = sorted((a[0] * cols + a[1], b[0] * cols + b[1]))
                    jw = (-1.0) ** occ[:, ia + 1:ib].sum(axis=1)
                    new = cfg.copy()
                    new[:, a[0], a[1]], new[:, b[0], b[1]] = cfg[:, b[0], b[1]], cfg[:, a[0], a[1]]
                    ne = state.ext_config(new, ROW)
                    cands.append(np.stack([ne[:, row, col], ne[:, row + 1, col], ne[:, row + 1, col + 1], ne[:, row, col + 1]], axis=-1))
                    terms.append((key, differ, jw))
                if any(t[1].any() for t in terms):
                    psi_ex = ctx.replace_plaquette_trace(row, col, np.stack(cands, axis=1), 1, 1)
                    for k, (key, differ, jw) in enumerate(terms):
                        eb = np.where(differ, -t2 * jw * np.conj(psi_ex[:, k] / np.where(psi == 0, 1.0, psi)), 0.0)
                        e += eb
                        if bonds is not None:
                            bonds[key][:, row, col] = eb
                if col < cols - 2:          # both LEFT chains advance over column col (set 1 under the row+1 override)
                    ctx.grow_bten2_step(LEFT, row)
                    ctx.bten2_select_set(0)
                    ctx.cfg_override_slice(HORIZONTAL, row + 1, None)
                    ctx.grow_bten2_step(LEFT, row)
                    ctx.bten2_select_set(1)
                    ctx.cfg_override_slice(HORIZONTAL, row + 1, flip[:, row + 1, :])
            ctx.cfg_override_slice(HORIZONTAL, row + 1, None)
            ctx.bten2_select_set(0)
            if row < rows - 2:
                ctx.shift_bmps_window(DOWN)
    finally:
        ctx.cfg_override_slice(HORIZONTAL, 0, None)
        ctx.bten2_select_set(0)
    return e


def nnn_hop_energy(ctx, state, configs, t2, bonds=None):
    """sum over plaquette diagonals of -t2 * jw * psi(S with the two sites exchanged) / psi(S), jw = Jordan-Wigner string
    of the sites strictly between the two in row-major order (square_spinless_fermion.h:161-200): one batched fresh
    contraction per diagonal.  bonds["dr"] / ["ur"] [n, rows-1, cols-1] receive the per-bond terms."""
    cfg = np.asarray(configs)
    n, rows, cols = cfg.shape
    occ = (np.asarray(state.nf)[cfg] % 2).reshape(n, -1)
    psi0 = evaluate_amplitude(ctx, state, cfg)
    dt = np.complex128 if state.is_complex else np.float64
    e = np.zeros(n, dt)
    if bonds is not None:
        bonds.update(dr=np.zeros((n, rows - 1, cols - 1), dt), ur=np.zeros((n, rows - 1, cols - 1), dt))
    for row in range(rows - 1):
        for col in range(cols - 1):
            for key, a, b in (("dr", (row, col), (row + 1, col + 1)), ("ur", (row + 1, col), (row, col + 1))):
                differ = cfg[:, a[0], a[1]] != cfg[:, b[0], b[1]]
                if not differ.any():
                    continue
                ia, ib = sorted((a[0] * cols + a[1], b[0] * cols + b[1]))
                jw = (-1.0) ** occ[:, ia + 1:ib].sum(axis=1)
                new = cfg.copy()
                new[:, a[0], a[1]], new[:, b[0], b[1]] = cfg[:, b[0], b[1]], cfg[:, a[0], a[1]]
                eb = np.where(differ, -t2 * jw * np.conj(evaluate_amplitude(ctx, state, new) / np.where(psi0 == 0, 1.0, psi0)), 0.0)
                e += eb
                if bonds is not None:
                    bonds[key][:, row, col] = eb
    return e


def spinless_fermion_observables(ctx, state, configs, t, V=0.0, t2=0.0):
    """Registry of SquareNNNModelMeasurementSolver<SquareSpinlessFermion>::EvaluateObservables
    (square_nnn_model_measurement_solver.h:33-210, square_spinless_fermion.h:87-115) for a batch of configurations:
    energy [n, 1], charge [n, rows*cols] (1 - config), bond_energy_h / _v and the (t2 = 0) diagonal bonds.
    Also returns psi_list [rows + cols][n]."""
    cfg = np.asarray(configs)
    n, rows, cols = cfg.shape
    dt = np.complex128 if state.is_complex else np.float64
    bonds = {"dr": np.zeros((n, rows - 1, cols - 1), dt), "ur": np.zeros((n, rows - 1, cols - 1), dt)}
    e, psis = spinless_fermion_energy(ctx, state, cfg, t, V, t2, bonds)
    return {"energy": e[:, None], "charge": (1.0 - cfg).reshape(n, -1), "bond_energy_h": bonds["h"].reshape(n, -1),
            "bond_energy_v": bonds["v"].reshape(n, -1), "bond_energy_dr": bonds["dr"].reshape(n, -1),
            "bond_energy_ur": bonds["ur"].reshape(n, -1)}, psis


def exact_sum_measure(ctx, state, all_configs, t, V=0.0, rank=0, size=1, batch=None, t2=0.0):
    """ExactSumMeasurerMPI (exact_summation_measurer.h:103-257) on the device: configurations rank, rank + size, ... in
    batches of walkers; returns (weighted sums by key, weight sum) -- sum both over ranks (all-reduce) and divide."""
    cfgs = np.asarray(all_configs)[rank::size]
    if len(np.asarray(all_configs)) == 0:
        raise ValueError("ExactSumMeasurerMPI: all_configs must not be empty")
    batch = batch or max(len(cfgs), 1)
    wsum, acc = 0.0, {}
    for b0 in range(0, len(cfgs), batch):
        part = cfgs[b0:b0 + batch]
        obs, psis = spinless_fermion_observables(ctx, state, part, t, V, t2)
        w = np.abs(psis[0]) ** 2                           # |psi(S)|^2: the sign decoration drops out
        wsum += float(w.sum())
        for key, vals in obs.items():
            acc[key] = acc.get(key, 0.0) + w @ vals
    return acc, wsum


def nn_energy(ctx, state, configs, bond_fn, bonds=None):
    """Nearest-neighbour local energy of a fermionic model for every configuration of the batch: the traversal of
    square_nnn_energy_solver.h:116-201 / bond_traversal_mixin.h:113-144 with the fermion interface (psi recomputed
    next to psi').  bond_fn(c1, c2) -> (diagonal energy, coefficient of psi(S with the two states exchanged)/psi(S)),
    arrays over the batch.  Returns (energy [n], psi_list [rows + cols][n]); `bonds` (a dict) receives the per-bond
    energies "h" [n, rows, cols-1] and "v" [n, rows-1, cols] that the measurement registry reports."""
    from .capi import LEFT, DOWN, RIGHT, UP, HORIZONTAL, VERTICAL
    cfg = np.asarray(configs)
    n, rows, cols = cfg.shape
    dt = np.complex128 if state.is_complex else np.float64       # QLTEN_Complex: E_loc = sum H conj(psi' / psi) (the reference's ComplexConjugate)
    e = np.zeros(n, dt)
    psis = []
    if bonds is not None:
        bonds.update(h=np.zeros((n, rows, cols - 1), dt), v=np.zeros((n, rows - 1, cols), dt))

    def bond(s1, s2, orient, order, ext):
        (r1, c1), (r2, c2) = s1, s2
        e_int, off = bond_fn(cfg[:, r1, c1], cfg[:, r2, c2])
        e_int = np.broadcast_to(np.asarray(e_int, dtype=np.float64), (n,))
        off = np.broadcast_to(np.asarray(off, dtype=np.float64), (n,))
        differ = off != 0.0
        if not differ.any():
            return e_int
        psi = ctx.trace(r1, c1, orient)                       # psi along the same path as psi' (sign consistency)
        new = cfg.copy()
        new[:, r1, c1], new[:, r2, c2] = cfg[:, r2, c2], cfg[:, r1, c1]
        ne = state.ext_config(new, order)
        cand = np.stack([ne[:, r1, c1], ne[:, r2, c2]], axis=-1)[:, None, :]
        psi_ex = ctx.replace_nn_trace(r1, c1, orient, cand)[:, 0]
        return e_int + np.where(differ, off * np.conj(psi_ex / np.where(psi == 0, 1.0, psi)), 0.0)

    ext = state.ext_config(cfg, ROW)
    ctx.set_configs(ext)
    ctx.generate_bmps_approach(UP)
    for row in range(rows):
        ctx.init_bten(LEFT, row)
        ctx.grow_full_bten(RIGHT, row, 1, True)
        psis.append(ctx.trace(row, 0, HORIZONTAL))
        for col in range(cols - 1):
            eb = bond((row, col), (row, col + 1), HORIZONTAL, ROW, ext)
            e += eb
            if bonds is not None:
                bonds["h"][:, row, col] = eb
            ctx.shift_bten_window(RIGHT)
        if row < rows - 1:
            ctx.shift_bmps_window(DOWN)
    ext = state.ext_config(cfg, COL)
    ctx.set_configs(ext)
    ctx.generate_bmps_approach(LEFT)
    for col in range(cols):
        ctx.init_bten(UP, col)
        ctx.grow_full_bten(DOWN, col, 2, True)
        psis.append(ctx.trace(0, col, VERTICAL))
        for row in range(rows - 1):
            eb = bond((row, col), (row + 1, col), VERTICAL, COL, ext)
            e += eb
            if bonds is not None:
                bonds["v"][:, row, col] = eb
            if row < rows - 2:
                ctx.shift_bten_window(DOWN)
        if col < cols - 1:
            ctx.shift_bmps_window(RIGHT)
    return e, np.array(psis)


def random_even_state(rows, cols, D, seed, n_odd=None, background=1.0, noise=0.1):
    """synthetic fermionic state for throughput / parity runs (the reference ships no fermionic state larger
    than 2x2): bond space = D states of which n_odd (default D // 2) are odd, site tensors N(0,1) on the
    parity-even entries plus a positive background on the all-even entry; state 0 occupied, 1 empty."""
    rng = np.random.default_rng(seed)
    n_odd = D // 2 if n_odd is None else n_odd
    bond = np.r_[np.zeros(D - n_odd, dtype=np.int64), np.ones(n_odd, dtype=np.int64)]
    one = np.zeros(1, dtype=np.int64)
    tensors, par = [[None] * cols for _ in range(rows)], [[None] * cols for _ in range(rows)]
    for r in range(rows):
        for c in range(cols):
            p = (one if c == 0 else bond, one if r == rows - 1 else bond, one if c == cols - 1 else bond, one if r == 0 else bond)
            par[r][c] = p
            tot = p[0][:, None, None, None] + p[1][None, :, None, None] + p[2][None, None, :, None] + p[3][None, None, None, :]
            comp = []
            for s, n in enumerate((1, 0)):
                # smooth positive background on the parity-allowed entries + noise (as the bosonic generator of
                # SURVEY 8d): keeps the network well conditioned so that chi-truncation is meaningful
                vec = [rng.uniform(0.5, 1.5, size=k) for k in tot.shape]
                bg = vec[0][:, None, None, None] * vec[1][None, :, None, None] * vec[2][None, None, :, None] * vec[3][None, None, None, :]
                a = (background * bg + noise * rng.standard_normal(tot.shape)) * ((tot + n) % 2 == 0)
                comp.append(a / np.sqrt(max(1.0, np.sum(a * a))) * 2.0)
            tensors[r][c] = comp
    return FermionState(tensors, par, [1, 0])
