"""Synthetic workloads of SURVEY.md section 8(d): seeded SITPS tensors and configuration lists.

The reference ships no state larger than 6x6, so throughput runs use this generator
(shapes of BASELINE.json configs C1-C4).  Host-side NumPy only; no device or oracle dependency.
"""
import numpy as np

CONFIGS = {
    # name: (L, D, chi, model)
    "C1": (4, 2, 4, "tfim"),
    "C2": (8, 4, 16, "tfim"),
    "C3": (10, 6, 24, "heisenberg"),
    "C4": (12, 8, 32, "heisenberg"),
}


def bond_dims(L, D, r, c):
    """(D_L, D_D, D_R, D_U) of site (r, c) on an L x L OBC lattice; boundary legs have dim 1."""
    return (1 if c == 0 else D, 1 if r == L - 1 else D, 1 if c == L - 1 else D, 1 if r == 0 else D)


def make_sitps(L, D, d=2, dtype=np.float64, noise=0.1):
    """T[r][c][s] = a (x) b (x) c (x) e + noise * N(0,1) with a,b,c,e ~ U(0.5,1.5), leg order (L,D,R,U),
    rng = default_rng(1000*L + 10*D + s_site), s_site = r*L + c; then a uniform per-site factor so
    that amplitudes are O(1) before the psi(S_ref) normalisation (each virtual bond sums ~D terms of
    magnitude ~1, 2L(L-1) bonds shared by L^2 sites)."""
    sitps = []
    pre = float(D) ** (-2.0 * (L - 1) / L)
    for r in range(L):
        row = []
        for c in range(L):
            rng = np.random.default_rng(1000 * L + 10 * D + r * L + c)
            shp = bond_dims(L, D, r, c)
            comps = []
            for s in range(d):
                vs = [rng.uniform(0.5, 1.5, size=n) for n in shp]
                t = np.einsum("i,j,k,l->ijkl", *vs) + noise * rng.standard_normal(shp)
                comps.append((t * pre).astype(dtype))
            row.append(comps)
        sitps.append(row)
    return sitps


def rescale_sitps(sitps, psi_ref):
    """Divide every site tensor by |psi(S_ref)|^(1/L^2) (mirrors NormalizeStateOrder1,
    include/qlpeps/algorithm/vmc_update/monte_carlo_engine.h:206-240)."""
    L = len(sitps)
    f = abs(psi_ref) ** (-1.0 / (L * L))
    return [[[t * f for t in comps] for comps in row] for row in sitps]


def checkerboard(L):
    return np.array([[(r + c) % 2 for c in range(L)] for r in range(L)], dtype=np.int32)


def make_configs(L, n_walkers, model="heisenberg", seed0=7):
    """Heisenberg: random Sz=0 shuffles, rng(seed0 + w).  TFIM: iid bits."""
    out = np.empty((n_walkers, L, L), dtype=np.int32)
    for w in range(n_walkers):
        rng = np.random.default_rng(seed0 + w)
        if model == "heisenberg":
            base = np.array([0, 1] * (L * L // 2) + [0] * (L * L % 2), dtype=np.int32)
            rng.shuffle(base)
            out[w] = base.reshape(L, L)
        else:
            out[w] = rng.integers(0, 2, size=(L, L), dtype=np.int32)
    return out


def sitps_to_flat(sitps, D, dtype=np.float64):
    """Pack into the C-ABI upload layout [row][col][s][L][D][R][U] zero-padded to D^4 per
    component (include/pepsgpu.h: pepsgpu_state_upload)."""
    L = len(sitps)
    d = len(sitps[0][0])
    flat = np.zeros((L, L, d, D, D, D, D), dtype=dtype)
    for r in range(L):
        for c in range(L):
            for s in range(d):
                t = sitps[r][c][s]
                flat[r, c, s, :t.shape[0], :t.shape[1], :t.shape[2], :t.shape[3]] = t
    return flat


REAL_FIXTURE = "tps_square_heisenberg4x4D8Double"


def tile_flat_state(flat_small, L):
    """A L x L state of the rank of a real PEPS: the site tensors of a small optimised state (flat upload layout
    [r][c][s][L][D][R][U] of an l x l lattice, l >= 4 even) repeated by POSITION CLASS -- the four corners, the edge
    tensors, and the interior tensors with the period of the small lattice's interior ((l-2) x (l-2)).  Leg dimensions
    match by construction (boundary legs 1, bulk legs D); the bond gauges of neighbouring copies do not, so this is not a
    physical state -- it is a workload whose site tensors have the singular spectra of a VMC-optimised PEPS (the
    reference ships no optimised state beyond 4 x 4: tests/slow_tests/test_data/tps_square_heisenberg4x4D8Double,
    test_boson_mc_peps_measure.cpp:55-62)."""
    l = flat_small.shape[0]
    assert flat_small.shape[1] == l and l >= 4 and L >= l
    per = l - 2

    def cls(x):
        return 0 if x == 0 else (l - 1 if x == L - 1 else 1 + (x - 1) % per)

    idx = [cls(x) for x in range(L)]
    return np.ascontiguousarray(flat_small[np.ix_(idx, idx)])


def flat_to_sitps(flat, dtype=np.float64):
    """Inverse of sitps_to_flat: per-site component arrays with their true leg dimensions (boundary legs 1)."""
    L, D = flat.shape[0], flat.shape[3]
    out = []
    for r in range(L):
        row = []
        for c in range(L):
            dl, dd, dr, du = bond_dims(L, D, r, c)
            row.append([np.ascontiguousarray(flat[r, c, s, :dl, :dd, :dr, :du]).astype(dtype) for s in range(flat.shape[2])])
        out.append(row)
    return out


def make_configs_near_neel(L, n_walkers, n_swaps=None, seed0=7):
    """Configurations of the kind a Monte-Carlo run on an antiferromagnetic state visits: the checkerboard with `n_swaps`
    random nearest-neighbour exchanges applied (default L*L/8), rng(seed0 + w).  (Uniformly random Sz = 0 shuffles have
    amplitudes ~25 orders of magnitude below the typical one on an optimised Heisenberg state: no chain ever sits there.)"""
    if n_swaps is None:
        n_swaps = L * L // 8
    out = np.empty((n_walkers, L, L), dtype=np.int32)
    base = checkerboard(L)
    for w in range(n_walkers):
        rng = np.random.default_rng(seed0 + w)
        cfg = base.copy()
        for _ in range(n_swaps):
            r, c = int(rng.integers(0, L)), int(rng.integers(0, L))
            if rng.integers(0, 2):
                r2, c2 = r, c + 1
            else:
                r2, c2 = r + 1, c
            if r2 >= L or c2 >= L:
                continue
            cfg[r, c], cfg[r2, c2] = cfg[r2, c2], cfg[r, c]
        out[w] = cfg
    return out
