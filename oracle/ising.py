"""K1 known-answer network: OBC classical 2D Ising partition function as a 2D tensor network and
its exact transfer-matrix free energy.

Restates tests/test_2d_tn/test_bmps_contractor.cpp:27-126 (SquareIsingModel, exact solution) and
:128-271 (OBCIsing2DTenNetWithoutZ2::SetUp, network construction).  Oracle = test infrastructure.
"""
import numpy as np

from .contractor import TensorNetwork2D


def exact_free_energy(lx, ly, temperature):
    """test_bmps_contractor.cpp:27-126: transfer matrix over columns of height ly."""
    if lx < ly:
        lx, ly = ly, lx
    dim = 1 << ly
    idx = np.arange(dim)
    bits = (idx[:, None] >> np.arange(ly)[None, :]) & 1            # bit b of config
    # CalHalfEnergyChain_ (:100-105): (#different NN bonds) - (ly-1)/2
    diff = np.sum(bits[:, :-1] != bits[:, 1:], axis=1)
    half = diff - (ly - 1) / 2.0
    # CalLadderEnergy_ (:108-112): 2*#different - ly
    x = idx[:, None] ^ idx[None, :]
    pop = np.zeros_like(x)
    for b in range(ly):
        pop += (x >> b) & 1
    ladder = 2.0 * pop - ly
    e = half[:, None] + half[None, :] + ladder
    tm = np.exp(-e / temperature)
    bvec = np.exp(-half / temperature)
    v = bvec.copy()
    for _ in range(lx - 1):
        v = v @ tm
    z = float(v @ bvec)
    return -np.log(z) / (lx * ly) * temperature


def build_ising_tn(lx=12, ly=12, beta=None):
    """test_bmps_contractor.cpp:153-258 without the random complex phases.  Returns
    (tn, sum_log_norms, beta): each site tensor is divided by its 2-norm as Normalize() does
    (:254) and the logs are accumulated."""
    if beta is None:
        beta = np.log(1 + np.sqrt(2.0)) / 2.0
    e = -1.0
    bw = np.array([[np.exp(-beta * e), np.exp(beta * e)],
                   [np.exp(beta * e), np.exp(-beta * e)]])                 # :154-159

    def core(shape):
        return np.zeros(shape)

    def absorb_two(c):
        # temp = Contract(bw,{1}, c,{3}); t = Contract(bw,{0}, temp,{3}); t.Transpose({2,3,0,1})
        temp = np.tensordot(bw, c, axes=([1], [3]))
        t = np.tensordot(bw, temp, axes=([0], [3]))
        return np.transpose(t, (2, 3, 0, 1))

    cm = core((2, 2, 2, 2))
    for i in range(2):
        cm[i, i, i, i] = 1.0
    t_m = absorb_two(cm)                                                   # :165-171

    c_up, c_left = core((2, 2, 2, 1)), core((1, 2, 2, 2))
    c_down, c_right = core((2, 1, 2, 2)), core((2, 2, 1, 2))
    for i in range(2):
        c_left[0, i, i, i] = 1.0
        c_up[i, i, i, 0] = 1.0
        c_down[i, 0, i, i] = 1.0
        c_right[i, i, 0, i] = 1.0
    # :190-196  temp = c_up.Transpose({3,0,1,2}); t_up = Contract(bw,{0}, temp,{3}); Transpose({2,3,0,1})
    temp = np.transpose(c_up, (3, 0, 1, 2))
    t_up = np.transpose(np.tensordot(bw, temp, axes=([0], [3])), (2, 3, 0, 1))
    t_left = absorb_two(c_left)                                            # :197-201
    t_right = np.transpose(np.tensordot(bw, c_right, axes=([1], [3])), (1, 2, 3, 0))   # :202-204
    t_down = absorb_two(c_down)                                            # :205-209

    c_lu, c_ll = core((1, 2, 2, 1)), core((1, 1, 2, 2))
    c_rl, c_ru = core((2, 1, 1, 2)), core((2, 2, 1, 1))
    for i in range(2):
        for j in range(2):
            c_lu[0, i, j, 0] = bw[i, j]                                    # :224-230
            c_rl[i, 0, 0, j] = bw[i, j]
    for i in range(2):
        c_ll[0, 0, i, i] = 1.0
        c_ru[i, i, 0, 0] = 1.0
    c_ll = absorb_two(c_ll)                                                # :236-242

    tn = TensorNetwork2D(ly, lx)
    for r in range(1, ly - 1):
        for c in range(1, lx - 1):
            tn.set((r, c), t_m)
    for r in range(1, ly - 1):
        tn.set((r, 0), t_left)
        tn.set((r, lx - 1), t_right)
    for c in range(1, lx - 1):
        tn.set((0, c), t_up)
        tn.set((ly - 1, c), t_down)
    tn.set((0, 0), c_lu)
    tn.set((ly - 1, 0), c_ll)
    tn.set((ly - 1, lx - 1), c_rl)
    tn.set((0, lx - 1), c_ru)
    log_norm = 0.0
    for r in range(ly):
        for c in range(lx):
            t = tn((r, c))
            nrm = np.linalg.norm(t)
            log_norm += np.log(nrm)
            tn.set((r, c), t / nrm)
    return tn, log_norm, beta


def exact_contract(tn):
    """Brute-force contraction of a small OBC network row by row (no truncation); used to pin
    amplitudes of the 2x2 / 4x4 fixtures independently of the BMPS machinery."""
    rows, cols = tn.rows, tn.cols
    # boundary vector over the down legs of a row: shape (d_0, d_1, ..., d_{cols-1})
    vec = np.ones((1,) * cols)
    for r in range(rows):
        # contract row r (legs L, D, R, U); U legs join `vec`, result indexed by D legs
        cur = None
        for c in range(cols):
            t = tn((r, c))
            if cur is None:
                assert t.shape[0] == 1
                # vec axes: (u0, u1, ...); contract u0
                cur = np.tensordot(t[0], vec, axes=([2], [0]))      # (D0, R0, u1, ...)
            else:
                # cur: (D0..D_{c-1}, R_{c-1}, u_c, ...)
                nd = c
                cur = np.tensordot(cur, t, axes=([nd, nd + 1], [0, 3]))   # (D.., u_{c+1}.., D_c, R_c)
                # move D_c, R_c after the D's
                nrest = cur.ndim - nd - 2
                perm = list(range(nd)) + [nd + nrest, nd + nrest + 1] + list(range(nd, nd + nrest))
                cur = np.transpose(cur, perm)
        assert cur.shape[cols] == 1
        vec = cur.reshape(cur.shape[:cols])
    return vec.reshape(-1)[0]
