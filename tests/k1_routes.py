"""The 21 contraction routes of Contract2DTNUsingBMPSContractor
(reference tests/test_2d_tn/test_bmps_contractor.cpp:273-405) as data, so that the oracle (CPU suite)
and the device (GPU suite) walk exactly the same sequence.  Every 'amp' entry must reproduce the
partition function; replacement traces re-insert the original tensors."""
from oracle.bmps import LEFT, DOWN, RIGHT, UP, HORIZONTAL, VERTICAL
from oracle.contractor import LEFTUP_TO_RIGHTDOWN as LU, LEFTDOWN_TO_RIGHTUP as LD

def routes(rows=12):
    """the walk for a lattice with `rows` rows (the vertical traces sit at rows - 2 and rows - 3, :303-311)"""
    return [(op[0],) + tuple((rows - 12 + x) if (op[0] in ("trace", "tnn") and op[3] == VERTICAL and i == 0) else x
                             for i, x in enumerate(op[1:])) for op in ROUTES]


def routes_by_pass(rows=12):
    """(horizontal ops, vertical ops): the row passes (UP / DOWN stacks, LEFT / RIGHT environments, HORIZONTAL traces) and the
    column passes of the walk.  A fermionic network is decorated per mode order (row-major for the row passes, column-major for
    the column passes: peps_amd/fermion.py), so the two groups run on two contexts; every environment an op needs is grown
    inside its own group."""
    hor, ver = [], []
    cur = hor
    for op in routes(rows):
        if op[0] == "grow_row": cur = hor
        elif op[0] == "grow_col": cur = ver
        cur.append(op)
    return hor, ver


def _walk(ops, c, tn, device):
    amps = []
    for op in ops:
        k, a = op[0], op[1:]
        if device:
            if k == "grow_row": c.grow_bmps_for_row(a[0])
            elif k == "grow_col": c.grow_bmps_for_col(a[0])
            elif k == "init_bten": c.init_bten(a[0], a[1])
            elif k == "grow_full_bten": c.grow_full_bten(a[0], a[1], a[2], True)
            elif k == "shift_bten": c.shift_bten_window(a[0])
            elif k == "init_bten2": c.init_bten2(a[0], a[1])
            elif k == "grow_full_bten2": c.grow_full_bten2(a[0], a[1], a[2], True)
            elif k == "shift_bten2": c.shift_bten2_window(a[0], a[1])
            elif k == "trace": amps.append(c.trace(a[0], a[1], a[2]))
            elif k == "tnn": amps.append(c.replace_tnn_trace(a[0], a[1], a[2]))
            elif k == "nnn": amps.append(c.replace_nnn_trace(a[0], a[1], a[2], a[3]))
            elif k == "sqrt5": amps.append(c.replace_sqrt5_trace(a[0], a[1], a[2], a[3]))
            else: raise KeyError(k)
        else:
            if k == "grow_row": c.GrowBMPSForRow(tn, a[0])
            elif k == "grow_col": c.GrowBMPSForCol(tn, a[0])
            elif k == "init_bten": c.InitBTen(tn, a[0], a[1])
            elif k == "grow_full_bten": c.GrowFullBTen(tn, a[0], a[1], a[2], True)
            elif k == "shift_bten": c.ShiftBTenWindow(tn, a[0])
            elif k == "init_bten2": c.InitBTen2(tn, a[0], a[1])
            elif k == "grow_full_bten2": c.GrowFullBTen2(tn, a[0], a[1], a[2], True)
            elif k == "shift_bten2": c.ShiftBTen2Window(tn, a[0], a[1])
            elif k == "trace": amps.append(c.Trace(tn, (a[0], a[1]), a[2]))
            elif k == "tnn":
                st = tnn_sites(*a)
                amps.append(c.ReplaceTNNSiteTrace(tn, st[0], a[2], tn(st[0]), tn(st[1]), tn(st[2])))
            elif k == "nnn":
                sl, sr = nnn_sites(a[0], a[1], a[2])
                amps.append(c.ReplaceNNNSiteTrace(tn, (a[0], a[1]), a[2], a[3], tn(sl), tn(sr)))
            elif k == "sqrt5":
                sl, sr = sqrt5_sites(*a)
                amps.append(c.ReplaceSqrt5DistTwoSiteTrace(tn, (a[0], a[1]), a[2], a[3], tn(sl), tn(sr)))
            else: raise KeyError(k)
    return amps


ROUTES = [
    ("grow_row", 2), ("init_bten", LEFT, 2), ("grow_full_bten", RIGHT, 2, 2),
    ("trace", 2, 0, HORIZONTAL), ("tnn", 2, 0, HORIZONTAL),
    ("shift_bten", RIGHT),
    ("trace", 2, 1, HORIZONTAL), ("tnn", 2, 1, HORIZONTAL),
    ("grow_col", 1), ("init_bten", DOWN, 1), ("grow_full_bten", UP, 1, 2),
    ("trace", 10, 1, VERTICAL),
    ("shift_bten", UP),
    ("trace", 9, 1, VERTICAL), ("tnn", 9, 1, VERTICAL),
    ("grow_row", 1), ("init_bten2", LEFT, 1), ("grow_full_bten2", RIGHT, 1, 2),
    ("nnn", 1, 0, LD, HORIZONTAL), ("nnn", 1, 0, LU, HORIZONTAL),
    ("shift_bten2", RIGHT, 1),
    ("nnn", 1, 1, LD, HORIZONTAL), ("nnn", 1, 1, LU, HORIZONTAL),
    ("sqrt5", 1, 0, LD, HORIZONTAL), ("sqrt5", 1, 1, LD, HORIZONTAL),
    ("sqrt5", 1, 0, LU, HORIZONTAL), ("sqrt5", 1, 1, LU, HORIZONTAL),
    ("grow_col", 1), ("grow_full_bten2", DOWN, 1, 2), ("grow_full_bten2", UP, 1, 2),
    ("nnn", 2, 1, LD, VERTICAL), ("nnn", 2, 1, LU, VERTICAL),
    ("shift_bten2", UP, 1),
    ("nnn", 1, 1, LD, VERTICAL), ("nnn", 1, 1, LU, VERTICAL),
    ("sqrt5", 1, 1, LD, VERTICAL), ("sqrt5", 1, 1, LU, VERTICAL),
]
N_AMPS = 21


def nnn_sites(r, c, d):
    """(left, right) sites of the diagonal of the plaquette with upper-left corner (r, c)"""
    return ((r, c), (r + 1, c + 1)) if d == LU else ((r + 1, c), (r, c + 1))


def sqrt5_sites(r, c, d, orient):
    if orient == HORIZONTAL:
        return ((r, c), (r + 1, c + 2)) if d == LU else ((r + 1, c), (r, c + 2))
    return ((r, c), (r + 2, c + 1)) if d == LU else ((r + 2, c), (r, c + 1))


def tnn_sites(r, c, orient):
    return [(r, c + k) if orient == HORIZONTAL else (r + k, c) for k in range(3)]


def run_oracle(c, tn, rows=12):
    amps = []
    for op in routes(rows):
        k, a = op[0], op[1:]
        if k == "grow_row": c.GrowBMPSForRow(tn, a[0])
        elif k == "grow_col": c.GrowBMPSForCol(tn, a[0])
        elif k == "init_bten": c.InitBTen(tn, a[0], a[1])
        elif k == "grow_full_bten": c.GrowFullBTen(tn, a[0], a[1], a[2], True)
        elif k == "shift_bten": c.ShiftBTenWindow(tn, a[0])
        elif k == "init_bten2": c.InitBTen2(tn, a[0], a[1])
        elif k == "grow_full_bten2": c.GrowFullBTen2(tn, a[0], a[1], a[2], True)
        elif k == "shift_bten2": c.ShiftBTen2Window(tn, a[0], a[1])
        elif k == "trace": amps.append(c.Trace(tn, (a[0], a[1]), a[2]))
        elif k == "tnn":
            s = tnn_sites(*a)
            amps.append(c.ReplaceTNNSiteTrace(tn, s[0], a[2], tn(s[0]), tn(s[1]), tn(s[2])))
        elif k == "nnn":
            sl, sr = nnn_sites(a[0], a[1], a[2])
            amps.append(c.ReplaceNNNSiteTrace(tn, (a[0], a[1]), a[2], a[3], tn(sl), tn(sr)))
        elif k == "sqrt5":
            sl, sr = sqrt5_sites(*a)
            amps.append(c.ReplaceSqrt5DistTwoSiteTrace(tn, (a[0], a[1]), a[2], a[3], tn(sl), tn(sr)))
        else:
            raise KeyError(k)
    return amps


def run_device(ctx, rows=12):
    """same walk through the C ABI; replacement = the walker's own states (no candidate table)"""
    amps = []
    for op in routes(rows):
        k, a = op[0], op[1:]
        if k == "grow_row": ctx.grow_bmps_for_row(a[0])
        elif k == "grow_col": ctx.grow_bmps_for_col(a[0])
        elif k == "init_bten": ctx.init_bten(a[0], a[1])
        elif k == "grow_full_bten": ctx.grow_full_bten(a[0], a[1], a[2], True)
        elif k == "shift_bten": ctx.shift_bten_window(a[0])
        elif k == "init_bten2": ctx.init_bten2(a[0], a[1])
        elif k == "grow_full_bten2": ctx.grow_full_bten2(a[0], a[1], a[2], True)
        elif k == "shift_bten2": ctx.shift_bten2_window(a[0], a[1])
        elif k == "trace": amps.append(ctx.trace(a[0], a[1], a[2]))
        elif k == "tnn": amps.append(ctx.replace_tnn_trace(a[0], a[1], a[2]))
        elif k == "nnn": amps.append(ctx.replace_nnn_trace(a[0], a[1], a[2], a[3]))
        elif k == "sqrt5": amps.append(ctx.replace_sqrt5_trace(a[0], a[1], a[2], a[3]))
        else:
            raise KeyError(k)
    return amps
