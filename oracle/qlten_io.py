"""Reader for the reference's on-disk state format (dense / TrivialRepQN and U1-like block files).

Format restated from SURVEY.md section 8c (reverse-engineered from the fixtures; writer =
SplitIndexTPS::Dump, include/qlpeps/two_dim_tn/tps/split_index_tps_impl.h:300-437, file names
split_index_tps.h:23-29, meta line :317-324).  Oracle = test infrastructure only.
"""
import os
import numpy as np


def _read_tokens(buf, pos, n):
    toks = []
    for _ in range(n):
        end = buf.index(b"\n", pos)
        toks.append(buf[pos:end])
        pos = end + 1
    return toks, pos


def load_qlten(path, complex_data=False, qn_fields=0):
    """Parse one .qlten file into a dense ndarray.

    Header (ASCII, newline separated): rank; per leg: n_sectors, per sector [qn fields]*qn_fields,
    degeneracy, sector_hash; dir (-1 IN / +1 OUT), dim, index_hash; n_blocks; per block `rank`
    sector coordinates; then raw little-endian payload (float64 or interleaved complex128), blocks
    in listed order, each row-major over its sector degeneracies.
    """
    with open(path, "rb") as f:
        buf = f.read()
    pos = 0
    (tok,), pos = _read_tokens(buf, pos, 1)
    rank = int(tok)
    legs = []
    for _ in range(rank):
        (tok,), pos = _read_tokens(buf, pos, 1)
        nsec = int(tok)
        degs = []
        for _s in range(nsec):
            toks, pos = _read_tokens(buf, pos, qn_fields + 2)
            degs.append(int(toks[qn_fields]))
        toks, pos = _read_tokens(buf, pos, 3)
        direction, dim = int(toks[0]), int(toks[1])
        assert sum(degs) == dim, (path, degs, dim)
        legs.append((degs, direction))
    (tok,), pos = _read_tokens(buf, pos, 1)
    nblocks = int(tok)
    blocks = []
    for _ in range(nblocks):
        toks, pos = _read_tokens(buf, pos, rank)
        blocks.append([int(t) for t in toks])
    dtype = np.complex128 if complex_data else np.float64
    shape = tuple(sum(d) for d, _ in legs)
    out = np.zeros(shape, dtype=dtype)
    offs = [np.concatenate([[0], np.cumsum(d)]) for d, _ in legs]
    item = 16 if complex_data else 8
    for coords in blocks:
        bshape = tuple(legs[k][0][coords[k]] for k in range(rank))
        n = int(np.prod(bshape))
        data = np.frombuffer(buf, dtype="<c16" if complex_data else "<f8", count=n, offset=pos)
        pos += n * item
        sl = tuple(slice(offs[k][coords[k]], offs[k][coords[k] + 1]) for k in range(rank))
        out[sl] = data.reshape(bshape)
    rest = buf[pos:]
    assert rest in (b"", b"\n"), "trailing bytes in %s: %d" % (path, len(rest))
    return out


def load_sitps(directory, complex_data=False, qn_fields=0):
    """Load a SplitIndexTPS dump: tps_meta.txt = 'rows cols phy_dim [bc]' and one
    tps_ten{row}_{col}_{component}.qlten per (site, physical state).  Returns sitps[r][c][s]
    (rank-4 arrays, legs L, D, R, U)."""
    with open(os.path.join(directory, "tps_meta.txt")) as f:
        toks = f.read().split()
    rows, cols, d = int(toks[0]), int(toks[1]), int(toks[2])
    sitps = [[[load_qlten(os.path.join(directory, "tps_ten%d_%d_%d.qlten" % (r, c, s)), complex_data, qn_fields)
               for s in range(d)] for c in range(cols)] for r in range(rows)]
    return sitps


def load_configuration(path, rows, cols):
    """configuration{rank}: text matrix (include/qlpeps/vmc_basic/configuration.h:284-310)."""
    with open(path) as f:
        vals = [int(x) for x in f.read().split()]
    return np.array(vals[:rows * cols], dtype=np.int64).reshape(rows, cols)


# ---------------------------------------------------------------------------------------------
# Writer (dense TrivialRepQN tensors): byte-identical to the files the reference writes.  The two
# hash fields were recovered from the fixtures (8 distinct (dir, dim) pairs over 168 files, all
# reproduced): sector_hash = qn_hash ^ degeneracy with qn_hash = 0; index_hash = VecHash(sector
# hashes) ^ std::hash<int>(dir), VecHash = the xxHash-style tuple hash (acc = P5; per lane
# acc += lane*P2, rotl 31, *= P1; acc += len ^ P5).
_M64 = (1 << 64) - 1
_P1, _P2, _P5 = 0x9E3779B185EBCA87, 0xC2B2AE3D27D4EB4F, 0x27D4EB2F165667C5


def trivial_index_hash(direction, dim):
    acc = (_P5 + dim * _P2) & _M64
    acc = ((acc << 31) | (acc >> 33)) & _M64
    acc = (acc * _P1) & _M64
    acc = (acc + (1 ^ _P5)) & _M64
    return acc ^ (1 if direction == 1 else _M64)


SITE_LEG_DIRS = (-1, 1, 1, -1)     # (L, D, R, U): IN, OUT, OUT, IN as in every fixture


def save_qlten(path, arr, dirs=SITE_LEG_DIRS):
    arr = np.asarray(arr)
    cplx = np.iscomplexobj(arr)
    head = ["%d" % arr.ndim]
    for k in range(arr.ndim):
        dim = arr.shape[k]
        head += ["1", "%d" % dim, "%d" % dim, "%d" % dirs[k], "%d" % dim, "%d" % trivial_index_hash(dirs[k], dim)]
    head += ["1"] + ["0"] * arr.ndim
    with open(path, "wb") as f:
        f.write(("\n".join(head) + "\n").encode())
        f.write(np.ascontiguousarray(arr, dtype="<c16" if cplx else "<f8").tobytes())
        f.write(b"\n")


def save_sitps(directory, sitps, with_bc=True):
    """SplitIndexTPS::Dump (split_index_tps_impl.h:300-330); meta = 'rows cols phy_dim [bc]' (bc 0 = open)."""
    os.makedirs(directory, exist_ok=True)
    rows, cols, d = len(sitps), len(sitps[0]), len(sitps[0][0])
    for r in range(rows):
        for c in range(cols):
            for s in range(d):
                save_qlten(os.path.join(directory, "tps_ten%d_%d_%d.qlten" % (r, c, s)), sitps[r][c][s])
    with open(os.path.join(directory, "tps_meta.txt"), "wb") as f:
        f.write(("%d %d %d" % (rows, cols, d) + (" 0" if with_bc else "")).encode())


def save_configuration(path, config):
    """Configuration::StreamWrite (configuration.h:457-464) + the .shape sidecar (:303-309)."""
    cfg = np.asarray(config)
    with open(path, "w") as f:
        for row in cfg:
            f.write(" ".join(str(int(x)) for x in row) + "\n")
    with open(path + ".shape", "w") as f:
        f.write("%d %d\n" % cfg.shape)
