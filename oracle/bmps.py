"""Boundary MPS: restatement of include/qlpeps/one_dim_tn/boundary_mps/bmps{.h,_impl.h}.

Bosonic (rank-3 BMPS tensors, rank-4 site tensors with leg order L=0, D=1, R=2, U=3,
tensor_network_2d.h:39-45).  Oracle = test infrastructure only.
"""
from dataclasses import dataclass
import numpy as np

from . import tensor as T

# include/qlpeps/basic.h:58-63 (enum BMPSPOSITION) and :19-22 (BondOrientation)
LEFT, DOWN, RIGHT, UP = 0, 1, 2, 3
HORIZONTAL, VERTICAL = 0, 1

SVD_COMPRESS = 0  # bmps.h:31-35 CompressMPSScheme (variational schemes are a "next" row)


def opposite(post):
    """basic.h:79-87"""
    return (post + 2) % 4


def orientation(post):
    """basic.h:71-73"""
    return post % 2


def rotate(orient):
    """basic.h:24-26"""
    return 1 - orient


@dataclass
class BMPSTruncateParams:
    """bmps.h:47-98"""
    D_min: int = 1
    D_max: int = 2 ** 62
    trunc_err: float = 0.0
    compress_scheme: int = SVD_COMPRESS

    @staticmethod
    def SVD(d_min, d_max, trunc_err):
        return BMPSTruncateParams(d_min, d_max, trunc_err, SVD_COMPRESS)


class BMPS:
    """bmps.h:153-363.  tensors[i] has legs (left, phys, right); UP/RIGHT are stored reversed
    (bmps.h:145-152)."""

    def __init__(self, position, tensors):
        self.position = position
        self.tensors = list(tensors)

    @staticmethod
    def boundary(position, phys_dims, dtype=np.float64):
        """bmps_impl.h:60-96: product state of (1, d, 1) tensors with element 1; d must be 1."""
        tens = []
        for d in phys_dims:
            assert d == 1
            t = np.zeros((1, d, 1), dtype=dtype)
            t[0, 0, 0] = 1.0
            tens.append(t)
        return BMPS(position, tens)

    def __len__(self):
        return len(self.tensors)

    def __getitem__(self, i):
        return self.tensors[i]

    def at_logical_col(self, col):
        """bmps.h:214-227"""
        if self.position in (UP, RIGHT):
            return self.tensors[len(self.tensors) - 1 - col]
        return self.tensors[col]

    def copy(self):
        return BMPS(self.position, [t.copy() for t in self.tensors])

    # ------------------------------------------------------------------
    def right_canonicalize_truncate(self, site, dmin, dmax, trunc_err):
        """bmps_impl.h:225-263: SVD(res[site], ldims=1) truncated; res[site] = vt;
        res[site-1] <- res[site-1] . (u s)."""
        u, s, vt, err, d = T.svd_trunc(self.tensors[site], 1, trunc_err, dmin, dmax)
        self.tensors[site] = vt
        us = u * s[None, :]
        self.tensors[site - 1] = T.contract(self.tensors[site - 1], [2], us, [0])
        return d, err

    def multiply_mpo(self, mpo, scheme, dmin, dmax, trunc_err):
        """bmps_impl.h:404-437.  `mpo` = list of rank-4 site tensors of the absorbed row/col in
        LOGICAL order; reversed here for UP/RIGHT as ReverseTransferMPOIfNeeded_ does (:694-699)."""
        assert len(mpo) == len(self.tensors)
        mpo = list(mpo)
        if self.position > 1:  # RIGHT or UP
            mpo.reverse()
        if scheme != SVD_COMPRESS and len(self.tensors) != 2:
            raise NotImplementedError("variational compression is a 'next' row (SURVEY 8f-3)")
        return self._multiply_mpo_svd_compress(mpo, dmin, dmax, trunc_err)

    def _multiply_mpo_svd_compress(self, mpo, dmin, dmax, trunc_err):
        """bmps_impl.h:756-862 (bosonic branch)."""
        n = len(self.tensors)
        pos = self.position
        pre_post = (pos + 3) % 4                      # bmps.h:283
        dtype = np.result_type(self.tensors[0].dtype, mpo[0].dtype)
        res = [None] * n
        # :769-772  r = IndexCombine(mpo_left, mps_left) transposed {2,0,1} -> (comb, mpo, mps)
        d1 = mpo[0].shape[pre_post]
        d2 = self.tensors[0].shape[0]
        r = np.transpose(T.index_combine(d1, d2, dtype), (2, 0, 1))
        for i in range(n):
            tmp1 = T.contract_cyclic(self.tensors[i], r, 0, 2, 1)        # :806
            tmp2 = T.contract_cyclic(tmp1, mpo[i], 3, pre_post, 2)       # :807
            if i < n - 1:
                tmp2 = np.transpose(tmp2, (1, 3, 2, 0))                   # :815-817
                res[i], r = T.qr(tmp2, 2)                                 # :821
            else:
                rb = T.index_combine(tmp2.shape[0], tmp2.shape[2], dtype)  # :827-831
                res[i] = T.contract(tmp2, [0, 2], rb, [0, 1])             # :838
                if res[i].size == 0 or not np.any(res[i]):
                    # :839-843 (GetActualDataSize()==0 <=> no non-zero block for a dense tensor)
                    raise RuntimeError("BMPS::MultiplyMPOSVDCompress_: Empty tensor at site %d" % i)
        out = BMPS(pos, res)
        self.last_actual_dmax = 1
        self.last_trunc_err_max = 0.0
        for i in range(n - 1, 0, -1):                                     # :853-857
            d, err = out.right_canonicalize_truncate(i, dmin, dmax, trunc_err)
            self.last_actual_dmax = max(self.last_actual_dmax, d)
            self.last_trunc_err_max = max(self.last_trunc_err_max, err)
        return out
