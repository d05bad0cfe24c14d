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

SVD_COMPRESS, VARIATION2Site, VARIATION1Site = 0, 1, 2  # bmps.h:31-35 CompressMPSScheme


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
    convergence_tol: float = None
    iter_max: int = None

    @staticmethod
    def SVD(d_min, d_max, trunc_err):
        return BMPSTruncateParams(d_min, d_max, trunc_err, SVD_COMPRESS)

    @staticmethod
    def Variational2Site(d_min, d_max, trunc_err, convergence_tol, iter_max):
        """bmps.h:81-88"""
        return BMPSTruncateParams(d_min, d_max, trunc_err, VARIATION2Site, convergence_tol, iter_max)

    @staticmethod
    def Variational1Site(d_min, d_max, trunc_err, convergence_tol, iter_max):
        """bmps.h:90-97"""
        return BMPSTruncateParams(d_min, d_max, trunc_err, VARIATION1Site, convergence_tol, iter_max)


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

    def multiply_mpo(self, mpo, scheme, dmin, dmax, trunc_err, convergence_tol=None, iter_max=None):
        """bmps_impl.h:404-437.  `mpo` = list of rank-4 site tensors of the absorbed row/col in
        LOGICAL order; reversed here for UP/RIGHT as ReverseTransferMPOIfNeeded_ does (:694-699)."""
        assert len(mpo) == len(self.tensors)
        mpo = list(mpo)
        if self.position > 1:  # RIGHT or UP
            mpo.reverse()
        if len(self.tensors) == 2 or scheme == SVD_COMPRESS:                      # :419-423
            return self._multiply_mpo_svd_compress(mpo, dmin, dmax, trunc_err)
        if scheme == VARIATION2Site:
            return self._multiply_mpo_2site_variational(mpo, dmin, dmax, trunc_err, convergence_tol, iter_max)
        if scheme == VARIATION1Site:
            return self._multiply_mpo_1site_variational(mpo, dmin, dmax, trunc_err, convergence_tol, iter_max)
        raise SystemExit("Do not support MPO multiplication method.")             # :432-435

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

    # ------------------------------------------------------------------
    # Variational compression (bmps_impl.h:864-1212, bosonic only: the reference asserts
    # !IsFermionic()).  Environments: lenv (res bond, mpo pre leg, mps bond),
    # renv (mps bond, mpo next leg, res bond)  (:701-728).  res_dag holds conj(res).
    def _variational_init_guess(self, mpo, dmin, dmax, trunc_err):
        """MakeVariationalInitGuess_ (:1174-1212): the BMPS truncated to bond dimension 2
        (left-canonicalise, then SVD sweep with Dmin=1, Dmax=2), times the MPO, SVD-compressed."""
        cp = self.copy()
        n = len(cp.tensors)
        for i in range(n - 1):                                       # Centralize(N-1): :119-176
            q, r = T.qr(cp.tensors[i], 2)
            cp.tensors[i] = q
            cp.tensors[i + 1] = T.contract(r, [1], cp.tensors[i + 1], [0])
        for i in range(n - 1, 0, -1):
            cp.right_canonicalize_truncate(i, 1, 2, 0.0)
        return cp._multiply_mpo_svd_compress(mpo, dmin, dmax, trunc_err)

    def _var_setup(self, mpo, res_dag):
        """MakeEnvironmentBoundaries_ + GrowRightEnvironments_ (:701-743)."""
        n = len(self.tensors)
        dtype = res_dag[0].dtype
        lenv0 = np.zeros((1, 1, 1), dtype=dtype); lenv0[0, 0, 0] = 1.0
        renv0 = np.zeros((1, 1, 1), dtype=dtype); renv0[0, 0, 0] = 1.0
        lenvs, renvs = [lenv0], [renv0]
        for i in range(n - 1, 1, -1):
            renvs.append(self._renv_step(mpo, i, renvs[-1], res_dag[i])[1])
        return lenvs, renvs

    def _renv_step(self, mpo, i, renv, res_dag_i):
        """:735-739 / :899-900: (mps[i] . renv) . mpo[i] -> (kr, l, opp, pre); closed with res_dag[i]."""
        pos = self.position
        t2 = T.contract_cyclic(self.tensors[i], renv, 2, 0, 1)
        t3 = T.contract_cyclic(t2, mpo[i], 1, pos, 2)
        nxt = None if res_dag_i is None else T.contract(t3, [2, 0], res_dag_i, [1, 2])
        return t3, nxt

    def _lenv_half(self, mpo, i, lenv):
        """:896-897: (lenv . mps[i]) . mpo[i] -> (r, k, next, opp)."""
        pre_post = (self.position + 3) % 4
        t0 = T.contract_cyclic(lenv, self.tensors[i], 2, 0, 1)
        return T.contract_cyclic(t0, mpo[i], 1, pre_post, 2)

    def _two_site(self, mpo, i, lenv, renv, trunc_err, dmin, dmax):
        """:895-908: the two-site tensor of the exact product, daggered, and its truncated SVD."""
        t1 = self._lenv_half(mpo, i, lenv)
        t3, _ = self._renv_step(mpo, i + 1, renv, None)
        theta = np.conj(T.contract(t1, [2, 0], t3, [3, 1]))                     # (k, opp, kr, opp')
        u, s, vt, _, _ = T.svd_trunc(theta, 2, trunc_err, dmin, dmax)
        return t1, t3, u, s, vt

    def _two_site_sweeps(self, mpo, res_dag, lenvs, renvs, trunc_err, dmin, dmax):
        """one left-to-right and one right-to-left pass of two-site updates (:892-945)."""
        n = len(self.tensors)
        s = None
        for i in range(n - 2):
            t1, _, u, s, _ = self._two_site(mpo, i, lenvs[-1], renvs[-1], trunc_err, dmin, dmax)
            res_dag[i] = u
            lenvs.append(np.transpose(T.contract(t1, [1, 3], u, [0, 1]), (2, 1, 0)))
            renvs.pop()
        for i in range(n - 2, 0, -1):
            _, t3, _, s, vt = self._two_site(mpo, i, lenvs[-1], renvs[-1], trunc_err, dmin, dmax)
            res_dag[i + 1] = np.transpose(vt, (0, 2, 1))
            renvs.append(T.contract(t3, [2, 0], res_dag[i + 1], [1, 2]))
            lenvs.pop()
        return s

    def _multiply_mpo_2site_variational(self, mpo, dmin, dmax, trunc_err, tol, max_iter):
        """MultiplyMPO2SiteVariationalCompress_ (:864-995)."""
        init = self._variational_init_guess(mpo, dmin, dmax, trunc_err)
        res_dag = [np.conj(t) for t in init.tensors]
        lenvs, renvs = self._var_setup(mpo, res_dag)
        s_last = None
        self.last_var_iters = 0
        for it in range(max_iter):
            s = self._two_site_sweeps(mpo, res_dag, lenvs, renvs, trunc_err, dmin, dmax)
            self.last_var_iters = it + 1
            if it == 0 or len(s) != len(s_last):                                # :946-949
                s_last = s
                continue
            if float(np.sum(np.abs(s - s_last))) / s[0] < tol:                  # :950-958
                break
            s_last = s
        _, t3, u, s, vt = self._two_site(mpo, 0, lenvs[-1], renvs[-1], trunc_err, dmin, dmax)
        res_dag[0] = u * s[None, None, :]                                       # :983-984
        res_dag[1] = np.transpose(vt, (0, 2, 1))
        return BMPS(self.position, [np.conj(t) for t in res_dag])               # FinalizeCompressedBMPS_

    def _multiply_mpo_1site_variational(self, mpo, dmin, dmax, trunc_err, tol, max_iter):
        """MultiplyMPO1SiteVariationalCompress_ (:997-1172)."""
        n = len(self.tensors)
        init = self._variational_init_guess(mpo, dmax, dmax, 0.0)               # :1012
        res_dag = [np.conj(t) for t in init.tensors]
        lenvs, renvs = self._var_setup(mpo, res_dag)
        self._two_site_sweeps(mpo, res_dag, lenvs, renvs, trunc_err, dmax, dmax)   # :1021-1079
        _, t3, u, s, vt = self._two_site(mpo, 0, lenvs[-1], renvs[-1], trunc_err, dmin, dmax)
        res_dag[0] = u * s[None, None, :]
        res_dag[1] = np.transpose(vt, (0, 2, 1))
        renvs.append(T.contract(t3, [2, 0], res_dag[1], [1, 2]))                # :1106-1107
        last_r_norm = 0.0
        self.last_var_iters = 0
        for it in range(max_iter):
            for i in range(n - 1):                                              # :1113-1128
                t1 = self._lenv_half(mpo, i, lenvs[-1])
                t2 = np.conj(T.contract(t1, [0, 2], renvs[-1], [0, 1]))         # (k, opp, kr)
                q, _ = T.qr(t2, 2)
                res_dag[i] = q
                lenvs.append(np.transpose(T.contract(t1, [1, 3], q, [0, 1]), (2, 1, 0)))
                renvs.pop()
            r_norm = 0.0
            for i in range(n - 1, 0, -1):                                       # :1130-1149
                t1, _ = self._renv_step(mpo, i, renvs[-1], None)                # (kr, l, opp, pre)
                t2 = np.conj(T.contract(t1, [3, 1], lenvs[-1], [1, 2]))         # (kr, opp, k)
                q, r = T.qr(t2, 2)
                res_dag[i] = np.transpose(q, (2, 1, 0))
                renvs.append(T.contract(t1, [2, 0], res_dag[i], [1, 2]))
                lenvs.pop()
                r_norm = float(np.linalg.norm(r))
            self.last_var_iters = it + 1
            if it == 0 or abs(r_norm - last_r_norm) / abs(r_norm) > tol:        # :1150-1155
                last_r_norm = r_norm
                continue
            break
        t1 = self._lenv_half(mpo, 0, lenvs[-1])
        res_dag[0] = np.conj(T.contract(t1, [0, 2], renvs[-1], [0, 1]))         # :1158-1164
        return BMPS(self.position, [np.conj(t) for t in res_dag])
