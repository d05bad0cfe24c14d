"""Stochastic reconfiguration with the O* samples resident in HBM.

Replaces, for the walkers of one GPU, SRSMatrix::operator* (optimizer/stochastic_reconfiguration_smatrix.h:37-99)
and its use inside the natural-gradient solve (utility/conjugate_gradient_solver.h): the S-matrix product is two
HBM-bound sweeps over the sample store of the device (pepsgpu_sr_matvec), the CG vectors are SITPS-shaped host
arrays (9 MB at C4), the cross-rank reduction is one all-reduce(sum) per product (replaces the MPI_Bcast of
mean_dot_v and the MPI_Reduce of the distributed product)."""
import numpy as np


class DeviceSRSMatrix:
    """S v = <(O*_i . v - mean . v) O*_i>_i + diag_shift v over the samples stored in `ctx` (and in the contexts of
    the other ranks when `dist` is an initialised torch.distributed module)."""

    def __init__(self, ctx, diag_shift=0.0, dist=None):
        self.ctx, self.diag_shift, self.dist = ctx, diag_shift, dist
        n_local = ctx.sr_count()
        tot = self._allreduce(np.array([float(n_local)]))[0]
        self.n_total = int(round(tot))
        self.mean = self._allreduce(ctx.sr_sum()) / self.n_total          # Ostar_mean

    def _allreduce(self, arr):
        if self.dist is None:
            return arr
        import torch
        t = torch.from_numpy(np.ascontiguousarray(arr))
        if self.dist.get_backend() == "nccl":
            t = t.cuda()
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return t.cpu().numpy()

    def __mul__(self, v):
        v = np.ascontiguousarray(v, dtype=np.float64)
        mean_dot_v = float(np.vdot(self.mean, v))
        res = self._allreduce(self.ctx.sr_matvec(v, mean_dot_v, 1.0 / self.n_total))
        if self.diag_shift != 0.0:
            res = res + self.diag_shift * v
        return res


def conjugate_gradient(matrix, b, x0=None, max_iter=100, relative_tolerance=1e-10, absolute_tolerance=0.0):
    """ConjugateGradientSolver (utility/conjugate_gradient_solver.h): returns (x, residual norm, iterations)"""
    b = np.asarray(b, dtype=np.float64)
    x = np.zeros_like(b) if x0 is None else np.array(x0, dtype=np.float64)
    tol_sq = max(relative_tolerance ** 2 * float(np.vdot(b, b)), absolute_tolerance ** 2)
    r = b - matrix * x
    rr = float(np.vdot(r, r))
    if rr <= tol_sq:
        return x, np.sqrt(rr), 0
    p = r.copy()
    it = 0
    for it in range(1, max_iter + 1):
        ap = matrix * p
        pap = float(np.vdot(p, ap))
        if not pap > 0.0:
            break
        alpha = rr / pap
        x += alpha * p
        r -= alpha * ap
        rr_new = float(np.vdot(r, r))
        if rr_new <= tol_sq:
            return x, np.sqrt(rr_new), it
        p = r + (rr_new / rr) * p
        rr = rr_new
    return x, np.sqrt(rr), it
