"""Stochastic reconfiguration -- oracle restatement (TEST INFRASTRUCTURE ONLY).

SRSMatrix::operator* (optimizer/stochastic_reconfiguration_smatrix.h:37-99, centred scalar projection) and the
conjugate-gradient solver (utility/conjugate_gradient_solver.h: standard CG, termination
||r||^2 <= max(rel_tol^2 ||b||^2, abs_tol^2)), on flat float64 vectors."""
import numpy as np


class SRSMatrix:
    def __init__(self, ostar_samples, ostar_mean=None, world_size=1, diag_shift=0.0):
        self.samples = [np.asarray(o, dtype=np.float64).ravel() for o in ostar_samples]
        self.mean = None if ostar_mean is None else np.asarray(ostar_mean, dtype=np.float64).ravel()
        self.world_size, self.diag_shift = world_size, diag_shift

    def __mul__(self, v0):
        v = np.asarray(v0, dtype=np.float64).ravel()
        mean_dot_v = 0.0 if self.mean is None else float(self.mean @ v)          # :38-42
        res = np.zeros_like(v)
        for o in self.samples:                                                      # :49-54
            res += (float(o @ v) - mean_dot_v) * o
        res *= 1.0 / (len(self.samples) * self.world_size)                          # :55
        if self.mean is not None and self.diag_shift != 0.0:                        # :75-77
            res += self.diag_shift * v
        return res


def conjugate_gradient(matvec, b, x0, max_iter=100, relative_tolerance=1e-10, absolute_tolerance=0.0):
    b = np.asarray(b, dtype=np.float64).ravel()
    x = np.asarray(x0, dtype=np.float64).ravel().copy()
    tol_sq = max(relative_tolerance ** 2 * float(b @ b), absolute_tolerance ** 2)
    r = b - matvec(x)
    rr = float(r @ r)
    if rr <= tol_sq:
        return x, np.sqrt(rr), 0
    p = r.copy()
    for it in range(1, max_iter + 1):
        ap = matvec(p)
        pap = float(p @ ap)
        if not pap > 0.0:
            break                                   # kIndefiniteMatrix
        alpha = rr / pap
        x += alpha * p
        r -= alpha * ap
        rr_new = float(r @ r)
        if rr_new <= tol_sq:
            return x, np.sqrt(rr_new), it
        p = r + (rr_new / rr) * p
        rr = rr_new
    return x, np.sqrt(rr), it
