"""Stochastic reconfiguration -- oracle restatement (TEST INFRASTRUCTURE ONLY).

SRSMatrix::operator* (optimizer/stochastic_reconfiguration_smatrix.h:37-99, centred scalar projection) and the
conjugate-gradient solver (utility/conjugate_gradient_solver.h: standard CG, termination
||r||^2 <= max(rel_tol^2 ||b||^2, abs_tol^2)), on flat float64 vectors."""
import numpy as np


class SRSMatrix:
    """TenElemT = double or complex: `a * b` of two SplitIndexTPS is the positive-definite pairing sum conj(a) b
    (split_index_tps.h:370-377, qlten::QuasiInnerProduct) = np.vdot."""

    def __init__(self, ostar_samples, ostar_mean=None, world_size=1, diag_shift=0.0):
        cplx = any(np.iscomplexobj(o) for o in ostar_samples)
        self.dtype = np.complex128 if cplx else np.float64
        self.samples = [np.asarray(o, dtype=self.dtype).ravel() for o in ostar_samples]
        self.mean = None if ostar_mean is None else np.asarray(ostar_mean, dtype=self.dtype).ravel()
        self.world_size, self.diag_shift = world_size, diag_shift

    def __mul__(self, v0):
        v = np.asarray(v0, dtype=self.dtype).ravel()
        mean_dot_v = 0.0 if self.mean is None else np.vdot(self.mean, v)         # :38-42
        res = np.zeros_like(v)
        for o in self.samples:                                                      # :49-54
            res += (np.vdot(o, v) - mean_dot_v) * o
        res *= 1.0 / (len(self.samples) * self.world_size)                          # :55
        if self.mean is not None and self.diag_shift != 0.0:                        # :75-77
            res += self.diag_shift * v
        return res


# CGTerminationReason (utility/conjugate_gradient_solver.h:74-80)
K_CONVERGED, K_MAX_ITERATIONS, K_INDEFINITE, K_BREAKDOWN, K_STAGNATED = 0, 1, 2, 3, 4


def conjugate_gradient_full(matvec, b, x0, max_iter=100, relative_tolerance=1e-4, absolute_tolerance=0.0,
                            residual_recompute_interval=20, orthogonality_threshold=0.5):
    """ConjugateGradientSolver (utility/conjugate_gradient_solver.h:181-276) with every branch of the reference:
    indefinite-matrix exit, stagnation detection, periodic residual recomputation, NaN/Inf exits, best-iterate
    tracking, orthogonality-based restart.  Parameter defaults = ConjugateGradientParams (optimizer_params.h:50-57).
    Returns (x, residual_norm, iterations, reason)."""
    cplx = np.iscomplexobj(b) or np.iscomplexobj(x0)        # TenElemT = QLTEN_Complex: a * b = sum conj(a) b, NormSquare = sum |a|^2
    dt = np.complex128 if cplx else np.float64
    b = np.asarray(b, dtype=dt).ravel()
    x0 = np.asarray(x0, dtype=dt).ravel()
    eps = np.finfo(np.float64).eps
    nsq = lambda v: float(np.vdot(v, v).real)
    tol_sq = max(relative_tolerance ** 2 * nsq(b), absolute_tolerance ** 2)
    r = b - matvec(x0)
    rr = nsq(r)
    if rr <= tol_sq:
        return x0.copy(), np.sqrt(rr), 0, K_CONVERGED
    p, x, best_x, best_rr = r.copy(), x0.copy(), x0.copy(), rr
    r_prev = r.copy()
    rkp1 = rr
    stagnation = 0
    for k in range(max_iter):
        rk = rkp1
        ap = matvec(p)
        pap = np.vdot(p, ap) if cplx else float(p @ ap)
        # detail::pap_is_valid (:142-148): real: pap > 0; complex: Re > 0 and |Im| < 1e-10
        ok = (pap.real > 0.0 and abs(pap.imag) < 1e-10) if cplx else (pap > 0.0)    # (+inf passes, NaN does not: the reference's plain comparisons)
        if not ok:
            return best_x, np.sqrt(best_rr), k, K_INDEFINITE
        alpha = rk / pap
        x = x + alpha * p
        if abs(alpha) ** 2 * nsq(p) < eps * eps * nsq(x):                # :227-236
            stagnation += 1
            if stagnation >= 3:
                return best_x, np.sqrt(best_rr), k + 1, K_STAGNATED
        else:
            stagnation = 0
        if residual_recompute_interval > 0 and (k % residual_recompute_interval) == residual_recompute_interval - 1:
            r = b - matvec(x)                                              # :238-239
        else:
            r = r - alpha * ap
        rkp1 = nsq(r)
        if not np.isfinite(rkp1):
            return best_x, np.sqrt(best_rr), k + 1, K_BREAKDOWN
        if rkp1 < best_rr:
            best_x, best_rr = x.copy(), rkp1
        if rkp1 <= tol_sq:
            return x, np.sqrt(rkp1), k + 1, K_CONVERGED
        if k > 0 and abs(np.vdot(r_prev, r).real) > orthogonality_threshold * rkp1:   # :259-266 (real part)
            p = r.copy()
            r_prev = r.copy()
            continue
        r_prev = r.copy()
        beta = rkp1 / rk
        if not np.isfinite(beta):
            return best_x, np.sqrt(best_rr), k + 1, K_BREAKDOWN
        p = r + beta * p
    return best_x, np.sqrt(best_rr), max_iter, K_MAX_ITERATIONS


def pseudo_inverse_cutoff(eigenvalues, r_pinv, a_pinv, soft_cutoff):
    """ApplyPseudoInverseCutoff (optimizer/minsr_eigensolve.h:44-78)."""
    ev = np.asarray(eigenvalues, dtype=np.float64)
    cutoff = r_pinv * (np.max(np.abs(ev)) if len(ev) else 0.0) + a_pinv
    out = np.zeros_like(ev)
    if soft_cutoff:
        den = ev ** 6 + cutoff ** 6
        nz = den != 0.0
        out[nz] = ev[nz] ** 5 / den[nz]
    else:
        keep = np.abs(ev) > cutoff
        out[keep] = 1.0 / ev[keep]
    return out


def minsr_tmatrix(ostar_samples):
    """MinSRTMatrix::Construct (optimizer/minsr_tmatrix.h:53-147): raw Gram ip_ij = O*_i * O*_j = sum conj(O*_i) O*_j, four-term
    centering (entry - m_i - conj(m_j) + c, :141-147), 1/Ns.  TenElemT = double or complex."""
    cplx = any(np.iscomplexobj(x) for x in ostar_samples)
    o = np.stack([np.asarray(x, dtype=np.complex128 if cplx else np.float64).ravel() for x in ostar_samples])
    ns = o.shape[0]
    g = o.conj() @ o.T
    m = g.sum(axis=1) / ns
    c = m.sum() / ns
    return (g - m[:, None] - m.conj()[None, :] + c) / ns


def minsr_direction(ostar_samples, ostar_mean, energy_samples, energy, r_pinv=1e-12, a_pinv=0.0, soft_cutoff=True):
    """Optimizer::CalculateMinSRDirection_ (optimizer/optimizer_impl.h:1126-1215) for one rank (or all ranks'
    samples concatenated): epsilon_bar = conj(E_loc - E) / Ns (:1139-1146), T, replicated eigensolve with pseudo-inverse cutoff
    (ReplicatedEigenSolveReal / Complex, minsr_eigensolve.h:101-237: y = Z lambda+ Z^H rhs), back-substitution
    delta = sum_i y_i O*_i - (sum_i y_i) Ostar_mean.  Returns (delta_theta, norm)."""
    cplx = any(np.iscomplexobj(x) for x in ostar_samples) or np.iscomplexobj(energy_samples) or np.iscomplexobj(ostar_mean)
    dt = np.complex128 if cplx else np.float64
    o = np.stack([np.asarray(x, dtype=dt).ravel() for x in ostar_samples])
    ns = o.shape[0]
    eps_bar = np.conj(np.asarray(energy_samples, dtype=dt) - energy) / ns
    t = minsr_tmatrix(ostar_samples)
    ev, z = np.linalg.eigh(t)
    y = z @ (pseudo_inverse_cutoff(ev, r_pinv, a_pinv, soft_cutoff) * (z.conj().T @ eps_bar))
    delta = y @ o - y.sum() * np.asarray(ostar_mean, dtype=dt).ravel()
    return delta, float(np.linalg.norm(delta))


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
