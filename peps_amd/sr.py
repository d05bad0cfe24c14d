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
        tot_sum = ctx.sr_sum()
        if np.iscomplexobj(tot_sum) and dist is not None:
            tot_sum = self._allreduce(np.ascontiguousarray(tot_sum).view(np.float64)).view(np.complex128)
        else:
            tot_sum = self._allreduce(tot_sum)
        self.mean = tot_sum / self.n_total          # Ostar_mean

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
        cplx = np.iscomplexobj(self.mean)               # complex contexts: a * b = sum conj(a) b (np.vdot), complex mean . v
        v = np.ascontiguousarray(v, dtype=np.complex128 if cplx else np.float64)
        mdv = np.vdot(self.mean, v)
        res = self.ctx.sr_matvec(v, mdv if cplx else float(mdv), 1.0 / self.n_total)
        if cplx and self.dist is not None:               # (torch has no complex all-reduce on every backend: as interleaved pairs)
            res = self._allreduce(np.ascontiguousarray(res).view(np.float64)).view(np.complex128)
        else:
            res = self._allreduce(res)
        if self.diag_shift != 0.0:
            res = res + self.diag_shift * v
        return res


K_CONVERGED, K_MAX_ITERATIONS, K_INDEFINITE, K_BREAKDOWN, K_STAGNATED = 0, 1, 2, 3, 4    # CGTerminationReason


def conjugate_gradient(matrix, b, x0=None, max_iter=100, relative_tolerance=1e-4, absolute_tolerance=0.0,
                       residual_recompute_interval=20, orthogonality_threshold=0.5, full_output=False):
    """ConjugateGradientSolver (utility/conjugate_gradient_solver.h:181-276) with host vectors: the multi-rank solve
    (`matrix` all-reduces every product).  Same branches as the reference and as the device-resident
    pepsgpu_sr_cg_solve: indefinite exit, stagnation, periodic residual recomputation, NaN/Inf exits, best iterate,
    orthogonality restart.  Returns (x, residual norm, iterations[, reason])."""
    cplx = np.iscomplexobj(b) or (x0 is not None and np.iscomplexobj(x0))      # TenElemT = QLTEN_Complex
    dt = np.complex128 if cplx else np.float64
    b = np.asarray(b, dtype=dt)
    x0 = np.zeros_like(b) if x0 is None else np.array(x0, dtype=dt)
    eps = np.finfo(np.float64).eps
    nsq = lambda v: float(np.vdot(v, v).real)

    def ret(x, rr, it, why):
        return (x, float(np.sqrt(rr)), it, why) if full_output else (x, float(np.sqrt(rr)), it)

    tol_sq = max(relative_tolerance ** 2 * nsq(b), absolute_tolerance ** 2)
    r = b - matrix * x0
    rr = nsq(r)
    if rr <= tol_sq:
        return ret(x0, rr, 0, K_CONVERGED)
    p, x, best_x, best = r.copy(), x0.copy(), x0.copy(), rr
    r_prev, rkp1, stagnation = r.copy(), rr, 0
    for k in range(max_iter):
        rk = rkp1
        ap = matrix * p
        pap = np.vdot(p, ap)
        # detail::pap_is_valid (:142-148): real pap > 0; complex Re > 0 and |Im| < 1e-10
        ok = (pap.real > 0.0 and abs(pap.imag) < 1e-10) if cplx else (pap > 0.0)    # (+inf passes, NaN does not: the reference's plain comparisons)
        if not ok:
            return ret(best_x, best, k, K_INDEFINITE)
        alpha = rk / (pap if cplx else float(pap))
        x = x + alpha * p
        if abs(alpha) ** 2 * nsq(p) < eps * eps * nsq(x):
            stagnation += 1
            if stagnation >= 3:
                return ret(best_x, best, k + 1, K_STAGNATED)
        else:
            stagnation = 0
        if residual_recompute_interval > 0 and (k % residual_recompute_interval) == residual_recompute_interval - 1:
            r = b - matrix * x
        else:
            r = r - alpha * ap
        rkp1 = nsq(r)
        if not np.isfinite(rkp1):
            return ret(best_x, best, k + 1, K_BREAKDOWN)
        if rkp1 < best:
            best_x, best = x.copy(), rkp1
        if rkp1 <= tol_sq:
            return ret(x, rkp1, k + 1, K_CONVERGED)
        if k > 0 and abs(np.vdot(r_prev, r).real) > orthogonality_threshold * rkp1:
            p, r_prev = r.copy(), r.copy()
            continue
        r_prev = r.copy()
        beta = rkp1 / rk
        if not np.isfinite(beta):
            return ret(best_x, best, k + 1, K_BREAKDOWN)
        p = r + beta * p
    return ret(best_x, best, max_iter, K_MAX_ITERATIONS)


def natural_gradient(ctx, gradient, diag_shift=0.0, dist=None, x0=None, **cg_params):
    """Optimizer::CalculateNaturalGradient (optimizer_impl.h): solve (S + diag_shift) x = gradient.  One rank: every CG
    vector stays on the device (pepsgpu_sr_cg_solve).  Several ranks: host vectors, one all-reduce per product.
    Returns (x, residual norm, iterations, reason)."""
    if dist is None:
        return ctx.sr_cg_solve(gradient, x0, diag_shift, **cg_params)
    return conjugate_gradient(DeviceSRSMatrix(ctx, diag_shift, dist), gradient, x0, full_output=True, **cg_params)


# ---------------------------------------------------------------------------------------------
# MinSR (Chen & Heyl 2024; optimizer/minsr_tmatrix.h, minsr_eigensolve.h, optimizer_impl.h:1126-1215)
def pseudo_inverse_cutoff(eigenvalues, r_pinv, a_pinv, soft_cutoff):
    """ApplyPseudoInverseCutoff (minsr_eigensolve.h:44-78)"""
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


class TorchRing:
    """The communication MinSRTMatrix::Construct needs, over torch.distributed (backend "nccl" = RCCL on the GPU box,
    "gloo" in the CPU tests): ring send/recv of sample batches, all-gather of small host vectors, all-reduce."""

    def __init__(self, dist):
        self.dist = dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.device = "cuda" if dist.get_backend() == "nccl" else "cpu"

    def ring_exchange(self, send_tensors, recv_tensors):
        """send to rank+1, receive from rank-1; rank P-1 sends first (minsr_tmatrix.h:96-104 breaks the cycle that way)"""
        nxt, prv = (self.rank + 1) % self.world, (self.rank - 1) % self.world
        ops = []
        for t in send_tensors:
            ops.append(self.dist.P2POp(self.dist.isend, t, nxt))
        for t in recv_tensors:
            ops.append(self.dist.P2POp(self.dist.irecv, t, prv))
        for req in self.dist.batch_isend_irecv(ops):
            req.wait()

    def allgather(self, arr):
        import torch
        if np.iscomplexobj(arr):          # complex values travel as interleaved (re, im) pairs (MPI_CXX_DOUBLE_COMPLEX in the reference)
            return self.allgather(np.ascontiguousarray(arr, dtype=np.complex128).view(np.float64)).view(np.complex128)
        t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float64)).to(self.device)
        out = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return np.concatenate([o.cpu().numpy() for o in out])

    def allreduce(self, arr):
        import torch
        if np.iscomplexobj(arr):
            return self.allreduce(np.ascontiguousarray(arr, dtype=np.complex128).view(np.float64)).view(np.complex128)
        t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float64)).to(self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return t.cpu().numpy()


class DeviceSampleBatch:
    """The local O* samples of a context as the backend of minsr_direction: Gram blocks on the device, batches
    exchanged as torch tensors whose device pointers go straight back into the library."""

    def __init__(self, ctx):
        self.ctx = ctx
        self.n = ctx.sr_count()

    def gram_local(self):
        return self.ctx.sr_gram()

    def export(self, device):
        import torch
        from . import capi
        sites = self.ctx.rows * self.ctx.cols
        dt = torch.float32 if self.ctx.dtype == capi.F32 else torch.complex128 if self.ctx.dtype == capi.C128 else torch.float64
        o = torch.empty((self.n, sites, self.ctx.D ** 4), dtype=dt, device=device)
        c = torch.empty((self.n, sites), dtype=torch.int32, device=device)
        self.ctx.sr_copy_samples(o.data_ptr(), c.data_ptr())
        return [o, c]

    def gram_with(self, batch):
        import torch
        torch.cuda.synchronize()
        return self.ctx.sr_gram(batch[0].data_ptr(), batch[1].data_ptr(), int(batch[0].shape[0]))

    def weighted_sum(self, y):
        return self.ctx.sr_weighted_sum(y)

    def sample_sum(self):
        return self.ctx.sr_sum()


def minsr_direction(batch, energy_samples, energy, r_pinv=1e-12, a_pinv=0.0, soft_cutoff=True, ring=None):
    """Optimizer::CalculateMinSRDirection_ (optimizer_impl.h:1126-1215): epsilon_bar -> T (ring exchange + four-term
    centering, MinSRTMatrix::Construct) -> replicated eigensolve with pseudo-inverse cutoff (Path B,
    minsr_eigensolve.h:101-155) -> back-substitution delta = sum_i y_i O*_i - (sum_i y_i) Ostar_mean.
    `batch`: the local samples (DeviceSampleBatch); `ring`: TorchRing or None for one rank; every rank must hold the
    same number of samples (as the reference, ns_global = ns_local * world).  Returns (delta_theta, norm), valid on
    every rank."""
    ns_local = batch.n
    world = 1 if ring is None else ring.world
    rank = 0 if ring is None else ring.rank
    ns = ns_local * world
    g0 = batch.gram_local()                                                          # round 0
    cplx = np.iscomplexobj(g0) or np.iscomplexobj(energy_samples)                    # TenElemT = QLTEN_Complex
    dt = np.complex128 if cplx else np.float64
    eps_local = np.conj(np.asarray(energy_samples, dtype=dt) - energy) / ns          # conj(Delta E) / Ns (optimizer_impl.h:1139-1146)
    assert eps_local.size == ns_local
    rows = np.zeros((ns_local, ns), dtype=dt)
    rows[:, rank * ns_local:(rank + 1) * ns_local] = g0
    if world > 1:
        cur = batch.export(ring.device)
        for rnd in range(1, world):                                                  # rounds 1..P-1
            nxt = [t.new_empty(t.shape) for t in cur]
            ring.ring_exchange(cur, nxt)
            src = (rank - rnd) % world
            rows[:, src * ns_local:(src + 1) * ns_local] = batch.gram_with(nxt)
            cur = nxt
    m_local = rows.sum(axis=1) / ns                                                  # four-term centering
    if world > 1:
        all_m = ring.allgather(m_local)
        eps_bar = ring.allgather(eps_local)
    else:
        all_m, eps_bar = m_local, eps_local
    c = all_m.sum() / ns
    rows = (rows - m_local[:, None] - np.conj(all_m)[None, :] + c) / ns              # entry - m_i - conj(m_j) + c (minsr_tmatrix.h:141-147)
    t_full = rows if world == 1 else ring.allgather(rows.ravel()).reshape(ns, ns)
    ev, z = np.linalg.eigh(t_full)                                                   # dsyev / zheev, replicated
    y = z @ (pseudo_inverse_cutoff(ev, r_pinv, a_pinv, soft_cutoff) * (z.conj().T @ eps_bar))
    y_local = y[rank * ns_local:(rank + 1) * ns_local]
    acc = np.concatenate([np.asarray(batch.weighted_sum(y_local)).ravel(), np.asarray(batch.sample_sum()).ravel()])
    if world > 1:
        acc = ring.allreduce(acc)
    half = acc.size // 2
    delta = acc[:half] - y.sum() * acc[half:] / ns
    return delta, float(np.linalg.norm(delta))
