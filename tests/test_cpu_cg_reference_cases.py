"""ConjugateGradientSolver (utility/conjugate_gradient_solver.h:181-276): every case of the reference's own
tests/test_utility/test_conjugate_gradient_solver.cpp, on the CPU, against BOTH restatements -- the host-vector solver of the product
(`peps_amd/sr.py`, the multi-rank path of the SR solve) and `oracle/sr.py` (the checker of the device-resident solve in
tests/test_gpu_sr.py).  Same matrices, right-hand sides, start vectors, parameters and expectations (converged / iterations / residual /
termination reason / solution) as the reference test, line numbers cited per case."""
import numpy as np
import pytest

from oracle import sr as osr


class _Mat:
    """MySquareMatrix of the reference test (my_vector_matrix.h): operator* on a vector"""

    def __init__(self, a):
        self.a = np.array(a)

    def __mul__(self, v):
        return self.a @ v


class _InfAfter(_Mat):
    """InfInjectingMatrix (:227-246): element 0 of the product is +inf from call `inf_after` + 1 on"""

    def __init__(self, a, inf_after):
        super().__init__(a)
        self.calls, self.inf_after = 0, inf_after

    def __mul__(self, v):
        self.calls += 1
        out = np.array(self.a @ v, dtype=np.float64)
        if self.calls > self.inf_after:
            out[0] = np.inf
        return out


def _solvers():
    from peps_amd import sr

    def host(mat, b, x0, **kw):
        return sr.conjugate_gradient(mat, np.array(b), np.array(x0), full_output=True, **kw)

    def oracle(mat, b, x0, **kw):
        return osr.conjugate_gradient_full(lambda v: mat * v, np.array(b), np.array(x0), **kw)
    return [("peps_amd.sr", host), ("oracle.sr", oracle)]


SOLVERS = _solvers()
IDS = [s[0] for s in SOLVERS]
A2 = [[4.0, 1.0], [1.0, 3.0]]
A2B = [[1.7014087728892546, -1.9258430571407281], [-1.9258430571407281, 2.7386516770021552]]
X0B = [-1.3252085986071422, 0.84568824604762072]


@pytest.mark.parametrize("name,solve", SOLVERS, ids=IDS)
def test_no_parallel_real_and_complex_systems(name, solve):
    """TestPlainCGSolver.NoParallel (:26-72): max_iter 100, relative_tolerance 1e-16"""
    a = [[1.0, 2.0, 3.0], [2.0, 5.0, 7.0], [3.0, 7.0, 15.0]]
    z = [[4.3, 1 + 2j, -3j], [1 - 2j, 5.0, 2 + 1j], [3j, 2 - 1j, 6.0]]
    cases = [(a, [11.0, 12.0, 13.0], [-1.0, 1.0, 0.0], [33.0, -8.0, -2.0]),
             (np.array(a, dtype=complex), np.array([11.0, 12.0, 13.0], dtype=complex), np.array([-1.0, 1.0, 0.0], dtype=complex), [33.0, -8.0, -2.0]),
             (z, [9.3 - 7.35j, -5j, 7 + 7j], [0j, 0j, 0j], [1 + 0.5j, -1 - 1.5j, 2 + 1j])]
    for mat, b, x0, want in cases:
        x, res, it, why = solve(_Mat(mat), b, x0, max_iter=100, relative_tolerance=1e-16)
        assert why == osr.K_CONVERGED and it < 100 and res < 1e-6                          # :33-36
        assert np.sum(np.abs(x - np.array(want)) ** 2) < 1e-13                             # :37-38


@pytest.mark.parametrize("name,solve", SOLVERS, ids=IDS)
def test_relative_tolerance_is_scale_independent(name, solve):
    """RelativeToleranceScaleIndependence (:74-104)"""
    for scale in (1.0, 1e6):
        x, res, it, why = solve(_Mat(A2), [scale, scale], [0.0, 0.0], max_iter=100, relative_tolerance=1e-10)
        want = np.array([2.0, 3.0]) * scale / 11.0
        assert why == osr.K_CONVERGED and it < 100
        assert np.sum((x - want) ** 2) / np.sum(want ** 2) < 1e-13


@pytest.mark.parametrize("name,solve", SOLVERS, ids=IDS)
def test_residue_restart_interval(name, solve):
    """SerialResidueRestart (:106-120): residual_recompute_interval = 5"""
    x, res, it, why = solve(_Mat(A2), [5.0, 4.0], [0.0, 0.0], max_iter=100, relative_tolerance=1e-16, residual_recompute_interval=5)
    assert why == osr.K_CONVERGED and np.sum((x - 1.0) ** 2) < 1e-13


@pytest.mark.parametrize("name,solve", SOLVERS, ids=IDS)
@pytest.mark.parametrize("b", [[0.0, 0.0], [1e-300, -1e-300]])
def test_zero_and_tiny_rhs(name, solve, b):
    """ZeroRhs / TinyRhs, relative only (:122-148): not converged; with an explicit absolute tolerance (:150-188): converged to zero"""
    _, _, _, why = solve(_Mat(A2B), b, X0B, max_iter=200, relative_tolerance=1e-10)
    assert why != osr.K_CONVERGED
    x, _, _, why = solve(_Mat(A2B), b, X0B, max_iter=200, relative_tolerance=1e-10, absolute_tolerance=1e-10)
    assert why == osr.K_CONVERGED and np.sum(x ** 2) < 1e-20


@pytest.mark.parametrize("name,solve", SOLVERS, ids=IDS)
def test_breakdown_and_non_convergence_are_reported(name, solve):
    """BreakdownDetection (:190-201): a singular matrix does not crash and is not reported converged;
    NonConvergenceReported (:203-213): max_iter = 1 leaves iterations == 1 and a positive residual"""
    _, _, _, why = solve(_Mat([[1.0, 0.0], [0.0, 0.0]]), [1.0, 1.0], [0.0, 0.0], max_iter=100, relative_tolerance=1e-10)
    assert why != osr.K_CONVERGED
    _, res, it, why = solve(_Mat(A2), [5.0, 4.0], [0.0, 0.0], max_iter=1, relative_tolerance=1e-30)
    assert why != osr.K_CONVERGED and it == 1 and res > 0.0


@pytest.mark.filterwarnings("ignore:invalid value encountered")        # 0 * inf in the residual update IS the case under test
@pytest.mark.parametrize("name,solve", SOLVERS, ids=IDS)
def test_termination_reasons(name, solve):
    """IndefiniteMatrixDetection (:219-233), NumericalBreakdownDetection (:248-268), StagnationDetection (:270-293)"""
    _, _, _, why = solve(_Mat(np.diag([1.0, -1.0, 1.0])), [0.0, 1.0, 0.0], [0.0, 0.0, 0.0], max_iter=100, relative_tolerance=1e-10)
    assert why == osr.K_INDEFINITE
    _, _, _, why = solve(_InfAfter(np.diag([2.0, 3.0, 5.0]), 1), [1.0, 2.0, 3.0], [0.0, 0.0, 0.0], max_iter=100, relative_tolerance=1e-10)
    assert why == osr.K_BREAKDOWN
    ev = np.array([10.0 ** (-i * 15.0 / 9.0) for i in range(10)])
    _, _, _, why = solve(_Mat(np.diag(ev)), ev, np.zeros(10), max_iter=10000, relative_tolerance=1e-30)
    assert why == osr.K_STAGNATED


@pytest.mark.parametrize("name,solve", SOLVERS, ids=IDS)
def test_orthogonality_restart_converges(name, solve):
    """OrthogonalityRestartConverges (:295-316): orthogonality_threshold = 0.01"""
    x, _, _, why = solve(_Mat(np.diag([2.0, 3.0, 5.0, 7.0])), [1.0, 2.0, 3.0, 4.0], [0.0, 0.0, 0.0, 0.0], max_iter=100,
                         relative_tolerance=1e-10, orthogonality_threshold=0.01)
    assert why == osr.K_CONVERGED and np.sum((x - np.array([0.5, 2.0 / 3.0, 0.6, 4.0 / 7.0])) ** 2) < 1e-13
