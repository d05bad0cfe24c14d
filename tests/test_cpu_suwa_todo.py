"""Suwa-Todo state update (`SuwaTodoStateUpdate`, suwa_todo_update.h:53-112) -- the cases of the reference's own
tests/test_monte_carlo_tools/test_suwa_todo_update.cpp, run on the CPU against BOTH restatements: the C++ of the host layer
(`qlpeps_gpu.h`, through `pepshost_suwa_todo_chain`: host code only, no device call) and `oracle/vmc.py`.

  * BasicFunctionality (:31-58): a single state stays, zero-weight states are never selected, one non-zero weight is absorbing;
  * SingleModeMarkovChainDistribution (:60-104): the stationary distribution of the chain equals the weights within 3 standard errors
    for the reference's three weight vectors (incl. a weight of 1e-30 and two equal maxima);
  * PottsModelTest (:258-291): q = 3 Potts model on the periodic 3 x 3 lattice swept with the update, against the exact energies the
    reference quotes (0.32711538 at 0.7 T_c, 7.16804143 at 1.5 T_c; re-derived here by enumeration);
  * and what the reference cannot test: the two restatements draw the same deviates -- identical chains from the same std::mt19937 seed."""
import itertools
import math

import numpy as np
import pytest

from oracle import vmc

CASES = [([1, 0.5, 0.3, 0.01, 0.06, 2], 0), ([1, 0.3, 1e-30], 0), ([9.6, 9.6, 1], 1)]       # test_suwa_todo_update.cpp:100-103


def _host():
    from peps_amd import hostapi
    return hostapi


def _oracle_chain(init, weights, seed, n):
    rng = vmc.StdMT19937(seed)
    st, out = init, np.empty(n, dtype=np.int64)
    for t in range(n):
        st = vmc.suwa_todo_state_update(st, weights, rng.u_longdouble)
        out[t] = st
    return out


def test_basic_functionality_both_restatements():
    host = _host()
    for chain in (lambda i, w, s, n: host.suwa_todo_chain(i, w, s, n), _oracle_chain):
        assert np.all(chain(0, [1.0], 5, 50) == 0)                               # :36-40
        assert not np.any(chain(0, [1.0, 0.0, 1.0], 6, 100) == 1)                # :43-49
        assert np.all(chain(1, [0.0, 1.0, 0.0], 7, 100) == 1)                    # :52-58


def test_host_chain_rejects_bad_arguments():
    host = _host()
    for init, w in ((3, [1.0, 1.0]), (0, [0.0, 1.0]), (0, [1.0, -0.5])):         # the reference's debug assertions (:59-67)
        with pytest.raises(ValueError):
            host.suwa_todo_chain(init, w, 1, 4)


@pytest.mark.parametrize("weights,init", CASES)
def test_identical_chains_cpp_and_oracle(weights, init):
    """deviate for deviate: std::mt19937 + uniform_real_distribution<long double> in C++, the restated generator in the oracle"""
    host = _host()
    for seed in (0, 20240115):
        assert np.array_equal(host.suwa_todo_chain(init, weights, seed, 4000), _oracle_chain(init, weights, seed, 4000))


@pytest.mark.parametrize("weights,init", CASES)
def test_single_mode_markov_chain_distribution(weights, init):
    """:60-98 with the reference's 5e5 iterations (C++ chain) and 5e4 (oracle): frequencies within 3 standard errors of w_i / sum w"""
    host = _host()
    pi = np.asarray(weights, dtype=np.float64) / np.sum(weights)
    for chain, n in ((host.suwa_todo_chain(init, weights, 11, 500000), 500000), (_oracle_chain(init, weights, 12, 50000), 50000)):
        freq = np.bincount(chain, minlength=len(weights)) / n
        tol = 3.0 * np.sqrt(pi * (1.0 - pi) / n)
        assert np.all(np.abs(freq - pi) <= tol + 1e-15), (freq, pi, tol)


def _potts_exact_energy(size, q, temperature):
    """CalculateExactEnergy (:170-216): E = sum over bonds [1 - delta], periodic, by enumeration"""
    n = size * size
    idx = np.arange(n).reshape(size, size)
    right, down = np.roll(idx, -1, axis=1).ravel(), np.roll(idx, -1, axis=0).ravel()
    cfg = np.array(list(itertools.product(range(q), repeat=n - 1)), dtype=np.int8)
    cfg = np.concatenate([cfg, np.zeros((cfg.shape[0], 1), dtype=np.int8)], axis=1)        # last spin fixed by the Z_q symmetry
    e = 2 * n - (cfg == cfg[:, right]).sum(axis=1) - (cfg == cfg[:, down]).sum(axis=1)
    wgt = np.exp(-e / temperature)
    return float((e * wgt).sum() / wgt.sum())


def _potts_mc_energy(size, q, temperature, seed, therm, samples):
    """PottsModel::MonteCarloSweep (:126-140): every site in turn, weights exp(-E_local / T) over its q states, Suwa-Todo update"""
    rng = vmc.StdMT19937(seed)
    spins = np.array([rng.raw() % q for _ in range(size * size)], dtype=np.int64).reshape(size, size)
    boltz = [math.exp(k / temperature) for k in range(5)]                        # exp(+neighbours equal / T)
    es = np.empty(samples)
    for sweep in range(therm + samples):
        for x in range(size):
            for y in range(size):
                nb = (spins[(x - 1) % size, y], spins[(x + 1) % size, y], spins[x, (y - 1) % size], spins[x, (y + 1) % size])
                w = [boltz[sum(1 for v in nb if v == j)] for j in range(q)]
                spins[x, y] = vmc.suwa_todo_state_update(int(spins[x, y]), w, rng.u_longdouble)
        if sweep >= therm:
            es[sweep - therm] = 2 * size * size - (spins == np.roll(spins, -1, axis=0)).sum() - (spins == np.roll(spins, -1, axis=1)).sum()
    return es


@pytest.mark.parametrize("ratio,e_ref", [(0.7, 0.32711538), (1.5, 7.16804143)])      # test_suwa_todo_update.cpp:8-12, :258-291
def test_potts_3x3_energy_against_the_reference_constants(ratio, e_ref):
    t_c = math.log(1.0 + math.sqrt(3.0))                                           # :263 (the value the reference computes)
    temperature = ratio * t_c
    assert abs(_potts_exact_energy(3, 3, temperature) - e_ref) < 5e-8              # the quoted constants, by enumeration
    es = _potts_mc_energy(3, 3, temperature, seed=7, therm=1500, samples=12000)
    bins = es[: (len(es) // 30) * 30].reshape(30, -1).mean(axis=1)                 # GetEnergyErr(30) (:149-168)
    err = bins.std(ddof=1) / math.sqrt(30)
    assert abs(es.mean() - e_ref) < 3.0 * err + 1e-12, (es.mean(), e_ref, err)
