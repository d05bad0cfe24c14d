"""CPU oracle: a float64 NumPy restatement of the QuantumLiquids/PEPS boundary-MPS hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``peps_amd/`` may import this package; only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` do,
and there only as the checker.

Every function cites the reference file:line it restates (paths relative to
``/root/reference``).  The arithmetic of the reference lives in the un-vendored
TensorToolkit (``qlten``) dependency; its semantics (``Contract``, ``QR``, truncating
``SVD``) are restated in ``oracle/tensor.py`` from the call sites.

Parity pinning (see tests/test_oracle_*.py, tests/golden/):
  K1  12x12 critical Ising partition function vs exact transfer matrix
      (tests/test_2d_tn/test_bmps_contractor.cpp:27-126,128-271,472-493), with SVD(10,30,1e-15),
      Variational2Site(10,30,1e-15,1e-14,10) and Variational1Site(10,30,1e-15,1e-14,10) on all 21 routes
  K3  PunchHole . site == Trace, EraseEnvsAfterUpdate + regrow (…:407-470)
  K4  2x2 fixtures, exact-summation energies (tests/test_algorithm/test_exact_summation_evaluator.cpp)
  K5  4x4 D=8 Heisenberg fixture, exact-sum energy / checkerboard amplitude
  K6  ExactSumMeasurerMPI registry of the 2x2 spinless-fermion simple-update state: energy, charge, per-bond
      energies (tests/test_algorithm/test_exact_summation_measurer.cpp:205-240 -> tests/golden/k4_exact_sum_measurer.json)
  K8  MCPEPSMeasurer regression vector of tests/test_model_solvers/test_square_xxz_measurer.cpp:204-381 (96 SpSm_cross values, seed 42,
      SVD(8, 16, 1e-15)): the Monte-Carlo chain, the order-1 rescale, the walker traces -- reproduced to 6e-16
      (tests/test_oracle_measure.py, tests/golden/xxz_spsm_cross_reference_golden.json)
  K9  MCPEPSMeasurer regression energy of the 6x6 fU1 t-J state, tests/test_model_solvers/test_tJ_model_solver.cpp:72-75,233-275
      (-14.74320489110316, seed 42, 20 sweeps): reproduced to 2e-15 (tests/test_oracle_fermion.py) -- the fermionic chain, truncation and signs
  and the reference's unit tests of SuwaTodoStateUpdate, ConjugateGradientSolver and Configuration I/O, case by case
      (tests/test_cpu_suwa_todo.py, test_cpu_cg_reference_cases.py, test_cpu_configuration_reference_cases.py)
The SVD truncation rule for trunc_err > 0 is "parity unpinned" beyond the instance K8 exercises (no reference binary can be produced
here).  Fermions: the Z2-graded algebra of oracle/graded.py is pinned on the reference's 2x2
spinless-fermion known answers (amplitudes, ratios, energies: tests/test_oracle_fermion.py); the
element-wise sign convention of fermionic GRADIENT tensors (CalGTenForFermionicTensors) stays unpinned.
"""
