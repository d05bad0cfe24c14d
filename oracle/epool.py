"""Local energies of a list of configurations from the float64 NumPy oracle, fanned over worker PROCESSES (one configuration
per task), started as a child interpreter that never touches the GPU -- bench.py's `energy_parity` sample (n >= 8 inside the
time one configuration takes) and nothing else.

TEST INFRASTRUCTURE ONLY.  Restates SquareNNNModelEnergySolver::CalEnergyAndHolesImpl
(include/qlpeps/algorithm/vmc_update/model_solvers/base/square_nnn_energy_solver.h:104-197) with the XXZ bond energy of
square_spin_onehalf_xxz_obc.h:72-104 through oracle/vmc.py.
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def start(flat, cfgs, chi, params=(1.0, 1.0, 0.0), nprocs=None, blas_threads=8):
    """non-blocking: returns a handle for collect()"""
    td = tempfile.mkdtemp(prefix="epool_")
    job = os.path.join(td, "job.npz")
    cfgs = np.ascontiguousarray(cfgs, dtype=np.int32)
    np.savez(job, flat=np.ascontiguousarray(flat, dtype=np.float64), cfgs=cfgs, chi=int(chi), params=np.array(params, dtype=np.float64),
             nprocs=int(nprocs or len(cfgs)))
    env = dict(os.environ, OPENBLAS_NUM_THREADS=str(blas_threads), OMP_NUM_THREADS=str(blas_threads), MKL_NUM_THREADS=str(blas_threads))
    p = subprocess.Popen([sys.executable, "-m", "oracle.epool", job], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env,
                         cwd=os.path.dirname(_HERE))
    return {"proc": p, "job": job, "dir": td, "n": len(cfgs)}


def collect(h, timeout=None):
    """(energies [n], amplitudes [n], wall seconds) or raises"""
    import shutil
    try:
        out, err = h["proc"].communicate(timeout=timeout)
        if h["proc"].returncode != 0:
            raise RuntimeError("oracle.epool child failed: " + out[-1000:] + err[-2000:])
        res = np.load(h["job"] + ".out.npz")
        return res["e"], res["a"], float(res["seconds"])
    except subprocess.TimeoutExpired:
        h["proc"].kill()
        raise
    finally:
        shutil.rmtree(h["dir"], ignore_errors=True)


def _one(args):
    flat, cfg, chi, params = args
    from peps_amd import synthetic          # host-side NumPy helpers only (no device, no library)
    from oracle import vmc
    from oracle.bmps import BMPSTruncateParams
    sitps = synthetic.flat_to_sitps(flat)
    tp = BMPSTruncateParams.SVD(chi, chi, 0.0)
    model = vmc.SquareSpinOneHalfXXZModelOBC(*[float(x) for x in params])
    comp = vmc.TPSWaveFunctionComponent(sitps, cfg, tp)
    out = model.CalEnergyAndHoles(sitps, comp, False)
    return float(out[0] if isinstance(out, (tuple, list)) else out), float(comp.amplitude)


def _child_main(job):
    import multiprocessing as mp
    import time
    d = np.load(job)
    flat, cfgs, chi, params, nprocs = d["flat"], d["cfgs"], int(d["chi"]), d["params"], int(d["nprocs"])
    t0 = time.perf_counter()
    ctx = mp.get_context("fork")
    with ctx.Pool(max(1, min(nprocs, len(cfgs)))) as pool:
        res = pool.map(_one, [(flat, c, chi, params) for c in cfgs], chunksize=1)
    np.savez(job + ".out.npz", e=np.array([r[0] for r in res]), a=np.array([r[1] for r in res]), seconds=time.perf_counter() - t0)


if __name__ == "__main__":
    _child_main(sys.argv[1])
