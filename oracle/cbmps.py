"""ctypes front of oracle/cbmps.c: the plain-C / LAPACK restatement of the reference's CPU path for one
EvaluateAmplitude (bmps_impl.h:756-862, :225-263; wave_function_component.h:187-212), run in the reference's execution
model (independent walkers, one per thread, BLAS threads = 1: monte_carlo_engine.h:97-98).

TEST INFRASTRUCTURE ONLY: the checker beside oracle/bmps.py and the `cpu_baseline` leg of bench.py.
"""
import ctypes as C
import glob
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, "cbmps.c")
LIB = os.path.join(_HERE, "lib", "libcbmps.so")
_lib = None


def build(force=False):
    """gcc -O2 -shared: oracle/lib/libcbmps.so (no BLAS at link time; LAPACK is dlopen'ed from SciPy's OpenBLAS)."""
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        os.makedirs(os.path.dirname(LIB), exist_ok=True)
        subprocess.run(["gcc", "-O2", "-std=c99", "-D_POSIX_C_SOURCE=200809L", "-fPIC", "-shared", "-Wall", "-o", LIB, SRC,
                        "-ldl", "-lpthread", "-lm"], check=True)
    return LIB


def blas_path():
    """the LP64 OpenBLAS inside the SciPy wheel (Fortran-ABI LAPACK: scipy_dgemm_, scipy_dgelqf_, scipy_dorglq_, scipy_dgesdd_)"""
    import scipy
    cands = sorted(glob.glob(os.path.join(os.path.dirname(scipy.__file__), "..", "scipy.libs", "libscipy_openblas-*.so")))
    if not cands:
        raise ImportError("SciPy's bundled OpenBLAS not found (scipy.libs/libscipy_openblas-*.so)")
    return os.path.abspath(cands[0])


def lib():
    global _lib
    if _lib is None:
        build()
        l = C.CDLL(LIB)
        l.cbmps_last_error.restype = C.c_char_p
        l.cbmps_init.argtypes = [C.c_char_p]
        l.cbmps_amplitudes.argtypes = [C.c_int] * 4 + [C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_int32), C.c_int,
                                                       C.POINTER(C.c_double), C.POINTER(C.c_double)]
        if l.cbmps_init(blas_path().encode()) != 0:
            raise ImportError("cbmps_init: %s" % l.cbmps_last_error().decode())
        _lib = l
    return _lib


def amplitudes(sitps_flat, configs, chi, nthreads=1):
    """sitps_flat: [L][L][d][D][D][D][D] float64 (zero padded, leg order L,D,R,U); configs [n][L][L] int32.
    Returns (amplitudes [n], wall seconds)."""
    flat = np.ascontiguousarray(sitps_flat, dtype=np.float64)
    cfg = np.ascontiguousarray(configs, dtype=np.int32)
    L, d, D = flat.shape[0], flat.shape[2], flat.shape[3]
    assert flat.shape == (L, L, d, D, D, D, D) and cfg.shape[1:] == (L, L)
    out = np.zeros(cfg.shape[0], dtype=np.float64)
    sec = C.c_double(0.0)
    rc = lib().cbmps_amplitudes(L, D, d, chi, flat.ctypes.data_as(C.POINTER(C.c_double)), cfg.shape[0],
                                cfg.ctypes.data_as(C.POINTER(C.c_int32)), int(nthreads),
                                out.ctypes.data_as(C.POINTER(C.c_double)), C.byref(sec))
    if rc != 0:
        raise RuntimeError("cbmps_amplitudes: %s" % lib().cbmps_last_error().decode())
    return out, sec.value


# ---- one walker per PROCESS (the reference's MPI model: one rank per core) -------------------------------------------------
# Run as a child program (`python -m oracle.cbmps job.npz`) so that the pool forks from a process that has never touched
# the GPU; bench.py starts it with subprocess and reads the result file.
def _chunk(args):
    flat, cfgs, chi = args
    if len(cfgs) == 0:
        return np.zeros(0), 0.0
    import time
    t0 = time.perf_counter()
    a, _ = amplitudes(flat, cfgs, chi, nthreads=1)
    return a, time.perf_counter() - t0


def amplitudes_multiprocess(sitps_flat, configs, chi, nprocs):
    """configs sharded round-robin over `nprocs` single-threaded worker processes; returns (amplitudes, wall seconds of the
    compute, processes used).  Starts a child interpreter (see above)."""
    d = amplitudes_multiprocess_detail(sitps_flat, configs, chi, nprocs)
    return d["amps"], d["seconds"], d["nprocs"]


def amplitudes_multiprocess_detail(sitps_flat, configs, chi, nprocs):
    """the same, with the busy seconds of every worker process (`proc_seconds`: how evenly the host ran them)"""
    import sys
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        job = os.path.join(td, "job.npz")
        np.savez(job, flat=np.ascontiguousarray(sitps_flat, dtype=np.float64), cfgs=np.ascontiguousarray(configs, dtype=np.int32),
                 chi=int(chi), nprocs=int(nprocs))
        env = dict(os.environ, OPENBLAS_NUM_THREADS="1", OMP_NUM_THREADS="1")
        r = subprocess.run([sys.executable, "-m", "oracle.cbmps", job], capture_output=True, text=True, env=env,
                           cwd=os.path.dirname(_HERE))
        if r.returncode != 0:
            raise RuntimeError("oracle.cbmps child failed: " + r.stdout[-2000:] + r.stderr[-2000:])
        res = np.load(os.path.join(td, "job.npz.out.npz"))
        return {"amps": res["amps"], "seconds": float(res["seconds"]), "nprocs": int(res["nprocs"]), "proc_seconds": res["proc_seconds"]}


def _child_main(job):
    import multiprocessing as mp
    import time
    d = np.load(job)
    flat, cfgs, chi, nprocs = d["flat"], d["cfgs"], int(d["chi"]), int(d["nprocs"])
    nprocs = max(1, min(nprocs, len(cfgs)))
    lib()                                          # build / load once in the parent; the forked workers inherit it
    shards = [cfgs[i::nprocs] for i in range(nprocs)]
    ctx = mp.get_context("fork")
    with ctx.Pool(nprocs) as pool:
        pool.map(_chunk, [(flat, cfgs[:0], chi)] * nprocs)      # workers up before the clock starts
        t0 = time.perf_counter()
        parts = pool.map(_chunk, [(flat, s, chi) for s in shards], chunksize=1)
        sec = time.perf_counter() - t0
    amps = np.zeros(len(cfgs))
    for i, (p, _) in enumerate(parts):
        amps[i::nprocs] = p
    np.savez(job + ".out.npz", amps=amps, seconds=sec, nprocs=nprocs, proc_seconds=np.array([t for _, t in parts]))


if __name__ == "__main__":
    import sys
    _child_main(sys.argv[1])
