"""ctypes binding of the C ABI declared in include/pepsgpu.h (libpepsgpu.so, HIP/gfx950).

This is the Python-side stub a maintainer would write to bind the library; it contains no
arithmetic and no CPU fallback: if the shared library is missing, import fails loudly.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PEPSGPU_LIB: another build of the same library (A/B measurements of kernel variants); there is no non-HIP implementation
LIB_PATH = os.environ.get("PEPSGPU_LIB") or os.path.join(_HERE, "lib", "libpepsgpu.so")

F32, F64 = 0, 1
C128 = 3          # complex float64 (TenElemT = QLTEN_Complex): every scalar / tensor output becomes complex128
LEFT, DOWN, RIGHT, UP = 0, 1, 2, 3
HORIZONTAL, VERTICAL = 0, 1
LEFTUP_TO_RIGHTDOWN, LEFTDOWN_TO_RIGHTUP = 0, 1   # basic.h:89-92 DIAGONAL_DIR
SVD_COMPRESS, VARIATION2SITE, VARIATION1SITE = 0, 1, 2   # CompressMPSScheme, bmps.h:31-35

_ERR = {1: ValueError, 2: RuntimeError, 3: RuntimeError, 4: IndexError, 5: RuntimeError}

SYMBOLS = [
    "pepsgpu_ctx_create", "pepsgpu_set_truncate_params", "pepsgpu_ctx_destroy", "pepsgpu_last_error", "pepsgpu_state_upload",
    "pepsgpu_walkers_set_configs", "pepsgpu_walkers_get_configs", "pepsgpu_n_walkers",
    "pepsgpu_grow_bmps_step", "pepsgpu_grow_full_bmps", "pepsgpu_grow_bmps_for_row", "pepsgpu_grow_bmps_for_col",
    "pepsgpu_shift_bmps_window", "pepsgpu_delete_inner_bmps", "pepsgpu_bmps_park", "pepsgpu_bmps_unpark", "pepsgpu_generate_bmps_approach",
    "pepsgpu_sweep_slice_exchange", "pepsgpu_sweep_slice_exchange_tab", "pepsgpu_sweep_slice_fullspace", "pepsgpu_nn_exchange_slice", "pepsgpu_walker_create", "pepsgpu_walker_clone", "pepsgpu_walker_destroy", "pepsgpu_walker_info", "pepsgpu_walker_set_mpo", "pepsgpu_walker_evolve",
    "pepsgpu_walker_evolve_step", "pepsgpu_walker_contract_row", "pepsgpu_walker_init_bten", "pepsgpu_walker_grow_bten_step",
    "pepsgpu_walker_shift_bten_window", "pepsgpu_walker_trace_with_bten", "pepsgpu_walker_clear_bten", "pepsgpu_walker_get_bmps_tensor",
    "pepsgpu_bmps_stack_size", "pepsgpu_get_bmps_tensor", "pepsgpu_init_bten", "pepsgpu_grow_full_bten",
    "pepsgpu_grow_bten_step", "pepsgpu_shift_bten_window", "pepsgpu_truncate_bten", "pepsgpu_bten_stack_size",
    "pepsgpu_trace", "pepsgpu_replace_nn_trace", "pepsgpu_replace_one_trace", "pepsgpu_punch_hole",
    "pepsgpu_init_bten2", "pepsgpu_grow_full_bten2", "pepsgpu_grow_bten2_step", "pepsgpu_shift_bten2_window",
    "pepsgpu_bten2_stack_size", "pepsgpu_replace_nnn_trace", "pepsgpu_replace_tnn_trace",
    "pepsgpu_bten2_select_set", "pepsgpu_cfg_override_slice", "pepsgpu_replace_plaquette_trace",
    "pepsgpu_replace_sqrt5_trace",
    "pepsgpu_grad_reset", "pepsgpu_grad_accumulate", "pepsgpu_grad_accumulate_states", "pepsgpu_grad_read", "pepsgpu_grad_device_ptr", "pepsgpu_grad_allreduce", "pepsgpu_bcast_state",
    "pepsgpu_comm_unique_id", "pepsgpu_comm_init", "pepsgpu_comm_size", "pepsgpu_comm_rank", "pepsgpu_comm_destroy",
    "pepsgpu_allreduce",
    "pepsgpu_sr_begin", "pepsgpu_sr_append", "pepsgpu_sr_count", "pepsgpu_sr_sum", "pepsgpu_sr_matvec", "pepsgpu_sr_matvec_c128",
    "pepsgpu_sr_cg_solve", "pepsgpu_sr_gram", "pepsgpu_sr_weighted_sum", "pepsgpu_sr_copy_samples",
    "pepsgpu_update_local", "pepsgpu_erase_envs_after_update", "pepsgpu_evaluate_amplitude",
    "pepsgpu_walker_flags", "pepsgpu_sync", "pepsgpu_stats", "pepsgpu_profile_enable", "pepsgpu_profile_read",
    "pepsgpu_diag_tgemm", "pepsgpu_diag_tgemm_chain", "pepsgpu_diag_chol", "pepsgpu_diag_chol_adaptive", "pepsgpu_diag_chol_pivot", "pepsgpu_diag_rows_qr", "pepsgpu_diag_suwa_todo", "pepsgpu_diag_gram_chol", "pepsgpu_diag_gram_cols", "pepsgpu_diag_gram_rows", "pepsgpu_diag_mgemm_dense", "pepsgpu_diag_jacobi", "pepsgpu_version",
]


def load_library(path=LIB_PATH):
    if not os.path.exists(path):
        raise ImportError("libpepsgpu.so not built (%s): run `python -c 'import __graft_entry__ as g; g.build()'`" % path)
    lib = C.CDLL(path)
    vp, ip, dp = C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_double)
    lib.pepsgpu_ctx_create.argtypes = [C.POINTER(vp)] + [C.c_int] * 8 + [C.c_double, C.c_int, C.c_int]
    lib.pepsgpu_set_truncate_params.argtypes = [vp, C.c_int, C.c_int, C.c_double, C.c_int, C.c_double, C.c_int]
    lib.pepsgpu_ctx_destroy.argtypes = [vp]
    lib.pepsgpu_ctx_destroy.restype = None
    lib.pepsgpu_last_error.argtypes = [vp]
    lib.pepsgpu_last_error.restype = C.c_char_p
    lib.pepsgpu_version.restype = C.c_char_p
    lib.pepsgpu_state_upload.argtypes = [vp, vp, C.c_int]
    lib.pepsgpu_walkers_set_configs.argtypes = [vp, C.c_int, ip]
    lib.pepsgpu_walkers_get_configs.argtypes = [vp, ip]
    lib.pepsgpu_n_walkers.argtypes = [vp]
    for name in ("grow_bmps_step", "grow_full_bmps", "grow_bmps_for_row", "grow_bmps_for_col", "shift_bmps_window",
                 "delete_inner_bmps", "generate_bmps_approach", "bmps_stack_size", "bten_stack_size",
                 "grow_bten_step", "shift_bten_window"):
        getattr(lib, "pepsgpu_" + name).argtypes = [vp, C.c_int]
    lib.pepsgpu_bmps_park.argtypes = [vp, C.c_int, C.c_int]
    lib.pepsgpu_bmps_unpark.argtypes = [vp, C.c_int]
    lib.pepsgpu_get_bmps_tensor.argtypes = [vp, C.c_int, C.c_int, C.c_int, ip, dp, dp]
    lib.pepsgpu_init_bten.argtypes = [vp, C.c_int, C.c_int]
    lib.pepsgpu_truncate_bten.argtypes = [vp, C.c_int, C.c_int]
    lib.pepsgpu_grow_full_bten.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.pepsgpu_trace.argtypes = [vp, C.c_int, C.c_int, C.c_int, dp]
    lib.pepsgpu_replace_nn_trace.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, ip, dp]
    lib.pepsgpu_replace_one_trace.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, ip, dp]
    lib.pepsgpu_punch_hole.argtypes = [vp, C.c_int, C.c_int, C.c_int, dp]
    for name in ("init_bten2", "grow_bten2_step", "shift_bten2_window"):
        getattr(lib, "pepsgpu_" + name).argtypes = [vp, C.c_int, C.c_int]
    lib.pepsgpu_grow_full_bten2.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.pepsgpu_bten2_stack_size.argtypes = [vp, C.c_int]
    lib.pepsgpu_replace_nnn_trace.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, ip, dp]
    lib.pepsgpu_bten2_select_set.argtypes = [vp, C.c_int]
    lib.pepsgpu_cfg_override_slice.argtypes = [vp, C.c_int, C.c_int, ip]
    lib.pepsgpu_replace_plaquette_trace.argtypes = [vp, C.c_int, C.c_int, C.c_int, ip, C.c_int, C.c_int, dp]
    lib.pepsgpu_replace_tnn_trace.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, ip, dp]
    lib.pepsgpu_replace_sqrt5_trace.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, ip, dp]
    lib.pepsgpu_grad_reset.argtypes = [vp]
    lib.pepsgpu_grad_accumulate.argtypes = [vp, dp, dp, C.c_int]
    lib.pepsgpu_grad_accumulate_states.argtypes = [vp, dp, dp, C.c_int, ip]
    lib.pepsgpu_grad_read.argtypes = [vp, dp, dp]
    lib.pepsgpu_grad_device_ptr.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_long)]
    lib.pepsgpu_grad_allreduce.argtypes = [vp]
    lib.pepsgpu_bcast_state.argtypes = [vp, C.c_int]
    lib.pepsgpu_comm_unique_id.argtypes = [vp]
    lib.pepsgpu_comm_init.argtypes = [vp, C.c_int, C.c_int, vp]
    lib.pepsgpu_comm_size.argtypes = [vp]
    lib.pepsgpu_comm_rank.argtypes = [vp]
    lib.pepsgpu_comm_destroy.argtypes = [vp]
    lib.pepsgpu_allreduce.argtypes = [vp, vp, C.c_long, C.c_int, C.c_int, C.c_int]
    lib.pepsgpu_sweep_slice_exchange.argtypes = [vp, C.c_int, C.c_int, C.c_int, dp, dp, ip, ip, ip]
    lib.pepsgpu_sweep_slice_exchange_tab.argtypes = [vp, C.c_int, C.c_int, C.c_int, dp, ip, dp, ip, ip, ip]
    lib.pepsgpu_sweep_slice_fullspace.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint32), dp, ip, ip]
    lib.pepsgpu_nn_exchange_slice.argtypes = [vp, C.c_int, C.c_int, C.c_int, dp, dp]
    lib.pepsgpu_walker_create.argtypes = [vp, C.c_int, C.c_int, ip]
    lib.pepsgpu_walker_clone.argtypes = [vp, C.c_int, ip]
    lib.pepsgpu_walker_destroy.argtypes = [vp, C.c_int]
    lib.pepsgpu_walker_info.argtypes = [vp, C.c_int, ip, ip, ip, ip]
    lib.pepsgpu_walker_set_mpo.argtypes = [vp, C.c_int, C.c_int, ip, dp, C.c_int]
    lib.pepsgpu_walker_evolve.argtypes = [vp, C.c_int]
    lib.pepsgpu_walker_evolve_step.argtypes = [vp, C.c_int]
    lib.pepsgpu_walker_contract_row.argtypes = [vp, C.c_int, C.c_int, dp]
    lib.pepsgpu_walker_init_bten.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.pepsgpu_walker_grow_bten_step.argtypes = [vp, C.c_int, C.c_int, C.c_int]
    lib.pepsgpu_walker_shift_bten_window.argtypes = [vp, C.c_int, C.c_int, C.c_int]
    lib.pepsgpu_walker_trace_with_bten.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, ip, dp, C.c_int, dp]
    lib.pepsgpu_walker_clear_bten.argtypes = [vp, C.c_int]
    lib.pepsgpu_walker_get_bmps_tensor.argtypes = [vp, C.c_int, C.c_int, ip, dp, dp]
    lib.pepsgpu_sr_begin.argtypes = [vp, C.c_int]
    lib.pepsgpu_sr_append.argtypes = [vp, dp]
    lib.pepsgpu_sr_count.argtypes = [vp]
    lib.pepsgpu_sr_sum.argtypes = [vp, dp]
    lib.pepsgpu_sr_matvec.argtypes = [vp, dp, C.c_double, C.c_double, dp]
    lib.pepsgpu_sr_matvec_c128.argtypes = [vp, dp, C.c_double, C.c_double, C.c_double, dp]
    lib.pepsgpu_sr_cg_solve.argtypes = [vp, dp, dp, C.c_double, C.c_int, C.c_double, C.c_double, C.c_int, C.c_double, dp, dp, ip, ip]
    lib.pepsgpu_sr_gram.argtypes = [vp, vp, vp, C.c_int, dp]
    lib.pepsgpu_sr_weighted_sum.argtypes = [vp, dp, dp]
    lib.pepsgpu_sr_copy_samples.argtypes = [vp, vp, vp]
    lib.pepsgpu_update_local.argtypes = [vp, C.c_int, ip, ip, C.POINTER(C.c_uint8)]
    lib.pepsgpu_erase_envs_after_update.argtypes = [vp, C.c_int, C.c_int]
    lib.pepsgpu_evaluate_amplitude.argtypes = [vp, dp]
    lib.pepsgpu_walker_flags.argtypes = [vp, ip]
    lib.pepsgpu_sync.argtypes = [vp]
    lib.pepsgpu_stats.argtypes = [vp, dp, C.c_int]
    lib.pepsgpu_profile_enable.argtypes = [vp, C.c_int]
    lib.pepsgpu_profile_read.argtypes = [vp, dp]
    lib.pepsgpu_diag_tgemm.argtypes = [C.c_int, C.c_int, ip, C.c_int, vp, C.c_size_t, vp, C.c_size_t, vp, C.c_size_t,
                                       C.c_int, C.c_long, C.c_long, C.c_long]
    lib.pepsgpu_diag_chol.argtypes = [C.c_int, dp, C.c_int, C.c_int, vp]
    lib.pepsgpu_diag_chol_adaptive.argtypes = [C.c_int, dp, C.c_int, C.c_int, vp, ip]
    lib.pepsgpu_diag_gram_chol.argtypes = [C.c_int, vp, C.c_int, C.c_int, C.c_int, vp, ip]
    lib.pepsgpu_diag_jacobi.argtypes = [C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, ip]
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = load_library()
    return _lib


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class _ReapingLib:
    """The library as a Context sees it: looking a function up first frees the BMPSWalker ids whose Python objects were finalised
    since the last call (Walker.__del__ only queues them), so the release happens on the thread that uses the context, between two
    calls, and before the next one can observe the walker."""

    def __init__(self, raw, ctx):
        object.__setattr__(self, "_raw", raw)
        object.__setattr__(self, "_ctx", ctx)

    def __getattr__(self, name):
        ctx = object.__getattribute__(self, "_ctx")
        raw = object.__getattribute__(self, "_raw")
        dead = ctx.__dict__.get("_dead_walkers")
        if dead and name != "pepsgpu_ctx_destroy":
            h = ctx.__dict__.get("_h")
            while dead and h:
                raw.pepsgpu_walker_destroy(h, dead.pop())
        return getattr(raw, name)


class Context:
    """Thin RAII wrapper of a pepsgpu_ctx: one walker batch on one GPU."""

    def __init__(self, rows, cols, D, phys_dim, chi, dtype=F32, device=0, max_walkers=256, chi_min=None,
                 trunc_err=0.0, scheme=0, convergence_tol=None, iter_max=None):
        self._l = _ReapingLib(lib(), self)
        self._dead_walkers = []
        self.rows, self.cols, self.D, self.d = rows, cols, D, phys_dim
        self.dtype = dtype
        self._ot = np.complex128 if dtype == C128 else np.float64      # type of every scalar / tensor the calls return
        h = C.c_void_p()
        rc = self._l.pepsgpu_ctx_create(C.byref(h), device, dtype, rows, cols, D, phys_dim,
                                        chi if chi_min is None else chi_min, chi, trunc_err, scheme, max_walkers)
        if rc != 0:
            raise _ERR.get(rc, RuntimeError)("pepsgpu_ctx_create failed (%d): %s"
                                             % (rc, self._l.pepsgpu_last_error(None).decode()))
        self._h = h
        self.n = 0
        if scheme != SVD_COMPRESS and convergence_tol is not None:
            self.set_truncate_params(chi if chi_min is None else chi_min, chi, trunc_err, scheme, convergence_tol, iter_max)

    def close(self):
        if getattr(self, "_h", None):
            self._l.pepsgpu_ctx_destroy(self._h)      # (drops every BMPSWalker of the context with it)
            self._h = None
        self._dead_walkers = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != 0:
            raise _ERR.get(rc, RuntimeError)("pepsgpu error %d: %s" % (rc, self._l.pepsgpu_last_error(self._h).decode()))

    # -- state / walkers --
    def set_truncate_params(self, chi_min, chi_max, trunc_err=0.0, scheme=0, convergence_tol=0.0, iter_max=0):
        """BMPSContractor::SetTruncateParams (bmps_contractor.h:216); scheme 0 SVD, 1 Variational2Site, 2 Variational1Site."""
        self._ck(self._l.pepsgpu_set_truncate_params(self._h, chi_min, chi_max, trunc_err, scheme, convergence_tol, iter_max))

    def state_upload(self, flat):
        flat = np.ascontiguousarray(flat)
        assert flat.shape == (self.rows, self.cols, self.d, self.D, self.D, self.D, self.D), flat.shape
        if np.iscomplexobj(flat):
            if self.dtype != C128:
                raise ValueError("a complex state needs a complex context (dtype=C128)")
            hd, flat = C128, np.ascontiguousarray(flat, dtype=np.complex128)
        else:
            hd = F32 if flat.dtype == np.float32 else F64
            if hd == F64:
                flat = flat.astype(np.float64, copy=False)
        self._ck(self._l.pepsgpu_state_upload(self._h, flat.ctypes.data_as(C.c_void_p), hd))

    def set_configs(self, configs):
        cfg = np.ascontiguousarray(configs, dtype=np.int32)
        assert cfg.ndim == 3 and cfg.shape[1:] == (self.rows, self.cols)
        self._ck(self._l.pepsgpu_walkers_set_configs(self._h, cfg.shape[0], _ip(cfg)))
        self.n = cfg.shape[0]

    def get_configs(self):
        out = np.zeros((self.n, self.rows, self.cols), dtype=np.int32)
        self._ck(self._l.pepsgpu_walkers_get_configs(self._h, _ip(out)))
        return out

    # -- contractor mirror --
    def grow_bmps_step(self, pos): self._ck(self._l.pepsgpu_grow_bmps_step(self._h, pos))
    def grow_full_bmps(self, pos): self._ck(self._l.pepsgpu_grow_full_bmps(self._h, pos))
    def grow_bmps_for_row(self, row): self._ck(self._l.pepsgpu_grow_bmps_for_row(self._h, row))
    def grow_bmps_for_col(self, col): self._ck(self._l.pepsgpu_grow_bmps_for_col(self._h, col))
    def shift_bmps_window(self, pos): self._ck(self._l.pepsgpu_shift_bmps_window(self._h, pos))
    def delete_inner_bmps(self, pos): self._ck(self._l.pepsgpu_delete_inner_bmps(self._h, pos))
    def bmps_park(self, pos, keep): self._ck(self._l.pepsgpu_bmps_park(self._h, pos, keep))
    def bmps_unpark(self, pos): self._ck(self._l.pepsgpu_bmps_unpark(self._h, pos))
    def generate_bmps_approach(self, pos): self._ck(self._l.pepsgpu_generate_bmps_approach(self._h, pos))
    def bmps_stack_size(self, pos): return self._l.pepsgpu_bmps_stack_size(self._h, pos)
    def bten_stack_size(self, pos): return self._l.pepsgpu_bten_stack_size(self._h, pos)
    def init_bten(self, pos, slice_num): self._ck(self._l.pepsgpu_init_bten(self._h, pos, slice_num))
    def grow_full_bten(self, pos, slice_num, remain_sites=2, init=True):
        self._ck(self._l.pepsgpu_grow_full_bten(self._h, pos, slice_num, remain_sites, int(init)))
    def grow_bten_step(self, pos): self._ck(self._l.pepsgpu_grow_bten_step(self._h, pos))
    def shift_bten_window(self, pos): self._ck(self._l.pepsgpu_shift_bten_window(self._h, pos))
    def truncate_bten(self, pos, length): self._ck(self._l.pepsgpu_truncate_bten(self._h, pos, length))

    def get_bmps_tensor(self, pos, level, idx):
        dims = np.zeros(3, dtype=np.int32)
        self._ck(self._l.pepsgpu_get_bmps_tensor(self._h, pos, level, idx, _ip(dims), None, None))
        data = np.zeros((self.n,) + tuple(int(x) for x in dims), dtype=self._ot)
        ls = np.zeros(self.n, dtype=np.float64)
        self._ck(self._l.pepsgpu_get_bmps_tensor(self._h, pos, level, idx, _ip(dims), _dp(data), _dp(ls)))
        return data, ls

    def sweep_slice_exchange(self, orientation, slice_num, uniforms, amplitude, pair_table=None):
        """one row / column of the NN-exchange sweep on the device; uniforms [n][nu] (next deviates of each walker's stream);
        pair_table [d * d][2] (optional): the pair of states the move proposes for the states (a, b) of a bond (fermionic extended
        states); returns (amplitude, consumed [n], accepted [n], slice_states [n][N])"""
        u = np.ascontiguousarray(uniforms, dtype=np.float64)
        assert u.ndim == 2 and u.shape[0] == self.n
        amp = np.array(amplitude, dtype=self._ot)
        N = self.cols if orientation == HORIZONTAL else self.rows
        cons, acc = np.zeros(self.n, dtype=np.int32), np.zeros(self.n, dtype=np.int32)
        st = np.zeros((self.n, N), dtype=np.int32)
        if pair_table is None:
            self._ck(self._l.pepsgpu_sweep_slice_exchange(self._h, orientation, slice_num, u.shape[1], _dp(u), _dp(amp), _ip(cons), _ip(acc), _ip(st)))
        else:
            tab = np.ascontiguousarray(pair_table, dtype=np.int32)
            assert tab.shape == (self.d * self.d, 2), tab.shape
            self._ck(self._l.pepsgpu_sweep_slice_exchange_tab(self._h, orientation, slice_num, u.shape[1], _dp(u), _ip(tab), _dp(amp), _ip(cons),
                                                              _ip(acc), _ip(st)))
        return amp, cons, acc, st

    def sweep_slice_fullspace(self, orientation, slice_num, phys_dim, engine_words, amplitude):
        """one row / column of the full-space NN updater (Suwa-Todo over phys_dim^2 states per bond) on the device; engine_words
        [n][2 (N - 1)] uint32 = the next raw outputs of each walker's mt19937; returns (amplitude, accepted [n], slice_states [n][N])"""
        N = self.cols if orientation == HORIZONTAL else self.rows
        wd = np.ascontiguousarray(engine_words, dtype=np.uint32)
        assert wd.shape == (self.n, 2 * (N - 1)), wd.shape
        amp = np.array(amplitude, dtype=self._ot)
        acc = np.zeros(self.n, dtype=np.int32)
        st = np.zeros((self.n, N), dtype=np.int32)
        self._ck(self._l.pepsgpu_sweep_slice_fullspace(self._h, orientation, slice_num, phys_dim, wd.ctypes.data_as(C.POINTER(C.c_uint32)),
                                                       _dp(amp), _ip(acc), _ip(st)))
        return amp, acc, st

    def nn_exchange_slice(self, orientation, slice_num, punch_holes=False):
        """psi [n] and the amplitudes with the two sites of every bond of the slice exchanged [n][N-1], one read-back"""
        N = self.cols if orientation == HORIZONTAL else self.rows
        psi, ex = np.zeros(self.n), np.zeros((self.n, N - 1))
        self._ck(self._l.pepsgpu_nn_exchange_slice(self._h, orientation, slice_num, int(punch_holes), _dp(psi), _dp(ex)))
        return psi, ex

    def get_walker(self, pos, level=-1):
        """BMPSContractor::GetWalker(tn, pos) (bmps_walker.h:51-58): a Walker object forked from the top of stack `pos`
        (level >= 0: from that level of the stack, BMPSWalker(tn, stack[level], pos, level + 1, params))"""
        wid = np.zeros(1, dtype=np.int32)
        self._ck(self._l.pepsgpu_walker_create(self._h, pos, level, _ip(wid)))
        return Walker(self, int(wid[0]))

    def trace(self, row, col, bond_dir):
        out = np.zeros(self.n, dtype=self._ot)
        self._ck(self._l.pepsgpu_trace(self._h, row, col, bond_dir, _dp(out)))
        return out

    def replace_nn_trace(self, row, col, bond_dir, cand_states):
        cand = np.ascontiguousarray(cand_states, dtype=np.int32)
        assert cand.ndim == 3 and cand.shape[0] == self.n and cand.shape[2] == 2
        out = np.zeros((self.n, cand.shape[1]), dtype=self._ot)
        self._ck(self._l.pepsgpu_replace_nn_trace(self._h, row, col, bond_dir, cand.shape[1], _ip(cand), _dp(out)))
        return out

    def replace_one_trace(self, row, col, orient, cand_states):
        cand = np.ascontiguousarray(cand_states, dtype=np.int32)
        assert cand.ndim == 2 and cand.shape[0] == self.n
        out = np.zeros((self.n, cand.shape[1]), dtype=self._ot)
        self._ck(self._l.pepsgpu_replace_one_trace(self._h, row, col, orient, cand.shape[1], _ip(cand), _dp(out)))
        return out

    # -- two-row environments, NNN / third-neighbour / sqrt(5) traces --
    def init_bten2(self, pos, slice_num1): self._ck(self._l.pepsgpu_init_bten2(self._h, pos, slice_num1))
    def grow_full_bten2(self, pos, slice_num1, remain_sites=2, init=True):
        self._ck(self._l.pepsgpu_grow_full_bten2(self._h, pos, slice_num1, remain_sites, int(init)))
    def grow_bten2_step(self, pos, slice_num1): self._ck(self._l.pepsgpu_grow_bten2_step(self._h, pos, slice_num1))
    def shift_bten2_window(self, pos, slice_num1): self._ck(self._l.pepsgpu_shift_bten2_window(self._h, pos, slice_num1))
    def bten2_stack_size(self, pos): return self._l.pepsgpu_bten2_stack_size(self._h, pos)

    def _cand(self, cand_states, ncols):
        """cand_states None -> no replacement (out [n]); else [n][n_cand][ncols] -> out [n][n_cand]"""
        if cand_states is None:
            return 0, None, np.zeros(self.n, dtype=self._ot)
        cand = np.ascontiguousarray(cand_states, dtype=np.int32)
        assert cand.ndim == 3 and cand.shape[0] == self.n and cand.shape[2] == ncols
        return cand.shape[1], cand, np.zeros((self.n, cand.shape[1]), dtype=self._ot)

    def bten2_select_set(self, which): self._ck(self._l.pepsgpu_bten2_select_set(self._h, int(which)))

    def cfg_override_slice(self, orient, num, states=None):
        """states [n][N] (extended states of one row / column read instead of the walkers' own), None: clear"""
        if states is None:
            self._ck(self._l.pepsgpu_cfg_override_slice(self._h, orient, num, None))
        else:
            st = np.ascontiguousarray(states, dtype=np.int32)
            # (the C ABI carries no length: the engine reads states[w * N + j], N = columns of a row / rows of a column)
            want = self.cols if orient == HORIZONTAL else self.rows
            assert st.ndim == 2 and st.shape[0] == self.n and st.shape[1] == want, (st.shape, self.n, want)
            self._ck(self._l.pepsgpu_cfg_override_slice(self._h, orient, num, _ip(st)))

    def replace_plaquette_trace(self, row, col, cand_states=None, left_set=0, right_set=0):
        """cand_states [n][n_cand][4]: states of (row, col), (row+1, col), (row+1, col+1), (row, col+1); None: the walkers' own"""
        nc, cand, out = self._cand(cand_states, 4)
        self._ck(self._l.pepsgpu_replace_plaquette_trace(self._h, row, col, nc, None if cand is None else _ip(cand), left_set, right_set,
                                                         _dp(out)))
        return out

    def replace_nnn_trace(self, row, col, nnn_dir, orient, cand_states=None):
        nc, cand, out = self._cand(cand_states, 2)
        self._ck(self._l.pepsgpu_replace_nnn_trace(self._h, row, col, nnn_dir, orient, nc,
                                                   None if cand is None else _ip(cand), _dp(out)))
        return out

    def replace_tnn_trace(self, row, col, orient, cand_states=None):
        nc, cand, out = self._cand(cand_states, 3)
        self._ck(self._l.pepsgpu_replace_tnn_trace(self._h, row, col, orient, nc,
                                                   None if cand is None else _ip(cand), _dp(out)))
        return out

    def replace_sqrt5_trace(self, row, col, link_dir, orient, cand_states=None):
        nc, cand, out = self._cand(cand_states, 2)
        self._ck(self._l.pepsgpu_replace_sqrt5_trace(self._h, row, col, link_dir, orient, nc,
                                                     None if cand is None else _ip(cand), _dp(out)))
        return out

    def punch_hole(self, row, col, orient):
        out = np.zeros((self.n, self.D, self.D, self.D, self.D), dtype=self._ot)
        self._ck(self._l.pepsgpu_punch_hole(self._h, row, col, orient, _dp(out)))
        return out

    def punch_hole_store(self, row, col, orient):
        self._ck(self._l.pepsgpu_punch_hole(self._h, row, col, orient, None))

    # -- stochastic reconfiguration: O* samples resident in HBM --
    def sr_begin(self, max_samples):
        self._ck(self._l.pepsgpu_sr_begin(self._h, max_samples))

    def sr_append(self, psi):
        psi = np.ascontiguousarray(psi, dtype=self._ot)
        self._ck(self._l.pepsgpu_sr_append(self._h, _dp(psi)))

    def sr_count(self):
        return self._l.pepsgpu_sr_count(self._h)

    def sr_sum(self):
        out = np.zeros((self.rows, self.cols, self.d, self.D, self.D, self.D, self.D), dtype=self._ot)
        self._ck(self._l.pepsgpu_sr_sum(self._h, _dp(out)))
        return out

    def sr_matvec(self, v, mean_dot_v, scale):
        v = np.ascontiguousarray(v, dtype=self._ot)
        out = np.zeros_like(v)
        if self._ot is np.complex128:
            m = complex(mean_dot_v)
            self._ck(self._l.pepsgpu_sr_matvec_c128(self._h, _dp(v), m.real, m.imag, float(scale), _dp(out)))
        else:
            self._ck(self._l.pepsgpu_sr_matvec(self._h, _dp(v), float(mean_dot_v), float(scale), _dp(out)))
        return out

    def sr_cg_solve(self, b, x0=None, diag_shift=0.0, max_iter=100, relative_tolerance=1e-4, absolute_tolerance=0.0,
                    residual_recompute_interval=20, orthogonality_threshold=0.5):
        """(S + diag_shift) x = b with every CG vector on the device; returns (x, residual_norm, iterations, reason)."""
        dt = np.complex128 if self.dtype == C128 else np.float64       # complex contexts: interleaved (re, im) pairs through the C ABI
        b = np.ascontiguousarray(b, dtype=dt)
        x0a = None if x0 is None else np.ascontiguousarray(x0, dtype=dt)
        x = np.zeros_like(b)
        res = np.zeros(1, dtype=np.float64)
        it = np.zeros(2, dtype=np.int32)
        self._ck(self._l.pepsgpu_sr_cg_solve(self._h, _dp(b), None if x0a is None else _dp(x0a), diag_shift, max_iter,
                                             relative_tolerance, absolute_tolerance, residual_recompute_interval,
                                             orthogonality_threshold, _dp(x), _dp(res), _ip(it[:1]), _ip(it[1:])))
        return x, float(res[0]), int(it[0]), int(it[1])

    def sr_gram(self, remote_samples_ptr=None, remote_configs_ptr=None, n_remote=0):
        """raw inner products of the local O* samples with themselves (default) or with a device-resident remote batch"""
        n = self.sr_count()
        dt = np.complex128 if self.dtype == C128 else np.float64       # complex contexts: ip_ij = sum conj(O*_i) O*_j, interleaved pairs
        out = np.zeros((n, n_remote if remote_samples_ptr else n), dtype=dt)
        self._ck(self._l.pepsgpu_sr_gram(self._h, remote_samples_ptr, remote_configs_ptr, n_remote, _dp(out)))
        return out

    def sr_weighted_sum(self, y):
        dt = np.complex128 if self.dtype == C128 else np.float64
        y = np.ascontiguousarray(y, dtype=dt)
        assert y.size == self.sr_count()
        out = np.zeros((self.rows, self.cols, self.d, self.D, self.D, self.D, self.D), dtype=dt)
        self._ck(self._l.pepsgpu_sr_weighted_sum(self._h, _dp(y), _dp(out)))
        return out

    def sr_copy_samples(self, dst_samples_ptr, dst_configs_ptr):
        self._ck(self._l.pepsgpu_sr_copy_samples(self._h, dst_samples_ptr, dst_configs_ptr))

    def grad_reset(self):
        self._ck(self._l.pepsgpu_grad_reset(self._h))

    def grad_accumulate(self, psi, eloc, exact_sum=False, states=None):
        """states ([n][rows][cols], optional): the component every stored hole belongs to, when it is not the walkers'
        current configuration (fermionic states: extended states of the row-major decoration)."""
        psi = np.ascontiguousarray(psi, dtype=np.float64)
        eloc = np.ascontiguousarray(eloc, dtype=np.float64)
        if states is None:
            self._ck(self._l.pepsgpu_grad_accumulate(self._h, _dp(psi), _dp(eloc), int(exact_sum)))
            return
        states = np.ascontiguousarray(states, dtype=np.int32)
        if states.shape != (len(psi), self.rows, self.cols):
            raise ValueError("states must be [n][rows][cols]")
        self._ck(self._l.pepsgpu_grad_accumulate_states(self._h, _dp(psi), _dp(eloc), int(exact_sum),
                                                        states.ctypes.data_as(C.POINTER(C.c_int32))))

    def grad_read(self):
        shp = (self.rows, self.cols, self.d, self.D, self.D, self.D, self.D)
        so, seo = np.zeros(shp), np.zeros(shp)
        self._ck(self._l.pepsgpu_grad_read(self._h, _dp(so), _dp(seo)))
        return so, seo

    # -- the exchange step: RCCL all-reduce of the accumulators (one rank = one context per GPU) --
    def grad_device_ptr(self):
        """(S_O pointer, S_EO pointer, elements): the float64 accumulators where they live in HBM."""
        so, seo, n = C.c_void_p(), C.c_void_p(), C.c_long()
        self._ck(self._l.pepsgpu_grad_device_ptr(self._h, C.byref(so), C.byref(seo), C.byref(n)))
        return so.value, seo.value, n.value

    def bcast_state(self, root=0):
        """ncclBroadcast of the flat SITPS in HBM from rank `root` over the context's communicator (comm_init)"""
        self._ck(self._l.pepsgpu_bcast_state(self._h, int(root)))

    def grad_allreduce(self):
        self._ck(self._l.pepsgpu_grad_allreduce(self._h))

    def comm_init(self, nranks, rank, unique_id=None):
        """unique_id: the 128 bytes of comm_unique_id() of rank 0 (None only for a single rank)."""
        buf = None if unique_id is None else C.create_string_buffer(bytes(unique_id), 128)
        self._ck(self._l.pepsgpu_comm_init(self._h, nranks, rank, buf))

    def comm_size(self): return self._l.pepsgpu_comm_size(self._h)
    def comm_rank(self): return self._l.pepsgpu_comm_rank(self._h)
    def comm_destroy(self): self._ck(self._l.pepsgpu_comm_destroy(self._h))

    def allreduce(self, arr, op="sum"):
        """in-place all-reduce of a host NumPy array (float32 / float64 / int32) over the context's communicator"""
        code = {np.dtype(np.float32): 0, np.dtype(np.float64): 1, np.dtype(np.int32): 2}[arr.dtype]
        assert arr.flags["C_CONTIGUOUS"]
        self._ck(self._l.pepsgpu_allreduce(self._h, arr.ctypes.data_as(C.c_void_p), arr.size, code, {"sum": 0, "max": 1}[op], 0))
        return arr

    def allreduce_device(self, ptr, n, dtype=F64, op="sum"):
        self._ck(self._l.pepsgpu_allreduce(self._h, C.c_void_p(ptr), n, {F32: 0, F64: 1}[dtype], {"sum": 0, "max": 1}[op], 1))

    def update_local(self, sites, new_states, accept_mask=None):
        sites = np.ascontiguousarray(sites, dtype=np.int32).reshape(-1, 2)
        ns = np.ascontiguousarray(new_states, dtype=np.int32).reshape(self.n, sites.shape[0])
        m = None
        if accept_mask is not None:
            m = np.ascontiguousarray(accept_mask, dtype=np.uint8)
        self._ck(self._l.pepsgpu_update_local(self._h, sites.shape[0], _ip(sites), _ip(ns),
                                              m.ctypes.data_as(C.POINTER(C.c_uint8)) if m is not None else None))

    def erase_envs_after_update(self, row, col):
        self._ck(self._l.pepsgpu_erase_envs_after_update(self._h, row, col))

    def evaluate_amplitude(self):
        out = np.zeros(self.n, dtype=self._ot)
        self._ck(self._l.pepsgpu_evaluate_amplitude(self._h, _dp(out)))
        return out

    def walker_flags(self):
        out = np.zeros(self.n, dtype=np.int32)
        self._ck(self._l.pepsgpu_walker_flags(self._h, _ip(out)))
        return out

    def sync(self):
        self._ck(self._l.pepsgpu_sync(self._h))

    PROF_CATS = ("contract", "gram_f64", "cholesky", "jacobi", "select", "normalize", "env", "jacobi_edge", "trunc_gram",
                 "trunc_apply", "contract_chain")

    def profile_enable(self, on=True):
        self._ck(self._l.pepsgpu_profile_enable(self._h, int(on)))

    def profile_read(self):
        out = np.zeros((len(self.PROF_CATS), 5), dtype=np.float64)
        self._ck(self._l.pepsgpu_profile_read(self._h, _dp(out)))
        return {name: {"ms": out[i, 0], "launches": int(out[i, 1]), "alg_flops": out[i, 2], "exec_flops": out[i, 3],
                       "bytes": out[i, 4]}
                for i, name in enumerate(self.PROF_CATS)}

    def stats(self):
        out = np.zeros(9, dtype=np.float64)
        self._ck(self._l.pepsgpu_stats(self._h, _dp(out), 9))
        return {"absorptions": int(out[0]), "jacobi_launches": int(out[1]), "jacobi_sweeps_sum": int(out[2]),
                "device_bytes": int(out[3]), "jacobi_sweeps_max": int(out[4]),
                "carry_live_fraction": float(out[5] / out[6]) if out[6] > 0 else None,
                "carry_live_max": int(out[7]), "absorptions_redone": int(out[8])}


def comm_unique_id():
    """ncclGetUniqueId through the library: 128 bytes that rank 0 hands to every rank (MPI_Bcast in the reference host,
    torch.distributed.broadcast_object_list in peps_amd/dist.py)."""
    buf = C.create_string_buffer(128)
    rc = lib().pepsgpu_comm_unique_id(buf)
    if rc != 0:
        raise RuntimeError("pepsgpu_comm_unique_id failed: %s" % lib().pepsgpu_last_error(None).decode())
    return buf.raw


# -- diagnostics (kernel unit tests) --
def diag_tgemm(dtype_in, dtype_out, I, J, K, sAi, sAk, sBk, sBj, sCi, sCj, A, B, C_init, nbatch, wA, wB, wC):
    ints = np.array(list(I) + list(J) + list(K) + list(sAi) + list(sAk) + list(sBk) + list(sBj) + list(sCi) + list(sCj),
                    dtype=np.int32)
    tin = np.float32 if dtype_in == F32 else np.float64
    tout = np.float32 if dtype_out == F32 else np.float64
    A = np.ascontiguousarray(A, dtype=tin)
    B = np.ascontiguousarray(B, dtype=tin)
    Cc = np.ascontiguousarray(C_init, dtype=tout).copy()
    rc = lib().pepsgpu_diag_tgemm(dtype_in, dtype_out, _ip(ints), 27, A.ctypes.data_as(C.c_void_p), A.size,
                                  B.ctypes.data_as(C.c_void_p), B.size, Cc.ctypes.data_as(C.c_void_p), Cc.size,
                                  nbatch, wA, wB, wC)
    if rc != 0:
        raise RuntimeError("diag_tgemm failed: %s" % lib().pepsgpu_last_error(None).decode())
    return Cc


def diag_tgemm_chain(R, A, W, live):
    """P[b][m,u,l2,a2] = sum R[b][m,l,a] A[b][a,p,a2] W[b][l,p,l2,u] over the live extents live[b] = (m, a, a2), through
    tgemm_chain_kernel with the descriptors of Engine::absorb_impl; returns (P, flags)."""
    R = np.ascontiguousarray(R, dtype=np.float32); A = np.ascontiguousarray(A, dtype=np.float32)
    W = np.ascontiguousarray(W, dtype=np.float32)
    nb, m, l, a = R.shape
    _, _, p, a2 = A.shape
    l2, u = W.shape[3], W.shape[4]
    dims = np.array([m, l, a, p, a2, l2, u], dtype=np.int32)
    lv = np.ascontiguousarray(live, dtype=np.int32)
    P = np.zeros((nb, m, u, l2, a2), dtype=np.float32)
    fl = np.zeros(nb, dtype=np.int32)
    f = lib().pepsgpu_diag_tgemm_chain
    f.argtypes = [C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]
    rc = f(_ip(dims), _ip(lv), nb, R.ctypes.data_as(C.c_void_p), A.ctypes.data_as(C.c_void_p), W.ctypes.data_as(C.c_void_p),
           P.ctypes.data_as(C.c_void_p), _ip(fl))
    if rc != 0:
        raise RuntimeError("diag_tgemm_chain failed: %s" % lib().pepsgpu_last_error(None).decode())
    return P, fl


def diag_chol(dtype_out, G):
    G = np.ascontiguousarray(G, dtype=np.float64)
    nb, n, _ = G.shape
    R = np.zeros((nb, n, n), dtype=np.float32 if dtype_out == F32 else np.float64)
    rc = lib().pepsgpu_diag_chol(dtype_out, _dp(G), n, nb, R.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise RuntimeError("diag_chol failed: %s" % lib().pepsgpu_last_error(None).decode())
    return R


def diag_chol_adaptive(dtype_out, G):
    """Low-rank + blocked Cholesky pair as the absorption runs it; returns (R, mlive)."""
    G = np.ascontiguousarray(G, dtype=np.float64)
    nb, n, _ = G.shape
    R = np.zeros((nb, n, n), dtype=np.float32 if dtype_out == F32 else np.float64)
    ml = np.zeros(nb, dtype=np.int32)
    rc = lib().pepsgpu_diag_chol_adaptive(dtype_out, _dp(G), n, nb, R.ctypes.data_as(C.c_void_p), _ip(ml))
    if rc != 0:
        raise RuntimeError("diag_chol_adaptive failed: %s" % lib().pepsgpu_last_error(None).decode())
    return R, ml


def diag_gram_chol(dtype, P):
    """gram_chol_lowrank_kernel alone; P = [nb][K][n]; returns (R [nb][n][n], mlive)."""
    t = np.float32 if dtype == F32 else np.float64
    P = np.ascontiguousarray(P, dtype=t)
    nb, K, n = P.shape
    R = np.zeros((nb, n, n), dtype=t)
    ml = np.zeros(nb, dtype=np.int32)
    rc = lib().pepsgpu_diag_gram_chol(dtype, P.ctypes.data_as(C.c_void_p), K, n, nb, R.ctypes.data_as(C.c_void_p), _ip(ml))
    if rc != 0:
        raise RuntimeError("diag_gram_chol failed: %s" % lib().pepsgpu_last_error(None).decode())
    return R, ml


def diag_gram_cols(dtype, P, klive=None):
    """gram_cols_f64_kernel alone; P = [nb][K][n]; returns G [nb][n][n] float64 (64 x 64 blocks on / above the diagonal)."""
    t = np.float32 if dtype == F32 else np.float64
    P = np.ascontiguousarray(P, dtype=t)
    nb, K, n = P.shape
    G = np.zeros((nb, n, n), dtype=np.float64)
    kl = None if klive is None else np.ascontiguousarray(klive, dtype=np.int32)
    f = lib().pepsgpu_diag_gram_cols
    f.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_double)]
    rc = f(dtype, P.ctypes.data_as(C.c_void_p), K, n, nb, None if kl is None else _ip(kl), _dp(G))
    if rc != 0:
        raise RuntimeError("diag_gram_cols failed: %s" % lib().pepsgpu_last_error(None).decode())
    return G


def diag_gram_rows(M, nrows):
    """gram_rows_f64_kernel alone; M = [nb][n][K] float32 (K a multiple of 16), nrows[b] live rows; returns G [nb][n][n] float64
    (64 x 64 blocks on / above the diagonal of the live part)."""
    M = np.ascontiguousarray(M, dtype=np.float32)
    nb, n, K = M.shape
    G = np.zeros((nb, n, n), dtype=np.float64)
    nr = np.ascontiguousarray(nrows, dtype=np.int32)
    f = lib().pepsgpu_diag_gram_rows
    f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_double)]
    rc = f(M.ctypes.data_as(C.c_void_p), n, K, nb, _ip(nr), _dp(G))
    if rc != 0:
        raise RuntimeError("diag_gram_rows failed: %s" % lib().pepsgpu_last_error(None).decode())
    return G


def diag_lds_gram_chol(which, X, nlive):
    """mid_gram_chol_kernel (which = 0: X [nb][n][K], R^T R = X X^T over the nlive[b] first rows) or colgram_dense_kernel (which = 1:
    X [nb][K][n], R^T R = X^T X over the nlive[b] first rows) alone; returns (R [nb][n][n] float32 -- rows beyond mlive are NaN --, mlive)."""
    X = np.ascontiguousarray(X, dtype=np.float32)
    nb = X.shape[0]
    n, K = (X.shape[1], X.shape[2]) if which == 0 else (X.shape[2], X.shape[1])
    R = np.zeros((nb, n, n), dtype=np.float32)
    ml = np.zeros(nb, dtype=np.int32)
    nl = np.ascontiguousarray(nlive, dtype=np.int32)
    f = lib().pepsgpu_diag_lds_gram_chol
    f.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32), C.c_void_p, C.POINTER(C.c_int32)]
    rc = f(which, X.ctypes.data_as(C.c_void_p), n, K, nb, _ip(nl), R.ctypes.data_as(C.c_void_p), _ip(ml))
    if rc != 0:
        raise RuntimeError("diag_lds_gram_chol failed: %s" % lib().pepsgpu_last_error(None).decode())
    return R, ml


def diag_chol_pivot(X, nlive, kcap=64):
    """The first compression of the dense truncation route (round 6): i8 row Gram with both triangles + chol_pivot_kernel.
    X [nb][n][K] float32, 128 < n <= 256; returns (R [nb][kcap][n] float32 in pivot order -- rows beyond mlive are NaN --, mlive)."""
    X = np.ascontiguousarray(X, dtype=np.float32)
    nb, n, K = X.shape
    R = np.zeros((nb, kcap, n), dtype=np.float32)
    ml = np.zeros(nb, dtype=np.int32)
    nl = np.ascontiguousarray(nlive, dtype=np.int32)
    f = lib().pepsgpu_diag_chol_pivot
    f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32), C.c_int, C.c_void_p, C.POINTER(C.c_int32)]
    rc = f(X.ctypes.data_as(C.c_void_p), n, K, nb, _ip(nl), kcap, R.ctypes.data_as(C.c_void_p), _ip(ml))
    if rc != 0:
        raise RuntimeError("diag_chol_pivot failed: %s" % lib().pepsgpu_last_error(None).decode())
    return R, ml


def diag_rows_qr(X, klive):
    """rows_qr_kernel alone: X [nb][k][len] float32 -> (V [nb][k][len] orthonormal live rows first, klive_out)."""
    X = np.ascontiguousarray(X, dtype=np.float32)
    nb, k, ln = X.shape
    V = np.zeros_like(X)
    ml = np.zeros(nb, dtype=np.int32)
    kl = np.ascontiguousarray(klive, dtype=np.int32)
    f = lib().pepsgpu_diag_rows_qr
    f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32), C.c_void_p, C.POINTER(C.c_int32)]
    rc = f(X.ctypes.data_as(C.c_void_p), k, ln, nb, _ip(kl), V.ctypes.data_as(C.c_void_p), _ip(ml))
    if rc != 0:
        raise RuntimeError("diag_rows_qr failed: %s" % lib().pepsgpu_last_error(None).decode())
    return V, ml


def diag_suwa_todo(weights, init, words):
    """the device's SuwaTodoStateUpdate: a chain of len(words) / 2 updates; words = raw uint32 outputs of std::mt19937"""
    w = np.ascontiguousarray(weights, dtype=np.float64)
    wd = np.ascontiguousarray(words, dtype=np.uint32)
    steps = len(wd) // 2
    out = np.zeros(steps, dtype=np.int32)
    f = lib().pepsgpu_diag_suwa_todo
    f.argtypes = [C.POINTER(C.c_double), C.c_int, C.c_int, C.POINTER(C.c_uint32), C.c_int, C.POINTER(C.c_int32)]
    rc = f(_dp(w), len(w), int(init), wd.ctypes.data_as(C.POINTER(C.c_uint32)), steps, _ip(out))
    if rc != 0:
        raise RuntimeError("diag_suwa_todo failed: %s" % lib().pepsgpu_last_error(None).decode())
    return out


def diag_mgemm_dense(R, Tt, a_dim, u_dim, k2_dim, tt_u_inner, m_live=None, a_live=None, k2_live=None):
    """mgemm_dense_kernel alone: R [nb][m][la], Tt [nb][la][u * k2] (inner order (u, k2), or (k2, u) with tt_u_inner) -> M [nb][m][u * k2]
    (rows beyond m_live[b] come back as NaN: untouched)."""
    R = np.ascontiguousarray(R, dtype=np.float32)
    Tt = np.ascontiguousarray(Tt, dtype=np.float32)
    nb, m, la = R.shape
    uk = u_dim * k2_dim
    assert Tt.shape == (nb, la, uk)
    M = np.zeros((nb, m, uk), dtype=np.float32)
    arrs = [None if x is None else np.ascontiguousarray(x, dtype=np.int32) for x in (m_live, a_live, k2_live)]
    f = lib().pepsgpu_diag_mgemm_dense
    ip = C.POINTER(C.c_int32)
    f.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 7 + [ip, ip, ip, C.c_void_p]
    rc = f(R.ctypes.data_as(C.c_void_p), Tt.ctypes.data_as(C.c_void_p), m, la, a_dim, u_dim, k2_dim, int(tt_u_inner), nb,
           *[None if a is None else _ip(a) for a in arrs], M.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise RuntimeError("diag_mgemm_dense failed: %s" % lib().pepsgpu_last_error(None).decode())
    return M


def diag_jacobi(dtype, M, k, force_global=False):
    t = np.float32 if dtype == F32 else np.float64
    M = np.ascontiguousarray(M, dtype=t).copy()
    nb, m, ln = M.shape
    Vt = np.zeros((nb, k, ln), dtype=t)
    S = np.zeros((nb, k), dtype=t)
    sw = np.zeros(nb, dtype=np.int32)
    rc = lib().pepsgpu_diag_jacobi(dtype, M.ctypes.data_as(C.c_void_p), m, ln, nb, k, Vt.ctypes.data_as(C.c_void_p),
                                   S.ctypes.data_as(C.c_void_p), int(force_global), _ip(sw))
    if rc != 0:
        raise RuntimeError("diag_jacobi failed: %s" % lib().pepsgpu_last_error(None).decode())
    return M, Vt, S, sw & 0xFF      # high bits: diagnostics (rows above the noise floor at entry)


class Walker:
    """BMPSContractor::BMPSWalker (bmps_contractor.h:357-646) for all Monte-Carlo walkers of a context: method names follow the
    reference.  The TransferMPO is set once (set_mpo / set_mpo_states / set_mpo_tensors); `opp_level` names the opposite
    boundary = level of the DOWN stack (down_stack[opp_level] in the reference's tests).
    Lifetime: the device copy is released by destroy(), at the end of a `with ctx.get_walker(...) as w:` block, or when the object is
    collected; Context.set_configs invalidates every walker of the context."""

    def __init__(self, ctx, wid):
        self.ctx, self.wid = ctx, wid

    def _info(self):
        v = np.zeros(4, dtype=np.int32)
        c = self.ctx
        c._ck(c._l.pepsgpu_walker_info(c._h, self.wid, _ip(v[0:1]), _ip(v[1:2]), _ip(v[2:3]), _ip(v[3:4])))
        return v

    def GetPosition(self): return int(self._info()[0])
    def GetStackSize(self): return int(self._info()[1])
    def GetBTenLeftCol(self): return int(self._info()[2])
    def GetBTenRightCol(self): return int(self._info()[3])

    def set_mpo(self, num):
        c = self.ctx
        c._ck(c._l.pepsgpu_walker_set_mpo(c._h, self.wid, num, None, None, 0))

    def set_mpo_states(self, num, states):
        c = self.ctx
        st = np.ascontiguousarray(states, dtype=np.int32)
        assert st.ndim == 2 and st.shape[0] == c.n
        c._ck(c._l.pepsgpu_walker_set_mpo(c._h, self.wid, num, _ip(st), None, 0))

    def set_mpo_tensors(self, num, tensors):
        """tensors [nt][N][D][D][D][D] (leg order L, D, R, U, zero padded to D), nt = 1 or n"""
        c = self.ctx
        t = np.ascontiguousarray(tensors, dtype=c._ot)
        assert t.ndim == 6 and t.shape[2:] == (c.D,) * 4
        c._ck(c._l.pepsgpu_walker_set_mpo(c._h, self.wid, num, None, _dp(t), t.shape[0]))

    def Evolve(self): self.ctx._ck(self.ctx._l.pepsgpu_walker_evolve(self.ctx._h, self.wid))
    def EvolveStep(self): self.ctx._ck(self.ctx._l.pepsgpu_walker_evolve_step(self.ctx._h, self.wid))

    def ContractRow(self, opp_level):
        c = self.ctx
        out = np.zeros(c.n, dtype=c._ot)
        c._ck(c._l.pepsgpu_walker_contract_row(c._h, self.wid, opp_level, _dp(out)))
        return out

    def InitBTenLeft(self, opp_level, target_col): self.ctx._ck(self.ctx._l.pepsgpu_walker_init_bten(self.ctx._h, self.wid, opp_level, LEFT, target_col))
    def InitBTenRight(self, opp_level, target_col): self.ctx._ck(self.ctx._l.pepsgpu_walker_init_bten(self.ctx._h, self.wid, opp_level, RIGHT, target_col))
    def GrowBTenLeftStep(self, opp_level): self.ctx._ck(self.ctx._l.pepsgpu_walker_grow_bten_step(self.ctx._h, self.wid, opp_level, LEFT))
    def GrowBTenRightStep(self, opp_level): self.ctx._ck(self.ctx._l.pepsgpu_walker_grow_bten_step(self.ctx._h, self.wid, opp_level, RIGHT))
    def ShiftBTenWindow(self, opp_level, position): self.ctx._ck(self.ctx._l.pepsgpu_walker_shift_bten_window(self.ctx._h, self.wid, opp_level, position))
    def ClearBTen(self): self.ctx._ck(self.ctx._l.pepsgpu_walker_clear_bten(self.ctx._h, self.wid))

    def _trace(self, opp_level, site_col, two, states, tensors):
        c = self.ctx
        out = np.zeros(c.n, dtype=c._ot)
        st = None if states is None else np.ascontiguousarray(states, dtype=np.int32)
        tt = None if tensors is None else np.ascontiguousarray(tensors, dtype=c._ot)
        nt = 0
        if tt is not None:
            tt = tt.reshape((-1,) + ((2,) if two else ()) + (c.D,) * 4)
            nt = tt.shape[0]
        c._ck(c._l.pepsgpu_walker_trace_with_bten(c._h, self.wid, opp_level, site_col, int(two), None if st is None else _ip(st),
                                                  None if tt is None else _dp(tt), nt, _dp(out)))
        return out

    def TraceWithBTen(self, opp_level, site_col, states=None, tensors=None):
        """states [n] (SITPS component per walker) or tensors [nt][D^4]; neither: the MPO's own tensor"""
        return self._trace(opp_level, site_col, False, states, tensors)

    def TraceWithTwoSiteBTen(self, opp_level, site_col, states=None, tensors=None):
        """states [n][2] or tensors [nt][2][D^4]"""
        return self._trace(opp_level, site_col, True, states, tensors)

    def GetBMPSTensor(self, idx):
        c = self.ctx
        dims = np.zeros(3, dtype=np.int32)
        c._ck(c._l.pepsgpu_walker_get_bmps_tensor(c._h, self.wid, idx, _ip(dims), None, None))
        data = np.zeros((c.n,) + tuple(int(x) for x in dims), dtype=c._ot)
        ls = np.zeros(c.n, dtype=np.float64)
        c._ck(c._l.pepsgpu_walker_get_bmps_tensor(c._h, self.wid, idx, _ip(dims), _dp(data), _dp(ls)))
        return data, ls

    def clone(self):
        """copy construction (`auto excited_walker = main_walker;`)"""
        wid = np.zeros(1, dtype=np.int32)
        self.ctx._ck(self.ctx._l.pepsgpu_walker_clone(self.ctx._h, self.wid, _ip(wid)))
        return Walker(self.ctx, int(wid[0]))

    def destroy(self):
        if self.wid is not None:
            self.ctx._ck(self.ctx._l.pepsgpu_walker_destroy(self.ctx._h, self.wid))
            self.wid = None

    # A walker is a deep copy of one boundary MPS for ALL Monte-Carlo walkers of the context: release it with the object, as the
    # destructor of the C++ BMPSWalker does.  The error code is ignored here -- set_configs / close of the context drop every walker
    # (a stale object then finds "no such walker"), and a context that is already closed has nothing left to free.
    def _release_quietly(self):
        wid, self.wid = self.wid, None
        ctx = self.ctx
        if wid is not None and getattr(ctx, "_h", None):
            try:
                ctx._l.pepsgpu_walker_destroy(ctx._h, wid)
            except Exception:
                pass

    def __del__(self):
        # The documented release paths are destroy() and the context manager.  A finalizer must not enter the engine (it may run on
        # another thread while a call on the same context is in flight: ctypes releases the GIL): it queues the id, the context frees
        # it at its next call from the owning thread (_ReapingLib) or drops it with close().
        try:
            wid, self.wid = self.wid, None
            if wid is not None and getattr(self.ctx, "_h", None):
                q = getattr(self.ctx, "_dead_walkers", None)
                if q is None:
                    q = self.ctx._dead_walkers = []
                q.append(wid)
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self._release_quietly()
        return False
