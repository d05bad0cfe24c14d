"""Kernel-time experiment for the order-256 Cholesky kernels: one diag_chol call on `nb` graded Gram matrices (run under rocprofv3
--kernel-trace --stats --output-format csv; the environment selects the kernel: PEPSGPU_CHOL_RESIDENT; the timing-only switch PEPSGPU_CR_DBG of calls 42 / 46 was removed
from the kernel after the measurement)."""
import sys
import numpy as np
sys.path.insert(0, ".")
from peps_amd import capi
nb, n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048, 256
rng = np.random.default_rng(1)
X = rng.standard_normal((64, 300, n)).astype(np.float32).astype(np.float64)
G64 = np.einsum("bri,brj->bij", X, X)
G = np.ascontiguousarray(np.tile(G64, (nb // 64, 1, 1)))
for _ in range(3):
    R = capi.diag_chol(capi.F32, G)
print("ok", R.shape, float(np.abs(R).max()))
