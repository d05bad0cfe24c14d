#!/bin/bash
# full GPU suite on the tree with the blocked-Cholesky v2, the branch-free GEMM operand loads and the unconditional factor-kernel loads
mkdir -p gpurun_out/r06
timeout 3000 python -m pytest tests -m gpu -q -x --tb=short 2>&1 | tail -8 | tee gpurun_out/r06/suite_call38.txt
