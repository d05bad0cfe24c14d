"""Round 6 measurement helper (not product): writes an instrumented copy of peps_amd/csrc into <dir> in which chol_blocked_kernel counts the
100 MHz wall clock per phase of a panel pair (update from the finished rows, staging, diagonal block, substitution, publish) and pepsgpu_diag_chol prints the per-walker means to stderr.  An optional number selects the k-steps of the update in flight.
usage: python scripts/chb_phase_patch.py <dir> [pfd]; then hipcc -shared ... -o peps_amd/lib/ab/<name>.so <dir>/capi.hip and PEPSGPU_LIB=<that>"""
import os, shutil, sys, glob
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = sys.argv[1]
pfd = int(sys.argv[2]) if len(sys.argv) > 2 else 0
os.makedirs(dst, exist_ok=True)
for f in glob.glob(os.path.join(ROOT, "peps_amd/csrc/*.h")) + [os.path.join(ROOT, "peps_amd/csrc/capi.hip")]:
    shutil.copy(f, dst)
os.makedirs(os.path.join(dst, "../../include"), exist_ok=True)
shutil.copy(os.path.join(ROOT, "include/pepsgpu.h"), os.path.join(dst, "../../include"))
p = os.path.join(dst, "linalg.h")
s = open(p).read()


def rep(a, b, cnt=1):
    global s
    lo = s.find("typedef double chb_f64x4")      # (the patterns are looked for from the kernel on; -1 in capi.hip: from the start)
    lo = max(lo, 0)
    assert a in s[lo:], a[:60]
    s = s[:lo] + s[lo:].replace(a, b, cnt)


rep("typedef double chb_f64x4 __attribute__((ext_vector_type(4)));",
    "typedef double chb_f64x4 __attribute__((ext_vector_type(4)));\n__device__ unsigned long long g_chb_phase[8];\n#define CHB_T() (tid == 0 ? wall_clock64() : 0ull)")
rep("    const int nprev = s_nlive;\n    // the panel's rows of G: requested now", "    const int nprev = s_nlive;\n    unsigned long long t0 = CHB_T(), t1;\n    // the panel's rows of G: requested now")
rep("    if (!dma && tid < n) {", "    __syncthreads();\n    t1 = CHB_T(); ph[0] += t1 - t0;\n    if (!dma && tid < n) {")
rep("    __syncthreads();\n    // ---- the 16 x 16 diagonal block, one wave, registers ----", "    __syncthreads();\n    t0 = CHB_T(); ph[1] += t0 - t1;\n    // ---- the 16 x 16 diagonal block, one wave, registers ----")
rep("    __syncthreads();\n    const unsigned livemask = s_livemask;", "    __syncthreads();\n    t1 = CHB_T(); ph[2] += t1 - t0;\n    const unsigned livemask = s_livemask;")
rep("    __syncthreads();\n    // publish the finished rows", "    __syncthreads();\n    t0 = CHB_T(); ph[3] += t0 - t1;\n    // publish the finished rows")
rep("    __threadfence_block();\n    __syncthreads();\n  }\n", "    __threadfence_block();\n    __syncthreads();\n    t1 = CHB_T(); ph[4] += t1 - t0;\n  }\n  if (tid == 0) for (int q = 0; q < 8; ++q) atomicAdd(&g_chb_phase[q], ph[q]);\n")
rep("  for (int jb = 0; jb < n; jb += CH_NB) {\n    const int nb = min(CH_NB, n - jb);\n    const int nprev = s_nlive;", "  unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};\n  const unsigned long long tk0 = CHB_T();\n  for (int jb = 0; jb < n; jb += CH_NB) {\n    const int nb = min(CH_NB, n - jb);\n    const int nprev = s_nlive;")
i = s.index("chol_blocked_kernel(double")
j = s.index("  if (mlive_out) return;\n  for (int e = tid + mlive * n; e < n * n; e += 256)", i)
s = s[:j] + "  if (tid == 0) { atomicAdd(&g_chb_phase[6], wall_clock64() - tk0); atomicAdd(&g_chb_phase[7], 1ull); }\n" + s[j:]
if pfd:
    rep("chol_blocked_kernel<T, 3, 4>", "chol_blocked_kernel<T, 3, %d>" % pfd, 2)      # (the launch instantiates four k-steps in flight)
open(p, "w").write(s)
p = os.path.join(dst, "capi.hip")
s = open(p).read()
rep("    if (dtype_out == 0) diag_chol_t<float>(G, n, nbatch, R_out); else diag_chol_t<double>(G, n, nbatch, R_out);\n  });",
    """    unsigned long long z[8] = {0,0,0,0,0,0,0,0}, h[8];
    PG_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_chb_phase), z, sizeof(z)));
    if (dtype_out == 0) diag_chol_t<float>(G, n, nbatch, R_out); else diag_chol_t<double>(G, n, nbatch, R_out);
    PG_CHECK_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_chb_phase), sizeof(h)));
    double nb = (double)h[7];
    fprintf(stderr, "[chb phases, us per walker (100 MHz clock)] update %.1f stage %.1f diag %.1f subst %.1f publish %.1f lds-update %.1f total %.1f (blocks %.0f)\\n",
            h[0] / nb / 100, h[1] / nb / 100, h[2] / nb / 100, h[3] / nb / 100, h[4] / nb / 100, h[5] / nb / 100, h[6] / nb / 100, nb);
  });""")
open(p, "w").write(s)
print("instrumented copy in", dst, "pfd", pfd or "default")
