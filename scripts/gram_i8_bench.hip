// Microbenchmark + exactness check of the exact-integer Gram (peps_amd/csrc/gram_i8.h) against the float64-MFMA Gram it replaces
// (gram_cols_lds_kernel, peps_amd/csrc/gram.h).  Not part of the library.
//   ./gram_i8_bench [walkers = 2048] [rows = 1536] [reps = 5]
// P[b] = A . B with a graded B (cond ~ 1e6), per-walker live rows K_b in [rows - 200, rows]; n = 256 columns, every eighth walker
// with 28 of 32 live inner columns.  One JSON line per kernel: time per launch, and on two walkers
//   err_exact : max |G - G_ref| / sqrt(G_ii G_jj), G_ref = long-double Gram of the float32 data
//   err_image : the same against the long-double Gram of the fixed-point image of P (what the integer kernel computes exactly)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "../peps_amd/csrc/gram.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
using namespace pepsgpu;

int main(int argc, char **argv) {
  const int nw = argc > 1 ? atoi(argv[1]) : 2048, rows = argc > 2 ? atoi(argv[2]) : 1536, reps = argc > 3 ? atoi(argv[3]) : 5;
  const int n = 256, ld = 256, inner = 32;
  std::mt19937_64 rng(7);
  std::normal_distribution<float> nd(0.f, 1.f);
  // two distinct walkers on the host, the rest of the batch repeats them
  std::vector<float> Ph((size_t)2 * rows * ld);
  for (int w = 0; w < 2; ++w) {
    std::vector<float> A((size_t)rows * 64), B((size_t)64 * n);
    for (auto &x : A) x = nd(rng);
    for (int m = 0; m < 64; ++m)
      for (int jx = 0; jx < n; ++jx) B[(size_t)m * n + jx] = nd(rng) * std::pow(10.f, -6.f * m / 63.f) * (1.f + 3.f * (jx % 7));
    for (int k = 0; k < rows; ++k)
      for (int jx = 0; jx < n; ++jx) {
        double s = 0;
        for (int m = 0; m < 64; ++m) s += (double)A[(size_t)k * 64 + m] * B[(size_t)m * n + jx];
        Ph[((size_t)w * rows + k) * ld + jx] = (float)(s + 1e-7 * nd(rng));      // full numerical rank, graded
      }
  }
  std::vector<int> kd(nw), il(nw);
  for (int b = 0; b < nw; ++b) { kd[b] = rows - (int)(rng() % 200); il[b] = b % 8 == 7 ? 28 : 32; }
  kd[0] = rows; kd[1] = rows - 37; il[0] = 32; il[1] = 28;
  float *P; double *G; int *kdev, *ildev;
  const long wP = (long)rows * ld;
  CK(hipMalloc(&P, sizeof(float) * wP * nw)); CK(hipMalloc(&G, sizeof(double) * n * n * nw));
  CK(hipMalloc(&kdev, sizeof(int) * nw)); CK(hipMalloc(&ildev, sizeof(int) * nw));
  for (int b = 0; b < nw; ++b) CK(hipMemcpy(P + wP * b, Ph.data() + (size_t)(b & 1) * wP, sizeof(float) * wP, hipMemcpyHostToDevice));
  CK(hipMemcpy(kdev, kd.data(), sizeof(int) * nw, hipMemcpyHostToDevice));
  CK(hipMemcpy(ildev, il.data(), sizeof(int) * nw, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  // references (walkers 0 and 1)
  std::vector<long double> Gex((size_t)2 * n * n, 0.0L), Gim((size_t)2 * n * n, 0.0L);
  for (int w = 0; w < 2; ++w) {
    const int K = kd[w], ilv = il[w];
    std::vector<double> X((size_t)K * n), Y((size_t)K * n);
    for (int k0 = 0; k0 < K; k0 += 64)
      for (int jx = 0; jx < n; ++jx) {
        const bool ok = jx % inner < ilv;
        float m = 0;
        for (int k = k0; k < std::min(K, k0 + 64); ++k) m = std::max(m, ok ? std::fabs(Ph[((size_t)w * rows + k) * ld + jx]) : 0.f);
        unsigned mb; memcpy(&mb, &m, 4);
        const int e = std::max(30, (int)(mb >> 23));
        for (int k = k0; k < std::min(K, k0 + 64); ++k) {
          const float x = ok ? Ph[((size_t)w * rows + k) * ld + jx] : 0.f;
          X[(size_t)k * n + jx] = x;
          Y[(size_t)k * n + jx] = std::ldexp(std::nearbyint(std::ldexp((double)x, 148 - e)), e - 148);
        }
      }
    for (int i = 0; i < n; ++i)
      for (int jx = i; jx < n; ++jx) {
        long double s = 0, t = 0;
        for (int k = 0; k < K; ++k) { s += (long double)X[(size_t)k * n + i] * X[(size_t)k * n + jx]; t += (long double)Y[(size_t)k * n + i] * Y[(size_t)k * n + jx]; }
        Gex[((size_t)w * n + i) * n + jx] = s; Gim[((size_t)w * n + i) * n + jx] = t;
      }
  }
  std::vector<double> Gh((size_t)2 * n * n);
  for (int variant = 0; variant < 7; ++variant) {
    float ms = 0;
    for (int rep = -1; rep < reps; ++rep) {
      CK(hipMemset(G, 0xff, sizeof(double) * n * n * 2));
      CK(hipEventRecord(e0));
      if (variant == 0) {
        const size_t smem = gram_cols_lds_smem_bytes();
        allow_dynamic_lds(reinterpret_cast<const void *>(&gram_cols_lds_kernel<float>), smem);
        hipLaunchKernelGGL(gram_cols_lds_kernel<float>, dim3(nw), dim3(512), smem, 0, P, wP, n, ld, kdev, 1, rows, G, (long)n * n, nullptr, inner,
                           ildev, nullptr, nullptr, 1);
      } else {
        const size_t smem = gram_cols_i8_smem_bytes();
#define GI_LAUNCH(D) do { allow_dynamic_lds(reinterpret_cast<const void *>(&gram_cols_i8_kernel<float, false, D>), smem); \
        hipLaunchKernelGGL((gram_cols_i8_kernel<float, false, D>), dim3(nw), dim3(512), smem, 0, P, wP, n, ld, kdev, 1, rows, G, (long)n * n, n, nullptr, inner, \
                           ildev, nullptr, nullptr, nullptr, 1); } while (0)
        if (variant == 1) GI_LAUNCH(0); else if (variant == 2) GI_LAUNCH(1); else if (variant == 3) GI_LAUNCH(2); else if (variant == 4) GI_LAUNCH(4); else if (variant == 5) GI_LAUNCH(7);
        else {
          allow_dynamic_lds(reinterpret_cast<const void *>(&gram_cols_i8_kernel<float, false, 0, 12>), smem);
          hipLaunchKernelGGL((gram_cols_i8_kernel<float, false, 0, 12>), dim3(nw), dim3(768), smem, 0, P, wP, n, ld, kdev, 1, rows, G, (long)n * n, n, nullptr, inner,
                             ildev, nullptr, nullptr, nullptr, 1);
        }
      }
      CK(hipGetLastError());
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float t; CK(hipEventElapsedTime(&t, e0, e1));
      if (rep >= 0) ms += t;
    }
    ms /= reps;
    CK(hipMemcpy(Gh.data(), G, sizeof(double) * n * n * 2, hipMemcpyDeviceToHost));
    double eex = 0, eim = 0;
    for (int w = 0; w < 2; ++w)
      for (int i = 0; i < n; ++i)
        for (int jx = i; jx < n; ++jx) {
          const long double di = Gex[((size_t)w * n + i) * n + i], dj = Gex[((size_t)w * n + jx) * n + jx];
          const double g = Gh[((size_t)w * n + i) * n + jx];
          if (di <= 0 || dj <= 0) { if (g != 0.0) eex = eim = 1e300; continue; }
          const long double sc = sqrtl(di * dj);
          eex = std::max(eex, (double)(fabsl(g - Gex[((size_t)w * n + i) * n + jx]) / sc));
          eim = std::max(eim, (double)(fabsl(g - Gim[((size_t)w * n + i) * n + jx]) / sc));
        }
    double flop = 0;
    for (int b = 0; b < nw; ++b) flop += 2.0 * kd[b] * n * (n + 1) / 2;
    printf("{\"kernel\": \"%s\", \"walkers\": %d, \"rows\": %d, \"n\": %d, \"ms\": %.4f, \"tflops_upper_triangle\": %.2f, \"err_exact\": %.3e, \"err_image\": %.3e}\n",
           variant == 0 ? "gram_cols_lds_kernel (f64 MFMA)" : variant == 1 ? "gram_cols_i8_kernel (9 x i8 MFMA, exact)" : variant == 2 ? "timing only: no drain" : variant == 3 ? "timing only: no digit pass" : variant == 4 ? "timing only: no MFMA" : variant == 5 ? "timing only: loads + barriers" : "gram_cols_i8_kernel, 12 waves (three per SIMD)", nw, rows, n, ms, flop / ms * 1e-9, eex, eim);
    fflush(stdout);
  }
  // ---- row Gram (the truncation input): M = nr x 256, G = M M^T, nr live rows per walker ----
  {
    const int Kr = 256, ldg = 256;
    std::vector<float> Mh((size_t)2 * 256 * Kr);
    for (int w = 0; w < 2; ++w)
      for (int i = 0; i < 256; ++i)
        for (int k = 0; k < Kr; ++k) Mh[((size_t)w * 256 + i) * Kr + k] = Ph[((size_t)w * rows + k) * ld + i];     // M = P^T of the first 256 rows
    std::vector<int> nr(nw);
    for (int b = 0; b < nw; ++b) nr[b] = 256 - (int)(rng() % 64);
    nr[0] = 256; nr[1] = 203;
    float *Md; int *nrd;
    CK(hipMalloc(&Md, sizeof(float) * 256 * Kr * nw)); CK(hipMalloc(&nrd, sizeof(int) * nw));
    for (int b = 0; b < nw; ++b) CK(hipMemcpy(Md + (size_t)256 * Kr * b, Mh.data() + (size_t)(b & 1) * 256 * Kr, sizeof(float) * 256 * Kr, hipMemcpyHostToDevice));
    CK(hipMemcpy(nrd, nr.data(), sizeof(int) * nw, hipMemcpyHostToDevice));
    std::vector<long double> Gx((size_t)2 * 256 * 256, 0.0L);
    for (int w = 0; w < 2; ++w)
      for (int i = 0; i < nr[w]; ++i)
        for (int jx = i; jx < nr[w]; ++jx) {
          long double sx = 0;
          for (int k = 0; k < Kr; ++k) sx += (long double)Mh[((size_t)w * 256 + i) * Kr + k] * Mh[((size_t)w * 256 + jx) * Kr + k];
          Gx[((size_t)w * 256 + i) * 256 + jx] = sx;
        }
    for (int variant = 0; variant < 3; ++variant) {
      float ms = 0;
      for (int rep = -1; rep < reps; ++rep) {
        CK(hipMemset(G, 0xff, sizeof(double) * 256 * 256 * 2));
        CK(hipEventRecord(e0));
        if (variant == 0) {
          hipLaunchKernelGGL(gram_rows_f64_kernel<float>, dim3((10 + 3) / 4, nw), dim3(256), 0, 0, Md, (long)256 * Kr, Kr, nrd, G, (long)256 * 256, ldg, nullptr,
                             nullptr, nullptr);
        } else if (variant == 1) {
          const size_t smem = gram_cols_i8_smem_bytes();
          allow_dynamic_lds(reinterpret_cast<const void *>(&gram_cols_i8_kernel<float, true>), smem);
          hipLaunchKernelGGL((gram_cols_i8_kernel<float, true>), dim3(nw), dim3(512), smem, 0, Md, (long)256 * Kr, 256, Kr, nullptr, 1, Kr, G, (long)256 * 256, ldg,
                             nullptr, 1, nullptr, nrd, nullptr, nullptr, 1);
        } else {
          const size_t smem = gram_cols_i8_smem_bytes();
          allow_dynamic_lds(reinterpret_cast<const void *>(&gram_cols_i8_kernel<float, true, 0, 12>), smem);
          hipLaunchKernelGGL((gram_cols_i8_kernel<float, true, 0, 12>), dim3(nw), dim3(768), smem, 0, Md, (long)256 * Kr, 256, Kr, nullptr, 1, Kr, G, (long)256 * 256, ldg,
                             nullptr, 1, nullptr, nrd, nullptr, nullptr, 1);
        }
        CK(hipGetLastError());
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float t; CK(hipEventElapsedTime(&t, e0, e1));
        if (rep >= 0) ms += t;
      }
      ms /= reps;
      CK(hipMemcpy(Gh.data(), G, sizeof(double) * 256 * 256 * 2, hipMemcpyDeviceToHost));
      double eex = 0;
      for (int w = 0; w < 2; ++w)
        for (int i = 0; i < nr[w]; ++i)
          for (int jx = i; jx < nr[w]; ++jx) {
            const long double sc = sqrtl(Gx[((size_t)w * 256 + i) * 256 + i] * Gx[((size_t)w * 256 + jx) * 256 + jx]);
            eex = std::max(eex, (double)(fabsl(Gh[((size_t)w * 256 + i) * 256 + jx] - Gx[((size_t)w * 256 + i) * 256 + jx]) / sc));
          }
      printf("{\"kernel\": \"%s\", \"walkers\": %d, \"K\": %d, \"ms\": %.4f, \"err_exact\": %.3e}\n",
             variant == 0 ? "gram_rows_f64_kernel (f64 MFMA)" : variant == 1 ? "gram_cols_i8_kernel<ROWS> (9 x i8 MFMA)" : "gram_cols_i8_kernel<ROWS>, 12 waves", nw, Kr, ms, eex);
      fflush(stdout);
    }
  }
  return 0;
}
