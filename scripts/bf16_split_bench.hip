// Pricing the bf16 split-precision MFMA route for the dense absorption contractions (north_star:
// "absorption GEMMs on MFMA (bf16/fp32 as the element type demands)").  Standalone microbenchmark, not part of the library.
//
// Batched GEMM  C[b] (M x N) = A[b] (M x K) . Bt[b] (N x K)^T , both operands k-contiguous in HBM (the layout the chained
// contraction kernel controls: the intermediate X of stage 1 is laid down by the kernel itself), at the two shapes of one dense C4
// site step per walker:
//     stage 1  X = R . A        M = 1920 (m l), K = 32 (a),      N = 256  (p a2)
//     stage 2  P = W . X        M = 64 (l2 u),  K = 64 (l p),    N = 7680 (m a2)
// Variant F32 : v_mfma_f32_32x32x2_f32 (64 FLOP/clk/SIMD, the vector rate; what the library's kernels use today).
// Variant BF3 : every f32 operand split into three bf16 pieces x = hi + mid + lo (exact: 3 x 8 significand bits) while it is
//               staged into LDS; six products hi.hi, hi.mid, mid.hi, mid.mid, hi.lo, lo.hi on v_mfma_f32_32x32x16_bf16
//               (1024 FLOP/clk/SIMD), f32 accumulation; the dropped terms (mid.lo, lo.mid, lo.lo) are 2^-24 relative.
// Block = 256 threads = 4 waves, tile 64 (M) x 128 (N), K in LDS chunks of 32; wave w owns columns 32 w .. 32 w + 31 of
// the tile as two 32 x 32 accumulators.  Prints one JSON line per (shape, variant): TFLOP/s (2 M N K flops per entry) and the
// error against a float64 product of the same float32 operands, relative to |A||B| (max over the checked entries).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

constexpr int BM = 64, BN = 128, KC = 32;     // K walked in LDS chunks of 32

// ---------------------------------------------------------------------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(const float *__restrict__ Ag, const float *__restrict__ Bg, float *__restrict__ Cg,
                                                          int M, int N, int store, int kreps) {
  // LDS images k-major: sA[k][BM + 1], sB[k][BN + 1] (a lane reads row r of k-plane h: consecutive floats, conflict free)
  __shared__ float sA[KC][BM + 4];
  __shared__ float sB[KC][BN + 4];
  const int b = blockIdx.z, m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float *A = Ag + (long)b * M * K + (long)m0 * K;
  const float *B = Bg + (long)b * N * K + (long)n0 * K;
  const int r = lane & 31, h = lane >> 5;
  f32x16 acc0 = {0}, acc1 = {0};
  for (int rep = 0; rep < kreps; ++rep)
  for (int kc = 0; kc < K; kc += KC) {
    if (kc || rep) __syncthreads();
    // stage: float4 along k (k-contiguous operands), transposed into the k-major images
    for (int e = tid; e < BM * KC / 4; e += 256) {
      const int rr = e / (KC / 4), k4 = (e - rr * (KC / 4)) * 4;
      const float4 v = *reinterpret_cast<const float4 *>(A + (long)rr * K + kc + k4);
      sA[k4][rr] = v.x; sA[k4 + 1][rr] = v.y; sA[k4 + 2][rr] = v.z; sA[k4 + 3][rr] = v.w;
    }
    for (int e = tid; e < BN * KC / 4; e += 256) {
      const int rr = e / (KC / 4), k4 = (e - rr * (KC / 4)) * 4;
      const float4 v = *reinterpret_cast<const float4 *>(B + (long)rr * K + kc + k4);
      sB[k4][rr] = v.x; sB[k4 + 1][rr] = v.y; sB[k4 + 2][rr] = v.z; sB[k4 + 3][rr] = v.w;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < KC; k += 2) {
      const float bv = sB[k + h][32 * wave + r];
      const float a0 = sA[k + h][r], a1 = sA[k + h][32 + r];
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bv, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bv, acc1, 0, 0, 0);
    }
  }
  // C/D layout: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
  float *C = Cg + (long)b * M * N;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int row = (q & 3) + 8 * (q >> 2) + 4 * h;
    if (store || acc0[q] != acc0[q]) C[(long)(m0 + row) * N + n0 + 32 * wave + r] = acc0[q];
    if (store || acc1[q] != acc1[q]) C[(long)(m0 + 32 + row) * N + n0 + 32 * wave + r] = acc1[q];
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
  unsigned r;
  asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}
// x = hi + mid + lo exactly (three round-to-nearest bf16 pieces of the running remainder); two elements at a time
__device__ __forceinline__ void split3(float x0, float x1, unsigned &hi, unsigned &mid, unsigned &lo) {
  hi = cvt_pk_bf16(x0, x1);
  const float h0 = __uint_as_float(hi << 16), h1 = __uint_as_float(hi & 0xFFFF0000u);
  const float r0 = x0 - h0, r1 = x1 - h1;
  mid = cvt_pk_bf16(r0, r1);
  const float m0 = __uint_as_float(mid << 16), m1 = __uint_as_float(mid & 0xFFFF0000u);
  lo = cvt_pk_bf16(r0 - m0, r1 - m1);
}

template <int K, int NPROD>
__global__ __launch_bounds__(256, 2) void gemm_bf3_kernel(const float *__restrict__ Ag, const float *__restrict__ Bg, float *__restrict__ Cg,
                                                          int M, int N, int store, int kreps) {
  constexpr int LD = KC + 8;                        // bf16 elements per LDS row (+16 bytes: conflict-free 16-byte row reads)
  __shared__ __attribute__((aligned(16))) unsigned short sA[3][BM][LD];
  __shared__ __attribute__((aligned(16))) unsigned short sB[3][BN][LD];
  const int b = blockIdx.z, m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float *A = Ag + (long)b * M * K + (long)m0 * K;
  const float *B = Bg + (long)b * N * K + (long)n0 * K;
  const int r = lane & 31, h = lane >> 5;
  f32x16 acc0 = {0}, acc1 = {0};
  for (int rep = 0; rep < kreps; ++rep)
  for (int kc = 0; kc < K; kc += KC) {
  if (kc || rep) __syncthreads();
  for (int e = tid; e < BM * KC / 4; e += 256) {
    const int rr = e / (KC / 4), k4 = (e - rr * (KC / 4)) * 4;
    const float4 v = *reinterpret_cast<const float4 *>(A + (long)rr * K + kc + k4);
    unsigned h0, q0, l0, h1, q1, l1;
    split3(v.x, v.y, h0, q0, l0);
    split3(v.z, v.w, h1, q1, l1);
    *reinterpret_cast<uint2 *>(&sA[0][rr][k4]) = make_uint2(h0, h1);
    *reinterpret_cast<uint2 *>(&sA[1][rr][k4]) = make_uint2(q0, q1);
    *reinterpret_cast<uint2 *>(&sA[2][rr][k4]) = make_uint2(l0, l1);
  }
  for (int e = tid; e < BN * KC / 4; e += 256) {
    const int rr = e / (KC / 4), k4 = (e - rr * (KC / 4)) * 4;
    const float4 v = *reinterpret_cast<const float4 *>(B + (long)rr * K + kc + k4);
    unsigned h0, q0, l0, h1, q1, l1;
    split3(v.x, v.y, h0, q0, l0);
    split3(v.z, v.w, h1, q1, l1);
    *reinterpret_cast<uint2 *>(&sB[0][rr][k4]) = make_uint2(h0, h1);
    *reinterpret_cast<uint2 *>(&sB[1][rr][k4]) = make_uint2(q0, q1);
    *reinterpret_cast<uint2 *>(&sB[2][rr][k4]) = make_uint2(l0, l1);
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < KC; k += 16) {
    bf16x8 a0[3], a1[3], bb[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      a0[p] = *reinterpret_cast<const bf16x8 *>(&sA[p][r][k + 8 * h]);
      a1[p] = *reinterpret_cast<const bf16x8 *>(&sA[p][32 + r][k + 8 * h]);
      bb[p] = *reinterpret_cast<const bf16x8 *>(&sB[p][32 * wave + r][k + 8 * h]);
    }
    // smallest terms first
    if (NPROD >= 6) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[0], bb[2], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[0], bb[2], acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[2], bb[0], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[2], bb[0], acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[1], bb[1], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[1], bb[1], acc1, 0, 0, 0);
    }
    if (NPROD >= 3) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[0], bb[1], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[0], bb[1], acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[1], bb[0], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[1], bb[0], acc1, 0, 0, 0);
    }
    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[0], bb[0], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[0], bb[0], acc1, 0, 0, 0);
  }
  }
  float *C = Cg + (long)b * M * N;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int row = (q & 3) + 8 * (q >> 2) + 4 * h;
    if (store || acc0[q] != acc0[q]) C[(long)(m0 + row) * N + n0 + 32 * wave + r] = acc0[q];
    if (store || acc1[q] != acc1[q]) C[(long)(m0 + 32 + row) * N + n0 + 32 * wave + r] = acc1[q];
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
struct Shape { const char *name; int M, K, N; };

template <typename F>
static double time_ms(F &&launch, int reps) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  launch(); launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a, 0));
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(b, 0));
  CK(hipEventSynchronize(b));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, a, b));
  CK(hipGetLastError());
  return ms / reps;
}

int main(int argc, char **argv) {
  const int nb = argc > 1 ? atoi(argv[1]) : 1024, reps = argc > 2 ? atoi(argv[2]) : 10;
  const Shape shapes[2] = {{"stage1 X=R.A (1920 x 32 x 256)", 1920, 32, 256}, {"stage2 P=W.X (64 x 64 x 7680)", 64, 64, 7680}};
  for (const Shape &s : shapes) {
    const size_t na = (size_t)nb * s.M * s.K, nbt = (size_t)nb * s.N * s.K, nc = (size_t)nb * s.M * s.N;
    std::vector<float> hA(na), hB(nbt), hC(nc);
    // operands with the dynamic range of the real tensors: log-uniform magnitudes over three decades, random signs
    unsigned long long st = 88172645463325252ull;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (double)(st >> 11) / 9007199254740992.0; };
    for (auto &v : hA) v = (float)((rnd() < 0.5 ? -1 : 1) * pow(10.0, -3.0 * rnd()));
    for (auto &v : hB) v = (float)((rnd() < 0.5 ? -1 : 1) * pow(10.0, -3.0 * rnd()));
    float *dA, *dB, *dC;
    CK(hipMalloc(&dA, na * 4)); CK(hipMalloc(&dB, nbt * 4)); CK(hipMalloc(&dC, nc * 4));
    CK(hipMemcpy(dA, hA.data(), na * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, hB.data(), nbt * 4, hipMemcpyHostToDevice));
    const dim3 grid(s.N / BN, s.M / BM, nb);
    const double flops = 2.0 * s.M * s.N * s.K * nb;
    for (int mode = 0; mode < 2; ++mode)      // 0: the GEMM as it is (results to HBM); 1: compute only (K loop x 16 on resident operands, no store)
    for (int variant = 0; variant < 4; ++variant) {
      const int store = mode == 0, kreps = mode == 0 ? 1 : 16;
      auto launch = [&]() {
        if (s.K == 32) {
          if (variant == 0) hipLaunchKernelGGL((gemm_f32_kernel<32>), grid, dim3(256), 0, 0, dA, dB, dC, s.M, s.N, store, kreps);
          else if (variant == 1) hipLaunchKernelGGL((gemm_bf3_kernel<32, 6>), grid, dim3(256), 0, 0, dA, dB, dC, s.M, s.N, store, kreps);
          else if (variant == 2) hipLaunchKernelGGL((gemm_bf3_kernel<32, 3>), grid, dim3(256), 0, 0, dA, dB, dC, s.M, s.N, store, kreps);
          else hipLaunchKernelGGL((gemm_bf3_kernel<32, 1>), grid, dim3(256), 0, 0, dA, dB, dC, s.M, s.N, store, kreps);
        } else {
          if (variant == 0) hipLaunchKernelGGL((gemm_f32_kernel<64>), grid, dim3(256), 0, 0, dA, dB, dC, s.M, s.N, store, kreps);
          else if (variant == 1) hipLaunchKernelGGL((gemm_bf3_kernel<64, 6>), grid, dim3(256), 0, 0, dA, dB, dC, s.M, s.N, store, kreps);
          else if (variant == 2) hipLaunchKernelGGL((gemm_bf3_kernel<64, 3>), grid, dim3(256), 0, 0, dA, dB, dC, s.M, s.N, store, kreps);
          else hipLaunchKernelGGL((gemm_bf3_kernel<64, 1>), grid, dim3(256), 0, 0, dA, dB, dC, s.M, s.N, store, kreps);
        }
      };
      CK(hipMemset(dC, 0, nc * 4));
      const double ms = time_ms(launch, reps);
      const char *vn[4] = {"f32 mfma 32x32x2", "bf16 x3, 6 products", "bf16 x3, 3 products (hi.hi, hi.mid, mid.hi)", "bf16 x1 (hi.hi only)"};
      if (mode == 1) {
        printf("{\"shape\": \"%s\", \"variant\": \"%s\", \"mode\": \"compute only (staging + split + MFMA x16, no store)\", \"batch\": %d, "
               "\"ms\": %.4f, \"tflops\": %.2f}\n", s.name, vn[variant], nb, ms, flops * kreps / ms * 1e-9);
        fflush(stdout);
        continue;
      }
      CK(hipMemcpy(hC.data(), dC, nc * 4, hipMemcpyDeviceToHost));
      // error against float64 on a sample of entries of batch entries 0 and nb - 1
      double emax = 0.0, erms = 0.0;
      long cnt = 0;
      for (int bb : {0, nb - 1})
        for (int i = 0; i < s.M; i += 7)
          for (int j = 0; j < s.N; j += 13) {
            double ref = 0.0, mag = 0.0;
            for (int k = 0; k < s.K; ++k) {
              const double x = hA[((size_t)bb * s.M + i) * s.K + k], y = hB[((size_t)bb * s.N + j) * s.K + k];
              ref += x * y; mag += fabs(x * y);
            }
            const double e = fabs(hC[((size_t)bb * s.M + i) * s.N + j] - ref) / mag;
            emax = fmax(emax, e); erms += e * e; ++cnt;
          }
      printf("{\"shape\": \"%s\", \"variant\": \"%s\", \"mode\": \"gemm (operands from HBM, result to HBM)\", \"batch\": %d, \"ms\": %.4f, "
             "\"tflops\": %.2f, \"hbm_GBps\": %.0f, \"err_max_rel_to_abs_sum\": %.3e, \"err_rms\": %.3e}\n", s.name, vn[variant], nb, ms,
             flops / ms * 1e-9, 4.0 * (na + nbt + nc) / ms * 1e-6, emax, sqrt(erms / cnt));
      fflush(stdout);
    }
    CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC));
  }
  return 0;
}
