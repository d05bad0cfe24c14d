// Stochastic-reconfiguration S-matrix on the device: the O* samples of the walkers stay in HBM
// and S v = <delta_i O*_i>, delta_i = O*_i . v - mean(O*) . v  (centred scalar projection,
// optimizer/stochastic_reconfiguration_smatrix.h:37-99) is two HBM-bound sweeps over them.
// An O* sample is stored the way the reference's SITPS-shaped sample is populated: one D^4 block per
// site, belonging to the component the walker's configuration selects (mc_energy_grad_evaluator.h:257-270),
// i.e. [sample][site][D^4] of T plus the configuration [sample][site].
#pragma once
#include "engine.h"

namespace pepsgpu {

// O*_i(site) = hole_i(site) / psi_i  from the resident hole store of the current walkers
template <typename T>
__global__ __launch_bounds__(256) void sr_append_kernel(const T *__restrict__ holes, const double *__restrict__ holes_ls,
                                                        const int *__restrict__ cfg, const double *__restrict__ logabs,
                                                        const double *__restrict__ sgn, T *__restrict__ o_out,
                                                        int *__restrict__ cfg_out, int sites, long slot,
                                                        const int *__restrict__ site_ne) {
  const int w = blockIdx.z, site = blockIdx.y;
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long base = ((long)w * sites + site) * slot;
  // only the first site_ne[site] elements of a slot hold the (compactly stored) hole; the rest is never written
  if (e < slot)
    o_out[base + e] = e < site_ne[site]
                          ? T(sgn[w] * exp(holes_ls[(long)w * sites + site] - logabs[w]) * (double)holes[base + e])
                          : T(0);
  if (e == 0) cfg_out[(long)w * sites + site] = cfg[(long)w * sites + site];
}

// delta[i] = sum_site < O*_i(site), v(site)[cfg_i(site)] > - shift - *shift_dev      (one block per sample)
template <typename T>
__global__ __launch_bounds__(256) void sr_delta_kernel(const T *__restrict__ o, const int *__restrict__ cfg,
                                                       const double *__restrict__ v, double shift, double *__restrict__ delta,
                                                       int sites, long slot, int dp, const double *__restrict__ shift_dev = nullptr) {
  __shared__ double s_red[4];
  const int i = blockIdx.x;
  double a = 0.0;
  for (int site = 0; site < sites; ++site) {
    const T *oi = o + ((long)i * sites + site) * slot;
    const double *vs = v + ((long)site * dp + cfg[(long)i * sites + site]) * slot;
    for (long e = threadIdx.x; e < slot; e += 256) a += (double)oi[e] * vs[e];
  }
  a = wave_sum(a);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) delta[i] = s_red[0] + s_red[1] + s_red[2] + s_red[3] - shift - (shift_dev ? *shift_dev : 0.0);
}

// ---- complex element type (TenElemT = QLTEN_Complex; SRSMatrix is templated over it, stochastic_reconfiguration_smatrix.h:36-99) ----
// O*_i(site) = conj(1 / psi_i) Dag(hole_i)(site) = conj(hole) * (psi / |psi|) / |psi|   (mc_energy_grad_evaluator.h:245-270)
template <typename T>
__global__ __launch_bounds__(256) void sr_append_cplx_kernel(const T *__restrict__ holes, const double *__restrict__ holes_ls,
                                                             const int *__restrict__ cfg, const double *__restrict__ logabs,
                                                             const double *__restrict__ ph_re, const double *__restrict__ ph_im,
                                                             T *__restrict__ o_out, int *__restrict__ cfg_out, int sites, long slot,
                                                             const int *__restrict__ site_ne) {
  const int w = blockIdx.z, site = blockIdx.y;
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long base = ((long)w * sites + site) * slot;
  if (e < slot) {
    T v = T(0);
    if (e < site_ne[site]) {
      const double f = exp(holes_ls[(long)w * sites + site] - logabs[w]);
      const double hr = (double)holes[base + e].re, hi = -(double)holes[base + e].im;            // Dag(hole)
      v = T(f * (hr * ph_re[w] - hi * ph_im[w]), f * (hr * ph_im[w] + hi * ph_re[w]));
    }
    o_out[base + e] = v;
  }
  if (e == 0) cfg_out[(long)w * sites + site] = cfg[(long)w * sites + site];
}
// delta[i] = < O*_i , v > - shift,  <a, b> = sum conj(a) b  (SplitIndexTPS::operator*, split_index_tps.h:370-377); interleaved pairs
template <typename T>
__global__ __launch_bounds__(256) void sr_delta_cplx_kernel(const T *__restrict__ o, const int *__restrict__ cfg,
                                                            const double *__restrict__ v, double shift_re, double shift_im,
                                                            double *__restrict__ delta, int sites, long slot, int dp,
                                                            const double *__restrict__ shift_dev = nullptr) {
  __shared__ double s_red[8];
  const int i = blockIdx.x;
  double ar = 0.0, ai = 0.0;
  for (int site = 0; site < sites; ++site) {
    const T *oi = o + ((long)i * sites + site) * slot;
    const double *vs = v + 2 * ((long)site * dp + cfg[(long)i * sites + site]) * slot;
    for (long e = threadIdx.x; e < slot; e += 256) {
      const double xr = (double)oi[e].re, xi = (double)oi[e].im, yr = vs[2 * e], yi = vs[2 * e + 1];
      ar += xr * yr + xi * yi;
      ai += xr * yi - xi * yr;
    }
  }
  ar = wave_sum(ar); ai = wave_sum(ai);
  if ((threadIdx.x & 63) == 0) { s_red[threadIdx.x >> 6] = ar; s_red[4 + (threadIdx.x >> 6)] = ai; }
  __syncthreads();
  if (threadIdx.x == 0) {
    delta[2 * i] = s_red[0] + s_red[1] + s_red[2] + s_red[3] - shift_re - (shift_dev ? shift_dev[0] : 0.0);
    delta[2 * i + 1] = s_red[4] + s_red[5] + s_red[6] + s_red[7] - shift_im - (shift_dev ? shift_dev[1] : 0.0);
  }
}
// out[site][s][e] = scale * sum_{i : cfg_i(site) == s} weight_i O*_i(site)[e], complex weights (nullptr: 1); interleaved pairs
template <typename T>
__global__ __launch_bounds__(256) void sr_accum_cplx_kernel(const T *__restrict__ o, const int *__restrict__ cfg,
                                                            const double *__restrict__ weight, double scale, double *__restrict__ out,
                                                            int n, int sites, long slot, int dp) {
  const int site = blockIdx.y;
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= slot) return;
  for (int s = 0; s < dp; ++s) {
    double ar = 0.0, ai = 0.0;
    for (int i = 0; i < n; ++i) {
      if (cfg[(long)i * sites + site] != s) continue;
      const double wr = weight ? weight[2 * i] : 1.0, wi = weight ? weight[2 * i + 1] : 0.0;
      const T x = o[((long)i * sites + site) * slot + e];
      ar += wr * (double)x.re - wi * (double)x.im;
      ai += wr * (double)x.im + wi * (double)x.re;
    }
    const long q = 2 * (((long)site * dp + s) * slot + e);
    out[q] = scale * ar; out[q + 1] = scale * ai;
  }
}

// part[blockIdx.x] = sum over the block's grid-stride share of a[e] b[e]; sr_dot_final_kernel adds the partials in a
// fixed order (no atomics: the CG scalars are reproducible run to run)
__global__ __launch_bounds__(256) void sr_dot_kernel(const double *__restrict__ a, const double *__restrict__ b, long n,
                                                     double *__restrict__ part) {
  __shared__ double s_red[4];
  double acc = 0.0;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) acc += a[e] * b[e];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = s_red[0] + s_red[1] + s_red[2] + s_red[3];
}
__global__ __launch_bounds__(64) void sr_dot_final_kernel(const double *__restrict__ part, int nblocks, double *__restrict__ out) {
  double acc = 0.0;
  for (int e = threadIdx.x; e < nblocks; e += 64) acc += part[e];
  acc = wave_sum(acc);
  if (threadIdx.x == 0) *out = acc;
}

// complex vectors as interleaved (re, im) pairs: part[2 b], part[2 b + 1] = the block's share of sum conj(a) b  (n complex entries)
__global__ __launch_bounds__(256) void sr_dotc_kernel(const double *__restrict__ a, const double *__restrict__ b, long n,
                                                      double *__restrict__ part) {
  __shared__ double s_red[8];
  double ar = 0.0, ai = 0.0;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
    const double xr = a[2 * e], xi = a[2 * e + 1], yr = b[2 * e], yi = b[2 * e + 1];
    ar += xr * yr + xi * yi;
    ai += xr * yi - xi * yr;
  }
  ar = wave_sum(ar); ai = wave_sum(ai);
  if ((threadIdx.x & 63) == 0) { s_red[threadIdx.x >> 6] = ar; s_red[4 + (threadIdx.x >> 6)] = ai; }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[2 * blockIdx.x] = s_red[0] + s_red[1] + s_red[2] + s_red[3];
    part[2 * blockIdx.x + 1] = s_red[4] + s_red[5] + s_red[6] + s_red[7];
  }
}
__global__ __launch_bounds__(64) void sr_dotc_final_kernel(const double *__restrict__ part, int nblocks, double *__restrict__ out) {
  double ar = 0.0, ai = 0.0;
  for (int e = threadIdx.x; e < nblocks; e += 64) { ar += part[2 * e]; ai += part[2 * e + 1]; }
  ar = wave_sum(ar); ai = wave_sum(ai);
  if (threadIdx.x == 0) { out[0] = ar; out[1] = ai; }
}
// y = alpha * x + beta * y, alpha complex, beta real (n complex entries)
__global__ __launch_bounds__(256) void sr_caxpby_kernel(double alpha_re, double alpha_im, const double *__restrict__ x, double beta,
                                                        double *__restrict__ y, long n) {
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
    const double xr = x[2 * e], xi = x[2 * e + 1];
    y[2 * e] = alpha_re * xr - alpha_im * xi + beta * y[2 * e];
    y[2 * e + 1] = alpha_re * xi + alpha_im * xr + beta * y[2 * e + 1];
  }
}

// y = alpha * x + beta * y
__global__ __launch_bounds__(256) void sr_axpby_kernel(double alpha, const double *__restrict__ x, double beta,
                                                       double *__restrict__ y, long n) {
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) y[e] = alpha * x[e] + beta * y[e];
}

// out[site][s][e] = scale * sum_{i : cfg_i(site) == s} weight_i O*_i(site)[e]   (weight == nullptr: 1)
template <typename T>
__global__ __launch_bounds__(256) void sr_accum_kernel(const T *__restrict__ o, const int *__restrict__ cfg,
                                                       const double *__restrict__ weight, double scale, double *__restrict__ out,
                                                       int n, int sites, long slot, int dp) {
  const int site = blockIdx.y;
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= slot) return;
  for (int s = 0; s < dp; ++s) {
    double a = 0.0;
    for (int i = 0; i < n; ++i) {
      if (cfg[(long)i * sites + site] != s) continue;
      a += (weight ? weight[i] : 1.0) * (double)o[((long)i * sites + site) * slot + e];
    }
    out[((long)site * dp + s) * slot + e] = scale * a;
  }
}

// state layout of the C ABI ([site][s][L][D][R][U], legs zero padded to D) <-> the compact per-slot layout the
// device keeps site tensors, holes and O* samples in (true leg dimensions, row-major inside the D^4 slot)
template <typename T>
void Engine<T>::sr_convert(const double *src, double *dst, bool to_compact) const {
  const size_t n = (size_t)Ly_ * Lx_ * dp_ * slot_ * kOut;      // complex: interleaved (re, im) pairs
  std::fill(dst, dst + n, 0.0);
  const size_t m = sr_map_c_.size();
  for (size_t k = 0; k < m; ++k) {
    const size_t from = kOut * (size_t)(to_compact ? sr_map_p_[k] : sr_map_c_[k]), to = kOut * (size_t)(to_compact ? sr_map_c_[k] : sr_map_p_[k]);
    for (int z = 0; z < kOut; ++z) dst[to + z] = src[from + z];
  }
}

template <typename T>
void Engine<T>::sr_begin(int max_samples) {
  {
  PG_REQUIRE(max_samples > 0, 1, "sr_begin: need a positive sample capacity");
  sr_release();
  const size_t sites = (size_t)Ly_ * Lx_;
  sr_o_ = (T *)arena_.alloc(sizeof(T) * (size_t)max_samples * sites * slot_);
  sr_cfg_ = (int *)arena_.alloc(sizeof(int) * (size_t)max_samples * sites);
  sr_delta_ = (double *)arena_.alloc(sizeof(double) * (size_t)max_samples * kOut);
  sr_v_ = (double *)arena_.alloc(sizeof(double) * sites * dp_ * slot_ * kOut);
  sr_out_ = (double *)arena_.alloc(sizeof(double) * sites * dp_ * slot_ * kOut);
  sr_map_c_.clear(); sr_map_p_.clear();
  for (int r = 0; r < Ly_; ++r)
    for (int c = 0; c < Lx_; ++c) {
      int dd[4];
      site_dims(r, c, dd);
      for (int s = 0; s < dp_; ++s) {
        const size_t base = ((size_t)(r * Lx_ + c) * dp_ + s) * slot_;
        size_t o = 0;
        for (int a = 0; a < dd[0]; ++a)
          for (int b = 0; b < dd[1]; ++b)
            for (int cc = 0; cc < dd[2]; ++cc)
              for (int e = 0; e < dd[3]; ++e, ++o) {
                sr_map_c_.push_back((uint32_t)(base + o));
                sr_map_p_.push_back((uint32_t)(base + (((size_t)a * D_ + b) * D_ + cc) * D_ + e));
              }
      }
    }
  sr_ne_ = (int *)arena_.alloc(sizeof(int) * sites);
  std::vector<int> ne(sites);
  for (int r = 0; r < Ly_; ++r)
    for (int c = 0; c < Lx_; ++c) {
      int dd[4];
      site_dims(r, c, dd);
      ne[r * Lx_ + c] = dd[0] * dd[1] * dd[2] * dd[3];
    }
  PG_CHECK_HIP(hipMemcpyAsync(sr_ne_, ne.data(), sizeof(int) * sites, hipMemcpyHostToDevice, stream_));
  PG_CHECK_HIP(hipStreamSynchronize(stream_));
  sr_cap_ = max_samples;
  sr_n_ = 0;
  }
}

template <typename T>
void Engine<T>::sr_release() {
  if (sr_o_) { arena_.free(sr_o_); arena_.free(sr_cfg_); arena_.free(sr_delta_); arena_.free(sr_v_); arena_.free(sr_out_); arena_.free(sr_ne_); }
  sr_ne_ = nullptr;
  sr_o_ = nullptr; sr_cfg_ = nullptr; sr_delta_ = nullptr; sr_v_ = nullptr; sr_out_ = nullptr;
  sr_cap_ = sr_n_ = 0;
}

template <typename T>
void Engine<T>::sr_append(const double *psi) {
  if constexpr (kCplx) {
    require_ready();
    PG_REQUIRE(sr_o_ != nullptr, 3, "sr_append: call pepsgpu_sr_begin first");
    PG_REQUIRE(holes_ != nullptr, 3, "sr_append: no holes stored (pepsgpu_punch_hole with out == NULL)");
    PG_REQUIRE(sr_n_ + nw_ <= sr_cap_, 1, "sr_append: sample store is full");
    std::vector<double> h(3 * (size_t)nw_);
    for (int w = 0; w < nw_; ++w) {
      const double a = std::hypot(psi[2 * w], psi[2 * w + 1]);
      PG_REQUIRE(a != 0.0, 5, "Wavefunction amplitude is near zero, causing division by zero.");
      h[w] = std::log(a); h[nw_ + w] = psi[2 * w] / a; h[2 * nw_ + w] = psi[2 * w + 1] / a;
    }
    ArenaScope scope(arena_);      // (d goes back to the arena when a HIP call below throws)
    double *d = (double *)arena_.alloc(sizeof(double) * h.size());
    PG_CHECK_HIP(hipMemcpyAsync(d, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice, stream_));
    const int sites = Ly_ * Lx_;
    hipLaunchKernelGGL(sr_append_cplx_kernel<T>, dim3((unsigned)((slot_ + 255) / 256), sites, nw_), dim3(256), 0, stream_,
                       (const T *)holes_, (const double *)holes_ls_, (const int *)cfg_, (const double *)d, (const double *)(d + nw_),
                       (const double *)(d + 2 * nw_), sr_o_ + (size_t)sr_n_ * sites * slot_, sr_cfg_ + (size_t)sr_n_ * sites, sites, slot_,
                       (const int *)sr_ne_);
    PG_CHECK_HIP(hipGetLastError());
    PG_CHECK_HIP(hipStreamSynchronize(stream_));
    arena_.free(d);
    sr_n_ += nw_;
  } else {
  require_ready();
  PG_REQUIRE(sr_o_ != nullptr, 3, "sr_append: call pepsgpu_sr_begin first");
  PG_REQUIRE(holes_ != nullptr, 3, "sr_append: no holes stored (pepsgpu_punch_hole with out == NULL)");
  PG_REQUIRE(sr_n_ + nw_ <= sr_cap_, 1, "sr_append: sample store is full");
  std::vector<double> h(2 * (size_t)nw_);
  for (int w = 0; w < nw_; ++w) {
    PG_REQUIRE(psi[w] != 0.0, 5, "Wavefunction amplitude is near zero, causing division by zero.");
    h[w] = std::log(std::fabs(psi[w]));
    h[nw_ + w] = psi[w] < 0 ? -1.0 : 1.0;
  }
  ArenaScope scope(arena_);
  double *d = (double *)arena_.alloc(sizeof(double) * h.size());
  PG_CHECK_HIP(hipMemcpyAsync(d, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice, stream_));
  const int sites = Ly_ * Lx_;
  hipLaunchKernelGGL(sr_append_kernel<T>, dim3((unsigned)((slot_ + 255) / 256), sites, nw_), dim3(256), 0, stream_,
                     (const T *)holes_, (const double *)holes_ls_, (const int *)cfg_, (const double *)d, (const double *)(d + nw_),
                     sr_o_ + (size_t)sr_n_ * sites * slot_, sr_cfg_ + (size_t)sr_n_ * sites, sites, slot_, (const int *)sr_ne_);
  PG_CHECK_HIP(hipGetLastError());
  PG_CHECK_HIP(hipStreamSynchronize(stream_));
  arena_.free(d);
  sr_n_ += nw_;
  }
}

template <typename T>
void Engine<T>::sr_sum(double *out) {
  if constexpr (kCplx) {
    PG_REQUIRE(sr_o_ != nullptr && sr_n_ > 0, 3, "sr_sum: no samples");
    const int sites = Ly_ * Lx_;
    const size_t n = (size_t)sites * dp_ * slot_ * 2;
    hipLaunchKernelGGL(sr_accum_cplx_kernel<T>, dim3((unsigned)((slot_ + 255) / 256), sites), dim3(256), 0, stream_, (const T *)sr_o_,
                       (const int *)sr_cfg_, (const double *)nullptr, 1.0, sr_out_, sr_n_, sites, slot_, dp_);
    PG_CHECK_HIP(hipGetLastError());
    std::vector<double> h(n);
    PG_CHECK_HIP(hipMemcpyAsync(h.data(), sr_out_, n * sizeof(double), hipMemcpyDeviceToHost, stream_));
    PG_CHECK_HIP(hipStreamSynchronize(stream_));
    sr_convert(h.data(), out, false);
  } else {
  PG_REQUIRE(sr_o_ != nullptr && sr_n_ > 0, 3, "sr_sum: no samples");
  const int sites = Ly_ * Lx_;
  const size_t n = (size_t)sites * dp_ * slot_;
  hipLaunchKernelGGL(sr_accum_kernel<T>, dim3((unsigned)((slot_ + 255) / 256), sites), dim3(256), 0, stream_, (const T *)sr_o_,
                     (const int *)sr_cfg_, (const double *)nullptr, 1.0, sr_out_, sr_n_, sites, slot_, dp_);
  PG_CHECK_HIP(hipGetLastError());
  std::vector<double> h(n);
  PG_CHECK_HIP(hipMemcpyAsync(h.data(), sr_out_, n * sizeof(double), hipMemcpyDeviceToHost, stream_));
  PG_CHECK_HIP(hipStreamSynchronize(stream_));
  sr_convert(h.data(), out, false);
  }
}

// out = scale * sum_i (O*_i . v - mean_dot_v) O*_i      (caller: all-reduce over ranks, + diag_shift * v)
template <typename T>
void Engine<T>::sr_matvec_cplx(const double *v, double mean_dot_v_re, double mean_dot_v_im, double scale, double *out) {
  if constexpr (kCplx) {
    PG_REQUIRE(sr_o_ != nullptr && sr_n_ > 0, 3, "sr_matvec: no samples");
    const int sites = Ly_ * Lx_;
    const size_t n = (size_t)sites * dp_ * slot_ * 2;
    std::vector<double> h(n);
    sr_convert(v, h.data(), true);
    PG_CHECK_HIP(hipMemcpyAsync(sr_v_, h.data(), n * sizeof(double), hipMemcpyHostToDevice, stream_));
    hipLaunchKernelGGL(sr_delta_cplx_kernel<T>, dim3(sr_n_), dim3(256), 0, stream_, (const T *)sr_o_, (const int *)sr_cfg_,
                       (const double *)sr_v_, mean_dot_v_re, mean_dot_v_im, sr_delta_, sites, slot_, dp_);
    PG_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(sr_accum_cplx_kernel<T>, dim3((unsigned)((slot_ + 255) / 256), sites), dim3(256), 0, stream_, (const T *)sr_o_,
                       (const int *)sr_cfg_, (const double *)sr_delta_, scale, sr_out_, sr_n_, sites, slot_, dp_);
    PG_CHECK_HIP(hipGetLastError());
    PG_CHECK_HIP(hipMemcpyAsync(h.data(), sr_out_, n * sizeof(double), hipMemcpyDeviceToHost, stream_));
    PG_CHECK_HIP(hipStreamSynchronize(stream_));
    sr_convert(h.data(), out, false);
  } else {
    PG_REQUIRE(mean_dot_v_im == 0.0, 1, "sr_matvec: a real context takes a real mean_dot_v");
    sr_matvec(v, mean_dot_v_re, scale, out);
  }
}

template <typename T>
void Engine<T>::sr_matvec(const double *v, double mean_dot_v, double scale, double *out) {
  if constexpr (kCplx) {
    sr_matvec_cplx(v, mean_dot_v, 0.0, scale, out);
  } else {
  PG_REQUIRE(sr_o_ != nullptr && sr_n_ > 0, 3, "sr_matvec: no samples");
  const int sites = Ly_ * Lx_;
  const size_t n = (size_t)sites * dp_ * slot_;
  std::vector<double> h(n);
  sr_convert(v, h.data(), true);
  PG_CHECK_HIP(hipMemcpyAsync(sr_v_, h.data(), n * sizeof(double), hipMemcpyHostToDevice, stream_));
  hipLaunchKernelGGL(sr_delta_kernel<T>, dim3(sr_n_), dim3(256), 0, stream_, (const T *)sr_o_, (const int *)sr_cfg_,
                     (const double *)sr_v_, mean_dot_v, sr_delta_, sites, slot_, dp_);
  PG_CHECK_HIP(hipGetLastError());
  hipLaunchKernelGGL(sr_accum_kernel<T>, dim3((unsigned)((slot_ + 255) / 256), sites), dim3(256), 0, stream_, (const T *)sr_o_,
                     (const int *)sr_cfg_, (const double *)sr_delta_, scale, sr_out_, sr_n_, sites, slot_, dp_);
  PG_CHECK_HIP(hipGetLastError());
  PG_CHECK_HIP(hipMemcpyAsync(h.data(), sr_out_, n * sizeof(double), hipMemcpyDeviceToHost, stream_));
  PG_CHECK_HIP(hipStreamSynchronize(stream_));
  sr_convert(h.data(), out, false);
  }
}

// ---------------------------------------------------------------------------------------------
// Device-resident natural-gradient solve: ConjugateGradientSolver (utility/conjugate_gradient_solver.h:181-276)
// on (S + diag_shift) x = b with every vector in HBM next to the samples (compact layout, f64).  One iteration =
// two sweeps over the sample store (S p) + a few axpy / dot kernels; the host reads back the handful of scalars
// its branches need (p.Ap, |p|^2 | |x|^2, |r|^2, r_prev.r) twice per iteration.  Every branch of the reference is
// kept: indefinite-matrix exit, stagnation detection, periodic residual recomputation, NaN/Inf exits, best-iterate
// tracking, orthogonality-based restart.
template <typename T>
void Engine<T>::sr_cg_solve(const double *b, const double *x0, double diag_shift, int max_iter, double rel_tol,
                            double abs_tol, int recompute_interval, double ortho_threshold, double *x_out,
                            double *residual_norm, int *iterations, int *reason) {
  if constexpr (kCplx) {
    // TenElemT = QLTEN_Complex (round 4): the same solver on interleaved (re, im) vectors.  a * b = sum conj(a) b, NormSquare real,
    // p * (A p) complex and valid when Re > 0 and |Im| < 1e-10 (detail::pap_is_valid, :142-148), alpha = rk / pap complex, beta real,
    // the restart test takes the real part of r_prev * r (:259-266).
    PG_REQUIRE(sr_o_ != nullptr && sr_n_ > 0, 3, "sr_cg_solve: no samples");
    ArenaScope scope(arena_);
    const int sites = Ly_ * Lx_;
    const long n = (long)sites * dp_ * slot_, n2 = 2 * n;
    const double scale = 1.0 / sr_n_;
    auto dvec = [&]() { return (double *)arena_.alloc(sizeof(double) * n2); };
    double *db = dvec(), *dx = dvec(), *dr = dvec(), *dp = dvec(), *dap = dvec(), *dbest = dvec(), *dprev = dvec(), *dmean = dvec();
    double *dsc = (double *)arena_.alloc(sizeof(double) * 16);
    constexpr int DOT_BLOCKS = 512;
    double *dpart = (double *)arena_.alloc(sizeof(double) * 2 * DOT_BLOCKS);
    auto nsq_into = [&](const double *a, double *out) {          // sum |a|^2: a real dot over the 2 n doubles
      hipLaunchKernelGGL(sr_dot_kernel, dim3(DOT_BLOCKS), dim3(256), 0, stream_, a, a, n2, dpart);
      hipLaunchKernelGGL(sr_dot_final_kernel, dim3(1), dim3(64), 0, stream_, (const double *)dpart, DOT_BLOCKS, out);
    };
    auto dotc_into = [&](const double *a, const double *b2, double *out) {
      hipLaunchKernelGGL(sr_dotc_kernel, dim3(DOT_BLOCKS), dim3(256), 0, stream_, a, b2, n, dpart);
      hipLaunchKernelGGL(sr_dotc_final_kernel, dim3(1), dim3(64), 0, stream_, (const double *)dpart, DOT_BLOCKS, out);
    };
    std::vector<double> h(n2);
    auto upload = [&](const double *src, double *dst) {
      if (src) {
        sr_convert(src, h.data(), true);
        PG_CHECK_HIP(hipMemcpyAsync(dst, h.data(), n2 * sizeof(double), hipMemcpyHostToDevice, stream_));
        PG_CHECK_HIP(hipStreamSynchronize(stream_));   // h is reused
      } else PG_CHECK_HIP(hipMemsetAsync(dst, 0, n2 * sizeof(double), stream_));
    };
    auto copy = [&](double *dst, const double *src) {
      PG_CHECK_HIP(hipMemcpyAsync(dst, src, n2 * sizeof(double), hipMemcpyDeviceToDevice, stream_));
    };
    const dim3 vg((unsigned)std::min<long>((n + 255) / 256, 2048)), vg2((unsigned)std::min<long>((n2 + 255) / 256, 2048));
    auto axpby = [&](double alpha, const double *x, double beta, double *y) {       // real scalars: element-wise on the doubles
      hipLaunchKernelGGL(sr_axpby_kernel, vg2, dim3(256), 0, stream_, alpha, x, beta, y, n2);
    };
    auto caxpby = [&](double ar, double ai, const double *x, double beta, double *y) {
      hipLaunchKernelGGL(sr_caxpby_kernel, vg, dim3(256), 0, stream_, ar, ai, x, beta, y, n);
    };
    auto read = [&](double *out, int k) {
      PG_CHECK_HIP(hipMemcpyAsync(out, dsc, sizeof(double) * k, hipMemcpyDeviceToHost, stream_));
      PG_CHECK_HIP(hipStreamSynchronize(stream_));
    };
    auto matvec = [&](const double *v, double *out) {
      dotc_into(dmean, v, dsc + 14);   // Ostar_mean * v = sum conj(mean) v
      hipLaunchKernelGGL(sr_delta_cplx_kernel<T>, dim3(sr_n_), dim3(256), 0, stream_, (const T *)sr_o_, (const int *)sr_cfg_, v, 0.0, 0.0,
                         sr_delta_, sites, slot_, dp_, (const double *)(dsc + 14));
      hipLaunchKernelGGL(sr_accum_cplx_kernel<T>, dim3((unsigned)((slot_ + 255) / 256), sites), dim3(256), 0, stream_, (const T *)sr_o_,
                         (const int *)sr_cfg_, (const double *)sr_delta_, scale, out, sr_n_, sites, slot_, dp_);
      PG_CHECK_HIP(hipGetLastError());
      if (diag_shift != 0.0) axpby(diag_shift, v, 1.0, out);
    };
    auto finish = [&](const double *xsrc, double res_sq, int iters, int why) {
      PG_CHECK_HIP(hipMemcpyAsync(h.data(), xsrc, n2 * sizeof(double), hipMemcpyDeviceToHost, stream_));
      PG_CHECK_HIP(hipStreamSynchronize(stream_));
      sr_convert(h.data(), x_out, false);
      *residual_norm = std::sqrt(res_sq); *iterations = iters; *reason = why;
      for (double *v : {db, dx, dr, dp, dap, dbest, dprev, dmean, dsc, dpart}) arena_.free(v);
    };
    enum { kConverged = 0, kMaxIterations = 1, kIndefiniteMatrix = 2, kNumericalBreakdown = 3, kStagnated = 4 };
    hipLaunchKernelGGL(sr_accum_cplx_kernel<T>, dim3((unsigned)((slot_ + 255) / 256), sites), dim3(256), 0, stream_, (const T *)sr_o_,
                       (const int *)sr_cfg_, (const double *)nullptr, scale, dmean, sr_n_, sites, slot_, dp_);
    upload(b, db);
    upload(x0, dx);
    double s[8];
    matvec(dx, dap);
    copy(dr, db);
    axpby(-1.0, dap, 1.0, dr);                      // r = b - A x0
    nsq_into(db, dsc); nsq_into(dr, dsc + 1); read(s, 2);
    const double tol_sq = std::max(rel_tol * rel_tol * s[0], abs_tol * abs_tol);
    double r_norm_sq = s[1];
    if (r_norm_sq <= tol_sq) return finish(dx, r_norm_sq, 0, kConverged);
    copy(dp, dr); copy(dbest, dx); copy(dprev, dr);
    double best = r_norm_sq, rkp1 = r_norm_sq;
    int stagnation = 0;
    const double eps = 2.220446049250313e-16;
    for (int k = 0; k < max_iter; ++k) {
      const double rk = rkp1;
      matvec(dp, dap);
      dotc_into(dp, dap, dsc); nsq_into(dp, dsc + 2); read(s, 3);
      const double pap_re = s[0], pap_im = s[1], pp = s[2];
      if (!(pap_re > 0.0 && std::fabs(pap_im) < 1e-10)) return finish(dbest, best, k, kIndefiniteMatrix);   // (+inf passes, as in the reference: the NaN it makes of the residual is the kNumericalBreakdown exit below)
      const double den = pap_re * pap_re + pap_im * pap_im;
      const double a_re = rk * pap_re / den, a_im = -rk * pap_im / den;      // alpha = rk / pap
      caxpby(a_re, a_im, dp, 1.0, dx);
      if (recompute_interval > 0 && (k % recompute_interval) == recompute_interval - 1) {
        matvec(dx, dr);
        axpby(1.0, db, -1.0, dr);                   // r = b - A x
      } else caxpby(-a_re, -a_im, dap, 1.0, dr);
      nsq_into(dx, dsc); nsq_into(dr, dsc + 1); dotc_into(dprev, dr, dsc + 2); read(s, 4);
      if ((a_re * a_re + a_im * a_im) * pp < eps * eps * s[0]) {
        if (++stagnation >= 3) return finish(dbest, best, k + 1, kStagnated);
      } else stagnation = 0;
      rkp1 = s[1];
      if (!std::isfinite(rkp1)) return finish(dbest, best, k + 1, kNumericalBreakdown);
      if (rkp1 < best) { copy(dbest, dx); best = rkp1; }
      if (rkp1 <= tol_sq) return finish(dx, rkp1, k + 1, kConverged);
      if (k > 0 && std::fabs(s[2]) > ortho_threshold * rkp1) {   // orthogonality-based restart (real part of r_prev * r)
        copy(dp, dr); copy(dprev, dr);
        continue;
      }
      copy(dprev, dr);
      const double beta = rkp1 / rk;
      if (!std::isfinite(beta)) return finish(dbest, best, k + 1, kNumericalBreakdown);
      axpby(1.0, dr, beta, dp);                      // p = r + beta p
    }
    finish(dbest, best, max_iter, kMaxIterations);
  } else {
  PG_REQUIRE(sr_o_ != nullptr && sr_n_ > 0, 3, "sr_cg_solve: no samples");
  ArenaScope scope(arena_);
  const int sites = Ly_ * Lx_;
  const long n = (long)sites * dp_ * slot_;
  const double scale = 1.0 / sr_n_;
  auto dvec = [&]() { return (double *)arena_.alloc(sizeof(double) * n); };
  double *db = dvec(), *dx = dvec(), *dr = dvec(), *dp = dvec(), *dap = dvec(), *dbest = dvec(), *dprev = dvec(), *dmean = dvec();
  double *dsc = (double *)arena_.alloc(sizeof(double) * 8);
  constexpr int DOT_BLOCKS = 512;
  double *dpart = (double *)arena_.alloc(sizeof(double) * DOT_BLOCKS);
  auto dot_into = [&](const double *a, const double *b2, double *out) {
    hipLaunchKernelGGL(sr_dot_kernel, dim3(DOT_BLOCKS), dim3(256), 0, stream_, a, b2, n, dpart);
    hipLaunchKernelGGL(sr_dot_final_kernel, dim3(1), dim3(64), 0, stream_, (const double *)dpart, DOT_BLOCKS, out);
  };
  std::vector<double> h(n);
  auto upload = [&](const double *src, double *dst) {
    if (src) {
      sr_convert(src, h.data(), true);
      PG_CHECK_HIP(hipMemcpyAsync(dst, h.data(), n * sizeof(double), hipMemcpyHostToDevice, stream_));
      PG_CHECK_HIP(hipStreamSynchronize(stream_));   // h is reused
    } else PG_CHECK_HIP(hipMemsetAsync(dst, 0, n * sizeof(double), stream_));
  };
  auto copy = [&](double *dst, const double *src) {
    PG_CHECK_HIP(hipMemcpyAsync(dst, src, n * sizeof(double), hipMemcpyDeviceToDevice, stream_));
  };
  const dim3 vg((unsigned)std::min<long>((n + 255) / 256, 2048));
  auto axpby = [&](double alpha, const double *x, double beta, double *y) {
    hipLaunchKernelGGL(sr_axpby_kernel, vg, dim3(256), 0, stream_, alpha, x, beta, y, n);
  };
  // dots[k] = a_k . b_k for up to four pairs, one read-back
  auto dots = [&](std::initializer_list<std::pair<const double *, const double *>> pairs, double *out) {
    int k = 0;
    for (auto &pr : pairs) dot_into(pr.first, pr.second, dsc + k++);
    PG_CHECK_HIP(hipMemcpyAsync(out, dsc, sizeof(double) * k, hipMemcpyDeviceToHost, stream_));
    PG_CHECK_HIP(hipStreamSynchronize(stream_));
  };
  // out = S v + diag_shift v     (SRSMatrix::operator*, stochastic_reconfiguration_smatrix.h:37-99)
  auto matvec = [&](const double *v, double *out) {
    dot_into(dmean, v, dsc + 7);   // Ostar_mean . v
    hipLaunchKernelGGL(sr_delta_kernel<T>, dim3(sr_n_), dim3(256), 0, stream_, (const T *)sr_o_, (const int *)sr_cfg_, v, 0.0,
                       sr_delta_, sites, slot_, dp_, (const double *)(dsc + 7));
    hipLaunchKernelGGL(sr_accum_kernel<T>, dim3((unsigned)((slot_ + 255) / 256), sites), dim3(256), 0, stream_, (const T *)sr_o_,
                       (const int *)sr_cfg_, (const double *)sr_delta_, scale, out, sr_n_, sites, slot_, dp_);
    PG_CHECK_HIP(hipGetLastError());
    if (diag_shift != 0.0) axpby(diag_shift, v, 1.0, out);
  };
  auto finish = [&](const double *xsrc, double res_sq, int iters, int why) {
    PG_CHECK_HIP(hipMemcpyAsync(h.data(), xsrc, n * sizeof(double), hipMemcpyDeviceToHost, stream_));
    PG_CHECK_HIP(hipStreamSynchronize(stream_));
    sr_convert(h.data(), x_out, false);
    *residual_norm = std::sqrt(res_sq); *iterations = iters; *reason = why;
    for (double *v : {db, dx, dr, dp, dap, dbest, dprev, dmean, dsc, dpart}) arena_.free(v);
  };
  enum { kConverged = 0, kMaxIterations = 1, kIndefiniteMatrix = 2, kNumericalBreakdown = 3, kStagnated = 4 };

  // Ostar_mean
  hipLaunchKernelGGL(sr_accum_kernel<T>, dim3((unsigned)((slot_ + 255) / 256), sites), dim3(256), 0, stream_, (const T *)sr_o_,
                     (const int *)sr_cfg_, (const double *)nullptr, scale, dmean, sr_n_, sites, slot_, dp_);
  upload(b, db);
  upload(x0, dx);
  double s[4];
  matvec(dx, dap);
  copy(dr, db);
  axpby(-1.0, dap, 1.0, dr);                      // r = b - A x0
  dots({{db, db}, {dr, dr}}, s);
  const double tol_sq = std::max(rel_tol * rel_tol * s[0], abs_tol * abs_tol);
  double r_norm_sq = s[1];
  if (r_norm_sq <= tol_sq) return finish(dx, r_norm_sq, 0, kConverged);
  copy(dp, dr); copy(dbest, dx); copy(dprev, dr);
  double best = r_norm_sq, rkp1 = r_norm_sq;
  int stagnation = 0;
  const double eps = 2.220446049250313e-16;
  for (int k = 0; k < max_iter; ++k) {
    const double rk = rkp1;
    matvec(dp, dap);
    dots({{dp, dap}, {dp, dp}}, s);
    const double pap = s[0], pp = s[1];
    if (!(pap > 0.0)) return finish(dbest, best, k, kIndefiniteMatrix);   // detail::pap_is_valid (:142-144): +inf passes, NaN does not
    const double alpha = rk / pap;
    axpby(alpha, dp, 1.0, dx);
    if (recompute_interval > 0 && (k % recompute_interval) == recompute_interval - 1) {
      matvec(dx, dr);
      axpby(1.0, db, -1.0, dr);                   // r = b - A x
    } else axpby(-alpha, dap, 1.0, dr);
    dots({{dx, dx}, {dr, dr}, {dprev, dr}}, s);
    if (alpha * alpha * pp < eps * eps * s[0]) {
      if (++stagnation >= 3) return finish(dbest, best, k + 1, kStagnated);
    } else stagnation = 0;
    rkp1 = s[1];
    if (!std::isfinite(rkp1)) return finish(dbest, best, k + 1, kNumericalBreakdown);
    if (rkp1 < best) { copy(dbest, dx); best = rkp1; }
    if (rkp1 <= tol_sq) return finish(dx, rkp1, k + 1, kConverged);
    if (k > 0 && std::fabs(s[2]) > ortho_threshold * rkp1) {   // orthogonality-based restart
      copy(dp, dr); copy(dprev, dr);
      continue;
    }
    copy(dprev, dr);
    const double beta = rkp1 / rk;
    if (!std::isfinite(beta)) return finish(dbest, best, k + 1, kNumericalBreakdown);
    axpby(1.0, dr, beta, dp);                      // p = r + beta p
  }
  finish(dbest, best, max_iter, kMaxIterations);
  }
}

// ---------------------------------------------------------------------------------------------
// MinSR building blocks (optimizer/minsr_tmatrix.h, optimizer_impl.h:1126-1215): raw Gram blocks of O* samples as one
// tensor GEMM over the compact parameter index (f64 accumulation), the weighted sum sum_i y_i O*_i, and a
// device-to-device copy of the local samples for the ring exchange of the multi-rank T matrix.
// A sample is expanded by its configuration into the SITPS-shaped vector the reference holds
// (component cfg_i(site) of every site, zeros elsewhere): the Gram block is then a plain GEMM over K = sites * d * D^4.
template <typename T>
__global__ __launch_bounds__(256) void sr_expand_kernel(const T *__restrict__ o, const int *__restrict__ cfg, T *__restrict__ out,
                                                        int sites, long slot, int dp, const int *__restrict__ site_ne) {
  const int i = blockIdx.z, site = blockIdx.y;
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= slot) return;
  const int c = cfg[(long)i * sites + site];
  const T v = e < site_ne[site] ? o[((long)i * sites + site) * slot + e] : T(0);
  for (int s = 0; s < dp; ++s) out[(((long)i * sites + site) * dp + s) * slot + e] = s == c ? v : T(0);
}

template <typename T>
void Engine<T>::sr_gram(const void *remote_o, const int32_t *remote_cfg, int n_remote, double *out) {
  if constexpr (kCplx) {
    // TenElemT = QLTEN_Complex (MinSRTMatrix is templated over it, minsr_tmatrix.h:38-150): ip_ij = O*_i * O*_j = sum conj(O*_i) O*_j
    // (SplitIndexTPS::operator*), one complex tensor GEMM with the A operand conjugated; out = interleaved (re, im) pairs [n][nb][2]
    PG_REQUIRE(sr_o_ != nullptr && sr_n_ > 0, 3, "sr_gram: no samples");
    PG_REQUIRE(!remote_o || (remote_cfg && n_remote > 0), 1, "sr_gram: bad remote batch");
    const int sites = Ly_ * Lx_;
    const long n = (long)sites * dp_ * slot_;
    PG_REQUIRE(n < (1l << 30), 1, "sr_gram: parameter count exceeds the 32-bit strides of the tensor GEMM");
    const int nb = remote_o ? n_remote : sr_n_;
    auto expand = [&](const T *o, const int *cfg, int cnt) {
      T *buf = (T *)arena_.alloc(sizeof(T) * (size_t)cnt * n);
      hipLaunchKernelGGL(sr_expand_kernel<T>, dim3((unsigned)((slot_ + 255) / 256), sites, cnt), dim3(256), 0, stream_, o, cfg, buf,
                         sites, slot_, dp_, (const int *)sr_ne_);
      PG_CHECK_HIP(hipGetLastError());
      return buf;
    };
    T *ea = expand((const T *)sr_o_, (const int *)sr_cfg_, sr_n_);
    T *eb = remote_o ? expand((const T *)remote_o, (const int *)remote_cfg, nb) : ea;
    const int bs = (int)std::min<long>(1024, ((1l << 30) - 1) / n) & ~63;
    PG_REQUIRE(bs >= 64, 1, "sr_gram: parameter count too large for the blocked Gram");
    const int groups = sites * dp_;
    std::vector<double> h(2 * (size_t)sr_n_ * nb, 0.0), part;
    for (int ib = 0; ib < sr_n_; ib += bs) {
      const int ni = std::min(bs, sr_n_ - ib);
      for (int jb = remote_o ? 0 : ib; jb < nb; jb += bs) {
        const int nj = std::min(bs, nb - jb);
        const long tiles = (long)((ni + 63) / 64) * ((nj + 63) / 64);
        int split = 1;
        for (int c = 1; c <= groups; ++c)
          if (groups % c == 0 && tiles * c <= 2048) split = c;
        const long chunk = n / split;
        T *d = (T *)arena_.alloc(sizeof(T) * (size_t)split * ni * nj);
        PG_CHECK_HIP(hipMemsetAsync(d, 0, sizeof(T) * (size_t)split * ni * nj, stream_));
        TGemmDesc g;
        g.I[2] = ni; g.sAi[2] = (int)n; g.sCi[2] = nj;
        g.K[2] = (int)chunk; g.sAk[2] = 1; g.sBk[2] = 1;
        g.J[2] = nj; g.sBj[2] = (int)n; g.sCj[2] = 1;
        g.wA = chunk; g.wB = chunk; g.wC = (long)ni * nj;
        g.nbatch = split;
        g.conjA = 1;
        tgemm_launch<T, T, T, T>(stream_, g, ea + (size_t)ib * n, eb + (size_t)jb * n, d);
        part.resize(2 * (size_t)split * ni * nj);
        PG_CHECK_HIP(hipMemcpyAsync(part.data(), d, sizeof(double) * part.size(), hipMemcpyDeviceToHost, stream_));
        PG_CHECK_HIP(hipStreamSynchronize(stream_));
        arena_.free(d);
        for (int sp = 0; sp < split; ++sp)
          for (int i2 = 0; i2 < ni; ++i2) {
            const double *src = &part[2 * (((size_t)sp * ni + i2) * nj)];
            double *dst = &h[2 * ((size_t)(ib + i2) * nb + jb)];
            for (int j2 = 0; j2 < 2 * nj; ++j2) dst[j2] += src[j2];
          }
      }
    }
    if (!remote_o)   // blocks below the diagonal were not computed: ip_ij = conj(ip_ji); ip_ii = sum |O*_i|^2 is real
      for (int i2 = 0; i2 < sr_n_; ++i2) {
        h[2 * ((size_t)i2 * nb + i2) + 1] = 0.0;
        for (int j2 = 0; j2 < i2; ++j2) {
          h[2 * ((size_t)i2 * nb + j2)] = h[2 * ((size_t)j2 * nb + i2)];
          h[2 * ((size_t)i2 * nb + j2) + 1] = -h[2 * ((size_t)j2 * nb + i2) + 1];
        }
      }
    std::copy(h.begin(), h.end(), out);
    if (eb != ea) arena_.free(eb);
    arena_.free(ea);
  } else {
  PG_REQUIRE(sr_o_ != nullptr && sr_n_ > 0, 3, "sr_gram: no samples");
  PG_REQUIRE(!remote_o || (remote_cfg && n_remote > 0), 1, "sr_gram: bad remote batch");
  const int sites = Ly_ * Lx_;
  const long n = (long)sites * dp_ * slot_;
  PG_REQUIRE(n < (1l << 31), 1, "sr_gram: parameter count exceeds the 32-bit strides of the tensor GEMM");
  const int nb = remote_o ? n_remote : sr_n_;
  auto expand = [&](const T *o, const int *cfg, int cnt) {
    T *buf = (T *)arena_.alloc(sizeof(T) * (size_t)cnt * n);
    hipLaunchKernelGGL(sr_expand_kernel<T>, dim3((unsigned)((slot_ + 255) / 256), sites, cnt), dim3(256), 0, stream_, o, cfg, buf,
                       sites, slot_, dp_, (const int *)sr_ne_);
    PG_CHECK_HIP(hipGetLastError());
    return buf;
  };
  T *ea = expand((const T *)sr_o_, (const int *)sr_cfg_, sr_n_);
  T *eb = remote_o ? expand((const T *)remote_o, (const int *)remote_cfg, nb) : ea;
  // Row blocks keep the 32-bit element offsets of the tensor GEMM in range (a sample is n elements long); the K index is
  // split over the batch dimension of the launch so that a few hundred samples still fill the chip (split-K, partial
  // results summed on the host in f64).
  const int bs = (int)std::min<long>(1024, ((1l << 31) - 1) / n) & ~63;
  PG_REQUIRE(bs >= 64, 1, "sr_gram: parameter count too large for the blocked Gram");
  const int groups = sites * dp_;                       // K splits must keep whole (site, state) slots together
  std::vector<double> h((size_t)sr_n_ * nb, 0.0), part;
  for (int ib = 0; ib < sr_n_; ib += bs) {
    const int ni = std::min(bs, sr_n_ - ib);
    for (int jb = remote_o ? 0 : ib; jb < nb; jb += bs) {
      const int nj = std::min(bs, nb - jb);
      const long tiles = (long)((ni + 63) / 64) * ((nj + 63) / 64);
      int split = 1;
      for (int c = 1; c <= groups; ++c)
        if (groups % c == 0 && tiles * c <= 2048) split = c;
      const long chunk = n / split;
      double *d = (double *)arena_.alloc(sizeof(double) * (size_t)split * ni * nj);
      PG_CHECK_HIP(hipMemsetAsync(d, 0, sizeof(double) * (size_t)split * ni * nj, stream_));
      TGemmDesc g;
      g.I[2] = ni; g.sAi[2] = (int)n; g.sCi[2] = nj;
      g.K[2] = (int)chunk; g.sAk[2] = 1; g.sBk[2] = 1;
      g.J[2] = nj; g.sBj[2] = (int)n; g.sCj[2] = 1;
      g.wA = chunk; g.wB = chunk; g.wC = (long)ni * nj;
      g.nbatch = split;
      g.upper_only = (!remote_o && ib == jb) ? 1 : 0;
      tgemm_launch<T, T, double, double>(stream_, g, ea + (size_t)ib * n, eb + (size_t)jb * n, d);
      part.resize((size_t)split * ni * nj);
      PG_CHECK_HIP(hipMemcpyAsync(part.data(), d, sizeof(double) * part.size(), hipMemcpyDeviceToHost, stream_));
      PG_CHECK_HIP(hipStreamSynchronize(stream_));
      arena_.free(d);
      for (int sp = 0; sp < split; ++sp)
        for (int i2 = 0; i2 < ni; ++i2) {
          const double *src = &part[((size_t)sp * ni + i2) * nj];
          double *dst = &h[(size_t)(ib + i2) * nb + jb];
          for (int j2 = 0; j2 < nj; ++j2) dst[j2] += src[j2];
        }
    }
  }
  if (!remote_o)   // tiles on and above the diagonal were computed
    for (int i2 = 0; i2 < sr_n_; ++i2)
      for (int j2 = 0; j2 < i2; ++j2) h[(size_t)i2 * nb + j2] = h[(size_t)j2 * nb + i2];
  std::copy(h.begin(), h.end(), out);
  if (eb != ea) arena_.free(eb);
  arena_.free(ea);
  }
}

template <typename T>
void Engine<T>::sr_weighted_sum(const double *y, double *out) {
  if constexpr (kCplx) {       // sum_i y_i O*_i with complex weights (interleaved pairs in, interleaved pairs out)
    PG_REQUIRE(sr_o_ != nullptr && sr_n_ > 0, 3, "sr_weighted_sum: no samples");
    const int sites = Ly_ * Lx_;
    const size_t n = (size_t)sites * dp_ * slot_ * 2;
    PG_CHECK_HIP(hipMemcpyAsync(sr_delta_, y, sizeof(double) * 2 * sr_n_, hipMemcpyHostToDevice, stream_));
    hipLaunchKernelGGL(sr_accum_cplx_kernel<T>, dim3((unsigned)((slot_ + 255) / 256), sites), dim3(256), 0, stream_, (const T *)sr_o_,
                       (const int *)sr_cfg_, (const double *)sr_delta_, 1.0, sr_out_, sr_n_, sites, slot_, dp_);
    PG_CHECK_HIP(hipGetLastError());
    std::vector<double> h(n);
    PG_CHECK_HIP(hipMemcpyAsync(h.data(), sr_out_, n * sizeof(double), hipMemcpyDeviceToHost, stream_));
    PG_CHECK_HIP(hipStreamSynchronize(stream_));
    sr_convert(h.data(), out, false);
  } else {
  PG_REQUIRE(sr_o_ != nullptr && sr_n_ > 0, 3, "sr_weighted_sum: no samples");
  const int sites = Ly_ * Lx_;
  const size_t n = (size_t)sites * dp_ * slot_;
  PG_CHECK_HIP(hipMemcpyAsync(sr_delta_, y, sizeof(double) * sr_n_, hipMemcpyHostToDevice, stream_));
  hipLaunchKernelGGL(sr_accum_kernel<T>, dim3((unsigned)((slot_ + 255) / 256), sites), dim3(256), 0, stream_, (const T *)sr_o_,
                     (const int *)sr_cfg_, (const double *)sr_delta_, 1.0, sr_out_, sr_n_, sites, slot_, dp_);
  PG_CHECK_HIP(hipGetLastError());
  std::vector<double> h(n);
  PG_CHECK_HIP(hipMemcpyAsync(h.data(), sr_out_, n * sizeof(double), hipMemcpyDeviceToHost, stream_));
  PG_CHECK_HIP(hipStreamSynchronize(stream_));
  sr_convert(h.data(), out, false);
  }
}

// device-to-device copy of the local sample store (for torch.distributed send / recv of the ring exchange)
template <typename T>
void Engine<T>::sr_copy_samples(void *dst_o, int32_t *dst_cfg) {
  {
  PG_REQUIRE(sr_o_ != nullptr && sr_n_ > 0, 3, "sr_copy_samples: no samples");
  const size_t sites = (size_t)Ly_ * Lx_;
  PG_CHECK_HIP(hipMemcpyAsync(dst_o, sr_o_, sizeof(T) * (size_t)sr_n_ * sites * slot_, hipMemcpyDeviceToDevice, stream_));
  PG_CHECK_HIP(hipMemcpyAsync(dst_cfg, sr_cfg_, sizeof(int) * (size_t)sr_n_ * sites, hipMemcpyDeviceToDevice, stream_));
  PG_CHECK_HIP(hipStreamSynchronize(stream_));
  }
}

}  // namespace pepsgpu
