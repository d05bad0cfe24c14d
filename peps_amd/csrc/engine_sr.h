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

// delta[i] = sum_site < O*_i(site), v(site)[cfg_i(site)] > - shift      (one block per sample)
template <typename T>
__global__ __launch_bounds__(256) void sr_delta_kernel(const T *__restrict__ o, const int *__restrict__ cfg,
                                                       const double *__restrict__ v, double shift, double *__restrict__ delta,
                                                       int sites, long slot, int dp) {
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
  if (threadIdx.x == 0) delta[i] = s_red[0] + s_red[1] + s_red[2] + s_red[3] - shift;
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
  const size_t n = (size_t)Ly_ * Lx_ * dp_ * slot_;
  std::fill(dst, dst + n, 0.0);
  const size_t m = sr_map_c_.size();
  if (to_compact) for (size_t k = 0; k < m; ++k) dst[sr_map_c_[k]] = src[sr_map_p_[k]];
  else for (size_t k = 0; k < m; ++k) dst[sr_map_p_[k]] = src[sr_map_c_[k]];
}

template <typename T>
void Engine<T>::sr_begin(int max_samples) {
  PG_REQUIRE(max_samples > 0, 1, "sr_begin: need a positive sample capacity");
  sr_release();
  const size_t sites = (size_t)Ly_ * Lx_;
  sr_o_ = (T *)arena_.alloc(sizeof(T) * (size_t)max_samples * sites * slot_);
  sr_cfg_ = (int *)arena_.alloc(sizeof(int) * (size_t)max_samples * sites);
  sr_delta_ = (double *)arena_.alloc(sizeof(double) * (size_t)max_samples);
  sr_v_ = (double *)arena_.alloc(sizeof(double) * sites * dp_ * slot_);
  sr_out_ = (double *)arena_.alloc(sizeof(double) * sites * dp_ * slot_);
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

template <typename T>
void Engine<T>::sr_release() {
  if (sr_o_) { arena_.free(sr_o_); arena_.free(sr_cfg_); arena_.free(sr_delta_); arena_.free(sr_v_); arena_.free(sr_out_); arena_.free(sr_ne_); }
  sr_ne_ = nullptr;
  sr_o_ = nullptr; sr_cfg_ = nullptr; sr_delta_ = nullptr; sr_v_ = nullptr; sr_out_ = nullptr;
  sr_cap_ = sr_n_ = 0;
}

template <typename T>
void Engine<T>::sr_append(const double *psi) {
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

template <typename T>
void Engine<T>::sr_sum(double *out) {
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

// out = scale * sum_i (O*_i . v - mean_dot_v) O*_i      (caller: all-reduce over ranks, + diag_shift * v)
template <typename T>
void Engine<T>::sr_matvec(const double *v, double mean_dot_v, double scale, double *out) {
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

}  // namespace pepsgpu
