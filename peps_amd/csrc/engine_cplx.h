// Row absorption for the COMPLEX element type: the Q-less algorithm of engine.h in its plain form -- static, zero padded
// shapes, no rank / bond adaptivity, generic tensor GEMM on the vector ALUs -- with the conjugations a complex SVD needs:
//
//   forward   P_i = R_i (A_i x W_i),   R_{i+1}^H R_{i+1} = P_i^H P_i          (Hermitian Gram, complex float64)
//   backward  T_i = (A_i x W_i) Y_{i+1},  M_i = R_i T_i,  rows of M_i --complex Jacobi--> sigma_k v_k^H,
//             Vt_i = chi largest rows, normalised  (= V^H of qlten::SVD, bmps_impl.h:235-238),   Y_i = T_i Vt_i^H
//
// Reference: BMPS::MultiplyMPOSVDCompress_ (bmps_impl.h:756-862) + RightCanonicalizeTruncate (:225-263) with
// TenElemT = QLTEN_Complex; no Dag() appears in the reference's absorption, the conjugates are inside qlten::QR / SVD.
// Parity-grade (kernels of linalg_cplx.h), not tuned: the throughput path of this library is real f32.
#pragma once
#include "engine.h"
#include "linalg_cplx.h"

namespace pepsgpu {

template <typename T>
typename Engine<T>::BMPSDev Engine<T>::absorb_simple(int pos, int num, const BMPSDev &in) {
  ArenaScope scope(arena_);
  const int N = mps_len(pos);
  const std::vector<DTen<T>> &cur = in.t;
  PG_REQUIRE((int)cur.size() == N, 3, "MultiplyMPO: MPS/MPO length mismatch");
  auto site_rc = [&](int i, int &r, int &c) {
    switch (pos) {
      case DOWN: r = num; c = i; break;
      case UP: r = num; c = N - 1 - i; break;
      case LEFT: r = i; c = num; break;
      default: r = N - 1 - i; c = num; break;
    }
  };
  const int ll = (pos + 3) % 4, lp = pos, lr = (pos + 1) % 4, lu = (pos + 2) % 4;
  BMPSDev out;

  // ---------------- forward ----------------
  std::vector<DTen<T>> R(N);
  R[0] = ones3();
  for (int i = 0; i + 1 < N; ++i) {
    int r, c, dd[4], st[4];
    site_rc(i, r, c);
    site_dims(r, c, dd);
    site_strides(r, c, st);
    const DTen<T> &A = cur[i];
    const int m = R[i].d[0], l = R[i].d[1], a = R[i].d[2];
    const int p = A.d[1], a2 = A.d[2];
    const int l2 = dd[lr], u = dd[lu];
    PG_REQUIRE(l == dd[ll] && a == A.d[0] && p == dd[lp], 3, "MultiplyMPO: bond dimension mismatch");
    DTen<T> X = alloc_ten(m * l, p, a2);
    DTen<T> P = alloc_ten(m, u, l2, a2);
    {
      TGemmDesc gx, gp;   // X[m,l,p,a2] = sum_a R[m,l,a] A[a,p,a2];  P[m,u,l2,a2] = sum_{l,p} W[l,p,l2,u] X[m,l,p,a2]
      gx.I[1] = m; gx.I[2] = l; gx.sAi[1] = l * a; gx.sAi[2] = a; gx.sCi[1] = l * p * a2; gx.sCi[2] = p * a2;
      gx.K[2] = a; gx.sAk[2] = 1; gx.sBk[2] = p * a2;
      gx.J[1] = p; gx.J[2] = a2; gx.sBj[1] = a2; gx.sBj[2] = 1; gx.sCj[1] = a2; gx.sCj[2] = 1;
      gx.wA = R[i].n; gx.wB = A.n; gx.wC = X.n; gx.nbatch = nw_;
      gp.I[1] = l2; gp.I[2] = u; gp.sAi[1] = st[lr]; gp.sAi[2] = st[lu]; gp.sCi[1] = a2; gp.sCi[2] = l2 * a2;
      gp.K[1] = l; gp.K[2] = p; gp.sAk[1] = st[ll]; gp.sAk[2] = st[lp]; gp.sBk[1] = p * a2; gp.sBk[2] = a2;
      gp.J[1] = m; gp.J[2] = a2; gp.sBj[1] = l * p * a2; gp.sBj[2] = 1; gp.sCj[1] = u * l2 * a2; gp.sCj[2] = 1;
      gp.wB = X.n; gp.wC = P.n; gp.nbatch = nw_;
      tgemm_launch<T, T, T, T>(stream_, gx, R[i].p, A.p, X.p);
      launch_site_gemm_a(gp, cfg_site(r, c), 1, X.p, P.p);
    }
    free_ten(X);
    const int rows = m * u, cols = l2 * a2;
    if (rows < cols) {          // any R with R^H R = P^H P serves, P itself included
      P.d[0] = rows; P.d[1] = l2; P.d[2] = a2; P.d[3] = 1;
      normalize(P.p, P.n, P.n, nw_, nullptr);
      R[i + 1] = P;
    } else {
      PG_REQUIRE(cols <= 1024, 1, "D * chi too large for the complex Cholesky kernel");
      Acc *G = (Acc *)arena_.alloc(sizeof(Acc) * (size_t)cols * cols * nw_);
      R[i + 1] = alloc_ten(cols, l2, a2);
      TGemmDesc g;            // G = P^H P
      g.I[2] = cols; g.sAi[2] = 1; g.sCi[2] = cols;
      g.K[2] = rows; g.sAk[2] = cols; g.sBk[2] = cols;
      g.J[2] = cols; g.sBj[2] = 1; g.sCj[2] = 1;
      g.wA = P.n; g.wB = P.n; g.wC = (long)cols * cols; g.nbatch = nw_;
      g.conjA = 1;
      tgemm_launch<T, T, Acc, Acc>(stream_, g, P.p, P.p, G);
      if constexpr (kCplx) {
        hipLaunchKernelGGL(chol_upper_cplx_kernel<T>, dim3(nw_), dim3(256), 0, stream_, (c128 *)G, (long)cols * cols, cols,
                           R[i + 1].p, R[i + 1].n, (int *)nullptr);
      } else {
        const size_t smem = chol_smem_bytes(cols);
        allow_dynamic_lds(reinterpret_cast<const void *>(&chol_upper_kernel<T>), smem);
        hipLaunchKernelGGL(chol_upper_kernel<T>, dim3(nw_), dim3(256), smem, stream_, (double *)G, (long)cols * cols, cols,
                           R[i + 1].p, R[i + 1].n, (int *)nullptr);
      }
      PG_CHECK_HIP(hipGetLastError());
      arena_.free(G);
      free_ten(P);
    }
  }

  // ---------------- backward ----------------
  out.t.resize(N);
  out.live.assign(N + 1, nullptr);
  out.logscale = (double *)arena_.alloc(sizeof(double) * nw_);
  PG_CHECK_HIP(hipMemcpyAsync(out.logscale, in.logscale, sizeof(double) * nw_, hipMemcpyDeviceToDevice, stream_));
  DTen<T> Y = ones3();   // [l2, a2, k2]
  for (int i = N - 1; i >= 0; --i) {
    int r, c, dd[4], st[4];
    site_rc(i, r, c);
    site_dims(r, c, dd);
    site_strides(r, c, st);
    const DTen<T> &A = cur[i];
    const int a = A.d[0], p = A.d[1], a2 = A.d[2];
    const int l = dd[ll], l2 = dd[lr], u = dd[lu];
    const int k2 = Y.d[2];
    PG_REQUIRE(Y.d[0] == l2 && Y.d[1] == a2 && p == dd[lp], 3, "MultiplyMPO: bond dimension mismatch (backward)");
    DTen<T> Z1 = alloc_ten(a, p, l2, k2);
    DTen<T> Tt = alloc_ten(l, a, u, k2);
    {
      TGemmDesc gz, gt;   // Z1[a,p,l2,k2] = sum_{a2} A[a,p,a2] Y[l2,a2,k2];  Tt[l,a,u,k2] = sum_{p,l2} W[l,p,l2,u] Z1[a,p,l2,k2]
      gz.I[1] = a; gz.I[2] = p; gz.sAi[1] = p * a2; gz.sAi[2] = a2; gz.sCi[1] = p * l2 * k2; gz.sCi[2] = l2 * k2;
      gz.K[2] = a2; gz.sAk[2] = 1; gz.sBk[2] = k2;
      gz.J[1] = l2; gz.J[2] = k2; gz.sBj[1] = a2 * k2; gz.sBj[2] = 1; gz.sCj[1] = k2; gz.sCj[2] = 1;
      gz.wA = A.n; gz.wB = Y.n; gz.wC = Z1.n; gz.nbatch = nw_;
      gt.I[1] = l; gt.I[2] = u; gt.sAi[1] = st[ll]; gt.sAi[2] = st[lu]; gt.sCi[1] = a * u * k2; gt.sCi[2] = k2;
      gt.K[1] = p; gt.K[2] = l2; gt.sAk[1] = st[lp]; gt.sAk[2] = st[lr]; gt.sBk[1] = l2 * k2; gt.sBk[2] = k2;
      gt.J[1] = a; gt.J[2] = k2; gt.sBj[1] = p * l2 * k2; gt.sBj[2] = 1; gt.sCj[1] = u * k2; gt.sCj[2] = 1;
      gt.wB = Z1.n; gt.wC = Tt.n; gt.nbatch = nw_;
      tgemm_launch<T, T, T, T>(stream_, gz, A.p, Y.p, Z1.p);
      launch_site_gemm_a(gt, cfg_site(r, c), 1, Z1.p, Tt.p);
    }
    free_ten(Z1);
    free_ten(Y);
    if (i == 0) {
      PG_REQUIRE(l == 1 && a == 1, 3, "MultiplyMPO: left boundary bond is not trivial");
      Tt.d[0] = 1; Tt.d[1] = u; Tt.d[2] = k2; Tt.d[3] = 1;
      normalize(Tt.p, Tt.n, Tt.n, nw_, out.logscale);
      out.t[0] = Tt;
      break;
    }
    const int m = R[i].d[0], la = l * a, uk = u * k2;
    PG_REQUIRE(R[i].d[1] == l && R[i].d[2] == a, 3, "MultiplyMPO: carry dimension mismatch");
    DTen<T> M = alloc_ten(m, uk, 1);
    {
      TGemmDesc g;   // M[m,(u,k2)] = sum_{(l,a)} R_i[m,(l,a)] Tt[(l,a),(u,k2)]
      g.I[2] = m; g.sAi[2] = la; g.sCi[2] = uk;
      g.K[1] = l; g.K[2] = a; g.sAk[1] = a; g.sAk[2] = 1; g.sBk[1] = a * uk; g.sBk[2] = uk;
      g.J[1] = u; g.J[2] = k2; g.sBj[1] = k2; g.sBj[2] = 1; g.sCj[1] = k2; g.sCj[2] = 1;
      g.wA = R[i].n; g.wB = Tt.n; g.wC = M.n; g.nbatch = nw_;
      tgemm_launch<T, T, T, T>(stream_, g, R[i].p, Tt.p, M.p);
    }
    if constexpr (kCplx) {
      hipLaunchKernelGGL(jacobi_rows_cplx_kernel<T>, dim3(nw_), dim3(1024), 0, stream_, M.p, M.n, m, uk, uk, 60, sweeps_);
    } else {
      hipLaunchKernelGGL(jacobi_rows_kernel<T>, dim3(nw_), dim3(1024), 0, stream_, M.p, M.n, m, uk, uk, 60, 0, sweeps_,
                         (const int *)nullptr, 1, 0);
    }
    PG_CHECK_HIP(hipGetLastError());
    ++n_jacobi_;
    const int k = std::min(chi_, std::min(m, uk));
    PG_REQUIRE(m <= 1024, 1, "bond dimension too large for select_rows_kernel");
    DTen<T> V = alloc_ten(k, u, k2);
    hipLaunchKernelGGL(select_rows_kernel<T>, dim3(nw_), dim3(256), 0, stream_, (const T *)M.p, M.n, m, uk, uk, k, V.p, V.n,
                       (T *)nullptr, 0L, (const int *)nullptr, 1, (int *)nullptr, trunc_err_, chi_min_, (double *)nullptr,
                       (const int *)nullptr, 1);
    PG_CHECK_HIP(hipGetLastError());
    free_ten(M);
    out.t[i] = V;
    DTen<T> Yn = alloc_ten(l, a, k);
    {
      TGemmDesc g;   // Y[(l,a),q] = sum_{(u,k2)} Tt[(l,a),(u,k2)] conj(Vt[q,(u,k2)])
      g.I[1] = l; g.I[2] = a; g.sAi[1] = a * uk; g.sAi[2] = uk; g.sCi[1] = a * k; g.sCi[2] = k;
      g.K[1] = u; g.K[2] = k2; g.sAk[1] = k2; g.sAk[2] = 1; g.sBk[1] = k2; g.sBk[2] = 1;
      g.J[2] = k; g.sBj[2] = uk; g.sCj[2] = 1;
      g.wA = Tt.n; g.wB = V.n; g.wC = Yn.n; g.nbatch = nw_;
      g.conjB = 1;
      tgemm_launch<T, T, T, T>(stream_, g, Tt.p, V.p, Yn.p);
    }
    normalize(Yn.p, Yn.n, Yn.n, nw_, out.logscale);
    free_ten(Tt);
    Y = Yn;
  }
  for (auto &t : R) arena_.free(t.p);
  out.kmax.assign(N + 1, -1);
  out.mlmax.assign(N, -1);
  out.depth = in.depth + 1;
  ++n_absorb_;
  return out;
}

}  // namespace pepsgpu
