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
        hipLaunchKernelGGL(chol_upper_cplx_kernel<T>, dim3(nw_), dim3(1024), 0, stream_, (c128 *)G, (long)cols * cols, cols,
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
    const int k = std::min(chi_, std::min(m, uk));
    PG_REQUIRE(m <= 1024, 1, "bond dimension too large for select_rows_kernel");
    DTen<T> V = alloc_ten(k, u, k2);
    // ---- dense sites, complex element type (round 5): the oversampled two-level route of the float64 engine (engine_impl.h has the
    // statement and the error argument) with the Hermitian forms: B^H B = M M^H, B2^H B2 = B B^H, rotated rows of B2 = sigma w^H,
    // w^H B = sigma u^H, Z = U^H M, complex Jacobi on the 2 chi rows of Z.  The complex one-sided Jacobi on the 256 x 256 block was
    // 96 % of a dense amplitude (6.6 amp/s at C4 whatever the batch).  Static shapes (this path has no live extents); walkers whose
    // factors keep more than 128 or fewer than chi + 4 rows, or whom the guard rejects, take the general kernel as before.
    int *rflag = nullptr;
    if constexpr (kCplx) {
      static const bool no_route = getenv("PEPSGPU_NO_C128_DENSE_ROUTE") != nullptr;
      const int kq = std::min(2 * k, (3 * std::min(m, uk)) / 4);
      // Round 6: the oversampled subspace from a RANDOMISED range finder -- a fixed table of signs times M, three re-orthonormalised steps
      // of subspace iteration (every half step a Cholesky-QR2 in complex float64, chol_solve_rows_cplx_kernel), then the same complex
      // Jacobi on Z = U M.  What lies outside the kq = 2 chi directions enters direction chi damped by (sigma_kq+1 / sigma_chi)^6
      // (~2e-8 on a state of the real spectrum): no Gram of M, no factorisation, no Jacobi on a 128 x 128 factor.  (The float64 route
      // selects its start rows by a pivoted factorisation, chol_pivot.h; a complex factor column does not fit a thread's registers.)
      // PEPSGPU_F64_PIVOT=0: the two-Cholesky route of round 5.
      static const int f64_pivot = getenv("PEPSGPU_F64_PIVOT") ? atoi(getenv("PEPSGPU_F64_PIVOT")) : 1;
      if (f64_pivot && !no_route && trunc_err_ == 0.0 && m >= 64 && m <= 256 && uk <= 256 && kq <= 64 && kq >= k + 8) {
        const int gb = (nw_ + 255) / 256;
        rflag = (int *)arena_.alloc(sizeof(int) * nw_);
        int *rowsM = (int *)arena_.alloc(sizeof(int) * nw_);
        hipLaunchKernelGGL(f64_route_init_kernel, dim3(gb), dim3(256), 0, stream_, (const int *)nullptr, 1, m, nw_, rowsM, rflag);
        arena_.free(rowsM);
        DTen<T> Om = alloc_ten(kq, m, 1);           // (only the first walker's slice is used: the table is shared, batch stride 0)
        hipLaunchKernelGGL(sign_table_kernel<T>, dim3((kq * m + 255) / 256), dim3(256), 0, stream_, Om.p, kq, m);
        PG_CHECK_HIP(hipGetLastError());
        DTen<T> Qz = alloc_ten(kq, uk, 1), Uz = alloc_ten(kq, m, 1);
        Acc *Sq = (Acc *)arena_.alloc(sizeof(Acc) * 64 * 64 * (size_t)nw_);
        auto orth = [&](DTen<T> &X, int len) {      // Cholesky-QR2 of the kq rows of X (in place)
          for (int pass = 0; pass < 2; ++pass) {
            TGemmDesc g;
            g.I[2] = kq; g.sAi[2] = len; g.sCi[2] = 64;
            g.K[2] = len; g.sAk[2] = 1; g.sBk[2] = 1;
            g.J[2] = kq; g.sBj[2] = len; g.sCj[2] = 1;
            g.wA = X.n; g.wB = X.n; g.wC = 64L * 64; g.nbatch = nw_;
            g.conjB = 1;
            g.upper_only = 1;
            tgemm_launch<T, T, Acc, Acc>(stream_, g, X.p, X.p, Sq);
            hipLaunchKernelGGL(chol_solve_rows_cplx_kernel, dim3(nw_), dim3(256), 0, stream_, (const c128 *)Sq, 64L * 64, 64, (c128 *)X.p, X.n, len, kq,
                               (const int *)nullptr);
            PG_CHECK_HIP(hipGetLastError());
          }
        };
        auto times_m = [&](const DTen<T> &U, long wU, DTen<T> &Zout) {       // Z = U M (kq x uk)
          TGemmDesc g;
          g.I[2] = kq; g.sAi[2] = m; g.sCi[2] = uk;
          g.K[2] = m; g.sAk[2] = 1; g.sBk[2] = uk;
          g.J[2] = uk; g.sBj[2] = 1; g.sCj[2] = 1;
          g.wA = wU; g.wB = M.n; g.wC = Zout.n; g.nbatch = nw_;
          tgemm_launch<T, T, T, T>(stream_, g, U.p, M.p, Zout.p);
        };
        auto times_mh = [&](const DTen<T> &Q, DTen<T> &Uout) {               // U = Q M^H (kq x m)
          TGemmDesc g;
          g.I[2] = kq; g.sAi[2] = uk; g.sCi[2] = m;
          g.K[2] = uk; g.sAk[2] = 1; g.sBk[2] = 1;
          g.J[2] = m; g.sBj[2] = uk; g.sCj[2] = 1;
          g.wA = Q.n; g.wB = M.n; g.wC = Uout.n; g.nbatch = nw_;
          g.conjB = 1;
          tgemm_launch<T, T, T, T>(stream_, g, Q.p, M.p, Uout.p);
        };
        times_m(Om, 0L, Qz);                          // the sketch: signs times M
        for (int it = 0; it < 3; ++it) {
          orth(Qz, uk);
          times_mh(Qz, Uz);
          orth(Uz, m);
          times_m(Uz, Uz.n, Qz);
        }
        hipLaunchKernelGGL(jacobi_rows_cplx_kernel<T>, dim3(nw_), dim3(1024), 0, stream_, Qz.p, Qz.n, kq, uk, uk, 60, sweeps_,
                           (const int *)rflag, 1);
        hipLaunchKernelGGL(select_rows_kernel<T>, dim3(nw_), dim3(256), 0, stream_, (const T *)Qz.p, Qz.n, kq, uk, uk, k, V.p, V.n,
                           (T *)nullptr, 0L, (const int *)nullptr, 1, (int *)nullptr, 0.0, chi_min_, (double *)nullptr, (const int *)rflag, 1);
        PG_CHECK_HIP(hipGetLastError());
        free_ten(Om); free_ten(Qz); free_ten(Uz);
        arena_.free(Sq);
      } else if (!no_route && trunc_err_ == 0.0 && m > 128 && m <= 256 && uk <= 256 && kq <= 64 && kq >= k + 8) {
        const int gb = (nw_ + 255) / 256, route_lo = std::min(kq, k + 4);
        constexpr double REDO_SCALE = 64.0;
        rflag = (int *)arena_.alloc(sizeof(int) * nw_);
        int *rowsM = (int *)arena_.alloc(sizeof(int) * nw_), *mB1 = (int *)arena_.alloc(sizeof(int) * nw_);
        int *mB2 = (int *)arena_.alloc(sizeof(int) * nw_), *kW = (int *)arena_.alloc(sizeof(int) * nw_);
        int *redo = (int *)arena_.alloc(sizeof(int) * nw_), *lvl = (int *)arena_.alloc(sizeof(int) * nw_);
        PG_CHECK_HIP(hipMemsetAsync(mB1, 0, sizeof(int) * nw_, stream_));
        PG_CHECK_HIP(hipMemsetAsync(mB2, 0, sizeof(int) * nw_, stream_));
        PG_CHECK_HIP(hipMemsetAsync(kW, 0, sizeof(int) * nw_, stream_));
        hipLaunchKernelGGL(f64_route_init_kernel, dim3(gb), dim3(256), 0, stream_, (const int *)nullptr, 1, m, nw_, rowsM, rflag);
        Acc *Gm = (Acc *)arena_.alloc(sizeof(Acc) * (size_t)m * m * nw_);
        DTen<T> B1 = alloc_ten(m, m, 1);
        auto gram_m = [&](const int *flag) {      // G = M M^H, upper triangle
          TGemmDesc g;
          g.I[2] = m; g.sAi[2] = uk; g.sCi[2] = m;
          g.K[2] = uk; g.sAk[2] = 1; g.sBk[2] = 1;
          g.J[2] = m; g.sBj[2] = uk; g.sCj[2] = 1;
          g.wA = M.n; g.wB = M.n; g.wC = (long)m * m; g.nbatch = nw_;
          g.conjB = 1;
          g.upper_only = 1;
          g.batch_flag = flag;
          tgemm_launch<T, T, Acc, Acc>(stream_, g, M.p, M.p, Gm);
        };
        gram_m(nullptr);
        constexpr double tscale = 1.0;
        hipLaunchKernelGGL(chol_upper_cplx_kernel<T>, dim3(nw_), dim3(1024), 0, stream_, (c128 *)Gm, (long)m * m, m, B1.p, B1.n, mB1,
                           (const int *)nullptr, tscale);
        // second chance for the walkers whose factor kept more than 128 rows: pivot threshold x REDO_SCALE (the guard prices it)
        hipLaunchKernelGGL(f64_route_redo_kernel, dim3(gb), dim3(256), 0, stream_, (const int *)mB1, 128, nw_, redo, lvl);
        gram_m(redo);
        hipLaunchKernelGGL(chol_upper_cplx_kernel<T>, dim3(nw_), dim3(1024), 0, stream_, (c128 *)Gm, (long)m * m, m, B1.p, B1.n, mB1,
                           (const int *)redo, tscale * REDO_SCALE);
        hipLaunchKernelGGL(f64_route_redo_kernel, dim3(gb), dim3(256), 0, stream_, (const int *)mB1, 128, nw_, redo, lvl, 2);
        gram_m(redo);
        hipLaunchKernelGGL(chol_upper_cplx_kernel<T>, dim3(nw_), dim3(1024), 0, stream_, (c128 *)Gm, (long)m * m, m, B1.p, B1.n, mB1,
                           (const int *)redo, tscale * REDO_SCALE * REDO_SCALE);
        PG_CHECK_HIP(hipGetLastError());
        arena_.free(Gm);
        hipLaunchKernelGGL(f64_route_check_kernel, dim3(gb), dim3(256), 0, stream_, rflag, mB1, route_lo, 128, nw_);
        Acc *G2 = (Acc *)arena_.alloc(sizeof(Acc) * (size_t)128 * 128 * nw_);
        PG_CHECK_HIP(hipMemsetAsync(G2, 0, sizeof(Acc) * (size_t)128 * 128 * nw_, stream_));   // (the factor kernel reads the full order)
        DTen<T> B2 = alloc_ten(128, 128, 1);
        {   // G2 = B B^H over the kept rows of B (<= 128; rows of B are m long)
          TGemmDesc g;
          g.I[2] = 128; g.sAi[2] = m; g.sCi[2] = 128;
          g.K[2] = m; g.sAk[2] = 1; g.sBk[2] = 1;
          g.J[2] = 128; g.sBj[2] = m; g.sCj[2] = 1;
          g.wA = B1.n; g.wB = B1.n; g.wC = 128L * 128; g.nbatch = nw_;
          g.dI[2].p = mB1; g.dJ[2].p = mB1;
          g.conjB = 1;
          g.upper_only = 1;
          g.batch_flag = rflag;
          tgemm_launch<T, T, Acc, Acc>(stream_, g, B1.p, B1.p, G2);
        }
        hipLaunchKernelGGL(chol_upper_cplx_kernel<T>, dim3(nw_), dim3(1024), 0, stream_, (c128 *)G2, 128L * 128, 128, B2.p, B2.n, mB2,
                           (const int *)rflag, 1.0);
        PG_CHECK_HIP(hipGetLastError());
        arena_.free(G2);
        hipLaunchKernelGGL(f64_route_check_kernel, dim3(gb), dim3(256), 0, stream_, rflag, mB2, route_lo, 128, nw_);
        hipLaunchKernelGGL(jacobi_rows_cplx_kernel<T>, dim3(nw_), dim3(1024), 0, stream_, B2.p, B2.n, 128, 128, 128, 60, sweeps_,
                           (const int *)rflag, 1);
        PG_CHECK_HIP(hipGetLastError());
        DTen<T> Wt = alloc_ten(kq, 128, 1), T1 = alloc_ten(kq, m, 1), Uq = alloc_ten(kq, m, 1), Zt = alloc_ten(kq, uk, 1);
        hipLaunchKernelGGL(select_rows_kernel<T>, dim3(nw_), dim3(256), 0, stream_, (const T *)B2.p, B2.n, 128, 128, 128, kq, Wt.p, Wt.n,
                           (T *)nullptr, 0L, (const int *)mB2, 1, kW, 0.0, 0, (double *)nullptr, (const int *)rflag, 1);
        PG_CHECK_HIP(hipGetLastError());
        {   // sigma_q u_q^H = w_q^H B
          TGemmDesc g;
          g.I[2] = kq; g.sAi[2] = 128; g.sCi[2] = m;
          g.K[2] = 128; g.sAk[2] = 1; g.sBk[2] = m;
          g.J[2] = m; g.sBj[2] = 1; g.sCj[2] = 1;
          g.wA = Wt.n; g.wB = B1.n; g.wC = T1.n; g.nbatch = nw_;
          g.dK[2].p = mB1;
          g.batch_flag = rflag;
          tgemm_launch<T, T, T, T>(stream_, g, Wt.p, B1.p, T1.p);
        }
        hipLaunchKernelGGL(select_rows_kernel<T>, dim3(nw_), dim3(256), 0, stream_, (const T *)T1.p, T1.n, kq, m, m, kq, Uq.p, Uq.n,
                           (T *)nullptr, 0L, (const int *)kW, 1, (int *)nullptr, 0.0, 0, (double *)nullptr, (const int *)rflag, 1);
        PG_CHECK_HIP(hipGetLastError());
        {   // Z = U^H M (kq x uk)
          TGemmDesc g;
          g.I[2] = kq; g.sAi[2] = m; g.sCi[2] = uk;
          g.K[2] = m; g.sAk[2] = 1; g.sBk[2] = uk;
          g.J[2] = uk; g.sBj[2] = 1; g.sCj[2] = 1;
          g.wA = Uq.n; g.wB = M.n; g.wC = Zt.n; g.nbatch = nw_;
          g.batch_flag = rflag;
          tgemm_launch<T, T, T, T>(stream_, g, Uq.p, M.p, Zt.p);
        }
        hipLaunchKernelGGL(jacobi_rows_cplx_kernel<T>, dim3(nw_), dim3(1024), 0, stream_, Zt.p, Zt.n, kq, uk, uk, 60, sweeps_,
                           (const int *)rflag, 1);
        hipLaunchKernelGGL(select_rows_kernel<T>, dim3(nw_), dim3(256), 0, stream_, (const T *)Zt.p, Zt.n, kq, uk, uk, k, V.p, V.n,
                           (T *)nullptr, 0L, (const int *)kW, 1, (int *)nullptr, 0.0, chi_min_, (double *)nullptr, (const int *)rflag, 1);
        constexpr double guard_tol = 1e-10;
        hipLaunchKernelGGL(f64_route_guard_kernel<T>, dim3(nw_), dim3(256), 0, stream_, (const T *)Zt.p, Zt.n, uk, (const int *)kW, k, guard_tol,
                           rflag, kq, (const int *)lvl, 5.7e-14 * tscale * REDO_SCALE, 5.7e-14 * tscale * REDO_SCALE * REDO_SCALE, 5.7e-14 * tscale,
                           tscale > 1.0 ? 1 : 0);
        PG_CHECK_HIP(hipGetLastError());
        if (dbg_sweeps_ && getenv("PEPSGPU_DEBUG_VERBOSE")) {   // diagnostics: who stays on the route
          std::vector<int> hf(nw_), h1(nw_), h2(nw_), hk(nw_);
          PG_CHECK_HIP(hipMemcpyAsync(hf.data(), rflag, nw_ * sizeof(int), hipMemcpyDeviceToHost, stream_));
          PG_CHECK_HIP(hipMemcpyAsync(h1.data(), mB1, nw_ * sizeof(int), hipMemcpyDeviceToHost, stream_));
          PG_CHECK_HIP(hipMemcpyAsync(h2.data(), mB2, nw_ * sizeof(int), hipMemcpyDeviceToHost, stream_));
          PG_CHECK_HIP(hipMemcpyAsync(hk.data(), kW, nw_ * sizeof(int), hipMemcpyDeviceToHost, stream_));
          PG_CHECK_HIP(hipStreamSynchronize(stream_));
          long on = 0, s1 = 0, s2 = 0, sk = 0, z1 = 0, z2 = 0;
          for (int w = 0; w < nw_; ++w) { on += hf[w] < 0; s1 += h1[w]; s2 += h2[w]; sk += hf[w] < 0 ? hk[w] : 0; z1 += h1[w] == 0; z2 += h2[w] == 0; }
          fprintf(stderr, "[pepsgpu] c128 dense route site %d (m = %d, uk = %d, kq = %d): %ld of %d walkers on the route; first factor rows mean %.1f (%ld off), second %.1f (%ld off), kept directions mean %.1f\n",
                  i, m, uk, kq, on, nw_, (double)s1 / nw_, z1, (double)s2 / nw_, z2, on ? (double)sk / on : 0.0);
        }
        free_ten(B1); free_ten(B2); free_ten(Wt); free_ten(T1); free_ten(Uq); free_ten(Zt);
        arena_.free(rowsM); arena_.free(mB1); arena_.free(mB2); arena_.free(kW); arena_.free(redo); arena_.free(lvl);
      }
    }
    // the general kernel: every walker, or -- behind the route -- the walkers that left it (rflag >= 0)
    if constexpr (kCplx) {
      hipLaunchKernelGGL(jacobi_rows_cplx_kernel<T>, dim3(nw_), dim3(1024), 0, stream_, M.p, M.n, m, uk, uk, 60, sweeps_,
                         (const int *)rflag, 0);
    } else {
      hipLaunchKernelGGL(jacobi_rows_kernel<T>, dim3(nw_), dim3(1024), 0, stream_, M.p, M.n, m, uk, uk, 60, 0, sweeps_,
                         (const int *)nullptr, 1, 0);
    }
    PG_CHECK_HIP(hipGetLastError());
    ++n_jacobi_;
    hipLaunchKernelGGL(select_rows_kernel<T>, dim3(nw_), dim3(256), 0, stream_, (const T *)M.p, M.n, m, uk, uk, k, V.p, V.n,
                       (T *)nullptr, 0L, (const int *)nullptr, 1, (int *)nullptr, trunc_err_, chi_min_, (double *)nullptr,
                       (const int *)rflag, rflag ? 0 : 1);
    PG_CHECK_HIP(hipGetLastError());
    if (rflag) { arena_.free(rflag); rflag = nullptr; }
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
