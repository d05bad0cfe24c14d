/* cbmps.c -- plain-C restatement of the reference's CPU path for one EvaluateAmplitude, on LAPACK.
 *
 * TEST INFRASTRUCTURE ONLY (checker + cpu_baseline of bench.py); nothing under peps_amd/ links or loads it.
 *
 * Restates, op for op, in float64:
 *   BMPS::MultiplyMPOSVDCompress_   /root/reference/include/qlpeps/one_dim_tn/boundary_mps/bmps_impl.h:756-862
 *       G1  tmp1 = Contract(mps[i], r)            :806      dgemm
 *       G2  tmp2 = Contract(tmp1, mpo[i])         :807      explicit transpose + dgemm (TensorToolkit transposes too)
 *       Transpose{1,3,2,0}                        :815-817  explicit copy
 *       QR(ldims = 2), explicit Q                 :821      dgelqf + dorglq on the row-major block (= geqrf + orgqr flops)
 *   BMPS::RightCanonicalizeTruncate :225-263      dgesdd (jobz 'S'), keep min(D_max, k), res[i-1] . (u s) dgemm
 *   TPSWaveFunctionComponent::EvaluateAmplitude   vmc_basic/wave_function_component.h:187-212: DOWN stack grown to
 *       row 1, then the top row closed against it (GrowFullBTen + Trace are restated here as the same chain of
 *       transfer contractions column by column; ~1 % of the work).
 * Execution model of the reference: one walker per MPI rank, BLAS threads = 1 per rank
 * (algorithm/vmc_update/monte_carlo_engine.h:97-98; examples/transverse_field_ising_vmc_optimize.cpp sets the
 * tensor-manipulation threads to 1).  Here: `nthreads` POSIX threads, each walking its own walkers, OpenBLAS pinned
 * to one thread, i.e. (i) nthreads = 1 and (ii) nthreads = host cores are the two baselines SURVEY.md 8(d) asks for.
 * LAPACK comes from the OpenBLAS that ships inside SciPy (Fortran ABI, LP64, symbols prefixed scipy_), opened with
 * dlopen at cbmps_init: no headers or link-time dependency needed.
 * Pinned against oracle/bmps.py (itself pinned on the reference's known answers) in tests/test_oracle_c.py.
 */
#include <dlfcn.h>
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef void (*dgemm_t)(const char *, const char *, const int *, const int *, const int *, const double *, const double *,
                        const int *, const double *, const int *, const double *, double *, const int *);
typedef void (*dgelqf_t)(const int *, const int *, double *, const int *, double *, double *, const int *, int *);
typedef void (*dorglq_t)(const int *, const int *, const int *, double *, const int *, const double *, double *, const int *,
                         int *);
typedef void (*dgesdd_t)(const char *, const int *, const int *, double *, const int *, double *, double *, const int *,
                         double *, const int *, double *, const int *, int *, int *);
typedef void (*setthr_t)(int);

static void *g_blas = NULL;
static dgemm_t p_dgemm;
static dgelqf_t p_dgelqf;
static dorglq_t p_dorglq;
static dgesdd_t p_dgesdd;
static setthr_t p_setthr;
static char g_err[256];

const char *cbmps_last_error(void) { return g_err; }

static void *sym2(const char *a, const char *b) {
  void *p = dlsym(g_blas, a);
  if (!p) p = dlsym(g_blas, b);
  return p;
}

int cbmps_init(const char *blas_path) {
  if (g_blas) return 0;
  g_blas = dlopen(blas_path, RTLD_NOW | RTLD_LOCAL);
  if (!g_blas) { snprintf(g_err, sizeof g_err, "dlopen(%s): %s", blas_path, dlerror()); return 1; }
  p_dgemm = (dgemm_t)sym2("scipy_dgemm_", "dgemm_");
  p_dgelqf = (dgelqf_t)sym2("scipy_dgelqf_", "dgelqf_");
  p_dorglq = (dorglq_t)sym2("scipy_dorglq_", "dorglq_");
  p_dgesdd = (dgesdd_t)sym2("scipy_dgesdd_", "dgesdd_");
  p_setthr = (setthr_t)sym2("scipy_openblas_set_num_threads", "openblas_set_num_threads");
  if (!p_dgemm || !p_dgelqf || !p_dorglq || !p_dgesdd) {
    snprintf(g_err, sizeof g_err, "BLAS/LAPACK symbols missing in %s", blas_path);
    return 2;
  }
  return 0;
}

/* row-major C(MxN) = op(A) op(B); ta/tb = 'N' or 'T' refer to the row-major operands */
static void gemm_rm(char ta, char tb, int M, int N, int K, const double *A, int lda, const double *B, int ldb, double *C,
                    int ldc) {
  const double one = 1.0, zero = 0.0;
  p_dgemm(&tb, &ta, &N, &M, &K, &one, B, &ldb, A, &lda, &zero, C, &ldc);
}

typedef struct { double *p; int d0, d1, d2; } Ten3;   /* (left, phys, right), row-major */

static double *xalloc(size_t n) {
  double *p = (double *)malloc((n ? n : 1) * sizeof(double));
  if (!p) { fprintf(stderr, "cbmps: out of memory\n"); abort(); }
  return p;
}

static int imin(int a, int b) { return a < b ? a : b; }

/* economy QR of the row-major block X (rows x cols): Q (rows x k) row-major into Qout, R (k x cols) row-major into Rout,
 * k = min(rows, cols).  The row-major X is the column-major X^T (cols x rows); X^T = L Qt  =>  X = Qt^T L^T = Q R. */
static void qr_rm(double *X, int rows, int cols, double *Qout, double *Rout, double **work, size_t *lwork_have) {
  const int k = imin(rows, cols);
  double *tau = xalloc((size_t)k);
  int info = 0, lw = -1;
  double wq;
  /* col-major matrix: M = cols, N = rows, lda = cols */
  p_dgelqf(&cols, &rows, X, &cols, tau, &wq, &lw, &info);
  size_t need = (size_t)wq;
  lw = -1;
  p_dorglq(&k, &rows, &k, X, &cols, tau, &wq, &lw, &info);
  if ((size_t)wq > need) need = (size_t)wq;
  if (need > *lwork_have) { free(*work); *work = xalloc(need); *lwork_have = need; }
  lw = (int)*lwork_have;
  p_dgelqf(&cols, &rows, X, &cols, tau, *work, &lw, &info);
  /* L = lower trapezoid (cols x k) col-major  ==  R^T;  R row-major (k x cols): R[i][j] = L[j][i] = X_cm[j + i*cols], j >= i */
  for (int i = 0; i < k; ++i)
    for (int j = 0; j < cols; ++j) Rout[(size_t)i * cols + j] = (j >= i) ? X[(size_t)j + (size_t)i * cols] : 0.0;
  /* Qt (k x rows) col-major, lda = cols  -> Q row-major (rows x k): Q[r][q] = Qt[q][r] = X_cm[q + r*cols] */
  p_dorglq(&k, &rows, &k, X, &cols, tau, *work, &lw, &info);
  for (int r = 0; r < rows; ++r)
    for (int q = 0; q < k; ++q) Qout[(size_t)r * k + q] = X[(size_t)q + (size_t)r * cols];
  free(tau);
}

typedef struct {
  int L, D, d, chi;
  const double *sitps;    /* [r][c][s][D][D][D][D] zero padded */
} Model;

static void site_dims(const Model *m, int r, int c, int *dd) {
  dd[0] = c == 0 ? 1 : m->D; dd[1] = r == m->L - 1 ? 1 : m->D; dd[2] = c == m->L - 1 ? 1 : m->D; dd[3] = r == 0 ? 1 : m->D;
}
/* compact copy (true leg dims, row-major (L, D, R, U)) of component s at (r, c) */
static void site_tensor(const Model *m, int r, int c, int s, double *out) {
  int dd[4];
  site_dims(m, r, c, dd);
  const int D = m->D;
  const double *src = m->sitps + (((size_t)(r * m->L + c) * m->d + s) * D * D * D * D);
  size_t o = 0;
  for (int a = 0; a < dd[0]; ++a)
    for (int b = 0; b < dd[1]; ++b)
      for (int cc = 0; cc < dd[2]; ++cc)
        for (int e = 0; e < dd[3]; ++e) out[o++] = src[(((size_t)a * D + b) * D + cc) * D + e];
}

/* BMPS::MultiplyMPOSVDCompress_ for the DOWN stack absorbing row `row` (bmps_impl.h:756-862), then :851-857 */
static void multiply_mpo_down(const Model *m, const int32_t *cfg, int row, Ten3 *mps, double **work, size_t *lwork) {
  const int n = m->L;
  Ten3 *res = (Ten3 *)calloc((size_t)n, sizeof(Ten3));
  double *W = xalloc((size_t)m->D * m->D * m->D * m->D);
  /* r = (comb, mpo_left, mps_left) = (1,1,1) */
  double *r = xalloc(1);
  r[0] = 1.0;
  int rm = 1, rl = 1, ra = 1;
  for (int i = 0; i < n; ++i) {
    int dd[4];
    site_dims(m, row, i, dd);
    site_tensor(m, row, i, cfg[row * n + i], W);
    const int a = mps[i].d0, p = mps[i].d1, a2 = mps[i].d2;
    const int l = dd[0], R = dd[2], U = dd[3];
    if (a != ra || l != rl || p != dd[1]) { fprintf(stderr, "cbmps: bond mismatch\n"); abort(); }
    /* G1 (:806) tmp1[(p,a2),(m,l)] = sum_a mps[a,(p,a2)] r[(m,l),a] */
    double *tmp1 = xalloc((size_t)p * a2 * rm * l);
    gemm_rm('T', 'T', p * a2, rm * l, a, mps[i].p, p * a2, r, a, tmp1, rm * l);
    /* bring (l, p) together for G2: t1t[(a2,m),(l,p)] */
    double *t1t = xalloc((size_t)a2 * rm * l * p);
    for (int ip = 0; ip < p; ++ip)
      for (int ia2 = 0; ia2 < a2; ++ia2)
        for (int im = 0; im < rm; ++im)
          for (int il = 0; il < l; ++il)
            t1t[(((size_t)ia2 * rm + im) * l + il) * p + ip] = tmp1[(((size_t)ip * a2 + ia2) * rm + im) * l + il];
    free(tmp1);
    /* G2 (:807) tmp2[(a2,m),(R,U)] = sum_{l,p} t1t[(a2,m),(l,p)] W[(l,p),(R,U)] */
    double *tmp2 = xalloc((size_t)a2 * rm * R * U);
    gemm_rm('N', 'N', a2 * rm, R * U, l * p, t1t, l * p, W, R * U, tmp2, R * U);
    free(t1t);
    if (i < n - 1) {
      /* Transpose{1,3,2,0} (:815-817): X[m,U,R,a2] */
      const int rows = rm * U, cols = R * a2;
      double *X = xalloc((size_t)rows * cols);
      for (int ia2 = 0; ia2 < a2; ++ia2)
        for (int im = 0; im < rm; ++im)
          for (int iR = 0; iR < R; ++iR)
            for (int iU = 0; iU < U; ++iU)
              X[(((size_t)im * U + iU) * R + iR) * a2 + ia2] = tmp2[(((size_t)ia2 * rm + im) * R + iR) * U + iU];
      free(tmp2);
      const int k = imin(rows, cols);
      res[i].p = xalloc((size_t)rows * k);
      res[i].d0 = rm; res[i].d1 = U; res[i].d2 = k;
      double *rn = xalloc((size_t)k * cols);
      qr_rm(X, rows, cols, res[i].p, rn, work, lwork);     /* :821 */
      free(X);
      free(r);
      r = rn; rm = k; rl = R; ra = a2;
    } else {
      /* last site (:826-849): (a2, R) are both 1 on an open boundary: res = tmp2 as (m, U, 1) */
      if (a2 != 1 || R != 1) { fprintf(stderr, "cbmps: right boundary bond is not trivial\n"); abort(); }
      res[i].p = tmp2; res[i].d0 = rm; res[i].d1 = U; res[i].d2 = 1;
      int any = 0;
      for (int e = 0; e < rm * U; ++e) any |= tmp2[e] != 0.0;
      if (!any) { fprintf(stderr, "BMPS::MultiplyMPOSVDCompress_: Empty tensor at site %d\n", i); }
    }
  }
  free(r);
  free(W);
  /* right-to-left truncation (:851-857 -> :225-263) */
  for (int i = n - 1; i >= 1; --i) {
    const int rr = res[i].d0, cc = res[i].d1 * res[i].d2, k = imin(rr, cc);
    /* row-major A (rr x cc) = col-major A^T (cc x rr): A^T = Ucm S VTcm  =>  vt_A = Ucm^T (k x cc row-major = Ucm as stored),
     * u_A = VTcm^T (rr x k row-major = VTcm as stored) */
    double *S = xalloc((size_t)k), *Ucm = xalloc((size_t)cc * k), *VTcm = xalloc((size_t)k * rr);
    int *iw = (int *)malloc(sizeof(int) * 8 * (size_t)k);
    int info = 0, lw = -1;
    double wq;
    p_dgesdd("S", &cc, &rr, res[i].p, &cc, S, Ucm, &cc, VTcm, &k, &wq, &lw, iw, &info);
    if ((size_t)wq > *lwork) { free(*work); *work = xalloc((size_t)wq); *lwork = (size_t)wq; }
    lw = (int)*lwork;
    p_dgesdd("S", &cc, &rr, res[i].p, &cc, S, Ucm, &cc, VTcm, &k, *work, &lw, iw, &info);
    if (info != 0) { fprintf(stderr, "cbmps: dgesdd info %d\n", info); abort(); }
    const int kk = imin(m->chi, k);       /* SVD(chi, chi, 0): keep min(D_max, k) */
    /* res[i] = vt (kk, phys, right): rows 0..kk of the row-major (k x cc) = Ucm col-major (cc x k) */
    double *vt = xalloc((size_t)kk * cc);
    for (int q = 0; q < kk; ++q) memcpy(vt + (size_t)q * cc, Ucm + (size_t)q * cc, sizeof(double) * cc);
    /* us (rr x kk) row-major: u_A[r][q] = VTcm_cm[q + r*k] */
    double *us = xalloc((size_t)rr * kk);
    for (int r2 = 0; r2 < rr; ++r2)
      for (int q = 0; q < kk; ++q) us[(size_t)r2 * kk + q] = VTcm[(size_t)q + (size_t)r2 * k] * S[q];
    free(res[i].p);
    res[i].p = vt; res[i].d0 = kk;
    /* res[i-1] (m', phys, rr) . us (:254) */
    const int rows = res[i - 1].d0 * res[i - 1].d1;
    double *nt = xalloc((size_t)rows * kk);
    gemm_rm('N', 'N', rows, kk, rr, res[i - 1].p, rr, us, kk, nt, kk);
    free(res[i - 1].p);
    res[i - 1].p = nt; res[i - 1].d2 = kk;
    free(S); free(Ucm); free(VTcm); free(iw); free(us);
  }
  for (int i = 0; i < n; ++i) { free(mps[i].p); mps[i] = res[i]; }
  free(res);
}

static double amplitude_one(const Model *m, const int32_t *cfg) {
  const int n = m->L;
  double *work = NULL;
  size_t lwork = 0;
  Ten3 *mps = (Ten3 *)calloc((size_t)n, sizeof(Ten3));
  for (int i = 0; i < n; ++i) { mps[i].p = xalloc(1); mps[i].p[0] = 1.0; mps[i].d0 = mps[i].d1 = mps[i].d2 = 1; }   /* bmps_impl.h:60-96 */
  for (int row = n - 1; row >= 1; --row) multiply_mpo_down(m, cfg, row, mps, &work, &lwork);   /* GrowBMPSForRow(0) */
  /* close row 0 against the DOWN boundary: E[l, a] -> E'[l', a'] = sum_{l,p,a} E[l,a] W[l,p,l'] B[a,p,a'] */
  double *W = xalloc((size_t)m->D * m->D * m->D * m->D);
  double *E = xalloc(1);
  E[0] = 1.0;
  int el = 1, ea = 1;
  for (int c = 0; c < n; ++c) {
    int dd[4];
    site_dims(m, 0, c, dd);
    site_tensor(m, 0, c, cfg[c], W);      /* (l, p, l2, 1) */
    const int l = dd[0], p = dd[1], l2 = dd[2], a = mps[c].d0, a2 = mps[c].d2;
    if (l != el || a != ea || p != mps[c].d1) { fprintf(stderr, "cbmps: top row bond mismatch\n"); abort(); }
    /* T1[l, (p, a2)] = sum_a E[l,a] B[a,(p,a2)] */
    double *T1 = xalloc((size_t)l * p * a2);
    gemm_rm('N', 'N', l, p * a2, a, E, a, mps[c].p, p * a2, T1, p * a2);
    /* E'[l2, a2] = sum_{l,p} W[(l,p), l2]^T T1[(l,p), a2] */
    double *En = xalloc((size_t)l2 * a2);
    gemm_rm('T', 'N', l2, a2, l * p, W, l2, T1, a2, En, a2);
    free(T1); free(E);
    E = En; el = l2; ea = a2;
  }
  const double amp = E[0];
  free(E); free(W); free(work);
  for (int i = 0; i < n; ++i) free(mps[i].p);
  free(mps);
  return amp;
}

typedef struct {
  const Model *m;
  const int32_t *cfgs;
  int n, tid, nthreads;
  double *out;
} Job;

static void *worker(void *arg) {
  Job *j = (Job *)arg;
  const int sites = j->m->L * j->m->L;
  for (int w = j->tid; w < j->n; w += j->nthreads) j->out[w] = amplitude_one(j->m, j->cfgs + (size_t)w * sites);
  return NULL;
}

/* amplitudes of n configurations, `nthreads` independent walkers at a time (one per thread, BLAS threads = 1);
 * seconds_out = wall time of the whole batch */
int cbmps_amplitudes(int L, int D, int d, int chi, const double *sitps_flat, int n, const int32_t *configs, int nthreads,
                     double *amps_out, double *seconds_out) {
  if (!g_blas) { snprintf(g_err, sizeof g_err, "cbmps_init not called"); return 1; }
  if (L < 2 || D < 1 || d < 1 || chi < 1 || n < 1 || nthreads < 1) { snprintf(g_err, sizeof g_err, "bad arguments"); return 2; }
  if (p_setthr) p_setthr(1);
  Model m = {L, D, d, chi, sitps_flat};
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  if (nthreads == 1) {
    Job j = {&m, configs, n, 0, 1, amps_out};
    worker(&j);
  } else {
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
    Job *jobs = (Job *)malloc(sizeof(Job) * (size_t)nthreads);
    for (int t = 0; t < nthreads; ++t) {
      jobs[t] = (Job){&m, configs, n, t, nthreads, amps_out};
      pthread_create(&th[t], NULL, worker, &jobs[t]);
    }
    for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
    free(th); free(jobs);
  }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  if (seconds_out) *seconds_out = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
  return 0;
}
