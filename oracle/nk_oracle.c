/* CPU ORACLE (C99 + OpenMP) — TEST INFRASTRUCTURE ONLY. Never linked into or called by the product
 * library; used by tests/ (checker), __graft_entry__.smoke() (checker) and bench.py's cpu_baseline leg.
 *
 * Restates, for speed at full problem sizes, the O(N) loops of the reference's Newton–Krylov step:
 *   - A*x for a concrete sparse Jacobian (the SparseMatrixCSC mul! that Krylov calls; here CSR, int32)
 *   - residual / JVP / Jacobian values of the 2-D Bratu problem and of the reference's
 *     Brusselator kernel (lib/NonlinearSolveFirstOrder/test/sparsity_tests__item1.jl:13-36)
 *   - restarted GMRES(m), MGS Arnoldi + Givens (Krylov.jl gmres, [EXT]: iterates parity-unpinned,
 *     see oracle/reference_restatement.py header)
 *   - NewtonDescent step `J δu = fu; δu *= -1; u += δu; fu = f(u)` with EisenstatWalkerForcing2 lag
 *     (lib/NonlinearSolveBase/src/descent/newton.jl:121-138,
 *      lib/NonlinearSolveFirstOrder/src/solve.jl:436-443, eisenstat_walker.jl:42-89)
 * It is validated against oracle/reference_restatement.py (NumPy) in tests/test_oracle_pins.py.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int orc_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
void orc_set_num_threads(int t) {
#ifdef _OPENMP
  omp_set_num_threads(t);
#else
  (void)t;
#endif
}

/* ------------------------------------------------------------------ BLAS-1 */
static double dot(int64_t n, const double *x, const double *y) {
  double s = 0.0;
#pragma omp parallel for reduction(+ : s) schedule(static)
  for (int64_t i = 0; i < n; ++i) s += x[i] * y[i];
  return s;
}
static void axpy(int64_t n, double a, const double *x, double *y) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) y[i] += a * x[i];
}
static double norm_inf(int64_t n, const double *x) {
  double m = 0.0;
#pragma omp parallel for reduction(max : m) schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    double a = fabs(x[i]);
    if (a > m || a != a) m = a;
  }
  return m;
}
double orc_dot(int64_t n, const double *x, const double *y) { return dot(n, x, y); }
double orc_norm_inf(int64_t n, const double *x) { return norm_inf(n, x); }

/* ------------------------------------------------------------------ CSR SpMV / transpose */
void orc_spmv(int64_t nrows, const int32_t *rowptr, const int32_t *col, const double *val,
              const double *x, double *y) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < nrows; ++i) {
    double s = 0.0;
    for (int32_t k = rowptr[i]; k < rowptr[i + 1]; ++k) s += val[k] * x[col[k]];
    y[i] = s;
  }
}
void orc_spmv_t(int64_t nrows, int64_t ncols, const int32_t *rowptr, const int32_t *col,
                const double *val, const double *x, double *y) {
  memset(y, 0, (size_t)ncols * sizeof(double));
  for (int64_t i = 0; i < nrows; ++i)
    for (int32_t k = rowptr[i]; k < rowptr[i + 1]; ++k) y[col[k]] += val[k] * x[i];
}

/* ------------------------------------------------------------------ 2-D Bratu */
typedef struct { int64_t ns; double c_lap, c_exp; } bratu_t;
static bratu_t bratu_make(int64_t ns, double lambda, double scale) {
  bratu_t b;
  double h = 1.0 / (double)(ns + 1);
  double s = (scale == 0.0) ? h * h : scale;
  b.ns = ns;
  b.c_lap = s / (h * h);
  b.c_exp = s * lambda;
  return b;
}
static inline double lap5(const double *u, int64_t ns, int64_t i, int64_t j) {
  int64_t k = j * ns + i;
  double s = 4.0 * u[k];
  if (i > 0) s -= u[k - 1];
  if (i < ns - 1) s -= u[k + 1];
  if (j > 0) s -= u[k - ns];
  if (j < ns - 1) s -= u[k + ns];
  return s;
}
void orc_bratu_residual(int64_t ns, double lambda, double scale, const double *u, double *f) {
  bratu_t b = bratu_make(ns, lambda, scale);
#pragma omp parallel for schedule(static)
  for (int64_t j = 0; j < ns; ++j)
    for (int64_t i = 0; i < ns; ++i)
      f[j * ns + i] = b.c_lap * lap5(u, ns, i, j) - b.c_exp * exp(u[j * ns + i]);
}
void orc_bratu_jvp(int64_t ns, double lambda, double scale, const double *u, const double *v, double *jv) {
  bratu_t b = bratu_make(ns, lambda, scale);
#pragma omp parallel for schedule(static)
  for (int64_t j = 0; j < ns; ++j)
    for (int64_t i = 0; i < ns; ++i) {
      int64_t k = j * ns + i;
      jv[k] = b.c_lap * lap5(v, ns, i, j) - b.c_exp * exp(u[k]) * v[k];
    }
}
/* CSR pattern: rows sorted, columns ascending (S, W, C, E, N), diagonal stored. nnz = 5N - 4ns. */
int64_t orc_bratu_nnz(int64_t ns) { return 5 * ns * ns - 4 * ns; }
void orc_bratu_pattern(int64_t ns, int32_t *rowptr, int32_t *col) {
  int64_t p = 0;
  for (int64_t j = 0; j < ns; ++j)
    for (int64_t i = 0; i < ns; ++i) {
      int64_t k = j * ns + i;
      rowptr[k] = (int32_t)p;
      if (j > 0) col[p++] = (int32_t)(k - ns);
      if (i > 0) col[p++] = (int32_t)(k - 1);
      col[p++] = (int32_t)k;
      if (i < ns - 1) col[p++] = (int32_t)(k + 1);
      if (j < ns - 1) col[p++] = (int32_t)(k + ns);
    }
  rowptr[ns * ns] = (int32_t)p;
}
void orc_bratu_jac_values(int64_t ns, double lambda, double scale, const double *u,
                          const int32_t *rowptr, double *val) {
  bratu_t b = bratu_make(ns, lambda, scale);
#pragma omp parallel for schedule(static)
  for (int64_t j = 0; j < ns; ++j)
    for (int64_t i = 0; i < ns; ++i) {
      int64_t k = j * ns + i;
      int64_t p = rowptr[k];
      if (j > 0) val[p++] = -b.c_lap;
      if (i > 0) val[p++] = -b.c_lap;
      val[p++] = 4.0 * b.c_lap - b.c_exp * exp(u[k]);
      if (i < ns - 1) val[p++] = -b.c_lap;
      if (j < ns - 1) val[p++] = -b.c_lap;
    }
}

/* ------------------------------------------------------------------ Brusselator (reference kernel) */
void orc_brusselator_residual(int64_t N, double A, double B, double alpha, double dx,
                              const double *u, double *du) {
  const double al = alpha / (dx * dx);
  const int64_t NN = N * N;
#pragma omp parallel for schedule(static)
  for (int64_t j = 0; j < N; ++j)
    for (int64_t i = 0; i < N; ++i) {
      double x = (double)i / (double)(N - 1), y = (double)j / (double)(N - 1);
      int64_t ip1 = (i + 1 == N) ? 0 : i + 1, im1 = (i == 0) ? N - 1 : i - 1;
      int64_t jp1 = (j + 1 == N) ? 0 : j + 1, jm1 = (j == 0) ? N - 1 : j - 1;
      int64_t k = i + N * j;
      double uu = u[k], vv = u[NN + k];
      double bf = (((x - 0.3) * (x - 0.3) + (y - 0.6) * (y - 0.6)) <= 0.1 * 0.1) ? 5.0 : 0.0;
      du[k] = al * (u[im1 + N * j] + u[ip1 + N * j] + u[i + N * jp1] + u[i + N * jm1] - 4.0 * uu) + B +
              uu * uu * vv - (A + 1.0) * uu + bf;
      du[NN + k] = al * (u[NN + im1 + N * j] + u[NN + ip1 + N * j] + u[NN + i + N * jp1] +
                         u[NN + i + N * jm1] - 4.0 * vv) + A * uu - uu * uu * vv;
    }
}

/* ------------------------------------------------------------------ operator dispatch for GMRES */
typedef struct {
  int kind; /* 0 = CSR, 1 = Bratu matrix-free JVP at u */
  int64_t n;
  const int32_t *rowptr, *col;
  const double *val;
  int64_t ns;
  double lambda, scale;
  const double *u;
} orc_op;

static void op_apply(const orc_op *A, const double *x, double *y) {
  if (A->kind == 0) orc_spmv(A->n, A->rowptr, A->col, A->val, x, y);
  else orc_bratu_jvp(A->ns, A->lambda, A->scale, A->u, x, y);
}

/* Chebyshev polynomial right preconditioner (Saad Alg. 12.1): y = p_d(A) v on [lmin, lmax]; work = 3 n doubles */
typedef struct { int degree; double lmin, lmax; double *work; } orc_cheb;
static void cheb_apply(const orc_op *A, const orc_cheb *C, const double *v, double *y) {
  const int64_t n = A->n;
  double *r = C->work, *d = C->work + n, *t = C->work + 2 * n;
  const double theta = 0.5 * (C->lmax + C->lmin), delta = 0.5 * (C->lmax - C->lmin), sigma1 = theta / delta;
  double rho = 1.0 / sigma1;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) { r[i] = v[i]; d[i] = v[i] / theta; y[i] = d[i]; }
  for (int k = 1; k < C->degree; ++k) {
    op_apply(A, d, t);
    const double rho_new = 1.0 / (2.0 * sigma1 - rho), c1 = rho_new * rho, c2 = 2.0 * rho_new / delta;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
      r[i] -= t[i];
      d[i] = c1 * d[i] + c2 * r[i];
      y[i] += d[i];
    }
    rho = rho_new;
  }
}
static const orc_cheb *g_cheb = NULL; /* set by the callers below; NULL = no preconditioner */
static double *g_cheb_tmp = NULL;
static void op_apply_prec(const orc_op *A, const double *x, double *y) {
  if (!g_cheb) { op_apply(A, x, y); return; }
  cheb_apply(A, g_cheb, x, g_cheb_tmp);
  op_apply(A, g_cheb_tmp, y);
}

/* GMRES(m), zero initial guess. Returns Arnoldi steps done; *converged, *rnorm0, *rnorm filled.
 * fixed_iters>0: exactly that many steps. work: (m+2)*n doubles. */
static int gmres_run(const orc_op *A, const double *b, double *x, double atol, double rtol, int m,
                     int itmax, int fixed_iters, double *work, int *converged, double *rnorm0,
                     double *rnorm_out) {
  const int64_t n = A->n;
  double *V = work;               /* (m+1) vectors */
  double *w = work + (size_t)(m + 1) * n;
  double *R = (double *)calloc((size_t)m * m, sizeof(double));
  double *cs = (double *)calloc(m, sizeof(double)), *sn = (double *)calloc(m, sizeof(double));
  double *g = (double *)calloc(m + 1, sizeof(double)), *h = (double *)calloc(m + 2, sizeof(double));
  double *yv = (double *)calloc(m, sizeof(double));
  memset(x, 0, (size_t)n * sizeof(double));
  double beta = sqrt(dot(n, b, b));
  *rnorm0 = *rnorm_out = beta;
  *converged = 0;
  int iters = 0;
  const int cap = fixed_iters > 0 ? fixed_iters : itmax;
  const double eps = fixed_iters > 0 ? -1.0 : atol + rtol * beta;
  if (beta == 0.0 || (fixed_iters <= 0 && beta <= eps)) { *converged = 1; goto out; }
  {
    const double *r = b;
    double *rbuf = NULL;
    for (;;) {
      double ib = 1.0 / beta;
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < n; ++i) V[i] = r[i] * ib;
      memset(g, 0, (size_t)(m + 1) * sizeof(double));
      g[0] = beta;
      int k = 0, done = 0;
      while (k < m && iters < cap) {
        op_apply_prec(A, V + (size_t)k * n, w);
        for (int i = 0; i <= k; ++i) { /* modified Gram–Schmidt */
          h[i] = dot(n, V + (size_t)i * n, w);
          axpy(n, -h[i], V + (size_t)i * n, w);
        }
        double hn = sqrt(dot(n, w, w));
        h[k + 1] = hn;
        for (int i = 0; i < k; ++i) {
          double t = cs[i] * h[i] + sn[i] * h[i + 1];
          h[i + 1] = -sn[i] * h[i] + cs[i] * h[i + 1];
          h[i] = t;
        }
        double d = hypot(h[k], h[k + 1]);
        if (d == 0.0) { cs[k] = 1.0; sn[k] = 0.0; } else { cs[k] = h[k] / d; sn[k] = h[k + 1] / d; }
        for (int i = 0; i < k; ++i) R[(size_t)i * m + k] = h[i];
        R[(size_t)k * m + k] = d;
        g[k + 1] = -sn[k] * g[k];
        g[k] = cs[k] * g[k];
        ++iters; ++k;
        *rnorm_out = fabs(g[k]);
        if (!(*rnorm_out == *rnorm_out) || isinf(*rnorm_out)) { done = 2; break; }
        if (fixed_iters <= 0 && *rnorm_out <= eps) { *converged = 1; done = 1; break; }
        if (hn == 0.0) { *converged = 1; done = 1; break; }
        double ih = 1.0 / hn;
        double *vn = V + (size_t)k * n;
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < n; ++i) vn[i] = w[i] * ih;
      }
      if (k > 0 && done != 2) {
        for (int i = k - 1; i >= 0; --i) {
          double s = g[i];
          for (int j = i + 1; j < k; ++j) s -= R[(size_t)i * m + j] * yv[j];
          yv[i] = s / R[(size_t)i * m + i];
        }
        for (int i = 0; i < k; ++i) axpy(n, yv[i], V + (size_t)i * n, x);
      }
      if (done || iters >= cap) { free(rbuf); break; }
      if (!rbuf) rbuf = (double *)malloc((size_t)n * sizeof(double));
      op_apply_prec(A, x, w); /* x is still in the preconditioned space here */
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < n; ++i) rbuf[i] = b[i] - w[i];
      r = rbuf;
      beta = sqrt(dot(n, r, r));
    }
  }
out:
  free(R); free(cs); free(sn); free(g); free(h); free(yv);
  if (g_cheb) { /* x = M⁻¹ z */
    cheb_apply(A, g_cheb, x, g_cheb_tmp);
    memcpy(x, g_cheb_tmp, (size_t)n * sizeof(double));
  }
  return iters;
}

int orc_gmres_csr(int64_t n, const int32_t *rowptr, const int32_t *col, const double *val,
                  const double *b, double *x, double atol, double rtol, int m, int itmax,
                  int fixed_iters, int *converged, double *rnorm0, double *rnorm) {
  orc_op A = {0, n, rowptr, col, val, 0, 0, 0, NULL};
  double *work = (double *)malloc((size_t)(m + 2) * n * sizeof(double));
  int it = gmres_run(&A, b, x, atol, rtol, m, itmax, fixed_iters, work, converged, rnorm0, rnorm);
  free(work);
  return it;
}

/* Newton–Krylov on Bratu: `nsteps` NewtonRaphson steps (no termination test inside — the caller
 * decides), linsolve = GMRES(m) on the assembled CSR (use_csr=1, values refilled every step as f.jac
 * does) or on the matrix-free JVP (use_csr=0). forcing=1: EisenstatWalkerForcing2 defaults with the
 * reference's one-step lag; forcing=0: rtol fixed. fixed_iters>0: fixed-work protocol.
 * Outputs per step: fnorm_inf[k] = ‖f(u_{k+1})‖∞, gmres_iters[k], eta[k]. u is updated in place. */
int orc_bratu_newton(int64_t ns, double lambda, double scale, double *u, int nsteps, int use_csr,
                     int m, int itmax, int fixed_iters, int forcing, double rtol, double *fnorm_inf,
                     int32_t *gmres_iters, double *eta_out) {
  const int64_t n = ns * ns;
  double *f = (double *)malloc((size_t)n * sizeof(double));
  double *dx = (double *)malloc((size_t)n * sizeof(double));
  double *work = (double *)malloc((size_t)(m + 2) * n * sizeof(double));
  int32_t *rowptr = NULL, *col = NULL;
  double *val = NULL;
  if (use_csr) {
    int64_t nnz = orc_bratu_nnz(ns);
    rowptr = (int32_t *)malloc((size_t)(n + 1) * sizeof(int32_t));
    col = (int32_t *)malloc((size_t)nnz * sizeof(int32_t));
    val = (double *)malloc((size_t)nnz * sizeof(double));
    orc_bratu_pattern(ns, rowptr, col);
  }
  orc_bratu_residual(ns, lambda, scale, u, f);
  double eta = 0.5, rn = sqrt(dot(n, f, f)), rn_prev = rn;
  const double gamma = 0.9, alpha = 2.0, eta_max = 0.9, sg_thr = 0.1;
  for (int k = 0; k < nsteps; ++k) {
    orc_op A;
    if (use_csr) {
      orc_bratu_jac_values(ns, lambda, scale, u, rowptr, val);
      orc_op t = {0, n, rowptr, col, val, 0, 0, 0, NULL};
      A = t;
    } else {
      orc_op t = {1, n, NULL, NULL, NULL, ns, lambda, scale, u};
      A = t;
    }
    double tol = rtol;
    if (forcing) {
      if (k == 0) { eta = 0.5; rn = rn_prev = sqrt(dot(n, f, f)); }
      else {
        double eprev = eta;
        eta = gamma * pow(rn / rn_prev, alpha);
        double esg = gamma * pow(eprev, alpha);
        if (esg > sg_thr && esg > eta) eta = esg;
        if (eta < 0.0) eta = 0.0;
        if (eta > eta_max) eta = eta_max;
      }
      tol = eta;
    }
    int conv; double r0, r1;
    gmres_iters[k] = gmres_run(&A, f, dx, 0.0, tol, m, itmax, fixed_iters, work, &conv, &r0, &r1);
    eta_out[k] = tol;
    if (forcing) { rn_prev = rn; rn = sqrt(dot(n, f, f)); }
    axpy(n, -1.0, dx, u);                       /* δu = -x ; u += δu */
    orc_bratu_residual(ns, lambda, scale, u, f); /* fu = f(u) */
    fnorm_inf[k] = norm_inf(n, f);
  }
  free(f); free(dx); free(work); free(rowptr); free(col); free(val);
  return 0;
}

/* Same Newton–Krylov loop with the Chebyshev(degree, ratio) right preconditioner (λmax = Gershgorin bound of the
 * Bratu Jacobian, λmin = λmax/ratio) and an early stop at ‖f‖∞ ≤ abstol: the CPU counterpart of the device
 * library's `precs` path, used for time-to-tolerance baselines. Returns the number of Newton steps taken. */
int orc_bratu_newton_cheb(int64_t ns, double lambda, double scale, double *u, int maxsteps, int use_csr, int m,
                          int itmax, int cheb_degree, double cheb_ratio, double abstol, double *fnorm_inf,
                          int32_t *gmres_iters) {
  const int64_t n = ns * ns;
  bratu_t b = bratu_make(ns, lambda, scale);
  double *f = (double *)malloc((size_t)n * sizeof(double));
  double *dx = (double *)malloc((size_t)n * sizeof(double));
  double *work = (double *)malloc((size_t)(m + 2) * n * sizeof(double));
  double *cw = (double *)malloc((size_t)4 * n * sizeof(double));
  int32_t *rowptr = NULL, *col = NULL;
  double *val = NULL;
  if (use_csr) {
    int64_t nnz = orc_bratu_nnz(ns);
    rowptr = (int32_t *)malloc((size_t)(n + 1) * sizeof(int32_t));
    col = (int32_t *)malloc((size_t)nnz * sizeof(int32_t));
    val = (double *)malloc((size_t)nnz * sizeof(double));
    orc_bratu_pattern(ns, rowptr, col);
  }
  orc_bratu_residual(ns, lambda, scale, u, f);
  double eta = 0.5, rn = sqrt(dot(n, f, f)), rn_prev = rn;
  const double gamma = 0.9, alpha = 2.0, eta_max = 0.9, sg_thr = 0.1;
  orc_cheb C = {cheb_degree, 0.0, 0.0, cw};
  int k = 0;
  for (; k < maxsteps; ++k) {
    orc_op A;
    if (use_csr) {
      orc_bratu_jac_values(ns, lambda, scale, u, rowptr, val);
      orc_op t = {0, n, rowptr, col, val, 0, 0, 0, NULL};
      A = t;
    } else {
      orc_op t = {1, n, NULL, NULL, NULL, ns, lambda, scale, u};
      A = t;
    }
    /* Gershgorin bound of J = c_lap·pentadiag − c_exp·diag(e^u): max_i (|4c − c_exp e^{u_i}| + 4c) */
    double gmax = 0.0;
    for (int64_t i = 0; i < n; ++i) {
      double d = fabs(4.0 * b.c_lap - b.c_exp * exp(u[i])) + 4.0 * b.c_lap;
      if (d > gmax) gmax = d;
    }
    C.lmax = gmax;
    C.lmin = gmax / cheb_ratio;
    g_cheb = cheb_degree > 0 ? &C : NULL;
    g_cheb_tmp = cw + 3 * n;
    if (k == 0) { eta = 0.5; rn = rn_prev = sqrt(dot(n, f, f)); }
    else {
      double eprev = eta;
      eta = gamma * pow(rn / rn_prev, alpha);
      double esg = gamma * pow(eprev, alpha);
      if (esg > sg_thr && esg > eta) eta = esg;
      if (eta < 0.0) eta = 0.0;
      if (eta > eta_max) eta = eta_max;
    }
    int conv; double r0, r1;
    gmres_iters[k] = gmres_run(&A, f, dx, 0.0, eta, m, itmax, 0, work, &conv, &r0, &r1);
    g_cheb = NULL;
    rn_prev = rn; rn = sqrt(dot(n, f, f));
    axpy(n, -1.0, dx, u);
    orc_bratu_residual(ns, lambda, scale, u, f);
    fnorm_inf[k] = norm_inf(n, f);
    if (fnorm_inf[k] <= abstol) { ++k; break; }
  }
  free(f); free(dx); free(work); free(cw); free(rowptr); free(col); free(val);
  return k;
}

/* ----------------------------------------------------------------------------------------------------------------
 * SimpleNewtonRaphson over an ensemble of small systems (lib/SimpleNonlinearSolve/src/raphson.jl:39-83 restated; see
 * oracle/reference_restatement.py::simple_newton_raphson), OpenMP over the systems. kind 0: f = u.*u .- p (n ≤ 16),
 * kind 1: the tutorial's p2_f (n = 4, docs/src/tutorials/nonlinear_solve_gpus.md:120-127). Analytic Jacobians,
 * Gaussian elimination with partial pivoting. TEST INFRASTRUCTURE / CPU baseline only. */
#define ORC_BMAX 16
static void ens_f(int kind, int n, const double *x, const double *p, double *f) {
  if (kind == 0) {
    for (int i = 0; i < n; ++i) f[i] = x[i] * x[i] - p[i];
  } else {
    f[0] = x[0] + p[0] * x[1];
    f[1] = sqrt(p[1]) * (x[2] - x[3]);
    f[2] = (x[1] - p[2] * x[2]) * (x[1] - p[2] * x[2]);
    f[3] = sqrt(p[3]) * (x[0] - x[3]) * (x[0] - x[3]);
  }
}
static void ens_jac(int kind, int n, const double *x, const double *p, double J[ORC_BMAX][ORC_BMAX]) {
  for (int i = 0; i < n; ++i)
    for (int k = 0; k < n; ++k) J[i][k] = 0.0;
  if (kind == 0) {
    for (int i = 0; i < n; ++i) J[i][i] = 2.0 * x[i];
  } else {
    const double s1 = sqrt(p[1]), s3 = sqrt(p[3]), d = x[1] - p[2] * x[2], e = x[0] - x[3];
    J[0][0] = 1.0; J[0][1] = p[0];
    J[1][2] = s1; J[1][3] = -s1;
    J[2][1] = 2.0 * d; J[2][2] = -2.0 * p[2] * d;
    J[3][0] = 2.0 * s3 * e; J[3][3] = -2.0 * s3 * e;
  }
}
static void ens_solve(int n, double A[ORC_BMAX][ORC_BMAX], double *b, double *dx) {
  for (int c = 0; c < n; ++c) {
    int piv = c;
    double best = fabs(A[c][c]);
    for (int r = c + 1; r < n; ++r)
      if (fabs(A[r][c]) > best) { best = fabs(A[r][c]); piv = r; }
    if (piv != c) {
      for (int k = c; k < n; ++k) { const double t = A[c][k]; A[c][k] = A[piv][k]; A[piv][k] = t; }
      const double t = b[c]; b[c] = b[piv]; b[piv] = t;
    }
    const double inv = 1.0 / A[c][c];
    for (int r = c + 1; r < n; ++r) {
      const double l = A[r][c] * inv;
      for (int k = c + 1; k < n; ++k) A[r][k] -= l * A[c][k];
      b[r] -= l * b[c];
    }
  }
  for (int r = n - 1; r >= 0; --r) {
    double s = b[r];
    for (int k = r + 1; k < n; ++k) s -= A[r][k] * dx[k];
    dx[r] = s / A[r][r];
  }
}
void orc_ensemble_newton(int kind, int n, int64_t nbatch, const double *u0, int u0_per_system, const double *p, int np,
                         double abstol, int maxiters, double *u_out, double *r_out, int32_t *retcode, int32_t *iters) {
#pragma omp parallel for schedule(static)
  for (int64_t b = 0; b < nbatch; ++b) {
    double x[ORC_BMAX], fx[ORC_BMAX], dx[ORC_BMAX], rhs[ORC_BMAX], J[ORC_BMAX][ORC_BMAX], A[ORC_BMAX][ORC_BMAX];
    const double *pp = p + b * np;
    for (int i = 0; i < n; ++i) x[i] = u0[(u0_per_system ? b * n : 0) + i];
    ens_f(kind, n, x, pp, fx);
    int allzero = 1, rc = 2, it = 0;
    for (int i = 0; i < n; ++i) allzero = allzero && (fx[i] == 0.0);
    if (allzero) rc = 1;
    else {
      ens_jac(kind, n, x, pp, J);
      for (it = 1; it <= maxiters; ++it) {
        for (int i = 0; i < n; ++i) { rhs[i] = fx[i]; for (int k = 0; k < n; ++k) A[i][k] = J[i][k]; }
        ens_solve(n, A, rhs, dx);
        for (int i = 0; i < n; ++i) x[i] -= dx[i];
        double nrm = 0.0; int nan = 0;
        for (int i = 0; i < n; ++i) { const double a = fabs(fx[i]); nan = nan || (a != a); nrm = a > nrm ? a : nrm; }
        if (!nan && nrm <= abstol) { rc = 1; break; }
        ens_f(kind, n, x, pp, fx);
        ens_jac(kind, n, x, pp, J);
      }
      if (it > maxiters) it = maxiters;
    }
    for (int i = 0; i < n; ++i) { u_out[b * n + i] = x[i]; r_out[b * n + i] = fx[i]; }
    retcode[b] = rc;
    iters[b] = it;
  }
}

/* ----------------------------------------------------------------------------------------------------------------
 * Tuned CPU leg for bench.py's `cpu_baseline` (TEST INFRASTRUCTURE / baseline only, like everything in this file).
 * Same fixed-work Newton step as the device benchmark — Jacobian value fill, GMRES(m) with exactly m Arnoldi steps,
 * x = V y, u −= x, residual, ‖f‖∞ — and the SAME orthogonalisation the device runs by default: CGS2 with delayed
 * re-orthogonalisation and one reduction per step (oracle/reference_restatement.py::gmres_dcgs2_1r, normalised-basis
 * form): two sweeps over the basis per Arnoldi step. Written for the host's memory system:
 *   - ONE persistent `omp parallel` region for the whole run; every loop uses the same static row partition;
 *   - parallel first touch: each thread initialises (and thereby places) its own rows of every array;
 *   - the sweeps keep a 512-row tile of u and z in L1 while the final columns stream past it;
 *   - reductions: per-thread partials, one barrier, every thread sums them in thread order (deterministic).
 * orc_stream_triad / orc_spmv_rate measure the box itself with the same partition.
 */
#include <time.h>
static double now_s(void) {
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}
#ifdef _OPENMP
#define TID() omp_get_thread_num()
#define NTHR() omp_get_num_threads()
#else
#define TID() 0
#define NTHR() 1
#endif
static void my_rows(int64_t n, int64_t gran, int t, int T, int64_t *lo, int64_t *hi) {
  int64_t units = n / gran;
  *lo = (units * t / T) * gran;
  *hi = (t == T - 1) ? n : (units * (t + 1) / T) * gran;
}

/* STREAM triad a = b + s c over 3 arrays of n doubles; returns the best GB/s of `reps` (24 n bytes per pass) */
double orc_stream_triad(int64_t n, int reps) {
  double *a = (double *)malloc((size_t)n * 8), *b = (double *)malloc((size_t)n * 8), *c = (double *)malloc((size_t)n * 8);
  double best = 0.0;
  if (!a || !b || !c) { free(a); free(b); free(c); return 0.0; }
#pragma omp parallel
  {
    int64_t lo, hi;
    my_rows(n, 8, TID(), NTHR(), &lo, &hi);
    for (int64_t i = lo; i < hi; ++i) { a[i] = 0.0; b[i] = 1.0; c[i] = 2.0; }
    for (int r = 0; r < reps; ++r) {
#pragma omp barrier
      double t0 = now_s();
      for (int64_t i = lo; i < hi; ++i) a[i] = b[i] + 3.0 * c[i];
#pragma omp barrier
      double dt = now_s() - t0;
#pragma omp master
      {
        double gb = 24.0 * (double)n / dt * 1e-9;
        if (gb > best) best = gb;
      }
    }
  }
  volatile double sink = a[n / 2];
  (void)sink;
  free(a); free(b); free(c);
  return best;
}

/* CSR SpMV rate on the Bratu ns×ns Jacobian pattern with first-touch placement: best GB/s of `reps`, in the same
 * algorithmic bytes as the device figure (12 nnz + 4 (n+1) + 16 n) */
double orc_spmv_rate(int64_t ns, int reps) {
  const int64_t n = ns * ns, nnz = orc_bratu_nnz(ns);
  int32_t *rowptr = (int32_t *)malloc((size_t)(n + 1) * 4), *col = (int32_t *)malloc((size_t)nnz * 4);
  int32_t *rp0 = (int32_t *)malloc((size_t)(n + 1) * 4), *c0 = (int32_t *)malloc((size_t)nnz * 4);
  double *val = (double *)malloc((size_t)nnz * 8), *x = (double *)malloc((size_t)n * 8), *y = (double *)malloc((size_t)n * 8);
  double best = 0.0;
  orc_bratu_pattern(ns, rp0, c0);
#pragma omp parallel
  {
    int64_t lo, hi;
    my_rows(n, ns, TID(), NTHR(), &lo, &hi);
    for (int64_t i = lo; i < hi; ++i) {
      rowptr[i] = rp0[i];
      x[i] = 1.0 + 1e-3 * (double)(i % 97);
      y[i] = 0.0;
      for (int32_t k = rp0[i]; k < rp0[i + 1]; ++k) { col[k] = c0[k]; val[k] = (c0[k] == i) ? 4.0 : -1.0; }
    }
#pragma omp master
    rowptr[n] = rp0[n];
    for (int r = 0; r < reps; ++r) {
#pragma omp barrier
      double t0 = now_s();
      for (int64_t i = lo; i < hi; ++i) {
        double s = 0.0;
        for (int32_t k = rowptr[i]; k < rowptr[i + 1]; ++k) s += val[k] * x[col[k]];
        y[i] = s;
      }
#pragma omp barrier
      double dt = now_s() - t0;
#pragma omp master
      {
        double gb = (12.0 * (double)nnz + 4.0 * (double)(n + 1) + 16.0 * (double)n) / dt * 1e-9;
        if (gb > best) best = gb;
      }
    }
  }
  volatile double sink = y[n / 2];
  (void)sink;
  free(rowptr); free(col); free(rp0); free(c0); free(val); free(x); free(y);
  return best;
}

#ifndef ORC_TILE
#define ORC_TILE 1024 /* rows of u, z kept in L1/L2 while the columns stream past */
#endif
#define ORC_MAXM 64
/* diagnostics: seconds thread 0 spent in [operator, dot sweep, reductions + scalar tail, axpy sweep, rest of the Newton step] */
static double g_phase[5];
void orc_phase_times(double *out) { for (int i = 0; i < 5; ++i) out[i] = g_phase[i]; }
#define ORC_PAD 8 /* doubles between per-thread reduction rows (false sharing) */

/* all-thread sum of cnt partials; part is [T][stride]; result in out[cnt] on every thread (identical order) */
static void team_sum(double *part, int stride, int cnt, const double *mine, double *out) {
  const int t = TID(), T = NTHR();
  for (int q = 0; q < cnt; ++q) part[(size_t)t * stride + q] = mine[q];
#pragma omp barrier
  for (int q = 0; q < cnt; ++q) {
    double s = 0.0;
    for (int r = 0; r < T; ++r) s += part[(size_t)r * stride + q];
    out[q] = s;
  }
#pragma omp barrier
}
static double team_max(double *part, int stride, double mine) {
  const int t = TID(), T = NTHR();
  part[(size_t)t * stride] = mine;
#pragma omp barrier
  double m = 0.0;
  for (int r = 0; r < T; ++r) {
    double a = part[(size_t)r * stride];
    if (a > m || a != a) m = a;
  }
#pragma omp barrier
  return m;
}

/* nsteps fixed-work Newton steps (m Arnoldi steps each, DCGS2-1R) on the assembled CSR Jacobian (use_csr=1) or the
 * matrix-free JVP. u (length ns²) is updated in place; fnorm_inf[k] = ‖f(u_{k+1})‖∞. Returns the wall seconds of the
 * step loop (set-up, first touch and the initial residual excluded, as on the device side). */
double orc_bratu_newton_fast(int64_t ns, double lambda, double scale, double *u_io, int nsteps, int use_csr, int m,
                             double *fnorm_inf) {
  const int64_t n = ns * ns, nnz = orc_bratu_nnz(ns);
  if (m < 1 || m > ORC_MAXM - 2) return -1.0;
  const bratu_t bp = bratu_make(ns, lambda, scale);
  int32_t *rowptr = NULL, *col = NULL, *rp0 = NULL, *c0 = NULL;
  double *val = NULL;
  if (use_csr) {
    rp0 = (int32_t *)malloc((size_t)(n + 1) * 4);
    c0 = (int32_t *)malloc((size_t)nnz * 4);
    orc_bratu_pattern(ns, rp0, c0);
    rowptr = (int32_t *)malloc((size_t)(n + 1) * 4);
    col = (int32_t *)malloc((size_t)nnz * 4);
    val = (double *)malloc((size_t)nnz * 8);
  }
  double *V = (double *)malloc((size_t)(m + 2) * n * 8);
  double *u = (double *)malloc((size_t)n * 8), *f = (double *)malloc((size_t)n * 8), *x = (double *)malloc((size_t)n * 8);
  int maxT = orc_num_threads();
  const int stride = 2 * ORC_MAXM + 4 + ORC_PAD;
  double *part = (double *)calloc((size_t)maxT * stride, 8);
  double elapsed = 0.0;
#pragma omp parallel
  {
    const int t = TID(), T = NTHR();
    int64_t lo, hi;
    my_rows(n, ns, t, T, &lo, &hi);
    /* ---- first touch: every thread places its own rows */
    for (int64_t i = lo; i < hi; ++i) { u[i] = u_io[i]; f[i] = 0.0; x[i] = 0.0; }
    for (int c = 0; c < m + 2; ++c) {
      double *vc = V + (size_t)c * n;
      for (int64_t i = lo; i < hi; ++i) vc[i] = 0.0;
    }
    if (use_csr) {
      for (int64_t i = lo; i < hi; ++i) {
        rowptr[i] = rp0[i];
        for (int32_t k = rp0[i]; k < rp0[i + 1]; ++k) { col[k] = c0[k]; val[k] = 0.0; }
      }
      if (t == 0) rowptr[n] = rp0[n];
    }
    /* thread-private Krylov scalars (identical on every thread) */
    double Hraw[(ORC_MAXM + 2) * (ORC_MAXM + 1)], R[ORC_MAXM * ORC_MAXM];
    double cs[ORC_MAXM], sn[ORC_MAXM], g[ORC_MAXM + 1], tprev[ORC_MAXM + 1], yv[ORC_MAXM];
    double mine[2 * ORC_MAXM + 4], red[2 * ORC_MAXM + 4];
    const int LH = ORC_MAXM + 1;
#pragma omp barrier
    /* initial residual */
    for (int64_t i = lo; i < hi; ++i) f[i] = bp.c_lap * lap5(u, ns, i % ns, i / ns) - bp.c_exp * exp(u[i]);
#pragma omp barrier
    double t0 = now_s();
    double tp = t0;
    if (t == 0) for (int i = 0; i < 5; ++i) g_phase[i] = 0.0;
#define ORC_PHASE(i) do { if (t == 0) { const double tn_ = now_s(); g_phase[i] += tn_ - tp; tp = tn_; } } while (0)
    for (int step = 0; step < nsteps; ++step) {
      /* ---- Jacobian values (f.jac): rows of this thread */
      if (use_csr) {
        for (int64_t k = lo; k < hi; ++k) {
          const int64_t i = k % ns, j = k / ns;
          int64_t p = rowptr[k];
          if (j > 0) val[p++] = -bp.c_lap;
          if (i > 0) val[p++] = -bp.c_lap;
          val[p++] = 4.0 * bp.c_lap - bp.c_exp * exp(u[k]);
          if (i < ns - 1) val[p++] = -bp.c_lap;
          if (j < ns - 1) val[p++] = -bp.c_lap;
        }
      }
      /* ---- GMRES(m), zero initial guess, exactly m Arnoldi steps: v_0 = f/‖f‖ */
      double s0 = 0.0;
      {
        double s8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int64_t i = lo;
        for (; i + 8 <= hi; i += 8)
          for (int q = 0; q < 8; ++q) s8[q] += f[i + q] * f[i + q];
        for (; i < hi; ++i) s8[0] += f[i] * f[i];
        s0 = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
      }
      team_sum(part, stride, 1, &s0, red);
      const double beta0 = sqrt(red[0]);
      const double ib = beta0 > 0.0 ? 1.0 / beta0 : 0.0;
      for (int64_t i = lo; i < hi; ++i) V[i] = f[i] * ib;
      for (int i = 0; i <= m; ++i) g[i] = 0.0;
      g[0] = beta0;
      int kdone = 0;
#pragma omp barrier
      ORC_PHASE(4);
      for (int k = 0; k <= m; ++k) {
        const int last = (k == m);
        double *uk = V + (size_t)k * n, *zk = V + (size_t)(k + 1) * n;
        /* z = A u_k on this thread's rows */
        if (!last) {
          if (use_csr) {
            for (int64_t i = lo; i < hi; ++i) {
              double s = 0.0;
              for (int32_t q = rowptr[i]; q < rowptr[i + 1]; ++q) s += val[q] * uk[col[q]];
              zk[i] = s;
            }
          } else {
            for (int64_t i = lo; i < hi; ++i)
              zk[i] = bp.c_lap * lap5(uk, ns, i % ns, i / ns) - bp.c_exp * exp(u[i]) * uk[i];
          }
        }
        ORC_PHASE(0);
        /* dot sweep: red = [V_jᵀu (k), u·u, V_jᵀz (k), u·z] */
        const int cnt = last ? k + 1 : 2 * k + 2;
        for (int q = 0; q < cnt; ++q) mine[q] = 0.0;
        /* eight independent partial sums per inner product, combined in a fixed order: the compiler can keep them in vector
         * registers without re-associating anything (a single scalar accumulator is a 4-cycle dependent-add chain per element) */
        for (int64_t r0 = lo; r0 < hi; r0 += ORC_TILE) {
          const int64_t r1 = r0 + ORC_TILE < hi ? r0 + ORC_TILE : hi;
          const int len = (int)(r1 - r0), len8 = len & ~7;
          const double *ut = uk + r0, *zt = zk + r0;
          {
            double a8[8] = {0, 0, 0, 0, 0, 0, 0, 0}, d8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int i = 0; i < len8; i += 8)
              for (int q = 0; q < 8; ++q) { a8[q] += ut[i + q] * ut[i + q]; if (!last) d8[q] += ut[i + q] * zt[i + q]; }
            for (int i = len8; i < len; ++i) { a8[0] += ut[i] * ut[i]; if (!last) d8[0] += ut[i] * zt[i]; }
            mine[k] += ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
            if (!last) mine[2 * k + 1] += ((d8[0] + d8[1]) + (d8[2] + d8[3])) + ((d8[4] + d8[5]) + (d8[6] + d8[7]));
          }
          if (!last) {
            /* four columns at a time share the loads of the u, z tile; `omp simd reduction` lets the compiler keep eight
             * vector accumulators (its association is fixed by the build, so runs are reproducible) */
            int j = 0;
            for (; j + 4 <= k; j += 4) {
              const double *v0 = V + (size_t)j * n + r0, *v1 = v0 + n, *v2 = v1 + n, *v3 = v2 + n;
              double a0 = 0, a1 = 0, a2 = 0, a3 = 0, b0 = 0, b1 = 0, b2 = 0, b3 = 0;
#pragma omp simd reduction(+ : a0, a1, a2, a3, b0, b1, b2, b3)
              for (int i = 0; i < len; ++i) {
                const double uu = ut[i], zz = zt[i];
                a0 += v0[i] * uu; b0 += v0[i] * zz;
                a1 += v1[i] * uu; b1 += v1[i] * zz;
                a2 += v2[i] * uu; b2 += v2[i] * zz;
                a3 += v3[i] * uu; b3 += v3[i] * zz;
              }
              mine[j] += a0; mine[j + 1] += a1; mine[j + 2] += a2; mine[j + 3] += a3;
              mine[k + 1 + j] += b0; mine[k + 2 + j] += b1; mine[k + 3 + j] += b2; mine[k + 4 + j] += b3;
            }
            for (; j < k; ++j) {
              const double *vj = V + (size_t)j * n + r0;
              double a0 = 0, b0 = 0;
#pragma omp simd reduction(+ : a0, b0)
              for (int i = 0; i < len; ++i) { a0 += vj[i] * ut[i]; b0 += vj[i] * zt[i]; }
              mine[j] += a0;
              mine[k + 1 + j] += b0;
            }
          }
        }
        ORC_PHASE(1);
        team_sum(part, stride, cnt, mine, red);
        if (k == 0) { /* v_0 is final: only the first projection of A v_0 */
          const double tl = red[1];
          for (int64_t i = lo; i < hi; ++i) zk[i] -= tl * uk[i];
          tprev[0] = tl;
#pragma omp barrier
          continue;
        }
        /* ---- scalar tail (every thread, identical): close Hessenberg column k−1 */
        double rr[ORC_MAXM], h[ORC_MAXM + 1], cc[ORC_MAXM + 1];
        double r2 = 0.0;
        for (int j = 0; j < k; ++j) { rr[j] = last ? 0.0 : red[j]; r2 += rr[j] * rr[j]; }
        double b2 = red[k] - r2;
        if (b2 < 0.0) b2 = 0.0;
        const double beta = sqrt(b2);
        for (int j = 0; j < k; ++j) h[j] = tprev[j] + rr[j];
        h[k] = beta;
        for (int j = 0; j <= k; ++j) Hraw[j * LH + (k - 1)] = h[j];
        {
          const int jc = k - 1;
          for (int i = 0; i < jc; ++i) {
            const double tt = cs[i] * h[i] + sn[i] * h[i + 1];
            h[i + 1] = -sn[i] * h[i] + cs[i] * h[i + 1];
            h[i] = tt;
          }
          const double dd = hypot(h[jc], h[jc + 1]);
          if (dd == 0.0) { cs[jc] = 1.0; sn[jc] = 0.0; } else { cs[jc] = h[jc] / dd; sn[jc] = h[jc + 1] / dd; }
          for (int i = 0; i < jc; ++i) R[i * ORC_MAXM + jc] = h[i];
          R[jc * ORC_MAXM + jc] = dd;
          g[jc + 1] = -sn[jc] * g[jc];
          g[jc] = cs[jc] * g[jc];
          kdone = jc + 1;
        }
        if (last) break;
        const double *gg = red + k + 1;
        const double dval = red[2 * k + 1];
        double rg = 0.0;
        for (int j = 0; j < k; ++j) rg += rr[j] * gg[j];
        for (int i = 0; i <= k; ++i) { /* c = H̄[0:k+1, 0:k] r (entries below the sub-diagonal are zero) */
          double s = 0.0;
          for (int j = (i > 0 ? i - 1 : 0); j < k; ++j) s += Hraw[i * LH + j] * rr[j];
          cc[i] = s;
        }
        const double isb = beta > 0.0 ? 1.0 / beta : 0.0;
        double ca[ORC_MAXM + 1], cb[ORC_MAXM + 1];
        for (int j = 0; j < k; ++j) {
          const double tt = (gg[j] - cc[j]) * isb;
          tprev[j] = tt;
          ca[j] = rr[j];                 /* v_k   = (u − Σ ca_j v_j) / β */
          cb[j] = cc[j] * isb + tt;      /* u_k+1 = z/β − Σ cb_j v_j − cbk v_k */
        }
        const double tlast = (dval - rg - beta * cc[k]) * isb * isb;
        tprev[k] = tlast;
        const double cbk = cc[k] * isb + tlast;
        ORC_PHASE(2);
        /* ---- axpy sweep (tile of u, z kept in L1 while the final columns stream past) */
        for (int64_t r0 = lo; r0 < hi; r0 += ORC_TILE) {
          const int64_t r1 = r0 + ORC_TILE < hi ? r0 + ORC_TILE : hi;
          double pu[ORC_TILE], pz[ORC_TILE];
          const int len = (int)(r1 - r0);
          for (int i = 0; i < len; ++i) { pu[i] = uk[r0 + i]; pz[i] = zk[r0 + i] * isb; }
          {
            int j = 0;
            for (; j + 4 <= k; j += 4) {  /* four columns per pass over the tile */
              const double *v0 = V + (size_t)j * n + r0, *v1 = v0 + n, *v2 = v1 + n, *v3 = v2 + n;
              const double a0 = ca[j], a1 = ca[j + 1], a2 = ca[j + 2], a3 = ca[j + 3];
              const double b0 = cb[j], b1 = cb[j + 1], b2 = cb[j + 2], b3 = cb[j + 3];
#pragma omp simd
              for (int i = 0; i < len; ++i) {
                pu[i] -= (a0 * v0[i] + a1 * v1[i]) + (a2 * v2[i] + a3 * v3[i]);
                pz[i] -= (b0 * v0[i] + b1 * v1[i]) + (b2 * v2[i] + b3 * v3[i]);
              }
            }
            for (; j < k; ++j) {
              const double *vj = V + (size_t)j * n + r0;
              const double a = ca[j], b = cb[j];
#pragma omp simd
              for (int i = 0; i < len; ++i) { pu[i] -= a * vj[i]; pz[i] -= b * vj[i]; }
            }
          }
          for (int i = 0; i < len; ++i) {
            const double vk = pu[i] * isb;
            uk[r0 + i] = vk;
            zk[r0 + i] = pz[i] - cbk * vk;
          }
        }
        ORC_PHASE(3);
#pragma omp barrier
        ORC_PHASE(2);
      }
      /* ---- y = R⁻¹ g ; x = V y ; u −= x ; f = F(u) ; ‖f‖∞ */
      for (int i = kdone - 1; i >= 0; --i) {
        double s = g[i];
        for (int j = i + 1; j < kdone; ++j) s -= R[i * ORC_MAXM + j] * yv[j];
        yv[i] = s / R[i * ORC_MAXM + i];
      }
      for (int64_t r0 = lo; r0 < hi; r0 += ORC_TILE) {
        const int64_t r1 = r0 + ORC_TILE < hi ? r0 + ORC_TILE : hi;
        double acc[ORC_TILE];
        const int len = (int)(r1 - r0);
        for (int i = 0; i < len; ++i) acc[i] = 0.0;
        for (int j = 0; j < kdone; ++j) {
          const double *vj = V + (size_t)j * n + r0;
          const double yj = yv[j];
          for (int i = 0; i < len; ++i) acc[i] += yj * vj[i];
        }
        for (int i = 0; i < len; ++i) { x[r0 + i] = acc[i]; u[r0 + i] -= acc[i]; }
      }
#pragma omp barrier
      double mx = 0.0;
      for (int64_t i = lo; i < hi; ++i) {
        const double fi = bp.c_lap * lap5(u, ns, i % ns, i / ns) - bp.c_exp * exp(u[i]);
        f[i] = fi;
        const double a = fabs(fi);
        if (a > mx || a != a) mx = a;
      }
      mx = team_max(part, stride, mx);
      if (t == 0) fnorm_inf[step] = mx;
    }
#pragma omp barrier
    if (t == 0) elapsed = now_s() - t0;
    for (int64_t i = lo; i < hi; ++i) u_io[i] = u[i];
  }
  free(rowptr); free(col); free(rp0); free(c0); free(val); free(V); free(u); free(f); free(x); free(part);
  return elapsed;
}

/* ----------------------------------------------------------------------------------------------------------------
 * The same fixed-work Newton step with the s-step Arnoldi process (the device's NK_ORTHO_SSTEP; the NumPy restatement is
 * oracle/reference_restatement.py::gmres_sstep): per block of sb ≤ s columns the monomial vectors X_j = A X_{j−1}
 * (X_0 = A v_k), then twice [C ; G] = [V_k X]ᵀX in ONE team reduction, RᵀR = G − CᵀC, X ← (X − V_k C) R⁻¹; the Hessenberg
 * columns follow from C = C₁ + C₂R₁, R = R₂R₁ and the old columns. Same host-side structure as orc_bratu_newton_fast (one
 * persistent parallel region, static row partition, first touch, row tiles kept in cache while the columns stream past).
 * Returns the wall seconds of the step loop, −1 on bad arguments, −2 when a block loses rank numerically (where the device
 * falls back to the column-by-column scheme). TEST INFRASTRUCTURE / baseline only.
 */
#define ORC_SMAX 16
/* S = RᵀR (upper R), Ri = R⁻¹; row-major sb×sb. Gd = the diagonal of XᵀX before the projection: a pivot that is not positive
 * relative to it (d ≤ 1e-12 (XᵀX)_aa) means the block lost rank numerically — the device's test (csrc/nk_sstep.hip::ss_factor) */
static int ss_chol_inv(int sb, const double *S, const double *Gd, double *Rm, double *Ri) {
  for (int e = 0; e < sb * sb; ++e) { Rm[e] = 0.0; Ri[e] = 0.0; }
  for (int a = 0; a < sb; ++a) {
    double d = S[a * sb + a];
    for (int p = 0; p < a; ++p) d -= Rm[p * sb + a] * Rm[p * sb + a];
    if (!(d > 1e-12 * Gd[a]) || isinf(d)) return 0;
    const double raa = sqrt(d);
    Rm[a * sb + a] = raa;
    for (int b = a + 1; b < sb; ++b) {
      double v = 0.5 * (S[a * sb + b] + S[b * sb + a]);
      for (int p = 0; p < a; ++p) v -= Rm[p * sb + a] * Rm[p * sb + b];
      Rm[a * sb + b] = v / raa;
    }
  }
  for (int b = 0; b < sb; ++b) {
    Ri[b * sb + b] = 1.0 / Rm[b * sb + b];
    for (int a = b - 1; a >= 0; --a) {
      double v = 0.0;
      for (int p = a + 1; p <= b; ++p) v -= Rm[a * sb + p] * Ri[p * sb + b];
      Ri[a * sb + b] = v / Rm[a * sb + a];
    }
  }
  return 1;
}
/* Chebyshev points of [−1, 1] in Leja order (csrc/nk_sstep.hip::nk_ss_leja_nodes, same loop) */
void orc_leja_nodes(int s, double *out) {
  double pts[ORC_SMAX];
  int used[ORC_SMAX];
  for (int i = 0; i < s; ++i) { pts[i] = cos((2.0 * i + 1.0) * 3.14159265358979323846 / (2.0 * s)); used[i] = 0; }
  for (int j = 0; j < s; ++j) {
    int best = -1;
    double bv = -1.0;
    for (int i = 0; i < s; ++i) {
      if (used[i]) continue;
      double v = (j == 0) ? fabs(pts[i]) : 1.0;
      for (int q = 0; q < j; ++q) v *= fabs(pts[i] - out[q]);
      if (v > bv) { bv = v; best = i; }
    }
    used[best] = 1;
    out[j] = pts[best];
  }
}
/* block widths the device's sweeps are compiled for (csrc/nk_sstep.hip::nk_ss_block_width) */
static int ss_block_width(int want) {
  if (want >= 15) return 15;
  if (want >= 8) return 8;
  if (want >= 6) return 6;
  if (want >= 4) return 4;
  return want >= 2 ? 2 : 1;
}
/* basis: 0 = monomial X_j = A X_{j−1}; 1 = Newton X_j = (A − θ_j I) X_{j−1} / σ with θ = Leja-ordered Chebyshev points of
 * the Gershgorin interval of J (CSR: the discs of the assembled rows; matrix-free: [−max d, 8c − min d], d = c_exp·eᵘ) and
 * σ = (hi − lo)/4 rounded to a power of two — recomputed for every Jacobian, as the device does. */
double orc_bratu_newton_fast_sstep2(int64_t ns, double lambda, double scale, double *u_io, int nsteps, int use_csr, int m,
                                    int s, int basis, double *fnorm_inf) {
  const int64_t n = ns * ns, nnz = orc_bratu_nnz(ns);
  if (m < 1 || m > ORC_MAXM - 2 || s < 1 || s > ORC_SMAX || (s > 8 && s != ss_block_width(s) && s != 16)) return -1.0;
  if (s == 16) return -1.0;
  double nodes[ORC_SMAX] = {0};
  if (basis) orc_leja_nodes(s, nodes);
  const bratu_t bp = bratu_make(ns, lambda, scale);
  int32_t *rowptr = NULL, *col = NULL, *rp0 = NULL, *c0 = NULL;
  double *val = NULL;
  if (use_csr) {
    rp0 = (int32_t *)malloc((size_t)(n + 1) * 4);
    c0 = (int32_t *)malloc((size_t)nnz * 4);
    orc_bratu_pattern(ns, rp0, c0);
    rowptr = (int32_t *)malloc((size_t)(n + 1) * 4);
    col = (int32_t *)malloc((size_t)nnz * 4);
    val = (double *)malloc((size_t)nnz * 8);
  }
  double *V = (double *)malloc((size_t)(m + 2) * n * 8);
  double *u = (double *)malloc((size_t)n * 8), *f = (double *)malloc((size_t)n * 8);
  const int maxT = orc_num_threads();
  const int KS = (ORC_MAXM + ORC_SMAX) * ORC_SMAX;  /* entries of a reduced block */
  const int stride = KS + ORC_PAD;
  double *part = (double *)calloc((size_t)maxT * stride, 8);
  double elapsed = 0.0;
  int broke = 0;
#pragma omp parallel
  {
    const int t = TID(), T = NTHR();
    int64_t lo, hi;
    my_rows(n, ns, t, T, &lo, &hi);
    for (int64_t i = lo; i < hi; ++i) { u[i] = u_io[i]; f[i] = 0.0; }
    for (int c = 0; c < m + 2; ++c) {
      double *vc = V + (size_t)c * n;
      for (int64_t i = lo; i < hi; ++i) vc[i] = 0.0;
    }
    if (use_csr) {
      for (int64_t i = lo; i < hi; ++i) {
        rowptr[i] = rp0[i];
        for (int32_t k = rp0[i]; k < rp0[i + 1]; ++k) { col[k] = c0[k]; val[k] = 0.0; }
      }
      if (t == 0) rowptr[n] = rp0[n];
    }
    /* thread-private Krylov scalars (identical on every thread) */
    double H[(ORC_MAXM + ORC_SMAX + 2) * ORC_MAXM], R[ORC_MAXM * ORC_MAXM];
    double cs[ORC_MAXM], sn[ORC_MAXM], g[ORC_MAXM + 1], yv[ORC_MAXM];
    double mine[(ORC_MAXM + ORC_SMAX) * ORC_SMAX], red[(ORC_MAXM + ORC_SMAX) * ORC_SMAX];
    double C1[ORC_MAXM * ORC_SMAX], R1[ORC_SMAX * ORC_SMAX], Ct[ORC_MAXM * ORC_SMAX], Rm[ORC_SMAX * ORC_SMAX],
        Ri[ORC_SMAX * ORC_SMAX], Sm[ORC_SMAX * ORC_SMAX], F[(ORC_MAXM + ORC_SMAX) * ORC_SMAX],
        NC[ORC_SMAX * (ORC_MAXM + ORC_SMAX)], Gd[ORC_SMAX], th[ORC_SMAX];
    double sigma = 1.0, isig = 1.0;
    for (int j = 0; j < ORC_SMAX; ++j) th[j] = 0.0;
    int bad = 0;
#pragma omp barrier
    for (int64_t i = lo; i < hi; ++i) f[i] = bp.c_lap * lap5(u, ns, i % ns, i / ns) - bp.c_exp * exp(u[i]);
#pragma omp barrier
    const double t0 = now_s();
    for (int step = 0; step < nsteps && !bad; ++step) {
      if (use_csr) {
        for (int64_t k = lo; k < hi; ++k) {
          const int64_t i = k % ns, j = k / ns;
          int64_t p = rowptr[k];
          if (j > 0) val[p++] = -bp.c_lap;
          if (i > 0) val[p++] = -bp.c_lap;
          val[p++] = 4.0 * bp.c_lap - bp.c_exp * exp(u[k]);
          if (i < ns - 1) val[p++] = -bp.c_lap;
          if (j < ns - 1) val[p++] = -bp.c_lap;
        }
      }
      if (basis) { /* bounds of J's spectrum → shifts and scale of the block basis */
        double mlo = -INFINITY, mhi = -INFINITY;
        if (use_csr) {
          for (int64_t i = lo; i < hi; ++i) {
            double rad = 0.0, d = 0.0;
            for (int32_t q = rowptr[i]; q < rowptr[i + 1]; ++q) {
              if (col[q] == i) d += val[q];
              else rad += fabs(val[q]);
            }
            if (-(d - rad) > mlo) mlo = -(d - rad);
            if (d + rad > mhi) mhi = d + rad;
          }
        } else {
          for (int64_t i = lo; i < hi; ++i) {
            const double d = bp.c_exp * exp(u[i]);
            if (d > mlo) mlo = d;
            if (-d > mhi) mhi = -d;
          }
        }
        mlo = team_max(part, stride, mlo);
        mhi = team_max(part, stride, mhi);
        if (!use_csr) mhi = 8.0 * bp.c_lap + mhi;
        const double ilo = -mlo, ihi = mhi, cc = 0.5 * (ilo + ihi), hh = 0.5 * (ihi - ilo);
        if (hh > 0.0 && !isinf(hh)) {
          for (int j = 0; j < s; ++j) th[j] = cc + hh * nodes[j];
          sigma = exp2(rint(log2(0.5 * hh)));
        } else {
          for (int j = 0; j < s; ++j) th[j] = 0.0;
          sigma = 1.0;
        }
        isig = 1.0 / sigma;
      }
      double s0 = 0.0;
      for (int64_t i = lo; i < hi; ++i) s0 += f[i] * f[i];
      team_sum(part, stride, 1, &s0, red);
      const double beta0 = sqrt(red[0]);
      const double ib = beta0 > 0.0 ? 1.0 / beta0 : 0.0;
      for (int64_t i = lo; i < hi; ++i) V[i] = f[i] * ib;
      for (int i = 0; i <= m; ++i) g[i] = 0.0;
      g[0] = beta0;
      int kdone = 0;
#pragma omp barrier
      int k = 1; /* orthonormal columns so far */
      while (k - 1 < m && !bad) {
        const int sb = ss_block_width((m - (k - 1)) < s ? (m - (k - 1)) : s), K = k + sb, ko = k - 1;
        double *X = V + (size_t)k * n;
        /* ---- matrix powers: X_j = A X_{j−1}, X_0 = A v_k (a barrier after each: the stencil reaches other threads' rows) */
        for (int j = 0; j < sb; ++j) {
          const double *src = V + (size_t)(k - 1 + j) * n;
          double *dst = X + (size_t)j * n;
          const double thj = th[j];
          if (use_csr) {
            for (int64_t i = lo; i < hi; ++i) {
              double a = 0.0;
              for (int32_t q = rowptr[i]; q < rowptr[i + 1]; ++q) a += val[q] * src[col[q]];
              dst[i] = basis ? isig * (a - thj * src[i]) : a;
            }
          } else {
            for (int64_t i = lo; i < hi; ++i) {
              const double a = bp.c_lap * lap5(src, ns, i % ns, i / ns) - bp.c_exp * exp(u[i]) * src[i];
              dst[i] = basis ? isig * (a - thj * src[i]) : a;
            }
          }
#pragma omp barrier
        }
        for (int pass = 0; pass < 2 && !bad; ++pass) {
          /* ---- [V_k X]ᵀ X on this thread's rows: the X tile stays in cache while the basis columns stream past */
          const int cnt = K * sb;
          for (int q = 0; q < cnt; ++q) mine[q] = 0.0;
          for (int64_t r0 = lo; r0 < hi; r0 += ORC_TILE) {
            const int64_t r1 = r0 + ORC_TILE < hi ? r0 + ORC_TILE : hi;
            const int len = (int)(r1 - r0);
            for (int j = 0; j < K; ++j) {
              const double *vj = V + (size_t)j * n + r0;
              for (int c = (j < k ? 0 : j - k); c < sb; ++c) {  /* the Gram part: upper triangle only */
                const double *xc = X + (size_t)c * n + r0;
                double a = 0.0;
#pragma omp simd reduction(+ : a)
                for (int i = 0; i < len; ++i) a += vj[i] * xc[i];
                mine[j * sb + c] += a;
              }
            }
          }
          team_sum(part, stride, cnt, mine, red);
          /* ---- scalar work (every thread, identical) */
          for (int j = 0; j < k; ++j)
            for (int c = 0; c < sb; ++c) Ct[j * sb + c] = red[j * sb + c];
          for (int a = 0; a < sb; ++a)
            for (int c = 0; c < sb; ++c) {
              double v = (c >= a) ? red[(k + a) * sb + c] : red[(k + c) * sb + a];
              for (int j = 0; j < k; ++j) v -= Ct[j * sb + a] * Ct[j * sb + c];
              Sm[a * sb + c] = v;
            }
          for (int a = 0; a < sb; ++a) Gd[a] = red[(k + a) * sb + a];
          if (!ss_chol_inv(sb, Sm, Gd, Rm, Ri)) { bad = 1; break; }
          /* ---- X ← (X − V_k Ct) R⁻¹ on this thread's rows */
          for (int64_t r0 = lo; r0 < hi; r0 += ORC_TILE) {
            const int64_t r1 = r0 + ORC_TILE < hi ? r0 + ORC_TILE : hi;
            const int len = (int)(r1 - r0);
            double xt[ORC_SMAX][ORC_TILE];
            for (int c = 0; c < sb; ++c) {
              const double *xc = X + (size_t)c * n + r0;
              for (int i = 0; i < len; ++i) xt[c][i] = xc[i];
            }
            for (int j = 0; j < k; ++j) {
              const double *vj = V + (size_t)j * n + r0;
              for (int c = 0; c < sb; ++c) {
                const double cf = Ct[j * sb + c];
                double *xc = xt[c];
#pragma omp simd
                for (int i = 0; i < len; ++i) xc[i] -= cf * vj[i];
              }
            }
            for (int c = sb - 1; c >= 0; --c) {  /* in place, last column first: column c needs the old columns ≤ c */
              double *xo = X + (size_t)c * n + r0;
              for (int i = 0; i < len; ++i) {
                double a = 0.0;
                for (int cc = 0; cc <= c; ++cc) a += xt[cc][i] * Ri[cc * sb + c];
                xo[i] = a;
              }
            }
          }
          if (pass == 0) {
            for (int e = 0; e < k * sb; ++e) C1[e] = Ct[e];
            for (int e = 0; e < sb * sb; ++e) R1[e] = Rm[e];
          }
#pragma omp barrier
        }
        if (bad) break;
        /* ---- C = C₁ + C₂R₁, R = R₂R₁ → F = [C ; R]; Hessenberg columns; Givens */
        for (int j = 0; j < k; ++j)
          for (int c = 0; c < sb; ++c) {
            double v = C1[j * sb + c];
            for (int a = 0; a <= c; ++a) v += Ct[j * sb + a] * R1[a * sb + c];
            F[j * sb + c] = v;
          }
        for (int a = 0; a < sb; ++a)
          for (int c = 0; c < sb; ++c) {
            double v = 0.0;
            for (int p = a; p <= c; ++p) v += Rm[a * sb + p] * R1[p * sb + c];
            F[(k + a) * sb + c] = (c >= a) ? v : 0.0;
          }
        for (int i = 0; i < K; ++i) NC[i] = sigma * F[i * sb] + (i == k - 1 ? th[0] : 0.0);
        for (int j = 1; j < sb; ++j)
          for (int i = 0; i < K; ++i) {
            double a = sigma * F[i * sb + j] + th[j] * F[i * sb + (j - 1)];
            if (i < k)
              for (int tt = (i > 0 ? i - 1 : 0); tt < ko; ++tt) a -= H[i * ORC_MAXM + tt] * F[tt * sb + (j - 1)];
            a -= NC[i] * F[(k - 1) * sb + (j - 1)];
            for (int q = 1; q < j; ++q) a -= NC[q * K + i] * F[(k + q - 1) * sb + (j - 1)];
            NC[j * K + i] = a / F[(k + j - 1) * sb + (j - 1)];
          }
        for (int j = 0; j < sb; ++j) {
          const int jc = ko + j;
          double *h = &NC[j * K];
          for (int i = 0; i <= jc + 1; ++i) H[i * ORC_MAXM + jc] = h[i];
          for (int i = 0; i < jc; ++i) {
            const double tt = cs[i] * h[i] + sn[i] * h[i + 1];
            h[i + 1] = -sn[i] * h[i] + cs[i] * h[i + 1];
            h[i] = tt;
          }
          const double dd = hypot(h[jc], h[jc + 1]);
          if (dd == 0.0) { cs[jc] = 1.0; sn[jc] = 0.0; } else { cs[jc] = h[jc] / dd; sn[jc] = h[jc + 1] / dd; }
          for (int i = 0; i < jc; ++i) R[i * ORC_MAXM + jc] = h[i];
          R[jc * ORC_MAXM + jc] = dd;
          g[jc + 1] = -sn[jc] * g[jc];
          g[jc] = cs[jc] * g[jc];
          kdone = jc + 1;
        }
        k += sb;
      }
      if (bad) break;
      /* ---- y = R⁻¹ g ; x = V y ; u −= x ; f = F(u) ; ‖f‖∞ */
      for (int i = kdone - 1; i >= 0; --i) {
        double a = g[i];
        for (int j = i + 1; j < kdone; ++j) a -= R[i * ORC_MAXM + j] * yv[j];
        yv[i] = a / R[i * ORC_MAXM + i];
      }
      for (int64_t r0 = lo; r0 < hi; r0 += ORC_TILE) {
        const int64_t r1 = r0 + ORC_TILE < hi ? r0 + ORC_TILE : hi;
        double acc[ORC_TILE];
        const int len = (int)(r1 - r0);
        for (int i = 0; i < len; ++i) acc[i] = 0.0;
        for (int j = 0; j < kdone; ++j) {
          const double *vj = V + (size_t)j * n + r0;
          const double yj = yv[j];
          for (int i = 0; i < len; ++i) acc[i] += yj * vj[i];
        }
        for (int i = 0; i < len; ++i) u[r0 + i] -= acc[i];
      }
#pragma omp barrier
      double mx = 0.0;
      for (int64_t i = lo; i < hi; ++i) {
        const double fi = bp.c_lap * lap5(u, ns, i % ns, i / ns) - bp.c_exp * exp(u[i]);
        f[i] = fi;
        const double a = fabs(fi);
        if (a > mx || a != a) mx = a;
      }
      mx = team_max(part, stride, mx);
      if (t == 0) fnorm_inf[step] = mx;
    }
    if (bad && t == 0) broke = 1;
#pragma omp barrier
    if (t == 0) elapsed = now_s() - t0;
    for (int64_t i = lo; i < hi; ++i) u_io[i] = u[i];
  }
  free(rowptr); free(col); free(rp0); free(c0); free(val); free(V); free(u); free(f); free(part);
  return broke ? -2.0 : elapsed;
}
double orc_bratu_newton_fast_sstep(int64_t ns, double lambda, double scale, double *u_io, int nsteps, int use_csr, int m,
                                   int s, double *fnorm_inf) {
  return orc_bratu_newton_fast_sstep2(ns, lambda, scale, u_io, nsteps, use_csr, m, s, 0, fnorm_inf);
}
