// Device-resident restarted GMRES(m) (seam 1: replaces solve!(cache.lincache) at
// lib/NonlinearSolveBase/ext/NonlinearSolveBaseLinearSolveExt.jl:26, i.e. LinearSolve.KrylovJL_GMRES →
// Krylov.gmres! [EXT]).
//
// Structure: the Krylov basis V (n × (m+1), column-major), the Hessenberg factor, the Givens rotations and the
// least-squares right-hand side all live in device memory. A whole restart cycle is enqueued without any
// host synchronisation: the small `k_givens` kernel decides convergence on the device and raises `ctl.done`,
// which every later kernel of the cycle reads first and returns on. The host reads the 128-byte control
// block once per cycle. Orthogonalisation: MGS (Krylov.jl's structure), CGS2 (two fused passes; 3 small
// all-reduces per Arnoldi step on multi-GPU) or CGS with the DGKS re-orthogonalisation test.
#include <math.h>
#include <string.h>

#include "nk_internal.h"

// ----------------------------------------------------------------------------- small device kernels
__global__ void k_gmres_begin(nk_gmres_ctl *ctl, const double *d_ss, double atol, double rtol, int fixed,
                              int first, double *g, int m) {
  if (threadIdx.x != 0) return;
  const double beta = sqrt(*d_ss);
  if (first) {
    ctl->rnorm0 = beta;
    ctl->tol = fixed ? -1.0 : atol + rtol * beta;
    ctl->failed = 0;
    ctl->converged = 0;
  }
  ctl->beta = beta;
  ctl->rnorm = beta;
  ctl->k = 0;
  ctl->need_reorth = 0;
  ctl->pad0 = 0;
  const int bad = !(beta == beta) || isinf(beta);
  if (bad) ctl->failed = 1;
  if (!bad && (beta == 0.0 || (ctl->tol >= 0.0 && beta <= ctl->tol))) ctl->converged = 1;
  ctl->done = (ctl->failed || ctl->converged) ? 1 : 0;
  ctl->inv_hn = (beta > 0.0 && !bad) ? 1.0 / beta : 0.0;
  g[0] = beta;
  for (int i = 1; i <= m; ++i) g[i] = 0.0;
}

// DGKS test after the first CGS pass: re-orthogonalise iff ‖w'‖² < ½‖w‖².  pad0 doubles as the
// "skip pass 2" flag read by the pass-2 kernels.
__global__ void k_dgks(nk_gmres_ctl *ctl, const double *h /*h[k+1] = ‖w‖²*/, const double *d_ss1, double *h2,
                       double *d_ss, double inv_nranks) {
  if (threadIdx.x != 0) return;
  if (ctl->done) { ctl->pad0 = 1; return; }
  const int k = ctl->k;
  const double before = h[k + 1], after = *d_ss1;
  const int need = (after < 0.5 * before) ? 1 : 0;
  ctl->need_reorth = need;
  ctl->pad0 = need ? 0 : 1;
  if (!need) {
    for (int i = 0; i <= k; ++i) h2[i] = 0.0;
    *d_ss = after * inv_nranks;  // the unconditional all-reduce that follows restores `after`
  }
}

// new Hessenberg column → apply old rotations, create the new one, update g and the residual estimate
__global__ __launch_bounds__(64) void k_givens(nk_gmres_ctl *ctl, double *h, const double *h2, const double *d_ss,
                                               double *R, double *cs, double *sn, double *g, int m) {
  if (ctl->done) return;
  __shared__ double sh[NK_MAX_NV + 2], sc[NK_MAX_NV + 2], ss_[NK_MAX_NV + 2];
  const int k = ctl->k, t = threadIdx.x;
  if (t <= k) {
    double v = h[t];
    if (h2) v += h2[t];
    sh[t] = v;
    if (t < k) { sc[t] = cs[t]; ss_[t] = sn[t]; }
  }
  __syncthreads();
  if (t != 0) return;
  double ssq = *d_ss;
  if (ssq < 0.0) ssq = 0.0;
  const double hn = sqrt(ssq);
  double hk = sh[0];
  // rotations i < k act on (h[i], h[i+1])
  for (int i = 0; i < k; ++i) {
    const double a = hk, b = sh[i + 1];
    const double tnew = sc[i] * a + ss_[i] * b;
    hk = -ss_[i] * a + sc[i] * b;
    R[(size_t)i * m + k] = tnew;
  }
  const double d = hypot(hk, hn);
  double c, s;
  if (d == 0.0) { c = 1.0; s = 0.0; } else { c = hk / d; s = hn / d; }
  cs[k] = c;
  sn[k] = s;
  R[(size_t)k * m + k] = d;
  const double gk = g[k];
  g[k + 1] = -s * gk;
  g[k] = c * gk;
  const double rn = fabs(g[k + 1]);
  ctl->k = k + 1;
  ctl->rnorm = rn;
  ctl->hn = hn;
  ctl->inv_hn = (hn > 0.0) ? 1.0 / hn : 0.0;
  if (!(rn == rn) || isinf(rn) || !(hn == hn)) { ctl->failed = 1; ctl->done = 1; }
  else if (ctl->tol >= 0.0 && rn <= ctl->tol) { ctl->converged = 1; ctl->done = 1; }
  else if (hn == 0.0) { ctl->converged = 1; ctl->done = 1; }  // happy breakdown
}

// y = R(0:k,0:k)^{-1} g(0:k)   (k = ctl->k)
__global__ __launch_bounds__(64) void k_backsolve(const nk_gmres_ctl *ctl, const double *R, const double *g, double *y,
                                                  int m) {
  if (threadIdx.x != 0) return;
  const int k = ctl->k;
  if (ctl->failed) {
    for (int i = 0; i < m; ++i) y[i] = 0.0;
    return;
  }
  for (int i = k - 1; i >= 0; --i) {
    double s = g[i];
    for (int j = i + 1; j < k; ++j) s -= R[(size_t)i * m + j] * y[j];
    y[i] = s / R[(size_t)i * m + i];
  }
  for (int i = k; i < m; ++i) y[i] = 0.0;
}

// ----------------------------------------------------------------------------- create / destroy / operators
extern "C" int nk_gmres_create(nk_ctx *ctx, int64_t n_local, int restart_m, int ortho, nk_gmres **out) {
  NK_REQUIRE(ctx && out, "NULL argument");
  NK_REQUIRE(n_local >= 0, "negative size");
  if (restart_m <= 0) restart_m = 30;
  NK_REQUIRE(restart_m < NK_MAX_NV, "restart m=%d too large (max %d)", restart_m, NK_MAX_NV - 1);
  NK_REQUIRE(ortho == NK_ORTHO_MGS || ortho == NK_ORTHO_CGS2 || ortho == NK_ORTHO_CGS, "bad ortho %d", ortho);
  NK_HIP(hipSetDevice(ctx->device));
  nk_gmres *G = new nk_gmres();
  G->ctx = ctx;
  G->n = n_local;
  G->m = restart_m;
  G->ortho = ortho;
  G->ldv = (n_local + 31) & ~(int64_t)31;  // 256-byte aligned columns
  if (G->ldv == 0) G->ldv = 32;
  const int m = restart_m;
  NK_TRY(nk_dev_alloc(&G->V, (size_t)G->ldv * (m + 1)));
  NK_TRY(nk_dev_alloc(&G->w, (size_t)G->ldv));
  NK_TRY(nk_dev_alloc(&G->r, (size_t)G->ldv));
  NK_TRY(nk_dev_alloc(&G->d_h, (size_t)NK_MAX_NV + 2));
  NK_TRY(nk_dev_alloc(&G->d_h2, (size_t)NK_MAX_NV + 2));
  NK_TRY(nk_dev_alloc(&G->d_R, (size_t)m * m));
  NK_TRY(nk_dev_alloc(&G->d_cs, (size_t)m + 1));
  NK_TRY(nk_dev_alloc(&G->d_sn, (size_t)m + 1));
  NK_TRY(nk_dev_alloc(&G->d_g, (size_t)m + 2));
  NK_TRY(nk_dev_alloc(&G->d_y, (size_t)m + 1));
  NK_TRY(nk_dev_alloc(&G->d_ss, (size_t)4));
  NK_TRY(nk_dev_alloc(&G->d_ctl, (size_t)1));
  NK_HIP(hipHostMalloc((void **)&G->h_ctl, sizeof(nk_gmres_ctl), hipHostMallocDefault));
  NK_HIP(hipMemset(G->d_ctl, 0, sizeof(nk_gmres_ctl)));
  NK_HIP(hipMemset(G->V, 0, (size_t)G->ldv * (m + 1) * sizeof(double)));
  *out = G;
  return NK_OK;
}
extern "C" int nk_gmres_destroy(nk_gmres *G) {
  if (!G) return NK_OK;
  hipFree(G->V); hipFree(G->w); hipFree(G->z); hipFree(G->r);
  hipFree(G->d_h); hipFree(G->d_h2); hipFree(G->d_R); hipFree(G->d_cs); hipFree(G->d_sn);
  hipFree(G->d_g); hipFree(G->d_y); hipFree(G->d_ss); hipFree(G->d_ctl);
  hipFree(G->d_u_own); hipFree(G->d_b); hipFree(G->d_x);
  hipHostFree(G->h_ctl);
  delete G;
  return NK_OK;
}
extern "C" int nk_gmres_set_operator_csr(nk_gmres *G, nk_csr *A) {
  NK_REQUIRE(G && A, "NULL argument");
  NK_REQUIRE(A->nrows == G->n, "operator size %lld != GMRES size %lld", (long long)A->nrows, (long long)G->n);
  G->op_kind = 1;
  G->A = A;
  return NK_OK;
}
extern "C" int nk_gmres_set_operator_jvp(nk_gmres *G, nk_problem *P, const double *u, int memspace) {
  NK_REQUIRE(G && P && u, "NULL argument");
  NK_REQUIRE(P->n_local == G->n, "problem size %lld != GMRES size %lld", (long long)P->n_local, (long long)G->n);
  NK_HIP(hipSetDevice(G->ctx->device));
  G->op_kind = 2;
  G->P = P;
  if (memspace == NK_DEVICE) {
    G->d_u = u;
  } else {
    if (!G->d_u_own) NK_TRY(nk_dev_alloc(&G->d_u_own, (size_t)G->n + 1));
    NK_HIP(hipMemcpyAsync(G->d_u_own, u, G->n * sizeof(double), hipMemcpyHostToDevice, G->ctx->stream));
    G->d_u = G->d_u_own;
  }
  return nk_problem_jvp_prepare(P, G->d_u);
}
extern "C" int nk_gmres_set_operator_fn(nk_gmres *G, nk_matvec_fn fn, void *user) {
  NK_REQUIRE(G && fn, "NULL argument");
  G->op_kind = 3;
  G->fn = fn;
  G->fn_user = user;
  return NK_OK;
}
extern "C" int nk_gmres_set_right_preconditioner(nk_gmres *G, nk_matvec_fn fn, void *user) {
  NK_REQUIRE(G, "NULL argument");
  G->prec = fn;
  G->prec_user = user;
  if (fn && !G->z) NK_TRY(nk_dev_alloc(&G->z, (size_t)G->ldv));
  return NK_OK;
}

// y = A x  (right-preconditioned: y = A M⁻¹ x)
static int op_apply(nk_gmres *G, const double *d_x, double *d_y, const int *d_skip) {
  nk_ctx *ctx = G->ctx;
  const double *src = d_x;
  if (G->prec) {
    if (G->prec(G->prec_user, d_x, G->z, (void *)ctx->stream) != 0) NK_FAIL(NK_E_CALLBACK, "preconditioner failed");
    src = G->z;
  }
  switch (G->op_kind) {
    case 1: return nk_csr_spmv_dev(G->A, src, d_y, d_skip);
    case 2: return nk_problem_jvp_dev(G->P, G->d_u, src, d_y, d_skip);
    case 3:
      ctx->stats.op_applies++;
      if (G->fn(G->fn_user, src, d_y, (void *)ctx->stream) != 0) NK_FAIL(NK_E_CALLBACK, "operator callback failed");
      return NK_OK;
    default: NK_FAIL(NK_E_INVALID, "GMRES has no operator");
  }
}

// ----------------------------------------------------------------------------- one Arnoldi step (enqueue only)
static int arnoldi_step(nk_gmres *G, int k) {
  nk_ctx *ctx = G->ctx;
  const int64_t n = G->n, ldv = G->ldv;
  const int *skip = &G->d_ctl->done;
  double *vk = G->V + (size_t)k * ldv;
  NK_TRY(op_apply(G, vk, G->w, skip));
  if (G->ortho == NK_ORTHO_MGS) {
    for (int i = 0; i <= k; ++i) {
      // h_i = v_i·w ; w -= h_i v_i   (the last axpy also yields ‖w‖²)
      NK_TRY(nk_blas_multidot(ctx, n, 1, G->V + (size_t)i * ldv, ldv, G->w, G->d_h + i, false, skip));
      NK_TRY(nk_blas_multiaxpy(ctx, n, 1, G->V + (size_t)i * ldv, ldv, G->d_h + i, -1.0, G->w,
                               i == k ? G->d_ss : nullptr, skip, nullptr));
    }
    hipLaunchKernelGGL(k_givens, dim3(1), dim3(64), 0, ctx->stream, G->d_ctl, G->d_h, (const double *)nullptr, G->d_ss,
                       G->d_R, G->d_cs, G->d_sn, G->d_g, G->m);
  } else if (G->ortho == NK_ORTHO_CGS2) {
    NK_TRY(nk_blas_multidot(ctx, n, k + 1, G->V, ldv, G->w, G->d_h, false, skip));
    NK_TRY(nk_blas_multiaxpy(ctx, n, k + 1, G->V, ldv, G->d_h, -1.0, G->w, nullptr, skip, nullptr));
    NK_TRY(nk_blas_multidot(ctx, n, k + 1, G->V, ldv, G->w, G->d_h2, false, skip));
    NK_TRY(nk_blas_multiaxpy(ctx, n, k + 1, G->V, ldv, G->d_h2, -1.0, G->w, G->d_ss, skip, nullptr));
    hipLaunchKernelGGL(k_givens, dim3(1), dim3(64), 0, ctx->stream, G->d_ctl, G->d_h, (const double *)G->d_h2, G->d_ss,
                       G->d_R, G->d_cs, G->d_sn, G->d_g, G->m);
  } else {  // CGS + DGKS
    const int *skip2 = &G->d_ctl->pad0;
    NK_TRY(nk_blas_multidot(ctx, n, k + 1, G->V, ldv, G->w, G->d_h, true, skip));  // h[k+1] = ‖w‖²
    NK_TRY(nk_blas_multiaxpy(ctx, n, k + 1, G->V, ldv, G->d_h, -1.0, G->w, G->d_ss + 1, skip, nullptr));
    // no re-orthogonalisation ⇒ k_dgks zeroes h2 and parks ss1/nranks in d_ss[0] (the unconditional
    // all-reduce inside the skipped pass-2 axpy restores ss1); otherwise pass 2 overwrites d_ss[0].
    hipLaunchKernelGGL(k_dgks, dim3(1), dim3(64), 0, ctx->stream, G->d_ctl, G->d_h, G->d_ss + 1, G->d_h2, G->d_ss,
                       1.0 / (double)ctx->nranks);
    NK_TRY(nk_blas_multidot(ctx, n, k + 1, G->V, ldv, G->w, G->d_h2, false, skip2));
    NK_TRY(nk_blas_multiaxpy(ctx, n, k + 1, G->V, ldv, G->d_h2, -1.0, G->w, G->d_ss, skip2, nullptr));
    hipLaunchKernelGGL(k_givens, dim3(1), dim3(64), 0, ctx->stream, G->d_ctl, G->d_h, (const double *)G->d_h2, G->d_ss,
                       G->d_R, G->d_cs, G->d_sn, G->d_g, G->m);
  }
  NK_HIP(hipGetLastError());
  // v_{k+1} = w / h_{k+1,k}
  return nk_blas_scale_to(ctx, n, &G->d_ctl->inv_hn, G->w, G->V + (size_t)(k + 1) * ldv, skip);
}

int nk_gmres_solve_dev(nk_gmres *G, const double *d_b, double *d_x, int use_x0, double atol, double rtol,
                       int maxiter, int fixed_iters, nk_gmres_info *info) {
  nk_ctx *ctx = G->ctx;
  const int64_t n = G->n, ldv = G->ldv;
  const int m = G->m;
  NK_REQUIRE(G->op_kind != 0, "GMRES has no operator");
  if (maxiter <= 0) maxiter = 300;
  const int cap = fixed_iters > 0 ? fixed_iters : maxiter;
  nk_gmres_info inf;
  memset(&inf, 0, sizeof(inf));
  ctx->stats.nsolve += 0;  // nsolve is counted by the nonlinear driver (LinearSolveJLCache functor)
  // r0 = b − A x0
  const double *rsrc = d_b;
  if (!use_x0) {
    NK_TRY(nk_blas_fill(ctx, n, 0.0, d_x));
  } else {
    if (G->prec) NK_FAIL(NK_E_UNSUPPORTED, "use_x0 with a right preconditioner is not supported");
    NK_TRY(op_apply(G, d_x, G->w, nullptr));
    NK_TRY(nk_blas_lincomb(ctx, n, 1.0, d_b, -1.0, G->w, G->r));
    rsrc = G->r;
  }
  int first = 1;
  int total_iters = 0;
  for (;;) {
    NK_TRY(nk_blas_sumsq(ctx, n, rsrc, G->d_ss));
    hipLaunchKernelGGL(k_gmres_begin, dim3(1), dim3(64), 0, ctx->stream, G->d_ctl, G->d_ss, atol, rtol,
                       fixed_iters > 0 ? 1 : 0, first, G->d_g, m);
    first = 0;
    NK_TRY(nk_blas_scale_to(ctx, n, &G->d_ctl->inv_hn, rsrc, G->V, &G->d_ctl->done));
    const int steps = (cap - total_iters) < m ? (cap - total_iters) : m;
    for (int k = 0; k < steps; ++k) NK_TRY(arnoldi_step(G, k));
    // x += M⁻¹ V y
    hipLaunchKernelGGL(k_backsolve, dim3(1), dim3(64), 0, ctx->stream, G->d_ctl, G->d_R, G->d_g, G->d_y, m);
    if (!G->prec) {
      NK_TRY(nk_blas_multiaxpy(ctx, n, m, G->V, ldv, G->d_y, 1.0, d_x, nullptr, nullptr, &G->d_ctl->k));
    } else {
      NK_TRY(nk_blas_fill(ctx, n, 0.0, G->w));
      NK_TRY(nk_blas_multiaxpy(ctx, n, m, G->V, ldv, G->d_y, 1.0, G->w, nullptr, nullptr, &G->d_ctl->k));
      if (G->prec(G->prec_user, G->w, G->z, (void *)ctx->stream) != 0) NK_FAIL(NK_E_CALLBACK, "preconditioner failed");
      NK_TRY(nk_blas_axpby(ctx, n, 1.0, G->z, 1.0, d_x));
    }
    NK_HIP(hipMemcpyAsync(G->h_ctl, G->d_ctl, sizeof(nk_gmres_ctl), hipMemcpyDeviceToHost, ctx->stream));
    NK_HIP(hipStreamSynchronize(ctx->stream));
    const nk_gmres_ctl c = *G->h_ctl;
    total_iters += c.k;
    inf.rnorm0 = c.rnorm0;
    inf.rnorm = c.rnorm;
    inf.converged = c.converged;
    inf.failed = c.failed;
    if (c.failed || c.converged || total_iters >= cap || steps == 0) break;
    inf.restarts++;
    // restart: r = b − A x
    if (G->prec) {
      // A x directly (x is in the original space): bypass the preconditioner
      nk_matvec_fn p = G->prec;
      G->prec = nullptr;
      int st = op_apply(G, d_x, G->w, nullptr);
      G->prec = p;
      NK_TRY(st);
    } else {
      NK_TRY(op_apply(G, d_x, G->w, nullptr));
    }
    NK_TRY(nk_blas_lincomb(ctx, n, 1.0, d_b, -1.0, G->w, G->r));
    rsrc = G->r;
  }
  inf.iters = total_iters;
  ctx->stats.gmres_iters += total_iters;
  if (info) *info = inf;
  return NK_OK;
}

extern "C" int nk_gmres_solve(nk_gmres *G, const double *b, double *x, int memspace, int use_x0, double atol,
                              double rtol, int maxiter, int fixed_iters, nk_gmres_info *info) {
  NK_REQUIRE(G && b && x, "NULL argument");
  NK_HIP(hipSetDevice(G->ctx->device));
  if (memspace == NK_DEVICE) return nk_gmres_solve_dev(G, b, x, use_x0, atol, rtol, maxiter, fixed_iters, info);
  nk_ctx *ctx = G->ctx;
  if (!G->d_b) NK_TRY(nk_dev_alloc(&G->d_b, (size_t)G->n + 1));
  if (!G->d_x) NK_TRY(nk_dev_alloc(&G->d_x, (size_t)G->n + 1));
  NK_HIP(hipMemcpyAsync(G->d_b, b, G->n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  if (use_x0) NK_HIP(hipMemcpyAsync(G->d_x, x, G->n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  NK_TRY(nk_gmres_solve_dev(G, G->d_b, G->d_x, use_x0, atol, rtol, maxiter, fixed_iters, info));
  NK_HIP(hipMemcpyAsync(x, G->d_x, G->n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  NK_HIP(hipStreamSynchronize(ctx->stream));
  return NK_OK;
}
