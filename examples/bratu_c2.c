/* Plain C against the C ABI (no Python, no C++): config C2 of BASELINE.json — 2-D Bratu 256², NewtonRaphson with the
 * concrete sparse Jacobian and the direct (banded LU) linsolve — followed by the same problem through the matrix-free
 * Newton–Krylov path. This is what a `ccall` binding does (julia/src/MI355XNewtonKrylov.jl), written out in C.
 *
 *   gcc -std=c99 -Iinclude examples/bratu_c2.c -Lnonlinearsolve.jl_amd/lib -lmi355x_nk -lm -o bratu_c2
 *   LD_LIBRARY_PATH=nonlinearsolve.jl_amd/lib ./bratu_c2 [n_side]
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "mi355x_nk.h"

#define CHECK(call)                                                                 \
  do {                                                                              \
    int st_ = (call);                                                               \
    if (st_ != NK_OK) {                                                             \
      fprintf(stderr, "%s failed (%d): %s\n", #call, st_, nk_last_error());        \
      return 1;                                                                     \
    }                                                                               \
  } while (0)

static int run(nk_problem *P, const nk_options *o, size_t n, const char *label, double *u, double *f) {
  nk_stats st;
  int retcode = 0;
  for (size_t i = 0; i < n; ++i) u[i] = 0.0; /* u0 = zeros */
  CHECK(nk_newton_solve(P, u, NK_HOST, o, u, f, &st, &retcode));
  double fmax = 0.0, umax = 0.0;
  for (size_t i = 0; i < n; ++i) {
    if (fabs(f[i]) > fmax) fmax = fabs(f[i]);
    if (u[i] > umax) umax = u[i];
  }
  printf("%s: retcode=%d nsteps=%lld nf=%lld njacs=%lld nfactors=%lld gmres_iters=%lld |h^2 F|_inf=%.3e max u=%.6f\n", label,
         retcode, (long long)st.nsteps, (long long)st.nf, (long long)st.njacs, (long long)st.nfactors,
         (long long)st.gmres_iters, fmax, umax);
  return retcode == NK_RET_SUCCESS ? 0 : 2;
}

int main(int argc, char **argv) {
  const int ns = argc > 1 ? atoi(argv[1]) : 256;
  const size_t n = (size_t)ns * (size_t)ns;
  nk_ctx *ctx = NULL;
  nk_problem *P = NULL;
  CHECK(nk_ctx_create(0, NULL, &ctx));
  const double params[2] = {(double)ns, 6.0}; /* {n_side, lambda}; residual scaled by h^2 */
  CHECK(nk_problem_create(ctx, NK_PROBLEM_BRATU2D, params, 2, &P));
  double *u = (double *)malloc(n * sizeof(double)), *f = (double *)malloc(n * sizeof(double));
  if (!u || !f) return 1;

  nk_options o;
  CHECK(nk_options_default(&o));
  o.abstol = 1e-8;
  o.maxiters = 50;
  o.algorithm = NK_ALG_NEWTON_RAPHSON;
  o.linsolve = NK_LINSOLVE_BANDED_LU; /* NewtonRaphson(): concrete sparse J + direct solve */
  int rc = run(P, &o, n, "direct ", u, f);

  o.linsolve = NK_LINSOLVE_GMRES_MATFREE; /* NewtonRaphson(linsolve = KrylovJL_GMRES()) on the JacobianOperator */
  o.forcing = NK_FORCING_EISENSTAT_WALKER2;
  o.gmres_restart = 30;
  o.gmres_maxiters = 2000;
  o.mg_nu = 2; /* precs = built-in multigrid V-cycle */
  rc |= run(P, &o, n, "krylov ", u, f);

  free(u);
  free(f);
  CHECK(nk_problem_destroy(P));
  CHECK(nk_ctx_destroy(ctx));
  return rc;
}
