/* Seam 1 replayed in plain C99: the call sequence NonlinearSolve.jl's own Newton loop performs against a `linsolve`
 * backend — lib/NonlinearSolveBase/ext/NonlinearSolveBaseLinearSolveExt.jl:16-32 (the LinearSolveJLCache functor:
 * nsolve += 1, update_A!, b, linu, solve!), :102-111 (set_lincache_A!: a NEW operator object every step for the
 * matrix-free path, lib/NonlinearSolveBase/src/jacobian.jl:260-262), :113-115 (update_tolerances!, pushed by
 * EisenstatWalkerForcing2 before every solve, lib/NonlinearSolveFirstOrder/src/eisenstat_walker.jl:50,77) — with the
 * Newton loop of lib/NonlinearSolveFirstOrder/src/solve.jl:325-465 written out on the host. This is what
 * julia/src/MI355XNewtonKrylov.jl's MI355XGMRES does through `ccall`, minus Julia.
 *
 * Three ways of handing `A` over, all with the vectors RESIDENT on the device (memspace = NK_DEVICE, buffers from
 * nk_device_alloc — a host language without a GPU array type needs nothing else):
 *   fn   : a callback operator that receives DEVICE pointers (here it applies the problem's JVP at the u it was built
 *          with — the shape of a StatefulJacobianOperator), rebuilt and re-registered every step
 *   jvp  : nk_gmres_set_operator_jvp — what the binding does when it recognises the device problem behind f.jvp
 *   csr  : a concrete sparse J, values refilled every step (f.jac), nk_gmres_set_operator_csr
 *   precs: csr + what the reference's documented `precs` return — `incompletelu(W, p) = (ilu(W), I)`,
 *          docs/src/tutorials/large_systems.md:257-260 — a LEFT preconditioner Pl, re-evaluated for every new A
 *          (lib/NonlinearSolveBase/src/linear_solve.jl:195-199; test/Core/core_tests__item21.jl:10-18): a device ILU(0) object of
 *          the concrete J (refactorised on the device for every new Jacobian, nk_precond_update), GMRES on Pl⁻¹ A, stopping on
 *          the preconditioned residual
 *
 *   gcc -std=c99 -Iinclude examples/linsolve_seam.c -Lnonlinearsolve.jl_amd/lib -lmi355x_nk -lm -o linsolve_seam
 *   LD_LIBRARY_PATH=nonlinearsolve.jl_amd/lib ./linsolve_seam [n_side] [out_prefix]
 * Prints one line per variant; with out_prefix also writes <prefix>_<variant>.bin (the final u, n doubles). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mi355x_nk.h"

#define CHECK(call)                                                                 \
  do {                                                                              \
    int st_ = (call);                                                               \
    if (st_ != NK_OK) {                                                             \
      fprintf(stderr, "%s failed (%d): %s\n", #call, st_, nk_last_error());        \
      return 1;                                                                     \
    }                                                                               \
  } while (0)

/* StatefulJacobianOperator(J, u, p): the operator object the nonlinear solver builds for every step */
typedef struct {
  nk_problem *P;
  const double *u_dev; /* the linearisation point, a device pointer */
  long applies;
} stateful_op;

/* mul!(Jv, A, v) — x and y are DEVICE pointers (nk_matvec_fn contract), so the JVP runs with memspace NK_DEVICE */
static int op_mul(void *user, const double *x, double *y, void *stream) {
  stateful_op *A = (stateful_op *)user;
  (void)stream; /* the problem shares the context's stream */
  A->applies++;
  return nk_jvp(A->P, A->u_dev, x, y, NK_DEVICE) == NK_OK ? 0 : 1;
}

enum { V_FN = 0, V_JVP = 1, V_CSR = 2, V_PRECS = 3 };
static const char *vname[] = {"fn", "jvp", "csr", "precs"};

static int newton(nk_ctx *ctx, nk_problem *P, int variant, int64_t n, const char *prefix) {
  const double abstol = 1e-8;
  const int maxiters = 50, restart = 30, inner_cap = 300;
  /* EisenstatWalkerForcing2() defaults (eisenstat_walker.jl:18-29) */
  const double eta0 = 0.5, eta_max = 0.9, gamma = 0.9, alpha = 2.0, sg_thr = 0.1;
  double *u = NULL, *fu = NULL, *du = NULL;
  CHECK(nk_device_alloc(ctx, n * 8, (void **)&u));
  CHECK(nk_device_alloc(ctx, n * 8, (void **)&fu));
  CHECK(nk_device_alloc(ctx, n * 8, (void **)&du));
  double *host = (double *)calloc((size_t)n, sizeof(double));
  if (!host) return 1;
  CHECK(nk_device_copy(ctx, u, host, n * 8, 0)); /* u0 = zeros */

  nk_gmres *G = NULL; /* the LinearCache: created once at init (construct_linear_solver, linear_solve.jl:116) */
  CHECK(nk_gmres_create(ctx, n, restart, NK_ORTHO_DCGS2, &G));
  nk_csr *J = NULL;
  nk_precond *Pl = NULL; /* what precs(A, p) returned last time (the object is reused: same pattern, new values) */
  if (variant == V_CSR || variant == V_PRECS) CHECK(nk_problem_jac_csr(P, &J));
  stateful_op A = {P, u, 0};

  CHECK(nk_residual(P, u, fu, NK_DEVICE)); /* init: one residual */
  double fnorm = 0.0, eta = eta0, rn = 0.0, rn_prev = 0.0;
  CHECK(nk_norm_inf(ctx, n, fu, &fnorm));
  long nsolve = 0, gmres_iters = 0;
  int step = 0, failed = 0;
  for (; step < maxiters && fnorm > abstol; ++step) {
    /* --- a new Jacobian (operator) for this step: jac_cache(u) */
    if (variant == V_FN) {
      A.u_dev = u;                                           /* StatefulJacobianOperator(J, u, p) */
      CHECK(nk_gmres_set_operator_fn(G, op_mul, &A));        /* lincache.A = new_A  (marks the cache fresh) */
    } else if (variant == V_JVP) {
      CHECK(nk_gmres_set_operator_jvp(G, P, u, NK_DEVICE));
    } else {
      CHECK(nk_jac_values(P, u, NK_DEVICE, J));              /* f.jac(J, u, p) */
      CHECK(nk_gmres_set_operator_csr(G, J));
      if (variant == V_PRECS) {                              /* (Pl, Pr) = precs(A, LinearSolveParameters(u, p)) for the fresh A */
        if (!Pl) CHECK(nk_precond_create_ilu0(J, NK_ILU_MULTICOLOR, &Pl));
        else CHECK(nk_precond_update(Pl));
        CHECK(nk_gmres_set_preconditioner(G, NK_SIDE_LEFT, Pl));   /* Pl = ILU(0) */
        CHECK(nk_gmres_set_preconditioner(G, NK_SIDE_RIGHT, NULL)); /* Pr = I     */
      }
    }
    /* --- pre_step_forcing!: η → update_tolerances!(lincache; reltol = η) */
    if (step == 0) {
      CHECK(nk_nrm2(ctx, n, fu, &rn));
      rn_prev = rn;
      eta = eta0;
    } else {
      const double eta_prev = eta;
      eta = gamma * pow(rn / rn_prev, alpha);
      const double eta_sg = gamma * pow(eta_prev, alpha);
      if (eta_sg > sg_thr && eta_sg > eta) eta = eta_sg;
      if (eta < 0.0) eta = 0.0;
      if (eta > eta_max) eta = eta_max;
    }
    /* --- the functor: nsolve += 1; b = fu; linu = δu; solve! */
    nsolve++;
    nk_gmres_info info;
    CHECK(nk_gmres_solve(G, fu, du, NK_DEVICE, /*use_x0*/ 0, /*atol*/ 0.0, /*rtol*/ eta, inner_cap, 0, &info));
    gmres_iters += info.iters;
    if (info.failed) { failed = 1; break; }                  /* ReturnCode.Failure ⇒ success = false */
    /* --- post_step_forcing!: ‖fu‖ before u moves (the one-step lag) */
    rn_prev = rn;
    CHECK(nk_nrm2(ctx, n, fu, &rn));
    /* --- δu ← −δu ; u += δu ; fu = f(u) ; termination on ‖fu‖∞ */
    CHECK(nk_axpy(ctx, n, -1.0, du, u));
    CHECK(nk_residual(P, u, fu, NK_DEVICE));
    CHECK(nk_norm_inf(ctx, n, fu, &fnorm));
  }
  CHECK(nk_device_copy(ctx, host, u, n * 8, 1));
  double umax = 0.0;
  for (int64_t i = 0; i < n; ++i)
    if (host[i] > umax) umax = host[i];
  printf("%s: steps=%d nsolve=%ld gmres_iters=%ld failed=%d fnorm_inf=%.3e max_u=%.9f callback_applies=%ld\n", vname[variant],
         step, nsolve, gmres_iters, failed, fnorm, umax, A.applies);
  if (prefix) {
    char path[512];
    snprintf(path, sizeof(path), "%s_%s.bin", prefix, vname[variant]);
    FILE *fp = fopen(path, "wb");
    if (!fp || fwrite(host, sizeof(double), (size_t)n, fp) != (size_t)n) return 1;
    fclose(fp);
  }
  free(host);
  CHECK(nk_gmres_destroy(G));
  if (Pl) CHECK(nk_precond_destroy(Pl));
  if (J) CHECK(nk_csr_destroy(J));
  CHECK(nk_device_free(ctx, u));
  CHECK(nk_device_free(ctx, fu));
  CHECK(nk_device_free(ctx, du));
  return (failed || fnorm > abstol) ? 2 : 0;
}

int main(int argc, char **argv) {
  const int ns = argc > 1 ? atoi(argv[1]) : 64;
  const char *prefix = argc > 2 ? argv[2] : NULL;
  const int64_t n = (int64_t)ns * ns;
  nk_ctx *ctx = NULL;
  nk_problem *P = NULL;
  CHECK(nk_ctx_create(0, NULL, &ctx));
  const double params[2] = {(double)ns, 6.0};
  CHECK(nk_problem_create(ctx, NK_PROBLEM_BRATU2D, params, 2, &P));
  int rc = 0;
  for (int v = V_FN; v <= V_PRECS; ++v) rc |= newton(ctx, P, v, n, prefix);
  CHECK(nk_problem_destroy(P));
  CHECK(nk_ctx_destroy(ctx));
  return rc;
}
