/* mi355x_nk.h — C ABI of libmi355x_nk.so: a MI355X (gfx950) Newton–Krylov inner loop that
 * drops in behind SciML/NonlinearSolve.jl's first-order step path.
 *
 * Every entry point replaces one seam of the reference (paths relative to the reference tree):
 *
 *   seam 1  linsolve backend   lib/NonlinearSolveBase/ext/NonlinearSolveBaseLinearSolveExt.jl:16-32
 *                              (LinearSolveJLCache functor → solve!(lincache)), tolerances pushed by
 *                              lib/NonlinearSolveFirstOrder/src/eisenstat_walker.jl:50,77
 *                              → nk_gmres_*                                     (GMRES(m) on device)
 *   seam 2  jac/jvp/vjp        lib/SciMLJacobianOperators/src/SciMLJacobianOperators.jl:167-182,238-243
 *                              (mul!(Jv, StatefulJacobianOperator, v)), f.jac at
 *                              lib/NonlinearSolveBase/src/jacobian.jl:237-258
 *                              → nk_residual / nk_jvp / nk_vjp / nk_jac_values / nk_spmv / nk_spmv_t
 *   seam 3  whole solver       ext/NonlinearSolvePETScExt.jl:38-167 pattern (SciMLBase.__solve) over
 *                              lib/NonlinearSolveFirstOrder/src/solve.jl:140-301 (init), :325-465 (step!)
 *                              lib/NonlinearSolveBase/src/solve.jl:360-387 (run to completion)
 *                              → nk_solver_init / _step / _solve / _reinit
 *
 * Conventions: C linkage, plain pointers and sizes, no C++ or torch types. All floating point data is
 * IEEE binary64. Every function returns an nk_status (0 = ok, <0 = error); the message of the last error
 * is available from nk_last_error(). Numerical non-convergence is a *retcode*, never an error status.
 * Vector arguments carry an explicit memory space (NK_HOST / NK_DEVICE). Host arrays are only read or
 * written during the call; device arrays must live on the context's device. Calls are asynchronous on the
 * context's HIP stream unless documented otherwise (anything returning scalars to the host synchronises).
 * A context is single-threaded. There is NO CPU fallback: without a HIP device nk_ctx_create fails.
 */
#ifndef MI355X_NK_H
#define MI355X_NK_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NK_VERSION_MAJOR 0
#define NK_VERSION_MINOR 1

/* ---------------------------------------------------------------- status / enums */
typedef enum {
  NK_OK = 0,
  NK_E_INVALID = -1,   /* bad argument */
  NK_E_HIP = -2,       /* HIP runtime error (no device, launch failure, ...) */
  NK_E_RCCL = -3,      /* RCCL error or librccl not loadable */
  NK_E_NOMEM = -4,
  NK_E_UNSUPPORTED = -5,
  NK_E_CALLBACK = -6,  /* a user callback returned non-zero */
  NK_E_SINGULAR = -7,  /* a preconditioner could not be built: zero / non-finite diagonal entry or ILU(0) pivot */
  NK_E_COMM = -8       /* a peer-mapped collective timed out (a rank stalled or died): results of this call are not valid */
} nk_status;

typedef enum { NK_HOST = 0, NK_DEVICE = 1 } nk_memspace;

/* SciMLBase.ReturnCode values that the first-order path can produce
 * (lib/NonlinearSolveFirstOrder/src/solve.jl:370,399,432; lib/NonlinearSolveBase/src/solve.jl:372-376,851;
 *  lib/NonlinearSolveBase/src/termination_conditions.jl:262,270,283,303,332). */
typedef enum {
  NK_RET_DEFAULT = 0,
  NK_RET_SUCCESS = 1,
  NK_RET_MAXITERS = 2,
  NK_RET_UNSTABLE = 3,
  NK_RET_STALLED = 4,
  NK_RET_INTERNAL_LINEAR_SOLVE_FAILED = 5,
  NK_RET_SHRINK_THRESHOLD_EXCEEDED = 6,
  NK_RET_MAXTIME = 7,
  NK_RET_FAILURE = 8,
  NK_RET_INTERNAL_LINESEARCH_FAILED = 9
} nk_retcode;

typedef enum {
  NK_PROBLEM_QUADRATIC = 1,     /* f(u,p) = u.*u .- p          common/common_rootfind_testing.jl:15-17 */
  NK_PROBLEM_BRATU2D = 2,       /* 5-point Bratu (not in the reference)                  */
  NK_PROBLEM_BRUSSELATOR2D = 3, /* lib/NonlinearSolveFirstOrder/test/sparsity_tests__item1.jl:13-36     */
  NK_PROBLEM_USER = 100         /* user callbacks (f, jvp, vjp, jac) — NonlinearFunction fields         */
} nk_problem_kind;

/* NK_ALG_GAUSS_NEWTON: NewtonDescent in NORMAL FORM, JᵀJ δ = Jᵀ f through the normal-form operator — what GaussNewton
 * (lib/NonlinearSolveFirstOrder/src/gauss_newton.jl:11-23) does on a NonlinearLeastSquaresProblem with a Krylov linsolve
 * (descent/newton.jl:58-95,107-118; StatefulJacobianNormalFormOperator, SciMLJacobianOperators.jl:252-291). */
/* NK_ALG_LEVENBERG_MARQUARDT: the damped normal equations (JᵀJ + λDᵀD) δ = Jᵀ f through the same normal-form operator
 * plus a diagonal (DampedNewtonDescent in :normal_form mode, descent/damped_newton.jl:186-200,297-313 — the mode a Krylov
 * `linsolve` selects), geodesic acceleration (descent/geodesic_acceleration.jl:98-136) and the damping-based trust region
 * (levenberg_marquardt.jl:159-168,247-268). Needs a concrete J (concrete_jac = Val(true)) and a Krylov linsolve. */
/* NK_ALG_PSEUDO_TRANSIENT: DampedNewtonDescent with SwitchedEvolutionRelaxation damping in :simple mode
 * (descent/damped_newton.jl:283-288): the Newton step on J + α⁻¹ I — the shift is added to the diagonal of the concrete J
 * (`dampen_jacobian!!`) or to the matrix-free operator. */
typedef enum { NK_ALG_NEWTON_RAPHSON = 0, NK_ALG_TRUST_REGION = 1, NK_ALG_GAUSS_NEWTON = 2,
               NK_ALG_LEVENBERG_MARQUARDT = 3, NK_ALG_PSEUDO_TRANSIENT = 4 } nk_algorithm;

/* which operator the Krylov solver sees as A (lib/NonlinearSolveBase/src/jacobian.jl:43-47,90-102) */
typedef enum {
  NK_LINSOLVE_GMRES_MATFREE = 0, /* StatefulJacobianOperator: fused device JVP                       */
  NK_LINSOLVE_GMRES_CSR = 1,     /* concrete sparse J (values refilled every step) + CSR SpMV        */
  NK_LINSOLVE_BANDED_LU = 2      /* concrete J, direct banded LU on device (config C2)                */
} nk_linsolve;

typedef enum {
  NK_ORTHO_MGS = 0,  /* modified Gram–Schmidt: the structure Krylov.jl's gmres uses [EXT]            */
  NK_ORTHO_CGS2 = 1, /* classical GS, always re-orthogonalised (2 fused passes)                       */
  NK_ORTHO_CGS = 2,  /* classical GS, re-orthogonalise only when ‖w'‖ < ‖w‖/√2 (DGKS)                 */
  NK_ORTHO_DCGS2 = 3, /* the default: CGS2 arithmetic with delayed re-orthogonalisation — two sweeps over the basis and
                       * (with the built-in linear operators) ONE reduction / all-reduce per Arnoldi step; operators
                       * reached through callbacks keep two reductions (and plain CGS2 when restart > 31)            */
  NK_ORTHO_DCGS2_1R = 4, /* insist on the one-reduction form (the pending vector's second projection and the new vector's
                       * first projection share a fused dot sweep; the Hessenberg column and the stopping test lag
                       * one step); falls back like NK_ORTHO_DCGS2 where the operator does not allow it            */
  NK_ORTHO_SSTEP = 5    /* s-step (communication-avoiding) Arnoldi: s operator applications build a block of basis vectors
                       * (A − θ_j I)…(A − θ_0 I) v — Newton basis with Leja-ordered Chebyshev shifts when real bounds of the
                       * operator's spectrum are known (Gershgorin discs of a CSR operator, the Bratu stencil's closed form,
                       * nk_gmres_set_spectrum_interval), monomial (θ = 0) otherwise — which is orthogonalised against the
                       * basis and within itself by block CGS in Pythagorean form, twice: three sweeps over the basis per
                       * s columns instead of two per column; the Gram blocks run on the FP64 matrix cores. Block size:
                       * nk_options.gmres_sstep / nk_gmres_set_block_size (0 = automatic: 15 Newton, 6 monomial). A block
                       * that loses rank numerically (Cholesky breakdown) finishes that solve with NK_ORTHO_DCGS2.           */
} nk_ortho;

/* basis of an s-step block (nk_options.gmres_sstep_basis / nk_gmres_set_sstep_basis) */
typedef enum {
  NK_SS_BASIS_AUTO = 0,     /* Newton where spectrum bounds are known, else monomial */
  NK_SS_BASIS_MONOMIAL = 1, /* A^j v (block size ≤ 8 advisable: κ grows like ρ^s)     */
  NK_SS_BASIS_NEWTON = 2    /* insist on the Newton basis (error if no bounds known)  */
} nk_ss_basis;

typedef enum { NK_FORCING_NONE = 0, NK_FORCING_EISENSTAT_WALKER2 = 1 } nk_forcing;

/* RadiusUpdateSchemes — lib/NonlinearSolveFirstOrder/src/trust_region.jl:59-147 */
typedef enum {
  NK_RUS_SIMPLE = 0, NK_RUS_NLSOLVE = 1, NK_RUS_NOCEDAL_WRIGHT = 2, NK_RUS_HEI = 3,
  NK_RUS_YUAN = 4, NK_RUS_BASTIN = 5, NK_RUS_FAN = 6
} nk_radius_update_scheme;

typedef enum { NK_COMM_NONE = 0, NK_COMM_RCCL = 1, NK_COMM_CALLBACKS = 2 } nk_comm_kind;

/* the nine SciMLBase termination modes (common/common_rootfind_testing.jl:3-13;
 * lib/NonlinearSolveBase/src/termination_conditions.jl:243-376). 0 is the reference's default (:385-389). */
typedef enum {
  NK_TM_ABSNORM_SAFEBEST = 0, NK_TM_NORM = 1, NK_TM_REL = 2, NK_TM_RELNORM = 3, NK_TM_RELNORM_SAFE = 4,
  NK_TM_RELNORM_SAFEBEST = 5, NK_TM_ABS = 6, NK_TM_ABSNORM = 7, NK_TM_ABSNORM_SAFE = 8
} nk_termination_mode;

/* ---------------------------------------------------------------- opaque handles */
typedef struct nk_ctx nk_ctx;         /* device, stream, communicator, scratch                     */
typedef struct nk_csr nk_csr;         /* row-partitioned CSR (f64 values, i32 indices) + halo plan */
typedef struct nk_problem nk_problem; /* residual / JVP / VJP / Jacobian provider                  */
typedef struct nk_gmres nk_gmres;     /* GMRES(m) workspace (LinearSolve "LinearCache" analogue)    */
typedef struct nk_solver nk_solver;   /* GeneralizedFirstOrderAlgorithmCache analogue               */
typedef struct nk_bandlu nk_lu;       /* banded LU factorisation (LinearSolve factorisation-cache analogue) */
typedef struct nk_batch nk_batch;     /* ensemble of small dense systems, one per GPU thread */

/* ---------------------------------------------------------------- plain structs */

/* NLStats (nf, njacs, nfactors, nsolve, nsteps) — incremented where the reference increments them
 * (lib/NonlinearSolveBase/src/utils.jl:201, jacobian.jl:238, ext/...LinearSolveExt.jl:20,84,
 *  lib/NonlinearSolveBase/src/solve.jl:844) — extended with device-side work counters. */
typedef struct {
  int64_t nf, njacs, nfactors, nsolve, nsteps;
  int64_t gmres_iters;     /* total Arnoldi steps                        */
  int64_t op_applies;      /* SpMV / JVP / VJP applications              */
  int64_t allreduces;      /* collective calls issued                    */
  int64_t halo_exchanges;
} nk_stats;

typedef struct {
  int32_t iters;        /* Arnoldi steps performed                                    */
  int32_t restarts;     /* completed restart cycles                                   */
  int32_t converged;    /* 1: ‖r‖ ≤ atol + rtol‖r0‖ ; 0: iteration cap hit            */
  int32_t failed;       /* 1: non-finite residual (maps to ReturnCode.Failure)        */
  double  rnorm0;       /* ‖b − A x0‖₂                                                */
  double  rnorm;        /* last (recurrence) residual norm estimate                   */
} nk_gmres_info;

/* One row per nonlinear step: the TraceMinimal fields (lib/NonlinearSolveBase/src/tracing.jl:259-309)
 * plus the Krylov / forcing / trust-region scalars. */
typedef struct {
  int32_t iter;
  int32_t gmres_iters;
  int32_t accepted;      /* trust region: step accepted (always 1 for Newton) */
  int32_t reserved;
  double  fnorm_inf;     /* ‖f(u)‖∞ after the step                            */
  double  step_norm2;    /* ‖δu‖₂                                             */
  double  eta;           /* Krylov reltol used                                */
  double  trust_region;  /* Δ after the update                                */
  double  rho;
} nk_trace_entry;

typedef struct {
  /* --- algorithm (raphson.jl:30-43, trust_region.jl:25-43) */
  int32_t algorithm;            /* nk_algorithm                                                     */
  int32_t linsolve;             /* nk_linsolve                                                      */
  int32_t maxiters;             /* 1000  (FirstOrder/src/solve.jl:142)                              */
  int32_t termination_norm;     /* internalnorm of the termination mode: 0 = maximum∘abs (default), 1 = 2-norm */
  double  abstol;               /* ≤0 → 3.0e-13 (common_defaults.jl:44-48)                          */
  double  reltol;               /* ≤0 → 3.0e-13; only forwarded to the linear solver                */
  double  maxtime;              /* seconds, ≤0 → none (Base/src/solve.jl:846-856)                   */
  /* --- Krylov protocol */
  int32_t gmres_restart;        /* m; ≤0 → 30                                                       */
  int32_t gmres_maxiters;       /* inner-iteration cap per linear solve; ≤0 → 300                   */
  int32_t gmres_ortho;          /* nk_ortho                                                         */
  int32_t gmres_fixed_iters;    /* >0: run exactly this many Arnoldi steps (fixed-work protocol)    */
  double  lin_abstol;           /* <0 → forward the nonlinear abstol (FirstOrder/src/solve.jl:203)  */
  double  lin_reltol;           /* <0 → forward the nonlinear reltol                                */
  /* --- forcing (eisenstat_walker.jl:18-29) */
  int32_t forcing;              /* nk_forcing                                                       */
  int32_t ew_safeguard;         /* 1                                                                */
  double  ew_eta0, ew_eta_max, ew_gamma, ew_alpha, ew_safeguard_threshold; /* .5 .9 .9 2 .1         */
  /* --- trust region (trust_region.jl:25-33,320-384); 0 → the scheme's default */
  int32_t radius_update_scheme; /* nk_radius_update_scheme                                          */
  int32_t max_shrink_times;     /* 32                                                               */
  double  max_trust_radius, initial_trust_radius, step_threshold, shrink_threshold,
          expand_threshold, shrink_factor, expand_factor;
  /* --- termination: AbsNormSafeBestTerminationMode(maximum∘abs; max_stalled_steps = 32)
   *     (termination_conditions.jl:243-336,385-389); numeric defaults of the mode struct are [EXT] */
  int32_t patience_steps;       /* 100                                                              */
  int32_t max_stalled_steps;    /* 32; <0 disables the stall test                                   */
  double  patience_objective_multiplier; /* 3                                                        */
  double  min_max_factor;       /* 1.3                                                              */
  double  protective_threshold; /* ≤0 → off                                                         */
  /* --- tracing */
  int32_t store_trace;          /* keep nk_trace_entry rows (costs one extra 2-norm per step)       */
  int32_t termination_mode;     /* nk_termination_mode; 0 = AbsNormSafeBest, the reference default  */
  /* --- preconditioner of the Krylov solver (`precs`): 0 none, >0 Chebyshev polynomial of that degree, its
   *     interval re-estimated for every new Jacobian (lambda_max by power iteration, lambda_min = max/ratio) */
  int32_t cheb_degree;
  int32_t linesearch;           /* 0 none (missing), 1 BackTracking, 2 LineSearchesJL(Static), 3 LineSearchesJL(StrongWolfe),
                                   4 LineSearchesJL(MoreThuente), 5 LineSearchesJL(HagerZhang) [EXT: LineSearches.jl defaults] — globalisation
                                   Val(:LineSearch), lib/NonlinearSolveFirstOrder/src/solve.jl:392-408               */
  double  cheb_ratio;           /* ≤1 → 30 */
  /* --- BackTracking line search (LineSearch.jl / LineSearches.jl [EXT]: c_1 = 1e-4, ρ_hi = 0.5, ρ_lo = 0.1,
   *     order = 3, iterations = 1000) on ϕ(α) = ½‖f(u + α δu)‖², ϕ'(0) = fuᵀ J δu */
  double  ls_c1, ls_rho_hi, ls_rho_lo;
  int32_t ls_order;             /* 2 | 3 */
  int32_t ls_maxiters;
  /* --- built-in multigrid V-cycle as the Krylov right preconditioner (BRATU2D, single rank), re-linearised for every
   *     new Jacobian like `precs(A, p)`: mg_nu smoothing steps (0 = off), coarsest grid side ≤ mg_coarse (0 → 31) */
  int32_t mg_nu;
  int32_t mg_coarse;
  /* --- concrete Jacobian of the built-in problems: 0 = closed-form values (what `f.jac` would supply), 1 = the
   *     colour-compressed assembly of `AutoSparse` + column colouring (ncolors seeded JVPs + decompression,
   *     lib/NonlinearSolveBase/src/jacobian.jl:244-247) every time the Jacobian is refreshed */
  int32_t jac_colored;
  /* --- LevenbergMarquardt (lib/NonlinearSolveFirstOrder/src/levenberg_marquardt.jl:37-64; constructor defaults in
   *     brackets): GeodesicAcceleration(DampedNewtonDescent(LM damping)) + LevenbergMarquardtTrustRegion */
  int32_t lm_disable_geodesic;          /* [0] 1 = plain damped Newton step (disable_geodesic = Val(true))           */
  double  lm_damping_initial;           /* [1]    λ₀                                                                  */
  double  lm_damping_increase_factor;   /* [2]    λ ← 2λ after a step that was not taken                              */
  double  lm_damping_decrease_factor;   /* [3]    λ ← λ/3 after an accepted step                                      */
  double  lm_min_damping_D;             /* [1e-8] floor of DᵀD = running max of diag(JᵀJ)                             */
  double  lm_alpha_geodesic;            /* [0.75] a step is taken only if 2‖a‖ ≤ α‖v‖                                 */
  double  lm_finite_diff_step_geodesic; /* [0.1]  h of the second directional derivative                              */
  double  lm_b_uphill;                  /* [1]    uphill moves: (1 − β)^b · ‖f_new‖ ≤ loss_old                        */
  /* --- PseudoTransient (lib/NonlinearSolveFirstOrder/src/pseudo_transient.jl:37-57): (J + α⁻¹ I) δ = f with switched
   *     evolution relaxation α⁻¹ ← α⁻¹·‖f‖₂/‖f_prev‖₂; mass matrix: identity, or a diagonal through
   *     nk_solver_set_mass_matrix_diagonal */
  double  pt_alpha_initial;             /* [1e-3] initial pseudo time step α                                          */
  int32_t gmres_sstep;                  /* [0]    NK_ORTHO_SSTEP: basis columns per block (1..16; 0 = automatic)      */
  int32_t gmres_sstep_basis;            /* [0]    nk_ss_basis                                                         */
  /* --- a built-in preconditioner object on the concrete J, rebuilt numerically for every new Jacobian like `precs(A, p)`
   *     (needs a concrete-J linsolve): 0 none, 1 Jacobi, 2 ILU(0) in the matrix's ordering, 3 ILU(0) multicolour,
   *     4 aggregation algebraic multigrid (nk_precond_create_amg with its defaults) */
  int32_t precond_kind;                 /* [0]                                                                        */
  int32_t precond_side;                 /* [1]    nk_side: the reference's tutorial precs return (Pl, I): left        */
} nk_options;

/* in-place callbacks of a user problem: NonlinearFunction{true}(f!; jvp, vjp, jac)
 * (signatures pinned by lib/SciMLJacobianOperators/test/core_tests__item2.jl:38-40,62-67 and
 *  lib/NonlinearSolveFirstOrder/test/rootfind_tests__item20.jl:17-32). All pointers are DEVICE pointers
 *  of local length n; `stream` is the hipStream_t the work must be ordered on. Return 0 on success. */
typedef int (*nk_residual_fn)(void *user, const double *u, double *f, void *stream);
typedef int (*nk_jvp_fn)(void *user, const double *v, const double *u, double *Jv, void *stream);
typedef int (*nk_vjp_fn)(void *user, const double *v, const double *u, double *vJ, void *stream);
typedef int (*nk_jacvals_fn)(void *user, const double *u, double *csr_vals, void *stream);
typedef struct {
  nk_residual_fn residual; /* required */
  nk_jvp_fn jvp;           /* NULL ⇒ forward differences through `residual`, (f(u+εv) − f(u))/ε, ε = √eps — the
                            * AutoFiniteDiff pushforward of SciMLJacobianOperators.jl:396-414 */
  nk_vjp_fn vjp;           /* required for TrustRegion on the matrix-free path */
  nk_jacvals_fn jac_values;/* NULL ⇒ concrete-J linsolves assemble J on the pattern given at create from ncolors seeded
                            * JVPs + decompression (greedy column colouring, cached per pattern; jacobian.jl:244-247) */
} nk_user_callbacks;

/* generic operator for nk_gmres: y = A x on device (an AbstractSciMLOperator / FunctionOperator) */
typedef int (*nk_matvec_fn)(void *user, const double *x, double *y, void *stream);

/* communicator callbacks (used for gloo/CPU-side collectives in tests, or any host transport).
 * Buffers are DEVICE pointers; the callback must have completed the operation (device-visible) before
 * it returns, after synchronising `stream` itself if it needs the data on the host. */
typedef struct {
  int (*allreduce)(void *user, double *buf, int count, int op /*0 sum, 1 max*/, void *stream);
  /* exchange bytes with every peer: send_bytes[p] bytes at send+send_off[p] to peer p, receive likewise */
  int (*alltoallv)(void *user, const void *send, const int64_t *send_off, const int64_t *send_bytes,
                   void *recv, const int64_t *recv_off, const int64_t *recv_bytes, void *stream);
  void *user;
} nk_comm_callbacks;

/* ---------------------------------------------------------------- library / context */
const char *nk_version(void);
const char *nk_last_error(void);              /* thread-local; never NULL */
int nk_device_count(int *count);

/* device_id: HIP ordinal. stream: the hipStream_t to enqueue on (NULL = the default/null stream). */
int nk_ctx_create(int device_id, void *stream, nk_ctx **out);
int nk_ctx_destroy(nk_ctx *ctx);
int nk_ctx_set_stream(nk_ctx *ctx, void *stream);
int nk_ctx_synchronize(nk_ctx *ctx);
/* deterministic=1 (default): fixed-order two-stage reductions, bitwise reproducible run to run. */
int nk_ctx_set_deterministic(nk_ctx *ctx, int deterministic);
/* Multi-rank CSR SpMV: run the halo exchange on a second stream while the row blocks that read no halo column are
 * computed (off by default; also enabled by NK_HALO_OVERLAP=1 in the environment at context creation). */
int nk_ctx_set_halo_overlap(nk_ctx *ctx, int on);

/* Device buffers for host languages that have no GPU array type of their own (a Julia session without AMDGPU.jl, plain C):
 * with them every vector argument of the ABI can be passed with memspace = NK_DEVICE and stays resident between calls.
 * nk_device_copy kind: 0 host→device, 1 device→host, 2 device→device; ordered on the context's stream, complete on return. */
int nk_device_alloc(nk_ctx *ctx, int64_t bytes, void **out);
int nk_device_free(nk_ctx *ctx, void *ptr);
int nk_device_copy(nk_ctx *ctx, void *dst, const void *src, int64_t bytes, int kind);
/* In-place updates of resident vectors (DEVICE pointers, local length n) for host languages whose resident vector type is a
 * library buffer (Julia's DeviceVector): y = a x + b y and y = a — with nk_dot / nk_nrm2 / nk_norm_inf / nk_axpy below, what
 * `@bb axpy!`, `copyto!` and the termination norms of the reference's step! need (FirstOrder/src/solve.jl:403,438,460). */
int nk_vec_axpby(nk_ctx *ctx, int64_t n, double a, const double *x, double b, double *y);
int nk_vec_fill(nk_ctx *ctx, int64_t n, double a, double *y);

/* Per-kernel-family timing (bench.py's roofline numbers). Off by default; when on, every launch of a profiled
 * family is issued with hipExtLaunchKernelGGL start/stop events, i.e. the kernel's own begin/end device
 * timestamps on the context's stream — the quantity rocprofv3's kernel trace reports. `launches` counts logical
 * operations (a multidot over 31 columns is one operation although it is two kernel launches). */
int nk_ctx_profile_enable(nk_ctx *ctx, int on);          /* on=1 also resets the accumulators */
int nk_ctx_profile_kernel_count(void);
int nk_ctx_profile_query(nk_ctx *ctx, int kernel_id, const char **name, int64_t *launches,
                         double *total_ms, double *total_algorithmic_bytes);

/* One process per GPU. Rank 0 creates an id, the host language broadcasts the 128 bytes
 * (torch.distributed / MPI.jl / Distributed.jl), every rank calls nk_ctx_comm_init_rccl. */
int nk_comm_unique_id(char id_out[128]);
int nk_ctx_comm_init_rccl(nk_ctx *ctx, int nranks, int rank, const char id[128]);
int nk_ctx_comm_init_callbacks(nk_ctx *ctx, int nranks, int rank, const nk_comm_callbacks *cb);
/* xGMI-native small collectives on top of either communicator (which keeps serving set-up and large exchanges): every
 * rank allocates an uncached arena on its GPU and exports it (hipIpcGetMemHandle, 64 bytes); the host language
 * all-gathers the handles; every rank maps all of them. Afterwards the Krylov all-reduces (≤ 128 doubles) and the halo
 * exchanges run as ONE small kernel each: a rank stores its partial sums / its halo entries straight into every peer's
 * arena, releases a sequence flag (system scope), polls the flags of its peers and combines in rank order — bitwise
 * reproducible, no RCCL launch (≈ 20 µs) on the critical path of an Arnoldi step.
 *   nk_ctx_comm_peer_handle : allocate the arena (arena_bytes ≤ 0: 64 MiB) and return its IPC handle
 *   nk_ctx_comm_enable_peer : handles = nranks × 64 bytes in rank order (this rank's own entry is ignored)
 *   nk_ctx_comm_peer_status : *enabled; *errors = time-outs seen by the device kernels (0 in a healthy run) */
#define NK_IPC_HANDLE_BYTES 64
int nk_ctx_comm_peer_handle(nk_ctx *ctx, int64_t arena_bytes, char handle_out[NK_IPC_HANDLE_BYTES]);
int nk_ctx_comm_enable_peer(nk_ctx *ctx, const char *handles);
int nk_ctx_comm_peer_status(nk_ctx *ctx, int *enabled, int64_t *errors);
/* the communicator's all-reduce on a DEVICE buffer, in place (op 0 = sum, 1 = max), through whatever transport the context
 * holds (peer-mapped arenas up to 512 values, RCCL, callbacks); blocking. For self-checks of a multi-GPU set-up
 * (tools/multi_gpu_selfcheck.py) and for host code that needs a global reduction of its own. */
int nk_ctx_comm_allreduce(nk_ctx *ctx, double *buf, int count, int op);
/* collective: a few all-reduces with known answers; if any rank sees a wrong value or a time-out the fast path is
 * switched off on every rank (*ok = 0) and the base transport serves all collectives */
int nk_ctx_comm_peer_selftest(nk_ctx *ctx, int *ok);
int nk_ctx_comm_peer_disable(nk_ctx *ctx);   /* before any problem / matrix was created on the context */
int nk_ctx_comm_info(nk_ctx *ctx, int *kind, int *nranks, int *rank);
/* *shared = 1 if several ranks of the communicator run on ONE device (found out collectively when the communicator was set up,
 * from the devices' PCI bus ids; NK_DEVICE_SHARED=0/1, set alike on every rank, overrides): such ranks do not use the one form that
 * needs a whole device per rank — the resident matrix-powers kernel on ranks. One process per GPU: 0.
 * nk_ctx_comm_init_rccl / nk_ctx_comm_init_callbacks are COLLECTIVE for this reason (one small all-reduce through the transport). */
int nk_ctx_comm_device_shared(nk_ctx *ctx, int *shared);

/* contiguous row-range partition used everywhere: rank r owns [r*n/P, (r+1)*n/P) rounded down to a
 * multiple of `granule` (a grid line for stencil problems). Pure host logic, no device needed. */
int nk_partition_range(int64_t n_global, int64_t granule, int nranks, int rank,
                       int64_t *row_begin, int64_t *row_end);

/* ---------------------------------------------------------------- sparse matrices */
/* Local rows [row_begin, row_begin+nrows_local) of a square n_global matrix, GLOBAL column ids.
 * index_bits 32|64, index_base 0|1 (Julia). Copies (and for nranks>1 builds the halo plan — collective).
 * vals may be NULL (pattern only; fill later with nk_csr_set_values / nk_jac_values). */
int nk_csr_create(nk_ctx *ctx, int64_t nrows_local, int64_t n_global, int64_t row_begin, int64_t nnz,
                  int index_bits, int index_base, const void *rowptr, const void *colind,
                  const double *vals, int memspace, nk_csr **out);
/* Julia's SparseMatrixCSC{Float64,Int} (colptr,rowval,nzval) of the WHOLE n × n matrix, as every rank holds it:
 * converts CSC→CSR once on the host and keeps this rank's row range (nk_partition_range with granule 1, or the
 * explicit range of the _rows form — whole grid lines for stencil problems). Collective on several ranks. */
int nk_csr_create_from_csc(nk_ctx *ctx, int64_t n, int64_t nnz, int index_bits, int index_base,
                           const void *colptr, const void *rowval, const double *nzval, nk_csr **out);
int nk_csr_create_from_csc_rows(nk_ctx *ctx, int64_t n, int64_t nnz, int index_bits, int index_base,
                                const void *colptr, const void *rowval, const double *nzval,
                                int64_t row_begin, int64_t nrows_local, nk_csr **out);
int nk_csr_destroy(nk_csr *A);
int nk_csr_set_values(nk_csr *A, const double *vals, int memspace);
/* New values for a matrix created by nk_csr_create_from_csc(_rows), given in the CSC order of that call (Julia:
 * `nonzeros(J)` of the SparseMatrixCSC `f.jac(J, u, p)` refreshes every Newton step): one gather on the device through the
 * permutation remembered at creation — no new conversion, no new pattern upload. nnz_csc = length of nzval. */
int nk_csr_set_values_csc(nk_csr *A, const double *nzval, int64_t nnz_csc, int memspace);
int nk_csr_get_values(nk_csr *A, double *vals, int memspace);
int nk_csr_info(nk_csr *A, int64_t *nrows_local, int64_t *n_global, int64_t *nnz, int64_t *n_halo);
double *nk_csr_values_device(nk_csr *A);      /* device pointer of the local values (nnz doubles) */
/* y = A x  (x, y local slices; halo exchanged internally).  nk_spmv_t: y = Aᵀ x (the contributions to entries other
 * ranks own return through the halo plan in reverse and are added in rank order: bitwise reproducible). */
int nk_spmv(nk_csr *A, const double *x, double *y, int memspace);
int nk_spmv_t(nk_csr *A, const double *x, double *y, int memspace);
/* Matrix powers — the s operator applications an s-step Arnoldi block makes back to back (the `mul!(Jv, A, v)` of
 * lib/SciMLJacobianOperators/src/SciMLJacobianOperators.jl:238-243, s times in a row on a concrete J):
 *   Y[:, p] = scale·(A Y[:, p−1] − θ_p Y[:, p−1]),  p = 0 … s−1,  Y[:, −1] = x   (theta: s host values, NULL = plain powers;
 * ldy ≥ local rows). A banded matrix small enough to be held in the chip's vector registers (≤ #CUs × 6144 rows, ≤ 5 / 8 / 16
 * entries per row, columns within ±1024 rows of their band; one rank) takes ONE launch that reads the matrix once
 * (*resident = 1, csrc/nk_powers.hip) — also a matrix made of two equal segments that is banded segment by segment (the
 * (i, j, species) ordering of a two-species system, docs/src/tutorials/large_systems.md) and / or whose bands close to a ring
 * (periodic boundaries); any other matrix takes s streaming SpMV launches. The columns are bit-identical either way.
 * NK_SPMV_POWERS=0 disables the resident kernel.
 * Several ranks: whether the resident kernel applies is decided ONCE per matrix by all ranks together — the first nk_csr_powers
 * or s-step linear solve on a row-partitioned matrix is COLLECTIVE (two small all-reduces through the communicator); every rank
 * must reach it, also a rank whose share would not qualify. */
int nk_csr_powers(nk_csr *A, const double *x, double *Y, int64_t ldy, int s, const double *theta, double scale, int memspace,
                  int *resident);
/* Which form of the resident kernel a CSR PATTERN (global column ids, one rank) fits on a device with num_cus compute units —
 * host arithmetic only, no device needed: layout[0] = 0 none (streaming launches), 1 plain bands, 2 segments and / or ring;
 * [1] slices of 1024 rows per band (and segment), [2] register slots per row (5 / 8 / 16), [3] bands (= workgroups),
 * [4] segments, [5] 1 = the bands of a segment close to a ring. (The device additionally needs all bands resident at once.) */
int nk_csr_powers_layout(int64_t nrows, const int32_t *rowptr, const int32_t *col, int num_cus, int layout[6]);
/* out_j = Σ_i A_ij² — diag(AᵀA), what LevenbergMarquardt's damping takes its DᵀD from (`sum!(abs2, J_diag_cache, J')`,
 * levenberg_marquardt.jl:133-148). Row-partitioned matrices: the same reverse halo exchange as nk_spmv_t. */
int nk_csr_colsumsq(nk_csr *A, double *out, int memspace);

/* ---------------------------------------------------------------- problems (seam 2) */
/* params: QUADRATIC {n, p}; BRATU2D {n_side, lambda, scale (0 → h², i.e. h²F)};
 *         BRUSSELATOR2D {N_g, A, B, alpha, dx}.  The problem partitions itself over the ctx's ranks. */
int nk_problem_create(nk_ctx *ctx, int kind, const double *params, int nparams, nk_problem **out);
/* csr pattern (local rows, global columns, 0-based int32/int64) is optional: NULL ⇒ no concrete J. */
int nk_problem_create_user(nk_ctx *ctx, int64_t n_local, int64_t n_global, int64_t row_begin,
                           const nk_user_callbacks *cb, void *user, nk_csr *jac_pattern,
                           nk_problem **out);
int nk_problem_destroy(nk_problem *P);
int nk_problem_size(nk_problem *P, int64_t *n_local, int64_t *n_global, int64_t *row_begin);
int nk_problem_set_params(nk_problem *P, const double *params, int nparams);  /* reinit!(cache; p) */
int nk_problem_initial_guess(nk_problem *P, double *u0, int memspace);
int nk_residual(nk_problem *P, const double *u, double *f, int memspace);
int nk_jvp(nk_problem *P, const double *u, const double *v, double *Jv, int memspace);
int nk_vjp(nk_problem *P, const double *u, const double *v, double *vJ, int memspace);
/* concrete sparse Jacobian: pattern once, closed-form values per call (f.jac(J,u,p), jacobian.jl:241) */
int nk_problem_jac_csr(nk_problem *P, nk_csr **out);
int nk_jac_values(nk_problem *P, const double *u, int memspace, nk_csr *J);
/* colour-compressed assembly: ncolors JVPs with seed vectors + decompression — the structure of
 * DI.jacobian! with AutoSparse (jacobian.jl:244-247; colouring ext/...SparseMatrixColoringsExt.jl:13-28) */
int nk_jac_values_colored(nk_problem *P, const double *u, int memspace, nk_csr *J, int *ncolors);

/* ---------------------------------------------------------------- GMRES (seam 1) */
int nk_gmres_create(nk_ctx *ctx, int64_t n_local, int restart_m, int ortho, nk_gmres **out);
int nk_gmres_destroy(nk_gmres *G);
int nk_gmres_set_operator_csr(nk_gmres *G, nk_csr *A);                    /* cache.A = SparseMatrix   */
int nk_gmres_set_operator_jvp(nk_gmres *G, nk_problem *P, const double *u, int memspace); /* Stateful… */
int nk_gmres_set_operator_fn(nk_gmres *G, nk_matvec_fn fn, void *user);
/* on != 0: the operator of the next solves is AᵀA for the CSR / problem operator that is set (normal form: the
 * transposed half is the distributed transposed SpMV or the problem's VJP); the caller passes b = Aᵀ f. */
int nk_gmres_set_normal_form(nk_gmres *G, int on);   /* AbstractSciMLOperator    */
/* NK_ORTHO_SSTEP: number of basis columns per block, 1..16; 0 = automatic (15 with the Newton basis, 6 with the monomial
 * one). The last block of a cycle is cut to fit the restart length. */
int nk_gmres_set_block_size(nk_gmres *G, int s);
/* NK_ORTHO_SSTEP: basis of a block (nk_ss_basis) */
int nk_gmres_set_sstep_basis(nk_gmres *G, int basis);
/* Real bounds lo < hi of the operator's spectrum (its real part) for operators the library cannot bound itself — callback
 * and matrix-free operators. They place the Newton-basis shifts of NK_ORTHO_SSTEP; approximate bounds are fine (the
 * shifts only condition the block, the Krylov space is the same). lo = hi = 0 forgets them. */
int nk_gmres_set_spectrum_interval(nk_gmres *G, double lo, double hi);
/* NK_ORTHO_SSTEP diagnostics: the block size in effect (automatic sizes narrow 15 → 8 → 4 after a block lost rank), whether
 * the last solve built Newton-basis blocks, and how many blocks have lost rank so far (each made its cycle run again —
 * narrower, or column by column). Any pointer may be NULL. */
int nk_gmres_get_sstep_state(nk_gmres *G, int *block_size, int *newton_basis, int *breakdowns);
/* Damped normal form: the operator becomes AᵀA + lambda·diag(d) (d: DEVICE vector of local length n, kept by reference;
 * NULL switches the damping off) — `dampen_jacobian!!(J_cache, JᵀJ, λ·DᵀD)` of DampedNewtonDescent's :normal_form mode
 * (lib/NonlinearSolveBase/src/descent/damped_newton.jl:297-313,356-370) without assembling JᵀJ. */
int nk_gmres_set_normal_form_damping(nk_gmres *G, const double *d_diag, double lambda);
/* Shifted operator: A + sigma·I for the operator that is set (0 switches it off) — `J + D` for a matrix-free Jacobian
 * operator under DampedNewtonDescent (`dampen_jacobian!!(::Any, J::AbstractSciMLOperator, D) = J + D`, damped_newton.jl:349). */
int nk_gmres_set_shift(nk_gmres *G, double sigma);
/* … and A + sigma·diag(m) with a DEVICE vector m of local length n (kept by reference; NULL = identity): the damping α⁻¹ M of
 * PseudoTransient(; mass_matrix = Diagonal(m)) (pseudo_transient.jl:102-120,149). */
int nk_gmres_set_shift_weights(nk_gmres *G, const double *d_m);
/* ---- preconditioners: the `precs(A, p) -> (Pl, Pr)` hook of the LinearSolve interface
 * (lib/NonlinearSolveBase/src/linear_solve.jl:195-199 wrap_preconditioners; test/Core/core_tests__item21.jl:10-18;
 *  docs/src/tutorials/large_systems.md:252-316, whose `precs` return `(Pl, I)`). GMRES solves Pl⁻¹ A Pr⁻¹ (Pr x) = Pl⁻¹ b:
 * the Arnoldi process runs on Pl⁻¹ A Pr⁻¹, and with a left preconditioner the stopping test
 * ‖Pl⁻¹(b − A x)‖ ≤ atol + rtol·‖Pl⁻¹(b − A x₀)‖ and nk_gmres_info.rnorm0 / rnorm refer to the PRECONDITIONED residual, as in
 * Krylov.jl's gmres [EXT]. Either side takes a callback (device or host pointers) or a built-in object; fn / P = NULL removes
 * that side. Callbacks must be linear maps y = P⁻¹ x. */
/* right preconditioner x = Pr⁻¹ z applied as a device callback */
int nk_gmres_set_right_preconditioner(nk_gmres *G, nk_matvec_fn fn, void *user);
/* left preconditioner y = Pl⁻¹ r applied as a device callback */
int nk_gmres_set_left_preconditioner(nk_gmres *G, nk_matvec_fn fn, void *user);
int nk_gmres_set_left_preconditioner_host(nk_gmres *G, nk_matvec_fn fn, void *user);
/* a built-in preconditioner object (below) on either side; the object stays the caller's (it must outlive its use here,
 * and nk_precond_update is the caller's to call when the matrix values change) */
typedef enum { NK_SIDE_RIGHT = 0, NK_SIDE_LEFT = 1 } nk_side;
typedef struct nk_precond nk_precond;
int nk_gmres_set_preconditioner(nk_gmres *G, int side, nk_precond *P);

/* Preconditioner objects built from a CSR matrix (csrc/nk_precond.hip) — what `precs(A, p)` returns for a general sparse
 * Jacobian (the reference's tutorial fills the slot with IncompleteLU.ilu(W) / an algebraic multigrid):
 *   Jacobi: M = diag(A).  ILU(0): A ≈ LU on the pattern of A, no fill, no pivoting, of the rank's LOCAL square block (halo
 *   columns dropped: block-Jacobi ILU(0) across ranks, no communication in the apply). Rows are scheduled by dependency levels;
 *   NK_ILU_NATURAL keeps the matrix's ordering (the classical preconditioner; a lexicographic stencil has 2n − 1 levels, walked
 *   inside one persistent workgroup — milliseconds per application at n = 1024²), NK_ILU_MULTICOLOR permutes the rows by a
 *   greedy colouring of the pattern (as many levels as colours, one wide launch each: the GPU form).
 * nk_precond_update refactorises for the matrix's current values; NK_E_SINGULAR on a zero pivot. x, y: local length n. */
typedef enum { NK_PRECOND_JACOBI = 1, NK_PRECOND_ILU0 = 2, NK_PRECOND_AMG = 3, NK_PRECOND_ILUT = 4 } nk_precond_kind;
typedef enum { NK_ILU_NATURAL = 0, NK_ILU_MULTICOLOR = 1 } nk_ilu_ordering;
int nk_precond_create_jacobi(nk_csr *A, nk_precond **out);
int nk_precond_create_ilu0(nk_csr *A, int ordering, nk_precond **out);
/* ILU with a drop tolerance — the tutorial's `incompletelu(W, p) = (ilu(W, τ = 50.0), I)` (docs/src/tutorials/large_systems.md:252-260;
 * IncompleteLU.jl [EXT]): Crout ILU (Li, Saad, Chow 2003), A ≈ (I + L) U with fill; an off-diagonal entry of U's row k / L's column
 * k is kept if its magnitude BEFORE the division by the pivot is ≥ tau (absolute; tau = 0: the complete LU without pivoting). The
 * factorisation runs on the HOST for every nk_precond_update (its pattern depends on the numbers — the reference's runs on the
 * CPU as well), of the rank's local block; the triangular solves of every application run on the device, level-scheduled from the
 * factors' pattern. nk_precond_ilu0_factors returns its factors as well. NK_E_SINGULAR on a zero pivot. */
int nk_precond_create_ilut(nk_csr *A, double tau, nk_precond **out);
/* Aggregation algebraic multigrid built from the matrix alone (csrc/nk_amg.hip) — the tutorial's
 * `precs = (A, p) -> (aspreconditioner(ruge_stuben(A)), I)` slot (docs/src/tutorials/large_systems.md:276-316): pairwise
 * aggregation (`passes` times per level: aggregates of ≤ 2^passes rows; strength threshold `theta`) fixed at creation from the
 * values of that moment, piecewise-constant transfers, Galerkin coarse matrices, `nu` Chebyshev steps on D⁻¹A over
 * [λmax / cheb_ratio, λmax] before and after, the coarse correction over-corrected by `overcorrection`, levels down to
 * ≤ coarse_max (≤ 128) rows, inverted densely. nk_precond_update keeps the aggregates and refreshes every number on the
 * device (Galerkin sums, D⁻¹, Gershgorin bounds, the coarse inverse) for the matrix's current values. A fixed linear
 * operator: usable as Pl or Pr. Several ranks: the rank's local block (block-Jacobi AMG). params NULL or zero fields:
 * defaults (nu 2, passes 2, theta 0.25, overcorrection 1.8, cheb_ratio 4, coarse_max 128). */
typedef struct nk_amg_params {
  int32_t nu, passes, coarse_max, matching;
  double theta, overcorrection, cheb_ratio;
} nk_amg_params;
int nk_amg_params_default(nk_amg_params *p);
int nk_precond_create_amg(nk_csr *A, const nk_amg_params *params, nk_precond **out);
/* how the aggregates were formed: 1 = the sequential pairwise pass (host set-up), 2 = handshaking (device set-up) */
int nk_precond_amg_matching(nk_precond *P, int *matching);
/* the hierarchy, for inspection: *levels (coarsest included); sizes / nnzs / lmax: up to `cap` entries each (NULL: skipped) */
int nk_precond_amg_info(nk_precond *P, int *levels, int cap, int64_t *sizes, int64_t *nnzs, double *lmax);
/* row → coarse row of level `level` (`count` = that level's rows; the coarsest level has none: NK_E_INVALID) */
int nk_precond_amg_aggregates(nk_precond *P, int level, int32_t *agg, int64_t count);
int nk_precond_update(nk_precond *P);
int nk_precond_apply(nk_precond *P, const double *x, double *y, int memspace);   /* y = M⁻¹ x */
int nk_precond_info(nk_precond *P, int *kind, int *levels_lower, int *levels_upper, int *ncolors);
/* the ILU(0) factors in the (permuted) ordering, for inspection: CSR of the local block — L strictly below the diagonal
 * (unit diagonal implied), U on and above — and perm[permuted row] = original row. Any pointer may be NULL. */
int nk_precond_ilu0_factors(nk_precond *P, int64_t *nnz, int32_t *rowptr, int32_t *col, double *val, int32_t *perm);
int nk_precond_destroy(nk_precond *P);
/* The same two hooks for operators that live in HOST memory (a Julia `mul!` on plain Arrays): the callback receives host
 * pointers, the library stages the vectors through pinned buffers around every call (2 × 8 n bytes over PCIe each). */
int nk_gmres_set_operator_fn_host(nk_gmres *G, nk_matvec_fn fn, void *user);
int nk_gmres_set_right_preconditioner_host(nk_gmres *G, nk_matvec_fn fn, void *user);
/* Built-in right preconditioner M⁻¹ = p_d(A): `degree` steps of the Chebyshev iteration on [lambda_min, lambda_max]
 * (operator applications only: no inner products, no all-reduce). lambda_max = 0 ⇒ the dominant eigenvalue is
 * bounded by Gershgorin (concrete CSR) or estimated by 30 power iterations ×1.15 (matrix-free / callback
 * operators) and lambda_min = lambda_max / ratio. The interval must not contain 0
 * (negative-definite operators are fine). degree ≤ 0 removes it. Call after the operator is set. */
int nk_gmres_set_chebyshev_preconditioner(nk_gmres *G, int degree, double lambda_min, double lambda_max, double ratio);
int nk_gmres_get_chebyshev_interval(nk_gmres *G, double *lambda_min, double *lambda_max);
/* Built-in right preconditioner M⁻¹ = one geometric multigrid V-cycle on the Jacobian of a BRATU2D problem linearised at u
 * (level operators by rediscretisation, bilinear transfers between non-nested grids, `nu` Chebyshev smoothing steps before
 * and after, banded LU on the coarsest grid of side ≤ coarse_max; 0 → 2 and 31) — the device counterpart of an algebraic-
 * multigrid `precs` (docs/src/tutorials/large_systems.md:244-316). Mesh-independent Krylov iteration counts. Call it again
 * for every new linearisation point; nu ≤ 0 or P == NULL removes it. Single rank. */
int nk_gmres_set_multigrid_preconditioner(nk_gmres *G, nk_problem *P, const double *u, int memspace, int nu, int coarse_max);
/* Solve A x = b. use_x0 = 0 ⇒ zero initial guess (our protocol); 1 ⇒ x holds x0.
 * Stop when ‖r‖₂ ≤ atol + rtol‖r0‖₂ or after maxiter Arnoldi steps; fixed_iters>0 overrides both. */
int nk_gmres_solve(nk_gmres *G, const double *b, double *x, int memspace, int use_x0,
                   double atol, double rtol, int maxiter, int fixed_iters, nk_gmres_info *info);

/* ---------------------------------------------------------------- direct factorisation (seam 1, `linsolve = nothing`)
 * Direct factorisation of a concrete banded sparse J on the device; factor once, solve many
 * (reuse_A_if_factorization, lib/NonlinearSolveBase/ext/NonlinearSolveBaseLinearSolveExt.jl:81-86). Single rank.
 * Engine (nk_lu_engine): block cyclic reduction — batched dense blocks of order b = bandwidth rounded to 32 (b <= 512) on FP64
 * MFMA, log2(n/b) dependent levels — or, for matrices of fewer than four block rows, a right-looking band LU (lower
 * bandwidth <= ~550). Pivots: on the diagonal first; the cyclic-reduction engine switches to row pivoting inside its dense
 * blocks when a diagonal pivot vanishes (NK_BCR_PIVOT=auto|always|never); *ok = 0 when the factorisation still broke down. */
int nk_lu_create(nk_csr *A, nk_lu **out);
int nk_lu_destroy(nk_lu *F);
int nk_lu_factor(nk_lu *F, nk_csr *A, int *ok);
int nk_lu_solve(nk_lu *F, const double *b, double *x, int memspace);
int nk_lu_info(nk_lu *F, int *kl, int *ku, int64_t *band_bytes);   /* band_bytes: device memory held by the factorisation */
/* which engine the factorisation object runs on: 0 = right-looking band LU (a chain of n/32 dependent block columns),
 * 1 = block cyclic reduction (log2(n/b) levels of batched dense b x b algebra on FP64 MFMA; chosen automatically when the
 * matrix has >= 4 block rows of order b = bandwidth rounded to 32, b <= 512; NK_DIRECT=band in the environment forces 0);
 * `block` = b (0 for the band LU), `levels` = number of reduction levels. */
int nk_lu_engine(nk_lu *F, int *engine, int *block, int *levels);

/* ---------------------------------------------------------------- Newton / TrustRegion (seam 3) */
int nk_options_default(nk_options *opts);
int nk_solver_init(nk_problem *P, const double *u0, int memspace, const nk_options *opts,
                   nk_solver **out);                       /* SciMLBase.__init */
int nk_solver_destroy(nk_solver *S);
int nk_solver_step(nk_solver *S);                          /* CommonSolve.step!  */
/* step!(cache; recompute_jacobian, evaluate_residual) (lib/NonlinearSolveBase/src/solve.jl:835-859 →
 * lib/NonlinearSolveFirstOrder/src/solve.jl:325-465). recompute_jacobian: -1 = nothing (the algorithm decides), 0, 1.
 * evaluate_residual = 0 is a HINT: honoured only when nk_solver_supports_deferred_residual says 1 (unglobalised Newton
 * step, AbsTerminationMode / AbsNormTerminationMode, no trace — FirstOrder/src/solve.jl:303-316); the step then ends
 * right after u += δu and nk_solver_refresh_residual (= refresh_residual!, :318-324) evaluates f(u) and runs the
 * termination check on demand. step and solve settle an outstanding deferral themselves. */
int nk_solver_step_ex(nk_solver *S, int recompute_jacobian, int evaluate_residual);
int nk_solver_supports_deferred_residual(nk_solver *S, int *yes);
int nk_solver_refresh_residual(nk_solver *S);
int nk_solver_solve(nk_solver *S, int *retcode);           /* CommonSolve.solve! */
int nk_solver_reinit(nk_solver *S, const double *u0, int memspace,
                     const double *params, int nparams);   /* SciMLBase.reinit!(cache, u0; p) */
/* PseudoTransient(; mass_matrix = Diagonal(m)) (pseudo_transient.jl:37-57): damping α⁻¹·diag(m); m has the local length of u,
 * is copied, and must be set before the first step (the reference fixes it at init). NULL returns to the identity. */
int nk_solver_set_mass_matrix_diagonal(nk_solver *S, const double *m, int memspace);
int nk_solver_get_u(nk_solver *S, double *u, int memspace);
int nk_solver_get_resid(nk_solver *S, double *f, int memspace);
int nk_solver_get_stats(nk_solver *S, nk_stats *stats);
/* The `precs(A, p) -> (Pl, Pr)` hook at solver level (KrylovJL_GMRES(precs = …); test/Core/core_tests__item21.jl:10-37):
 * `fn` is called once when it is installed — LinearSolve evaluates precs when the linear cache is built [EXT] — and then
 * right after every Jacobian refresh (a new `A` marks the LinearSolve cache fresh: lib/NonlinearSolveBase/ext/
 * NonlinearSolveBaseLinearSolveExt.jl:64-111), before that step's linear solve; never by nk_solver_reinit. It receives the
 * solver's GMRES object — install Pl / Pr on it with nk_gmres_set_left/right_preconditioner(_host) or
 * nk_gmres_set_preconditioner —, the concrete Jacobian (NULL on the matrix-free path) and the iterate (DEVICE pointer, local
 * length n: the `u` of LinearSolveParameters(u, p)). Non-zero return = NK_E_CALLBACK. fn = NULL removes the hook. */
typedef int (*nk_precs_fn)(void *user, nk_gmres *G, nk_csr *A, const double *u);
int nk_solver_set_precs(nk_solver *S, nk_precs_fn fn, void *user);
/* the solver's GMRES object and concrete Jacobian (NULL where the linsolve has none), e.g. to build preconditioner objects on */
nk_gmres *nk_solver_gmres(nk_solver *S);
nk_csr *nk_solver_jacobian(nk_solver *S);
int nk_solver_get_retcode(nk_solver *S, int *retcode, int *nsteps, int *force_stop);
int nk_solver_get_scalars(nk_solver *S, double *fnorm_inf, double *trust_region, double *eta);
int nk_solver_get_trace(nk_solver *S, nk_trace_entry *rows, int capacity, int *nrows);
/* one call: init + solve + results (what SciMLBase.__solve of the extension algorithm does) */
int nk_newton_solve(nk_problem *P, const double *u0, int memspace, const nk_options *opts,
                    double *u_out, double *resid_out, nk_stats *stats, int *retcode);

/* ---------------------------------------------------------------- ensembles of small systems (kernel generation)
 * The reference's "GPU acceleration over large parameter searches" (docs/src/tutorials/nonlinear_solve_gpus.md:70-176):
 * SimpleNewtonRaphson (lib/SimpleNonlinearSolve/src/raphson.jl:39-83) for thousands of small (n ≤ 64) systems
 * f(u, p_b) = 0, one system per GPU thread. `source` is HIP C++ defining
 *     template <typename T> __device__ void nk_f(const T *u, const double *p, T *f);
 * (and, with flags & 1, `__device__ void nk_jac(const double *u, const double *p, double *J)`, row-major n×n). It is
 * compiled at run time (hiprtc, gfx950) with the solver kernel; without nk_jac the Jacobian comes from forward-mode dual
 * numbers (AutoForwardDiff, the reference default). Semantics per system: iszero(f(u0)) ⇒ Success; δ = J \ f (partial
 * pivoting), u −= δ, then AbsNormTerminationMode(maximum∘abs) on the residual of the previous iterate; default abstol
 * eps^(4/5), maxiters 1000; per-system retcode (NK_RET_SUCCESS | NK_RET_MAXITERS) and iteration count.
 * nk_batch_compile_check compiles only (no device needed). */
int nk_batch_compile_check(const char *source, int n, int nparams, int flags, int64_t *code_bytes);
int nk_batch_create(nk_ctx *ctx, const char *source, int n, int nparams, int flags, nk_batch **out);
int nk_batch_destroy(nk_batch *B);
/* u0: n doubles shared by all systems (u0_per_system = 0) or nbatch×n; p: nbatch×nparams; outputs nbatch×n, nbatch×n,
 * nbatch, nbatch (retcode/iters nullable); abstol ≤ 0 and maxiters ≤ 0 select the defaults. */
int nk_batch_solve(nk_batch *B, int64_t nbatch, const double *u0, int u0_per_system, const double *p, int memspace,
                   double abstol, int maxiters, double *u_out, double *resid_out, int32_t *retcode_out, int32_t *iters_out);
/* SimpleTrustRegion (lib/SimpleNonlinearSolve/src/trust_region.jl:57-229, default radius update) per system; thresholds and
 * factors ≤ 0, max_shrink_times < 0 select the reference defaults (1e-4, 0.25, 0.75, 0.25, 2, 32). Retcodes: NK_RET_SUCCESS,
 * NK_RET_MAXITERS, NK_RET_SHRINK_THRESHOLD_EXCEEDED. */
int nk_batch_solve_trust_region(nk_batch *B, int64_t nbatch, const double *u0, int u0_per_system, const double *p, int memspace,
                                double abstol, int maxiters, double step_threshold, double shrink_threshold,
                                double expand_threshold, double shrink_factor, double expand_factor, int max_shrink_times,
                                double *u_out, double *resid_out, int32_t *retcode_out, int32_t *iters_out);

/* ---------------------------------------------------------------- BLAS-1 building blocks (exported for
 * the bench / tests; all on the ctx stream, results of reductions are all-reduced over the ranks) */
int nk_dot(nk_ctx *ctx, int64_t n, const double *x, const double *y, double *result);
int nk_nrm2(nk_ctx *ctx, int64_t n, const double *x, double *result);
int nk_norm_inf(nk_ctx *ctx, int64_t n, const double *x, double *result);
int nk_axpy(nk_ctx *ctx, int64_t n, double a, const double *x, double *y);
/* Krylov basis kernels: V is column-major n × nv with leading dimension ldv (device).
 * h[j] = V[:,j]·w  and  w -= V h (returns ‖w‖² after the update in *wnorm2 if non-NULL). */
int nk_multidot(nk_ctx *ctx, int64_t n, int nv, const double *V, int64_t ldv, const double *w, double *h_host);
int nk_multiaxpy(nk_ctx *ctx, int64_t n, int nv, const double *V, int64_t ldv, const double *h_host,
                 double *w, double *wnorm2);
/* fused CGS2 pass on a lazily-normalised basis (v_j = s_j ṽ_j): w -= V (h∘s); h2[j] = s_j ṽ_j·w (j<nv), h2[nv] = ‖w‖² */
int nk_fused_axpy_dot(nk_ctx *ctx, int64_t n, int nv, const double *V, int64_t ldv, const double *h_host,
                      const double *s_host, double *w, double *h2_host);

#ifdef __cplusplus
}
#endif
#endif /* MI355X_NK_H */
