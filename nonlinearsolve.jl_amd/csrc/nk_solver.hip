// Newton–Raphson / TrustRegion driver (seam 3): a host-side restatement of
//   lib/NonlinearSolveFirstOrder/src/solve.jl:140-301 (init), :325-465 (step!), :108-133 (reinit!)
//   lib/NonlinearSolveBase/src/solve.jl:360-387 (run to completion), :835-859 (step! bookkeeping)
//   lib/NonlinearSolveBase/src/descent/newton.jl:97-141, dogleg.jl:86-151, steepest.jl:57-80
//   lib/NonlinearSolveFirstOrder/src/trust_region.jl:204-258,292-317,320-384,396-514
//   lib/NonlinearSolveFirstOrder/src/eisenstat_walker.jl:42-107
//   lib/NonlinearSolveBase/src/termination_conditions.jl:243-336,414-453
//   lib/NonlinearSolveFirstOrder/src/levenberg_marquardt.jl:37-64,72-168,204-293 (LevenbergMarquardt),
//   lib/NonlinearSolveBase/src/descent/damped_newton.jl:224-345 (:normal_form mode), geodesic_acceleration.jl:98-136
// All vectors stay in device memory; per nonlinear step the host reads back a handful of scalars.
#include <math.h>
#include <string.h>

#include <chrono>

#include "nk_internal.h"

struct nk_solver {
  nk_problem *P = nullptr;
  nk_ctx *ctx = nullptr;
  nk_stats ctx_base{};  // context-wide counters (operator applications, all-reduces, halo exchanges) at (re)initialisation
  nk_options o{};
  int64_t n = 0;
  // vectors. The iterate lives in a pool of three buffers: `u` (current) and `best_u` (the retained best iterate of the
  // safe-best termination modes; may be the same buffer) point into it, and every new iterate / trial point is written
  // into a buffer that is neither — taking a step or keeping the best iterate is a pointer assignment, never a copy.
  double *ubuf[3] = {nullptr, nullptr, nullptr};
  double *u = nullptr, *fu = nullptr, *du = nullptr, *best_u = nullptr;
  double *u_trial = nullptr, *fu_trial = nullptr, *du_newton = nullptr, *du_cauchy = nullptr, *Jdu = nullptr,
         *JTfu = nullptr, *c1 = nullptr, *c2 = nullptr, *tr_du = nullptr, *stage = nullptr;
  double *stage2 = nullptr;   // iterative-refinement correction of the direct path
  bool fu_deferred = false;   // step!(…; evaluate_residual = false) left the residual of the new iterate unevaluated
  double fnorm2 = 0.0;        // ‖fu‖₂ of the current residual (Eisenstat–Walker reads it without another reduction)
  nk_gmres *G = nullptr;
  nk_csr *J = nullptr;
  bool own_J = false;
  nk_bandlu *B = nullptr;   // direct linsolve (NK_LINSOLVE_BANDED_LU)
  bool lu_valid = false;
  // driver state
  int nsteps = 0, retcode = NK_RET_DEFAULT;
  bool force_stop = false, make_new_jacobian = true;
  nk_stats stats{};
  double total_time = 0.0;
  uint64_t u_version = 0;
  // termination cache
  double abstol = 0, reltol = 0, best_obj = 0, initial_obj = 0, fnorm_inf = 0;
  double tc_u0_norm = 0;  // ‖u0‖₂ for the relative stall test
  int tc_nsteps = 0, tc_retcode = NK_RET_DEFAULT;
  std::vector<double> objectives_trace, step_norm_trace;
  // forcing
  double eta = 0, rnorm = 0, rnorm_prev = 0, lin_abstol = 0, lin_reltol = 0;
  // trust region
  double max_tr = 0, init_tr = 0, tr = 0, rho = 0, step_thr = 0, shrink_thr = 0, expand_thr = 0, shrink_f = 0,
         expand_f = 0, p1 = 0, p2 = 0, p3 = 0, p4 = 0;
  int shrink_counter = 0;
  bool last_accepted = false;
  int last_gmres_iters = 0;
  // LevenbergMarquardt: damping cache (λ, λ_factor, DᵀD), geodesic acceleration (v, a), LM trust region (v_cache, ‖v_old‖)
  double lm_lam = 0, lm_lam_factor = 0, lm_norm_v_old = 0, lm_beta = 0;
  bool lm_tr_accepted = false, lm_geo_accepted = false;
  double *lm_dtd = nullptr, *lm_diag = nullptr, *lm_v = nullptr, *lm_a = nullptr, *lm_vcache = nullptr, *lm_rhs = nullptr;
  nk_normal_plan *nplan = nullptr;  // LevenbergMarquardt with the direct linsolve: the assembled JᵀJ + λDᵀD
  // PseudoTransient: SwitchedEvolutionRelaxation cache (α⁻¹, the residual norm it was last scaled with) and the shift that
  // currently sits on the diagonal of the concrete J
  double pt_ainv = 0, pt_res = 0, pt_applied = 0;
  double *pt_mass = nullptr;  // diagonal of the mass matrix M (NULL: identity): the damping is α⁻¹ M
  std::vector<nk_trace_entry> trace;
  // preconditioning behind the `precs` hook: a built-in object on the concrete J (nk_options.precond_kind) and / or a callback
  nk_precond *prec_obj = nullptr;
  // speculative Jacobian fill (plain Newton on a built-in problem with a concrete J): the NEXT step's J(u_new) is written into a
  // second value set while the host waits for this step's norms — the queue is not empty during the termination test's round
  // trip. The live J is untouched until the next step takes the set (refresh_J); a solve that terminates never sees it.
  bool spec_valid = false;
  bool begun_ahead = false;     // … and the next linear solve's cycle begin rode in that fill's first workgroup (speculate_J)
  double *d_rhs_gersh = nullptr;   // the residual kernel's Gershgorin partials of J(u_new) (what that begin reduces)
  uint64_t spec_version = 0, spec_params = 0;
  // the step's last residual kernel also wrote f into column 0 of the Krylov basis and its Σ f² partials into d_rhs_ss: the next
  // step's linear solve starts from them if nothing has moved since (nk_gmres_preloaded_rhs)
  double *d_rhs_ss = nullptr;
  bool pre_valid = false;
  uint64_t pre_uver = 0, pre_params = 0;
  const double *pre_fu = nullptr;
  int pre_grid = 0;
  int rhs_gersh_grid = 0;      // > 0: d_rhs_gersh holds the Gershgorin partials of J at the iterate the residual was taken at
  const double *spec_u = nullptr;               // the iterate the set was filled at
  nk_csr_valstate spec_state{};                 // the filled set (valid) / the spare buffers (not valid)
  nk_precs_fn precs = nullptr;
  void *precs_user = nullptr;
};

// u_new = u + sign·du (out of place: the old iterate stays intact in its buffer) ; partial Σ (u_new − u_old)²  (the stall
// test's ‖u − uprev‖₂, termination_conditions.jl:311-316). sign = −1 takes the linear solve's x as it is (δu = −x).
__global__ __launch_bounds__(NK_BLOCK) void k_newton_update(int64_t n, double sign, const double *__restrict__ du,
                                                            const double *__restrict__ u, double *__restrict__ unew,
                                                            double *__restrict__ partials) {
  __shared__ double sm[4];
  double ss = 0.0;
  const int64_t stride = (int64_t)gridDim.x * NK_BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x; i < n; i += stride) {
    const double uo = u[i], un = uo + sign * du[i], d = un - uo;
    unew[i] = un;
    ss += d * d;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = ss;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}
__global__ __launch_bounds__(NK_BLOCK) void k_sum_partials(const double *__restrict__ partials, int nblk,
                                                           double *__restrict__ out) {
  __shared__ double sm[4];
  double v = 0.0;
  for (int i = threadIdx.x; i < nblk; i += NK_BLOCK) v += partials[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) *out = sm[0] + sm[1] + sm[2] + sm[3];
}

// one pass over (fu, u): Σ(fu+u)², max|fu+u|, max(|fu| − reltol·|u+fu|)  — the quantities the Norm/Rel* termination
// modes need (check_convergence, termination_conditions.jl:338-376; apply_norm(f, du, u) = f(du .+ u), utils.jl:102)
__global__ __launch_bounds__(NK_BLOCK) void k_tc_pair(int64_t n, const double *__restrict__ fu,
                                                      const double *__restrict__ u, double reltol,
                                                      double *__restrict__ partials) {
  __shared__ double sm[12];
  double ss = 0.0, mx = 0.0, viol = -__builtin_inf();
  const int64_t stride = (int64_t)gridDim.x * NK_BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x; i < n; i += stride) {
    const double f = fu[i], s = f + u[i], as = fabs(s);
    ss += s * s;
    mx = (as > mx || as != as) ? as : mx;
    const double vv = fabs(f) - reltol * as;
    viol = (vv > viol || vv != vv) ? vv : viol;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    ss += __shfl_xor(ss, o, 64);
    const double m2 = __shfl_xor(mx, o, 64), v2 = __shfl_xor(viol, o, 64);
    mx = (m2 > mx || m2 != m2) ? m2 : mx;
    viol = (v2 > viol || v2 != v2) ? v2 : viol;
  }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { sm[w] = ss; sm[4 + w] = mx; sm[8 + w] = viol; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = sm[0] + sm[1] + sm[2] + sm[3], b = sm[4], c = sm[8];
    for (int k = 1; k < 4; ++k) {
      b = (sm[4 + k] > b || sm[4 + k] != sm[4 + k]) ? sm[4 + k] : b;
      c = (sm[8 + k] > c || sm[8 + k] != sm[8 + k]) ? sm[8 + k] : c;
    }
    partials[blockIdx.x] = a;
    partials[gridDim.x + blockIdx.x] = b;
    partials[2 * gridDim.x + blockIdx.x] = c;
  }
}
__global__ __launch_bounds__(NK_BLOCK) void k_tc_pair_reduce(const double *__restrict__ partials, int nblk,
                                                             double *__restrict__ out) {
  __shared__ double sm[12];
  double ss = 0.0, mx = 0.0, viol = -__builtin_inf();
  for (int i = threadIdx.x; i < nblk; i += NK_BLOCK) {
    ss += partials[i];
    const double b = partials[nblk + i], c = partials[2 * nblk + i];
    mx = (b > mx || b != b) ? b : mx;
    viol = (c > viol || c != c) ? c : viol;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    ss += __shfl_xor(ss, o, 64);
    const double m2 = __shfl_xor(mx, o, 64), v2 = __shfl_xor(viol, o, 64);
    mx = (m2 > mx || m2 != m2) ? m2 : mx;
    viol = (v2 > viol || v2 != v2) ? v2 : viol;
  }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { sm[w] = ss; sm[4 + w] = mx; sm[8 + w] = viol; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = sm[0] + sm[1] + sm[2] + sm[3], b = sm[4], c = sm[8];
    for (int k = 1; k < 4; ++k) {
      b = (sm[4 + k] > b || sm[4 + k] != sm[4 + k]) ? sm[4 + k] : b;
      c = (sm[8 + k] > c || sm[8 + k] != sm[8 + k]) ? sm[8 + k] : c;
    }
    out[0] = a; out[1] = b; out[2] = c;
  }
}

extern "C" int nk_options_default(nk_options *o) {
  NK_REQUIRE(o, "NULL argument");
  memset(o, 0, sizeof(*o));
  o->algorithm = NK_ALG_NEWTON_RAPHSON;
  o->linsolve = NK_LINSOLVE_GMRES_MATFREE;
  o->maxiters = 1000;
  o->abstol = 0.0;
  o->reltol = 0.0;
  o->maxtime = 0.0;
  o->gmres_restart = 30;
  o->gmres_maxiters = 300;
  // s-step Arnoldi (Newton basis where the operator's spectrum can be bounded, else monomial): three sweeps over the basis per
  // block of columns; a block that loses rank finishes that solve with delayed CGS2, and restarts the s-step form cannot serve
  // (m > 78) run column by column (NK_ORTHO_DCGS2: CGS2 arithmetic, two sweeps and one reduction per column)
  o->gmres_ortho = NK_ORTHO_SSTEP;
  o->gmres_fixed_iters = 0;
  o->lin_abstol = -1.0;
  o->lin_reltol = -1.0;
  o->forcing = NK_FORCING_NONE;
  o->ew_safeguard = 1;
  o->ew_eta0 = 0.5;
  o->ew_eta_max = 0.9;
  o->ew_gamma = 0.9;
  o->ew_alpha = 2.0;
  o->ew_safeguard_threshold = 0.1;
  o->radius_update_scheme = NK_RUS_SIMPLE;
  o->max_shrink_times = 32;
  o->step_threshold = 1.0 / 10000;  // TrustRegion() constructor values (trust_region.jl:25-33)
  o->shrink_threshold = 0.25;
  o->expand_threshold = 0.75;
  o->shrink_factor = 0.25;
  o->expand_factor = 2.0;
  o->patience_steps = 100;
  o->max_stalled_steps = 32;
  o->patience_objective_multiplier = 3.0;
  o->min_max_factor = 1.3;
  o->protective_threshold = 0.0;
  o->store_trace = 0;
  o->linesearch = 0;
  o->ls_c1 = 1e-4;
  o->ls_rho_hi = 0.5;
  o->ls_rho_lo = 0.1;
  o->ls_order = 3;
  o->ls_maxiters = 1000;
  o->mg_nu = 0;
  o->mg_coarse = 0;
  o->jac_colored = 0;
  o->lm_disable_geodesic = 0;  // LevenbergMarquardt() constructor values (levenberg_marquardt.jl:37-43)
  o->lm_damping_initial = 1.0;
  o->lm_damping_increase_factor = 2.0;
  o->lm_damping_decrease_factor = 3.0;
  o->lm_min_damping_D = 1e-8;
  o->lm_alpha_geodesic = 0.75;
  o->lm_finite_diff_step_geodesic = 0.1;
  o->lm_b_uphill = 1.0;
  o->pt_alpha_initial = 1e-3;  // PseudoTransient() (pseudo_transient.jl:38)
  o->gmres_sstep = 0;
  o->gmres_sstep_basis = NK_SS_BASIS_AUTO;
  o->precond_kind = 0;
  o->precond_side = NK_SIDE_LEFT;
  return NK_OK;
}

static const double DEFAULT_TOL = 3.0e-13;  // common_defaults.jl:44-48

static bool is_tr(const nk_solver *S) { return S->o.algorithm == NK_ALG_TRUST_REGION; }
// GaussNewton: normal form only when the linear solver needs a square A (Krylov); a factorising solver takes J δ = f as it is
// (needs_square_A(nothing) = false, descent/newton.jl:71, linear_solve.jl:225)
static bool normal_form(const nk_solver *S) {
  return S->o.algorithm == NK_ALG_GAUSS_NEWTON && S->o.linsolve != NK_LINSOLVE_BANDED_LU;
}
static bool is_lm(const nk_solver *S) { return S->o.algorithm == NK_ALG_LEVENBERG_MARQUARDT; }
static bool is_pt(const nk_solver *S) { return S->o.algorithm == NK_ALG_PSEUDO_TRANSIENT; }
static bool concrete(const nk_solver *S) { return S->o.linsolve != NK_LINSOLVE_GMRES_MATFREE; }
static bool direct(const nk_solver *S) { return S->o.linsolve == NK_LINSOLVE_BANDED_LU; }

// ---- scalar helpers (device reductions → pinned host, one synchronisation)
static int fetch(nk_solver *S, int count, double *out, const std::function<int()> &before_wait = nullptr) {
  return nk_scalars_to_host(S->ctx, S->ctx->d_scal, count, out, before_wait);
}
static double *slot(nk_solver *S, int i) { return S->ctx->d_scal + i; }

// a pool buffer that holds neither the current nor the retained best iterate
static double *spare_u(nk_solver *S) {
  for (double *b : S->ubuf)
    if (b != S->u && b != S->best_u) return b;
  return nullptr;  // unreachable: three buffers, at most two in use
}
// ‖fu‖∞ and ‖fu‖₂² of the current residual (+ optionally the stall norm's partial sums) → ONE stage-2 launch, one fetch
static int residual_norms(nk_solver *S, const double *stall_partials, int stall_n, double *step_norm,
                          const std::function<int()> &before_wait = nullptr, int have_partials = 0,
                          const std::function<int(const nk_fold_norms &, bool *)> &fold_into = nullptr) {
  double v[3] = {0, 0, 0};
  NK_TRY(nk_blas_norms_inf2_to_host(S->ctx, S->n, S->fu, slot(S, 0), stall_partials, stall_n, v, before_wait, have_partials,
                                    fold_into));
  S->fnorm_inf = v[0];
  S->fnorm2 = sqrt(v[1]);
  if (step_norm) *step_norm = sqrt(v[2]);
  return NK_OK;
}

// jac_cache(u) for a concrete J: closed-form values, or — jac_colored — the colour-compressed assembly
static bool speculation_allowed(const nk_solver *S) {
  static const bool off = (getenv("NK_SPECULATIVE_JAC") && atoi(getenv("NK_SPECULATIVE_JAC")) == 0) || getenv("NK_GMRES_GRAPH");
  // (TrustRegion — the fill at the trial point before the host knows whether it is accepted — was measured on config C5: no
  //  gain, the set is wasted on every rejected step; not enabled)
  return !off && S->o.algorithm == NK_ALG_NEWTON_RAPHSON && !S->o.linesearch && S->o.linsolve != NK_LINSOLVE_GMRES_MATFREE &&
         S->P->kind != NK_PROBLEM_USER && !S->o.jac_colored && S->precs == nullptr && S->J != nullptr && !S->J->raw_exposed;
}
// J(u) of the iterate just formed into the spare value set (enqueued behind the step's last kernels, before the host waits)
// (`version`: the iterate's version at which the set may be taken)
// (fold / folded: the stage-2 reduction of the step's norms rides in the fill kernel's first workgroup — nk_fold_norms)
// The next step's linear solve may have its cycle begin run in that fill's first workgroup as well (nk_gmres_begin_ahead): one
// launch less on the step's critical path. Only where nothing the host learns in between can change the begin: no forcing (the
// tolerance of the next solve is this one's), no preconditioner that is rebuilt per Jacobian, fixed work, not the last step.
static bool begin_ahead_allowed(const nk_solver *S) {
  return speculation_allowed(S) && S->o.forcing != NK_FORCING_EISENSTAT_WALKER2 && S->o.cheb_degree <= 0 && S->o.mg_nu <= 0 &&
         !S->o.precond_kind && !S->o.store_trace && S->G != nullptr && S->G->op_kind == 1 && S->G->A == S->J &&
         S->o.gmres_fixed_iters > 0 && S->nsteps + 2 <= S->o.maxiters && S->pre_valid && S->rhs_gersh_grid > 0;
}
static int speculate_J(nk_solver *S, const double *u_at, uint64_t version, const nk_fold_norms *fold = nullptr, bool *folded = nullptr) {
  S->spec_valid = false;
  S->begun_ahead = false;
  if (folded) *folded = false;
  if (!speculation_allowed(S)) return NK_OK;
  nk_csr *J = S->J;
  if (!S->spec_state.d_val) NK_TRY(nk_csr_alloc_values(J, &S->spec_state.d_val));
  const nk_csr_valstate live = nk_csr_get_valstate(J);
  nk_csr_valstate spare = S->spec_state;
  spare.t_values_stale = true; spare.bounds_valid = false; spare.bounds_pending = false;
  nk_csr_set_valstate(J, spare);
  nk_ss_begin_args beg{};
  bool begin_rides = false;
  if (fold != nullptr && begin_ahead_allowed(S)) {
    const int brc = nk_gmres_begin_ahead(S->G, S->fu, S->d_rhs_ss, S->pre_grid, S->d_rhs_gersh, S->rhs_gersh_grid, S->lin_abstol,
                                         S->lin_reltol, S->o.gmres_maxiters, S->o.gmres_fixed_iters, &beg, &begin_rides);
    if (brc != NK_OK) { nk_csr_set_valstate(J, live); return brc; }
  }
  const int rc = nk_problem_jac_values_dev(S->P, u_at, J, fold, folded, begin_rides ? &beg : nullptr);
  if (begin_rides) {
    if (rc == NK_OK && folded && *folded) {
      nk_gmres_ahead_values(S->G, nk_csr_get_valstate(J).d_val);
      S->begun_ahead = true;
    } else {
      nk_gmres_drop_ahead(S->G);
    }
  }
  S->spec_state = nk_csr_get_valstate(J);       // (the fill may have grown the partials buffer)
  nk_csr_set_valstate(J, live);
  if (rc != NK_OK) return rc;
  S->spec_valid = true;
  S->spec_version = version;
  S->spec_u = u_at;
  S->spec_params = S->P->params_version;
  return NK_OK;
}
// a begin that ran ahead for a solve that will not take it: forgotten — and, because it reduced the SPARE value set's bounds into
// the matrix's bounds word, the live set's bounds are recomputed on demand
static void drop_begun_ahead(nk_solver *S) {
  if (!S->begun_ahead) return;
  S->begun_ahead = false;
  nk_gmres_drop_ahead(S->G);
  if (S->J) nk_csr_invalidate_bounds(S->J);
  S->spec_valid = false;
}
static int refresh_J(nk_solver *S) {
  if (S->spec_valid && S->spec_version == S->u_version && S->spec_u == S->u && S->spec_params == S->P->params_version &&
      speculation_allowed(S)) {
    // the values of J(u) are already there: the two sets change places (the old live buffers are the next spare ones)
    const nk_csr_valstate live = nk_csr_get_valstate(S->J);
    nk_csr_set_valstate(S->J, S->spec_state);
    S->spec_state = live;
    S->spec_valid = false;
    S->begun_ahead = false;   // (nk_gmres_solve_dev skips its begin — or starts from scratch if anything about the solve differs)
  } else {
    drop_begun_ahead(S);
    S->spec_valid = false;
    if (S->o.jac_colored && S->P->kind != NK_PROBLEM_USER) NK_TRY(nk_problem_jac_colored_dev(S->P, S->u, S->J));
    else NK_TRY(nk_problem_jac_values_dev(S->P, S->u, S->J));
  }
  S->stats.njacs++;
  S->lu_valid = false;
  S->pt_applied = 0.0;  // fresh values: no damping on the diagonal yet
  return NK_OK;
}
static int apply_J(nk_solver *S, const double *v, double *out) {
  if (concrete(S)) return nk_csr_spmv_dev(S->J, v, out, nullptr);
  return nk_problem_jvp_dev(S->P, S->u, v, out, nullptr);
}
static int apply_JT(nk_solver *S, const double *u_at, const double *v, double *out) {
  if (concrete(S) && u_at == S->u) return nk_csr_spmv_t_dev(S->J, v, out);
  return nk_problem_vjp_dev(S->P, u_at, v, out);
}

// ---- trust-region defaults (trust_region.jl:320-384)
static void tr_defaults(nk_solver *S) {
  const int m = S->o.radius_update_scheme;
  auto pick = [](double v, double d) { return v == 0.0 ? d : v; };
  double d;
  d = (m == NK_RUS_HEI) ? 0.0 : (m == NK_RUS_YUAN) ? 1e-3 : (m == NK_RUS_BASTIN) ? 0.05 : 1e-4;
  S->step_thr = pick(S->o.step_threshold, d);
  d = (m == NK_RUS_HEI) ? 0.0 : (m == NK_RUS_NLSOLVE || m == NK_RUS_BASTIN) ? 0.05 : 0.25;
  S->shrink_thr = pick(S->o.shrink_threshold, d);
  d = (m == NK_RUS_NLSOLVE || m == NK_RUS_BASTIN) ? 0.9 : (m == NK_RUS_HEI) ? 0.0 : 0.75;
  S->expand_thr = pick(S->o.expand_threshold, d);
  d = (m == NK_RUS_NLSOLVE) ? 0.5 : (m == NK_RUS_HEI) ? 0.0 : (m == NK_RUS_BASTIN) ? 0.05 : 0.25;
  S->shrink_f = pick(S->o.shrink_factor, d);
  S->expand_f = pick(S->o.expand_factor, 2.0);
  S->p1 = S->p2 = S->p3 = S->p4 = 0.0;
  switch (m) {
    case NK_RUS_NLSOLVE: S->p1 = 0.5; break;
    case NK_RUS_HEI: S->p1 = 5.0; S->p2 = 0.1; S->p3 = 0.15; S->p4 = 0.15; break;
    case NK_RUS_YUAN: S->p1 = 2.0; S->p2 = 1.0 / 6; S->p3 = 6.0; break;
    case NK_RUS_FAN: S->p1 = 0.1; S->p2 = 0.25; S->p3 = 12.0; S->p4 = 1e18; break;
    case NK_RUS_BASTIN: S->p1 = 2.5; S->p2 = 0.25; break;
    default: break;
  }
}
static int tr_radii(nk_solver *S) {
  const int m = S->o.radius_update_scheme;
  nk_ctx *ctx = S->ctx;
  NK_TRY(nk_blas_sumsq(ctx, S->n, S->u, slot(S, 0)));
  NK_TRY(nk_blas_sumsq(ctx, S->n, S->fu, slot(S, 1)));
  NK_TRY(nk_blas_minmax(ctx, S->n, S->u, slot(S, 2)));
  double v[4];
  NK_TRY(fetch(S, 4, v));
  const double u0_norm = sqrt(v[0]), fu_norm = sqrt(v[1]), umax = v[2], umin = -v[3];
  if (S->o.max_trust_radius != 0.0) S->max_tr = S->o.max_trust_radius;
  else if (m == NK_RUS_SIMPLE || m == NK_RUS_NOCEDAL_WRIGHT) S->max_tr = fmax(fu_norm, umax - umin);
  else S->max_tr = INFINITY;
  if (S->o.initial_trust_radius != 0.0) S->init_tr = S->o.initial_trust_radius;
  else if (m == NK_RUS_NLSOLVE) S->init_tr = u0_norm > 0 ? u0_norm : 1.0;
  else if (m == NK_RUS_HEI || m == NK_RUS_BASTIN) S->init_tr = 1.0;
  else if (m == NK_RUS_FAN) S->init_tr = pow(fu_norm, 0.99) / 10.0;
  else S->init_tr = S->max_tr / 11.0;
  if (m == NK_RUS_YUAN) {
    NK_TRY(apply_JT(S, S->u, S->fu, S->JTfu));
    NK_TRY(nk_blas_sumsq(ctx, S->n, S->JTfu, slot(S, 0)));
    NK_TRY(fetch(S, 1, v));
    S->init_tr = S->p1 * sqrt(v[0]);
  }
  return NK_OK;
}

// ---- termination cache: the nine SciMLBase termination modes (termination_conditions.jl:243-376)
enum {
  TM_ABSNORM_SAFEBEST = 0,  // default_termination_mode(::NonlinearProblem, Val(:regular))  (:385-389)
  TM_NORM = 1, TM_REL = 2, TM_RELNORM = 3, TM_RELNORM_SAFE = 4, TM_RELNORM_SAFEBEST = 5,
  TM_ABS = 6, TM_ABSNORM = 7, TM_ABSNORM_SAFE = 8
};
static bool tm_safe(int m) { return m == TM_ABSNORM_SAFEBEST || m == TM_ABSNORM_SAFE || m == TM_RELNORM_SAFE || m == TM_RELNORM_SAFEBEST; }
static bool tm_best(int m) { return m == TM_ABSNORM_SAFEBEST || m == TM_RELNORM_SAFEBEST; }
static bool tm_rel(int m) { return m == TM_RELNORM_SAFE || m == TM_RELNORM_SAFEBEST; }
static bool tm_needs_pair(int m) { return m == TM_NORM || m == TM_REL || m == TM_RELNORM || tm_rel(m); }

struct tc_quant {
  double nf = 0;      // internalnorm(fu)
  double nfu = 0;     // internalnorm(fu .+ u)
  double relviol = 0; // max_i(|fu_i| − reltol |u_i + fu_i|)  (RelTerminationMode: converged iff ≤ 0)
};
// device reductions for the current (fu, u); ‖fu‖∞ is already in S->fnorm_inf
static int tc_quantities(nk_solver *S, tc_quant *q) {
  nk_ctx *ctx = S->ctx;
  const int mode = S->o.termination_mode;
  const bool l2 = S->o.termination_norm == 1;
  q->nf = S->fnorm_inf;
  int cnt = 0;
  if (l2) { NK_TRY(nk_blas_sumsq(ctx, S->n, S->fu, slot(S, 8))); cnt = 1; }
  if (tm_needs_pair(mode)) {
    const int grid = nk_grid_for(S->n, NK_BLOCK * 4, NK_MAX_RED_BLOCKS);
    NK_LAUNCH(ctx, k_tc_pair, dim3(grid), dim3(NK_BLOCK), S->n, S->fu, S->u, S->reltol, ctx->d_partials);
    NK_LAUNCH(ctx, k_tc_pair_reduce, dim3(1), dim3(NK_BLOCK), (const double *)ctx->d_partials, grid, slot(S, 9));
    NK_HIP(hipGetLastError());
    NK_TRY(nk_comm_allreduce(ctx, slot(S, 9), 1, 0));
    NK_TRY(nk_comm_allreduce(ctx, slot(S, 10), 2, 1));
    cnt = 4;
  }
  if (cnt) {
    double v[4] = {0, 0, 0, 0};
    NK_TRY(nk_scalars_to_host(ctx, slot(S, 8), cnt, v));
    if (l2) q->nf = sqrt(v[0]);
    if (cnt == 4) { q->nfu = l2 ? sqrt(v[1]) : v[2]; q->relviol = v[3]; }
  }
  return NK_OK;
}
static double tc_objective(const nk_solver *S, const tc_quant &q) {
  if (tm_rel(S->o.termination_mode)) return q.nf / (q.nfu + 2.220446049250313e-16 * S->reltol);  // eps(reltol)
  return q.nf;
}

static int tc_reinit(nk_solver *S) {
  S->tc_retcode = NK_RET_DEFAULT;
  S->tc_nsteps = 0;
  tc_quant q;
  NK_TRY(tc_quantities(S, &q));
  S->initial_obj = tm_safe(S->o.termination_mode) ? tc_objective(S, q) : INFINITY;
  S->best_obj = S->initial_obj;
  S->objectives_trace.assign(S->o.patience_steps > 0 ? S->o.patience_steps : 1, 0.0);
  if (S->o.max_stalled_steps >= 0) S->step_norm_trace.assign(S->o.max_stalled_steps > 0 ? S->o.max_stalled_steps : 1, 0.0);
  else S->step_norm_trace.clear();
  if (tm_rel(S->o.termination_mode) && !S->step_norm_trace.empty()) {
    NK_TRY(nk_blas_sumsq(S->ctx, S->n, S->u, slot(S, 0)));
    double v;
    NK_TRY(fetch(S, 1, &v));
    S->tc_u0_norm = sqrt(v);
  }
  return NK_OK;
}
// sets *stop when the solve must end; step_norm = ‖u − uprev‖₂
static int tc_check(nk_solver *S, double step_norm, bool *stop) {
  *stop = false;
  const int mode = S->o.termination_mode;
  tc_quant q;
  NK_TRY(tc_quantities(S, &q));
  if (!tm_safe(mode)) {  // plain modes: check_convergence only (termination_conditions.jl:232-241)
    bool conv = false;
    switch (mode) {
      case TM_NORM: conv = (q.nf <= S->abstol) || (q.nf <= S->reltol * q.nfu); break;
      case TM_REL: conv = (q.relviol <= 0.0); break;
      case TM_RELNORM: conv = (q.nf <= S->reltol * q.nfu); break;
      case TM_ABS: conv = (S->fnorm_inf <= S->abstol); break;
      case TM_ABSNORM: conv = (q.nf <= S->abstol); break;
      default: break;
    }
    if (conv) { S->tc_retcode = NK_RET_SUCCESS; *stop = true; }
    return NK_OK;
  }
  const double objective = tc_objective(S, q);
  const double criteria = tm_rel(mode) ? S->reltol : S->abstol;
  if (!isfinite(objective)) { S->tc_retcode = NK_RET_UNSTABLE; *stop = true; return NK_OK; }
  if (S->o.protective_threshold > 0.0 &&
      objective > S->initial_obj * S->o.protective_threshold * (double)S->P->n_global) {
    S->tc_retcode = NK_RET_UNSTABLE; *stop = true; return NK_OK;
  }
  if (tm_best(mode) && objective < S->best_obj) {
    S->best_obj = objective;
    S->best_u = S->u;  // the buffer stays untouched until a better iterate replaces it (spare_u never hands it out)
  }
  if (objective <= criteria) { S->tc_retcode = NK_RET_SUCCESS; *stop = true; return NK_OK; }
  S->tc_nsteps += 1;
  const int L = (int)S->objectives_trace.size();
  S->objectives_trace[(S->tc_nsteps - 1) % L] = objective;
  if (objective <= S->o.patience_objective_multiplier * criteria && S->tc_nsteps > S->o.patience_steps) {
    const int cnt = S->tc_nsteps < L ? S->tc_nsteps : L;
    double mn = INFINITY, mx = -INFINITY;
    for (int i = 0; i < cnt; ++i) { mn = fmin(mn, S->objectives_trace[i]); mx = fmax(mx, S->objectives_trace[i]); }
    if (mn < S->o.min_max_factor * mx) { S->tc_retcode = NK_RET_STALLED; *stop = true; return NK_OK; }
  }
  if (!S->step_norm_trace.empty()) {
    const int L2 = (int)S->step_norm_trace.size();
    S->step_norm_trace[(S->tc_nsteps - 1) % L2] = step_norm;
    if (S->tc_nsteps > S->o.max_stalled_steps) {
      double mx = -INFINITY;
      for (double v : S->step_norm_trace) mx = fmax(mx, v);
      const bool stalled = tm_rel(mode) ? (mx <= S->reltol * (mx + S->tc_u0_norm)) : (mx <= S->abstol);
      if (stalled) { S->tc_retcode = NK_RET_STALLED; *stop = true; return NK_OK; }
    }
  }
  S->tc_retcode = NK_RET_FAILURE;
  return NK_OK;
}
// update_from_termination_cache! (termination_conditions.jl:440-453)
static int rollback_to_best(nk_solver *S) {
  if (!tm_best(S->o.termination_mode)) return NK_OK;  // only the *Best modes retain an iterate
  if (S->best_u == S->u) return NK_OK;  // the last iterate is the best one: its residual is already in fu
  S->u = S->best_u;
  S->u_version++;
  nk_problem_invalidate(S->P);
  NK_TRY(nk_problem_residual_dev(S->P, S->u, S->fu));
  S->stats.nf++;
  return residual_norms(S, nullptr, 0, nullptr);
}

// ---- init
static int solver_start(nk_solver *S, bool first = true) {  // everything after u has been set (first = init, else reinit!)
  nk_ctx *ctx = S->ctx;
  // the problem's linearisation caches (exp(u) diagonal, f(u) for forward differences) are keyed on the pointer of u:
  // u changes in place between solves, and a new solver may receive a just-freed address — start from "not linearised"
  nk_problem_invalidate(S->P);
  NK_TRY(nk_problem_residual_dev(S->P, S->u, S->fu));
  S->stats = nk_stats{};
  S->ctx_base.op_applies = S->ctx->stats.op_applies;
  S->ctx_base.allreduces = S->ctx->stats.allreduces;
  S->ctx_base.halo_exchanges = S->ctx->stats.halo_exchanges;
  S->nsteps = 0;
  S->retcode = NK_RET_DEFAULT;
  S->force_stop = false;
  S->make_new_jacobian = true;
  S->total_time = 0.0;
  S->trace.clear();
  S->u_version++;
  S->best_u = S->u;
  S->fu_deferred = false;
  NK_TRY(residual_norms(S, nullptr, 0, nullptr));
  NK_TRY(tc_reinit(S));
  // jacobian.jl:104-118 evaluates J once at init (njacs = 1 before the first step); reinit! does not (JacobianCache's reinit!
  // only swaps p, jacobian.jl:184-186) — the first step re-evaluates it either way (make_new_jacobian)
  if (concrete(S) && first) NK_TRY(refresh_J(S));
  NK_TRY(nk_blas_fill(ctx, S->n, 0.0, S->du));  // descent/newton.jl:34-36
  S->eta = S->o.ew_eta0;
  S->rnorm = S->rnorm_prev = S->fnorm2;
  S->lin_abstol = S->o.lin_abstol >= 0.0 ? S->o.lin_abstol : S->abstol;  // FirstOrder/src/solve.jl:203
  S->lin_reltol = S->o.lin_reltol >= 0.0 ? S->o.lin_reltol : S->reltol;
  if (is_tr(S)) {
    tr_defaults(S);
    NK_TRY(tr_radii(S));
    S->tr = S->init_tr;
    S->rho = 0.0;
    S->shrink_counter = 0;
    S->last_accepted = false;
  }
  if (is_pt(S)) {  // SwitchedEvolutionRelaxationCache init / reinit! (pseudo_transient.jl:107-131)
    S->pt_ainv = 1.0 / S->o.pt_alpha_initial;
    S->pt_res = S->fnorm2;
    if (first) S->pt_applied = 0.0;  // (the Jacobian values were just refilled; after reinit! the first step's refill resets it)
    if (S->G) NK_TRY(nk_gmres_set_shift(S->G, 0.0));
  }
  if (is_lm(S)) {  // init / reinit! of the damping cache, the LM trust region and the geodesic cache
    S->lm_lam = S->o.lm_damping_initial;                    // levenberg_marquardt.jl:72-89,119-131
    S->lm_lam_factor = S->o.lm_damping_increase_factor;
    NK_TRY(nk_blas_fill(ctx, S->n, S->o.lm_min_damping_D, S->lm_dtd));
    NK_TRY(nk_blas_copy(ctx, S->n, S->u, S->lm_vcache));    // `@bb v = copy(u)` (:212), reinit!: copyto!(v_cache, u0)
    S->lm_norm_v_old = INFINITY;
    S->lm_tr_accepted = false;
    S->lm_geo_accepted = false;
    S->lm_beta = NAN;
    S->tr = S->lm_lam;  // what nk_solver_get_scalars reports in the trust-region slot
  }
  return NK_OK;
}

extern "C" int nk_solver_init(nk_problem *P, const double *u0, int memspace, const nk_options *opts, nk_solver **out) {
  NK_REQUIRE(P && u0 && opts && out, "NULL argument");
  nk_ctx *ctx = P->ctx;
  NK_HIP(hipSetDevice(ctx->device));
  NK_REQUIRE(opts->algorithm == NK_ALG_NEWTON_RAPHSON || opts->algorithm == NK_ALG_TRUST_REGION ||
                 opts->algorithm == NK_ALG_GAUSS_NEWTON || opts->algorithm == NK_ALG_LEVENBERG_MARQUARDT ||
                 opts->algorithm == NK_ALG_PSEUDO_TRANSIENT, "bad algorithm");
  NK_REQUIRE(!(opts->algorithm == NK_ALG_PSEUDO_TRANSIENT && !(opts->pt_alpha_initial > 0.0)),
             "PseudoTransient: alpha_initial must be positive");
  NK_REQUIRE(!(opts->algorithm == NK_ALG_PSEUDO_TRANSIENT && opts->forcing != NK_FORCING_NONE),
             "PseudoTransient takes no forcing term (pseudo_transient.jl:37-57)");
  if (opts->algorithm == NK_ALG_LEVENBERG_MARQUARDT) {
    NK_REQUIRE(opts->linsolve == NK_LINSOLVE_GMRES_CSR || opts->linsolve == NK_LINSOLVE_BANDED_LU,
               "LevenbergMarquardt needs a concrete Jacobian (concrete_jac = Val(true), levenberg_marquardt.jl:62)");
    NK_REQUIRE(opts->linesearch == 0 && opts->forcing == NK_FORCING_NONE,
               "LevenbergMarquardt takes neither a line search nor a forcing term (levenberg_marquardt.jl:37-64)");
    NK_REQUIRE(opts->lm_damping_initial > 0.0 && opts->lm_damping_increase_factor > 0.0 &&
                   opts->lm_damping_decrease_factor > 0.0 && opts->lm_finite_diff_step_geodesic > 0.0,
               "LevenbergMarquardt: damping_initial, the damping factors and finite_diff_step_geodesic must be positive");
  }
  NK_REQUIRE(opts->linsolve == NK_LINSOLVE_GMRES_MATFREE || opts->linsolve == NK_LINSOLVE_GMRES_CSR ||
                 opts->linsolve == NK_LINSOLVE_BANDED_LU,
             "unknown linsolve %d", opts->linsolve);
  NK_REQUIRE(!(opts->linsolve == NK_LINSOLVE_BANDED_LU && opts->forcing != NK_FORCING_NONE),
             "a forcing term needs an iterative linear solver");
  NK_REQUIRE(opts->termination_mode >= 0 && opts->termination_mode <= 8, "bad termination_mode %d", opts->termination_mode);
  NK_REQUIRE(opts->termination_norm == 0 || opts->termination_norm == 1, "bad termination_norm %d", opts->termination_norm);
  NK_REQUIRE(opts->linesearch >= 0 && opts->linesearch <= 5, "bad linesearch %d", opts->linesearch);
  NK_REQUIRE(!(opts->algorithm == NK_ALG_TRUST_REGION && opts->linesearch != 0),
             "TrustRegion and LineSearch methods are algorithmically incompatible (FirstOrder/src/solve.jl:221-223)");
  NK_REQUIRE(!(opts->algorithm == NK_ALG_TRUST_REGION && opts->forcing != NK_FORCING_NONE),
             "TrustRegion does not accept a forcing term (trust_region.jl:25-43)");
  nk_solver *S = new nk_solver();
  S->P = P;
  S->ctx = ctx;
  auto guard = nk_make_guard(S, [](nk_solver *s) { nk_solver_destroy(s); });
  S->o = *opts;
  if (S->o.maxiters <= 0) S->o.maxiters = 1000;
  if (S->o.gmres_restart <= 0) S->o.gmres_restart = 30;
  if (S->o.gmres_maxiters <= 0) S->o.gmres_maxiters = 300;
  if (S->o.patience_steps <= 0) S->o.patience_steps = 100;
  if (S->o.max_shrink_times <= 0) S->o.max_shrink_times = 32;
  S->abstol = opts->abstol > 0.0 ? opts->abstol : DEFAULT_TOL;
  S->reltol = opts->reltol > 0.0 ? opts->reltol : DEFAULT_TOL;
  const int64_t n = S->n = P->n_local;
  const size_t na = (size_t)n + 2;
  for (int b = 0; b < 3; ++b) NK_TRY(nk_dev_alloc(&S->ubuf[b], na));
  S->u = S->best_u = S->ubuf[0];
  NK_TRY(nk_dev_alloc(&S->fu, na));
  NK_TRY(nk_dev_alloc(&S->du, na));
  if (!is_tr(S) && S->o.linesearch) {
    NK_TRY(nk_dev_alloc(&S->fu_trial, na));
    NK_TRY(nk_dev_alloc(&S->Jdu, na));
  }
  if (normal_form(S)) NK_TRY(nk_dev_alloc(&S->JTfu, na));
  if (is_lm(S)) {
    NK_TRY(nk_dev_alloc(&S->fu_trial, na));
    NK_TRY(nk_dev_alloc(&S->Jdu, na));
    for (double **b : {&S->lm_dtd, &S->lm_diag, &S->lm_v, &S->lm_a, &S->lm_vcache, &S->lm_rhs}) NK_TRY(nk_dev_alloc(b, na));
  }
  if (is_tr(S)) {
    NK_TRY(nk_dev_alloc(&S->fu_trial, na));
    NK_TRY(nk_dev_alloc(&S->du_newton, na));
    NK_TRY(nk_dev_alloc(&S->du_cauchy, na));
    NK_TRY(nk_dev_alloc(&S->Jdu, na));
    NK_TRY(nk_dev_alloc(&S->JTfu, na));
    NK_TRY(nk_dev_alloc(&S->c1, na));
    NK_TRY(nk_dev_alloc(&S->c2, na));
  }
  if (concrete(S)) {
    NK_TRY(nk_problem_jac_csr(P, &S->J));
    S->own_J = (P->kind != NK_PROBLEM_USER);
  }
  if (direct(S)) {
    if (is_lm(S)) {  // the factorising solver gets the assembled normal matrix JᵀJ + λDᵀD (see lm_damped_solve)
      NK_TRY(nk_normal_plan_create(S->J, &S->nplan));
      NK_TRY(nk_bandlu_create(nk_normal_plan_matrix(S->nplan), &S->B));
    } else {
      NK_TRY(nk_bandlu_create(S->J, &S->B));
    }
    NK_TRY(nk_dev_alloc(&S->stage, na));
  } else {
    NK_TRY(nk_gmres_create(ctx, n, S->o.gmres_restart, S->o.gmres_ortho, &S->G));
    NK_TRY(nk_gmres_set_block_size(S->G, S->o.gmres_sstep));   // (validates 0..16)
    NK_TRY(nk_gmres_set_sstep_basis(S->G, S->o.gmres_sstep_basis));
    if (concrete(S)) NK_TRY(nk_gmres_set_operator_csr(S->G, S->J));
  }
  NK_HIP(hipMemcpyAsync(S->u, u0, n * sizeof(double),
                        memspace == NK_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream));
  NK_TRY(solver_start(S));
  *out = guard.release();
  return NK_OK;
}

extern "C" int nk_solver_destroy(nk_solver *S) {
  if (!S) return NK_OK;
  hipStreamSynchronize(S->ctx->stream);
  if (S->P) nk_problem_invalidate(S->P);  // the vectors the problem was linearised at are about to be freed
  double *bufs[] = {S->ubuf[0], S->ubuf[1], S->ubuf[2], S->fu, S->du, S->fu_trial, S->du_newton, S->du_cauchy,
                    S->Jdu, S->JTfu, S->c1, S->c2, S->tr_du, S->stage, S->stage2, S->lm_dtd, S->lm_diag, S->lm_v,
                    S->lm_a, S->lm_vcache, S->lm_rhs, S->pt_mass};
  for (double *b : bufs) hipFree(b);
  hipFree(S->d_rhs_ss);
  hipFree(S->d_rhs_gersh);
  hipFree(S->spec_state.d_val);     // (whichever value set is the spare one now; the live one belongs to J)
  hipFree(S->spec_state.d_gersh);
  nk_gmres_destroy(S->G);
  nk_precond_destroy(S->prec_obj);
  nk_bandlu_destroy(S->B);
  nk_normal_plan_destroy(S->nplan);
  if (S->own_J) nk_csr_destroy(S->J);
  delete S;
  return NK_OK;
}

// ---- EisenstatWalkerForcing2 (eisenstat_walker.jl:42-89)
static int pre_step_forcing(nk_solver *S, int iter) {
  const nk_options &o = S->o;
  if (iter == 0) {
    S->eta = o.ew_eta0;
    S->rnorm = S->rnorm_prev = S->fnorm2;  // ‖fu‖₂, reduced together with ‖fu‖∞ when the residual was evaluated
  } else {
    const double eta_prev = S->eta;
    S->eta = o.ew_gamma * pow(S->rnorm / S->rnorm_prev, o.ew_alpha);
    if (o.ew_safeguard) {
      const double eta_sg = o.ew_gamma * pow(eta_prev, o.ew_alpha);
      if (eta_sg > o.ew_safeguard_threshold && eta_sg > S->eta) S->eta = eta_sg;
    }
    S->eta = fmin(fmax(S->eta, 0.0), o.ew_eta_max);
  }
  S->lin_reltol = S->eta;  // LinearSolve.update_tolerances!(lincache; reltol = η)
  return NK_OK;
}

// ---- NewtonDescent.solve! : J δ = fu through GMRES, then δu = −δ  (newton.jl:121-138)
// negate = false leaves x (δu = −x) in du_out: the plain Newton update then runs with sign −1 and saves a pass
static int newton_descent(nk_solver *S, double *du_out, bool *ok, bool new_jacobian, bool negate = true) {
  S->stats.nsolve++;
  if (is_pt(S)) {
    // SwitchedEvolutionRelaxation solve! (pseudo_transient.jl:152-164): α⁻¹ ← α⁻¹·‖f‖₂/‖f_prev‖₂; then
    // dampen_jacobian!! (damped_newton.jl:283-288,349-366): the step is taken on J + α⁻¹ I
    S->pt_ainv *= S->fnorm2 / S->pt_res;
    S->pt_res = S->fnorm2;
    if (concrete(S)) {  // the shift lives on the diagonal of the stored values; a re-used J carries the previous one
      NK_TRY(nk_csr_add_to_diagonal_dev(S->J, S->pt_ainv - S->pt_applied, S->pt_mass));
      S->pt_applied = S->pt_ainv;
      S->lu_valid = false;
    } else {
      NK_TRY(nk_gmres_set_shift(S->G, S->pt_ainv));
    }
  }
  if (direct(S)) {
    // update_A!(cache, ::AbstractFactorization, A, reuse): refactorise unless the caller asked for reuse
    // (ext/NonlinearSolveBaseLinearSolveExt.jl:81-86; reuse_A_if_factorization = !new_jacobian, newton.jl:125)
    bool lu_ok = true;
    if (new_jacobian || !S->lu_valid) {
      int fok = 0;
      NK_TRY(nk_bandlu_factor(S->B, S->J, &fok));
      S->stats.nfactors++;
      S->lu_valid = fok != 0;
      lu_ok = fok != 0;
    }
    S->last_gmres_iters = 0;
    double v[2] = {NAN, 1.0};
    if (lu_ok) {
      NK_TRY(nk_bandlu_solve(S->B, S->fu, du_out));
      // The direct engines (block cyclic reduction, band LU) pivot on the diagonal only, so the solve is verified: ‖J x − b‖₂ ≤ 1e-10 ‖b‖₂, with one step of iterative
      // refinement before giving up on the factorisation.
      for (int pass = 0; pass < 2; ++pass) {
        NK_TRY(nk_csr_spmv_dev(S->J, du_out, S->stage, nullptr));
        NK_TRY(nk_blas_lincomb(S->ctx, S->n, 1.0, S->fu, -1.0, S->stage, S->stage));  // r = b − J x
        const double *xs[2] = {S->stage, S->fu}, *ys[2] = {S->stage, S->fu};
        NK_TRY(nk_blas_multi_reduce(S->ctx, S->n, 2, xs, ys, nullptr, nullptr, 0, 0, slot(S, 0)));
        NK_TRY(fetch(S, 2, v));
        if (!(v[0] == v[0]) || sqrt(v[0]) <= 1e-10 * sqrt(v[1]) + 1e-300 || pass == 1) break;
        if (!S->stage2) NK_TRY(nk_dev_alloc(&S->stage2, (size_t)S->n + 2));
        NK_TRY(nk_bandlu_solve(S->B, S->stage, S->stage2));                             // J dx = r
        NK_TRY(nk_blas_axpby(S->ctx, S->n, 1.0, S->stage2, 1.0, du_out));
      }
      lu_ok = (v[0] == v[0]) && (sqrt(v[0]) <= 1e-8 * sqrt(v[1]) + 1e-300);
      if (!lu_ok && S->B->bcr) {  // the diagonal-pivot factorisation was not accurate enough: once more with row pivoting
        int changed = 0;
        NK_TRY(nk_bcr_set_pivoting(S->B->bcr, 1, &changed));
        if (changed) {
          int fok = 0;
          NK_TRY(nk_bandlu_factor(S->B, S->J, &fok));
          S->stats.nfactors++;
          S->lu_valid = fok != 0;
          if (fok) {
            NK_TRY(nk_bandlu_solve(S->B, S->fu, du_out));
            NK_TRY(nk_csr_spmv_dev(S->J, du_out, S->stage, nullptr));
            NK_TRY(nk_blas_lincomb(S->ctx, S->n, 1.0, S->fu, -1.0, S->stage, S->stage));
            const double *xs[2] = {S->stage, S->fu}, *ys[2] = {S->stage, S->fu};
            NK_TRY(nk_blas_multi_reduce(S->ctx, S->n, 2, xs, ys, nullptr, nullptr, 0, 0, slot(S, 0)));
            NK_TRY(fetch(S, 2, v));
            lu_ok = (v[0] == v[0]) && (sqrt(v[0]) <= 1e-8 * sqrt(v[1]) + 1e-300);
          }
        }
      }
    }
    if (!lu_ok) {
      // The reference's default linear solver falls back (LU → QR) when the factorisation is singular or inaccurate; here
      // the fallback is GMRES on the same concrete J — only if that fails too is the linear solve reported as failed.
      if (!S->G) {
        NK_TRY(nk_gmres_create(S->ctx, S->n, 60, NK_ORTHO_DCGS2, &S->G));
        NK_TRY(nk_gmres_set_operator_csr(S->G, S->J));
      }
      nk_gmres_info fi;
      NK_TRY(nk_gmres_solve_dev(S->G, S->fu, du_out, 0, 0.0, 1e-12, 3000, 0, &fi));
      S->last_gmres_iters = fi.iters;
      S->stats.gmres_iters += fi.iters;
      S->lu_valid = false;
      *ok = fi.converged && !fi.failed;
      if (*ok) {  // a lucky breakdown on a singular J "converges" with a non-finite or meaningless x: trust the true residual only
        NK_TRY(nk_csr_spmv_dev(S->J, du_out, S->stage, nullptr));
        NK_TRY(nk_blas_lincomb(S->ctx, S->n, 1.0, S->fu, -1.0, S->stage, S->stage));
        const double *xs[2] = {S->stage, S->fu}, *ys[2] = {S->stage, S->fu};
        NK_TRY(nk_blas_multi_reduce(S->ctx, S->n, 2, xs, ys, nullptr, nullptr, 0, 0, slot(S, 0)));
        NK_TRY(fetch(S, 2, v));
        *ok = (v[0] == v[0]) && (sqrt(v[0]) <= 1e-8 * sqrt(v[1]) + 1e-300);
      }
      if (!*ok) return NK_OK;
    } else {
      *ok = true;
    }
    if (!negate) return NK_OK;
    return nk_blas_lincomb(S->ctx, S->n, -1.0, du_out, 0.0, du_out, du_out);
  }
  nk_gmres_info info;
  const double *rhs = S->fu;
  if (normal_form(S)) {  // JᵀJ δ = Jᵀ fu (descent/newton.jl:107-118)
    NK_TRY(apply_JT(S, S->u, S->fu, S->JTfu));
    NK_TRY(nk_gmres_set_normal_form(S->G, 1));
    rhs = S->JTfu;
  }
  if (S->pre_valid && S->pre_uver == S->u_version && S->pre_fu == rhs && S->pre_params == S->P->params_version)
    nk_gmres_preloaded_rhs(S->G, rhs, S->d_rhs_ss, S->pre_grid);
  S->pre_valid = false;
  NK_TRY(nk_gmres_solve_dev(S->G, rhs, du_out, 0, S->lin_abstol, S->lin_reltol, S->o.gmres_maxiters,
                            S->o.gmres_fixed_iters, &info));
  S->last_gmres_iters = info.iters;
  S->stats.gmres_iters += info.iters;
  *ok = !info.failed;
  if (!*ok || !negate) return NK_OK;
  return nk_blas_lincomb(S->ctx, S->n, -1.0, du_out, 0.0, du_out, du_out);
}

// c1 = a·g ; c2 = N − c1 ; partial sums of c2·c2 and c1·c2 — the dogleg segment in one pass (dogleg.jl:136-147)
__global__ __launch_bounds__(NK_BLOCK) void k_dogleg_segment(int64_t n, double a, const double *__restrict__ g,
                                                             const double *__restrict__ N, double *__restrict__ c1,
                                                             double *__restrict__ c2, double *__restrict__ partials) {
  __shared__ double sm[8];
  double s22 = 0.0, s12 = 0.0;
  const int64_t stride = (int64_t)gridDim.x * NK_BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x; i < n; i += stride) {
    const double x1 = a * g[i], x2 = N[i] - x1;
    c1[i] = x1;
    c2[i] = x2;
    s22 += x2 * x2;
    s12 += x1 * x2;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s22 += __shfl_xor(s22, o, 64); s12 += __shfl_xor(s12, o, 64); }
  if ((threadIdx.x & 63) == 0) { sm[threadIdx.x >> 6] = s22; sm[4 + (threadIdx.x >> 6)] = s12; }
  __syncthreads();
  if (threadIdx.x == 0) {
    partials[blockIdx.x] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
    partials[gridDim.x + blockIdx.x] = (sm[4] + sm[5]) + (sm[6] + sm[7]);
  }
}

// ---- Dogleg.solve! (dogleg.jl:86-151). duJJdu = NaN ⇒ "not computed". *have_JTfu: JTfu holds Jᵀfu of this (u, fu).
static int dogleg(nk_solver *S, bool *ok, double *duJJdu_out, bool new_jacobian, bool *have_JTfu) {
  nk_ctx *ctx = S->ctx;
  const int64_t n = S->n;
  *duJJdu_out = NAN;
  *have_JTfu = false;
  NK_TRY(newton_descent(S, S->du_newton, ok, new_jacobian));
  if (!*ok) return NK_OK;
  double v[4];
  NK_TRY(nk_blas_sumsq(ctx, n, S->du_newton, slot(S, 0)));
  NK_TRY(fetch(S, 1, v));
  if (sqrt(v[0]) <= S->tr) return nk_blas_copy(ctx, n, S->du_newton, S->du);
  // δu_cauchy = −Jᵀ fu (steepest.jl:75-77); Jᵀfu itself is what the trust-region scheme needs again for ρ
  NK_TRY(apply_JT(S, S->u, S->fu, S->JTfu));
  *have_JTfu = true;
  NK_TRY(nk_blas_lincomb(ctx, n, -1.0, S->JTfu, 0.0, S->JTfu, S->du_cauchy));
  NK_TRY(apply_J(S, S->du_cauchy, S->Jdu));
  {
    const double *xs[2] = {S->du_cauchy, S->Jdu}, *ys[2] = {S->du_cauchy, S->Jdu};
    NK_TRY(nk_blas_multi_reduce(ctx, n, 2, xs, ys, nullptr, nullptr, 0, 0, slot(S, 0)));
  }
  NK_TRY(fetch(S, 2, v));
  const double l_grad = sqrt(v[0]), dJJd = v[1];
  const double d_cauchy = (l_grad * l_grad * l_grad) / dJJd;
  if (d_cauchy >= S->tr) {
    const double lam = S->tr / l_grad;
    *duJJdu_out = lam * lam * dJJd;
    return nk_blas_lincomb(ctx, n, lam, S->du_cauchy, 0.0, S->du_cauchy, S->du);
  }
  {
    const int grid = nk_grid_for(n, NK_BLOCK * 4, NK_MAX_RED_BLOCKS / 2);
    NK_LAUNCH(ctx, k_dogleg_segment, dim3(grid), dim3(NK_BLOCK), n, d_cauchy / l_grad, (const double *)S->du_cauchy,
              (const double *)S->du_newton, S->c1, S->c2, ctx->d_partials_ss);
    NK_HIP(hipGetLastError());
    NK_TRY(nk_blas_multi_reduce(ctx, 0, 0, nullptr, nullptr, nullptr, ctx->d_partials_ss, 2, grid, slot(S, 0)));
  }
  NK_TRY(fetch(S, 2, v));
  const double a = v[0], b = 2.0 * v[1], c = d_cauchy * d_cauchy - S->tr * S->tr;
  const double aux = fmax(0.0, b * b - 4.0 * a * c);
  const double tau = (-b + sqrt(aux)) / (2.0 * a);
  return nk_blas_lincomb(ctx, n, 1.0, S->c1, tau, S->c2, S->du);
}

// ---- GenericTrustRegionSchemeCache solve! (trust_region.jl:396-514)
// u_trial = u + du with the partial sums of Σ du² and Σ (u_trial − u)² (the latter is the stall test's displacement)
__global__ __launch_bounds__(NK_BLOCK) void k_tr_trial(int64_t n, const double *__restrict__ u, const double *__restrict__ du,
                                                       double *__restrict__ ut, double *__restrict__ partials) {
  __shared__ double sm[8];
  double sd = 0.0, ss = 0.0;
  const int64_t stride = (int64_t)gridDim.x * NK_BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x; i < n; i += stride) {
    const double uo = u[i], d = du[i], un = uo + d, e = un - uo;
    ut[i] = un;
    sd += d * d;
    ss += e * e;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { sd += __shfl_xor(sd, o, 64); ss += __shfl_xor(ss, o, 64); }
  if ((threadIdx.x & 63) == 0) { sm[threadIdx.x >> 6] = sd; sm[4 + (threadIdx.x >> 6)] = ss; }
  __syncthreads();
  if (threadIdx.x == 0) {
    partials[blockIdx.x] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
    partials[gridDim.x + blockIdx.x] = (sm[4] + sm[5]) + (sm[6] + sm[7]);
  }
}

// One trial-point kernel, one multi-reduction, ONE fetch: ‖f(u+δu)‖², ‖f‖², δu·Jᵀf, ‖Jδu‖² (unless the dogleg supplied it),
// ‖f(u+δu)‖∞, ‖δu‖², ‖(u+δu) − u‖² all arrive together (the unfused form issued six two-launch reductions and a copy).
static int tr_solve(nk_solver *S, double duJJdu, bool have_JTfu, bool *accepted, double *step_norm, double *fnew2_out) {
  nk_ctx *ctx = S->ctx;
  const int64_t n = S->n;
  const int m = S->o.radius_update_scheme;
  S->u_trial = spare_u(S);
  const int tgrid = nk_grid_for(n, NK_BLOCK * 4, NK_MAX_RED_BLOCKS / 2);
  NK_LAUNCH(ctx, k_tr_trial, dim3(tgrid), dim3(NK_BLOCK), n, (const double *)S->u, (const double *)S->du, S->u_trial,
            ctx->d_partials_ss);
  NK_HIP(hipGetLastError());
  nk_problem_invalidate(S->P);  // a buffer the problem may have been linearised at has new contents
  NK_TRY(nk_problem_residual_dev(S->P, S->u_trial, S->fu_trial));
  S->stats.nf++;
  const bool have = !isnan(duJJdu);
  if (!have) NK_TRY(apply_J(S, S->du, S->Jdu));
  if (!have_JTfu) NK_TRY(apply_JT(S, S->u, S->fu, S->JTfu));
  double v[7];
  {
    const double *xs[4] = {S->fu_trial, S->fu, S->du, S->Jdu}, *ys[4] = {S->fu_trial, S->fu, S->JTfu, S->Jdu};
    NK_TRY(nk_blas_multi_reduce(ctx, n, have ? 3 : 4, xs, ys, S->fu_trial, ctx->d_partials_ss, 2, tgrid, slot(S, 0)));
    NK_TRY(fetch(S, (have ? 3 : 4) + 3, v));
  }
  const int o = have ? 3 : 4;  // v[o] = ‖f_trial‖∞, v[o+1] = ‖δu‖², v[o+2] = ‖u_trial − u‖²
  if (!have) duJJdu = v[3];
  const double fnew2 = v[0], f2 = v[1], nd = sqrt(v[o + 1]);
  const double fnew_inf = v[o];
  *step_norm = sqrt(v[o + 2]);
  *fnew2_out = fnew2;
  const double num = (fnew2 - f2) / 2.0, denom = v[2] + duJJdu / 2.0;
  S->rho = num / denom;
  const double rho = S->rho;
  S->last_accepted = rho > S->step_thr;
  switch (m) {
    case NK_RUS_SIMPLE:
      if (rho < S->shrink_thr) { S->tr *= S->shrink_f; S->shrink_counter++; }
      else {
        S->shrink_counter = 0;
        if (rho > S->expand_thr && rho > S->step_thr) S->tr = S->expand_f * S->tr;
      }
      break;
    case NK_RUS_NLSOLVE:
      if (rho < S->shrink_thr) { S->tr *= S->shrink_f; S->shrink_counter++; }
      else {
        S->shrink_counter = 0;
        if (rho >= S->expand_thr) S->tr = S->expand_f * nd;
        else if (rho >= S->p1) S->tr = fmax(S->tr, S->expand_f * nd);
      }
      break;
    case NK_RUS_NOCEDAL_WRIGHT:
      if (rho < S->shrink_thr) { S->tr = S->shrink_f * nd; S->shrink_counter++; }
      else {
        S->shrink_counter = 0;
        if (rho > S->expand_thr && fabs(nd - S->tr) < 1e-6 * S->tr) S->tr = S->expand_f * S->tr;
      }
      break;
    case NK_RUS_HEI: {
      const double r = rho, c2 = S->shrink_thr, M = S->p1, g1 = S->p3, g2 = S->p4, beta = S->p2;
      const double rf = (r >= c2) ? (2 * (M - 1 - g2) * atan(r - c2) + (1 + g2)) / M_PI
                                  : (1 - g1 - beta) * (exp(r - c2) + beta / (1 - g1 - beta));
      const double tr_new = rf * nd;
      if (tr_new < S->tr) S->shrink_counter++;
      else S->shrink_counter = 0;
      S->tr = tr_new;
      break;
    }
    case NK_RUS_YUAN: {
      if (rho < S->shrink_thr) { S->p1 = S->p2 * S->p1; S->shrink_counter++; }
      else {
        if (rho >= S->expand_thr && 2 * nd > S->tr) S->p1 = S->p3 * S->p1;
        S->shrink_counter = 0;
      }
      nk_problem_invalidate(S->P);
      NK_TRY(apply_JT(S, S->u_trial, S->fu_trial, S->JTfu));
      NK_TRY(nk_blas_sumsq(ctx, n, S->JTfu, slot(S, 0)));
      double t;
      NK_TRY(fetch(S, 1, &t));
      S->tr = S->p1 * sqrt(t);
      break;
    }
    case NK_RUS_FAN:
      if (rho < S->shrink_thr) { S->p1 *= S->p2; S->shrink_counter++; }
      else {
        S->shrink_counter = 0;
        if (rho > S->expand_thr) S->p1 = fmin(S->p1 * S->p3, S->p4);
      }
      S->tr = S->p1 * pow(sqrt(fnew2), 0.99);
      break;
    case NK_RUS_BASTIN:
      if (rho > S->step_thr) {
        // retrospective ratio with J at the trial point (trust_region.jl:490-504)
        nk_problem_invalidate(S->P);
        NK_TRY(nk_problem_jvp_dev(S->P, S->u_trial, S->du, S->Jdu, nullptr));
        NK_TRY(nk_problem_vjp_dev(S->P, S->u_trial, S->fu_trial, S->JTfu));
        NK_TRY(nk_blas_sumsq(ctx, n, S->JTfu, slot(S, 0)));
        NK_TRY(nk_problem_vjp_dev(S->P, S->u_trial, S->Jdu, S->JTfu));
        NK_TRY(nk_blas_sumsq(ctx, n, S->JTfu, slot(S, 1)));
        NK_TRY(nk_blas_sumsq(ctx, n, S->du, slot(S, 2)));
        double t[3];
        NK_TRY(fetch(S, 3, t));
        const double rho2 = num / (t[0] + t[1] / 2.0);
        if (rho2 >= S->expand_thr) S->tr = S->p1 * sqrt(t[2]);
        S->shrink_counter = 0;
      } else {
        S->tr *= S->p2;
        S->shrink_counter++;
      }
      break;
    default: NK_FAIL(NK_E_INVALID, "bad radius update scheme");
  }
  S->tr = fmin(S->tr, S->max_tr);
  *accepted = S->last_accepted;
  S->fnorm_inf = S->last_accepted ? fnew_inf : S->fnorm_inf;
  return NK_OK;
}

// ---- BackTracking line search on ϕ(α) = ½‖f(u + α δu)‖² (LineSearches.jl BackTracking restated, [EXT]):
// sufficient decrease ϕ(α) ≤ ϕ(0) + c₁ α ϕ'(0); quadratic, then cubic interpolation, safeguarded to
// [ρ_lo α, ρ_hi α]. Every ϕ evaluation is one residual (stats.nf += 1, as the reference's line-search cache does).
static int ls_phi(nk_solver *S, double alpha, double *phi) {
  S->u_trial = spare_u(S);
  nk_problem_invalidate(S->P);
  NK_TRY(nk_blas_lincomb(S->ctx, S->n, 1.0, S->u, alpha, S->du, S->u_trial));
  NK_TRY(nk_problem_residual_dev(S->P, S->u_trial, S->fu_trial));
  S->stats.nf++;
  NK_TRY(nk_blas_sumsq(S->ctx, S->n, S->fu_trial, slot(S, 0)));
  double v;
  NK_TRY(fetch(S, 1, &v));
  *phi = 0.5 * v;
  return NK_OK;
}
static int backtracking(nk_solver *S, double *alpha_out, bool *failed) {
  const nk_options &o = S->o;
  *failed = false;
  // ϕ(0) and ϕ'(0) = fuᵀ (J δu)   (PseudoTransient: the stored J carries the damping α⁻¹ I — take the true product)
  if (is_pt(S)) NK_TRY(nk_problem_jvp_dev(S->P, S->u, S->du, S->Jdu, nullptr));
  else NK_TRY(apply_J(S, S->du, S->Jdu));
  NK_TRY(nk_blas_sumsq(S->ctx, S->n, S->fu, slot(S, 0)));
  NK_TRY(nk_blas_dot(S->ctx, S->n, S->fu, S->Jdu, slot(S, 1)));
  double v[2];
  NK_TRY(fetch(S, 2, v));
  const double phi0 = 0.5 * v[0], dphi0 = v[1];
  double a1 = 1.0, a2 = 1.0, phx0 = phi0, phx1 = phi0;
  NK_TRY(ls_phi(S, a1, &phx1));
  int iterfinite = 0;
  const int iterfinitemax = 1074;  // -log2(eps(Float64)) style bound used by LineSearches.jl
  while (!isfinite(phx1) && iterfinite < iterfinitemax) {
    ++iterfinite;
    a1 = a2;
    a2 = a1 / 2.0;
    NK_TRY(ls_phi(S, a2, &phx1));
  }
  int iteration = 0;
  while (phx1 > phi0 + o.ls_c1 * a2 * dphi0) {
    ++iteration;
    if (iteration > o.ls_maxiters) { *failed = true; break; }
    double atmp;
    if (o.ls_order == 2 || iteration == 1) {
      atmp = -(dphi0 * a2 * a2) / (2.0 * (phx1 - phi0 - dphi0 * a2));
    } else {
      const double div = 1.0 / (a1 * a1 * a2 * a2 * (a2 - a1));
      const double ca = (a1 * a1 * (phx1 - phi0 - dphi0 * a2) - a2 * a2 * (phx0 - phi0 - dphi0 * a1)) * div;
      const double cb = (-a1 * a1 * a1 * (phx1 - phi0 - dphi0 * a2) + a2 * a2 * a2 * (phx0 - phi0 - dphi0 * a1)) * div;
      if (fabs(ca) <= 2.220446049250313e-16) atmp = dphi0 / (2.0 * cb);  // isapprox(a, 0; atol = eps)
      else {
        const double disc = fmax(cb * cb - 3.0 * ca * dphi0, 0.0);
        atmp = (-cb + sqrt(disc)) / (3.0 * ca);
      }
    }
    a1 = a2;
    atmp = (atmp == atmp) ? fmin(atmp, a2 * o.ls_rho_hi) : a2 * o.ls_rho_hi;  // NaNMath.min
    a2 = (atmp == atmp) ? fmax(atmp, a2 * o.ls_rho_lo) : a2 * o.ls_rho_lo;    // NaNMath.max
    phx0 = phx1;
    NK_TRY(ls_phi(S, a2, &phx1));
  }
  *alpha_out = a2;
  return NK_OK;
}

// ---- LineSearchesJL(; method = Static | StrongWolfe | MoreThuente | HagerZhang) [EXT: LineSearch.jl's wrapper around LineSearches.jl,
// the methods of lib/NonlinearSolveFirstOrder/test/rootfind_tests__item2.jl:40-46] on ϕ(α) = ½‖f(u + α δu)‖²,
// ϕ'(α) = f(u + α δu)ᵀ J(u + α δu) δu. Restated from the published algorithms with LineSearches.jl's default parameters
// (oracle/reference_restatement.py::_lsjl is the same code in Python; its Moré–Thuente step function is pinned against SciPy's
// MINPACK-2 dcstep). Every ϕ / ϕ' / (ϕ, ϕ') evaluation is one residual at the trial point (nf += 1) — plus, for ϕ', one
// Jacobian-vector product there — one fused two-scalar reduction, one fetch.
static int ls_phidphi(nk_solver *S, double alpha, double *phi, double *dphi) {
  S->u_trial = spare_u(S);
  nk_problem_invalidate(S->P);
  NK_TRY(nk_blas_lincomb(S->ctx, S->n, 1.0, S->u, alpha, S->du, S->u_trial));
  NK_TRY(nk_problem_residual_dev(S->P, S->u_trial, S->fu_trial));
  S->stats.nf++;
  double v[2] = {0.0, 0.0};
  if (dphi) {
    NK_TRY(nk_problem_jvp_dev(S->P, S->u_trial, S->du, S->Jdu, nullptr));
    const double *xs[2] = {S->fu_trial, S->fu_trial}, *ys[2] = {S->fu_trial, S->Jdu};
    NK_TRY(nk_blas_multi_reduce(S->ctx, S->n, 2, xs, ys, nullptr, nullptr, 0, 0, slot(S, 0)));
    NK_TRY(fetch(S, 2, v));
    *dphi = v[1];
  } else {
    NK_TRY(nk_blas_sumsq(S->ctx, S->n, S->fu_trial, slot(S, 0)));
    NK_TRY(fetch(S, 1, v));
  }
  *phi = 0.5 * v[0];
  return NK_OK;
}
// LineSearches.Static: the proposed step, halved while ϕ is not finite
static int ls_static(nk_solver *S, double *alpha) {
  double a = 1.0, pa;
  NK_TRY(ls_phidphi(S, a, &pa, nullptr));
  for (int it = 0; !isfinite(pa) && it < 52; ++it) {
    a /= 2.0;
    NK_TRY(ls_phidphi(S, a, &pa, nullptr));
  }
  *alpha = a;
  return NK_OK;
}
// LineSearches.StrongWolfe (Nocedal & Wright alg. 3.5 / 3.6, cubic interpolation, c₁ = 1e-4, c₂ = 0.9, ρ = 2)
static double ls_sw_interp(double a1, double a2, double p1, double p2, double d1, double d2) {
  const double q1 = d1 + d2 - 3.0 * (p1 - p2) / (a1 - a2);
  const double rad = q1 * q1 - d1 * d2;
  const double q2 = rad >= 0.0 ? sqrt(rad) : NAN;
  return a2 - (a2 - a1) * ((d2 + q2 - q1) / (d2 - d1 + 2.0 * q2));
}
static int ls_sw_zoom(nk_solver *S, double alo, double ahi, double phi0, double dphi0, double *out) {
  const double c1 = 1e-4, c2 = 0.9;
  double aj = NAN;
  for (int it = 0; it < 10; ++it) {
    double plo, dlo, phi_, dhi, pj, dj;
    NK_TRY(ls_phidphi(S, alo, &plo, &dlo));
    NK_TRY(ls_phidphi(S, ahi, &phi_, &dhi));
    aj = (alo < ahi) ? ls_sw_interp(alo, ahi, plo, phi_, dlo, dhi) : ls_sw_interp(ahi, alo, phi_, plo, dhi, dlo);
    NK_TRY(ls_phidphi(S, aj, &pj, nullptr));
    if (pj > phi0 + c1 * aj * dphi0 || pj > plo) {
      ahi = aj;
    } else {
      NK_TRY(ls_phidphi(S, aj, &pj, &dj));
      if (fabs(dj) <= -c2 * dphi0) break;
      if (dj * (ahi - alo) >= 0.0) ahi = alo;
      alo = aj;
    }
  }
  *out = aj;
  return NK_OK;
}
static int ls_strongwolfe(nk_solver *S, double phi0, double dphi0, double *alpha) {
  const double c1 = 1e-4, c2 = 0.9, rho = 2.0, a_max = 65536.0;
  double a_prev = 0.0, a_i = 1.0, p_prev = phi0, p_i, d_i, tmp;
  for (int i = 1; a_i < a_max; ++i) {
    NK_TRY(ls_phidphi(S, a_i, &p_i, nullptr));
    if (p_i > phi0 + c1 * a_i * dphi0 || (p_i >= p_prev && i > 1)) {
      NK_TRY(ls_sw_zoom(S, a_prev, a_i, phi0, dphi0, alpha));
      return ls_phidphi(S, *alpha, &tmp, nullptr);  // the method returns (α*, ϕ(α*)): one more evaluation
    }
    NK_TRY(ls_phidphi(S, a_i, &p_i, &d_i));
    if (fabs(d_i) <= -c2 * dphi0) { *alpha = a_i; return NK_OK; }
    if (d_i >= 0.0) {
      NK_TRY(ls_sw_zoom(S, a_i, a_prev, phi0, dphi0, alpha));
      return ls_phidphi(S, *alpha, &tmp, nullptr);
    }
    a_prev = a_i;
    p_prev = p_i;
    a_i *= rho;
  }
  *alpha = a_max;
  return ls_phidphi(S, a_max, &tmp, nullptr);
}
// MINPACK cstep (Moré & Thuente 1994): safeguarded cubic / quadratic step + update of the interval of uncertainty
struct mt_state { double stx, fx, dgx, sty, fy, dgy, alpha, f, dg; bool bracketed; int info; };
static void ls_cstep(mt_state &m, double amin, double amax) {
  double &stx = m.stx, &fx = m.fx, &dgx = m.dgx, &sty = m.sty, &fy = m.fy, &dgy = m.dgy, &alpha = m.alpha;
  const double f = m.f, dg = m.dg;
  m.info = 0;
  if ((m.bracketed && (alpha <= fmin(stx, sty) || alpha >= fmax(stx, sty))) || dgx * (alpha - stx) >= 0.0 || amax < amin) return;
  const double sgnd = dg * (dgx / fabs(dgx));
  bool bound;
  double af;
  if (f > fx) {
    m.info = 1; bound = true;
    const double theta = 3.0 * (fx - f) / (alpha - stx) + dgx + dg;
    const double sc = fmax(fabs(theta), fmax(fabs(dgx), fabs(dg)));
    double gamma = sc * sqrt((theta / sc) * (theta / sc) - (dgx / sc) * (dg / sc));
    if (alpha < stx) gamma = -gamma;
    const double pp = gamma - dgx + theta, q = gamma - dgx + gamma + dg, r = pp / q;
    const double ac = stx + r * (alpha - stx);
    const double aq = stx + ((dgx / ((fx - f) / (alpha - stx) + dgx)) / 2.0) * (alpha - stx);
    af = (fabs(ac - stx) < fabs(aq - stx)) ? ac : (ac + aq) / 2.0;
    m.bracketed = true;
  } else if (sgnd < 0.0) {
    m.info = 2; bound = false;
    const double theta = 3.0 * (fx - f) / (alpha - stx) + dgx + dg;
    const double sc = fmax(fabs(theta), fmax(fabs(dgx), fabs(dg)));
    double gamma = sc * sqrt((theta / sc) * (theta / sc) - (dgx / sc) * (dg / sc));
    if (alpha > stx) gamma = -gamma;
    const double pp = gamma - dg + theta, q = gamma - dg + gamma + dgx, r = pp / q;
    const double ac = alpha + r * (stx - alpha);
    const double aq = alpha + (dg / (dg - dgx)) * (stx - alpha);
    af = (fabs(ac - alpha) > fabs(aq - alpha)) ? ac : aq;
    m.bracketed = true;
  } else if (fabs(dg) < fabs(dgx)) {
    m.info = 3; bound = true;
    const double theta = 3.0 * (fx - f) / (alpha - stx) + dgx + dg;
    const double sc = fmax(fabs(theta), fmax(fabs(dgx), fabs(dg)));
    double gamma = sc * sqrt(fmax(0.0, (theta / sc) * (theta / sc) - (dgx / sc) * (dg / sc)));
    if (alpha > stx) gamma = -gamma;
    const double pp = gamma - dg + theta, q = gamma + dgx - dg + gamma, r = pp / q;
    double ac;
    if (r < 0.0 && gamma != 0.0) ac = alpha + r * (stx - alpha);
    else if (alpha > stx) ac = amax;
    else ac = amin;
    const double aq = alpha + (dg / (dg - dgx)) * (stx - alpha);
    if (m.bracketed) af = (fabs(alpha - ac) < fabs(alpha - aq)) ? ac : aq;
    else af = (fabs(alpha - ac) > fabs(alpha - aq)) ? ac : aq;
  } else {
    m.info = 4; bound = false;
    if (m.bracketed) {
      const double theta = 3.0 * (f - fy) / (sty - alpha) + dgy + dg;
      const double sc = fmax(fabs(theta), fmax(fabs(dgy), fabs(dg)));
      double gamma = sc * sqrt((theta / sc) * (theta / sc) - (dgy / sc) * (dg / sc));
      if (alpha > sty) gamma = -gamma;
      const double pp = gamma - dg + theta, q = gamma - dg + gamma + dgy, r = pp / q;
      af = alpha + r * (sty - alpha);
    } else if (alpha > stx) af = amax;
    else af = amin;
  }
  if (f > fx) { sty = alpha; fy = f; dgy = dg; }
  else {
    if (sgnd < 0.0) { sty = stx; fy = fx; dgy = dgx; }
    stx = alpha; fx = f; dgx = dg;
  }
  af = fmax(amin, fmin(amax, af));
  alpha = af;
  if (m.bracketed && bound) {
    if (sty > stx) alpha = fmin(stx + (2.0 / 3.0) * (sty - stx), alpha);
    else alpha = fmax(stx + (2.0 / 3.0) * (sty - stx), alpha);
  }
}
// LineSearches.MoreThuente (f_tol = 1e-4, gtol = 0.9, x_tol = 1e-8, alphamin = 1e-16, alphamax = 65536, maxfev = 100)
static int ls_morethuente(nk_solver *S, double phi0, double dphi0, double *alpha_out) {
  const double f_tol = 1e-4, gtol = 0.9, x_tol = 1e-8, amin = 1e-16, amax = 65536.0;
  const int maxfev = 100;
  int info = 0, info_cstep = 1, nfev = 0;
  bool stage1 = true;
  const double finit = phi0, dgtest = f_tol * dphi0;
  double width = amax - amin, width1 = 2.0 * width;
  mt_state m;
  m.stx = 0.0; m.fx = finit; m.dgx = dphi0;
  m.sty = 0.0; m.fy = finit; m.dgy = dphi0;
  m.bracketed = false;
  m.info = 1;
  double alpha = fmin(fmax(1.0, amin), amax), f, dg, stmin, stmax;
  NK_TRY(ls_phidphi(S, alpha, &f, &dg));
  nfev++;
  for (int itf = 0; (!isfinite(f) || !isfinite(dg)) && itf < 52; ++itf) {
    alpha /= 2.0;
    NK_TRY(ls_phidphi(S, alpha, &f, &dg));
    nfev++;
    m.stx = 0.875 * alpha;
  }
  for (;;) {
    if (m.bracketed) { stmin = fmin(m.stx, m.sty); stmax = fmax(m.stx, m.sty); }
    else { stmin = m.stx; stmax = alpha + 4.0 * (alpha - m.stx); }
    stmin = fmax(amin, stmin);
    stmax = fmin(amax, stmax);
    alpha = fmin(fmax(alpha, amin), amax);
    if ((m.bracketed && (alpha <= stmin || alpha >= stmax)) || nfev >= maxfev - 1 || info_cstep == 0 ||
        (m.bracketed && stmax - stmin <= x_tol * stmax))
      alpha = m.stx;
    NK_TRY(ls_phidphi(S, alpha, &f, &dg));  // (the first pass evaluates the initial step a second time, as LineSearches.jl does)
    nfev++;
    const double ftest1 = finit + alpha * dgtest;
    if ((m.bracketed && (alpha <= stmin || alpha >= stmax)) || info_cstep == 0) info = 6;
    if (alpha == amax && f <= ftest1 && dg <= dgtest) info = 5;
    if (alpha == amin && (f > ftest1 || dg >= dgtest)) info = 4;
    if (nfev >= maxfev) info = 3;
    if (m.bracketed && stmax - stmin <= x_tol * stmax) info = 2;
    if (f <= ftest1 && fabs(dg) <= -gtol * dphi0) info = 1;
    if (info != 0) break;
    if (stage1 && f <= ftest1 && dg >= fmin(f_tol, gtol) * dphi0) stage1 = false;
    m.alpha = alpha;
    if (stage1 && f <= m.fx && f > ftest1) {  // the modified function ψ(α) = ϕ(α) − ϕ(0) − f_tol ϕ'(0) α
      mt_state mm = m;
      mm.fx = m.fx - m.stx * dgtest; mm.fy = m.fy - m.sty * dgtest; mm.f = f - alpha * dgtest;
      mm.dgx = m.dgx - dgtest; mm.dgy = m.dgy - dgtest; mm.dg = dg - dgtest;
      ls_cstep(mm, stmin, stmax);
      m.stx = mm.stx; m.sty = mm.sty; m.alpha = mm.alpha; m.bracketed = mm.bracketed; m.info = mm.info;
      m.fx = mm.fx + mm.stx * dgtest; m.fy = mm.fy + mm.sty * dgtest;
      m.dgx = mm.dgx + dgtest; m.dgy = mm.dgy + dgtest;
    } else {
      m.f = f; m.dg = dg;
      ls_cstep(m, stmin, stmax);
    }
    alpha = m.alpha;
    info_cstep = m.info;
    if (m.bracketed) {
      if (fabs(m.sty - m.stx) >= (2.0 / 3.0) * width1) alpha = m.stx + (m.sty - m.stx) / 2.0;
      width1 = width;
      width = fabs(m.sty - m.stx);
    }
  }
  *alpha_out = alpha;
  return NK_OK;
}
// LineSearches.HagerZhang (Hager & Zhang 2005: bracket B0–B3, secant² S1–S4, update U0–U3 with bisection θ = ½, Wolfe /
// approximate Wolfe tests; δ = 0.1, σ = 0.9, ρ = 5, ε = 1e-6, γ = 0.66, ≤ 50 iterations, ψ₃ = 0.1). The method's exceptions
// (non-descent direction, iteration limit, lost bracket) are reported as a failed line search at the best step so far.
struct hz_state {
  nk_solver *S;
  std::vector<double> a, v, d;  // step lengths, ϕ, ϕ′ of every evaluation (index 0: α = 0)
  double phi_0, dphi_0, phi_lim;
  bool lost = false;
};
static int hz_eval(hz_state &h, double alpha, double *p, double *dp) {
  NK_TRY(ls_phidphi(h.S, alpha, p, dp));
  h.a.push_back(alpha); h.v.push_back(*p); h.d.push_back(*dp);
  return NK_OK;
}
static bool hz_wolfe(const hz_state &h, double c, double pc, double dc) {
  const double delta = 0.1, sigma = 0.9;
  const bool w1 = delta * h.dphi_0 >= (pc - h.phi_0) / c && dc >= sigma * h.dphi_0;
  const bool w2 = (2.0 * delta - 1.0) * h.dphi_0 >= dc && dc >= sigma * h.dphi_0 && pc <= h.phi_lim;
  return w1 || w2;
}
static int hz_bisect(hz_state &h, int *ia, int *ib) {
  double a = h.a[*ia], b = h.a[*ib];
  while (b - a > nextafter(b, INFINITY) - b) {
    const double dd = (a + b) / 2.0;
    double pd, gd;
    NK_TRY(hz_eval(h, dd, &pd, &gd));
    const int id = (int)h.a.size() - 1;
    if (gd >= 0.0) { *ib = id; return NK_OK; }
    if (pd <= h.phi_lim) { a = dd; *ia = id; }
    else { b = dd; *ib = id; }
  }
  return NK_OK;
}
static int hz_update(hz_state &h, int ia, int ib, int ic, int *oa, int *ob) {
  const double a = h.a[ia], b = h.a[ib], c = h.a[ic];
  *oa = ia; *ob = ib;
  if (c < a || c > b) return NK_OK;
  if (h.d[ic] >= 0.0) { *ob = ic; return NK_OK; }
  if (h.v[ic] <= h.phi_lim) { *oa = ic; return NK_OK; }
  *ob = ic;
  return hz_bisect(h, oa, ob);
}
static double hz_secant(double a, double b, double da, double db) { return (a * db - b * da) / (db - da); }
static int hz_secant2(hz_state &h, int ia, int ib, bool *iswolfe, int *oA, int *oB) {
  const double a0 = h.a[ia], b0 = h.a[ib], da = h.d[ia], db = h.d[ib];
  *iswolfe = false;
  if (!(da < 0.0 && db >= 0.0)) { h.lost = true; *oA = ia; *oB = ib; return NK_OK; }
  double c = hz_secant(a0, b0, da, db), pc, dc;
  NK_TRY(hz_eval(h, c, &pc, &dc));
  int ic = (int)h.a.size() - 1;
  if (hz_wolfe(h, c, pc, dc)) { *iswolfe = true; *oA = *oB = ic; return NK_OK; }
  int iA, iB;
  NK_TRY(hz_update(h, ia, ib, ic, &iA, &iB));
  const double a = h.a[iA], b = h.a[iB];
  if (iB == ic) c = hz_secant(h.a[ib], h.a[iB], h.d[ib], h.d[iB]);
  else if (iA == ic) c = hz_secant(h.a[ia], h.a[iA], h.d[ia], h.d[iA]);
  if ((iA == ic || iB == ic) && a <= c && c <= b) {
    NK_TRY(hz_eval(h, c, &pc, &dc));
    ic = (int)h.a.size() - 1;
    if (hz_wolfe(h, c, pc, dc)) { *iswolfe = true; *oA = *oB = ic; return NK_OK; }
    int jA, jB;
    NK_TRY(hz_update(h, iA, iB, ic, &jA, &jB));
    iA = jA; iB = jB;
  }
  *oA = iA; *oB = iB;
  return NK_OK;
}
static int ls_hagerzhang(nk_solver *S, double phi_0, double dphi_0, double *alpha_out, bool *failed) {
  const double rho = 5.0, eps_hz = 1e-6, gamma = 0.66, psi3 = 0.1, feps = 2.220446049250313e-16;
  const int lsmax = 50;
  double alphamax = INFINITY;
  *failed = false;
  if (!(isfinite(phi_0) && isfinite(dphi_0)) || dphi_0 >= feps * fabs(phi_0)) { *alpha_out = 0.0; *failed = true; return NK_OK; }
  hz_state h;
  h.S = S;
  h.a.push_back(0.0); h.v.push_back(phi_0); h.d.push_back(dphi_0);
  h.phi_0 = phi_0; h.dphi_0 = dphi_0;
  h.phi_lim = phi_0 + eps_hz * fabs(phi_0);
  double c = 1.0, phi_c, dphi_c;
  NK_TRY(ls_phidphi(S, c, &phi_c, &dphi_c));
  for (int itf = 1; !(isfinite(phi_c) && isfinite(dphi_c)) && itf < 53; ++itf) {
    c *= psi3;
    NK_TRY(ls_phidphi(S, c, &phi_c, &dphi_c));
  }
  if (!(isfinite(phi_c) && isfinite(dphi_c))) { *alpha_out = 0.0; return NK_OK; }
  h.a.push_back(c); h.v.push_back(phi_c); h.d.push_back(dphi_c);
  bool bracketed = false;
  int ia = 0, ib = 1, it = 1;
  while (!bracketed && it < lsmax) {  // B0–B3
    if (dphi_c >= 0.0) {
      ib = (int)h.a.size() - 1;
      for (int i = ib - 1; i >= 0; --i)
        if (h.v[i] <= h.phi_lim) { ia = i; break; }
      bracketed = true;
    } else if (h.v.back() > h.phi_lim) {
      ib = (int)h.a.size() - 1;
      ia = 0;
      NK_TRY(hz_bisect(h, &ia, &ib));
      bracketed = true;
    } else {
      const double cold = c;
      if (nextafter(cold, INFINITY) >= alphamax) { *alpha_out = cold; return NK_OK; }
      c = fmin(c * rho, alphamax);
      NK_TRY(ls_phidphi(S, c, &phi_c, &dphi_c));
      for (int itf = 1; !(isfinite(phi_c) && isfinite(dphi_c)) && c > nextafter(cold, INFINITY) && itf < 53; ++itf) {
        alphamax = c;
        c = (cold + c) / 2.0;
        NK_TRY(ls_phidphi(S, c, &phi_c, &dphi_c));
      }
      if (!(isfinite(phi_c) && isfinite(dphi_c))) { *alpha_out = cold; return NK_OK; }
      if (dphi_c < 0.0 && c == alphamax) { *alpha_out = c; return NK_OK; }
      h.a.push_back(c); h.v.push_back(phi_c); h.d.push_back(dphi_c);
    }
    ++it;
  }
  while (it < lsmax) {  // L1–L3
    const double a = h.a[ia], b = h.a[ib];
    if (b - a <= nextafter(b, INFINITY) - b) { *alpha_out = a; return NK_OK; }
    bool isw;
    int iA, iB;
    NK_TRY(hz_secant2(h, ia, ib, &isw, &iA, &iB));
    if (h.lost) { *alpha_out = h.a[ia]; *failed = true; return NK_OK; }
    if (isw) { *alpha_out = h.a[iA]; return NK_OK; }
    const double A = h.a[iA], B = h.a[iB];
    if (B - A < gamma * (b - a)) {
      if (nextafter(h.v[ia], INFINITY) >= h.v[ib] && nextafter(h.v[iA], INFINITY) >= h.v[iB]) { *alpha_out = A; return NK_OK; }
      ia = iA; ib = iB;
    } else {
      double pc, dc;
      NK_TRY(hz_eval(h, (A + B) / 2.0, &pc, &dc));
      int ja, jb;
      NK_TRY(hz_update(h, iA, iB, (int)h.a.size() - 1, &ja, &jb));
      ia = ja; ib = jb;
    }
    ++it;
  }
  *alpha_out = h.a[ia];  // iteration limit: LineSearchException in LineSearches.jl
  *failed = true;
  return NK_OK;
}

static int linesearch_lsjl(nk_solver *S, double *alpha_out, bool *failed) {
  *failed = false;
  double phi0, dphi0;
  NK_TRY(ls_phidphi(S, 0.0, &phi0, &dphi0));
  if (dphi0 >= 0.0) {  // not a descent direction: the full step, reported as a failed line search
    *alpha_out = 1.0;
    *failed = true;
    return NK_OK;
  }
  switch (S->o.linesearch) {
    case 2: return ls_static(S, alpha_out);
    case 3: return ls_strongwolfe(S, phi0, dphi0, alpha_out);
    case 4: return ls_morethuente(S, phi0, dphi0, alpha_out);
    case 5: return ls_hagerzhang(S, phi0, dphi0, alpha_out, failed);
    default: NK_FAIL(NK_E_INVALID, "bad linesearch %d", S->o.linesearch);
  }
}

// ---- LevenbergMarquardt
// DᵀD ← max(DᵀD, diag(JᵀJ)) — update_levenberg_marquardt_diagonal!! (levenberg_marquardt.jl:270-293)
__global__ __launch_bounds__(NK_BLOCK) void k_lm_dtd_max(int64_t n, const double *__restrict__ diag, double *__restrict__ dtd) {
  const int64_t stride = (int64_t)gridDim.x * NK_BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x; i < n; i += stride) {
    const double a = dtd[i], b = diag[i];
    dtd[i] = (b > a) ? b : a;  // Base.max would propagate a NaN of diag(JᵀJ); a NaN Jacobian ends the solve as Unstable anyway
  }
}
// fu_c ← (2/h)·((f(u + h v) − fu)/h − J v) — the second directional derivative (geodesic_acceleration.jl:112-117)
__global__ __launch_bounds__(NK_BLOCK) void k_lm_geo_rhs(int64_t n, double h, const double *__restrict__ fu,
                                                         const double *__restrict__ Jv, double *__restrict__ fc) {
  const int64_t stride = (int64_t)gridDim.x * NK_BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x; i < n; i += stride)
    fc[i] = (2.0 / h) * ((fc[i] - fu[i]) / h - Jv[i]);
}
// DampedNewtonDescent.solve! in :normal_form mode (damped_newton.jl:297-313): (JᵀJ + λDᵀD) x = Jᵀ rhs, δ = −x.
// recompute_A (the velocity solve, idx = Val(1)) refreshes DᵀD from the current J and fixes the damping λ·DᵀD for this step;
// the acceleration solve reuses it (:307-309).
static int lm_damped_solve(nk_solver *S, const double *rhs_f, double *out, bool recompute_A, bool *ok) {
  nk_ctx *ctx = S->ctx;
  S->stats.nsolve++;
  if (recompute_A) {
    NK_TRY(nk_csr_colsumsq_dev(S->J, S->lm_diag));
    const int grid = nk_grid_for(S->n, NK_BLOCK * 4, 2048);
    NK_LAUNCH(ctx, k_lm_dtd_max, dim3(grid), dim3(NK_BLOCK), S->n, (const double *)S->lm_diag, S->lm_dtd);
    NK_HIP(hipGetLastError());
    if (direct(S)) {
      // factorising linear solver: the reference takes the QR least-squares form min ‖[J; √(λDᵀD)] x − [f; 0]‖
      // (damped_newton.jl:258-296); the device factorises the normal equations of the same problem, JᵀJ + λDᵀD assembled on
      // the pattern of JᵀJ — same minimiser, the conditioning of J squared
      NK_TRY(nk_normal_plan_values(S->nplan, S->J, S->lm_lam, S->lm_dtd));
      int fok = 0;
      NK_TRY(nk_bandlu_factor(S->B, nk_normal_plan_matrix(S->nplan), &fok));
      S->stats.nfactors++;
      S->lu_valid = fok != 0;
    } else {
      NK_TRY(nk_gmres_set_normal_form(S->G, 1));
      NK_TRY(nk_gmres_set_normal_form_damping(S->G, S->lm_dtd, S->lm_lam));
    }
  }
  NK_TRY(nk_csr_spmv_t_dev(S->J, rhs_f, S->lm_rhs));
  if (direct(S)) {
    S->last_gmres_iters = 0;
    *ok = S->lu_valid;
    if (!S->lu_valid) return nk_blas_fill(ctx, S->n, 0.0, out);
    NK_TRY(nk_bandlu_solve(S->B, S->lm_rhs, out));
    return nk_blas_lincomb(ctx, S->n, -1.0, out, 0.0, out, out);
  }
  nk_gmres_info info;
  NK_TRY(nk_gmres_solve_dev(S->G, S->lm_rhs, out, 0, S->lin_abstol, S->lin_reltol, S->o.gmres_maxiters,
                            S->o.gmres_fixed_iters, &info));
  S->last_gmres_iters = info.iters;
  S->stats.gmres_iters += info.iters;
  *ok = !info.failed;
  return nk_blas_lincomb(ctx, S->n, -1.0, out, 0.0, out, out);
}
// callback_into_cache!(topcache, ::LevenbergMarquardtDampingCache) (levenberg_marquardt.jl:159-168)
static void lm_callback(nk_solver *S) {
  const bool geo_ok = S->o.lm_disable_geodesic ? true : S->lm_geo_accepted;  // last_step_accepted default: true
  if (S->lm_tr_accepted && geo_ok) S->lm_lam_factor = 1.0 / S->o.lm_damping_decrease_factor;
  S->lm_lam *= S->lm_lam_factor;
  S->lm_lam_factor = S->o.lm_damping_increase_factor;
  S->tr = S->lm_lam;
}
static int check_and_update(nk_solver *S, double step_norm);
static int internal_step(nk_solver *S, int recompute, bool evaluate_residual);
// the rest of step! (FirstOrder/src/solve.jl:365-462) for LevenbergMarquardt: GeodesicAcceleration.solve!
// (geodesic_acceleration.jl:98-136), LevenbergMarquardtTrustRegionCache solve! (levenberg_marquardt.jl:247-268)
static int lm_step(nk_solver *S, bool new_jacobian, bool evaluate_residual) {
  nk_ctx *ctx = S->ctx;
  const int64_t n = S->n;
  const bool geo = !S->o.lm_disable_geodesic;
  bool ok = true, success = true;
  NK_TRY(lm_damped_solve(S, S->fu, S->lm_v, true, &ok));
  if (!geo && !ok) {  // DampedNewtonDescent alone reports linsolve_success = false (damped_newton.jl:334-337)
    if (new_jacobian) {
      S->retcode = NK_RET_INTERNAL_LINEAR_SOLVE_FAILED;
      S->force_stop = true;
      return NK_OK;
    }
    S->make_new_jacobian = true;
    return internal_step(S, 1, evaluate_residual);
  }
  double nv2 = NAN;
  if (geo) {  // (an inner failure is not propagated: geodesic's solve! reads `.δu` of the inner results only)
    const double h = S->o.lm_finite_diff_step_geodesic;
    double *uc = spare_u(S);
    nk_problem_invalidate(S->P);
    NK_TRY(nk_blas_lincomb(ctx, n, 1.0, S->u, h, S->lm_v, uc));
    NK_TRY(nk_problem_residual_dev(S->P, uc, S->fu_trial));  // Utils.evaluate_f!! — not counted in stats.nf
    NK_TRY(apply_J(S, S->lm_v, S->Jdu));
    const int grid = nk_grid_for(n, NK_BLOCK * 4, 2048);
    NK_LAUNCH(ctx, k_lm_geo_rhs, dim3(grid), dim3(NK_BLOCK), n, h, (const double *)S->fu, (const double *)S->Jdu, S->fu_trial);
    NK_HIP(hipGetLastError());
    NK_TRY(lm_damped_solve(S, S->fu_trial, S->lm_a, false, &ok));
    double v[2];
    const double *xs[2] = {S->lm_v, S->lm_a}, *ys[2] = {S->lm_v, S->lm_a};
    NK_TRY(nk_blas_multi_reduce(ctx, n, 2, xs, ys, nullptr, nullptr, 0, 0, slot(S, 0)));
    NK_TRY(fetch(S, 2, v));
    nv2 = v[0];
    S->lm_geo_accepted = 2.0 * sqrt(v[1]) <= sqrt(v[0]) * S->o.lm_alpha_geodesic;
    success = S->lm_geo_accepted;
    if (success) NK_TRY(nk_blas_lincomb(ctx, n, 1.0, S->lm_v, 0.5, S->lm_a, S->du));  // δu = v + a/2
  } else {
    NK_TRY(nk_blas_copy(ctx, n, S->lm_v, S->du));
  }
  bool accepted = false;
  double step_norm = 0.0;
  if (success) {
    S->make_new_jacobian = true;
    // β = cos(v, v_old); trial point; loss = ‖f(u + δu)‖₂ — one trial kernel, one multi-reduction, one fetch
    S->u_trial = spare_u(S);
    const int tgrid = nk_grid_for(n, NK_BLOCK * 4, NK_MAX_RED_BLOCKS / 2);
    NK_LAUNCH(ctx, k_tr_trial, dim3(tgrid), dim3(NK_BLOCK), n, (const double *)S->u, (const double *)S->du, S->u_trial,
              ctx->d_partials_ss);
    NK_HIP(hipGetLastError());
    nk_problem_invalidate(S->P);
    NK_TRY(nk_problem_residual_dev(S->P, S->u_trial, S->fu_trial));
    S->stats.nf++;
    double v[6];
    const double *xs[3] = {S->lm_v, S->lm_v, S->fu_trial}, *ys[3] = {S->lm_vcache, S->lm_v, S->fu_trial};
    NK_TRY(nk_blas_multi_reduce(ctx, n, 3, xs, ys, S->fu_trial, ctx->d_partials_ss, 2, tgrid, slot(S, 0)));
    NK_TRY(fetch(S, 6, v));  // v·v_cache, v·v, ‖f_new‖², ‖f_new‖∞, ‖δu‖², ‖u_trial − u‖²
    (void)nv2;
    const double norm_v = sqrt(v[1]);
    const double beta = v[0] / (norm_v * S->lm_norm_v_old);
    S->lm_beta = beta;
    const double loss = sqrt(v[2]);
    const double lhs = pow(1.0 - beta, S->o.lm_b_uphill) * loss;  // NaN for a negative base with a fractional exponent
    if (lhs <= INFINITY) {  // loss_old is initialised to Inf and never written again by the reference (:206-268)
      accepted = true;
      S->lm_norm_v_old = norm_v;
      double *t = S->lm_vcache; S->lm_vcache = S->lm_v; S->lm_v = t;  // copyto!(v_cache, v)
      S->u = S->u_trial;
      t = S->fu; S->fu = S->fu_trial; S->fu_trial = t;
      S->fnorm2 = loss;
      S->fnorm_inf = v[3];
      S->u_version++;
      step_norm = sqrt(v[5]);
    } else {
      S->make_new_jacobian = false;
    }
    S->lm_tr_accepted = accepted;
    NK_TRY(check_and_update(S, step_norm));
  } else {
    S->make_new_jacobian = false;
  }
  if (S->o.store_trace) {
    nk_trace_entry e;
    memset(&e, 0, sizeof(e));
    e.iter = S->nsteps + 1;
    e.gmres_iters = S->last_gmres_iters;
    e.accepted = accepted ? 1 : 0;
    e.fnorm_inf = S->fnorm_inf;
    NK_TRY(nk_blas_sumsq(ctx, n, S->du, slot(S, 0)));
    double t;
    NK_TRY(fetch(S, 1, &t));
    e.step_norm2 = sqrt(t);
    e.eta = S->lin_reltol;
    e.trust_region = S->lm_lam;  // the damping in force during this step
    e.rho = S->lm_beta;
    S->trace.push_back(e);
  }
  lm_callback(S);
  return NK_OK;
}

// check_and_update! (termination_conditions.jl:414-426)
static int check_and_update(nk_solver *S, double step_norm) {
  bool stop = false;
  NK_TRY(tc_check(S, step_norm, &stop));
  if (stop) {
    S->retcode = S->tc_retcode;
    NK_TRY(rollback_to_best(S));
    S->force_stop = true;
  }
  return NK_OK;
}
// supports_deferred_residual (FirstOrder/src/solve.jl:303-316): only the unglobalised step, only a residual-only
// termination mode (AbsTerminationMode / AbsNormTerminationMode, termination_conditions.jl:43-45), only without a trace
static bool supports_deferred_residual(const nk_solver *S) {
  if (is_tr(S) || is_lm(S) || S->o.linesearch) return false;
  if (!(S->o.termination_mode == TM_ABS || S->o.termination_mode == TM_ABSNORM)) return false;
  return !S->o.store_trace;
}
// refresh_residual! (FirstOrder/src/solve.jl:318-324)
static int refresh_residual(nk_solver *S) {
  if (!S->fu_deferred) return NK_OK;
  S->fu_deferred = false;
  NK_TRY(nk_problem_residual_dev(S->P, S->u, S->fu));
  S->stats.nf++;
  NK_TRY(residual_norms(S, nullptr, 0, nullptr));
  return check_and_update(S, 0.0);
}

// precs(A, p) for the Jacobian just refreshed: the built-in object is refactorised (created on first use) and installed on
// its side; the caller's hook then installs whatever it returns
static int refresh_precs(nk_solver *S) {
  // M ≈ J is a preconditioner for J, not for JᵀJ (+ λDᵀD): LevenbergMarquardt / GaussNewton / normal-form solves leave the
  // built-in objects alone (a user's `precs` sees the operator it is asked about and may do better)
  if (S->o.precond_kind && !normal_form(S) && !is_lm(S)) {
    if (!S->prec_obj) {
      NK_REQUIRE(concrete(S) && S->J, "nk_options.precond_kind needs a concrete-J linsolve");
      if (S->o.precond_kind == 1) NK_TRY(nk_precond_create_jacobi(S->J, &S->prec_obj));
      else if (S->o.precond_kind == 4) NK_TRY(nk_precond_create_amg(S->J, nullptr, &S->prec_obj));
      else NK_TRY(nk_precond_create_ilu0(S->J, S->o.precond_kind == 3 ? NK_ILU_MULTICOLOR : NK_ILU_NATURAL, &S->prec_obj));
    } else {
      NK_TRY(nk_precond_update(S->prec_obj));
    }
    NK_TRY(nk_gmres_set_preconditioner(S->G, S->o.precond_side == NK_SIDE_RIGHT ? NK_SIDE_RIGHT : NK_SIDE_LEFT, S->prec_obj));
  }
  if (S->precs && S->precs(S->precs_user, S->G, concrete(S) ? S->J : nullptr, S->u) != 0)
    NK_FAIL(NK_E_CALLBACK, "precs callback failed");
  return NK_OK;
}
extern "C" int nk_solver_set_precs(nk_solver *S, nk_precs_fn fn, void *user) {
  NK_REQUIRE(S, "NULL argument");
  NK_REQUIRE(!fn || S->G, "precs needs a Krylov linsolve");
  S->precs = fn;
  S->precs_user = user;
  // LinearSolve evaluates precs when the linear cache is built [EXT]: once now, for the operator the cache was built with
  if (fn && fn(user, S->G, concrete(S) ? S->J : nullptr, S->u) != 0) NK_FAIL(NK_E_CALLBACK, "precs callback failed");
  return NK_OK;
}
extern "C" nk_gmres *nk_solver_gmres(nk_solver *S) { return S ? S->G : nullptr; }
extern "C" nk_csr *nk_solver_jacobian(nk_solver *S) { return S ? S->J : nullptr; }

// ---- InternalAPI.step! (FirstOrder/src/solve.jl:325-465)
static int internal_step(nk_solver *S, int recompute /*-1 nothing, 0 false, 1 true*/, bool evaluate_residual) {
  nk_ctx *ctx = S->ctx;
  const int64_t n = S->n;
  // the descent is taken from the residual at the iterate it starts from: settle an outstanding deferral first
  NK_TRY(refresh_residual(S));  // (as in the reference, the step goes on even if this check terminated the solve)
  const bool defer_residual = !evaluate_residual && supports_deferred_residual(S);
  bool new_jacobian;
  if ((recompute < 0 || recompute == 1) && S->make_new_jacobian) {
    if (concrete(S)) {
      NK_TRY(refresh_J(S));
    } else {
      NK_TRY(nk_gmres_set_operator_jvp(S->G, S->P, S->u, NK_DEVICE));  // StatefulJacobianOperator(J, u, p)
    }
    new_jacobian = true;
    if (!direct(S) && S->o.cheb_degree > 0)  // precs(A, p) is re-evaluated for every new A
      NK_TRY(nk_gmres_set_chebyshev_preconditioner(S->G, S->o.cheb_degree, 0.0, 0.0, S->o.cheb_ratio));
    if (!direct(S) && S->o.mg_nu > 0)
      NK_TRY(nk_gmres_set_multigrid_preconditioner(S->G, S->P, S->u, NK_DEVICE, S->o.mg_nu, S->o.mg_coarse));
    if (!direct(S)) NK_TRY(refresh_precs(S));
  } else {
    new_jacobian = false;
    drop_begun_ahead(S);
    S->spec_valid = false;   // (a value set filled ahead is only ever taken by the step that follows its fill)
  }
  if (is_lm(S)) return lm_step(S, new_jacobian, evaluate_residual);
  const bool has_forcing = S->o.forcing == NK_FORCING_EISENSTAT_WALKER2;
  if (has_forcing) NK_TRY(pre_step_forcing(S, S->nsteps));

  bool ok = true, have_JTfu = false;
  double duJJdu = NAN;
  // plain Newton step: the update takes x of J x = fu as it is (u − x), sparing the δu = −x pass
  const bool fold_sign = !is_tr(S) && !S->o.linesearch;
  // plain Newton through GMRES: u_new = u − x is formed by the pass that forms x (nk_gmres_arm_fused_update)
  const bool fuse_update = fold_sign && !direct(S) && !is_pt(S) && !normal_form(S) && S->G != nullptr;
  // (armed for THIS descent only: an error exit below must not leave the object armed with pointers into the u pool — a later
  //  zero-guess, single-cycle solve on the same object would write a u buffer)
  struct fu_guard_t { nk_gmres *g; ~fu_guard_t() { if (g) (void)nk_gmres_take_fused_update(g, nullptr); } } fu_guard{fuse_update ? S->G : nullptr};
  if (fuse_update) nk_gmres_arm_fused_update(S->G, S->u, spare_u(S), -1.0, ctx->d_partials_ss);
  if (is_tr(S)) NK_TRY(dogleg(S, &ok, &duJJdu, new_jacobian, &have_JTfu));
  else NK_TRY(newton_descent(S, S->du, &ok, new_jacobian, !fold_sign));
  int fused_grid = 0;
  const bool update_fused = fuse_update && nk_gmres_take_fused_update(S->G, &fused_grid) && ok;
  if (!ok) {
    if (new_jacobian) {
      S->retcode = NK_RET_INTERNAL_LINEAR_SOLVE_FAILED;
      S->force_stop = true;
      return NK_OK;
    }
    S->make_new_jacobian = true;
    return internal_step(S, 1, evaluate_residual);
  }
  if (has_forcing) {  // post_step_forcing!: ‖fu‖ BEFORE u moves (one-step lag) — cached with the residual's other norms
    S->rnorm_prev = S->rnorm;
    S->rnorm = S->fnorm2;
  }
  S->make_new_jacobian = true;
  bool accepted = true;
  double step_norm = 0.0, du_norm = NAN;
  if (is_tr(S)) {
    double fnew2 = 0.0;
    NK_TRY(tr_solve(S, duJJdu, have_JTfu, &accepted, &step_norm, &fnew2));
    if (accepted) {  // take the trial point: pointer assignments, no copies
      S->u = S->u_trial;
      double *t = S->fu; S->fu = S->fu_trial; S->fu_trial = t;
      S->fnorm2 = sqrt(fnew2);
      S->u_version++;
    } else {
      S->make_new_jacobian = false;
      step_norm = 0.0;  // u did not move: the stall test sees ‖u − u_cache‖₂ = 0 (termination_conditions.jl:311-316)
    }
    if (S->shrink_counter > S->o.max_shrink_times) {
      S->retcode = NK_RET_SHRINK_THRESHOLD_EXCEEDED;
      S->force_stop = true;
    }
  } else {
    if (S->o.linesearch) {  // Val(:LineSearch): α from the line search, then axpy!(α, δu, u)  (solve.jl:392-408)
      double alpha = 1.0;
      bool lsfail = false;
      if (S->o.linesearch == 1) NK_TRY(backtracking(S, &alpha, &lsfail));
      else NK_TRY(linesearch_lsjl(S, &alpha, &lsfail));
      if (lsfail) {
        S->retcode = NK_RET_INTERNAL_LINESEARCH_FAILED;
        S->force_stop = true;
      }
      if (alpha != 1.0) NK_TRY(nk_blas_lincomb(ctx, n, alpha, S->du, 0.0, S->du, S->du));  // δu ← α δu, then u += δu
    }
    const int grid = update_fused ? fused_grid : nk_grid_for(n, NK_BLOCK * 4, NK_MAX_RED_BLOCKS);
    double *un = spare_u(S);
    if (!update_fused) {
      nk_prof_scope prof_(ctx, NK_K_NEWTON_UPDATE, 24.0 * (double)n);
      NK_LAUNCH(ctx, k_newton_update, dim3(grid), dim3(NK_BLOCK), n, fold_sign ? -1.0 : 1.0, (const double *)S->du,
                (const double *)S->u, un, ctx->d_partials_ss);
    }
    NK_HIP(hipGetLastError());
    S->u = un;
    S->u_version++;
    nk_problem_invalidate(S->P);
    if (defer_residual) {  // the driver asked for it and nothing would observe the difference: refresh_residual! pays later
      S->fu_deferred = true;
      return NK_OK;
    }
    int norm_grid = 0;   // (> 0: the residual kernel has left the norms' stage-1 partials in ctx->d_partials)
    // … and, for the next step's linear solve (plain Newton: its right-hand side is this f), f in column 0 of the Krylov basis
    double *v0 = (fold_sign && !direct(S) && !is_pt(S) && !normal_form(S) && S->G != nullptr) ? nk_gmres_rhs_column(S->G) : nullptr;
    if (v0 != nullptr && S->d_rhs_ss == nullptr) NK_TRY(nk_dev_alloc(&S->d_rhs_ss, (size_t)NK_MAX_RED_BLOCKS));
    // (… and, for a cycle begin that may run ahead of the next solve, the Gershgorin partials of J(u_new): the Bratu residual kernel
    //  evaluates the exponential the discs' centres need anyway)
    const bool want_gersh = v0 != nullptr && S->P->kind == NK_PROBLEM_BRATU2D && nk_ctx_is_single(ctx) && speculation_allowed(S);
    if (want_gersh && S->d_rhs_gersh == nullptr) NK_TRY(nk_dev_alloc(&S->d_rhs_gersh, (size_t)2 * NK_MAX_RED_BLOCKS));
    NK_TRY(nk_problem_residual_norms_dev(S->P, S->u, S->fu, ctx->d_partials, &norm_grid, v0, v0 ? S->d_rhs_ss : nullptr,
                                         want_gersh ? S->d_rhs_gersh : nullptr));
    S->rhs_gersh_grid = (want_gersh && norm_grid > 0 && S->P->j0 == 0 && S->P->j1 == S->P->ns && !S->P->replicated) ? norm_grid : 0;
    S->pre_valid = norm_grid > 0 && v0 != nullptr;
    if (S->pre_valid) {
      S->pre_uver = S->u_version; S->pre_params = S->P->params_version; S->pre_fu = S->fu; S->pre_grid = norm_grid;
    }
    if (norm_grid == 0) NK_TRY(nk_problem_residual_dev(S->P, S->u, S->fu));
    S->stats.nf++;
    // ‖f‖∞, ‖f‖₂, ‖u − u_prev‖₂: one fetch — and behind the kernels that produce them, before the host waits, the next step's
    // Jacobian values (speculate_J)
    // (where the fill kernel can carry the norms' stage-2 reduction — fold_into — the fill is enqueued by THAT call, in place of
    //  k_reduce_inf2; otherwise behind it, by before_wait)
    bool spec_done = false;
    std::function<int(const nk_fold_norms &, bool *)> fold_into;
    if (speculation_allowed(S) && S->P->kind == NK_PROBLEM_BRATU2D && nk_ctx_is_single(ctx))   // (the fill kernel that can carry it)
      fold_into = [S, &spec_done](const nk_fold_norms &f, bool *folded) -> int {
        spec_done = true;
        return speculate_J(S, S->u, S->u_version, &f, folded);
      };
    NK_TRY(residual_norms(S, ctx->d_partials_ss, grid, &step_norm,
                          [S, &spec_done]() -> int { return spec_done ? NK_OK : speculate_J(S, S->u, S->u_version); }, norm_grid,
                          fold_into));
    if (S->o.store_trace) {
      NK_TRY(nk_blas_sumsq(ctx, n, S->du, slot(S, 0)));
      double v;
      NK_TRY(fetch(S, 1, &v));
      du_norm = sqrt(v);
    }
  }
  NK_TRY(check_and_update(S, step_norm));
  if (S->force_stop) drop_begun_ahead(S);
  if (S->o.store_trace) {
    nk_trace_entry e;
    memset(&e, 0, sizeof(e));
    e.iter = S->nsteps + 1;
    e.gmres_iters = S->last_gmres_iters;
    e.accepted = accepted ? 1 : 0;
    e.fnorm_inf = S->fnorm_inf;
    if (is_tr(S)) {
      NK_TRY(nk_blas_sumsq(ctx, n, S->du, slot(S, 0)));
      double v;
      NK_TRY(fetch(S, 1, &v));
      du_norm = sqrt(v);
    }
    e.step_norm2 = du_norm;
    e.eta = S->lin_reltol;
    e.trust_region = is_tr(S) ? S->tr : NAN;
    e.rho = is_tr(S) ? S->rho : NAN;
    S->trace.push_back(e);
  }
  return NK_OK;
}

// CommonSolve.step!(cache; recompute_jacobian, evaluate_residual) (Base/src/solve.jl:835-859)
extern "C" int nk_solver_step_ex(nk_solver *S, int recompute_jacobian, int evaluate_residual) {
  NK_REQUIRE(S, "NULL argument");
  NK_REQUIRE(recompute_jacobian >= -1 && recompute_jacobian <= 1, "recompute_jacobian must be -1 (nothing), 0 or 1");
  NK_HIP(hipSetDevice(S->ctx->device));
  if (S->force_stop || S->nsteps >= S->o.maxiters) return NK_OK;
  const auto t0 = std::chrono::steady_clock::now();
  NK_TRY(internal_step(S, recompute_jacobian, evaluate_residual != 0));
  S->stats.nsteps++;
  S->nsteps++;
  if (S->o.maxtime > 0.0) {
    S->total_time += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (!S->force_stop && S->retcode == NK_RET_DEFAULT && S->total_time >= S->o.maxtime) {
      S->retcode = NK_RET_MAXTIME;
      S->force_stop = true;
    }
  }
  return NK_OK;
}

extern "C" int nk_solver_step(nk_solver *S) { return nk_solver_step_ex(S, -1, 1); }
extern "C" int nk_solver_supports_deferred_residual(nk_solver *S, int *yes) {
  NK_REQUIRE(S && yes, "NULL argument");
  *yes = supports_deferred_residual(S) ? 1 : 0;
  return NK_OK;
}
extern "C" int nk_solver_refresh_residual(nk_solver *S) {
  NK_REQUIRE(S, "NULL argument");
  NK_HIP(hipSetDevice(S->ctx->device));
  return refresh_residual(S);
}

extern "C" int nk_solver_solve(nk_solver *S, int *retcode) {  // _run_cache_to_completion!
  NK_REQUIRE(S, "NULL argument");
  while (!S->force_stop && S->nsteps < S->o.maxiters) NK_TRY(nk_solver_step(S));
  if (S->retcode == NK_RET_DEFAULT) S->retcode = (S->nsteps >= S->o.maxiters) ? NK_RET_MAXITERS : NK_RET_SUCCESS;
  // a driver may have stepped with evaluate_residual = false: bring the residual forward before it is reported
  NK_TRY(refresh_residual(S));
  NK_TRY(rollback_to_best(S));
  NK_HIP(hipStreamSynchronize(S->ctx->stream));
  if (retcode) *retcode = S->retcode;
  return NK_OK;
}

extern "C" int nk_solver_reinit(nk_solver *S, const double *u0, int memspace, const double *params, int nparams) {
  NK_REQUIRE(S, "NULL argument");
  NK_HIP(hipSetDevice(S->ctx->device));
  if (params) NK_TRY(nk_problem_set_params(S->P, params, nparams));
  if (u0)
    NK_HIP(hipMemcpyAsync(S->u, u0, S->n * sizeof(double),
                          memspace == NK_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, S->ctx->stream));
  return solver_start(S, false);
}

// PseudoTransient(; mass_matrix = Diagonal(m)) (pseudo_transient.jl:37-57,102-120): the damping term becomes α⁻¹ M. The matrix
// is part of the algorithm, fixed at init in the reference — set it before the first step (NULL returns to the identity).
extern "C" int nk_solver_set_mass_matrix_diagonal(nk_solver *S, const double *m, int memspace) {
  NK_REQUIRE(S, "NULL argument");
  NK_REQUIRE(is_pt(S), "a mass matrix belongs to PseudoTransient (algorithm = NK_ALG_PSEUDO_TRANSIENT)");
  NK_HIP(hipSetDevice(S->ctx->device));
  if (S->pt_applied != 0.0 && concrete(S)) {  // take the damping that is on the stored diagonal back off first
    NK_TRY(nk_csr_add_to_diagonal_dev(S->J, -S->pt_applied, S->pt_mass));
    S->pt_applied = 0.0;
    S->lu_valid = false;
  }
  if (m == nullptr) {
    NK_HIP(hipStreamSynchronize(S->ctx->stream));
    hipFree(S->pt_mass);
    S->pt_mass = nullptr;
  } else {
    if (!S->pt_mass) NK_TRY(nk_dev_alloc(&S->pt_mass, (size_t)S->n + 1));
    NK_HIP(hipMemcpyAsync(S->pt_mass, m, S->n * sizeof(double),
                          memspace == NK_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, S->ctx->stream));
    if (memspace != NK_DEVICE) NK_HIP(hipStreamSynchronize(S->ctx->stream));
  }
  if (S->G) NK_TRY(nk_gmres_set_shift_weights(S->G, S->pt_mass));
  return NK_OK;
}

static int copy_out(nk_solver *S, const double *src, double *dst, int memspace) {
  NK_HIP(hipMemcpyAsync(dst, src, S->n * sizeof(double),
                        memspace == NK_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, S->ctx->stream));
  NK_HIP(hipStreamSynchronize(S->ctx->stream));
  return NK_OK;
}
extern "C" int nk_solver_get_u(nk_solver *S, double *u, int memspace) {
  NK_REQUIRE(S && u, "NULL argument");
  return copy_out(S, S->u, u, memspace);
}
extern "C" int nk_solver_get_resid(nk_solver *S, double *f, int memspace) {
  NK_REQUIRE(S && f, "NULL argument");
  return copy_out(S, S->fu, f, memspace);
}
extern "C" int nk_solver_get_stats(nk_solver *S, nk_stats *st) {
  NK_REQUIRE(S && st, "NULL argument");
  *st = S->stats;
  st->op_applies = S->ctx->stats.op_applies - S->ctx_base.op_applies;  // since this cache was (re)initialised
  st->allreduces = S->ctx->stats.allreduces - S->ctx_base.allreduces;
  st->halo_exchanges = S->ctx->stats.halo_exchanges - S->ctx_base.halo_exchanges;
  return NK_OK;
}
extern "C" int nk_solver_get_retcode(nk_solver *S, int *retcode, int *nsteps, int *force_stop) {
  NK_REQUIRE(S, "NULL argument");
  if (retcode) *retcode = S->retcode;
  if (nsteps) *nsteps = S->nsteps;
  if (force_stop) *force_stop = S->force_stop ? 1 : 0;
  return NK_OK;
}
extern "C" int nk_solver_get_scalars(nk_solver *S, double *fnorm_inf, double *trust_region, double *eta) {
  NK_REQUIRE(S, "NULL argument");
  if (fnorm_inf) *fnorm_inf = S->fnorm_inf;
  if (trust_region) *trust_region = S->tr;
  if (eta) *eta = S->eta;
  return NK_OK;
}
extern "C" int nk_solver_get_trace(nk_solver *S, nk_trace_entry *rows, int capacity, int *nrows) {
  NK_REQUIRE(S && nrows, "NULL argument");
  *nrows = (int)S->trace.size();
  if (rows)
    for (int i = 0; i < capacity && i < *nrows; ++i) rows[i] = S->trace[i];
  return NK_OK;
}

extern "C" int nk_newton_solve(nk_problem *P, const double *u0, int memspace, const nk_options *opts, double *u_out,
                               double *resid_out, nk_stats *stats, int *retcode) {
  nk_solver *S = nullptr;
  NK_TRY(nk_solver_init(P, u0, memspace, opts, &S));
  int st = nk_solver_solve(S, retcode);
  if (st == NK_OK && u_out) st = nk_solver_get_u(S, u_out, memspace);
  if (st == NK_OK && resid_out) st = nk_solver_get_resid(S, resid_out, memspace);
  if (st == NK_OK && stats) st = nk_solver_get_stats(S, stats);
  nk_solver_destroy(S);
  return st;
}
