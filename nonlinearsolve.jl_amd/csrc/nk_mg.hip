// Geometric multigrid V-cycle as a right preconditioner for the Bratu Jacobian — what the reference reaches through
// `precs(A, p) -> (Pl, Pr)` with an AlgebraicMultigrid.jl preconditioner (docs/src/tutorials/large_systems.md:244-316;
// hook: test/Core/core_tests__item21.jl:10-18). The Chebyshev polynomial (nk_gmres.hip) removes the restart stagnation
// but its degree grows with the grid; a V-cycle keeps the Krylov iteration count mesh independent (7 at every size).
//
//   levels     n_l = n_{l−1}/2 interior points per side down to ≤ coarse_max; level operator by REDISCRETISATION:
//              J_l = scale·(Δ_{h_l} − λ diag(exp(u_l))) with the fine level's `scale` and u_l = R u_{l−1} — each level is a
//              Bratu problem object, so the matrix-free stencil JVP kernel serves every level
//   transfers  bilinear interpolation at the points' physical positions (the grids need not be nested: n = 1024 → 512 →
//              …); restriction = row-normalised transpose, evaluated as a gather (≤ 5×5 fine points per coarse point)
//   smoother   ν steps of the Chebyshev iteration on [λmax/4, λmax], λmax = 8·scale/h_l² (Gershgorin for the stencil)
//   coarsest   banded LU (nk_band.hip) of the assembled coarse Jacobian, refactored whenever u changes
// BRATU2D. Row-partitioned runs (several ranks): every level is partitioned by grid lines like the fine problem (the level
// problems are ordinary partitioned Bratu objects, so the smoother's stencil JVP brings its own halo exchange); the
// transfers are tensor products, so only their line direction crosses ranks — a rank gathers the few ghost lines of the
// other level it needs through a halo plan built once per transfer (the partitions of two levels do not line up exactly);
// the coarsest level is gathered to every rank (one all-reduce of a zero-padded vector) and solved redundantly.
// Restated on the CPU by oracle/reference_restatement.py::BratuMultigrid (the arithmetic does not depend on the partition).
//
// BRUSSELATOR2D (config C5): the same V-cycle for the two coupled species on the periodic N × N grid.
//   levels     N_l = N/2^l while even and > coarse_max (default 8); level operator by rediscretisation with spacing 2^l·dx at
//              the full-weighting restriction of the linearisation point — each level is a Brusselator problem object, so
//              the matrix-free JVP kernel serves every level
//   transfers  nested periodic grids: bilinear prolongation, full-weighting restriction (= ¼ Pᵀ), species by species
//   smoother   ν Chebyshev steps on −[λmax/4, λmax]: the Jacobian is diffusion dominated with a NEGATIVE spectrum;
//              λmax = 8α/dx_l² + A + 1 + 2·max|u|·max|v| + max|u|² bounds every Gershgorin disc of every level
//   coarsest   banded LU of the assembled coarse Jacobian (the reaction block may be indefinite there: solved exactly)
// Restated by oracle/reference_restatement.py::BrusselatorMultigrid.
#include <math.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "nk_internal.h"

struct nk_mg_level {
  int64_t ns = 0, n = 0;
  nk_problem *P = nullptr;   // Bratu problem of this level (level 0: the caller's, not owned)
  double *u = nullptr;       // linearisation point (level 0: the caller's vector, not owned)
  double *b = nullptr, *x = nullptr, *r = nullptr, *d = nullptr, *t = nullptr;
  double lmax = 0;
  // transfers to / from the next coarser level
  int32_t *pI0 = nullptr;    // per fine index: left coarse neighbour (−1 … nc−1)
  double *pw1 = nullptr;     // weight of the right neighbour
  int32_t *rlo = nullptr;    // per coarse index: first fine index of its support
  double *rw = nullptr;      // 5 normalised 1-D weights per coarse index
  // several ranks: lines owned on this level, and the ghost lines of the neighbouring levels the transfers read
  int64_t j0 = 0, j1 = 0;    // owned lines [j0, j1)
  nk_halo gh_c;              // prolongation: ghost lines of the COARSER level's vectors; needed coarse lines [pc_lo, pc_hi)
  int64_t pc_lo = 0, pc_hi = 0;
  nk_halo gh_f;              // restriction: ghost lines of THIS level's vectors; needed fine lines [rf_lo, rf_hi)
  int64_t rf_lo = 0, rf_hi = 0;
};
// a vector of one level seen through its owner's lines plus gathered ghost lines: line J (global) lives in `below`
// (lines [lo, own_lo)), `own` (lines [own_lo, own_hi)) or `above` (lines [own_hi, hi))
struct mg_lines {
  const double *below, *own, *above;
  int lo, hi;      // needed lines [lo, hi) (reads outside are clamped; the callers give them weight 0)
  int j0, j1;      // owned lines [j0, j1): `own` starts at line j0
  int a0;          // first ghost line above: max(lo, j1)
  int n;           // points per line
  __device__ __forceinline__ double at(int J, int I) const {
    const int Jc = J < lo ? lo : (J >= hi ? hi - 1 : J);
    if (Jc < j0) return below[(size_t)(Jc - lo) * n + I];
    if (Jc < j1) return own[(size_t)(Jc - j0) * n + I];
    return above[(size_t)(Jc - a0) * n + I];
  }
};

struct nk_mg {
  nk_ctx *ctx = nullptr;
  int kind = 0;              // 0 Bratu (Dirichlet, scalar), 1 Brusselator (periodic, two species)
  double base_alpha = 0, base_dx = 0, brus_A = 0;  // Brusselator: α, dx of level 0 and the reaction parameter A
  int nu = 2;
  std::vector<nk_mg_level> lv;
  nk_csr *Jc = nullptr;
  nk_bandlu *LU = nullptr;
  // several ranks: the coarsest level once more, whole on every rank (gathered right-hand side / iterate / solution)
  nk_problem *Prep = nullptr;
  double *rep_u = nullptr, *rep_b = nullptr, *rep_x = nullptr;
  // HIP graphs of the V-cycle body, one per early-exit flag pointer (nullptr / the GMRES control block's flag)
  hipStream_t cap_stream = nullptr;
  hipGraphExec_t gexec[2] = {nullptr, nullptr};
  const int *gskip[2] = {nullptr, nullptr};
  bool graph_broken = false;  // capture / instantiate failed once: plain launches from then on
};

// ----------------------------------------------------------------------------- kernels
#define MG_SKIP(p) if ((p) != nullptr && *(p) != 0) return
// x_f += P e_c   (bilinear, homogeneous Dirichlet: neighbours outside the coarse grid are zero)
__global__ __launch_bounds__(NK_BLOCK) void k_mg_prolong_add(int nf, int nc, const int32_t *__restrict__ I0,
                                                             const double *__restrict__ w1, const double *__restrict__ ec,
                                                             double *__restrict__ xf, const int *d_skip) {
  MG_SKIP(d_skip);
  const int k = blockIdx.x * NK_BLOCK + threadIdx.x;
  if (k >= nf * nf) return;
  const int j = k / nf, i = k - j * nf;
  const int Il = I0[i], Jl = I0[j];
  const double wi1 = w1[i], wi0 = 1.0 - wi1, wj1 = w1[j], wj0 = 1.0 - wj1;
  // clamped loads + 0/1 masks keep the four loads unconditional
  const int ia = Il < 0 ? 0 : Il, ib = Il + 1 >= nc ? nc - 1 : Il + 1;
  const int ja = Jl < 0 ? 0 : Jl, jb = Jl + 1 >= nc ? nc - 1 : Jl + 1;
  const double ma = Il >= 0 ? 1.0 : 0.0, mb = Il + 1 < nc ? 1.0 : 0.0, na = Jl >= 0 ? 1.0 : 0.0, nb = Jl + 1 < nc ? 1.0 : 0.0;
  const double v = wj0 * na * (wi0 * ma * ec[ja * nc + ia] + wi1 * mb * ec[ja * nc + ib]) +
                   wj1 * nb * (wi0 * ma * ec[jb * nc + ia] + wi1 * mb * ec[jb * nc + ib]);
  xf[k] += v;
}
// r_c = R r_f, R = row-normalised Pᵀ = (tensor product of the 1-D normalised weights)
__global__ __launch_bounds__(NK_BLOCK) void k_mg_restrict(int nf, int nc, const int32_t *__restrict__ lo,
                                                          const double *__restrict__ w, const double *__restrict__ rf,
                                                          double *__restrict__ rc, const int *d_skip) {
  MG_SKIP(d_skip);
  const int k = blockIdx.x * NK_BLOCK + threadIdx.x;
  if (k >= nc * nc) return;
  const int J = k / nc, I = k - J * nc;
  const int i0 = lo[I], j0 = lo[J];
  double s = 0.0;
#pragma unroll
  for (int b = 0; b < 5; ++b) {
    const int j = min(j0 + b, nf - 1);
    const double wj = w[J * 5 + b];
    double row = 0.0;
#pragma unroll
    for (int a = 0; a < 5; ++a) row += w[I * 5 + a] * rf[j * nf + min(i0 + a, nf - 1)];
    s += wj * row;
  }
  rc[k] = s;
}
// the same two transfers on a line-partitioned hierarchy: the thread grid covers the OWNED lines of the target level, the
// source level is read through mg_lines (owned + ghost lines)
__global__ __launch_bounds__(NK_BLOCK) void k_mg_prolong_add_dist(int nf, int nc, int jf0, int nlf, const int32_t *__restrict__ I0,
                                                                  const double *__restrict__ w1, mg_lines ec,
                                                                  double *__restrict__ xf, const int *d_skip) {
  MG_SKIP(d_skip);
  const int k = blockIdx.x * NK_BLOCK + threadIdx.x;
  if (k >= nf * nlf) return;
  const int jl = k / nf, i = k - jl * nf, j = jf0 + jl;
  const int Il = I0[i], Jl = I0[j];
  const double wi1 = w1[i], wi0 = 1.0 - wi1, wj1 = w1[j], wj0 = 1.0 - wj1;
  const int ia = Il < 0 ? 0 : Il, ib = Il + 1 >= nc ? nc - 1 : Il + 1;
  const int ja = Jl < 0 ? 0 : Jl, jb = Jl + 1 >= nc ? nc - 1 : Jl + 1;
  const double ma = Il >= 0 ? 1.0 : 0.0, mb = Il + 1 < nc ? 1.0 : 0.0, na = Jl >= 0 ? 1.0 : 0.0, nb = Jl + 1 < nc ? 1.0 : 0.0;
  const double v = wj0 * na * (wi0 * ma * ec.at(ja, ia) + wi1 * mb * ec.at(ja, ib)) +
                   wj1 * nb * (wi0 * ma * ec.at(jb, ia) + wi1 * mb * ec.at(jb, ib));
  xf[k] += v;
}
__global__ __launch_bounds__(NK_BLOCK) void k_mg_restrict_dist(int nf, int nc, int jc0, int nlc, const int32_t *__restrict__ lo,
                                                               const double *__restrict__ w, mg_lines rf,
                                                               double *__restrict__ rc, const int *d_skip) {
  MG_SKIP(d_skip);
  const int k = blockIdx.x * NK_BLOCK + threadIdx.x;
  if (k >= nc * nlc) return;
  const int Jl = k / nc, I = k - Jl * nc, J = jc0 + Jl;
  const int i0 = lo[I], j0 = lo[J];
  double s = 0.0;
#pragma unroll
  for (int b = 0; b < 5; ++b) {
    const int j = min(j0 + b, nf - 1);
    const double wj = w[J * 5 + b];
    double row = 0.0;
#pragma unroll
    for (int a = 0; a < 5; ++a) row += w[I * 5 + a] * rf.at(j, min(i0 + a, nf - 1));
    s += wj * row;
  }
  rc[k] = s;
}
// out[full coarse vector] = 0 except this rank's lines (the all-reduce that follows assembles the whole vector everywhere)
__global__ __launch_bounds__(NK_BLOCK) void k_mg_scatter_lines(int64_t nfull, int64_t off, int64_t cnt, const double *__restrict__ loc,
                                                               double *__restrict__ full) {
  const int64_t i = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x;
  if (i >= nfull) return;
  full[i] = (i >= off && i < off + cnt) ? loc[i - off] : 0.0;
}

// Chebyshev smoother: the first step; the later steps and the residual are fused into the stencil JVP's row epilogue
// d = r/θ ; x = (zero ? 0 : x) + d ; r_copy = r (optional: the recurrence continues on a private copy of b)
__global__ __launch_bounds__(NK_BLOCK) void k_mg_cheb_first(int64_t n, const double *__restrict__ r, double inv_theta, int zero,
                                                            double *__restrict__ d, double *__restrict__ x,
                                                            double *__restrict__ r_copy, const int *d_skip) {
  MG_SKIP(d_skip);
  const int64_t i = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x;
  if (i >= n) return;
  const double rv = r[i], dd = rv * inv_theta;
  d[i] = dd;
  x[i] = zero ? dd : x[i] + dd;
  if (r_copy) r_copy[i] = rv;
}
static inline dim3 g1(int64_t n) { return dim3((unsigned)((n + NK_BLOCK - 1) / NK_BLOCK)); }

// ---- Brusselator: nested periodic grids, two species stored one after the other (idx = i + N·j + N²·s)
// r_c = ¼ Pᵀ r_f (full weighting): 1/16 · [1 2 1; 2 4 2; 1 2 1] around the fine point (2I, 2J), periodic.
// Slabs of whole lines (several ranks; every rank's first line is even on every level but the coarsest): coarse local line Jl
// sits on fine local line 2 Jl; the line below fine line 0 is the lower ghost line `lo` ([species 0 | species 1], nullptr =
// wrap inside the slab), the line above 2 Jl + 1 is always owned.
__global__ __launch_bounds__(NK_BLOCK) void k_mgb_restrict(int nf, int nlf, const double *__restrict__ rf,
                                                           const double *__restrict__ lo, double *__restrict__ rc,
                                                           const int *d_skip) {
  MG_SKIP(d_skip);
  const int nc = nf >> 1, nlc = nlf >> 1;
  const int k = blockIdx.x * NK_BLOCK + threadIdx.x;
  if (k >= 2 * nc * nlc) return;
  const int sp = k / (nc * nlc), kk = k - sp * nc * nlc, J = kk / nc, I = kk - J * nc;
  const double *f = rf + (size_t)sp * nf * nlf;
  const int i = 2 * I, j = 2 * J;
  const int im = (i == 0) ? nf - 1 : i - 1, ip = i + 1, jp = j + 1;  // i, j even ⇒ i + 1 < nf, j + 1 < nlf
  const double *below = (j > 0) ? f + (size_t)(j - 1) * nf : (lo ? lo + (size_t)sp * nf : f + (size_t)(nlf - 1) * nf);
  const double c = f[j * nf + i];
  const double e = (f[j * nf + im] + f[j * nf + ip]) + (below[i] + f[jp * nf + i]);
  const double d = (below[im] + below[ip]) + (f[jp * nf + im] + f[jp * nf + ip]);
  rc[k] = (4.0 * c + 2.0 * e + d) * (1.0 / 16.0);
}
// x_f += P e_c (bilinear on the nested periodic grid); the coarse line above the slab is the upper ghost line `hi`
__global__ __launch_bounds__(NK_BLOCK) void k_mgb_prolong_add(int nf, int nlf, const double *__restrict__ ec,
                                                              const double *__restrict__ hi, double *__restrict__ xf,
                                                              const int *d_skip) {
  MG_SKIP(d_skip);
  const int nc = nf >> 1, nlc = nlf >> 1;
  const int k = blockIdx.x * NK_BLOCK + threadIdx.x;
  if (k >= 2 * nf * nlf) return;
  const int sp = k / (nf * nlf), kk = k - sp * nf * nlf, j = kk / nf, i = kk - j * nf;
  const double *c = ec + (size_t)sp * nc * nlc;
  const int I0 = i >> 1, J0 = j >> 1;
  const int I1 = (i & 1) ? ((I0 + 1 == nc) ? 0 : I0 + 1) : I0;
  // (an even index reads the same coarse point twice with weight ½ + ½ — the loads stay unconditional)
  const double *l0 = c + (size_t)J0 * nc;
  const double *l1 = (j & 1) ? ((J0 + 1 < nlc) ? c + (size_t)(J0 + 1) * nc : (hi ? hi + (size_t)sp * nc : c)) : l0;
  xf[k] += 0.25 * ((l0[I0] + l0[I1]) + (l1[I0] + l1[I1]));
}
// coarsest level on several ranks: this rank's lines of both species into the full (i, j, species) vector, zero elsewhere
__global__ __launch_bounds__(NK_BLOCK) void k_mgb_scatter_lines(int N, int j0, int nl, const double *__restrict__ loc,
                                                                double *__restrict__ full) {
  const int64_t e = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x;
  if (e >= 2ll * N * N) return;
  const int sp = (int)(e / ((int64_t)N * N));
  const int64_t kk = e - (int64_t)sp * N * N;
  const int j = (int)(kk / N), i = (int)(kk - (int64_t)j * N);
  full[e] = (j >= j0 && j < j0 + nl) ? loc[(size_t)i + (size_t)N * (j - j0) + (size_t)N * nl * sp] : 0.0;
}
__global__ __launch_bounds__(NK_BLOCK) void k_mgb_take_lines(int N, int j0, int nl, const double *__restrict__ full,
                                                             double *__restrict__ loc) {
  const int64_t e = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x;
  if (e >= 2ll * N * nl) return;
  const int sp = (int)(e / ((int64_t)N * nl));
  const int64_t kk = e - (int64_t)sp * N * nl;
  const int jl = (int)(kk / N), i = (int)(kk - (int64_t)jl * N);
  loc[e] = full[(size_t)i + (size_t)N * (j0 + jl) + (size_t)N * N * sp];
}
// one Chebyshev step after the first, unfused: r −= J d (t holds J d); d = c1 d + c2 r; x += d
__global__ __launch_bounds__(NK_BLOCK) void k_mg_cheb_step(int64_t n, double c1, double c2, const double *__restrict__ t,
                                                           double *__restrict__ r, double *__restrict__ d,
                                                           double *__restrict__ x, const int *d_skip) {
  MG_SKIP(d_skip);
  const int64_t i = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x;
  if (i >= n) return;
  const double rn = r[i] - t[i], dn = c1 * d[i] + c2 * rn;
  r[i] = rn;
  d[i] = dn;
  x[i] += dn;
}
// out = b − t
__global__ __launch_bounds__(NK_BLOCK) void k_mg_sub(int64_t n, const double *__restrict__ b, const double *__restrict__ t,
                                                     double *__restrict__ out, const int *d_skip) {
  MG_SKIP(d_skip);
  const int64_t i = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x;
  if (i < n) out[i] = b[i] - t[i];
}

// ----------------------------------------------------------------------------- setup
static void interp_tables(int nf, int nc, std::vector<int32_t> &I0, std::vector<double> &w1, std::vector<int32_t> &lo,
                          std::vector<double> &w) {
  const double h = 1.0 / (nf + 1), H = 1.0 / (nc + 1);
  I0.resize(nf);
  w1.resize(nf);
  for (int i = 0; i < nf; ++i) {
    const double t = (i + 1) * h / H;  // coarse coordinate (coarse interior point I, 1-based, sits at t = I)
    const int fl = (int)floor(t);
    I0[i] = fl - 1;                    // 0-based left neighbour; −1 / nc are the zero boundary values
    w1[i] = t - fl;
  }
  lo.assign(nc, 0);
  w.assign((size_t)nc * 5, 0.0);
  std::vector<double> sum(nc, 0.0);
  std::vector<int> first(nc, -1);
  for (int i = 0; i < nf; ++i)
    for (int side = 0; side < 2; ++side) {
      const int I = I0[i] + side;
      const double wt = side ? w1[i] : 1.0 - w1[i];
      if (I < 0 || I >= nc || wt == 0.0) continue;
      if (first[I] < 0) first[I] = i;
      const int a = i - first[I];
      if (a < 5) w[(size_t)I * 5 + a] += wt;
      sum[I] += wt;
    }
  for (int I = 0; I < nc; ++I) {
    lo[I] = first[I] < 0 ? 0 : first[I];
    for (int a = 0; a < 5; ++a) w[(size_t)I * 5 + a] /= (sum[I] > 0.0 ? sum[I] : 1.0);
  }
}

void nk_mg_destroy(nk_mg *M) {
  if (!M) return;
  for (size_t l = 0; l < M->lv.size(); ++l) {
    nk_mg_level &L = M->lv[l];
    if (l > 0) { nk_problem_destroy(L.P); hipFree(L.u); }
    hipFree(L.b); hipFree(L.x); hipFree(L.r); hipFree(L.d); hipFree(L.t);
    hipFree(L.pI0); hipFree(L.pw1); hipFree(L.rlo); hipFree(L.rw);
  }
  for (nk_mg_level &L : M->lv) { nk_halo_free(&L.gh_c); nk_halo_free(&L.gh_f); }
  if (M->LU) nk_bandlu_destroy(M->LU);
  if (M->Jc) nk_csr_destroy(M->Jc);
  if (M->Prep) nk_problem_destroy(M->Prep);
  hipFree(M->rep_u); hipFree(M->rep_b); hipFree(M->rep_x);
  for (hipGraphExec_t &g : M->gexec) if (g) hipGraphExecDestroy(g);
  if (M->cap_stream) hipStreamDestroy(M->cap_stream);
  delete M;
}

// build the hierarchy for problem P (level vectors, transfer tables, level problems); values follow in nk_mg_update
static mg_lines make_lines(const nk_halo &H, const double *own, int64_t lo, int64_t hi, int64_t j0, int64_t j1, int64_t n) {
  mg_lines m;
  const int64_t nb = std::max<int64_t>(0, std::min(hi, j0) - lo);
  m.below = H.d_recv;
  m.above = H.d_recv ? H.d_recv + nb * n : nullptr;
  m.own = own;
  m.lo = (int)lo; m.hi = (int)hi; m.j0 = (int)j0; m.j1 = (int)j1;
  m.a0 = (int)std::max(lo, j1);
  m.n = (int)n;
  return m;
}
// ghost lines [lo, hi) \ [j0, j1) of a level vector with `n` points per line, as global entry ids
static void ghost_needs(int64_t lo, int64_t hi, int64_t j0, int64_t j1, int64_t n, std::vector<int64_t> &needs) {
  needs.clear();
  for (int64_t J = lo; J < hi; ++J) {
    if (J >= j0 && J < j1) continue;
    for (int64_t I = 0; I < n; ++I) needs.push_back(J * n + I);
  }
}

// Brusselator hierarchy: see the file header. Several ranks: every level is split by lines like the fine problem; a level
// is coarsened only while every rank keeps an even number of lines (then slab boundaries stay even on both levels and the
// transfers need exactly the problem's own one-line halo: the line below the slab for the restriction, the coarse line above
// it for the prolongation); the coarsest level is gathered to every rank and solved redundantly.
static int mgb_create(nk_problem *P, int nu, int coarse_max, nk_mg **out) {
  nk_ctx *ctx = P->ctx;
  const int R = ctx->nranks;
  NK_REQUIRE(P->ns < 23000, "grid too large for 32-bit point indices");
  if (nu <= 0) nu = 2;
  if (coarse_max < 4) coarse_max = 8;
  nk_mg *M = new nk_mg();
  auto guard = nk_make_guard(M, [](nk_mg *g) { nk_mg_destroy(g); });
  M->ctx = ctx;
  M->kind = 1;
  M->nu = nu;
  M->brus_A = P->params[1];
  M->base_alpha = P->params[3];
  M->base_dx = P->params[4];
  int64_t N = P->ns;
  double dx = P->params[4];
  for (int l = 0;; ++l) {
    nk_mg_level L;
    L.ns = N;
    if (l == 0) L.P = P;
    else {
      const double par[5] = {(double)N, P->params[1], P->params[2], P->params[3], dx};
      if (nk_problem_create(ctx, NK_PROBLEM_BRUSSELATOR2D, par, 5, &L.P) != NK_OK) return NK_E_HIP;
    }
    L.j0 = L.P->j0;
    L.j1 = L.P->j1;
    L.n = 2 * N * (L.j1 - L.j0);
    if (l > 0) NK_TRY(nk_dev_alloc(&L.u, (size_t)L.n + 1));
    for (double **b : {&L.b, &L.x, &L.r, &L.d, &L.t}) NK_TRY(nk_dev_alloc(b, (size_t)L.n + 1));
    // coarsen while the grid halves evenly — on several ranks: while every rank's slab does (N divisible by 2 R, and the
    // coarse level still has two lines per rank)
    bool coarsest = N <= coarse_max || (N % 2) != 0 || N / 2 < 4;
    if (R > 1 && !coarsest) coarsest = (N % (2 * R)) != 0 || N / 2 < 2 * R || (L.j0 % 2) != 0 || ((L.j1 - L.j0) % 2) != 0;
    M->lv.push_back(L);
    if (coarsest) break;
    N /= 2;
    dx *= 2.0;
  }
  if (M->lv.size() > 1) {
    nk_mg_level &C = M->lv.back();
    if (R > 1) {
      const double par[5] = {(double)C.ns, P->params[1], P->params[2], P->params[3], dx};
      NK_TRY(nk_problem_create_brus_replicated(ctx, par, &M->Prep));
      const size_t nfull = (size_t)(2 * C.ns * C.ns) + 1;
      NK_TRY(nk_dev_alloc(&M->rep_u, nfull));
      NK_TRY(nk_dev_alloc(&M->rep_b, nfull));
      NK_TRY(nk_dev_alloc(&M->rep_x, nfull));
      NK_TRY(nk_problem_jac_csr(M->Prep, &M->Jc));
    } else {
      NK_TRY(nk_problem_jac_csr(C.P, &M->Jc));
    }
    NK_TRY(nk_bandlu_create(M->Jc, &M->LU, 1));  // coarsest level: a handful of block columns — the band LU
  }
  *out = guard.release();
  return NK_OK;
}

int nk_mg_create(nk_problem *P, int nu, int coarse_max, nk_mg **out) {
  nk_ctx *ctx = P->ctx;
  const int R = ctx->nranks;
  if (P->kind == NK_PROBLEM_BRUSSELATOR2D) return mgb_create(P, nu, coarse_max, out);
  NK_REQUIRE(P->kind == NK_PROBLEM_BRATU2D, "the multigrid preconditioner is built for BRATU2D and BRUSSELATOR2D problems");
  NK_REQUIRE(P->ns < 46000, "grid too large for 32-bit point indices");
  if (nu <= 0) nu = 2;
  if (coarse_max < 3) coarse_max = 31;  // MI355X, 1024² with set-up: 63 → 13.5 ms, 31 → 10.1, 15 → 9.5, 7 → 9.7
  nk_mg *M = new nk_mg();
  auto guard = nk_make_guard(M, [](nk_mg *g) { nk_mg_destroy(g); });
  M->ctx = ctx;
  M->nu = nu;
  const double hf = 1.0 / (double)(P->ns + 1);
  const double scale = P->c_lap * hf * hf, lambda = P->params[1];
  const int64_t min_coarse = std::max<int64_t>(3, 2 * (int64_t)R);  // every rank keeps at least two lines of every level
  std::vector<std::vector<int32_t>> hI0, hlo;  // host copies of the 1-D transfer tables (ghost-line ranges, several ranks)
  int64_t ns = P->ns;
  for (int l = 0;; ++l) {
    nk_mg_level L;
    L.ns = ns;
    const double h = 1.0 / (double)(ns + 1);
    L.lmax = 8.0 * scale / (h * h);
    if (l == 0) L.P = P;
    else {
      const double par[3] = {(double)ns, lambda, scale};
      if (nk_problem_create(ctx, NK_PROBLEM_BRATU2D, par, 3, &L.P) != NK_OK) return NK_E_HIP;
    }
    L.j0 = L.P->j0;
    L.j1 = L.P->j1;
    L.n = ns * (L.j1 - L.j0);  // local entries (all of them on one rank)
    if (l > 0) NK_TRY(nk_dev_alloc(&L.u, (size_t)L.n + 1));
    NK_TRY(nk_dev_alloc(&L.b, (size_t)L.n + 1));
    NK_TRY(nk_dev_alloc(&L.x, (size_t)L.n + 1));
    NK_TRY(nk_dev_alloc(&L.r, (size_t)L.n + 1));
    NK_TRY(nk_dev_alloc(&L.d, (size_t)L.n + 1));
    NK_TRY(nk_dev_alloc(&L.t, (size_t)L.n + 1));
    const bool coarsest = ns <= coarse_max || ns / 2 < min_coarse;
    hI0.emplace_back();
    hlo.emplace_back();
    if (!coarsest) {
      const int nf = (int)ns, nc = (int)(ns / 2);
      std::vector<int32_t> I0, lo;
      std::vector<double> w1, w;
      interp_tables(nf, nc, I0, w1, lo, w);
      NK_TRY(nk_dev_alloc(&L.pI0, (size_t)nf));
      NK_TRY(nk_dev_alloc(&L.pw1, (size_t)nf));
      NK_TRY(nk_dev_alloc(&L.rlo, (size_t)nc));
      NK_TRY(nk_dev_alloc(&L.rw, (size_t)nc * 5));
      NK_HIP(nk_memcpy(ctx, L.pI0, I0.data(), nf * sizeof(int32_t), hipMemcpyHostToDevice));
      NK_HIP(nk_memcpy(ctx, L.pw1, w1.data(), nf * sizeof(double), hipMemcpyHostToDevice));
      NK_HIP(nk_memcpy(ctx, L.rlo, lo.data(), nc * sizeof(int32_t), hipMemcpyHostToDevice));
      NK_HIP(nk_memcpy(ctx, L.rw, w.data(), (size_t)nc * 5 * sizeof(double), hipMemcpyHostToDevice));
      hI0.back() = I0;
      hlo.back() = lo;
    }
    M->lv.push_back(L);
    if (coarsest) break;
    ns /= 2;
  }
  if (R > 1) {  // ghost-line plans of the transfers (collective; the same sequence on every rank)
    for (size_t l = 0; l + 1 < M->lv.size(); ++l) {
      nk_mg_level &F = M->lv[l], &C = M->lv[l + 1];
      const int64_t nf = F.ns, nc = C.ns;
      // prolongation: my fine lines read the coarse lines I0[j], I0[j] + 1
      int64_t lo = nc, hi = 0;
      for (int64_t j = F.j0; j < F.j1; ++j) {
        lo = std::min<int64_t>(lo, std::max<int64_t>(0, hI0[l][j]));
        hi = std::max<int64_t>(hi, std::min<int64_t>(nc, (int64_t)hI0[l][j] + 2));
      }
      if (hi < lo) hi = lo;
      F.pc_lo = lo;
      F.pc_hi = hi;
      std::vector<int64_t> needs;
      ghost_needs(lo, hi, C.j0, C.j1, nc, needs);
      NK_TRY(nk_halo_build_from_needs(ctx, C.j0 * nc, (C.j1 - C.j0) * nc, needs, &F.gh_c));
      // restriction: my coarse lines read the fine lines lo[J] … lo[J] + 4 (clamped)
      lo = nf;
      hi = 0;
      for (int64_t J = C.j0; J < C.j1; ++J) {
        lo = std::min<int64_t>(lo, hlo[l][J]);
        hi = std::max<int64_t>(hi, std::min<int64_t>(nf - 1, (int64_t)hlo[l][J] + 4) + 1);
      }
      if (hi < lo) hi = lo;
      F.rf_lo = lo;
      F.rf_hi = hi;
      ghost_needs(lo, hi, F.j0, F.j1, nf, needs);
      NK_TRY(nk_halo_build_from_needs(ctx, F.j0 * nf, (F.j1 - F.j0) * nf, needs, &F.gh_f));
    }
  }
  nk_mg_level &C = M->lv.back();
  if (M->lv.size() > 1) {
    if (R > 1) {
      NK_TRY(nk_problem_create_bratu_replicated(ctx, C.ns, lambda, scale, &M->Prep));
      NK_TRY(nk_dev_alloc(&M->rep_u, (size_t)(C.ns * C.ns) + 1));
      NK_TRY(nk_dev_alloc(&M->rep_b, (size_t)(C.ns * C.ns) + 1));
      NK_TRY(nk_dev_alloc(&M->rep_x, (size_t)(C.ns * C.ns) + 1));
      NK_TRY(nk_problem_jac_csr(M->Prep, &M->Jc));
    } else {
      NK_TRY(nk_problem_jac_csr(C.P, &M->Jc));
    }
    NK_TRY(nk_bandlu_create(M->Jc, &M->LU, 1));  // coarsest level: a handful of block columns — the band LU
  }
  *out = guard.release();
  return NK_OK;
}

// every rank's lines of the coarsest level → the whole vector on every rank (zero-padded scatter + one all-reduce)
static int mg_gather_coarse(nk_mg *M, const double *loc, double *full) {
  nk_ctx *ctx = M->ctx;
  nk_mg_level &C = M->lv.back();
  const int64_t nfull = (M->kind == 1 ? 2 : 1) * C.ns * C.ns;
  if (M->kind == 1)
    NK_LAUNCH(ctx, k_mgb_scatter_lines, g1(nfull), dim3(NK_BLOCK), (int)C.ns, (int)C.j0, (int)(C.j1 - C.j0), loc, full);
  else
    NK_LAUNCH(ctx, k_mg_scatter_lines, g1(nfull), dim3(NK_BLOCK), nfull, C.j0 * C.ns, C.n, loc, full);
  NK_HIP(hipGetLastError());
  for (int64_t o = 0; o < nfull; o += 1 << 20) {  // (int-sized messages)
    const int c = (int)std::min<int64_t>(1 << 20, nfull - o);
    NK_TRY(nk_comm_allreduce(ctx, full + o, c, 0));
  }
  return NK_OK;
}
// transfers: F level → coarser level C (restriction of `src`, a level-F vector) and back
static int mg_restrict(nk_mg *M, nk_mg_level &F, nk_mg_level &C, const double *src, double *dst, const int *d_skip) {
  nk_ctx *ctx = M->ctx;
  if (M->kind == 1) {
    const double *lo = nullptr, *hi = nullptr;
    if (ctx->nranks > 1) NK_TRY(nk_problem_ghost_lines(F.P, src, &lo, &hi));
    NK_LAUNCH(ctx, k_mgb_restrict, g1(C.n), dim3(NK_BLOCK), (int)F.ns, (int)(F.j1 - F.j0), src, lo, dst, d_skip);
  } else if (ctx->nranks == 1) {
    NK_LAUNCH(ctx, k_mg_restrict, g1(C.n), dim3(NK_BLOCK), (int)F.ns, (int)C.ns, (const int32_t *)F.rlo, (const double *)F.rw, src,
              dst, d_skip);
  } else {
    NK_TRY(nk_halo_exchange(ctx, &F.gh_f, src));
    const mg_lines ln = make_lines(F.gh_f, src, F.rf_lo, F.rf_hi, F.j0, F.j1, F.ns);
    NK_LAUNCH(ctx, k_mg_restrict_dist, g1(C.n), dim3(NK_BLOCK), (int)F.ns, (int)C.ns, (int)C.j0, (int)(C.j1 - C.j0),
              (const int32_t *)F.rlo, (const double *)F.rw, ln, dst, d_skip);
  }
  NK_HIP(hipGetLastError());
  return NK_OK;
}
static int mg_prolong_add(nk_mg *M, nk_mg_level &F, nk_mg_level &C, const double *ec, double *xf, const int *d_skip) {
  nk_ctx *ctx = M->ctx;
  if (M->kind == 1) {
    const double *lo = nullptr, *hi = nullptr;
    if (ctx->nranks > 1) NK_TRY(nk_problem_ghost_lines(C.P, ec, &lo, &hi));
    NK_LAUNCH(ctx, k_mgb_prolong_add, g1(F.n), dim3(NK_BLOCK), (int)F.ns, (int)(F.j1 - F.j0), ec, hi, xf, d_skip);
  } else if (ctx->nranks == 1) {
    NK_LAUNCH(ctx, k_mg_prolong_add, g1(F.n), dim3(NK_BLOCK), (int)F.ns, (int)C.ns, (const int32_t *)F.pI0, (const double *)F.pw1, ec,
              xf, d_skip);
  } else {
    NK_TRY(nk_halo_exchange(ctx, &F.gh_c, ec));
    const mg_lines ln = make_lines(F.gh_c, ec, F.pc_lo, F.pc_hi, C.j0, C.j1, C.ns);
    NK_LAUNCH(ctx, k_mg_prolong_add_dist, g1(F.n), dim3(NK_BLOCK), (int)F.ns, (int)C.ns, (int)F.j0, (int)(F.j1 - F.j0),
              (const int32_t *)F.pI0, (const double *)F.pw1, ln, xf, d_skip);
  }
  NK_HIP(hipGetLastError());
  return NK_OK;
}

// new linearisation point: restrict u down the hierarchy, refresh exp(u_l) of every level, refactor the coarsest J
int nk_mg_update(nk_mg *M, const double *d_u) {
  M->lv[0].u = const_cast<double *>(d_u);
  if (M->kind == 1) {  // λmax of every level: 8α/dx_l² + the reaction rows' Gershgorin bound at the fine linearisation point
    nk_ctx *ctx = M->ctx;
    const int64_t nn = M->lv[0].n / 2;                        // this rank's entries of one species
    NK_TRY(nk_blas_minmax(ctx, nn, d_u, ctx->d_scal));        // (max, −min) of u (all-reduced)
    NK_TRY(nk_blas_minmax(ctx, nn, d_u + nn, ctx->d_scal + 2));  // … of v
    double v[4];
    NK_TRY(nk_scalars_to_host(ctx, ctx->d_scal, 4, v));
    const double U = fmax(fabs(v[0]), fabs(v[1])), V = fmax(fabs(v[2]), fabs(v[3]));
    const double rb = M->brus_A + 1.0 + 2.0 * U * V + U * U;
    double dx = M->base_dx;
    for (nk_mg_level &L : M->lv) {
      L.lmax = -(8.0 * M->base_alpha / (dx * dx) + rb);
      dx *= 2.0;
    }
  }
  for (size_t l = 0; l + 1 < M->lv.size(); ++l) {
    nk_mg_level &F = M->lv[l], &Cc = M->lv[l + 1];
    NK_TRY(mg_restrict(M, F, Cc, F.u, Cc.u, nullptr));
  }
  for (size_t l = 1; l < M->lv.size(); ++l) NK_TRY(nk_problem_jvp_prepare(M->lv[l].P, M->lv[l].u));
  if (M->LU) {
    nk_mg_level &C = M->lv.back();
    if (M->Prep) {  // several ranks: the coarsest iterate whole on every rank, Jacobian and factorisation redundantly
      NK_TRY(mg_gather_coarse(M, C.u, M->rep_u));
      NK_TRY(nk_problem_jac_values_dev(M->Prep, M->rep_u, M->Jc));
    } else {
      NK_TRY(nk_problem_jac_values_dev(C.P, C.u, M->Jc));
    }
    int ok = 0;
    NK_TRY(nk_bandlu_factor(M->LU, M->Jc, &ok));
    NK_REQUIRE(ok, "multigrid: the coarsest Jacobian has a zero pivot");
  }
  NK_HIP(hipGetLastError());
  return NK_OK;
}

// ν Chebyshev steps for J_l x = b on [λmax/4, λmax]; zero_guess: x starts at 0 (then r = b)
// b − J x in ONE kernel: the stencil JVP's row epilogue subtracts from b (nk_spmv_epi mode 2)
static int mg_residual(nk_mg *M, nk_mg_level &L, const double *x, double *out, const int *d_skip) {
  if (M->kind == 1) {  // (the Brusselator JVP kernel has no row epilogue: J x, then b − J x)
    NK_TRY(nk_problem_jvp_dev(L.P, L.u, x, out, d_skip));
    NK_LAUNCH(M->ctx, k_mg_sub, g1(L.n), dim3(NK_BLOCK), L.n, (const double *)L.b, (const double *)out, out, d_skip);
    NK_HIP(hipGetLastError());
    return NK_OK;
  }
  nk_spmv_epi ep;
  ep.mode = 2;
  ep.r = L.b;
  return nk_problem_jvp_dev(L.P, L.u, x, out, d_skip, nullptr, &ep);
}

static int mg_smooth(nk_mg *M, nk_mg_level &L, bool zero_guess, const int *d_skip) {
  nk_ctx *ctx = M->ctx;
  const double lmax = L.lmax, lmin = lmax / 4.0;
  const double theta = 0.5 * (lmax + lmin), delta = 0.5 * (lmax - lmin), sigma = theta / delta;
  double rho = 1.0 / sigma;
  if (!zero_guess) {
    NK_TRY(mg_residual(M, L, L.x, L.r, d_skip));
    NK_LAUNCH(ctx, k_mg_cheb_first, g1(L.n), dim3(NK_BLOCK), L.n, (const double *)L.r, 1.0 / theta, 0, L.d, L.x,
              (double *)nullptr, d_skip);
  } else {  // x0 = 0: r = b; the recurrence below updates r in place, so it gets its own copy in the same sweep
    NK_LAUNCH(ctx, k_mg_cheb_first, g1(L.n), dim3(NK_BLOCK), L.n, (const double *)L.b, 1.0 / theta, 1, L.d, L.x,
              M->nu > 1 ? L.r : (double *)nullptr, d_skip);
  }
  if (M->kind == 1) {
    for (int k = 1; k < M->nu; ++k) {
      const double rho_new = 1.0 / (2.0 * sigma - rho);
      NK_TRY(nk_problem_jvp_dev(L.P, L.u, L.d, L.t, d_skip));
      NK_LAUNCH(ctx, k_mg_cheb_step, g1(L.n), dim3(NK_BLOCK), L.n, rho_new * rho, 2.0 * rho_new / delta, (const double *)L.t,
                L.r, L.d, L.x, d_skip);
      rho = rho_new;
    }
    NK_HIP(hipGetLastError());
    return NK_OK;
  }
  double *dcur = L.d, *dalt = L.t;
  for (int k = 1; k < M->nu; ++k) {
    // r −= J d ; d_new = c1 d + c2 r ; x += d_new — fused into the JVP's row epilogue (mode 1); d ping-pongs because the
    // stencil reads the neighbours' old d
    const double rho_new = 1.0 / (2.0 * sigma - rho);
    nk_spmv_epi ep;
    ep.mode = 1;
    ep.c1 = rho_new * rho;
    ep.c2 = 2.0 * rho_new / delta;
    ep.r = L.r;
    ep.dnew = dalt;
    ep.yacc = L.x;
    NK_TRY(nk_problem_jvp_dev(L.P, L.u, dcur, dalt, d_skip, nullptr, &ep));
    double *tmp = dcur; dcur = dalt; dalt = tmp;
    rho = rho_new;
  }
  return NK_OK;
}

// the V-cycle proper: lv[0].b → lv[0].x (every pointer and scalar argument is fixed for the lifetime of the hierarchy)
static int mg_vcycle_body(nk_mg *M, const int *d_skip);

// dst = V-cycle(src). The body is ≈ 70 small launches whose arguments never change (only the DATA behind the level
// vectors, exp(u_l) and the coarse LU factors do), so it is captured once into a HIP graph per flag pointer and replayed:
// host launch work per kernel disappears (measured gain on MI355X is small — 5.9 → 5.7 ms for the 1024² solve — because the
// coarse-level kernels sit at the ≈ 5 µs execution floor either way). NK_MG_GRAPH=0 keeps the plain launches; the
// profiled pass (hipExtLaunchKernelGGL with events) always uses them. Capture runs on a private stream because the
// context's stream may be the legacy default stream, which cannot be captured.
int nk_mg_apply(nk_mg *M, const double *src, double *dst, const int *d_skip) {
  nk_ctx *ctx = M->ctx;
  const int nl = (int)M->lv.size();
  if (nl == 1) {  // a single (small) level: smoothing only
    nk_mg_level &L = M->lv[0];
    NK_TRY(nk_blas_copy(ctx, L.n, src, L.b));
    NK_TRY(mg_smooth(M, L, true, d_skip));
    return nk_blas_copy(ctx, L.n, L.x, dst);
  }
  NK_TRY(nk_blas_copy(ctx, M->lv[0].n, src, M->lv[0].b));
  static const bool use_graph = !(getenv("NK_MG_GRAPH") && atoi(getenv("NK_MG_GRAPH")) == 0);
  if (use_graph && !ctx->prof.on && !M->graph_broken && ctx->nranks == 1 && M->kind == 0) {  // (Brusselator: λmax moves with u)  // (collectives carry per-call sequence numbers)
    const int slot = d_skip ? 1 : 0;
    if (M->gexec[slot] && M->gskip[slot] != d_skip) {  // a different flag pointer: re-capture
      hipGraphExecDestroy(M->gexec[slot]);
      M->gexec[slot] = nullptr;
    }
    if (!M->gexec[slot]) {
      if (!M->cap_stream && hipStreamCreateWithFlags(&M->cap_stream, hipStreamNonBlocking) != hipSuccess) M->graph_broken = true;
      if (!M->graph_broken) {
        NK_HIP(hipStreamSynchronize(ctx->stream));  // the capture stream is not ordered behind the context's stream
        hipStream_t user_stream = ctx->stream;
        ctx->stream = M->cap_stream;
        hipGraph_t graph = nullptr;
        hipError_t e = hipStreamBeginCapture(M->cap_stream, hipStreamCaptureModeThreadLocal);
        const int st = (e == hipSuccess) ? mg_vcycle_body(M, d_skip) : NK_E_HIP;
        if (e == hipSuccess) e = hipStreamEndCapture(M->cap_stream, &graph);
        ctx->stream = user_stream;
        if (st == NK_OK && e == hipSuccess && graph) e = hipGraphInstantiate(&M->gexec[slot], graph, nullptr, nullptr, 0);
        if (graph) hipGraphDestroy(graph);
        if (st != NK_OK || e != hipSuccess || !M->gexec[slot]) {  // no graphs on this runtime: plain launches from now on
          M->gexec[slot] = nullptr;
          M->graph_broken = true;
          (void)hipGetLastError();
        }
        M->gskip[slot] = d_skip;
      }
    }
    if (M->gexec[slot]) NK_HIP(hipGraphLaunch(M->gexec[slot], ctx->stream));
    else NK_TRY(mg_vcycle_body(M, d_skip));
  } else {
    NK_TRY(mg_vcycle_body(M, d_skip));
  }
  return nk_blas_copy(ctx, M->lv[0].n, M->lv[0].x, dst);
}

static int mg_vcycle_body(nk_mg *M, const int *d_skip) {
  nk_ctx *ctx = M->ctx;
  const int nl = (int)M->lv.size();
  for (int l = 0; l + 1 < nl; ++l) {
    nk_mg_level &F = M->lv[l], &C = M->lv[l + 1];
    NK_TRY(mg_smooth(M, F, true, d_skip));
    NK_TRY(mg_residual(M, F, F.x, F.r, d_skip));
    NK_TRY(mg_restrict(M, F, C, F.r, C.b, d_skip));
  }
  {
    nk_mg_level &C = M->lv[nl - 1];
    if (M->Prep) {
      NK_TRY(mg_gather_coarse(M, C.b, M->rep_b));
      NK_TRY(nk_bandlu_solve(M->LU, M->rep_b, M->rep_x));
      if (M->kind == 1)
        NK_LAUNCH(ctx, k_mgb_take_lines, g1(C.n), dim3(NK_BLOCK), (int)C.ns, (int)C.j0, (int)(C.j1 - C.j0),
                  (const double *)M->rep_x, C.x);
      else
        NK_TRY(nk_blas_copy(ctx, C.n, M->rep_x + C.j0 * C.ns, C.x));
    } else {
      NK_TRY(nk_bandlu_solve(M->LU, C.b, C.x));
    }
  }
  for (int l = nl - 2; l >= 0; --l) {
    nk_mg_level &F = M->lv[l], &C = M->lv[l + 1];
    NK_TRY(mg_prolong_add(M, F, C, C.x, F.x, d_skip));
    NK_TRY(mg_smooth(M, F, false, d_skip));
  }
  NK_HIP(hipGetLastError());
  return NK_OK;
}

int nk_mg_levels(const nk_mg *M) { return (int)M->lv.size(); }
