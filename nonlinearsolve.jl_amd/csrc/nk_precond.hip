// Preconditioner objects built from a CSR matrix — what a `precs(A, p)` hook returns for a general sparse Jacobian
// (the reference: lib/NonlinearSolveBase/src/linear_solve.jl:195-199; test/Core/core_tests__item21.jl:10-18;
// docs/src/tutorials/large_systems.md:252-316, where the slots are filled with IncompleteLU.ilu(W) and an algebraic
// multigrid as `Pl`). Usable on either side of the device GMRES (nk_gmres_set_preconditioner) and standalone (nk_precond_apply).
//
//   NK_PRECOND_JACOBI   M = diag(A)
//   NK_PRECOND_ILU0     A ≈ L U on the pattern of A (no fill, no pivoting; L unit lower), of the rank's LOCAL square block
//                       (halo columns are dropped: block-Jacobi ILU(0) across ranks — no communication in the apply).
//   NK_PRECOND_AMG      aggregation algebraic multigrid (nk_amg.hip) — the scalable one of the three.
//
// ILU(0) on a GPU is a scheduling problem: row i of the factorisation (and of both triangular solves) can start when the
// rows it refers to are done. Rows are grouped into LEVELS (level(i) = 1 + max level of the rows i depends on); a level is a
// data-parallel kernel over its rows (one thread per row, the row's entries in CSR order: results equal a sequential sweep
// bit for bit). Two orderings:
//   NK_ILU_NATURAL     the matrix as it is. For a lexicographic 5-point stencil the levels are the anti-diagonals of the grid:
//                      2n − 1 levels of ≤ n rows — a dependency chain, not a parallel workload. Such schedules (many narrow
//                      levels) run inside ONE persistent workgroup that walks the levels with a barrier in between (≈ 1 µs per
//                      level instead of a ≈ 5 µs launch): the classical preconditioner, exact parity with a sequential ILU(0),
//                      but milliseconds per application at n = 1024².
//   NK_ILU_MULTICOLOR  rows permuted by a greedy distance-1 colouring of the pattern (red–black for the 5-point stencil):
//                      as many levels as colours (2–8), each level one wide launch — the GPU form of ILU(0) (weaker than the
//                      natural ordering by a constant factor in Krylov iterations, two orders of magnitude faster to apply).
// The symbolic phase (permutation, levels, the update plan of the IKJ factorisation: which entry of row k meets which entry
// of row i) runs once per pattern on the host; nk_precond_update refactorises for the matrix's current values on the device.
#include "nk_internal.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>
#include <vector>

struct nk_precond {
  nk_ctx *ctx = nullptr;
  int kind = 0;
  nk_csr *A = nullptr;
  int64_t n = 0;
  // Jacobi
  double *d_dinv = nullptr;
  // ILU(0)
  int ordering = 0, ncolors = 0;
  int64_t nnzp = 0;
  int32_t *d_perm = nullptr;                                 // permuted row → original row
  int32_t *d_rp = nullptr, *d_ci = nullptr, *d_dg = nullptr;  // permuted local block: rowptr, sorted columns, diagonal position
  int32_t *d_src = nullptr;                                  // position of every entry in A->d_val
  double *d_lu = nullptr;
  int32_t *d_planptr = nullptr, *d_planq = nullptr, *d_plans = nullptr;
  int32_t *d_rowsL = nullptr, *d_ptrL = nullptr, *d_rowsU = nullptr, *d_ptrU = nullptr;
  std::vector<int32_t> h_ptrL, h_ptrU;
  bool chainL = false, chainU = false;
  double *d_y = nullptr, *d_z = nullptr, *d_xin = nullptr, *d_xout = nullptr;
  int *d_fail = nullptr;
  bool factored = false;
  // ILU(τ): the drop tolerance; the factors' pattern is a function of the values, so every update re-plans (ilut_update)
  double tau = 0.0;
  // algebraic multigrid (nk_amg.hip)
  struct nk_amg *amg = nullptr;
};

// ----------------------------------------------------------------------------- Jacobi
__global__ __launch_bounds__(NK_BLOCK) void k_jacobi_setup(int64_t n, const int32_t *__restrict__ rowptr,
                                                           const int32_t *__restrict__ col, const double *__restrict__ val,
                                                           double *__restrict__ dinv, int *fail) {
  const int64_t stride = (int64_t)gridDim.x * NK_BLOCK;
  for (int64_t r = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x; r < n; r += stride) {
    double d = 0.0;
    for (int32_t p = rowptr[r]; p < rowptr[r + 1]; ++p)
      if (col[p] == r) d += val[p];
    if (d == 0.0 || d != d) *fail = 1;
    dinv[r] = 1.0 / d;
  }
}
__global__ __launch_bounds__(NK_BLOCK) void k_jacobi_apply(int64_t n, const double *__restrict__ dinv,
                                                           const double *__restrict__ x, double *__restrict__ y,
                                                           const int *d_skip) {
  if (d_skip != nullptr && *d_skip != 0) return;
  const int64_t stride = (int64_t)gridDim.x * NK_BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x; i < n; i += stride) y[i] = dinv[i] * x[i];
}

// ----------------------------------------------------------------------------- ILU(0): device kernels
// IKJ factorisation of row i: for every entry (i, k), k < i, in ascending k: l = a_ik / u_kk; a_ij −= l·u_kj for the j > k that
// both rows hold (the plan lists those pairs). Same operations in the same order as the sequential algorithm.
__device__ __forceinline__ void ilu_factor_row(int i, const int32_t *__restrict__ rp, const int32_t *__restrict__ ci,
                                               const int32_t *__restrict__ dg, double *lu,
                                               const int32_t *__restrict__ planptr, const int32_t *__restrict__ planq,
                                               const int32_t *__restrict__ plans, int *fail) {
  for (int32_t p = rp[i]; p < dg[i]; ++p) {
    const double piv = lu[dg[ci[p]]];
    const double l = lu[p] / piv;
    lu[p] = l;
    for (int32_t e = planptr[p]; e < planptr[p + 1]; ++e) lu[planq[e]] -= l * lu[plans[e]];
  }
  const double d = lu[dg[i]];
  if (d == 0.0 || d != d) *fail = 1;
}
// forward substitution with the unit lower factor: y_i = b_i − Σ_{k<i} l_ik y_k (b gathered through the permutation)
__device__ __forceinline__ void ilu_lower_row(int i, const int32_t *__restrict__ rp, const int32_t *__restrict__ ci,
                                              const int32_t *__restrict__ dg, const double *__restrict__ lu,
                                              const int32_t *__restrict__ perm, const double *__restrict__ b, double *y) {
  double s = b[perm ? perm[i] : i];
  for (int32_t p = rp[i]; p < dg[i]; ++p) s -= lu[p] * y[ci[p]];
  y[i] = s;
}
// backward substitution: z_i = (y_i − Σ_{j>i} u_ij z_j) / u_ii; the result also goes to out[perm[i]]
__device__ __forceinline__ void ilu_upper_row(int i, const int32_t *__restrict__ rp, const int32_t *__restrict__ ci,
                                              const int32_t *__restrict__ dg, const double *__restrict__ lu,
                                              const int32_t *__restrict__ perm, const double *__restrict__ y, double *z,
                                              double *out) {
  double s = y[i];
  for (int32_t p = dg[i] + 1; p < rp[i + 1]; ++p) s -= lu[p] * z[ci[p]];
  s /= lu[dg[i]];
  z[i] = s;
  out[perm ? perm[i] : i] = s;
}

struct ilu_dev {
  const int32_t *rp, *ci, *dg, *perm, *planptr, *planq, *plans;
  double *lu;
  int *fail;
};
// one level per launch (wide levels: the multicolour ordering)
__global__ __launch_bounds__(NK_BLOCK) void k_ilu_factor_level(ilu_dev d, const int32_t *__restrict__ rows, int lo, int hi) {
  const int idx = lo + blockIdx.x * NK_BLOCK + threadIdx.x;
  if (idx < hi) ilu_factor_row(rows[idx], d.rp, d.ci, d.dg, d.lu, d.planptr, d.planq, d.plans, d.fail);
}
__global__ __launch_bounds__(NK_BLOCK) void k_ilu_lower_level(ilu_dev d, const int32_t *__restrict__ rows, int lo, int hi,
                                                              const double *__restrict__ b, double *y, const int *d_skip) {
  if (d_skip != nullptr && *d_skip != 0) return;
  const int idx = lo + blockIdx.x * NK_BLOCK + threadIdx.x;
  if (idx < hi) ilu_lower_row(rows[idx], d.rp, d.ci, d.dg, d.lu, d.perm, b, y);
}
__global__ __launch_bounds__(NK_BLOCK) void k_ilu_upper_level(ilu_dev d, const int32_t *__restrict__ rows, int lo, int hi,
                                                              const double *__restrict__ y, double *z, double *out,
                                                              const int *d_skip) {
  if (d_skip != nullptr && *d_skip != 0) return;
  const int idx = lo + blockIdx.x * NK_BLOCK + threadIdx.x;
  if (idx < hi) ilu_upper_row(rows[idx], d.rp, d.ci, d.dg, d.lu, d.perm, y, z, out);
}
// all levels inside ONE persistent workgroup (many narrow levels: the natural ordering of a stencil): a barrier per level
// instead of a launch per level. __syncthreads orders the workgroup's global stores before the next level's loads.
constexpr int ILU_CHAIN_THREADS = 1024;
__global__ __launch_bounds__(ILU_CHAIN_THREADS) void k_ilu_factor_chain(ilu_dev d, const int32_t *__restrict__ rows,
                                                                        const int32_t *__restrict__ ptr, int nlev) {
  for (int lev = 0; lev < nlev; ++lev) {
    for (int idx = ptr[lev] + threadIdx.x; idx < ptr[lev + 1]; idx += ILU_CHAIN_THREADS)
      ilu_factor_row(rows[idx], d.rp, d.ci, d.dg, d.lu, d.planptr, d.planq, d.plans, d.fail);
    __threadfence_block();
    __syncthreads();
  }
}
__global__ __launch_bounds__(ILU_CHAIN_THREADS) void k_ilu_lower_chain(ilu_dev d, const int32_t *__restrict__ rows,
                                                                       const int32_t *__restrict__ ptr, int nlev,
                                                                       const double *__restrict__ b, double *y,
                                                                       const int *d_skip) {
  if (d_skip != nullptr && *d_skip != 0) return;
  for (int lev = 0; lev < nlev; ++lev) {
    for (int idx = ptr[lev] + threadIdx.x; idx < ptr[lev + 1]; idx += ILU_CHAIN_THREADS)
      ilu_lower_row(rows[idx], d.rp, d.ci, d.dg, d.lu, d.perm, b, y);
    __threadfence_block();
    __syncthreads();
  }
}
__global__ __launch_bounds__(ILU_CHAIN_THREADS) void k_ilu_upper_chain(ilu_dev d, const int32_t *__restrict__ rows,
                                                                       const int32_t *__restrict__ ptr, int nlev,
                                                                       const double *__restrict__ y, double *z, double *out,
                                                                       const int *d_skip) {
  if (d_skip != nullptr && *d_skip != 0) return;
  for (int lev = 0; lev < nlev; ++lev) {
    for (int idx = ptr[lev] + threadIdx.x; idx < ptr[lev + 1]; idx += ILU_CHAIN_THREADS)
      ilu_upper_row(rows[idx], d.rp, d.ci, d.dg, d.lu, d.perm, y, z, out);
    __threadfence_block();
    __syncthreads();
  }
}
__global__ __launch_bounds__(NK_BLOCK) void k_ilu_gather_values(int64_t nnzp, const int32_t *__restrict__ src,
                                                                const double *__restrict__ aval, double *__restrict__ lu) {
  const int64_t stride = (int64_t)gridDim.x * NK_BLOCK;
  for (int64_t e = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x; e < nnzp; e += stride) lu[e] = aval[src[e]];
}

// ----------------------------------------------------------------------------- ILU(0): symbolic phase (host)
template <typename T>
static int upload(nk_ctx *ctx, T **dst, const std::vector<T> &v) {
  NK_TRY(nk_dev_alloc(dst, v.size() + 1));
  if (!v.empty()) NK_HIP(nk_memcpy(ctx, *dst, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return NK_OK;
}
// greedy distance-1 colouring of the symmetrised pattern in natural order (smallest free colour); rows are then ordered by
// (colour, original index) — restated in oracle/reference_restatement.py::multicolor_permutation
static void multicolor_perm(int64_t n, const std::vector<int32_t> &rp, const std::vector<int32_t> &ci,
                            std::vector<int32_t> &perm, int *ncolors) {
  std::vector<int32_t> trp(n + 1, 0), tci;   // transpose pattern of the local block (for unsymmetric patterns)
  for (int64_t i = 0; i < n; ++i)
    for (int32_t p = rp[i]; p < rp[i + 1]; ++p)
      if (ci[p] < n && ci[p] != i) trp[ci[p] + 1]++;
  for (int64_t i = 0; i < n; ++i) trp[i + 1] += trp[i];
  tci.resize(trp[n]);
  {
    std::vector<int32_t> fill(trp.begin(), trp.end() - 1);
    for (int64_t i = 0; i < n; ++i)
      for (int32_t p = rp[i]; p < rp[i + 1]; ++p)
        if (ci[p] < n && ci[p] != i) tci[fill[ci[p]]++] = (int32_t)i;
  }
  std::vector<int32_t> color(n, -1), mark;
  int nc = 0;
  for (int64_t i = 0; i < n; ++i) {
    mark.assign(nc + 1, 0);
    for (int32_t p = rp[i]; p < rp[i + 1]; ++p)
      if (ci[p] < n && ci[p] != i && color[ci[p]] >= 0) mark[color[ci[p]]] = 1;
    for (int32_t p = trp[i]; p < trp[i + 1]; ++p)
      if (color[tci[p]] >= 0) mark[color[tci[p]]] = 1;
    int c = 0;
    while (c < nc && mark[c]) ++c;
    color[i] = c;
    if (c == nc) ++nc;
  }
  perm.resize(n);
  std::iota(perm.begin(), perm.end(), 0);
  std::stable_sort(perm.begin(), perm.end(), [&](int32_t a, int32_t b) { return color[a] < color[b]; });
  *ncolors = nc;
}

static int ilu_symbolic(nk_precond *P) {
  nk_csr *A = P->A;
  const int64_t n = A->nrows;
  NK_REQUIRE((int64_t)A->h_rowptr.size() == n + 1, "ILU(0): the matrix keeps no host copy of its pattern");
  const std::vector<int32_t> &rp0 = A->h_rowptr, &ci0 = A->h_col;
  std::vector<int32_t> perm, iperm(n);
  if (P->ordering == NK_ILU_MULTICOLOR) {
    multicolor_perm(n, rp0, ci0, perm, &P->ncolors);
  } else {
    perm.resize(n);
    std::iota(perm.begin(), perm.end(), 0);
  }
  for (int64_t i = 0; i < n; ++i) iperm[perm[i]] = (int32_t)i;
  // permuted local block: row i = original row perm[i], columns renumbered and sorted; src = position in A's value array
  std::vector<int32_t> rp(n + 1, 0), ci, src, dg(n, -1);
  {
    std::vector<std::pair<int32_t, int32_t>> row;
    for (int64_t i = 0; i < n; ++i) {
      const int32_t o = perm[i];
      row.clear();
      for (int32_t p = rp0[o]; p < rp0[o + 1]; ++p)
        if (ci0[p] < n) row.emplace_back(iperm[ci0[p]], p);
      std::sort(row.begin(), row.end());
      for (auto &e : row) {
        if (e.first == i) dg[i] = (int32_t)ci.size();
        ci.push_back(e.first);
        src.push_back(e.second);
      }
      rp[i + 1] = (int32_t)ci.size();
      NK_REQUIRE(dg[i] >= 0, "ILU(0): row %lld has no stored diagonal entry", (long long)o);
    }
  }
  P->nnzp = (int64_t)ci.size();
  // update plan: for the lower entry p = (i, k): the entries q of row i behind it whose column row k holds above its diagonal
  std::vector<int32_t> planptr(P->nnzp + 1, 0), planq, plans;
  for (int64_t i = 0; i < n; ++i) {
    for (int32_t p = rp[i]; p < rp[i + 1]; ++p) {
      planptr[p] = (int32_t)planq.size();
      if (p >= dg[i]) continue;
      const int32_t k = ci[p];
      int32_t q = p + 1, s = dg[k] + 1;
      const int32_t qe = rp[i + 1], se = rp[k + 1];
      while (q < qe && s < se) {
        if (ci[q] == ci[s]) { planq.push_back(q); plans.push_back(s); ++q; ++s; }
        else if (ci[q] < ci[s]) ++q;
        else ++s;
      }
    }
  }
  planptr[P->nnzp] = (int32_t)planq.size();
  // level schedules: L (also the factorisation's) from the rows' lower entries, U from the upper ones
  auto levels = [&](bool lower, std::vector<int32_t> &rows, std::vector<int32_t> &ptr) {
    std::vector<int32_t> lev(n, 0);
    int32_t nlev = 0;
    if (lower) {
      for (int64_t i = 0; i < n; ++i) {
        int32_t l = 0;
        for (int32_t p = rp[i]; p < dg[i]; ++p) l = std::max(l, lev[ci[p]] + 1);
        lev[i] = l;
        nlev = std::max(nlev, l + 1);
      }
    } else {
      for (int64_t i = n - 1; i >= 0; --i) {
        int32_t l = 0;
        for (int32_t p = dg[i] + 1; p < rp[i + 1]; ++p) l = std::max(l, lev[ci[p]] + 1);
        lev[i] = l;
        nlev = std::max(nlev, l + 1);
      }
    }
    ptr.assign(nlev + 1, 0);
    for (int64_t i = 0; i < n; ++i) ptr[lev[i] + 1]++;
    for (int32_t l = 0; l < nlev; ++l) ptr[l + 1] += ptr[l];
    rows.resize(n);
    std::vector<int32_t> fill(ptr.begin(), ptr.end() - 1);
    for (int64_t i = 0; i < n; ++i) rows[fill[lev[i]]++] = (int32_t)i;   // ascending row index inside a level
  };
  std::vector<int32_t> rowsL, rowsU;
  levels(true, rowsL, P->h_ptrL);
  levels(false, rowsU, P->h_ptrU);
  // many narrow levels → the persistent single-workgroup walk (≈ 1 µs per level of ≤ 1024 rows) beats a launch per level (≈ 5 µs)
  auto chain_pays = [&](const std::vector<int32_t> &ptr) {
    const int nlev = (int)ptr.size() - 1;
    double chain_us = 0.0;
    for (int l = 0; l < nlev; ++l) chain_us += 1.0 * ((ptr[l + 1] - ptr[l] + ILU_CHAIN_THREADS - 1) / ILU_CHAIN_THREADS);
    return nlev > 16 && chain_us < 5.0 * nlev;
  };
  P->chainL = chain_pays(P->h_ptrL);
  P->chainU = chain_pays(P->h_ptrU);
  if (P->ordering != NK_ILU_MULTICOLOR) P->ncolors = 0;
  if (P->ordering == NK_ILU_MULTICOLOR) NK_TRY(upload(P->ctx, &P->d_perm, perm));
  NK_TRY(upload(P->ctx, &P->d_rp, rp));
  NK_TRY(upload(P->ctx, &P->d_ci, ci));
  NK_TRY(upload(P->ctx, &P->d_dg, dg));
  NK_TRY(upload(P->ctx, &P->d_src, src));
  NK_TRY(upload(P->ctx, &P->d_planptr, planptr));
  NK_TRY(upload(P->ctx, &P->d_planq, planq));
  NK_TRY(upload(P->ctx, &P->d_plans, plans));
  NK_TRY(upload(P->ctx, &P->d_rowsL, rowsL));
  NK_TRY(upload(P->ctx, &P->d_ptrL, P->h_ptrL));
  NK_TRY(upload(P->ctx, &P->d_rowsU, rowsU));
  NK_TRY(upload(P->ctx, &P->d_ptrU, P->h_ptrU));
  NK_TRY(nk_dev_alloc(&P->d_lu, (size_t)P->nnzp + 1));
  NK_TRY(nk_dev_alloc(&P->d_y, (size_t)n + 1));
  NK_TRY(nk_dev_alloc(&P->d_z, (size_t)n + 1));
  return NK_OK;
}

// ----------------------------------------------------------------------------- ILU(τ): Crout ILU with a drop tolerance
// The tutorial's OTHER precs, `incompletelu(W, p) = (ilu(W, τ = 50.0), I)` (docs/src/tutorials/large_systems.md:252-260;
// IncompleteLU.jl [EXT: not in the tree] implements Li, Saad, Chow, "Crout versions of ILU for general sparse matrices", SIAM J.
// Sci. Comput. 25 (2003)): A ≈ (I + L) U with fill, by the Crout order — step k forms row k of U and column k of L from the
// rows / columns finished before it —
//     z = A[k, k:] − Σ_{i<k, l_ki ≠ 0} l_ki · U[i, k:]          u_kk = z_k ;  u_kj = z_j kept if |z_j| ≥ τ  (j > k)
//     w = A[k+1:, k] − Σ_{i<k, u_ik ≠ 0} u_ik · L[k+1:, i]       l_ik = w_i / u_kk kept if |w_i| ≥ τ          (i > k)
// (the drop test compares the entry BEFORE the division by the pivot, absolutely — IncompleteLU.jl's rule as far as it can be told
// without its source; oracle/reference_restatement.py::ilut restates exactly this). τ = 0 is the complete LU without pivoting.
// The factorisation is sequential and its pattern depends on the numbers: it runs on the HOST for every new Jacobian — as the
// reference's does — on the rank's local block (halo columns dropped: block-Jacobi across ranks); the two triangular solves of
// every application run on the device, level-scheduled from the factors' actual pattern with the kernels of ILU(0).
static int ilut_update(nk_precond *P) {
  nk_csr *A = P->A;
  nk_ctx *ctx = P->ctx;
  const int64_t n = A->nrows;
  NK_REQUIRE((int64_t)A->h_rowptr.size() == n + 1, "ILU(τ): the matrix keeps no host copy of its pattern");
  const std::vector<int32_t> &rp0 = A->h_rowptr, &ci0 = A->h_col;
  std::vector<double> av((size_t)A->nnz);
  NK_HIP(hipMemcpyAsync(av.data(), A->d_val, (size_t)A->nnz * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  NK_HIP(hipStreamSynchronize(ctx->stream));
  const double tau = P->tau;
  // the local block by columns (for the L columns): CSC with ascending rows
  std::vector<int32_t> cp(n + 1, 0), cr;
  std::vector<double> cv;
  {
    for (int64_t i = 0; i < n; ++i)
      for (int32_t p = rp0[i]; p < rp0[i + 1]; ++p)
        if (ci0[p] < n) cp[ci0[p] + 1]++;
    for (int64_t j = 0; j < n; ++j) cp[j + 1] += cp[j];
    cr.resize(cp[n]);
    cv.resize(cp[n]);
    std::vector<int32_t> fill(cp.begin(), cp.end() - 1);
    for (int64_t i = 0; i < n; ++i)
      for (int32_t p = rp0[i]; p < rp0[i + 1]; ++p)
        if (ci0[p] < n) { const int32_t q = fill[ci0[p]]++; cr[q] = (int32_t)i; cv[q] = av[p]; }
  }
  // U by rows (diagonal first, then ascending columns), L by columns (ascending rows); and, for the Crout sums, for every row k the
  // finished L entries in it (column i, value) and for every column k the finished U entries in it (row i, value)
  struct ent { int32_t idx; double v; };
  std::vector<std::vector<ent>> Urow(n), Lcol(n), Lrow(n), Ucol(n);
  std::vector<double> diag(n, 0.0);
  std::vector<double> wz(n, 0.0);
  std::vector<char> mark(n, 0);
  std::vector<int32_t> idx;
  // where the entries ≥ k of U's row i / L's column i begin (k only grows: the cursors only advance)
  std::vector<int32_t> ufirst(n, 0), lfirst(n, 0);
  for (int64_t k = 0; k < n; ++k) {
    // ---- row k of U
    idx.clear();
    // (a row without a stored diagonal entry starts from 0 there: fill may still create the pivot, as in the reference's ilu —
    //  if it does not, the zero-pivot test below reports the row)
    mark[k] = 1; idx.push_back((int32_t)k); wz[k] = 0.0;
    for (int32_t p = rp0[k]; p < rp0[k + 1]; ++p) {
      const int32_t j = ci0[p];
      if (j < k || j >= n) continue;
      if (!mark[j]) { mark[j] = 1; idx.push_back(j); wz[j] = 0.0; }
      wz[j] += av[p];
    }
    for (const ent &l : Lrow[k]) {             // ascending i (the columns were finished in that order)
      const int32_t i = l.idx;
      const std::vector<ent> &ur = Urow[i];
      int32_t &f = ufirst[i];
      while (f < (int32_t)ur.size() && ur[f].idx < k) ++f;
      for (int32_t q = f; q < (int32_t)ur.size(); ++q) {
        const int32_t j = ur[q].idx;
        if (!mark[j]) { mark[j] = 1; idx.push_back(j); wz[j] = 0.0; }
        wz[j] -= l.v * ur[q].v;
      }
    }
    const double piv = wz[k];
    if (piv == 0.0 || !std::isfinite(piv)) {
      for (int32_t j : idx) mark[j] = 0;
      P->factored = false;
      NK_FAIL(NK_E_SINGULAR, "ILU(τ): zero or non-finite pivot in row %lld (no pivoting)", (long long)k);
    }
    diag[k] = piv;
    std::sort(idx.begin(), idx.end());
    for (int32_t j : idx) {
      mark[j] = 0;
      if (j > k && std::fabs(wz[j]) >= tau && wz[j] != 0.0) {
        Urow[k].push_back({j, wz[j]});
        Ucol[j].push_back({(int32_t)k, wz[j]});
      }
    }
    // ---- column k of L
    idx.clear();
    for (int32_t p = cp[k]; p < cp[k + 1]; ++p) {
      const int32_t r = cr[p];
      if (r <= k) continue;
      if (!mark[r]) { mark[r] = 1; idx.push_back(r); wz[r] = 0.0; }
      wz[r] += cv[p];
    }
    for (const ent &u : Ucol[k]) {             // ascending i
      const int32_t i = u.idx;
      const std::vector<ent> &lc = Lcol[i];
      int32_t &f = lfirst[i];
      while (f < (int32_t)lc.size() && lc[f].idx <= k) ++f;
      for (int32_t q = f; q < (int32_t)lc.size(); ++q) {
        const int32_t r = lc[q].idx;
        if (!mark[r]) { mark[r] = 1; idx.push_back(r); wz[r] = 0.0; }
        wz[r] -= u.v * lc[q].v;
      }
    }
    std::sort(idx.begin(), idx.end());
    for (int32_t r : idx) {
      mark[r] = 0;
      if (std::fabs(wz[r]) >= tau && wz[r] != 0.0) {
        const double l = wz[r] / piv;
        Lcol[k].push_back({r, l});
        Lrow[r].push_back({(int32_t)k, l});
      }
    }
  }
  // ---- the factors as ONE row-major matrix [L strictly lower | diagonal | U strictly upper], columns ascending: what the
  // level-scheduled kernels of ILU(0) walk
  std::vector<int32_t> rp(n + 1, 0), ci, dg(n, 0);
  std::vector<double> lu;
  for (int64_t i = 0; i < n; ++i) {
    for (const ent &l : Lrow[i]) { ci.push_back(l.idx); lu.push_back(l.v); }
    dg[i] = (int32_t)ci.size();
    ci.push_back((int32_t)i);
    lu.push_back(diag[i]);
    for (const ent &u : Urow[i]) { ci.push_back(u.idx); lu.push_back(u.v); }
    rp[i + 1] = (int32_t)ci.size();
  }
  P->nnzp = (int64_t)ci.size();
  auto levels = [&](bool lower, std::vector<int32_t> &rows, std::vector<int32_t> &ptr) {
    std::vector<int32_t> lev(n, 0);
    int32_t nlev = 0;
    if (lower) {
      for (int64_t i = 0; i < n; ++i) {
        int32_t l = 0;
        for (int32_t p = rp[i]; p < dg[i]; ++p) l = std::max(l, lev[ci[p]] + 1);
        lev[i] = l;
        nlev = std::max(nlev, l + 1);
      }
    } else {
      for (int64_t i = n - 1; i >= 0; --i) {
        int32_t l = 0;
        for (int32_t p = dg[i] + 1; p < rp[i + 1]; ++p) l = std::max(l, lev[ci[p]] + 1);
        lev[i] = l;
        nlev = std::max(nlev, l + 1);
      }
    }
    ptr.assign(nlev + 1, 0);
    for (int64_t i = 0; i < n; ++i) ptr[lev[i] + 1]++;
    for (int32_t l = 0; l < nlev; ++l) ptr[l + 1] += ptr[l];
    rows.resize(n);
    std::vector<int32_t> fill(ptr.begin(), ptr.end() - 1);
    for (int64_t i = 0; i < n; ++i) rows[fill[lev[i]]++] = (int32_t)i;
  };
  std::vector<int32_t> rowsL, rowsU;
  levels(true, rowsL, P->h_ptrL);
  levels(false, rowsU, P->h_ptrU);
  auto chain_pays = [&](const std::vector<int32_t> &ptr) {
    const int nlev = (int)ptr.size() - 1;
    double chain_us = 0.0;
    for (int l = 0; l < nlev; ++l) chain_us += 1.0 * ((ptr[l + 1] - ptr[l] + ILU_CHAIN_THREADS - 1) / ILU_CHAIN_THREADS);
    return nlev > 16 && chain_us < 5.0 * nlev;
  };
  P->chainL = chain_pays(P->h_ptrL);
  P->chainU = chain_pays(P->h_ptrU);
  // (the pattern changes with the values: the device copies are replaced)
  hipFree(P->d_rp); hipFree(P->d_ci); hipFree(P->d_dg); hipFree(P->d_lu);
  hipFree(P->d_rowsL); hipFree(P->d_ptrL); hipFree(P->d_rowsU); hipFree(P->d_ptrU);
  P->d_rp = P->d_ci = P->d_dg = P->d_rowsL = P->d_ptrL = P->d_rowsU = P->d_ptrU = nullptr;
  P->d_lu = nullptr;
  NK_TRY(upload(P->ctx, &P->d_rp, rp));
  NK_TRY(upload(P->ctx, &P->d_ci, ci));
  NK_TRY(upload(P->ctx, &P->d_dg, dg));
  NK_TRY(upload(P->ctx, &P->d_rowsL, rowsL));
  NK_TRY(upload(P->ctx, &P->d_ptrL, P->h_ptrL));
  NK_TRY(upload(P->ctx, &P->d_rowsU, rowsU));
  NK_TRY(upload(P->ctx, &P->d_ptrU, P->h_ptrU));
  NK_TRY(upload(P->ctx, &P->d_lu, lu));
  if (!P->d_y) NK_TRY(nk_dev_alloc(&P->d_y, (size_t)n + 1));
  if (!P->d_z) NK_TRY(nk_dev_alloc(&P->d_z, (size_t)n + 1));
  P->factored = true;
  return NK_OK;
}

static ilu_dev ilu_view(const nk_precond *P) {
  ilu_dev d;
  d.rp = P->d_rp; d.ci = P->d_ci; d.dg = P->d_dg; d.perm = P->d_perm;
  d.planptr = P->d_planptr; d.planq = P->d_planq; d.plans = P->d_plans;
  d.lu = P->d_lu; d.fail = P->d_fail;
  return d;
}

// ----------------------------------------------------------------------------- public entry points
static int precond_new(nk_csr *A, int kind, nk_precond **out, nk_precond **Pp) {
  NK_REQUIRE(A && out, "NULL argument");
  NK_HIP(hipSetDevice(A->ctx->device));
  nk_precond *P = new nk_precond();
  P->ctx = A->ctx;
  P->kind = kind;
  P->A = A;
  P->n = A->nrows;
  if (nk_dev_alloc(&P->d_fail, (size_t)2) != NK_OK) { delete P; NK_FAIL(NK_E_NOMEM, "out of device memory"); }
  nk_memset(P->ctx, P->d_fail, 0, 2 * sizeof(int));
  *Pp = P;
  return NK_OK;
}
extern "C" int nk_precond_create_jacobi(nk_csr *A, nk_precond **out) {
  nk_precond *P = nullptr;
  NK_TRY(precond_new(A, NK_PRECOND_JACOBI, out, &P));
  auto guard = nk_make_guard(P, [](nk_precond *p) { nk_precond_destroy(p); });
  NK_TRY(nk_dev_alloc(&P->d_dinv, (size_t)P->n + 1));
  NK_TRY(nk_precond_update(P));
  *out = guard.release();
  return NK_OK;
}
extern "C" int nk_precond_create_ilu0(nk_csr *A, int ordering, nk_precond **out) {
  NK_REQUIRE(ordering == NK_ILU_NATURAL || ordering == NK_ILU_MULTICOLOR, "bad ILU(0) ordering %d", ordering);
  nk_precond *P = nullptr;
  NK_TRY(precond_new(A, NK_PRECOND_ILU0, out, &P));
  auto guard = nk_make_guard(P, [](nk_precond *p) { nk_precond_destroy(p); });
  P->ordering = ordering;
  NK_TRY(ilu_symbolic(P));
  NK_TRY(nk_precond_update(P));
  *out = guard.release();
  return NK_OK;
}
extern "C" int nk_precond_create_ilut(nk_csr *A, double tau, nk_precond **out) {
  NK_REQUIRE(tau >= 0.0 && std::isfinite(tau), "ILU(τ): the drop tolerance must be finite and ≥ 0");
  nk_precond *P = nullptr;
  NK_TRY(precond_new(A, NK_PRECOND_ILUT, out, &P));
  auto guard = nk_make_guard(P, [](nk_precond *p) { nk_precond_destroy(p); });
  P->tau = tau;
  NK_TRY(nk_precond_update(P));
  *out = guard.release();
  return NK_OK;
}
extern "C" int nk_precond_create_amg(nk_csr *A, const nk_amg_params *params, nk_precond **out) {
  nk_precond *P = nullptr;
  NK_TRY(precond_new(A, NK_PRECOND_AMG, out, &P));
  auto guard = nk_make_guard(P, [](nk_precond *p) { nk_precond_destroy(p); });
  NK_TRY(nk_amg_create(A, params, &P->amg));   // (creates the hierarchy and its numbers for the current values)
  P->factored = true;
  *out = guard.release();
  return NK_OK;
}
extern "C" int nk_precond_amg_info(nk_precond *P, int *levels, int cap, int64_t *sizes, int64_t *nnzs, double *lmax) {
  NK_REQUIRE(P && P->kind == NK_PRECOND_AMG && P->amg, "not an AMG preconditioner");
  const int nl = nk_amg_levels(P->amg);
  if (levels) *levels = nl;
  for (int l = 0; l < nl && l < cap; ++l)
    NK_TRY(nk_amg_level_info(P->amg, l, sizes ? sizes + l : nullptr, nnzs ? nnzs + l : nullptr, lmax ? lmax + l : nullptr));
  return NK_OK;
}
extern "C" int nk_precond_amg_matching(nk_precond *P, int *matching) {
  NK_REQUIRE(P && P->kind == NK_PRECOND_AMG && P->amg && matching, "not an AMG preconditioner");
  *matching = nk_amg_matching(P->amg);
  return NK_OK;
}
extern "C" int nk_precond_amg_aggregates(nk_precond *P, int level, int32_t *agg, int64_t count) {
  NK_REQUIRE(P && P->kind == NK_PRECOND_AMG && P->amg && agg, "not an AMG preconditioner");
  const int32_t *h = nk_amg_aggregates(P->amg, level);
  int64_t n = 0;
  NK_REQUIRE(h && nk_amg_level_info(P->amg, level, &n, nullptr, nullptr) == NK_OK && n == count,
             "AMG: level %d has no aggregates or not %lld rows", level, (long long)count);
  memcpy(agg, h, (size_t)n * sizeof(int32_t));
  return NK_OK;
}
extern "C" int nk_precond_destroy(nk_precond *P) {
  if (!P) return NK_OK;
  nk_amg_destroy(P->amg);
  hipFree(P->d_dinv); hipFree(P->d_perm); hipFree(P->d_rp); hipFree(P->d_ci); hipFree(P->d_dg); hipFree(P->d_src);
  hipFree(P->d_lu); hipFree(P->d_planptr); hipFree(P->d_planq); hipFree(P->d_plans);
  hipFree(P->d_rowsL); hipFree(P->d_ptrL); hipFree(P->d_rowsU); hipFree(P->d_ptrU);
  hipFree(P->d_y); hipFree(P->d_z); hipFree(P->d_xin); hipFree(P->d_xout); hipFree(P->d_fail);
  delete P;
  return NK_OK;
}

// refactorise for the matrix's current values (call whenever they change — `precs(A, p)` is re-evaluated for every new A)
extern "C" int nk_precond_update(nk_precond *P) {
  NK_REQUIRE(P, "NULL argument");
  nk_ctx *ctx = P->ctx;
  NK_HIP(hipSetDevice(ctx->device));
  nk_csr *A = P->A;
  const int64_t n = P->n;
  if (P->kind == NK_PRECOND_AMG) {
    P->factored = false;
    NK_TRY(nk_amg_update(P->amg));
    P->factored = true;
    return NK_OK;
  }
  if (P->kind == NK_PRECOND_ILUT) {
    P->factored = false;
    return ilut_update(P);
  }
  NK_HIP(hipMemsetAsync(P->d_fail, 0, sizeof(int), ctx->stream));
  if (n > 0) {
    if (P->kind == NK_PRECOND_JACOBI) {
      nk_prof_scope prof_(ctx, NK_K_OTHER, 12.0 * (double)A->nnz + 12.0 * (double)n);
      NK_LAUNCH(ctx, k_jacobi_setup, dim3(nk_grid_for(n, NK_BLOCK, 4096)), dim3(NK_BLOCK), n, A->d_rowptr, A->d_col, A->d_val,
                P->d_dinv, P->d_fail);
    } else {
      nk_prof_scope prof_(ctx, NK_K_OTHER, 28.0 * (double)P->nnzp);
      NK_LAUNCH(ctx, k_ilu_gather_values, dim3(nk_grid_for(P->nnzp, NK_BLOCK * 4, 4096)), dim3(NK_BLOCK), P->nnzp,
                (const int32_t *)P->d_src, (const double *)A->d_val, P->d_lu);
      const ilu_dev d = ilu_view(P);
      const int nlev = (int)P->h_ptrL.size() - 1;
      if (P->chainL) {
        NK_LAUNCH(ctx, k_ilu_factor_chain, dim3(1), dim3(ILU_CHAIN_THREADS), d, (const int32_t *)P->d_rowsL,
                  (const int32_t *)P->d_ptrL, nlev);
      } else {
        for (int l = 0; l < nlev; ++l) {
          const int lo = P->h_ptrL[l], hi = P->h_ptrL[l + 1];
          if (hi > lo)
            NK_LAUNCH(ctx, k_ilu_factor_level, dim3((hi - lo + NK_BLOCK - 1) / NK_BLOCK), dim3(NK_BLOCK), d,
                      (const int32_t *)P->d_rowsL, lo, hi);
        }
      }
    }
    NK_HIP(hipGetLastError());
  }
  int fail = 0;
  NK_HIP(hipMemcpyAsync(&fail, P->d_fail, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  NK_HIP(hipStreamSynchronize(ctx->stream));
  P->factored = (fail == 0);   // factors with a zero / non-finite pivot are not usable: apply refuses until a good update
  if (fail) NK_FAIL(NK_E_SINGULAR, P->kind == NK_PRECOND_JACOBI ? "Jacobi preconditioner: zero or non-finite diagonal entry"
                                                                 : "ILU(0): zero or non-finite pivot (no pivoting)");
  return NK_OK;
}

// y = M⁻¹ x on device vectors of local length n (x and y may alias)
int nk_precond_apply_dev(nk_precond *P, const double *d_x, double *d_y, const int *d_skip) {
  nk_ctx *ctx = P->ctx;
  const int64_t n = P->n;
  if (n == 0) return NK_OK;
  if (!P->factored)
    NK_FAIL(NK_E_SINGULAR, "preconditioner object has no valid factors (its last update met a zero or non-finite pivot)");
  if (P->kind == NK_PRECOND_AMG) {
    if (d_x != d_y) return nk_amg_apply_dev(P->amg, d_x, d_y, d_skip);
    // in place: the V-cycle reads b while it builds x
    if (!P->d_xin) NK_TRY(nk_dev_alloc(&P->d_xin, (size_t)n + 1));
    NK_HIP(hipMemcpyAsync(P->d_xin, d_x, n * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    return nk_amg_apply_dev(P->amg, P->d_xin, d_y, d_skip);
  }
  if (P->kind == NK_PRECOND_JACOBI) {
    nk_prof_scope prof_(ctx, NK_K_OTHER, 24.0 * (double)n);
    NK_LAUNCH(ctx, k_jacobi_apply, dim3(nk_grid_for(n, NK_BLOCK * 4, 4096)), dim3(NK_BLOCK), n, (const double *)P->d_dinv, d_x,
              d_y, d_skip);
    NK_HIP(hipGetLastError());
    return NK_OK;
  }
  const ilu_dev d = ilu_view(P);
  nk_prof_scope prof_(ctx, NK_K_OTHER, 12.0 * (double)P->nnzp + 40.0 * (double)n);
  {
    const int nlev = (int)P->h_ptrL.size() - 1;
    if (P->chainL) {
      NK_LAUNCH(ctx, k_ilu_lower_chain, dim3(1), dim3(ILU_CHAIN_THREADS), d, (const int32_t *)P->d_rowsL,
                (const int32_t *)P->d_ptrL, nlev, d_x, P->d_y, d_skip);
    } else {
      for (int l = 0; l < nlev; ++l) {
        const int lo = P->h_ptrL[l], hi = P->h_ptrL[l + 1];
        if (hi > lo)
          NK_LAUNCH(ctx, k_ilu_lower_level, dim3((hi - lo + NK_BLOCK - 1) / NK_BLOCK), dim3(NK_BLOCK), d,
                    (const int32_t *)P->d_rowsL, lo, hi, d_x, P->d_y, d_skip);
      }
    }
  }
  {
    const int nlev = (int)P->h_ptrU.size() - 1;
    if (P->chainU) {
      NK_LAUNCH(ctx, k_ilu_upper_chain, dim3(1), dim3(ILU_CHAIN_THREADS), d, (const int32_t *)P->d_rowsU,
                (const int32_t *)P->d_ptrU, nlev, (const double *)P->d_y, P->d_z, d_y, d_skip);
    } else {
      for (int l = 0; l < nlev; ++l) {
        const int lo = P->h_ptrU[l], hi = P->h_ptrU[l + 1];
        if (hi > lo)
          NK_LAUNCH(ctx, k_ilu_upper_level, dim3((hi - lo + NK_BLOCK - 1) / NK_BLOCK), dim3(NK_BLOCK), d,
                    (const int32_t *)P->d_rowsU, lo, hi, (const double *)P->d_y, P->d_z, d_y, d_skip);
      }
    }
  }
  NK_HIP(hipGetLastError());
  return NK_OK;
}
extern "C" int nk_precond_apply(nk_precond *P, const double *x, double *y, int memspace) {
  NK_REQUIRE(P && x && y, "NULL argument");
  nk_ctx *ctx = P->ctx;
  NK_HIP(hipSetDevice(ctx->device));
  if (memspace == NK_DEVICE) return nk_precond_apply_dev(P, x, y, nullptr);
  if (!P->d_xin) NK_TRY(nk_dev_alloc(&P->d_xin, (size_t)P->n + 1));
  if (!P->d_xout) NK_TRY(nk_dev_alloc(&P->d_xout, (size_t)P->n + 1));
  NK_HIP(hipMemcpyAsync(P->d_xin, x, P->n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  NK_TRY(nk_precond_apply_dev(P, P->d_xin, P->d_xout, nullptr));
  NK_HIP(hipMemcpyAsync(y, P->d_xout, P->n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  NK_HIP(hipStreamSynchronize(ctx->stream));
  return NK_OK;
}
extern "C" int nk_precond_info(nk_precond *P, int *kind, int *levels_lower, int *levels_upper, int *ncolors) {
  NK_REQUIRE(P, "NULL argument");
  if (kind) *kind = P->kind;
  const bool ilu = P->kind == NK_PRECOND_ILU0 || P->kind == NK_PRECOND_ILUT;
  if (levels_lower) *levels_lower = ilu ? (int)P->h_ptrL.size() - 1 : (P->kind == NK_PRECOND_AMG ? nk_amg_levels(P->amg) : 0);
  if (levels_upper) *levels_upper = ilu ? (int)P->h_ptrU.size() - 1 : 0;
  if (ncolors) *ncolors = P->ncolors;
  return NK_OK;
}
// the factors in the permuted ordering, for parity tests: CSR of the local block (rowptr n+1, cols and values nnz — L strictly
// below the diagonal with its unit diagonal implied, U on and above) and the permutation (permuted row → original row)
extern "C" int nk_precond_ilu0_factors(nk_precond *P, int64_t *nnz, int32_t *rowptr, int32_t *col, double *val, int32_t *perm) {
  NK_REQUIRE(P && (P->kind == NK_PRECOND_ILU0 || P->kind == NK_PRECOND_ILUT), "not an incomplete-LU preconditioner");
  NK_HIP(hipSetDevice(P->ctx->device));
  NK_HIP(hipStreamSynchronize(P->ctx->stream));
  if (nnz) *nnz = P->nnzp;
  if (rowptr) NK_HIP(nk_memcpy(P->ctx, rowptr, P->d_rp, (P->n + 1) * sizeof(int32_t), hipMemcpyDeviceToHost));
  if (col) NK_HIP(nk_memcpy(P->ctx, col, P->d_ci, P->nnzp * sizeof(int32_t), hipMemcpyDeviceToHost));
  if (val) NK_HIP(nk_memcpy(P->ctx, val, P->d_lu, P->nnzp * sizeof(double), hipMemcpyDeviceToHost));
  if (perm) {
    if (P->d_perm) NK_HIP(nk_memcpy(P->ctx, perm, P->d_perm, P->n * sizeof(int32_t), hipMemcpyDeviceToHost));
    else for (int64_t i = 0; i < P->n; ++i) perm[i] = (int32_t)i;
  }
  return NK_OK;
}
nk_csr *nk_precond_matrix(nk_precond *P) { return P->A; }
int64_t nk_precond_size(nk_precond *P) { return P->n; }
