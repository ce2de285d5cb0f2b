// Aggregation algebraic multigrid built from a CSR matrix ALONE — the `precs(A, p)` slot the reference's tutorial fills with
// AlgebraicMultigrid.jl's ruge_stuben / smoothed_aggregation as Pl (docs/src/tutorials/large_systems.md:276-316): the
// preconditioner that makes a GENERAL sparse Jacobian at n = 1e6 converge in a handful of Krylov iterations (ILU(0) and
// Jacobi do not: DESIGN.md §5d; the geometric V-cycle of nk_mg.hip knows the two built-in grids only).
//
//   coarsening   pairwise aggregation, `passes` times per level (aggregates of ≤ 2^passes rows): rows in natural order, an
//                unmatched row takes its unmatched neighbour of largest strength s_ij = −a_ij·sign(a_ii) if s_ij ≥ θ·max_k s_ik.
//                HOST, once per pattern (from the values at creation); the aggregates are kept for the object's life.
//   transfers    piecewise constant: restriction = sum over an aggregate's rows (ascending), prolongation = injection.
//   coarse A     Galerkin TᵀA T: every coarse entry = the sum of the fine entries it covers, in ascending position — a gather
//                plan per level, one thread per coarse entry: bitwise reproducible, refreshed on the DEVICE for every new A.
//   smoother     ν Chebyshev steps on D⁻¹A over [λmax/ratio, λmax], λmax = max_i Σ_j |a_ij| / |a_ii| (device reduction);
//                the steps ride in the CSR SpMV's row epilogue (modes 1 / 2 / 4 of nk_spmv_epi: no separate vector pass).
//   coarsest     ≤ 128 rows: dense inverse by the in-register Gauss–Jordan of nk_bcr.hip (row pivoting), applied as one
//                GEMV launch; if coarsening stalls above that, 4ν smoothing steps.
//   cycle        V, with the coarse correction over-corrected by ω (1.8): plain aggregation under-estimates smooth error by
//                about a factor of two. A fixed linear operator — plain GMRES may use it on either side.
// Several ranks: the rank's LOCAL square block (halo columns dropped) — block-Jacobi AMG, no communication in the apply.
// CPU restatement: oracle/reference_restatement.py::AggregationAMG (same aggregates, V-cycle equal to 1e-12 relative).
#include "nk_internal.h"


#include <algorithm>
#include <cmath>
#include <cstring>
#include <functional>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <thread>
#include <vector>

struct amg_level {
  int64_t n = 0, nnz = 0, nc = 0;
  nk_csr *A = nullptr;       // level matrix (level 0: the caller's matrix, or a private copy of its local block)
  bool own_A = false;
  int32_t *d_agg = nullptr;                      // row → coarse row
  int32_t *d_aggptr = nullptr, *d_aggmem = nullptr;   // coarse row → its rows, ascending
  int32_t *d_gptr = nullptr, *d_gidx = nullptr;       // coarse entry → the fine entries it sums, ascending
  double *d_dinv = nullptr;
  double *b = nullptr, *x = nullptr, *r = nullptr, *d0 = nullptr, *d1 = nullptr;   // work vectors (level 0: b, x are the caller's)
  double lmax = 1.0;
  std::vector<int32_t> h_agg;                    // host copy of d_agg (device set-up: fetched on first request)
};
struct nk_amg {
  nk_ctx *ctx = nullptr;
  nk_csr *A = nullptr;
  nk_amg_params prm{};
  std::vector<amg_level> lv;   // lv.back() is the coarsest level (no aggregates)
  int32_t *d_src0 = nullptr;   // level 0 is a private local block: position of its entries in A->d_val
  bool dense = false;
  double *d_inv = nullptr;     // coarsest level: dense inverse, column-major, leading dimension ldinv (n rounded up to 8)
  int ldinv = 0;
  double *d_lmax = nullptr, *d_lpart = nullptr;
  int *d_fail = nullptr, *d_nparts = nullptr;
  std::vector<int> nparts;     // workgroups of k_amg_diag per level (= partial maxima to reduce)
  int lpart_stride = 0;
  bool ready = false;
  int matching = 1;            // how the aggregates were formed: 1 = sequential pairwise pass (host), 2 = handshaking (device)
};

extern "C" int nk_amg_params_default(nk_amg_params *p) {
  NK_REQUIRE(p, "NULL argument");
  p->nu = 2; p->passes = 2; p->coarse_max = 128; p->matching = 0;
  p->theta = 0.25; p->overcorrection = 1.8; p->cheb_ratio = 4.0;
  return NK_OK;
}

// ----------------------------------------------------------------------------- host: aggregates and Galerkin patterns
struct host_csr {
  int64_t n = 0;
  std::vector<int32_t> rp, ci;
  std::vector<double> v;
};
static void pairwise_pass(const host_csr &A, double theta, std::vector<int32_t> &cid, int32_t &nc) {
  const int64_t n = A.n;
  cid.assign(n, -1);
  nc = 0;
  for (int64_t i = 0; i < n; ++i) {
    if (cid[i] >= 0) continue;
    double dg = 0.0;
    for (int32_t k = A.rp[i]; k < A.rp[i + 1]; ++k) if (A.ci[k] == i) dg += A.v[k];
    const double sg = dg < 0.0 ? -1.0 : 1.0;
    double smax = 0.0, bv = 0.0;
    int32_t best = -1;
    for (int32_t k = A.rp[i]; k < A.rp[i + 1]; ++k) {
      const int32_t j = A.ci[k];
      if (j == i) continue;
      const double sv = -A.v[k] * sg;
      if (sv > smax) smax = sv;
      if (cid[j] < 0 && sv > bv) { bv = sv; best = j; }
    }
    cid[i] = nc;
    if (best >= 0 && bv >= theta * smax) cid[best] = nc;
    ++nc;
  }
}
// C = TᵀA T for the aggregate map cid (columns sorted); emap[k] = the entry of C fine entry k is summed into (values: ascending k)
// Threaded over coarse rows, two passes (count, fill): a coarse row's fine entries are gathered as (coarse column, fine position)
// keys and sorted in a small local buffer — the summation order of every coarse entry is still "ascending fine position", whatever
// the number of threads (round 4: one thread, a heap vector and std::sort per row: 34 + 11 + 24 ms for the first level of 1024²).
static int amg_host_threads() {
  static const int t = [] {
    const char *e = getenv("NK_AMG_THREADS");
    int v = e ? atoi(e) : (int)std::thread::hardware_concurrency();
    return v < 1 ? 1 : (v > 16 ? 16 : v);
  }();
  return t;
}
template <class F>
static void amg_parallel_for(int64_t n, F body /* (int64_t lo, int64_t hi) */) {
  const int T = (n < 20000) ? 1 : amg_host_threads();
  if (T == 1) { body(0, n); return; }
  std::vector<std::thread> th;
  for (int t = 0; t < T; ++t) th.emplace_back([=] { body(n * t / T, n * (t + 1) / T); });
  for (auto &x : th) x.join();
}
static void galerkin_host(const host_csr &A, const std::vector<int32_t> &cid, int32_t nc, host_csr &C, std::vector<int32_t> &emap) {
  const int64_t n = A.n, nnz = (int64_t)A.ci.size();
  std::vector<int32_t> mptr(nc + 1, 0), mem(n);
  for (int64_t i = 0; i < n; ++i) mptr[cid[i] + 1]++;
  for (int32_t I = 0; I < nc; ++I) mptr[I + 1] += mptr[I];
  {
    std::vector<int32_t> fill(mptr.begin(), mptr.end() - 1);
    for (int64_t i = 0; i < n; ++i) mem[fill[cid[i]]++] = (int32_t)i;
  }
  C.n = nc;
  C.rp.assign(nc + 1, 0);
  emap.assign(nnz, -1);
  // the keys of coarse row I, sorted by (coarse column, fine position)
  auto gather = [&](int32_t I, std::vector<uint64_t> &keys) {
    keys.clear();
    for (int32_t t = mptr[I]; t < mptr[I + 1]; ++t) {
      const int32_t i = mem[t];
      for (int32_t k = A.rp[i]; k < A.rp[i + 1]; ++k) keys.push_back(((uint64_t)(uint32_t)cid[A.ci[k]] << 32) | (uint32_t)k);
    }
    if (keys.size() <= 24) {   // (the usual case: ≤ 4 rows of a stencil) insertion sort, no call
      for (size_t a = 1; a < keys.size(); ++a) {
        const uint64_t v = keys[a];
        size_t b = a;
        for (; b > 0 && keys[b - 1] > v; --b) keys[b] = keys[b - 1];
        keys[b] = v;
      }
    } else {
      std::sort(keys.begin(), keys.end());
    }
  };
  amg_parallel_for(nc, [&](int64_t lo, int64_t hi) {
    std::vector<uint64_t> keys;
    keys.reserve(64);
    for (int64_t I = lo; I < hi; ++I) {
      gather((int32_t)I, keys);
      int32_t cnt = 0;
      for (size_t t = 0; t < keys.size(); ++t) cnt += (t == 0 || (keys[t] >> 32) != (keys[t - 1] >> 32)) ? 1 : 0;
      C.rp[I + 1] = cnt;
    }
  });
  for (int32_t I = 0; I < nc; ++I) C.rp[I + 1] += C.rp[I];
  C.ci.assign((size_t)C.rp[nc], 0);
  C.v.assign((size_t)C.rp[nc], 0.0);
  amg_parallel_for(nc, [&](int64_t lo, int64_t hi) {
    std::vector<uint64_t> keys;
    keys.reserve(64);
    for (int64_t I = lo; I < hi; ++I) {
      gather((int32_t)I, keys);
      int32_t e = C.rp[I] - 1;
      for (size_t t = 0; t < keys.size(); ++t) {
        if (t == 0 || (keys[t] >> 32) != (keys[t - 1] >> 32)) C.ci[++e] = (int32_t)(keys[t] >> 32);
        const int32_t k = (int32_t)(uint32_t)keys[t];
        emap[k] = e;
        C.v[e] += A.v[k];   // ascending fine position within the entry
      }
    }
  });
}

// ----------------------------------------------------------------------------- device kernels
__global__ __launch_bounds__(NK_BLOCK) void k_amg_gather(int64_t nnz, const int32_t *__restrict__ src, const double *__restrict__ in,
                                                         double *__restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * NK_BLOCK;
  for (int64_t e = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x; e < nnz; e += stride) out[e] = in[src[e]];
}
__global__ __launch_bounds__(NK_BLOCK) void k_amg_galerkin(int64_t nnzc, const int32_t *__restrict__ gptr, const int32_t *__restrict__ gidx,
                                                           const double *__restrict__ fine, double *__restrict__ coarse) {
  const int64_t stride = (int64_t)gridDim.x * NK_BLOCK;
  for (int64_t e = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x; e < nnzc; e += stride) {
    double s = 0.0;
    for (int32_t t = gptr[e]; t < gptr[e + 1]; ++t) s += fine[gidx[t]];
    coarse[e] = s;
  }
}
// dinv_i = 1 / a_ii; per-block maximum of Σ_j |a_ij| / |a_ii| (the Gershgorin bound of D⁻¹A)
__global__ __launch_bounds__(NK_BLOCK) void k_amg_diag(int64_t n, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                       const double *__restrict__ val, double *__restrict__ dinv,
                                                       double *__restrict__ part, int *fail) {
  __shared__ double red[4];
  const int64_t stride = (int64_t)gridDim.x * NK_BLOCK;
  double m = 0.0;
  for (int64_t r = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x; r < n; r += stride) {
    double d = 0.0, s = 0.0;
    for (int32_t p = rowptr[r]; p < rowptr[r + 1]; ++p) {
      const double v = val[p];
      if (col[p] == r) d += v;
      s += fabs(v);
    }
    if (d == 0.0 || d != d || isinf(d)) *fail = 1;
    dinv[r] = 1.0 / d;
    m = fmax(m, s / fabs(d));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
}
__global__ __launch_bounds__(NK_BLOCK) void k_amg_lmax_final(int nlev, int stride, const int *__restrict__ nparts,
                                                             const double *__restrict__ part, double *__restrict__ lmax) {
  __shared__ double red[4];
  for (int l = 0; l < nlev; ++l) {
    double m = 0.0;
    for (int i = threadIdx.x; i < nparts[l]; i += NK_BLOCK) m = fmax(m, part[(size_t)l * stride + i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) lmax[l] = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
  }
}
// the coarsest matrix as a dense column-major ld × ld array, ld = n rounded up to a multiple of 8 with an identity border (the
// pivoting Gauss–Jordan of nk_bcr.hip walks its pivots in groups of eight)
__global__ __launch_bounds__(NK_BLOCK) void k_amg_dense_fill(int n, int ld, const int32_t *__restrict__ rowptr,
                                                             const int32_t *__restrict__ col, const double *__restrict__ val,
                                                             double *__restrict__ M) {
  for (int64_t e = threadIdx.x; e < (int64_t)ld * ld; e += NK_BLOCK) M[e] = 0.0;
  __syncthreads();
  for (int r = threadIdx.x; r < ld; r += NK_BLOCK) {
    if (r >= n) { M[(int64_t)r + (int64_t)r * ld] = 1.0; continue; }
    for (int32_t p = rowptr[r]; p < rowptr[r + 1]; ++p) M[(int64_t)r + (int64_t)col[p] * ld] += val[p];
  }
}
// x = M b, M dense column-major (leading dimension ld), n ≤ 128: lanes run along the rows, coalesced
__global__ __launch_bounds__(128) void k_amg_dense_apply(int n, int ld, const double *__restrict__ M, const double *__restrict__ b,
                                                         double *__restrict__ x, const int *d_skip) {
  if (d_skip != nullptr && *d_skip != 0) return;
  __shared__ double sb[128];
  const int t = threadIdx.x;
  if (t < n) sb[t] = b[t];
  __syncthreads();
  if (t >= n) return;
  double s = 0.0;
  for (int j = 0; j < n; ++j) s += M[(int64_t)t + (int64_t)j * ld] * sb[j];
  x[t] = s;
}
// first Chebyshev step. zero = 1: from x = 0 (r ← b, d = D⁻¹b/θ, x = d); zero = 0: r holds b − A x (d = D⁻¹r/θ, x += d)
__global__ __launch_bounds__(NK_BLOCK) void k_amg_cheb_first(int64_t n, int zero, const double *__restrict__ b,
                                                             const double *__restrict__ dinv, double inv_theta,
                                                             double *__restrict__ r, double *__restrict__ d, double *__restrict__ x,
                                                             const int *d_skip) {
  if (d_skip != nullptr && *d_skip != 0) return;
  const int64_t stride = (int64_t)gridDim.x * NK_BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x; i < n; i += stride) {
    if (zero) {
      const double bb = b[i], dd = dinv[i] * bb * inv_theta;
      r[i] = bb;
      d[i] = dd;
      x[i] = dd;
    } else {
      const double dd = dinv[i] * r[i] * inv_theta;
      d[i] = dd;
      x[i] += dd;
    }
  }
}
__global__ __launch_bounds__(NK_BLOCK) void k_amg_restrict(int64_t nc, const int32_t *__restrict__ aggptr, const int32_t *__restrict__ aggmem,
                                                           const double *__restrict__ r, double *__restrict__ bc, const int *d_skip) {
  if (d_skip != nullptr && *d_skip != 0) return;
  const int64_t stride = (int64_t)gridDim.x * NK_BLOCK;
  for (int64_t I = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x; I < nc; I += stride) {
    double s = 0.0;
    for (int32_t t = aggptr[I]; t < aggptr[I + 1]; ++t) s += r[aggmem[t]];
    bc[I] = s;
  }
}
__global__ __launch_bounds__(NK_BLOCK) void k_amg_prolong(int64_t n, const int32_t *__restrict__ agg, const double *__restrict__ xc,
                                                          double omega, double *__restrict__ x, const int *d_skip) {
  if (d_skip != nullptr && *d_skip != 0) return;
  const int64_t stride = (int64_t)gridDim.x * NK_BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x; i < n; i += stride) x[i] += omega * xc[agg[i]];
}

// ----------------------------------------------------------------------------- device set-up (round 5): handshake matching + Galerkin plans
// Everything the host set-up above does, on the device, for a matrix without halo columns. A parallel pairwise pass cannot
// reproduce the sequential "rows in natural order" rule, so the matching is its own (deterministic, order-free) rule —
// HANDSHAKING: every unmatched row names its best unmatched neighbour, two rows that name each other are paired, AMG_HS_ROUNDS
// rounds. "Best": largest strength s_ij = −a_ij·sign(a_ii) among s_ij ≥ θ·max_k s_ik, ties by a key both ends of an edge compute
// alike — for the edge a < b, d = b − a: [d, far first (variant 1), or 2³¹ − 1 − d, near first (variant 0)] ≫ [⌊a/d⌋ even first] ≫
// [a 32-bit hash of (a, b)] ≫ [the larger column]. On a lexicographically numbered grid with equal couplings ⌊a/d⌋ is the
// position along the grid line: whole lines pair up in ONE round and the aggregates are as regular as the sequential rule's
// (variant 0 reproduces its aggregates on even-sized and periodic grids; variant 1 keeps the pairs aligned on odd-sized ones:
// a level that variant 0 does not coarsen by 2^passes within 2 % is coarsened with both and keeps the one with fewer aggregates).
// The Galerkin plans: one thread per coarse row writes the keys (coarse column ≪ 32 | fine position) of its members' entries
// into its segment of a key buffer, sorts the segment (Shell sort: ≤ 2^passes stencil rows), counts and then emits the distinct
// coarse columns; the sorted low words ARE the gather list ("ascending fine position within a coarse entry", as on the host).
// CPU restatement: oracle/reference_restatement.py::amg_handshake_pass / AggregationAMG(matching="handshake").
constexpr int AMG_HS_ROUNDS = 8;
__host__ __device__ inline uint32_t amg_edge_hash(uint32_t a, uint32_t b) {
  uint32_t h = (a * 0x9E3779B1u) ^ (b * 0x85EBCA77u);
  h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
  return h;
}
__host__ __device__ inline uint64_t amg_edge_key(int32_t i, int32_t j, int variant) {
  const uint32_t a = (uint32_t)(i < j ? i : j), b = (uint32_t)(i < j ? j : i);
  const uint32_t d = b > a ? b - a : 1u;
  const uint64_t k2 = variant == 1 ? (uint64_t)d : (uint64_t)(0x7FFFFFFFu - d);
  const uint64_t par = 1u - ((a / d) & 1u);
  return (k2 << 33) | (par << 32) | (uint64_t)amg_edge_hash(a, b);
}
// per row: sign of the diagonal, θ·max strength; *fail = 1 if a row's columns are not strictly ascending
__global__ __launch_bounds__(NK_BLOCK) void k_amg_strength(int64_t n, const int32_t *__restrict__ rp, const int32_t *__restrict__ ci,
                                                           const double *__restrict__ v, double theta, double *__restrict__ thr,
                                                           double *__restrict__ sgn, int *fail) {
  const int64_t i = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x;
  if (i >= n) return;
  double dg = 0.0;
  bool bad = false;
  for (int32_t k = rp[i]; k < rp[i + 1]; ++k) {
    if (ci[k] == i) dg += v[k];
    if (k > rp[i] && ci[k] <= ci[k - 1]) bad = true;
  }
  const double sg = dg < 0.0 ? -1.0 : 1.0;
  double smax = 0.0;
  for (int32_t k = rp[i]; k < rp[i + 1]; ++k)
    if (ci[k] != i) smax = fmax(smax, -v[k] * sg);
  thr[i] = theta * smax;
  sgn[i] = sg;
  if (bad) *fail = 1;
}
__global__ __launch_bounds__(NK_BLOCK) void k_amg_propose(int64_t n, const int32_t *__restrict__ rp, const int32_t *__restrict__ ci,
                                                          const double *__restrict__ v, const double *__restrict__ thr,
                                                          const double *__restrict__ sgn, const int32_t *__restrict__ match,
                                                          int32_t *__restrict__ best, int variant) {
  const int64_t i = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x;
  if (i >= n) return;
  int32_t bj = -1;
  if (match[i] < 0) {
    const double sg = sgn[i], th = thr[i];
    double bs = 0.0;
    uint64_t bk = 0;
    for (int32_t k = rp[i]; k < rp[i + 1]; ++k) {
      const int32_t j = ci[k];
      if (j == i) continue;
      const double s = -v[k] * sg;
      if (!(s > 0.0) || !(s >= th) || match[j] >= 0) continue;
      const uint64_t key = amg_edge_key((int32_t)i, j, variant);
      if (bj < 0 || s > bs || (s == bs && (key > bk || (key == bk && j > bj)))) { bs = s; bk = key; bj = j; }
    }
  }
  best[i] = bj;
}
__global__ __launch_bounds__(NK_BLOCK) void k_amg_accept(int64_t n, const int32_t *__restrict__ best, int32_t *__restrict__ match) {
  const int64_t i = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x;
  if (i >= n) return;
  const int32_t j = best[i];
  if (j >= 0 && best[j] == (int32_t)i) match[i] = j;   // (best is −1 for matched rows: only unmatched pairs get here)
}
__global__ __launch_bounds__(NK_BLOCK) void k_amg_rep_flag(int64_t n, const int32_t *__restrict__ match, int32_t *__restrict__ flag) {
  const int64_t i = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x;
  if (i > n) return;
  flag[i] = (i < n && (match[i] < 0 || (int32_t)i < match[i])) ? 1 : 0;   // (flag[n] = 0: the scan's last entry is the count)
}
// aggregates numbered in the order of their smallest row; pair[2I], pair[2I + 1]: the rows of aggregate I (−1: a singleton)
__global__ __launch_bounds__(NK_BLOCK) void k_amg_number(int64_t n, const int32_t *__restrict__ match, const int32_t *__restrict__ num,
                                                         int32_t *__restrict__ cid, int32_t *__restrict__ pair) {
  const int64_t i = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x;
  if (i >= n) return;
  const int32_t m = match[i];
  if (m < 0 || (int32_t)i < m) {
    const int32_t I = num[i];
    cid[i] = I;
    pair[2 * (int64_t)I] = (int32_t)i;
    pair[2 * (int64_t)I + 1] = m;
  } else {
    cid[i] = num[m];
  }
}
// the rows of every aggregate after one more pass: the union of its two parts' rows, ascending, −1 behind them (W = 2 Wp)
__global__ __launch_bounds__(NK_BLOCK) void k_amg_compose(int64_t nc, int Wp, const int32_t *__restrict__ memp,
                                                          const int32_t *__restrict__ pair, int32_t *__restrict__ mem) {
  const int64_t I = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x;
  if (I >= nc) return;
  int32_t buf[16];
  int cnt = 0;
  for (int h = 0; h < 2; ++h) {
    const int32_t a = pair[2 * I + h];
    if (a < 0) continue;
    for (int t = 0; t < Wp; ++t) {
      const int32_t r = memp[(int64_t)a * Wp + t];
      if (r < 0) break;
      int p = cnt++;
      for (; p > 0 && buf[p - 1] > r; --p) buf[p] = buf[p - 1];
      buf[p] = r;
    }
  }
  for (int t = 0; t < 2 * Wp; ++t) mem[I * 2 * Wp + t] = t < cnt ? buf[t] : -1;
}
__global__ __launch_bounds__(NK_BLOCK) void k_amg_map(int64_t n, const int32_t *__restrict__ cid, int32_t *__restrict__ agg) {
  const int64_t i = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x;
  if (i < n) agg[i] = cid[agg[i]];
}
// members → number of keys (fine entries) / of rows per aggregate; entry nc of both = 0 (the scans' totals)
__global__ __launch_bounds__(NK_BLOCK) void k_amg_keycount(int64_t nc, int W, const int32_t *__restrict__ mem, const int32_t *__restrict__ rp,
                                                           int32_t *__restrict__ len, int32_t *__restrict__ rows) {
  const int64_t I = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x;
  if (I > nc) return;
  int32_t L = 0, R = 0;
  if (I < nc)
    for (int t = 0; t < W; ++t) {
      const int32_t r = mem[I * W + t];
      if (r < 0) break;
      L += rp[r + 1] - rp[r];
      ++R;
    }
  len[I] = L;
  if (rows) rows[I] = R;
}
__global__ __launch_bounds__(NK_BLOCK) void k_amg_keysort(int64_t nc, int W, const int32_t *__restrict__ mem, const int32_t *__restrict__ rp,
                                                          const int32_t *__restrict__ ci, const int32_t *__restrict__ cid,
                                                          const int32_t *__restrict__ kptr, uint64_t *__restrict__ keys,
                                                          int32_t *__restrict__ cnt) {
  const int64_t I = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x;
  if (I > nc) return;
  if (I == nc) { cnt[I] = 0; return; }
  uint64_t *K = keys + kptr[I];
  int L = 0;
  for (int t = 0; t < W; ++t) {
    const int32_t r = mem[I * W + t];
    if (r < 0) break;
    for (int32_t k = rp[r]; k < rp[r + 1]; ++k) K[L++] = ((uint64_t)(uint32_t)cid[ci[k]] << 32) | (uint32_t)k;
  }
  // Shell sort (all keys distinct: the fine position is part of them)
  const int gaps[7] = {301, 132, 57, 23, 10, 4, 1};
  for (int g = 0; g < 7; ++g) {
    const int gap = gaps[g];
    if (gap >= L && gap > 1) continue;
    for (int a = gap; a < L; ++a) {
      const uint64_t x = K[a];
      int b = a;
      for (; b >= gap && K[b - gap] > x; b -= gap) K[b] = K[b - gap];
      K[b] = x;
    }
  }
  int32_t c = 0;
  for (int t = 0; t < L; ++t) c += (t == 0 || (K[t] >> 32) != (K[t - 1] >> 32)) ? 1 : 0;
  cnt[I] = c;
}
__global__ __launch_bounds__(NK_BLOCK) void k_amg_keyfill(int64_t nc, const int32_t *__restrict__ kptr, const uint64_t *__restrict__ keys,
                                                          const int32_t *__restrict__ crp, int32_t *__restrict__ cci,
                                                          int32_t *__restrict__ gptr, int32_t *__restrict__ gidx) {
  const int64_t I = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x;
  if (I >= nc) return;
  const int32_t k0 = kptr[I], k1 = kptr[I + 1];
  int32_t e = crp[I] - 1;
  for (int32_t t = k0; t < k1; ++t) {
    const uint64_t x = keys[t];
    if (t == k0 || (x >> 32) != (keys[t - 1] >> 32)) {
      ++e;
      cci[e] = (int32_t)(x >> 32);
      gptr[e] = t;
    }
    gidx[t] = (int32_t)(uint32_t)x;
  }
  if (I == nc - 1) gptr[crp[nc]] = k1;
}
__global__ __launch_bounds__(NK_BLOCK) void k_amg_members_fill(int64_t nc, int W, const int32_t *__restrict__ mem,
                                                               const int32_t *__restrict__ aptr, int32_t *__restrict__ amem) {
  const int64_t I = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x;
  if (I >= nc) return;
  int32_t o = aptr[I];
  for (int t = 0; t < W; ++t) {
    const int32_t r = mem[I * W + t];
    if (r < 0) break;
    amem[o++] = r;
  }
}
__global__ __launch_bounds__(NK_BLOCK) void k_amg_iota(int64_t n, int32_t *__restrict__ a) {
  const int64_t i = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x;
  if (i < n) a[i] = (int32_t)i;
}

// one allocation for the set-up's temporaries (bump-allocated, rewound per level)
struct amg_arena {
  char *base = nullptr;
  size_t cap = 0, top = 0;
  ~amg_arena() { hipFree(base); }
  template <class T>
  int get(T **p, size_t count) {
    const size_t bytes = (count * sizeof(T) + 255) & ~(size_t)255;
    if (top + bytes > cap) NK_FAIL(NK_E_NOMEM, "AMG set-up: the temporary arena (%zu MB) is too small", cap >> 20);
    *p = reinterpret_cast<T *>(base + top);
    top += bytes;
    return NK_OK;
  }
};
static inline dim3 amg_g1(int64_t n) { return dim3((unsigned)((n + NK_BLOCK - 1) / NK_BLOCK > 0 ? (n + NK_BLOCK - 1) / NK_BLOCK : 1)); }
// out[0 .. n) = exclusive prefix sums of in[0 .. n) (n includes the caller's trailing zero: out[n − 1] is the total).
// An in-tree scan (no vendor primitive in the shipped path): a workgroup scans 2048 entries — eight per thread in registers, the
// wavefront's 64 thread sums by a shuffle ladder, the four wavefront sums through LDS — and leaves its total; the totals are
// scanned by the same routine (two levels serve 4 M entries, three 8 G) and added back. in == out is allowed.
constexpr int AMG_SCAN_PER = 8, AMG_SCAN_TILE = NK_BLOCK * AMG_SCAN_PER;
__global__ __launch_bounds__(NK_BLOCK) void k_amg_scan_tile(const int32_t *in, int32_t *out, int64_t n,
                                                            int32_t *__restrict__ totals) {
  __shared__ int32_t s_w[NK_BLOCK / 64];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int64_t base = (int64_t)blockIdx.x * AMG_SCAN_TILE + (int64_t)t * AMG_SCAN_PER;
  int32_t v[AMG_SCAN_PER];
#pragma unroll
  for (int q = 0; q < AMG_SCAN_PER; ++q) v[q] = (base + q < n) ? in[base + q] : 0;
  int32_t mine = 0;
#pragma unroll
  for (int q = 0; q < AMG_SCAN_PER; ++q) { const int32_t x = v[q]; v[q] = mine; mine += x; }   // exclusive within the thread
  int32_t incl = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int32_t up = __shfl_up(incl, o, 64);
    if (lane >= o) incl += up;
  }
  if (lane == 63) s_w[wv] = incl;
  __syncthreads();
  int32_t off = incl - mine;
  for (int w = 0; w < wv; ++w) off += s_w[w];
#pragma unroll
  for (int q = 0; q < AMG_SCAN_PER; ++q)
    if (base + q < n) out[base + q] = v[q] + off;
  if (totals != nullptr && t == NK_BLOCK - 1) totals[blockIdx.x] = off + mine;
}
__global__ __launch_bounds__(NK_BLOCK) void k_amg_scan_add(int32_t *__restrict__ out, int64_t n, const int32_t *__restrict__ offs) {
  const int32_t o = offs[blockIdx.x];
  const int64_t base = (int64_t)blockIdx.x * AMG_SCAN_TILE;
  for (int q = threadIdx.x; q < AMG_SCAN_TILE; q += NK_BLOCK)
    if (base + q < n) out[base + q] += o;
}
static int amg_scan(nk_ctx *ctx, amg_arena &ar, const int32_t *in, int32_t *out, int64_t n) {
  if (n <= 0) return NK_OK;
  const int64_t nb = (n + AMG_SCAN_TILE - 1) / AMG_SCAN_TILE;
  int32_t *totals = nullptr;
  if (nb > 1) NK_TRY(ar.get(&totals, (size_t)nb));
  hipLaunchKernelGGL(k_amg_scan_tile, dim3((unsigned)nb), dim3(NK_BLOCK), 0, ctx->stream, in, out, n, totals);
  if (nb > 1) {
    NK_TRY(amg_scan(ctx, ar, totals, totals, nb));
    hipLaunchKernelGGL(k_amg_scan_add, dim3((unsigned)nb), dim3(NK_BLOCK), 0, ctx->stream, out, n, (const int32_t *)totals);
  }
  NK_HIP(hipGetLastError());
  return NK_OK;
}
static int amg_fetch_int(nk_ctx *ctx, const int32_t *d, int32_t *h) {
  NK_HIP(hipMemcpyAsync(h, d, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  NK_HIP(hipStreamSynchronize(ctx->stream));
  return NK_OK;
}
struct amg_dcsr { int64_t n = 0, nnz = 0; const int32_t *rp = nullptr, *ci = nullptr; const double *v = nullptr; };
// one pairwise pass: cid (n), pair (2 nc), nc
static int amg_handshake_pass_dev(nk_ctx *ctx, amg_arena &ar, const amg_dcsr &F, double theta, int variant, int32_t **cid_out,
                                  int32_t **pair_out, int32_t *nc_out, int *d_fail) {
  const int64_t n = F.n;
  double *thr, *sgn;
  int32_t *match, *best, *flag, *num, *cid, *pair;
  NK_TRY(ar.get(&thr, n)); NK_TRY(ar.get(&sgn, n));
  NK_TRY(ar.get(&match, n)); NK_TRY(ar.get(&best, n)); NK_TRY(ar.get(&flag, n + 1)); NK_TRY(ar.get(&num, n + 1));
  NK_TRY(ar.get(&cid, n));
  NK_LAUNCH(ctx, k_amg_strength, amg_g1(n), dim3(NK_BLOCK), n, F.rp, F.ci, F.v, theta, thr, sgn, d_fail);
  NK_HIP(hipMemsetAsync(match, 0xFF, n * sizeof(int32_t), ctx->stream));
  for (int r = 0; r < AMG_HS_ROUNDS; ++r) {
    NK_LAUNCH(ctx, k_amg_propose, amg_g1(n), dim3(NK_BLOCK), n, F.rp, F.ci, F.v, (const double *)thr, (const double *)sgn,
              (const int32_t *)match, best, variant);
    NK_LAUNCH(ctx, k_amg_accept, amg_g1(n), dim3(NK_BLOCK), n, (const int32_t *)best, match);
  }
  NK_LAUNCH(ctx, k_amg_rep_flag, amg_g1(n + 1), dim3(NK_BLOCK), n, (const int32_t *)match, flag);
  NK_TRY(amg_scan(ctx, ar, flag, num, n + 1));
  int32_t nc = 0;
  NK_TRY(amg_fetch_int(ctx, num + n, &nc));
  NK_TRY(ar.get(&pair, 2 * (size_t)nc + 2));
  NK_LAUNCH(ctx, k_amg_number, amg_g1(n), dim3(NK_BLOCK), n, (const int32_t *)match, (const int32_t *)num, cid, pair);
  *cid_out = cid; *pair_out = pair; *nc_out = nc;
  return NK_OK;
}
// the Galerkin plan of the aggregates (cid: row → aggregate, mem: nc × W member rows) on the pattern of F:
// crp / cci (pattern of TᵀF T), gptr / gidx (coarse entry → the fine entries it sums). `persist`: gptr / gidx are allocations of
// their own (kept by the level), else arena memory.
struct amg_plan { int32_t nc = 0, nnzc = 0; int32_t *crp = nullptr, *cci = nullptr, *gptr = nullptr, *gidx = nullptr; };
static int amg_galerkin_plan_dev(nk_ctx *ctx, amg_arena &ar, const amg_dcsr &F, const int32_t *cid, const int32_t *mem, int W, int32_t nc,
                                 bool persist, amg_plan *out, int32_t **persist_gptr = nullptr, int32_t **persist_gidx = nullptr) {
  int32_t *len, *kptr, *cnt, *crp;
  uint64_t *keys;
  NK_TRY(ar.get(&len, (size_t)nc + 1)); NK_TRY(ar.get(&kptr, (size_t)nc + 1)); NK_TRY(ar.get(&cnt, (size_t)nc + 1));
  NK_TRY(ar.get(&crp, (size_t)nc + 1)); NK_TRY(ar.get(&keys, (size_t)F.nnz + 1));
  NK_LAUNCH(ctx, k_amg_keycount, amg_g1((int64_t)nc + 1), dim3(NK_BLOCK), (int64_t)nc, W, mem, F.rp, len, (int32_t *)nullptr);
  NK_TRY(amg_scan(ctx, ar, len, kptr, (int64_t)nc + 1));
  NK_LAUNCH(ctx, k_amg_keysort, amg_g1((int64_t)nc + 1), dim3(NK_BLOCK), (int64_t)nc, W, mem, F.rp, F.ci, cid, (const int32_t *)kptr, keys, cnt);
  NK_TRY(amg_scan(ctx, ar, cnt, crp, (int64_t)nc + 1));
  int32_t nnzc = 0;
  NK_TRY(amg_fetch_int(ctx, crp + nc, &nnzc));
  int32_t *cci, *gptr = nullptr, *gidx = nullptr;
  NK_TRY(ar.get(&cci, (size_t)nnzc + 1));
  if (persist) {   // (the caller's level owns them from the moment they exist: nk_amg_destroy frees what a failed set-up leaves)
    NK_TRY(nk_dev_alloc(persist_gptr, (size_t)nnzc + 2));
    NK_TRY(nk_dev_alloc(persist_gidx, (size_t)F.nnz + 1));
    gptr = *persist_gptr; gidx = *persist_gidx;
  } else {
    NK_TRY(ar.get(&gptr, (size_t)nnzc + 2)); NK_TRY(ar.get(&gidx, (size_t)F.nnz + 1));
  }
  out->gptr = gptr; out->gidx = gidx;   // (owned by the caller from here on)
  NK_LAUNCH(ctx, k_amg_keyfill, amg_g1(nc), dim3(NK_BLOCK), (int64_t)nc, (const int32_t *)kptr, (const uint64_t *)keys, (const int32_t *)crp,
            cci, gptr, gidx);
  out->nc = nc; out->nnzc = nnzc; out->crp = crp; out->cci = cci;
  return NK_OK;
}
// `passes` pairwise passes on F with one tie-break variant: agg (n), mem (nc × 2^passes), nc — arena memory
static int amg_coarsen_dev(nk_ctx *ctx, amg_arena &ar, const amg_dcsr &F, const nk_amg_params &prm, int variant, int32_t **agg_out,
                           int32_t **mem_out, int32_t *nc_out, int *d_fail) {
  amg_dcsr cur = F;
  int32_t *agg = nullptr, *mem = nullptr, nc = (int32_t)F.n;
  int W = 1;
  for (int p = 0; p < prm.passes; ++p) {
    int32_t *cid, *pair;
    NK_TRY(amg_handshake_pass_dev(ctx, ar, cur, prm.theta, variant, &cid, &pair, &nc, d_fail));
    if (p == 0) {
      agg = cid;       // (row → aggregate of the first pass; later passes map it on)
      mem = pair;
    } else {
      int32_t *m2;
      NK_TRY(ar.get(&m2, (size_t)nc * 2 * W + 2));
      NK_LAUNCH(ctx, k_amg_compose, amg_g1(nc), dim3(NK_BLOCK), (int64_t)nc, W, (const int32_t *)mem, (const int32_t *)pair, m2);
      NK_LAUNCH(ctx, k_amg_map, amg_g1(F.n), dim3(NK_BLOCK), F.n, (const int32_t *)cid, agg);
      mem = m2;
    }
    W *= 2;
    if (p + 1 < prm.passes) {   // the next pass matches on the Galerkin matrix of this pass's pairs
      amg_plan pl;
      NK_TRY(amg_galerkin_plan_dev(ctx, ar, cur, cid, pair, 2, nc, false, &pl));
      double *cv;
      NK_TRY(ar.get(&cv, (size_t)pl.nnzc + 1));
      NK_LAUNCH(ctx, k_amg_galerkin, dim3(nk_grid_for(pl.nnzc, NK_BLOCK, 4096)), dim3(NK_BLOCK), (int64_t)pl.nnzc, (const int32_t *)pl.gptr,
                (const int32_t *)pl.gidx, cur.v, cv);
      cur.n = nc; cur.nnz = pl.nnzc; cur.rp = pl.crp; cur.ci = pl.cci; cur.v = cv;
    }
  }
  *agg_out = agg; *mem_out = mem; *nc_out = nc;
  return NK_OK;
}
static int amg_setup_device(nk_amg *M, const std::function<void(const char *)> &lap) {
  nk_ctx *ctx = M->ctx;
  nk_csr *A = M->A;
  const nk_amg_params &prm = M->prm;
  const int W = 1 << prm.passes;
  amg_arena ar;
  ar.cap = (size_t)256 * (size_t)A->nrows + (size_t)96 * (size_t)A->nnz + ((size_t)64 << 20);
  NK_HIP(hipMalloc((void **)&ar.base, ar.cap));
  int *d_fail = nullptr;
  NK_TRY(nk_dev_alloc(&d_fail, (size_t)2));
  auto fguard = nk_make_guard(d_fail, [](int *p) { hipFree(p); });
  NK_HIP(hipMemsetAsync(d_fail, 0, 2 * sizeof(int), ctx->stream));
  M->lv.emplace_back();
  {
    amg_level &L0 = M->lv.back();
    L0.n = A->nrows; L0.nnz = A->nnz; L0.A = A;
  }
  lap("arena");
  while (M->lv.back().n > prm.coarse_max && (int)M->lv.size() < 24) {
    ar.top = 0;
    const int l = (int)M->lv.size() - 1;
    nk_csr *FA = M->lv[l].A;
    amg_dcsr F;
    F.n = FA->nrows; F.nnz = FA->nnz; F.rp = FA->d_rowptr; F.ci = FA->d_col; F.v = FA->d_val;
    int32_t *agg = nullptr, *mem = nullptr, nc = 0;
    NK_TRY(amg_coarsen_dev(ctx, ar, F, prm, 0, &agg, &mem, &nc, d_fail));
    if (l == 0) {
      int fail = 0;
      NK_HIP(nk_memcpy(ctx, &fail, d_fail, sizeof(int), hipMemcpyDeviceToHost));
      NK_REQUIRE(!fail, "AMG: the columns of some row are not sorted / unique");
    }
    if ((int64_t)nc * W > F.n + F.n / 50) {   // not (nearly) a full coarsening: the other tie-break may align the pairs better
      int32_t *agg1 = nullptr, *mem1 = nullptr, nc1 = 0;
      NK_TRY(amg_coarsen_dev(ctx, ar, F, prm, 1, &agg1, &mem1, &nc1, d_fail));
      if (nc1 < nc) { agg = agg1; mem = mem1; nc = nc1; }
    }
    lap("  matching");
    if ((double)nc > 0.8 * (double)F.n) break;   // coarsening stalled: this level is the coarsest
    amg_plan pl;
    amg_level &L = M->lv[l];
    NK_TRY(amg_galerkin_plan_dev(ctx, ar, F, agg, mem, W, nc, true, &pl, &L.d_gptr, &L.d_gidx));
    L.nc = nc;
    // the aggregates: row → coarse row, coarse row → its rows
    NK_TRY(nk_dev_alloc(&L.d_agg, (size_t)F.n + 1));
    NK_HIP(hipMemcpyAsync(L.d_agg, agg, F.n * sizeof(int32_t), hipMemcpyDeviceToDevice, ctx->stream));
    {
      int32_t *len, *rows;
      NK_TRY(ar.get(&len, (size_t)nc + 1)); NK_TRY(ar.get(&rows, (size_t)nc + 1));
      NK_LAUNCH(ctx, k_amg_keycount, amg_g1((int64_t)nc + 1), dim3(NK_BLOCK), (int64_t)nc, W, (const int32_t *)mem, F.rp, len, rows);
      NK_TRY(nk_dev_alloc(&L.d_aggptr, (size_t)nc + 2));
      NK_TRY(nk_dev_alloc(&L.d_aggmem, (size_t)F.n + 1));
      NK_TRY(amg_scan(ctx, ar, rows, L.d_aggptr, (int64_t)nc + 1));
      NK_LAUNCH(ctx, k_amg_members_fill, amg_g1(nc), dim3(NK_BLOCK), (int64_t)nc, W, (const int32_t *)mem, (const int32_t *)L.d_aggptr, L.d_aggmem);
    }
    lap("  level Galerkin plan");
    // the next level's matrix object (its pattern through the host: the SpMV's row-block descriptors are built there)
    std::vector<int32_t> hrp((size_t)nc + 1), hci((size_t)pl.nnzc);
    NK_HIP(hipMemcpyAsync(hrp.data(), pl.crp, ((size_t)nc + 1) * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    if (pl.nnzc) NK_HIP(hipMemcpyAsync(hci.data(), pl.cci, (size_t)pl.nnzc * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    NK_HIP(hipStreamSynchronize(ctx->stream));
    std::vector<int64_t> gc(hci.begin(), hci.end());
    M->lv.emplace_back();
    amg_level &C = M->lv.back();
    C.n = nc; C.nnz = pl.nnzc;
    NK_TRY(nk_csr_create_local(ctx, nc, nc, 0, hrp, gc, nullptr, &C.A, true));
    C.own_A = true;
    NK_LAUNCH(ctx, k_amg_galerkin, dim3(nk_grid_for(pl.nnzc, NK_BLOCK, 4096)), dim3(NK_BLOCK), (int64_t)pl.nnzc, (const int32_t *)pl.gptr,
              (const int32_t *)pl.gidx, F.v, C.A->d_val);
    lap("  next level's matrix");
  }
  NK_HIP(hipStreamSynchronize(ctx->stream));
  // work vectors
  for (size_t l = 0; l < M->lv.size(); ++l) {
    amg_level &L = M->lv[l];
    NK_TRY(nk_dev_alloc(&L.d_dinv, (size_t)L.n + 1));
    NK_TRY(nk_dev_alloc(&L.r, (size_t)L.n + 1));
    NK_TRY(nk_dev_alloc(&L.d0, (size_t)L.n + 1));
    NK_TRY(nk_dev_alloc(&L.d1, (size_t)L.n + 1));
    if (l > 0) {
      NK_TRY(nk_dev_alloc(&L.b, (size_t)L.n + 1));
      NK_TRY(nk_dev_alloc(&L.x, (size_t)L.n + 1));
    }
  }
  lap("work vectors");
  return NK_OK;
}

// ----------------------------------------------------------------------------- create / destroy
void nk_amg_destroy(nk_amg *M) {
  if (!M) return;
  for (size_t l = 0; l < M->lv.size(); ++l) {
    amg_level &L = M->lv[l];
    if (L.own_A && L.A) nk_csr_destroy(L.A);
    hipFree(L.d_agg); hipFree(L.d_aggptr); hipFree(L.d_aggmem); hipFree(L.d_gptr); hipFree(L.d_gidx); hipFree(L.d_dinv);
    if (l > 0) { hipFree(L.b); hipFree(L.x); }
    hipFree(L.r); hipFree(L.d0); hipFree(L.d1);
  }
  hipFree(M->d_src0); hipFree(M->d_inv); hipFree(M->d_lmax); hipFree(M->d_lpart); hipFree(M->d_fail); hipFree(M->d_nparts);
  delete M;
}
template <class T>
static int upload(nk_ctx *ctx, T **dst, const std::vector<T> &src) {
  NK_TRY(nk_dev_alloc(dst, src.size() + 1));
  if (!src.empty()) NK_HIP(nk_memcpy(ctx, *dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
  return NK_OK;
}
int nk_amg_create(nk_csr *A, const nk_amg_params *prm, nk_amg **out) {
  nk_ctx *ctx = A->ctx;
  nk_amg *M = new nk_amg();
  auto guard = nk_make_guard(M, [](nk_amg *m) { nk_amg_destroy(m); });
  M->ctx = ctx;
  M->A = A;
  nk_amg_params_default(&M->prm);
  if (prm) {
    if (prm->nu > 0) M->prm.nu = prm->nu;
    if (prm->passes > 0) M->prm.passes = prm->passes;
    if (prm->matching > 0) M->prm.matching = prm->matching;
    if (prm->coarse_max > 0) M->prm.coarse_max = prm->coarse_max;
    if (prm->theta > 0.0) M->prm.theta = prm->theta;
    if (prm->overcorrection > 0.0) M->prm.overcorrection = prm->overcorrection;
    if (prm->cheb_ratio > 1.0) M->prm.cheb_ratio = prm->cheb_ratio;
  }
  NK_REQUIRE(M->prm.nu <= 16 && M->prm.passes <= 4 && M->prm.coarse_max <= 128, "AMG: nu ≤ 16, passes ≤ 4, coarse_max ≤ 128");
  NK_REQUIRE(M->prm.matching >= 0 && M->prm.matching <= 2, "AMG: matching is 0 (automatic), 1 (sequential, host) or 2 (handshake, device)");
  const int64_t n = A->nrows;
  // (NK_AMG_TIMING=1: the set-up's phases on stderr)
  static const bool timing = getenv("NK_AMG_TIMING") && atoi(getenv("NK_AMG_TIMING")) != 0;
  auto tnow = [] { return std::chrono::steady_clock::now(); };
  auto t_last = tnow();
  auto lap = [&](const char *what) {
    if (!timing) return;
    hipStreamSynchronize(ctx->stream);
    const auto t = tnow();
    fprintf(stderr, "[amg set-up] %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t - t_last).count());
    t_last = t;
  };
  // where the hierarchy is set up: on the device (handshake matching) for a matrix without halo columns, on the host (the
  // sequential pairwise pass) for a rank's local block of a distributed matrix — params.matching 1 / 2 or NK_AMG_SETUP=host /
  // device ask for one of them
  const bool has_halo = !A->halo_gcols.empty();
  bool on_device = !has_halo;
  if (M->prm.matching == 1) on_device = false;
  if (const char *e = getenv("NK_AMG_SETUP")) {
    if (!strcmp(e, "host")) on_device = false;
    else if (!strcmp(e, "device")) on_device = !has_halo;
  }
  NK_REQUIRE(!(M->prm.matching == 2 && has_halo), "AMG: the device set-up (matching = 2) takes a matrix without halo columns");
  M->matching = on_device ? 2 : 1;
  if (on_device) {
    NK_TRY(amg_setup_device(M, lap));
  } else {
    // ---- level 0 on the host: the rank's local square block with the values of this moment
    host_csr H;
    H.n = n;
    std::vector<double> vals((size_t)A->nnz);
    NK_HIP(hipStreamSynchronize(ctx->stream));
    if (A->nnz) NK_HIP(nk_memcpy(ctx, vals.data(), A->d_val, A->nnz * sizeof(double), hipMemcpyDeviceToHost));
    std::vector<int32_t> src0;
    if (!has_halo) {   // the whole matrix: its pattern and values as they are
      H.rp.assign(A->h_rowptr.begin(), A->h_rowptr.end());
      H.ci.assign(A->h_col.begin(), A->h_col.end());
      H.v = std::move(vals);
    } else {
      H.rp.assign(n + 1, 0);
      H.ci.reserve((size_t)A->nnz); H.v.reserve((size_t)A->nnz); src0.reserve((size_t)A->nnz);
      for (int64_t r = 0; r < n; ++r) {
        for (int32_t k = A->h_rowptr[r]; k < A->h_rowptr[r + 1]; ++k)
          if (A->h_col[k] < n) { H.ci.push_back(A->h_col[k]); H.v.push_back(vals[k]); src0.push_back(k); }
        H.rp[r + 1] = (int32_t)H.ci.size();
      }
    }
    for (int64_t r = 0; r < n; ++r)   // sorted columns are part of the contract of the aggregation order
      for (int32_t k = H.rp[r] + 1; k < H.rp[r + 1]; ++k)
        NK_REQUIRE(H.ci[k] > H.ci[k - 1], "AMG: the columns of row %lld are not sorted / unique", (long long)r);
    lap("level 0 to the host");
    // ---- coarsen
    std::vector<host_csr> mats;
    mats.push_back(std::move(H));
    std::vector<std::vector<int32_t>> aggs, emaps;
    while (mats.back().n > M->prm.coarse_max && (int)mats.size() < 24) {
      const host_csr &F = mats.back();
      std::vector<int32_t> agg((size_t)F.n);
      std::iota(agg.begin(), agg.end(), 0);
      host_csr cur = F;
      int32_t nc = 0;
      for (int p = 0; p < M->prm.passes; ++p) {
        std::vector<int32_t> cid, em;
        pairwise_pass(cur, M->prm.theta, cid, nc);
        lap("  pairwise pass");
        host_csr nxt;
        galerkin_host(cur, cid, nc, nxt, em);
        lap("  pass Galerkin");
        cur = std::move(nxt);
        for (auto &a : agg) a = cid[a];
      }
      if ((double)nc > 0.8 * (double)F.n) break;   // coarsening stalled: this level is the coarsest
      host_csr C;
      std::vector<int32_t> emap;
      galerkin_host(F, agg, nc, C, emap);          // one-stage sums: what a value refresh recomputes
      lap("  level Galerkin");
      aggs.push_back(std::move(agg));
      emaps.push_back(std::move(emap));
      mats.push_back(std::move(C));
    }
    lap("coarsening (total above)");
    // ---- device objects
    const int nlev = (int)mats.size();   // (of this branch; the same number as below)
    M->lv.resize(nlev);
    for (int l = 0; l < nlev; ++l) {
      amg_level &L = M->lv[l];
      const host_csr &F = mats[l];
      L.n = F.n;
      L.nnz = (int64_t)F.ci.size();
      if (l == 0 && !has_halo) {
        L.A = A;
      } else {
        std::vector<int64_t> gc(F.ci.begin(), F.ci.end());
        NK_TRY(nk_csr_create_local(ctx, F.n, F.n, 0, F.rp, gc, nullptr, &L.A, true));
        L.own_A = true;
      }
      NK_TRY(nk_dev_alloc(&L.d_dinv, (size_t)L.n + 1));
      NK_TRY(nk_dev_alloc(&L.r, (size_t)L.n + 1));
      NK_TRY(nk_dev_alloc(&L.d0, (size_t)L.n + 1));
      NK_TRY(nk_dev_alloc(&L.d1, (size_t)L.n + 1));
      if (l > 0) {
        NK_TRY(nk_dev_alloc(&L.b, (size_t)L.n + 1));
        NK_TRY(nk_dev_alloc(&L.x, (size_t)L.n + 1));
      }
      if (l + 1 < nlev) {
        const std::vector<int32_t> &agg = aggs[l], &emap = emaps[l];
        L.nc = mats[l + 1].n;
        L.h_agg = agg;
        std::vector<int32_t> aptr(L.nc + 1, 0), amem((size_t)L.n);
        for (int64_t i = 0; i < L.n; ++i) aptr[agg[i] + 1]++;
        for (int64_t I = 0; I < L.nc; ++I) aptr[I + 1] += aptr[I];
        {
          std::vector<int32_t> fill(aptr.begin(), aptr.end() - 1);
          for (int64_t i = 0; i < L.n; ++i) amem[fill[agg[i]]++] = (int32_t)i;
        }
        const int64_t nnzc = (int64_t)mats[l + 1].ci.size();
        std::vector<int32_t> gptr(nnzc + 1, 0), gidx((size_t)L.nnz);
        for (int64_t k = 0; k < L.nnz; ++k) gptr[emap[k] + 1]++;
        for (int64_t e = 0; e < nnzc; ++e) gptr[e + 1] += gptr[e];
        {
          std::vector<int32_t> fill(gptr.begin(), gptr.end() - 1);
          for (int64_t k = 0; k < L.nnz; ++k) gidx[fill[emap[k]]++] = (int32_t)k;
        }
        NK_TRY(upload(M->ctx, &L.d_agg, agg));
        NK_TRY(upload(M->ctx, &L.d_aggptr, aptr));
        NK_TRY(upload(M->ctx, &L.d_aggmem, amem));
        NK_TRY(upload(M->ctx, &L.d_gptr, gptr));
        NK_TRY(upload(M->ctx, &L.d_gidx, gidx));
      }
    }
    if (has_halo) NK_TRY(upload(M->ctx, &M->d_src0, src0));
    lap("device objects");
  }
  const int nlev = (int)M->lv.size();
  M->dense = M->lv.back().n <= M->prm.coarse_max;
  M->ldinv = (int)((M->lv.back().n + 7) / 8 * 8);
  if (M->dense && M->lv.back().n > 0) NK_TRY(nk_dev_alloc(&M->d_inv, (size_t)M->ldinv * M->ldinv));
  M->lpart_stride = 1024;
  M->nparts.assign(nlev, 0);
  for (int l = 0; l < nlev; ++l) M->nparts[l] = M->lv[l].n > 0 ? nk_grid_for(M->lv[l].n, NK_BLOCK, M->lpart_stride) : 0;
  NK_TRY(upload(M->ctx, &M->d_nparts, M->nparts));
  NK_TRY(nk_dev_alloc(&M->d_lpart, (size_t)nlev * M->lpart_stride));
  NK_TRY(nk_dev_alloc(&M->d_lmax, (size_t)nlev + 1));
  NK_TRY(nk_dev_alloc(&M->d_fail, (size_t)2));
  NK_TRY(nk_amg_update(M));
  lap("first numeric update");
  *out = guard.release();
  return NK_OK;
}

// the numbers of the hierarchy for the matrix's CURRENT values (same pattern): Galerkin sums, D⁻¹, λmax, the coarse inverse
int nk_amg_update(nk_amg *M) {
  nk_ctx *ctx = M->ctx;
  const int nlev = (int)M->lv.size();
  M->ready = false;
  NK_HIP(hipMemsetAsync(M->d_fail, 0, 2 * sizeof(int), ctx->stream));
  if (M->d_src0 && M->lv[0].nnz > 0)
    NK_LAUNCH(ctx, k_amg_gather, dim3(nk_grid_for(M->lv[0].nnz, NK_BLOCK * 4, 4096)), dim3(NK_BLOCK), M->lv[0].nnz,
              (const int32_t *)M->d_src0, (const double *)M->A->d_val, M->lv[0].A->d_val);
  for (int l = 0; l < nlev; ++l) {
    amg_level &L = M->lv[l];
    if (L.n == 0) continue;
    nk_prof_scope prof_(ctx, NK_K_OTHER, 12.0 * (double)L.nnz + 12.0 * (double)L.n);
    NK_LAUNCH(ctx, k_amg_diag, dim3(M->nparts[l]), dim3(NK_BLOCK), L.n, (const int32_t *)L.A->d_rowptr, (const int32_t *)L.A->d_col,
              (const double *)L.A->d_val, L.d_dinv, M->d_lpart + (size_t)l * M->lpart_stride, M->d_fail);
    if (l + 1 < nlev) {
      amg_level &Cn = M->lv[l + 1];
      NK_LAUNCH(ctx, k_amg_galerkin, dim3(nk_grid_for(Cn.nnz, NK_BLOCK, 4096)), dim3(NK_BLOCK), Cn.nnz, (const int32_t *)L.d_gptr,
                (const int32_t *)L.d_gidx, (const double *)L.A->d_val, Cn.A->d_val);
      Cn.A->t_values_stale = true; Cn.A->bounds_valid = false; Cn.A->bounds_pending = false;
    }
  }
  NK_LAUNCH(ctx, k_amg_lmax_final, dim3(1), dim3(NK_BLOCK), nlev, M->lpart_stride, (const int *)M->d_nparts, (const double *)M->d_lpart,
            M->d_lmax);
  amg_level &Lc = M->lv.back();
  if (M->dense && Lc.n > 0) {
    NK_LAUNCH(ctx, k_amg_dense_fill, dim3(1), dim3(NK_BLOCK), (int)Lc.n, M->ldinv, (const int32_t *)Lc.A->d_rowptr,
              (const int32_t *)Lc.A->d_col, (const double *)Lc.A->d_val, M->d_inv);
    NK_TRY(nk_dense_invert128_dev(ctx, M->d_inv, M->ldinv, M->ldinv, M->d_fail + 1));
  }
  std::vector<double> lm(nlev, 1.0);
  int fail[2] = {0, 0};
  NK_HIP(hipMemcpyAsync(lm.data(), M->d_lmax, nlev * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  NK_HIP(hipMemcpyAsync(fail, M->d_fail, 2 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  NK_HIP(hipStreamSynchronize(ctx->stream));
  NK_HIP(hipGetLastError());
  if (fail[0]) NK_FAIL(NK_E_SINGULAR, "AMG: zero or non-finite diagonal entry on some level");
  if (fail[1]) NK_FAIL(NK_E_SINGULAR, "AMG: the coarsest-level matrix is singular");
  for (int l = 0; l < nlev; ++l) {
    if (!(lm[l] > 0.0) || !std::isfinite(lm[l])) NK_FAIL(NK_E_SINGULAR, "AMG: no finite Gershgorin bound on level %d", l);
    M->lv[l].lmax = lm[l];
  }
  M->ready = true;
  return NK_OK;
}

// ----------------------------------------------------------------------------- apply: one V-cycle
// steps 2 … nsteps of the Chebyshev iteration on level L (the first step is k_amg_cheb_first); the current correction ends in *dcur
static int amg_cheb_rest(nk_amg *M, amg_level &L, int nsteps, double **dcur, double **dnext, const int *d_skip) {
  const double lmin = L.lmax / M->prm.cheb_ratio, theta = 0.5 * (L.lmax + lmin), delta = 0.5 * (L.lmax - lmin);
  const double sigma1 = theta / delta;
  double rho = 1.0 / sigma1;
  for (int k = 1; k < nsteps; ++k) {
    const double rho_new = 1.0 / (2.0 * sigma1 - rho);
    nk_spmv_epi ep;
    ep.mode = 1;
    ep.c1 = rho_new * rho;
    ep.c2 = 2.0 * rho_new / delta;
    ep.r = L.r;
    ep.dnew = *dnext;
    ep.yacc = L.x;
    ep.dinv = L.d_dinv;
    NK_TRY(nk_csr_spmv_dev(L.A, *dcur, nullptr, d_skip, nullptr, &ep));   // r −= A d; d' = c1 d + c2 D⁻¹ r; x += d'
    std::swap(*dcur, *dnext);
    rho = rho_new;
  }
  return NK_OK;
}
int nk_amg_apply_dev(nk_amg *M, const double *d_b, double *d_x, const int *d_skip) {
  nk_ctx *ctx = M->ctx;
  if (!M->ready) NK_FAIL(NK_E_SINGULAR, "AMG: the hierarchy has no valid numbers (its last update failed)");
  // the smoothers' and residuals' SpMVs on the levels are the preconditioner's own work, not applications of the Jacobian
  // operator: the statistics' counter is put back behind the cycle (so runs under different preconditioners stay comparable)
  struct op_guard { nk_ctx *c; int64_t v; ~op_guard() { c->stats.op_applies = v; } } og{ctx, ctx->stats.op_applies};
  const int nlev = (int)M->lv.size();
  if (M->lv[0].n == 0) return NK_OK;
  M->lv[0].b = const_cast<double *>(d_b);
  M->lv[0].x = d_x;
  const int nu = M->prm.nu;
  auto grid = [](int64_t n) { return dim3(nk_grid_for(n, NK_BLOCK * 4, 4096)); };
  std::vector<double *> dcur(nlev), dnext(nlev);
  auto coarsest = [&](amg_level &L) -> int {
    if (M->dense) {
      NK_LAUNCH(ctx, k_amg_dense_apply, dim3(1), dim3(128), (int)L.n, M->ldinv, (const double *)M->d_inv, (const double *)L.b, L.x, d_skip);
      return NK_OK;
    }
    const double lmin = L.lmax / M->prm.cheb_ratio, theta = 0.5 * (L.lmax + lmin);
    double *dc = L.d0, *dn = L.d1;
    NK_LAUNCH(ctx, k_amg_cheb_first, grid(L.n), dim3(NK_BLOCK), L.n, 1, (const double *)L.b, (const double *)L.d_dinv, 1.0 / theta, L.r,
              dc, L.x, d_skip);
    return amg_cheb_rest(M, L, 4 * nu, &dc, &dn, d_skip);
  };
  for (int l = 0; l + 1 < nlev; ++l) {   // ---- down: pre-smooth from zero, residual, restrict
    amg_level &L = M->lv[l];
    const double lmin = L.lmax / M->prm.cheb_ratio, theta = 0.5 * (L.lmax + lmin);
    dcur[l] = L.d0; dnext[l] = L.d1;
    nk_prof_scope prof_(ctx, NK_K_OTHER, 40.0 * (double)L.n);
    NK_LAUNCH(ctx, k_amg_cheb_first, grid(L.n), dim3(NK_BLOCK), L.n, 1, (const double *)L.b, (const double *)L.d_dinv, 1.0 / theta, L.r,
              dcur[l], L.x, d_skip);
    NK_TRY(amg_cheb_rest(M, L, nu, &dcur[l], &dnext[l], d_skip));
    nk_spmv_epi ep;
    ep.mode = 4;
    ep.r = L.r;
    NK_TRY(nk_csr_spmv_dev(L.A, dcur[l], nullptr, d_skip, nullptr, &ep));                  // r −= A d: now r = b − A x
    NK_LAUNCH(ctx, k_amg_restrict, grid(L.nc), dim3(NK_BLOCK), L.nc, (const int32_t *)L.d_aggptr, (const int32_t *)L.d_aggmem,
              (const double *)L.r, M->lv[l + 1].b, d_skip);
  }
  NK_TRY(coarsest(M->lv.back()));
  for (int l = nlev - 2; l >= 0; --l) {   // ---- up: over-corrected prolongation, post-smooth
    amg_level &L = M->lv[l];
    const double lmin = L.lmax / M->prm.cheb_ratio, theta = 0.5 * (L.lmax + lmin);
    nk_prof_scope prof_(ctx, NK_K_OTHER, 40.0 * (double)L.n);
    NK_LAUNCH(ctx, k_amg_prolong, grid(L.n), dim3(NK_BLOCK), L.n, (const int32_t *)L.d_agg, (const double *)M->lv[l + 1].x,
              M->prm.overcorrection, L.x, d_skip);
    nk_spmv_epi ep;
    ep.mode = 2;
    ep.r = L.b;
    NK_TRY(nk_csr_spmv_dev(L.A, L.x, L.r, d_skip, nullptr, &ep));                          // r = b − A x
    NK_LAUNCH(ctx, k_amg_cheb_first, grid(L.n), dim3(NK_BLOCK), L.n, 0, (const double *)L.b, (const double *)L.d_dinv, 1.0 / theta, L.r,
              dcur[l], L.x, d_skip);
    NK_TRY(amg_cheb_rest(M, L, nu, &dcur[l], &dnext[l], d_skip));
  }
  NK_HIP(hipGetLastError());
  return NK_OK;
}
int nk_amg_levels(const nk_amg *M) { return (int)M->lv.size(); }
int nk_amg_level_info(const nk_amg *M, int l, int64_t *n, int64_t *nnz, double *lmax) {
  if (l < 0 || l >= (int)M->lv.size()) return NK_E_INVALID;
  if (n) *n = M->lv[l].n;
  if (nnz) *nnz = M->lv[l].nnz;
  if (lmax) *lmax = M->lv[l].lmax;
  return NK_OK;
}
const int32_t *nk_amg_aggregates(const nk_amg *Mc, int l) {
  nk_amg *M = const_cast<nk_amg *>(Mc);
  if (l < 0 || l + 1 >= (int)M->lv.size()) return nullptr;
  amg_level &L = M->lv[l];
  if (L.h_agg.empty() && L.n > 0 && L.d_agg) {   // (device set-up: the map has not been to the host yet)
    L.h_agg.resize((size_t)L.n);
    if (nk_memcpy(M->ctx, L.h_agg.data(), L.d_agg, (size_t)L.n * sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess) return nullptr;
  }
  return L.h_agg.data();
}
int nk_amg_matching(const nk_amg *M) { return M->matching; }
