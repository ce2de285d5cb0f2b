// Residual / JVP / VJP / Jacobian-value kernels of the built-in problems and the nk_problem API (seam 2).
//
//   QUADRATIC      f = u.*u .- p                         common/common_rootfind_testing.jl:15-17
//   BRATU2D        F_k = s[(4u_k − Σ nb)/h² − λ e^{u_k}]  SURVEY.md §8d (not in the reference)
//   BRUSSELATOR2D  brusselator_2d_loop                   lib/NonlinearSolveFirstOrder/test/sparsity_tests__item1.jl:13-36
//   USER           device callbacks (f!, jvp!, vjp!, jac!) = NonlinearFunction{true} fields
//
// Grid problems are partitioned by whole grid lines; a rank owns lines [j0, j1). Off-rank lines arrive in
// the halo buffer (lower neighbour's line(s) first, then the upper neighbour's).
// Algorithmic bytes: residual 16 n; matrix-free JVP 24 n (v, d = c_exp·e^u precomputed once per Newton
// step, Jv); Jacobian value fill ≈ 8 nnz + 8 n.
#include <math.h>
#include <string.h>

#include <algorithm>

#include "nk_internal.h"

#define SKIP_GUARD(d_skip) \
  if ((d_skip) != nullptr && *(d_skip) != 0) return;

// ============================================================================ quadratic
__global__ __launch_bounds__(NK_BLOCK) void k_quad_residual(int64_t n, double p, const double *__restrict__ u,
                                                            double *__restrict__ f) {
  const int64_t i = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x;
  if (i < n) f[i] = u[i] * u[i] - p;
}
__global__ __launch_bounds__(NK_BLOCK) void k_quad_jvp(int64_t n, const double *__restrict__ u,
                                                       const double *__restrict__ v, double *__restrict__ jv,
                                                       const int *d_skip, const double *__restrict__ out_scale) {
  SKIP_GUARD(d_skip);
  const double os = out_scale ? *out_scale : 1.0;
  const int64_t i = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x;
  if (i < n) jv[i] = os * (2.0 * u[i] * v[i]);
}
__global__ __launch_bounds__(NK_BLOCK) void k_quad_jac(int64_t n, const double *__restrict__ u,
                                                       double *__restrict__ vals) {
  const int64_t i = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x;
  if (i < n) vals[i] = 2.0 * u[i];
}

// ============================================================================ Bratu 2-D
// local point k = jl*ns + i, jl = j - j0. lo/hi: halo lines (nullptr at the physical boundary).
// Branch-free on purpose (see k_spmv_stream): every neighbour load is unconditional on a clamped address and the
// boundary is applied as a 0/1 weight, so the five loads of a point are in flight together.
__device__ __forceinline__ double bratu_lap(const double *__restrict__ u, const double *__restrict__ lo,
                                            const double *__restrict__ hi, int64_t ns, int64_t nl, int64_t i,
                                            int64_t jl, int64_t k) {
  const double *pw = u + (i > 0 ? k - 1 : k);
  const double *pe = u + (i < ns - 1 ? k + 1 : k);
  const double *ps = (jl > 0) ? (u + k - ns) : (lo ? lo + i : u + k);
  const double *pn = (jl < nl - 1) ? (u + k + ns) : (hi ? hi + i : u + k);
  const double c = u[k], w = *pw, e = *pe, s = *ps, n = *pn;
  const double mw = (i > 0) ? 1.0 : 0.0, me = (i < ns - 1) ? 1.0 : 0.0;
  const double ms = (jl > 0 || lo) ? 1.0 : 0.0, mn = (jl < nl - 1 || hi) ? 1.0 : 0.0;
  return 4.0 * c - mw * w - me * e - ms * s - mn * n;
}
__global__ __launch_bounds__(NK_BLOCK) void k_bratu_residual(int64_t ns, int64_t nl, double c_lap, double c_exp,
                                                             const double *__restrict__ u,
                                                             const double *__restrict__ lo,
                                                             const double *__restrict__ hi, double *__restrict__ f) {
  const int64_t k = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x;
  if (k >= ns * nl) return;
  const int64_t jl = k / ns, i = k - jl * ns;
  f[k] = c_lap * bratu_lap(u, lo, hi, ns, nl, i, jl, k) - c_exp * exp(u[k]);
}
// The residual and, in the same pass, the per-workgroup partial results of its norms — max |f| in partials[0 .. grid), Σ f² in
// partials[grid .. 2 grid) — with k_absmax_sumsq's assignment of entries to threads and its order of combination: the bits of
// ‖f‖∞ and ‖f‖₂ are those of the two-launch form (nk_blas.hip). One pass over f less per Newton step.
__global__ __launch_bounds__(NK_BLOCK) void k_bratu_residual_norms(int64_t ns, int64_t nl, double c_lap, double c_exp,
                                                                   const double *__restrict__ u, const double *__restrict__ lo,
                                                                   const double *__restrict__ hi, double *__restrict__ f,
                                                                   double *__restrict__ partials, double *__restrict__ f_copy,
                                                                   double *__restrict__ ss_copy, double *__restrict__ gpart) {
  // f_copy / ss_copy (nullable): f once more — into column 0 of the Krylov basis the next linear solve starts from — and the
  // Σ f² partials once more, where that solve's first kernel finds ‖b‖² (nk_gmres_preloaded_rhs)
  // gpart (nullable; one rank: jl is the global grid line): the Gershgorin partials of J(u) — centre 4 c_lap − c_exp·eᵘ, radius =
  // the |off-diagonals| added in CSR order —, exactly k_bratu_jac's expressions: {max −(d − r), max (d + r)} per workgroup
  __shared__ double sm[16];
  double m = 0.0, s = 0.0, mlo = -INFINITY, mhi = -INFINITY;
  const int64_t n = ns * nl, stride = (int64_t)gridDim.x * NK_BLOCK;
  // Four points per trip, ALL their loads requested before the first is used (round 6): the rolled loop made two dependent memory
  // round trips per point — u[k] for the exponential, then the neighbours — four points in a row per thread at 1024²: 8 round
  // trips ≈ the launch's 11 µs. A thread's points are visited in the same (ascending) order: the partial results keep their bits.
  constexpr int U = 4;
  const bool small = n < ((int64_t)1 << 31);   // (32-bit division: a 64-bit one is ≈ 40 instructions per point)
  for (int64_t k0 = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x; k0 < n; k0 += U * stride) {
    double c[U], w[U], e[U], so[U], no[U];
    int64_t kk[U], ii[U], jj[U];
#pragma unroll
    for (int q = 0; q < U; ++q) {
      const int64_t kq = k0 + q * stride, k = kq < n ? kq : n - 1;   // (past the end: the last point again, not stored)
      const int64_t jl = small ? (int64_t)((uint32_t)k / (uint32_t)ns) : k / ns, i = k - jl * ns;
      kk[q] = k; ii[q] = i; jj[q] = jl;
      const double *pw = u + (i > 0 ? k - 1 : k);
      const double *pe = u + (i < ns - 1 ? k + 1 : k);
      const double *ps = (jl > 0) ? (u + k - ns) : (lo ? lo + i : u + k);
      const double *pn = (jl < nl - 1) ? (u + k + ns) : (hi ? hi + i : u + k);
      c[q] = u[k]; w[q] = *pw; e[q] = *pe; so[q] = *ps; no[q] = *pn;
    }
#pragma unroll
    for (int q = 0; q < U; ++q) {
      const int64_t k = kk[q], i = ii[q], jl = jj[q];
      const double mw = (i > 0) ? 1.0 : 0.0, me = (i < ns - 1) ? 1.0 : 0.0;
      const double ms = (jl > 0 || lo) ? 1.0 : 0.0, mn = (jl < nl - 1 || hi) ? 1.0 : 0.0;
      const double lap = 4.0 * c[q] - mw * w[q] - me * e[q] - ms * so[q] - mn * no[q];   // (bratu_lap's expression)
      const double eu = exp(c[q]);
      const double v = c_lap * lap - c_exp * eu;
      if (k0 + q * stride < n) {
        if (gpart != nullptr) {
          const double d = 4.0 * c_lap - c_exp * eu;
          double rad = 0.0;
          if (jl > 0) rad += c_lap;
          if (i > 0) rad += c_lap;
          if (i < ns - 1) rad += c_lap;
          if (jl < ns - 1) rad += c_lap;
          mlo = fmax(mlo, -(d - rad));
          mhi = fmax(mhi, d + rad);
        }
        f[k] = v;
        if (f_copy) f_copy[k] = v;
        const double av = fabs(v);
        m = (m != m || av != av) ? __builtin_nan("") : (m > av ? m : av);
        s += v * v;
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const double mo = __shfl_xor(m, o, 64);
    m = (m != m || mo != mo) ? __builtin_nan("") : (m > mo ? m : mo);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) { sm[threadIdx.x >> 6] = m; sm[4 + (threadIdx.x >> 6)] = s; }
  __syncthreads();
  if (threadIdx.x == 0) {
    auto nmax = [](double a, double b) { return (a != a || b != b) ? __builtin_nan("") : (a > b ? a : b); };
    partials[blockIdx.x] = nmax(nmax(sm[0], sm[1]), nmax(sm[2], sm[3]));
    partials[gridDim.x + blockIdx.x] = (sm[4] + sm[5]) + (sm[6] + sm[7]);
    if (ss_copy) ss_copy[blockIdx.x] = (sm[4] + sm[5]) + (sm[6] + sm[7]);
  }
  if (gpart != nullptr) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mlo = fmax(mlo, __shfl_xor(mlo, o, 64)); mhi = fmax(mhi, __shfl_xor(mhi, o, 64)); }
    if ((threadIdx.x & 63) == 0) { sm[8 + (threadIdx.x >> 6)] = mlo; sm[12 + (threadIdx.x >> 6)] = mhi; }
    __syncthreads();
    if (threadIdx.x == 0) {
      gpart[blockIdx.x] = fmax(fmax(sm[8], sm[9]), fmax(sm[10], sm[11]));
      gpart[gridDim.x + blockIdx.x] = fmax(fmax(sm[12], sm[13]), fmax(sm[14], sm[15]));
    }
  }
}
__global__ __launch_bounds__(NK_BLOCK) void k_bratu_diag(int64_t n, double c_exp, const double *__restrict__ u,
                                                         double *__restrict__ d) {
  const int64_t k = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x;
  if (k < n) d[k] = c_exp * exp(u[k]);
}
__global__ __launch_bounds__(NK_BLOCK) void k_bratu_jvp(int64_t ns, int64_t nl, double c_lap,
                                                        const double *__restrict__ d, const double *__restrict__ v,
                                                        const double *__restrict__ lo, const double *__restrict__ hi,
                                                        double *__restrict__ jv, const int *d_skip,
                                                        const double *__restrict__ out_scale, const nk_spmv_epi epi) {
  SKIP_GUARD(d_skip);
  const double os = out_scale ? *out_scale : 1.0;
  const int64_t k = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x;
  if (k >= ns * nl) return;
  const int64_t jl = k / ns, i = k - jl * ns;
  const double vk = v[k];
  const double res = c_lap * bratu_lap(v, lo, hi, ns, nl, i, jl, k) - d[k] * vk;
  if (epi.mode == 0) {
    jv[k] = os * res;
  } else if (epi.mode == 3) {  // Newton-basis step of the s-step Arnoldi process: scale·(J v − θ v)
    jv[k] = os * (res - (*epi.theta) * vk);
  } else if (epi.mode == 2) {  // fused residual: out = b − J v (b = epi.r)
    jv[k] = epi.r[k] - res;
  } else {  // fused Chebyshev step (v = d_old): r −= J d; d_new = c1 d_old + c2 r; y += d_new
    const double rr = epi.r[k] - res;
    epi.r[k] = rr;
    const double dn = epi.c1 * vk + epi.c2 * rr;
    epi.dnew[k] = dn;
    epi.yacc[k] += dn;
  }
}
// Tiled form of k_bratu_jvp: a thread owns column i of TY consecutive grid lines, so the TY+2 values of its column are
// loaded once (6 instead of 12 vertical loads for TY = 4), the index arithmetic is 32-bit with no division (blockIdx.y is
// the line tile), and every load of the tile is issued before the first use. Same arithmetic, term by term, as bratu_lap.
template <int TY>
__global__ __launch_bounds__(NK_BLOCK) void k_bratu_jvp_tile(int ns, int nl, double c_lap, const double *__restrict__ d,
                                                             const double *__restrict__ v,
                                                             const double *__restrict__ lo,
                                                             const double *__restrict__ hi, double *__restrict__ jv,
                                                             const int *d_skip, const double *__restrict__ out_scale,
                                                             const nk_spmv_epi epi) {
  const int skip = d_skip ? *d_skip : 0;
  const double os = out_scale ? *out_scale : 1.0;
  const int i = blockIdx.x * NK_BLOCK + threadIdx.x;
  const int j0 = blockIdx.y * TY;
  if (i >= ns) return;
  const int iw = i > 0 ? i - 1 : i, ie = i < ns - 1 ? i + 1 : i;
  const double mw = (i > 0) ? 1.0 : 0.0, me = (i < ns - 1) ? 1.0 : 0.0;
  double col[TY + 2], w[TY], e[TY], dg[TY], rr[TY];
#pragma unroll
  for (int q = 0; q < TY + 2; ++q) {
    int jl = j0 - 1 + q;
    jl = jl < 0 ? 0 : (jl > nl - 1 ? nl - 1 : jl);
    col[q] = v[(size_t)jl * ns + i];
  }
#pragma unroll
  for (int r = 0; r < TY; ++r) {
    const int jl = (j0 + r < nl) ? j0 + r : nl - 1;
    const size_t row = (size_t)jl * ns;
    w[r] = v[row + iw];
    e[r] = v[row + ie];
    dg[r] = d[row + i];
    rr[r] = (epi.mode == 1 || epi.mode == 2) ? epi.r[row + i] : 0.0;
  }
  if (j0 == 0 && lo) col[0] = lo[i];                                  // uniform per workgroup
  const double hiv = (j0 + TY >= nl && hi) ? hi[i] : 0.0;             // line nl: the upper neighbour rank's first line
  if (skip) return;
#pragma unroll
  for (int r = 0; r < TY; ++r) {
    const int jl = j0 + r;
    if (jl < nl) {
      const double ms = (jl > 0 || lo) ? 1.0 : 0.0, mn = (jl < nl - 1 || hi) ? 1.0 : 0.0;
      const double c = col[r + 1];
      const double nn = (jl == nl - 1) ? hiv : col[r + 2];  // (weight mn = 0 without an upper neighbour)
      const double res = c_lap * (4.0 * c - mw * w[r] - me * e[r] - ms * col[r] - mn * nn) - dg[r] * c;
      const size_t k = (size_t)jl * ns + i;
      if (epi.mode == 0) {
        jv[k] = os * res;
      } else if (epi.mode == 3) {  // Newton-basis step of the s-step Arnoldi process: scale·(J v − θ v)
        jv[k] = os * (res - (*epi.theta) * c);
      } else if (epi.mode == 2) {  // fused residual: out = b − J v (b = epi.r)
        jv[k] = rr[r] - res;
      } else {  // fused Chebyshev step (v = d_old): r −= J d; d_new = c1 d_old + c2 r; y += d_new
        const double rn = rr[r] - res;
        epi.r[k] = rn;
        const double dn = epi.c1 * c + epi.c2 * rn;
        epi.dnew[k] = dn;
        epi.yacc[k] += dn;
      }
    }
  }
}
// values in pattern order [S?][W?][C][E?][N?] (columns ascending); j = global grid line. A workgroup owns 256 consecutive
// rows = one contiguous run of ≤ 1280 non-zeros: every thread lays its row's ≤ 5 values out in LDS, then the run goes to
// memory with lane-contiguous stores (a thread writing its own row straight to memory stores 8 bytes every 40: the first
// version of this kernel ran at 2.1 TB/s, 24 µs at 1024²).
__global__ __launch_bounds__(NK_BLOCK) void k_bratu_jac(int64_t ns, int64_t nl, int64_t j0, double c_lap,
                                                        double c_exp, const double *__restrict__ u,
                                                        const int32_t *__restrict__ rowptr,
                                                        double *__restrict__ vals, double *__restrict__ gpart,
                                                        const nk_fold_norms fold, const nk_ss_begin_args beg) {
  __shared__ double sv[5 * NK_BLOCK];
  __shared__ double sg[8];
  __shared__ int32_t s_p0, s_p1;
  // fold.partials != nullptr: workgroup 0 fills nothing — it reduces a residual kernel's norm partials and hands them to the host
  // (nk_fold_norms); the rows then belong to workgroups 1 … (the grid is one larger)
  const int fb = fold.partials != nullptr ? 1 : 0;
  if (fb && blockIdx.x == 0) {
    nk_reduce_inf2_body(fold, sv);
    if (beg.ctl != nullptr) {   // … and the next linear solve's cycle begin (nk_gmres_begin_ahead)
      __syncthreads();
      nk_ss_begin_body(beg, sv + 16);
    }
    return;
  }
  const int bid = (int)blockIdx.x - fb, nbl = (int)gridDim.x - fb;
  const int64_t n = ns * nl, r0 = (int64_t)bid * NK_BLOCK;
  const int64_t k = r0 + threadIdx.x;
  const int64_t rlast = (r0 + NK_BLOCK < n ? r0 + NK_BLOCK : n);
  if (threadIdx.x == 0) { s_p0 = rowptr[r0]; s_p1 = rowptr[rlast]; }
  int32_t p = 0;
  double d = 0.0;
  int64_t i = 0, j = 0;
  if (k < n) {
    const int64_t jl = k / ns;
    i = k - jl * ns;
    j = j0 + jl;
    p = rowptr[k];
    d = 4.0 * c_lap - c_exp * exp(u[k]);
  }
  __syncthreads();
  const int32_t p0 = s_p0, nnzb = s_p1 - s_p0;
  if (k < n) {
    int q = p - p0;
    if (j > 0) sv[q++] = -c_lap;
    if (i > 0) sv[q++] = -c_lap;
    sv[q++] = d;
    if (i < ns - 1) sv[q++] = -c_lap;
    if (j < ns - 1) sv[q++] = -c_lap;
  }
  __syncthreads();
  for (int q = threadIdx.x; q < nnzb; q += NK_BLOCK) vals[p0 + q] = sv[q];
  if (gpart != nullptr) {
    // the row's Gershgorin disc on the fly — centre d, radius = the |off-diagonals| added in CSR order (S, W, E, N), exactly as
    // k_csr_gershgorin's pass over the finished matrix would (bitwise the same bounds, 19 µs and 67 MB less per Jacobian)
    double mlo = -INFINITY, mhi = -INFINITY;
    if (k < n) {
      double rad = 0.0;
      if (j > 0) rad += c_lap;
      if (i > 0) rad += c_lap;
      if (i < ns - 1) rad += c_lap;
      if (j < ns - 1) rad += c_lap;
      mlo = -(d - rad);
      mhi = d + rad;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mlo = fmax(mlo, __shfl_xor(mlo, o, 64)); mhi = fmax(mhi, __shfl_xor(mhi, o, 64)); }
    if ((threadIdx.x & 63) == 0) { sg[threadIdx.x >> 6] = mlo; sg[4 + (threadIdx.x >> 6)] = mhi; }
    __syncthreads();
    if (threadIdx.x == 0) {
      gpart[bid] = fmax(fmax(sg[0], sg[1]), fmax(sg[2], sg[3]));
      gpart[nbl + bid] = fmax(fmax(sg[4], sg[5]), fmax(sg[6], sg[7]));
    }
  }
}

// ============================================================================ Brusselator 2-D
// local layout (i, jl, species): idx = i + N*jl + N*nl*s. lo/hi: [species0 line | species1 line] of the
// periodic neighbours; nullptr → wrap inside the local slab (single rank).
struct brus_par { int64_t N, nl, j0; double A, B, alpha; };

__device__ __forceinline__ double brus_lap(const double *__restrict__ w, const double *__restrict__ lo,
                                           const double *__restrict__ hi, const brus_par &q, int64_t i, int64_t jl) {
  const int64_t N = q.N, nl = q.nl;
  const int64_t ip1 = (i + 1 == N) ? 0 : i + 1, im1 = (i == 0) ? N - 1 : i - 1;
  double s = w[im1 + N * jl] + w[ip1 + N * jl] - 4.0 * w[i + N * jl];
  if (jl + 1 < nl) s += w[i + N * (jl + 1)];
  else s += hi ? hi[i] : w[i];                      // wrap to local line 0
  if (jl > 0) s += w[i + N * (jl - 1)];
  else s += lo ? lo[i] : w[i + N * (nl - 1)];       // wrap to local last line
  return s;
}
__global__ __launch_bounds__(NK_BLOCK) void k_brus_residual(brus_par q, const double *__restrict__ U,
                                                            const double *__restrict__ lo,
                                                            const double *__restrict__ hi, double *__restrict__ F) {
  const int64_t t = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x;
  const int64_t nn = q.N * q.nl;
  if (t >= nn) return;
  const int64_t jl = t / q.N, i = t - jl * q.N;
  const double x = (double)i / (double)(q.N - 1), y = (double)(q.j0 + jl) / (double)(q.N - 1);
  const double bf = (((x - 0.3) * (x - 0.3) + (y - 0.6) * (y - 0.6)) <= 0.1 * 0.1) ? 5.0 : 0.0;
  const double uu = U[t], vv = U[nn + t];
  const double *lo1 = lo ? lo + q.N : nullptr, *hi1 = hi ? hi + q.N : nullptr;
  F[t] = q.alpha * brus_lap(U, lo, hi, q, i, jl) + q.B + uu * uu * vv - (q.A + 1.0) * uu + bf;
  F[nn + t] = q.alpha * brus_lap(U + nn, lo1, hi1, q, i, jl) + q.A * uu - uu * uu * vv;
}
// transpose=0: J*v ; transpose=1: Jᵀ*v
__global__ __launch_bounds__(NK_BLOCK) void k_brus_jvp(brus_par q, int transpose, const double *__restrict__ U,
                                                       const double *__restrict__ V, const double *__restrict__ lo,
                                                       const double *__restrict__ hi, double *__restrict__ JV,
                                                       const int *d_skip, const double *__restrict__ out_scale) {
  SKIP_GUARD(d_skip);
  const double os = out_scale ? *out_scale : 1.0;
  const int64_t t = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x;
  const int64_t nn = q.N * q.nl;
  if (t >= nn) return;
  const int64_t jl = t / q.N, i = t - jl * q.N;
  const double uu = U[t], vv = U[nn + t], a = V[t], b = V[nn + t];
  const double *lo1 = lo ? lo + q.N : nullptr, *hi1 = hi ? hi + q.N : nullptr;
  const double la = q.alpha * brus_lap(V, lo, hi, q, i, jl), lb = q.alpha * brus_lap(V + nn, lo1, hi1, q, i, jl);
  const double d11 = 2.0 * uu * vv - (q.A + 1.0), d12 = uu * uu, d21 = q.A - 2.0 * uu * vv, d22 = -uu * uu;
  if (!transpose) {
    JV[t] = os * (la + d11 * a + d12 * b);
    JV[nn + t] = os * (lb + d21 * a + d22 * b);
  } else {
    JV[t] = os * (la + d11 * a + d21 * b);
    JV[nn + t] = os * (lb + d12 * a + d22 * b);
  }
}
// role per non-zero: 0 α (neighbour) · 1 ∂F1/∂u · 2 ∂F1/∂v · 3 ∂F2/∂v · 4 ∂F2/∂u
__global__ __launch_bounds__(NK_BLOCK) void k_brus_jac(int64_t nnz, int64_t nn, double A, double alpha,
                                                       const uint8_t *__restrict__ role,
                                                       const int32_t *__restrict__ node, const double *__restrict__ U,
                                                       double *__restrict__ vals) {
  const int64_t p = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x;
  if (p >= nnz) return;
  const int32_t t = node[p];
  const double uu = U[t], vv = U[nn + t];
  double r;
  switch (role[p]) {
    case 0: r = alpha; break;
    case 1: r = -4.0 * alpha + 2.0 * uu * vv - (A + 1.0); break;
    case 2: r = uu * uu; break;
    case 3: r = -4.0 * alpha - uu * uu; break;
    default: r = A - 2.0 * uu * vv; break;
  }
  vals[p] = r;
}
__global__ __launch_bounds__(NK_BLOCK) void k_brus_u0(brus_par q, double *__restrict__ U) {
  const int64_t t = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x;
  const int64_t nn = q.N * q.nl;
  if (t >= nn) return;
  const int64_t jl = t / q.N, i = t - jl * q.N;
  const double x = (double)i / (double)(q.N - 1), y = (double)(q.j0 + jl) / (double)(q.N - 1);
  const double a = y * (1.0 - y), b = x * (1.0 - x);
  U[t] = 22.0 * a * sqrt(a);        // 22 (y(1-y))^(3/2)   sparsity_tests__item1.jl:40-50
  U[nn + t] = 27.0 * b * sqrt(b);
}

// ============================================================================ host side
static inline int grid1(int64_t n) { return (int)((n + NK_BLOCK - 1) / NK_BLOCK); }

static brus_par brus_params(const nk_problem *P) {
  brus_par q;
  q.N = P->ns;
  q.nl = P->j1 - P->j0;
  q.j0 = P->j0;
  q.A = P->params[1];
  q.B = P->params[2];
  const double dx = P->params[4];
  q.alpha = P->params[3] / (dx * dx);
  return q;
}

static int setup_grid_partition(nk_problem *P, int64_t ns, int dof_per_node, bool periodic) {
  nk_ctx *ctx = P->ctx;
  const int R = ctx->nranks, r = ctx->rank;
  NK_REQUIRE(ns >= R, "grid side %lld smaller than the number of ranks %d", (long long)ns, R);
  int64_t b, e;
  NK_TRY(nk_partition_range(ns, 1, R, r, &b, &e));
  P->ns = ns;
  P->j0 = b;
  P->j1 = e;
  const int64_t nl = e - b;
  P->n_local = ns * nl * dof_per_node;
  P->n_global = ns * ns * dof_per_node;
  P->row_begin = ns * b * dof_per_node;
  if (R == 1) return NK_OK;
  // who needs what from me: my first line goes to the rank below (it is their "upper" line), my last line
  // to the rank above. Receive order per peer: [their-lower-request][their-upper-request].
  std::vector<std::vector<int32_t>> send(R);
  std::vector<int64_t> recv(R, 0);
  const int below = (r == 0) ? (periodic ? R - 1 : -1) : r - 1;
  const int above = (r == R - 1) ? (periodic ? 0 : -1) : r + 1;
  auto push_line = [&](std::vector<int32_t> &v, int64_t jl) {
    for (int s = 0; s < dof_per_node; ++s)
      for (int64_t i = 0; i < ns; ++i) v.push_back((int32_t)(i + ns * jl + ns * nl * s));
  };
  // A peer p receives from me first what serves as p's LOWER halo (my last line, if I am below p), then
  // what serves as p's UPPER halo (my first line, if I am above p).
  if (above >= 0) push_line(send[above], nl - 1);  // I am below `above`
  if (below >= 0) push_line(send[below], 0);       // I am above `below`
  if (below >= 0) recv[below] += ns * dof_per_node;
  if (above >= 0) recv[above] += ns * dof_per_node;
  return nk_halo_setup(ctx, &P->halo, send, recv);
}

// pointers to my lower / upper halo lines inside the receive buffer (nullptr if none)
static void halo_lines(const nk_problem *P, int dof_per_node, bool periodic, const double **lo, const double **hi) {
  *lo = *hi = nullptr;
  const nk_ctx *ctx = P->ctx;
  const int R = ctx->nranks, r = ctx->rank;
  if (R == 1 || P->replicated) return;
  const int below = (r == 0) ? (periodic ? R - 1 : -1) : r - 1;
  const int above = (r == R - 1) ? (periodic ? 0 : -1) : r + 1;
  const int64_t line = P->ns * dof_per_node;
  if (below >= 0) *lo = P->halo.d_recv + P->halo.recv_off[below];
  if (above >= 0) {
    // if the same peer is both below and above (R == 2, periodic) its block is [lower part][upper part]
    const int64_t shift = (above == below) ? line : 0;
    *hi = P->halo.d_recv + P->halo.recv_off[above] + shift;
  }
}

extern "C" int nk_problem_create(nk_ctx *ctx, int kind, const double *params, int nparams, nk_problem **out) {
  NK_REQUIRE(ctx && params && out, "NULL argument");
  NK_REQUIRE(nparams >= 0 && nparams <= 8, "nparams out of range");
  NK_HIP(hipSetDevice(ctx->device));
  nk_problem *P = new nk_problem();
  P->ctx = ctx;
  P->kind = kind;
  P->nparams = nparams;
  for (int i = 0; i < nparams; ++i) P->params[i] = params[i];
  int st = NK_OK;
  if (kind == NK_PROBLEM_QUADRATIC) {
    if (nparams < 2) { delete P; NK_FAIL(NK_E_INVALID, "QUADRATIC needs {n, p}"); }
    const int64_t n = (int64_t)params[0];
    int64_t b, e;
    st = nk_partition_range(n, 1, ctx->nranks, ctx->rank, &b, &e);
    P->n_global = n;
    P->row_begin = b;
    P->n_local = e - b;
  } else if (kind == NK_PROBLEM_BRATU2D) {
    if (nparams < 2) { delete P; NK_FAIL(NK_E_INVALID, "BRATU2D needs {n_side, lambda[, scale]}"); }
    const int64_t ns = (int64_t)params[0];
    const double lam = params[1], sc = nparams >= 3 ? params[2] : 0.0;
    const double h = 1.0 / (double)(ns + 1);
    const double s = (sc == 0.0) ? h * h : sc;
    P->c_lap = s / (h * h);
    P->c_exp = s * lam;
    st = setup_grid_partition(P, ns, 1, false);
  } else if (kind == NK_PROBLEM_BRUSSELATOR2D) {
    if (nparams < 5) { delete P; NK_FAIL(NK_E_INVALID, "BRUSSELATOR2D needs {N, A, B, alpha, dx}"); }
    const int64_t N = (int64_t)params[0];
    if (N < 3) { delete P; NK_FAIL(NK_E_INVALID, "BRUSSELATOR2D needs N >= 3"); }
    st = setup_grid_partition(P, N, 2, true);
  } else {
    delete P;
    NK_FAIL(NK_E_INVALID, "unknown built-in problem kind %d", kind);
  }
  if (st != NK_OK) { delete P; return st; }
  *out = P;
  return NK_OK;
}

// a Bratu problem every rank holds in full (the multigrid's coarsest level, solved redundantly): no partition, no halo
int nk_problem_create_bratu_replicated(nk_ctx *ctx, int64_t ns, double lambda, double scale, nk_problem **out) {
  nk_problem *P = new nk_problem();
  P->ctx = ctx;
  P->kind = NK_PROBLEM_BRATU2D;
  P->replicated = true;
  P->nparams = 3;
  P->params[0] = (double)ns;
  P->params[1] = lambda;
  P->params[2] = scale;
  const double h = 1.0 / (double)(ns + 1), s = (scale == 0.0) ? h * h : scale;
  P->c_lap = s / (h * h);
  P->c_exp = s * lambda;
  P->ns = ns;
  P->j0 = 0;
  P->j1 = ns;
  P->n_local = P->n_global = ns * ns;
  P->row_begin = 0;
  *out = P;
  return NK_OK;
}

// the same for the Brusselator (multigrid, coarsest level on several ranks)
int nk_problem_create_brus_replicated(nk_ctx *ctx, const double *params5, nk_problem **out) {
  nk_problem *P = new nk_problem();
  P->ctx = ctx;
  P->kind = NK_PROBLEM_BRUSSELATOR2D;
  P->replicated = true;
  P->nparams = 5;
  for (int i = 0; i < 5; ++i) P->params[i] = params5[i];
  const int64_t N = (int64_t)params5[0];
  P->ns = N;
  P->j0 = 0;
  P->j1 = N;
  P->n_local = P->n_global = 2 * N * N;
  P->row_begin = 0;
  *out = P;
  return NK_OK;
}
// ghost lines of a vector laid out like the problem's unknowns (grid problems on several ranks): exchanges the halo and
// returns the lines below / above the owned slab ([species 0 line | species 1 line] for the Brusselator); nullptr where the
// slab wraps onto itself (one rank, replicated problems)
int nk_problem_ghost_lines(nk_problem *P, const double *d_v, const double **lo, const double **hi) {
  NK_TRY(nk_halo_exchange(P->ctx, &P->halo, d_v));
  if (P->kind == NK_PROBLEM_BRUSSELATOR2D) halo_lines(P, 2, true, lo, hi);
  else halo_lines(P, 1, false, lo, hi);
  return NK_OK;
}

extern "C" int nk_problem_create_user(nk_ctx *ctx, int64_t n_local, int64_t n_global, int64_t row_begin,
                                      const nk_user_callbacks *cb, void *user, nk_csr *jac_pattern,
                                      nk_problem **out) {
  NK_REQUIRE(ctx && cb && cb->residual && out, "NULL argument / residual callback");
  nk_problem *P = new nk_problem();
  P->ctx = ctx;
  P->kind = NK_PROBLEM_USER;
  P->n_local = n_local;
  P->n_global = n_global;
  P->row_begin = row_begin;
  P->cb = *cb;
  P->user = user;
  P->user_pattern = jac_pattern;
  *out = P;
  return NK_OK;
}

extern "C" int nk_problem_destroy(nk_problem *P) {
  if (!P) return NK_OK;
  hipFree(P->d_diag);
  nk_powers_plan_destroy(P->pw);
  hipFree(P->d_fd_f0);
  hipFree(P->d_fd_up);
  hipFree(P->d_fd_f1);
  for (double *&t : P->d_tmp) hipFree(t);
  nk_halo_free(&P->halo);
  nk_csr_destroy(P->lin_J);
  delete P;
  return NK_OK;
}
extern "C" int nk_problem_size(nk_problem *P, int64_t *n_local, int64_t *n_global, int64_t *row_begin) {
  NK_REQUIRE(P, "NULL argument");
  if (n_local) *n_local = P->n_local;
  if (n_global) *n_global = P->n_global;
  if (row_begin) *row_begin = P->row_begin;
  return NK_OK;
}
extern "C" int nk_problem_set_params(nk_problem *P, const double *params, int nparams) {
  NK_REQUIRE(P && params, "NULL argument");
  NK_REQUIRE(nparams == P->nparams, "nparams mismatch (%d vs %d)", nparams, P->nparams);
  if (P->kind == NK_PROBLEM_QUADRATIC) {
    NK_REQUIRE((int64_t)params[0] == P->n_global, "cannot change the problem size");
  } else if (P->kind == NK_PROBLEM_BRATU2D || P->kind == NK_PROBLEM_BRUSSELATOR2D) {
    NK_REQUIRE((int64_t)params[0] == P->ns, "cannot change the grid size");
  }
  for (int i = 0; i < nparams; ++i) P->params[i] = params[i];
  P->params_version++;
  if (P->kind == NK_PROBLEM_BRATU2D) {
    const double h = 1.0 / (double)(P->ns + 1);
    const double sc = nparams >= 3 ? params[2] : 0.0, s = (sc == 0.0) ? h * h : sc;
    P->c_lap = s / (h * h);
    P->c_exp = s * params[1];
  }
  return NK_OK;
}

// ---------------------------------------------------------------------------- device-level operations
// f(u) and the stage-1 partials of ‖f‖∞, ‖f‖₂² in ONE launch, where the problem has such a kernel: *grid_out = the number of
// workgroups (partials laid out as k_absmax_sumsq leaves them), 0 = not available (nothing was launched).
int nk_problem_residual_norms_dev(nk_problem *P, const double *d_u, double *d_f, double *partials, int *grid_out, double *f_copy,
                                  double *ss_copy, double *gpart) {
  static const bool off = getenv("NK_FUSED_RESIDUAL_NORMS") && atoi(getenv("NK_FUSED_RESIDUAL_NORMS")) == 0;   // A/B switch
  *grid_out = 0;
  nk_ctx *ctx = P->ctx;
  const int64_t n = P->n_local;
  if (off || n == 0 || P->kind != NK_PROBLEM_BRATU2D) return NK_OK;
  const int grid = nk_grid_for(n, NK_BLOCK * 4, NK_MAX_RED_BLOCKS);   // (k_absmax_sumsq's grid: nk_blas_norms_inf2)
  nk_prof_scope prof_(ctx, NK_K_RESIDUAL, (f_copy ? 24.0 : 16.0) * (double)n);
  const double *lo, *hi;
  NK_TRY(nk_halo_exchange(ctx, &P->halo, d_u));
  halo_lines(P, 1, false, &lo, &hi);
  // (the discs' radii depend on the GLOBAL position of a row: one rank with the whole grid only)
  const bool whole = ctx->nranks == 1 && P->j0 == 0 && P->j1 == P->ns && !P->replicated;
  NK_LAUNCH(ctx, k_bratu_residual_norms, dim3(grid), dim3(NK_BLOCK), P->ns, P->j1 - P->j0, P->c_lap, P->c_exp, d_u, lo, hi, d_f,
            partials, f_copy, ss_copy, whole ? gpart : (double *)nullptr);
  NK_HIP(hipGetLastError());
  *grid_out = grid;
  return NK_OK;
}
int nk_problem_residual_dev(nk_problem *P, const double *d_u, double *d_f) {
  nk_ctx *ctx = P->ctx;
  const int64_t n = P->n_local;
  if (n == 0) return NK_OK;
  nk_prof_scope prof_(ctx, NK_K_RESIDUAL, 16.0 * (double)n);
  switch (P->kind) {
    case NK_PROBLEM_QUADRATIC:
      NK_LAUNCH(ctx, k_quad_residual, dim3(grid1(n)), dim3(NK_BLOCK), n, P->params[1], d_u, d_f);
      break;
    case NK_PROBLEM_BRATU2D: {
      const double *lo, *hi;
      NK_TRY(nk_halo_exchange(ctx, &P->halo, d_u));
      halo_lines(P, 1, false, &lo, &hi);
      NK_LAUNCH(ctx, k_bratu_residual, dim3(grid1(n)), dim3(NK_BLOCK), P->ns, P->j1 - P->j0,
                         P->c_lap, P->c_exp, d_u, lo, hi, d_f);
      break;
    }
    case NK_PROBLEM_BRUSSELATOR2D: {
      const double *lo, *hi;
      NK_TRY(nk_halo_exchange(ctx, &P->halo, d_u));
      halo_lines(P, 2, true, &lo, &hi);
      NK_LAUNCH(ctx, k_brus_residual, dim3(grid1(n / 2)), dim3(NK_BLOCK), brus_params(P), d_u, lo,
                         hi, d_f);
      break;
    }
    case NK_PROBLEM_USER:
      if (P->cb.residual(P->user, d_u, d_f, (void *)ctx->stream) != 0) NK_FAIL(NK_E_CALLBACK, "residual callback failed");
      break;
    default:
      NK_FAIL(NK_E_INVALID, "bad problem kind");
  }
  NK_HIP(hipGetLastError());
  return NK_OK;
}

// J(d_u) into the problem's private copy of the jac_prototype pattern: by f.jac when the user supplied it, otherwise by
// colour-compressed differences of the JVP (the AutoSparse(AutoFiniteDiff) analogue) — what prepare_vjp / prepare_jvp fall
// back to when there is no vjp / jvp (SciMLJacobianOperators.jl:307-322, 379-392; the reference's own cache is dense)
static int user_lin_J(nk_problem *P, const double *d_u) {
  NK_REQUIRE(P->user_pattern, "user problem has neither the operator callback nor a jac_prototype to build J from");
  if (!P->lin_J) NK_TRY(nk_csr_clone_pattern(P->user_pattern, &P->lin_J));
  if (P->d_u_linJ == d_u) return NK_OK;
  if (P->cb.jac_values) {
    if (P->cb.jac_values(P->user, d_u, P->lin_J->d_val, (void *)P->ctx->stream) != 0)
      NK_FAIL(NK_E_CALLBACK, "jac_values callback failed");
    P->lin_J->t_values_stale = true; P->lin_J->bounds_valid = false; P->lin_J->bounds_pending = false;
  } else {
    NK_TRY(nk_problem_jac_colored_dev(P, d_u, P->lin_J));
  }
  P->d_u_linJ = d_u;
  return NK_OK;
}

// Bounds of the Bratu Jacobian's spectrum from its Gershgorin discs (centre 4c − d_k, radius ≤ 4c, d = c_exp·exp(u)):
// [−max d, 8c − min d], left on the device as {−lo, hi} = {max d, 8c + max(−d)} for the s-step Newton basis (nk_sstep.hip)
__global__ __launch_bounds__(NK_BLOCK) void k_minmax_stage1(int64_t n, const double *__restrict__ x, double *__restrict__ part) {
  __shared__ double red[8];
  double a = -INFINITY, c = -INFINITY;
  const int64_t stride = (int64_t)gridDim.x * NK_BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x; i < n; i += stride) {
    const double v = x[i];
    a = fmax(a, v);
    c = fmax(c, -v);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { a = fmax(a, __shfl_xor(a, o, 64)); c = fmax(c, __shfl_xor(c, o, 64)); }
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = a; red[4 + (threadIdx.x >> 6)] = c; }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[blockIdx.x] = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    part[gridDim.x + blockIdx.x] = fmax(fmax(red[4], red[5]), fmax(red[6], red[7]));
  }
}
__global__ __launch_bounds__(256) void k_bratu_interval_final(int nblk, const double *__restrict__ part, double c8,
                                                              double *__restrict__ out2) {
  __shared__ double red[8];
  double a = -INFINITY, c = -INFINITY;
  for (int i = threadIdx.x; i < nblk; i += 256) { a = fmax(a, part[i]); c = fmax(c, part[nblk + i]); }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { a = fmax(a, __shfl_xor(a, o, 64)); c = fmax(c, __shfl_xor(c, o, 64)); }
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = a; red[4 + (threadIdx.x >> 6)] = c; }
  __syncthreads();
  if (threadIdx.x == 0) {
    out2[0] = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    out2[1] = c8 + fmax(fmax(red[4], red[5]), fmax(red[6], red[7]));
  }
}
int nk_problem_spectrum_interval_dev(nk_problem *P, const double *d_u, double *d_out2) {
  NK_REQUIRE(P->kind == NK_PROBLEM_BRATU2D && P->n_local > 0, "internal: spectrum bounds exist for the Bratu stencil only");
  nk_ctx *ctx = P->ctx;
  if (P->d_u_lin != d_u || !P->d_diag) NK_TRY(nk_problem_jvp_prepare(P, d_u));
  const int grid = nk_grid_for(P->n_local, NK_BLOCK * 4, 1024);
  NK_LAUNCH(ctx, k_minmax_stage1, dim3(grid), dim3(NK_BLOCK), P->n_local, (const double *)P->d_diag, ctx->d_partials);
  NK_LAUNCH(ctx, k_bratu_interval_final, dim3(1), dim3(256), grid, (const double *)ctx->d_partials, 8.0 * P->c_lap, d_out2);
  NK_HIP(hipGetLastError());
  return NK_OK;
}

int nk_problem_jvp_prepare(nk_problem *P, const double *d_u) {
  P->d_u_lin = d_u;
  if (P->kind == NK_PROBLEM_BRATU2D) {
    if (!P->d_diag) NK_TRY(nk_dev_alloc(&P->d_diag, (size_t)P->n_local + 1));
    if (P->n_local)
      NK_LAUNCH(P->ctx, k_bratu_diag, dim3(grid1(P->n_local)), dim3(NK_BLOCK), P->n_local,
                         P->c_exp, d_u, P->d_diag);
    NK_HIP(hipGetLastError());
  }
  if (P->kind == NK_PROBLEM_USER && !P->cb.jvp && P->cb.jac_values && P->user_pattern)
    return user_lin_J(P, d_u);  // prepare_jvp's second choice: f.jac, then J·v (SciMLJacobianOperators.jl:379-392)
  if (P->kind == NK_PROBLEM_USER && !P->cb.jvp && P->n_local) {  // forward differences need f at the linearisation point
    if (!P->d_fd_f0) {
      NK_TRY(nk_dev_alloc(&P->d_fd_f0, (size_t)P->n_local + 1));
      NK_TRY(nk_dev_alloc(&P->d_fd_up, (size_t)P->n_local + 1));
      NK_TRY(nk_dev_alloc(&P->d_fd_f1, (size_t)P->n_local + 1));
    }
    if (P->cb.residual(P->user, d_u, P->d_fd_f0, (void *)P->ctx->stream) != 0)
      NK_FAIL(NK_E_CALLBACK, "residual callback failed");
  }
  return NK_OK;
}

// Forward-difference directional derivative Jv ≈ (f(u + εv) − f(u))/ε with ε = √eps — what the reference's
// JacobianOperator does for a NonlinearFunction without jvp under AutoFiniteDiff (DI.pushforward!,
// SciMLJacobianOperators.jl:396-414; step rule FiniteDiff.jl [EXT]: absstep = relstep = √eps at t = 0).
#define NK_FD_EPS 1.4901161193847656e-08
__global__ __launch_bounds__(NK_BLOCK) void k_fd_perturb(int64_t n, const double *__restrict__ u,
                                                         const double *__restrict__ v, double eps,
                                                         double *__restrict__ up) {
  const int64_t stride = (int64_t)gridDim.x * NK_BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x; i < n; i += stride) up[i] = u[i] + eps * v[i];
}
__global__ __launch_bounds__(NK_BLOCK) void k_fd_diff(int64_t n, const double *__restrict__ f1,
                                                      const double *__restrict__ f0, double inv_eps,
                                                      double *__restrict__ jv) {
  const int64_t stride = (int64_t)gridDim.x * NK_BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x; i < n; i += stride) jv[i] = (f1[i] - f0[i]) * inv_eps;
}

int nk_problem_jvp_dev(nk_problem *P, const double *d_u, const double *d_v, double *d_jv, const int *d_skip,
                       const double *d_out_scale, const nk_spmv_epi *epi) {
  if (epi && epi->mode != 0 && P->kind != NK_PROBLEM_BRATU2D)  // (modes: 1 fused Chebyshev step, 2 fused residual b − Jv)
    NK_FAIL(NK_E_UNSUPPORTED, "internal: fused epilogue only exists for the Bratu JVP");
  nk_spmv_epi ep{};
  if (epi) ep = *epi;
  nk_ctx *ctx = P->ctx;
  const int64_t n = P->n_local;
  ctx->stats.op_applies++;
  if (n == 0) return NK_OK;
  if (P->kind == NK_PROBLEM_BRATU2D && (P->d_u_lin != d_u || !P->d_diag)) NK_TRY(nk_problem_jvp_prepare(P, d_u));
  const bool jac_jvp = P->kind == NK_PROBLEM_USER && !P->cb.jvp && P->cb.jac_values && P->user_pattern;
  if (jac_jvp) {
    if (d_out_scale) NK_FAIL(NK_E_INVALID, "internal: output scale is not supported for callback JVPs");
    NK_TRY(user_lin_J(P, d_u));
    return nk_csr_spmv_dev(P->lin_J, d_v, d_jv, d_skip);
  }
  if (P->kind == NK_PROBLEM_USER && !P->cb.jvp && (P->d_u_lin != d_u || !P->d_fd_f0)) NK_TRY(nk_problem_jvp_prepare(P, d_u));
  nk_prof_scope prof_(ctx, NK_K_JVP, 24.0 * (double)n);
  switch (P->kind) {
    case NK_PROBLEM_QUADRATIC:
      NK_LAUNCH(ctx, k_quad_jvp, dim3(grid1(n)), dim3(NK_BLOCK), n, d_u, d_v, d_jv, d_skip,
                         d_out_scale);
      break;
    case NK_PROBLEM_BRATU2D: {
      const double *lo, *hi;
      NK_TRY(nk_halo_exchange(ctx, &P->halo, d_v));
      halo_lines(P, 1, false, &lo, &hi);
      static const bool flat = getenv("NK_BRATU_JVP_FLAT") != nullptr;  // A/B switch: one point per thread
      const int64_t nl = P->j1 - P->j0;
      if (flat || P->ns >= (1ll << 30)) {
        NK_LAUNCH(ctx, k_bratu_jvp, dim3(grid1(n)), dim3(NK_BLOCK), P->ns, nl, P->c_lap, P->d_diag, d_v, lo, hi, d_jv,
                  d_skip, d_out_scale, ep);
      } else {
        constexpr int TY = 4;
        const dim3 grid((unsigned)((P->ns + NK_BLOCK - 1) / NK_BLOCK), (unsigned)((nl + TY - 1) / TY));
        NK_LAUNCH(ctx, k_bratu_jvp_tile<TY>, grid, dim3(NK_BLOCK), (int)P->ns, (int)nl, P->c_lap, P->d_diag, d_v, lo, hi,
                  d_jv, d_skip, d_out_scale, ep);
      }
      break;
    }
    case NK_PROBLEM_BRUSSELATOR2D: {
      const double *lo, *hi;
      NK_TRY(nk_halo_exchange(ctx, &P->halo, d_v));
      halo_lines(P, 2, true, &lo, &hi);
      NK_LAUNCH(ctx, k_brus_jvp, dim3(grid1(n / 2)), dim3(NK_BLOCK), brus_params(P), 0, d_u, d_v,
                         lo, hi, d_jv, d_skip, d_out_scale);
      break;
    }
    case NK_PROBLEM_USER:
      if (d_out_scale) NK_FAIL(NK_E_INVALID, "internal: output scale is not supported for callback JVPs");
      if (!P->cb.jvp) {  // forward differences through the residual callback
        NK_LAUNCH(ctx, k_fd_perturb, dim3(grid1(n)), dim3(NK_BLOCK), n, d_u, d_v, NK_FD_EPS, P->d_fd_up);
        if (P->cb.residual(P->user, P->d_fd_up, P->d_fd_f1, (void *)ctx->stream) != 0)
          NK_FAIL(NK_E_CALLBACK, "residual callback failed");
        NK_LAUNCH(ctx, k_fd_diff, dim3(grid1(n)), dim3(NK_BLOCK), n, P->d_fd_f1, P->d_fd_f0, 1.0 / NK_FD_EPS, d_jv);
        break;
      }
      if (P->cb.jvp(P->user, d_v, d_u, d_jv, (void *)ctx->stream) != 0) NK_FAIL(NK_E_CALLBACK, "jvp callback failed");
      break;
    default:
      NK_FAIL(NK_E_INVALID, "bad problem kind");
  }
  NK_HIP(hipGetLastError());
  return NK_OK;
}

int nk_problem_vjp_dev(nk_problem *P, const double *d_u, const double *d_v, double *d_vj) {
  nk_ctx *ctx = P->ctx;
  const int64_t n = P->n_local;
  if (P->kind == NK_PROBLEM_QUADRATIC || P->kind == NK_PROBLEM_BRATU2D)  // symmetric Jacobians
    return nk_problem_jvp_dev(P, d_u, d_v, d_vj, nullptr);
  ctx->stats.op_applies++;
  if (n == 0) return NK_OK;
  if (P->kind == NK_PROBLEM_BRUSSELATOR2D) {
    const double *lo, *hi;
    NK_TRY(nk_halo_exchange(ctx, &P->halo, d_v));
    halo_lines(P, 2, true, &lo, &hi);
    NK_LAUNCH(ctx, k_brus_jvp, dim3(grid1(n / 2)), dim3(NK_BLOCK), brus_params(P), 1, d_u, d_v, lo,
                       hi, d_vj, (const int *)nullptr, (const double *)nullptr);
    NK_HIP(hipGetLastError());
    return NK_OK;
  }
  if (P->kind == NK_PROBLEM_USER) {
    if (!P->cb.vjp) {  // prepare_vjp without f.vjp: f.jac' · v; without f.jac: the finite-difference Jacobian on the pattern
      NK_REQUIRE(P->user_pattern, "user problem has neither a vjp callback nor a jac_prototype to build Jᵀ from "
                                  "(prepare_vjp, SciMLJacobianOperators.jl:307-322)");
      NK_TRY(user_lin_J(P, d_u));
      return nk_csr_spmv_t_dev(P->lin_J, d_v, d_vj);
    }
    if (P->cb.vjp(P->user, d_v, d_u, d_vj, (void *)ctx->stream) != 0) NK_FAIL(NK_E_CALLBACK, "vjp callback failed");
    return NK_OK;
  }
  NK_FAIL(NK_E_INVALID, "bad problem kind");
}

// ---------------------------------------------------------------------------- Jacobian pattern / values
// internal global index of grid node (i, j), species s for a line-partitioned problem
static int64_t grid_gidx(int64_t ns, int dof, int R, int64_t i, int64_t j, int s) {
  // owner of line j and its range
  int p = (int)(((j + 1) * R - 1) / ns);  // initial guess, then fix up
  if (p < 0) p = 0;
  if (p >= R) p = R - 1;
  for (;;) {
    const int64_t b = ns * p / R, e = ns * (p + 1) / R;
    if (j < b) --p;
    else if (j >= e) ++p;
    else {
      const int64_t nl = e - b;
      return ns * b * dof + i + ns * (j - b) + ns * nl * s;
    }
  }
}

extern "C" int nk_problem_jac_csr(nk_problem *P, nk_csr **out) {
  NK_REQUIRE(P && out, "NULL argument");
  nk_ctx *ctx = P->ctx;
  NK_HIP(hipSetDevice(ctx->device));
  std::vector<int32_t> rp;
  std::vector<int64_t> gc;
  if (P->kind == NK_PROBLEM_QUADRATIC) {
    rp.resize(P->n_local + 1);
    gc.resize(P->n_local);
    for (int64_t i = 0; i < P->n_local; ++i) { rp[i] = (int32_t)i; gc[i] = P->row_begin + i; }
    rp[P->n_local] = (int32_t)P->n_local;
    return nk_csr_create_local(ctx, P->n_local, P->n_global, P->row_begin, rp, gc, nullptr, out);
  }
  if (P->kind == NK_PROBLEM_BRATU2D) {
    const int64_t ns = P->ns;
    rp.reserve(P->n_local + 1);
    gc.reserve(5 * P->n_local);
    for (int64_t j = P->j0; j < P->j1; ++j)
      for (int64_t i = 0; i < ns; ++i) {
        const int64_t k = j * ns + i;  // contiguous line partition ⇒ internal global index = lexicographic
        rp.push_back((int32_t)gc.size());
        if (j > 0) gc.push_back(k - ns);
        if (i > 0) gc.push_back(k - 1);
        gc.push_back(k);
        if (i < ns - 1) gc.push_back(k + 1);
        if (j < ns - 1) gc.push_back(k + ns);
      }
    rp.push_back((int32_t)gc.size());
    return nk_csr_create_local(ctx, P->n_local, P->n_global, P->row_begin, rp, gc, nullptr, out, P->replicated);
  }
  if (P->kind == NK_PROBLEM_BRUSSELATOR2D) {
    const int64_t N = P->ns, nl = P->j1 - P->j0, nn = N * nl;
    const int R = P->replicated ? 1 : ctx->nranks;  // (a replicated problem numbers its unknowns like a single rank)
    std::vector<uint8_t> role;
    std::vector<int32_t> node;
    rp.reserve(P->n_local + 1);
    struct ent { int64_t c; uint8_t role; };
    for (int s = 0; s < 2; ++s)
      for (int64_t jl = 0; jl < nl; ++jl)
        for (int64_t i = 0; i < N; ++i) {
          const int64_t j = P->j0 + jl;
          const int64_t ip1 = (i + 1 == N) ? 0 : i + 1, im1 = (i == 0) ? N - 1 : i - 1;
          const int64_t jp1 = (j + 1 == N) ? 0 : j + 1, jm1 = (j == 0) ? N - 1 : j - 1;
          ent e[6] = {{grid_gidx(N, 2, R, im1, j, s), 0}, {grid_gidx(N, 2, R, ip1, j, s), 0},
                      {grid_gidx(N, 2, R, i, jp1, s), 0}, {grid_gidx(N, 2, R, i, jm1, s), 0},
                      {grid_gidx(N, 2, R, i, j, s), (uint8_t)(s == 0 ? 1 : 3)},
                      {grid_gidx(N, 2, R, i, j, 1 - s), (uint8_t)(s == 0 ? 2 : 4)}};
          std::sort(e, e + 6, [](const ent &a, const ent &b) { return a.c < b.c; });
          rp.push_back((int32_t)gc.size());
          for (int t = 0; t < 6; ++t) {
            gc.push_back(e[t].c);
            role.push_back(e[t].role);
            node.push_back((int32_t)(i + N * jl));
          }
        }
    rp.push_back((int32_t)gc.size());
    NK_TRY(nk_csr_create_local(ctx, P->n_local, P->n_global, P->row_begin, rp, gc, nullptr, out, P->replicated));
    nk_csr *Jc = *out;
    NK_TRY(nk_dev_alloc(&Jc->d_role, role.size()));
    NK_TRY(nk_dev_alloc(&Jc->d_node, node.size()));
    NK_HIP(nk_memcpy(ctx, Jc->d_role, role.data(), role.size(), hipMemcpyHostToDevice));
    NK_HIP(nk_memcpy(ctx, Jc->d_node, node.data(), node.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    (void)nn;
    return NK_OK;
  }
  if (P->kind == NK_PROBLEM_USER) {
    NK_REQUIRE(P->user_pattern, "user problem was created without a Jacobian pattern");
    *out = P->user_pattern;
    return NK_OK;
  }
  NK_FAIL(NK_E_INVALID, "bad problem kind");
}

int nk_problem_jac_values_dev(nk_problem *P, const double *d_u, nk_csr *J, const nk_fold_norms *fold, bool *folded,
                              const nk_ss_begin_args *begin) {
  nk_ctx *ctx = P->ctx;
  const int64_t n = P->n_local;
  if (folded) *folded = false;
  J->t_values_stale = true; J->bounds_valid = false; J->bounds_pending = false;
  if (n == 0) return NK_OK;
  nk_prof_scope prof_(ctx, NK_K_JACFILL, 8.0 * (double)J->nnz + 8.0 * (double)n);
  switch (P->kind) {
    case NK_PROBLEM_QUADRATIC:
      NK_LAUNCH(ctx, k_quad_jac, dim3(grid1(n)), dim3(NK_BLOCK), n, d_u, J->d_val);
      break;
    case NK_PROBLEM_BRATU2D:
    {
      // one rank: the fill also leaves the Gershgorin bounds of the new Jacobian (the s-step Newton basis asks for them before
      // every linear solve); several ranks: the bounds are all-reduced, computed on demand
      const int g = grid1(n);
      double *gpart = nullptr;
      const bool take_fold = fold != nullptr && folded != nullptr && fold->partials != nullptr && ctx->nranks == 1;
      const bool begin_rides = take_fold && begin != nullptr && begin->ctl != nullptr;
      if (begin_rides) {
        // the cycle begin in this launch's first workgroup reduces the bounds of THIS Jacobian (from the residual kernel's
        // partials) into the matrix's bounds word: nothing pending, nothing for this kernel to compute
        NK_REQUIRE(begin->ival != nullptr && begin->ival == nk_csr_bounds_word(J), "internal: a cycle begin folded into the fill must reduce into the matrix's bounds word");
        J->bounds_valid = true;
      } else if (ctx->nranks == 1 && J->nblocks > 0) {
        if (!J->d_gersh || J->gersh_cap < 2 * g) {
          hipFree(J->d_gersh);
          J->d_gersh = nullptr;
          const int cap = 2 * (g > J->nblocks ? g : J->nblocks) + 2;
          NK_TRY(nk_dev_alloc(&J->d_gersh, (size_t)cap));
          J->gersh_cap = cap;
        }
        gpart = J->d_gersh;
      }
      NK_LAUNCH(ctx, k_bratu_jac, dim3(g + (take_fold ? 1 : 0)), dim3(NK_BLOCK), P->ns, P->j1 - P->j0, P->j0, P->c_lap, P->c_exp, d_u,
                J->d_rowptr, J->d_val, gpart, take_fold ? *fold : nk_fold_norms{},
                begin_rides ? *begin : nk_ss_begin_args{});
      if (take_fold) *folded = true;
      if (gpart) NK_TRY(nk_csr_bounds_from_partials(J, gpart, g));
      break;
    }
    case NK_PROBLEM_BRUSSELATOR2D: {
      NK_REQUIRE(J->d_role && J->d_node, "CSR was not created by nk_problem_jac_csr for a Brusselator problem");
      const nk_csr *ex = J;
      const brus_par q = brus_params(P);
      NK_LAUNCH(ctx, k_brus_jac, dim3(grid1(J->nnz)), dim3(NK_BLOCK), J->nnz, q.N * q.nl, q.A,
                         q.alpha, ex->d_role, ex->d_node, d_u, J->d_val);
      break;
    }
    case NK_PROBLEM_USER:
      if (!P->cb.jac_values) return nk_problem_jac_colored_dev(P, d_u, J);  // AutoSparse(AutoFiniteDiff) analogue
      if (P->cb.jac_values(P->user, d_u, J->d_val, (void *)ctx->stream) != 0)
        NK_FAIL(NK_E_CALLBACK, "jac_values callback failed");
      break;
    default:
      NK_FAIL(NK_E_INVALID, "bad problem kind");
  }
  NK_HIP(hipGetLastError());
  return NK_OK;
}

// ---------------------------------------------------------------------------- exported wrappers with memspace
static int stage(nk_problem *P, int slot, const double *src, int memspace, const double **dst) {
  if (memspace == NK_DEVICE) { *dst = src; return NK_OK; }
  if (!P->d_tmp[slot]) NK_TRY(nk_dev_alloc(&P->d_tmp[slot], (size_t)P->n_local + 1));
  NK_HIP(hipMemcpyAsync(P->d_tmp[slot], src, P->n_local * sizeof(double), hipMemcpyHostToDevice, P->ctx->stream));
  *dst = P->d_tmp[slot];
  return NK_OK;
}
static int out_begin(nk_problem *P, int slot, double *dst, int memspace, double **dev) {
  if (memspace == NK_DEVICE) { *dev = dst; return NK_OK; }
  if (!P->d_tmp[slot]) NK_TRY(nk_dev_alloc(&P->d_tmp[slot], (size_t)P->n_local + 1));
  *dev = P->d_tmp[slot];
  return NK_OK;
}
static int out_end(nk_problem *P, double *dst, int memspace, const double *dev) {
  if (memspace == NK_DEVICE) return NK_OK;
  NK_HIP(hipMemcpyAsync(dst, dev, P->n_local * sizeof(double), hipMemcpyDeviceToHost, P->ctx->stream));
  NK_HIP(hipStreamSynchronize(P->ctx->stream));
  return NK_OK;
}

extern "C" int nk_problem_initial_guess(nk_problem *P, double *u0, int memspace) {
  NK_REQUIRE(P && u0, "NULL argument");
  NK_HIP(hipSetDevice(P->ctx->device));
  double *d;
  NK_TRY(out_begin(P, 0, u0, memspace, &d));
  switch (P->kind) {
    case NK_PROBLEM_QUADRATIC: NK_TRY(nk_blas_fill(P->ctx, P->n_local, 1.0, d)); break;
    case NK_PROBLEM_BRATU2D: NK_TRY(nk_blas_fill(P->ctx, P->n_local, 0.0, d)); break;
    case NK_PROBLEM_BRUSSELATOR2D:
      if (P->n_local)
        NK_LAUNCH(P->ctx, k_brus_u0, dim3(grid1(P->n_local / 2)), dim3(NK_BLOCK), brus_params(P), d);
      NK_HIP(hipGetLastError());
      break;
    default: NK_FAIL(NK_E_UNSUPPORTED, "no built-in initial guess for this problem kind");
  }
  return out_end(P, u0, memspace, d);
}
extern "C" int nk_residual(nk_problem *P, const double *u, double *f, int memspace) {
  NK_REQUIRE(P && u && f, "NULL argument");
  NK_HIP(hipSetDevice(P->ctx->device));
  const double *du;
  double *df;
  NK_TRY(stage(P, 0, u, memspace, &du));
  NK_TRY(out_begin(P, 1, f, memspace, &df));
  NK_TRY(nk_problem_residual_dev(P, du, df));
  return out_end(P, f, memspace, df);
}
static int jvp_any(nk_problem *P, const double *u, const double *v, double *jv, int memspace, bool transpose) {
  NK_REQUIRE(P && u && v && jv, "NULL argument");
  NK_HIP(hipSetDevice(P->ctx->device));
  const double *du, *dv;
  double *dj;
  NK_TRY(stage(P, 0, u, memspace, &du));
  NK_TRY(stage(P, 1, v, memspace, &dv));
  NK_TRY(out_begin(P, 2, jv, memspace, &dj));
  nk_problem_invalidate(P);  // force re-linearisation: the caller's u may have changed in place
  NK_TRY(transpose ? nk_problem_vjp_dev(P, du, dv, dj) : nk_problem_jvp_dev(P, du, dv, dj, nullptr));
  return out_end(P, jv, memspace, dj);
}
extern "C" int nk_jvp(nk_problem *P, const double *u, const double *v, double *Jv, int memspace) {
  return jvp_any(P, u, v, Jv, memspace, false);
}
extern "C" int nk_vjp(nk_problem *P, const double *u, const double *v, double *vJ, int memspace) {
  return jvp_any(P, u, v, vJ, memspace, true);
}
extern "C" int nk_jac_values(nk_problem *P, const double *u, int memspace, nk_csr *J) {
  NK_REQUIRE(P && u && J, "NULL argument");
  NK_HIP(hipSetDevice(P->ctx->device));
  const double *du;
  NK_TRY(stage(P, 0, u, memspace, &du));
  NK_TRY(nk_problem_jac_values_dev(P, du, J));
  if (memspace != NK_DEVICE) NK_HIP(hipStreamSynchronize(P->ctx->stream));
  return NK_OK;
}

// ---------------------------------------------------------------------------- colour-compressed assembly
// Column colouring by greedy distance-2 (structurally orthogonal columns), seeds s_c = 1 on colour c,
// B[:,c] = J s_c by the matrix-free JVP, decompression vals[p] = B[row(p), colour(col(p))]. This is the
// shape of DI.jacobian! with AutoSparse + column colouring (jacobian.jl:244-247).
__global__ __launch_bounds__(NK_BLOCK) void k_seed(int64_t n, const int32_t *__restrict__ color, int c,
                                                   double *__restrict__ s) {
  const int64_t i = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x;
  if (i < n) s[i] = (color[i] == c) ? 1.0 : 0.0;
}
__global__ __launch_bounds__(NK_BLOCK) void k_decompress(int64_t nrows, const int32_t *__restrict__ rowptr,
                                                         const int32_t *__restrict__ colcolor_of_nnz, int c,
                                                         const double *__restrict__ Bc, double *__restrict__ vals) {
  const int64_t r = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x;
  if (r >= nrows) return;
  for (int32_t p = rowptr[r]; p < rowptr[r + 1]; ++p)
    if (colcolor_of_nnz[p] == c) vals[p] = Bc[r];
}

// greedy distance-2 column colouring, natural order (columns sharing a row get different colours) —
// SparseMatrixColorings' GreedyColoringAlgorithm() default, [EXT]. rp/col: a square pattern with n rows.
static int greedy_column_coloring(int64_t n, const std::vector<int32_t> &rp, const std::vector<int32_t> &col,
                                  std::vector<int32_t> &color) {
  const int64_t nnz = (int64_t)col.size();
  std::vector<int32_t> cnt(n + 1, 0);
  for (int64_t p = 0; p < nnz; ++p) cnt[col[p] + 1]++;
  for (int64_t c = 0; c < n; ++c) cnt[c + 1] += cnt[c];
  std::vector<int32_t> rows_of_col(nnz), fillp(cnt.begin(), cnt.end() - 1);
  for (int64_t r = 0; r < n; ++r)
    for (int32_t p = rp[r]; p < rp[r + 1]; ++p) rows_of_col[fillp[col[p]]++] = (int32_t)r;
  color.assign(n, -1);
  std::vector<int32_t> mark(n + 1, -1);
  int ncolors = 0;
  for (int64_t c = 0; c < n; ++c) {
    for (int32_t q = cnt[c]; q < cnt[c + 1]; ++q) {
      const int32_t r = rows_of_col[q];
      for (int32_t p = rp[r]; p < rp[r + 1]; ++p) {
        const int32_t oc = color[col[p]];
        if (oc >= 0) mark[oc] = (int32_t)c;
      }
    }
    int32_t k = 0;
    while (mark[k] == (int32_t)c) ++k;
    color[c] = k;
    if (k + 1 > ncolors) ncolors = k + 1;
  }
  return ncolors;
}

// the WHOLE pattern of a built-in grid problem in the internal global numbering (every rank can write it down): what a
// row-partitioned run colours, so that all ranks agree on the colour of every column without communication
static int grid_global_pattern(const nk_problem *P, std::vector<int32_t> &rp, std::vector<int32_t> &col) {
  const int R = P->ctx->nranks;
  const int64_t N = P->ns;
  rp.clear();
  col.clear();
  if (P->kind == NK_PROBLEM_BRATU2D) {
    for (int64_t j = 0; j < N; ++j)
      for (int64_t i = 0; i < N; ++i) {
        const int64_t k = j * N + i;
        rp.push_back((int32_t)col.size());
        if (j > 0) col.push_back((int32_t)(k - N));
        if (i > 0) col.push_back((int32_t)(k - 1));
        col.push_back((int32_t)k);
        if (i < N - 1) col.push_back((int32_t)(k + 1));
        if (j < N - 1) col.push_back((int32_t)(k + N));
      }
    rp.push_back((int32_t)col.size());
    return NK_OK;
  }
  if (P->kind == NK_PROBLEM_BRUSSELATOR2D) {
    for (int p = 0; p < R; ++p) {  // rows in the order of their internal ids: rank, species, grid line, point
      const int64_t b = N * p / R, e = N * (p + 1) / R;
      for (int s = 0; s < 2; ++s)
        for (int64_t j = b; j < e; ++j)
          for (int64_t i = 0; i < N; ++i) {
            const int64_t ip1 = (i + 1 == N) ? 0 : i + 1, im1 = (i == 0) ? N - 1 : i - 1;
            const int64_t jp1 = (j + 1 == N) ? 0 : j + 1, jm1 = (j == 0) ? N - 1 : j - 1;
            int64_t c[6] = {grid_gidx(N, 2, R, im1, j, s), grid_gidx(N, 2, R, ip1, j, s), grid_gidx(N, 2, R, i, jp1, s),
                            grid_gidx(N, 2, R, i, jm1, s), grid_gidx(N, 2, R, i, j, s), grid_gidx(N, 2, R, i, j, 1 - s)};
            std::sort(c, c + 6);
            rp.push_back((int32_t)col.size());
            for (int t = 0; t < 6; ++t)
              if (t == 0 || c[t] != c[t - 1]) col.push_back((int32_t)c[t]);
          }
    }
    rp.push_back((int32_t)col.size());
    return NK_OK;
  }
  NK_FAIL(NK_E_UNSUPPORTED, "coloured assembly on several ranks needs a built-in grid problem (the pattern of a user problem "
                            "is only known slice by slice)");
}

// column colours of J's local index space [owned | halo], computed once per pattern
static int csr_ensure_coloring(nk_problem *P, nk_csr *J) {
  if (J->ncolors > 0) return NK_OK;
  const int64_t n = J->nrows, nloc = n + (int64_t)J->halo_gcols.size();
  std::vector<int32_t> color_loc(nloc > 0 ? nloc : 1, 0);
  int ncolors = 0;
  if (J->ctx->nranks == 1) {
    std::vector<int32_t> color;
    ncolors = greedy_column_coloring(n, J->h_rowptr, J->h_col, color);
    for (int64_t c = 0; c < n; ++c) color_loc[c] = color[c];
  } else {
    NK_REQUIRE(P->n_global < (1ll << 31), "pattern too large for a 32-bit colouring");
    std::vector<int32_t> rp, col, color;
    NK_TRY(grid_global_pattern(P, rp, col));
    NK_REQUIRE((int64_t)rp.size() - 1 == J->n_global, "pattern/problem size mismatch");
    ncolors = greedy_column_coloring(J->n_global, rp, col, color);
    for (int64_t c = 0; c < n; ++c) color_loc[c] = color[J->row_begin + c];
    for (size_t h = 0; h < J->halo_gcols.size(); ++h) color_loc[n + h] = color[J->halo_gcols[h]];
  }
  std::vector<int32_t> nnzcolor(J->nnz);
  for (int64_t p = 0; p < J->nnz; ++p) nnzcolor[p] = color_loc[J->h_col[p]];
  NK_TRY(nk_dev_alloc(&J->d_color, (size_t)n + 1));
  NK_TRY(nk_dev_alloc(&J->d_nnzcolor, (size_t)J->nnz + 1));
  NK_TRY(nk_dev_alloc(&J->d_seed, (size_t)n + 1));
  NK_TRY(nk_dev_alloc(&J->d_B, (size_t)n + 1));
  if (n) NK_HIP(nk_memcpy(J->ctx, J->d_color, color_loc.data(), n * sizeof(int32_t), hipMemcpyHostToDevice));
  if (J->nnz) NK_HIP(nk_memcpy(J->ctx, J->d_nnzcolor, nnzcolor.data(), J->nnz * sizeof(int32_t), hipMemcpyHostToDevice));
  J->ncolors = ncolors;
  return NK_OK;
}

int nk_problem_jac_colored_dev(nk_problem *P, const double *d_u, nk_csr *J) {
  nk_ctx *ctx = P->ctx;
  NK_REQUIRE(J->nrows == P->n_local && J->n_global == P->n_global, "pattern/problem size mismatch");
  NK_TRY(csr_ensure_coloring(P, J));
  const int64_t n = J->nrows;
  NK_TRY(nk_problem_jvp_prepare(P, d_u));  // the caller's u may have changed in place
  for (int c = 0; c < J->ncolors; ++c) {
    NK_LAUNCH(ctx, k_seed, dim3(grid1(n)), dim3(NK_BLOCK), n, J->d_color, c, J->d_seed);
    NK_TRY(nk_problem_jvp_dev(P, d_u, J->d_seed, J->d_B, nullptr));
    NK_LAUNCH(ctx, k_decompress, dim3(grid1(n)), dim3(NK_BLOCK), n, J->d_rowptr, J->d_nnzcolor, c, J->d_B, J->d_val);
  }
  NK_HIP(hipGetLastError());
  J->t_values_stale = true; J->bounds_valid = false; J->bounds_pending = false;
  return NK_OK;
}

extern "C" int nk_jac_values_colored(nk_problem *P, const double *u, int memspace, nk_csr *J, int *ncolors_out) {
  NK_REQUIRE(P && u && J, "NULL argument");
  NK_HIP(hipSetDevice(P->ctx->device));
  const double *du;
  NK_TRY(stage(P, 0, u, memspace, &du));
  NK_TRY(nk_problem_jac_colored_dev(P, du, J));
  NK_HIP(hipStreamSynchronize(P->ctx->stream));
  if (ncolors_out) *ncolors_out = J->ncolors;
  return NK_OK;
}
