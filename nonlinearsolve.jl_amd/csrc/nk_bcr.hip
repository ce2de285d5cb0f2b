// Block cyclic reduction — the parallel engine of the direct `linsolve` (config C2: `linsolve = nothing` on a concrete sparse
// J; factorisation reuse semantics of lib/NonlinearSolveBase/ext/NonlinearSolveBaseLinearSolveExt.jl:81-86 as in nk_band.hip).
//
// A matrix of bandwidth ≤ b is block tridiagonal with b × b blocks (sub-diagonal A_i, diagonal D_i, super-diagonal C_i,
// m = ⌈n/b⌉ block rows). The right-looking band LU of nk_band.hip walks a chain of n/32 dependent block columns (2048 links
// of ≈ 28 µs at C2: 57 ms, ≈ 0.15 TFLOP/s). Cyclic reduction trades ≈ 7× the flops for a chain of log₂ m levels whose work is
// batched dense b × b algebra — FP64 MFMA GEMMs (`v_mfma_f64_16x16x4_f64`) and in-register Gauss–Jordan inversions:
//   level:  for every odd row k     D_k ← D_k⁻¹
//           for every even row j    P_j = A_j D_{j−1}⁻¹,  Q_j = C_j D_{j+1}⁻¹
//                                   D'_{j/2} = D_j − P_j C_{j−1} − Q_j A_{j+1},  A'_{j/2} = −P_j A_{j−1},  C'_{j/2} = −Q_j C_{j+1}
//           recurse on the even rows (⌈m/2⌉ of them); the last level inverts its single block
//   solve:  down  f'_{j/2} = f_j − P_j f_{j−1} − Q_j f_{j+1};   bottom  x = D⁻¹ f;
//           up    x_k = D_k⁻¹ (f_k − A_k x_{k−1} − C_k x_{k+1})  for the odd rows of each level
// Inversion of a block: n ≤ 128 — one workgroup, the matrix in REGISTERS (8 × 8 entries per thread), one LDS broadcast of
// the pivot row and column and one barrier per pivot; larger blocks — 2 × 2 Schur-complement recursion on halves (two
// inversions + six GEMMs). Partial pivoting INSIDE the blocks inverted by one workgroup (≤ 128; implicit row pivoting, see
// k_bcr_inv128); no pivoting across the halves of the Schur recursion or across block rows — the Jacobians of the grid problems
// are diagonally dominant / M-matrices and so are their Schur complements; a vanishing or non-finite pivot raises the failure
// flag, and the nonlinear driver verifies ‖J x − b‖ after every direct solve (one step of iterative refinement, then GMRES on
// the same J) — nk_solver.hip, newton_descent.
// Algorithmic work at C2 (n = 65 536, b = 256, m = 256): ≈ 58 GFLOP of b³ products + 255 block inversions per factorisation
// (band LU: 8.6 GFLOP); a solve streams every stored block once (≈ 1 GB). Memory: ≈ 8 m b² doubles (1.1 GB at C2).
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "nk_internal.h"

typedef double nk_d4 __attribute__((ext_vector_type(4)));

// ----------------------------------------------------------------------------- batched GEMM on FP64 MFMA
// C = beta·C0 + alpha·A·B, column-major, any M, N, K (edges are zero-filled / guarded). Workgroup tile 64 × 64, four
// wavefronts in a 2 × 2 arrangement, each 32 × 32 = 2 × 2 MFMA tiles of 16 × 16; K in slabs of 32 staged through LDS
// (k-major, row pitch 80 doubles: the two k-rows a half-wave reads land in disjoint bank ranges).
// v_mfma_f64_16x16x4_f64 operand maps (cdna_hip_programming.md): A lane l ↦ A[l & 15][l >> 4], B lane l ↦ B[l >> 4][l & 15],
// D register r of lane l ↦ D[(l >> 4) + 4 r][l & 15].
struct bcr_gemm_args {
  const double *A; int64_t sA; int lda;
  const double *B; int64_t sB; int ldb;
  const double *C0; int64_t sC0; int ldc0;
  double *C; int64_t sC; int ldc;
  int M, N, K;
  double alpha, beta;
  int tri;  // structural zeros of one operand (level 0 of a banded matrix): 0 none; 1 A upper triangular (A[m][k] = 0 for k < m);
            // 2 B lower triangular (B[k][n] = 0 for k < n); 3 B upper triangular (k > n); 4 A lower triangular (k > m)
};
constexpr int GT = 64, GK = 32, GLD = 80;
__global__ __launch_bounds__(256) void k_bcr_gemm(bcr_gemm_args g) {
  __shared__ double As[GK * GLD], Bs[GK * GLD];
  const int t = threadIdx.x, l = t & 63, w = t >> 6, wm = w & 1, wn = w >> 1;
  const int tiles_m = (g.M + GT - 1) / GT;
  const int tm = blockIdx.x % tiles_m, tn = blockIdx.x / tiles_m;
  const int m0 = tm * GT, n0 = tn * GT;
  const double *__restrict__ A = g.A + (int64_t)blockIdx.y * g.sA;
  const double *__restrict__ B = g.B + (int64_t)blockIdx.y * g.sB;
  nk_d4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = (nk_d4){0.0, 0.0, 0.0, 0.0};
  // the K range this tile has to visit (whole slabs)
  int kbeg = 0, kend = g.K;
  if (g.tri == 1) kbeg = (m0 / GK) * GK;
  else if (g.tri == 2) kbeg = (n0 / GK) * GK;
  else if (g.tri == 3) kend = min(g.K, n0 + GT);
  else if (g.tri == 4) kend = min(g.K, m0 + GT);
  double av[8], bv[8];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int idx = t + 256 * e;
      const int m = idx & 63, k = idx >> 6;
      const bool ok = (m0 + m < g.M) && (k0 + k < g.K);
      av[e] = ok ? A[(int64_t)(m0 + m) + (int64_t)(k0 + k) * g.lda] : 0.0;
      const int kb = idx & 31, n = idx >> 5;
      const bool okb = (k0 + kb < g.K) && (n0 + n < g.N);
      bv[e] = okb ? B[(int64_t)(k0 + kb) + (int64_t)(n0 + n) * g.ldb] : 0.0;
    }
  };
  if (kbeg < kend) fetch(kbeg);
  for (int k0 = kbeg; k0 < kend; k0 += GK) {
    __syncthreads();  // the previous slab has been consumed
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int idx = t + 256 * e;
      As[(idx >> 6) * GLD + (idx & 63)] = av[e];
      Bs[(idx & 31) * GLD + (idx >> 5)] = bv[e];
    }
    __syncthreads();
    if (k0 + GK < kend) fetch(k0 + GK);  // the next slab's loads fly while this one is multiplied
#pragma unroll
    for (int kk = 0; kk < GK / 4; ++kk) {
      const int kr = (kk * 4 + (l >> 4)) * GLD;
      const double a0 = As[kr + wm * 32 + (l & 15)], a1 = As[kr + wm * 32 + 16 + (l & 15)];
      const double b0 = Bs[kr + wn * 32 + (l & 15)], b1 = Bs[kr + wn * 32 + 16 + (l & 15)];
      acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
    }
  }
  const double *__restrict__ C0 = g.C0 ? g.C0 + (int64_t)blockIdx.y * g.sC0 : nullptr;
  double *__restrict__ C = g.C + (int64_t)blockIdx.y * g.sC;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm * 32 + mi * 16 + (l >> 4) + 4 * r, col = n0 + wn * 32 + ni * 16 + (l & 15);
        if (row < g.M && col < g.N) {
          double v = g.alpha * acc[mi][ni][r];
          if (g.beta != 0.0) v += g.beta * C0[(int64_t)row + (int64_t)col * g.ldc0];
          C[(int64_t)row + (int64_t)col * g.ldc] = v;
        }
      }
}

// ----------------------------------------------------------------------------- batched in-place inverse, n ≤ 128
// Gauss–Jordan on the diagonal pivots, the matrix held in registers: thread (ti, tj) of a 16 × 32 arrangement owns rows
// 8 ti … 8 ti + 7 and columns 4 tj … 4 tj + 3 (rows/columns ≥ n behave as an identity border; 64 entries per thread on 256
// threads overflowed the VGPR file into AGPRs: 108 µs per inversion). Per pivot k the owners of
// column k and of row k publish them to LDS (double buffered: ONE barrier per pivot), every thread then applies
// a_ij ← a_ij − a_ik a_kj / a_kk to its 32 entries; row k, column k and the pivot itself take their Gauss–Jordan values.
// The local index of row/column k inside its owner (k mod 8) is the unrolled inner loop counter, so every register index is
// a compile-time constant.
__device__ __forceinline__ double bcr_rcp(double p) {  // v_rcp_f64 + two Newton steps (≈ 1 ulp; a division costs ≈ 5× more)
  double x = __builtin_amdgcn_rcp(p);
  x = x * (2.0 - p * x);
  x = x * (2.0 - p * x);
  return x;
}
// PIVOT: partial (row) pivoting inside the block, implicit — the pivot of column k is the largest entry among the rows not
// yet used; no row is moved (a row exchange would need run-time register indices): the pivot row is published from wherever
// it lives (a predicated select over the eight local rows of its owners), the elimination skips it by position, and the
// bookkeeping is undone when the result is stored: with p_k the pivot row of column k and σ its inverse, entry (i, j) of the
// register image is entry (σ(i), p_j) of the inverse. Two barriers per pivot instead of one.
__device__ __forceinline__ unsigned bcr_wave_max_u32(unsigned v) {  // max over the 64 lanes, uniform result
  unsigned t;
  t = __builtin_amdgcn_update_dpp(0u, v, 0x111, 0xf, 0xf, false); v = v > t ? v : t;  // row_shr:1
  t = __builtin_amdgcn_update_dpp(0u, v, 0x112, 0xf, 0xf, false); v = v > t ? v : t;  // row_shr:2
  t = __builtin_amdgcn_update_dpp(0u, v, 0x114, 0xf, 0xf, false); v = v > t ? v : t;  // row_shr:4
  t = __builtin_amdgcn_update_dpp(0u, v, 0x118, 0xf, 0xf, false); v = v > t ? v : t;  // row_shr:8 — lane 15 of a row: the row's max
  t = __builtin_amdgcn_update_dpp(0u, v, 0x142, 0xa, 0xf, false); v = v > t ? v : t;  // row_bcast:15 into rows 1, 3
  t = __builtin_amdgcn_update_dpp(0u, v, 0x143, 0xc, 0xf, false); v = v > t ? v : t;  // row_bcast:31 into rows 2, 3
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
template <bool PIVOT>
__global__ __launch_bounds__(512) void k_bcr_inv128(double *__restrict__ mats, int64_t stride, int ld, int n, int *fail) {
  __shared__ double colb[2][128], rowb[2][128];
  __shared__ int prow[128], sigma[128], usedf[128];
  double *__restrict__ M = mats + (int64_t)blockIdx.x * stride;
  const int t = threadIdx.x, ti = t & 15, tj = t >> 4;  // rows 8 ti …, columns 4 tj … (tj = 0 … 31)
  double a[8][4];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int i = ti * 8 + r, j = tj * 4 + c;
      a[r][c] = (i < n && j < n) ? M[(int64_t)i + (int64_t)j * ld] : (i == j ? 1.0 : 0.0);
    }
  if (PIVOT) {
    if (t < 128) { usedf[t] = (t >= n) ? 1 : 0; prow[t] = t; sigma[t] = t; }
    __syncthreads();
  }
  bool bad = false;
#pragma unroll 1
  for (int kb = 0; kb < 16; ++kb) {
    if (kb * 8 >= n) break;  // the identity border needs no elimination
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int k = kb * 8 + r, buf = r & 1;
      const int cj = 2 * kb + (r >> 2), cl = r & 3;  // owner column group and local column of column k
      if (tj == cj) {
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) colb[buf][ti * 8 + rr] = a[rr][cl];
      }
      int p = k;
      if (PIVOT) {
        __syncthreads();
        // Every wavefront finds the pivot row for itself (same data, same order: same answer). The magnitudes are compared as
        // float bit patterns (+2; a NaN ranks 1, a used row 0: the choice always falls on an unused row, so the pivot rows form
        // a permutation whatever the data — the permuted store below relies on that) through a DPP max reduction — six
        // v_max_u32_dpp; a shuffle butterfly on (double, index) pairs cost 18 LDS round trips per pivot, 3× the whole step.
        const int l = t & 63;
        const float f0 = fabsf((float)colb[buf][l]), f1 = fabsf((float)colb[buf][l + 64]);
        unsigned k0 = (f0 == f0) ? __float_as_uint(f0) + 2u : 1u, k1 = (f1 == f1) ? __float_as_uint(f1) + 2u : 1u;
        if (usedf[l]) k0 = 0u;
        if (usedf[l + 64]) k1 = 0u;
        const unsigned km = bcr_wave_max_u32(k0 > k1 ? k0 : k1);
        const unsigned long long b0 = __ballot(k0 == km), b1 = __ballot(k1 == km);
        p = b0 ? (int)__ffsll((long long)b0) - 1 : 64 + (int)__ffsll((long long)b1) - 1;
        if (ti == (p >> 3)) {
          const int rp = p & 7;
#pragma unroll
          for (int cc = 0; cc < 4; ++cc) {
            double v = a[0][cc];
#pragma unroll
            for (int q = 1; q < 8; ++q) v = (rp == q) ? a[q][cc] : v;
            rowb[buf][tj * 4 + cc] = v;
          }
        }
      } else if (ti == kb) {
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) rowb[buf][tj * 4 + cc] = a[r][cc];
      }
      __syncthreads();
      // (the bookkeeping is written only now: before the barrier other wavefronts may still be reading usedf for THIS pivot)
      if (PIVOT && t == 0) { prow[k] = p; usedf[p] = 1; }
      const double piv = colb[buf][p];
      bad = bad || !(fabs(piv) > 1e-290);  // zero, denormal-small or NaN
      const double pinv = bcr_rcp(piv);
      double mr[8], rk[4];
#pragma unroll
      for (int q = 0; q < 8; ++q) mr[q] = colb[buf][ti * 8 + q];
#pragma unroll
      for (int q = 0; q < 4; ++q) rk[q] = rowb[buf][tj * 4 + q] * pinv;
#pragma unroll
      for (int rr = 0; rr < 8; ++rr)
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) a[rr][cc] -= mr[rr] * rk[cc];
      if (PIVOT) {
        const int rp = p & 7;
        const bool prow_owner = (ti == (p >> 3));
        if (prow_owner) {
#pragma unroll
          for (int q = 0; q < 8; ++q)
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) a[q][cc] = (rp == q) ? rk[cc] : a[q][cc];
        }
        if (tj == cj) {
#pragma unroll
          for (int rr = 0; rr < 8; ++rr) a[rr][cl] = -mr[rr] * pinv;
          if (prow_owner) {
#pragma unroll
            for (int q = 0; q < 8; ++q) a[q][cl] = (rp == q) ? pinv : a[q][cl];
          }
        }
      } else {
        if (ti == kb) {
#pragma unroll
          for (int cc = 0; cc < 4; ++cc) a[r][cc] = rk[cc];
        }
        if (tj == cj) {
#pragma unroll
          for (int rr = 0; rr < 8; ++rr) a[rr][cl] = -mr[rr] * pinv;
        }
        if (ti == kb && tj == cj) a[r][cl] = pinv;
      }
    }
  }
  if (bad && t == 0) *fail = 1;
  if (PIVOT) {
    __syncthreads();
    if (t < n) sigma[prow[t]] = t;
    __syncthreads();
  }
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int i = ti * 8 + r, j = tj * 4 + c;
      if (i < n && j < n) {
        if (PIVOT) M[(int64_t)sigma[i] + (int64_t)prow[j] * ld] = a[r][c];
        else M[(int64_t)i + (int64_t)j * ld] = a[r][c];
      }
    }
}

// ----------------------------------------------------------------------------- assembly, copies
// CSR → the dense blocks of level 0 (zeroed beforehand); rows ≥ n of the last block row get a unit diagonal
__global__ __launch_bounds__(NK_BLOCK) void k_bcr_fill(int64_t n, int b, int m, const int32_t *__restrict__ rowptr,
                                                       const int32_t *__restrict__ col, const double *__restrict__ val,
                                                       double *__restrict__ A, double *__restrict__ D, double *__restrict__ C,
                                                       int *fail) {
  const int64_t r = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x;
  const int64_t bb = (int64_t)b * b;
  if (r >= (int64_t)m * b) return;
  const int64_t bi = r / b;
  const int ri = (int)(r - bi * b);
  if (r >= n) {
    D[bi * bb + ri + (int64_t)ri * b] = 1.0;
    return;
  }
  for (int32_t p = rowptr[r]; p < rowptr[r + 1]; ++p) {
    const int64_t c = col[p], bj = c / b;
    const int ci = (int)(c - bj * b);
    double *dst = (bj == bi) ? D : (bj == bi - 1) ? A : (bj == bi + 1) ? C : nullptr;
    if (!dst) { *fail = 1; continue; }  // outside the block tridiagonal (cannot happen for bandwidth ≤ b)
    dst[bi * bb + ri + (int64_t)ci * b] = val[p];
  }
}
// dst block i ← src block 2 i
__global__ __launch_bounds__(NK_BLOCK) void k_bcr_copy_even(int64_t bb, int64_t count, const double *__restrict__ src,
                                                            double *__restrict__ dst) {
  const int64_t e = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x;
  if (e >= bb * count) return;
  const int64_t i = e / bb, o = e - i * bb;
  dst[e] = src[2 * i * bb + o];
}

// ----------------------------------------------------------------------------- solve sweeps
// One kernel for the three block-row products of a solve, out = y0 ∓ M1 v1 ∓ M2 v2, spread over b/32 workgroups per block row
// (a workgroup per block row streamed its 2–3 blocks of 512 KB at the latency-bound rate of one CU: 2.4 ms per solve at C2):
// a workgroup owns a slab of 32 rows; thread (row, g) runs over the columns g, g + 8, …, eight loads in flight per matrix, and
// the eight column groups are summed through LDS in a fixed order.
//   mode 0 (down)   i ↦ j = 2i:    f2_i = f_j − P_i f_{j−1} − Q_i f_{j+1}
//   mode 1 (up, 1)  i ↦ k = 2i+1:  t_k  = x_k − A_k x_{k−1} − C_k x_{k+1}
//   mode 2 (up, 2)  i ↦ k = 2i+1:  x_k  = D_k⁻¹ t_k            (mode 3: the bottom level, k = 0, t_0 = D_0⁻¹ f_0)
struct bcr_gemv_args {
  int b, m, mode;
  const double *M1, *M2;   // block arrays (P,Q | A,C | D,–)
  const double *vin;       // f (mode 0), x (mode 1), t (mode 2), f (mode 3)
  double *out;             // f2 (mode 0), t (mode 1), x (mode 2), t (mode 3)
};
__global__ __launch_bounds__(256) void k_bcr_gemv(bcr_gemv_args g) {
  __shared__ double v1[512], v2[512], red[8][33];
  const int b = g.b, m = g.m, i = blockIdx.x, slab = blockIdx.y;
  const int64_t bb = (int64_t)b * b;
  int rowblk, outblk;            // block row of the inputs' centre / of the output
  const double *M1 = nullptr, *M2 = nullptr, *x1 = nullptr, *x2 = nullptr, *y0 = nullptr;
  double sign = -1.0;
  if (g.mode == 0) {
    const int j = 2 * i;
    rowblk = j; outblk = i;
    y0 = g.vin + (int64_t)j * b;
    if (j >= 1) { M1 = g.M1 + i * bb; x1 = g.vin + (int64_t)(j - 1) * b; }
    if (j + 1 < m) { M2 = g.M2 + i * bb; x2 = g.vin + (int64_t)(j + 1) * b; }
  } else if (g.mode == 1) {
    const int k = 2 * i + 1;
    rowblk = k; outblk = k;
    y0 = g.vin + (int64_t)k * b;
    M1 = g.M1 + k * bb; x1 = g.vin + (int64_t)(k - 1) * b;
    if (k + 1 < m) { M2 = g.M2 + k * bb; x2 = g.vin + (int64_t)(k + 1) * b; }
  } else {
    const int k = (g.mode == 2) ? 2 * i + 1 : 0;
    rowblk = k; outblk = k;
    M1 = g.M1 + k * bb; x1 = g.vin + (int64_t)k * b;
    sign = 1.0;
  }
  (void)rowblk;
  for (int c = threadIdx.x; c < b; c += 256) {
    v1[c] = x1 ? x1[c] : 0.0;
    v2[c] = x2 ? x2[c] : 0.0;
  }
  __syncthreads();
  const int r = threadIdx.x & 31, cg = threadIdx.x >> 5, row = slab * 32 + r;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  if (M1) {
    const double *__restrict__ p = M1 + row;
    for (int c = cg; c < b; c += 32) {   // b is a multiple of 32: columns c, c+8, c+16, c+24
      s0 += p[(int64_t)c * b] * v1[c];
      s1 += p[(int64_t)(c + 8) * b] * v1[c + 8];
      s2 += p[(int64_t)(c + 16) * b] * v1[c + 16];
      s3 += p[(int64_t)(c + 24) * b] * v1[c + 24];
    }
  }
  if (M2) {
    const double *__restrict__ p = M2 + row;
    for (int c = cg; c < b; c += 32) {
      s0 += p[(int64_t)c * b] * v2[c];
      s1 += p[(int64_t)(c + 8) * b] * v2[c + 8];
      s2 += p[(int64_t)(c + 16) * b] * v2[c + 16];
      s3 += p[(int64_t)(c + 24) * b] * v2[c + 24];
    }
  }
  red[cg][r] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (threadIdx.x < 32) {
    double s = ((red[0][r] + red[1][r]) + (red[2][r] + red[3][r])) + ((red[4][r] + red[5][r]) + (red[6][r] + red[7][r]));
    s *= sign;
    if (y0) s += y0[row];
    g.out[(int64_t)outblk * b + row] = s;
  }
}
// up: the even rows take the coarser level's solution, x_{2i} = x2_i
__global__ __launch_bounds__(NK_BLOCK) void k_bcr_scatter(int b, int64_t m2, const double *__restrict__ x2, double *__restrict__ x) {
  const int64_t e = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x;
  if (e >= m2 * b) return;
  const int64_t i = e / b, r = e - i * b;
  x[2 * i * b + r] = x2[e];
}

// ----------------------------------------------------------------------------- host side
struct bcr_level {
  int m = 0;                                              // block rows of this level
  double *A = nullptr, *D = nullptr, *C = nullptr;        // m blocks each; after the factorisation D_k = D_k⁻¹ for odd k
  double *P = nullptr, *Q = nullptr;                      // ⌈m/2⌉ blocks each (index i = j/2 of the even row j)
  double *f = nullptr;                                    // m·b right-hand side / solution of this level
  double *t = nullptr;                                    // m·b: the odd rows' right-hand sides between the two products
};
struct nk_bcr {
  nk_ctx *ctx = nullptr;
  int64_t n = 0;
  int b = 0;
  std::vector<bcr_level> lv;
  double *ws = nullptr;   // workspace of the recursive block inversion: 2 depths × {T, W} × batch × (b/2)²… sized 4·batch·b²/… below
  int64_t ws_slot = 0;    // doubles per (depth, T|W) slot
  int *d_fail = nullptr;
  // Pivoting policy (NK_BCR_PIVOT = auto | always | never; default auto): the diagonal-pivot inversion is 2.5× faster per
  // block (87 vs 215 µs: 5.0 vs 7.4 ms per C2 factorisation), so a factorisation starts on it; a vanishing / non-finite pivot —
  // or the nonlinear driver's residual check failing (nk_bcr_set_pivoting) — switches THIS object to row pivoting for good.
  bool pivot = false, pivot_locked = false;
};

static int bcr_gemm(nk_bcr *S, int batch, int M, int N, int K, double alpha, const double *A, int64_t sA, int lda,
                    const double *B, int64_t sB, int ldb, double beta, const double *C0, int64_t sC0, int ldc0, double *C,
                    int64_t sC, int ldc, int tri = 0) {
  if (batch <= 0 || M <= 0 || N <= 0) return NK_OK;
  bcr_gemm_args g;
  g.A = A; g.sA = sA; g.lda = lda;
  g.B = B; g.sB = sB; g.ldb = ldb;
  g.C0 = C0; g.sC0 = sC0; g.ldc0 = ldc0;
  g.C = C; g.sC = sC; g.ldc = ldc;
  g.M = M; g.N = N; g.K = K;
  g.alpha = alpha; g.beta = beta;
  g.tri = tri;
  const int tiles = ((M + GT - 1) / GT) * ((N + GT - 1) / GT);
  for (int b0 = 0; b0 < batch; b0 += 65535) {  // gridDim.y limit
    const int nb = std::min(65535, batch - b0);
    bcr_gemm_args h = g;
    h.A += (int64_t)b0 * sA; h.B += (int64_t)b0 * sB; h.C += (int64_t)b0 * sC;
    if (h.C0) h.C0 += (int64_t)b0 * sC0;
    NK_LAUNCH(S->ctx, k_bcr_gemm, dim3(tiles, nb), dim3(256), h);
  }
  NK_HIP(hipGetLastError());
  return NK_OK;
}

// in-place inverse of `batch` n × n matrices (leading dimension ld, distance `stride`): n ≤ 128 directly, otherwise by the
// 2 × 2 Schur-complement formulas on [E F; G H] with E of order n1:
//   E ← E⁻¹; T = G E; W = E F; H ← (H − T F)⁻¹; F ← −W H; G ← −H T; E ← E − F T
static int bcr_invert(nk_bcr *S, double *M, int64_t stride, int ld, int n, int batch, int depth) {
  if (batch <= 0) return NK_OK;
  if (n <= 128) {
    if (!S->pivot) NK_LAUNCH(S->ctx, k_bcr_inv128<false>, dim3(batch), dim3(512), M, stride, ld, n, S->d_fail);
    else NK_LAUNCH(S->ctx, k_bcr_inv128<true>, dim3(batch), dim3(512), M, stride, ld, n, S->d_fail);
    NK_HIP(hipGetLastError());
    return NK_OK;
  }
  NK_REQUIRE(depth < 2, "block order %d too large for the block inversion", n);
  const int n1 = (n <= 256) ? 128 : 256, n2 = n - n1;
  double *E = M, *F = M + (int64_t)n1 * ld, *G = M + n1, *H = M + n1 + (int64_t)n1 * ld;
  double *T = S->ws + (int64_t)(2 * depth) * S->ws_slot, *W = S->ws + (int64_t)(2 * depth + 1) * S->ws_slot;
  const int64_t sT = (int64_t)n2 * n1;
  NK_TRY(bcr_invert(S, E, stride, ld, n1, batch, depth + 1));
  NK_TRY(bcr_gemm(S, batch, n2, n1, n1, 1.0, G, stride, ld, E, stride, ld, 0.0, nullptr, 0, 0, T, sT, n2));
  NK_TRY(bcr_gemm(S, batch, n1, n2, n1, 1.0, E, stride, ld, F, stride, ld, 0.0, nullptr, 0, 0, W, sT, n1));
  NK_TRY(bcr_gemm(S, batch, n2, n2, n1, -1.0, T, sT, n2, F, stride, ld, 1.0, H, stride, ld, H, stride, ld));
  NK_TRY(bcr_invert(S, H, stride, ld, n2, batch, depth + 1));
  NK_TRY(bcr_gemm(S, batch, n1, n2, n2, -1.0, W, sT, n1, H, stride, ld, 0.0, nullptr, 0, 0, F, stride, ld));
  NK_TRY(bcr_gemm(S, batch, n2, n1, n2, -1.0, H, stride, ld, T, sT, n2, 0.0, nullptr, 0, 0, G, stride, ld));
  NK_TRY(bcr_gemm(S, batch, n1, n1, n2, -1.0, F, stride, ld, T, sT, n2, 1.0, E, stride, ld, E, stride, ld));
  return NK_OK;
}

int nk_dense_invert128_dev(nk_ctx *ctx, double *d_M, int ld, int n, int *d_fail) {
  NK_REQUIRE(n >= 1 && n <= 128 && ld >= n, "dense inverse: 1 ≤ n ≤ 128");
  NK_LAUNCH(ctx, k_bcr_inv128<true>, dim3(1), dim3(512), d_M, (int64_t)0, ld, n, d_fail);
  NK_HIP(hipGetLastError());
  return NK_OK;
}

void nk_bcr_destroy(nk_bcr *S) {
  if (!S) return;
  for (size_t l = 0; l < S->lv.size(); ++l) {
    bcr_level &L = S->lv[l];
    hipFree(L.A); hipFree(L.D); hipFree(L.C); hipFree(L.P); hipFree(L.Q); hipFree(L.f); hipFree(L.t);
  }
  hipFree(S->ws);
  hipFree(S->d_fail);
  delete S;
}

// bytes the engine would allocate for an n × n matrix of half bandwidth ≤ b
int64_t nk_bcr_bytes(int64_t n, int b) {
  int64_t m = (n + b - 1) / b, blocks = 0;
  const int64_t m0 = m;
  for (;;) {
    blocks += 3 * m + (m > 1 ? 2 * ((m + 1) / 2) : 0);
    if (m == 1) break;
    m = (m + 1) / 2;
  }
  return (blocks * (int64_t)b * b + 4 * std::max<int64_t>(1, m0 / 2) * ((int64_t)b * b / 2)) * 8;
}

int nk_bcr_create(nk_ctx *ctx, int64_t n, int b, nk_bcr **out) {
  NK_REQUIRE(b >= 32 && b % 32 == 0 && b <= 512, "block order %d not supported by the cyclic-reduction engine", b);
  nk_bcr *S = new nk_bcr();
  auto guard = nk_make_guard(S, [](nk_bcr *s) { nk_bcr_destroy(s); });
  S->ctx = ctx;
  S->n = n;
  S->b = b;
  const int64_t bb = (int64_t)b * b;
  int m = (int)((n + b - 1) / b);
  const int m0 = m;
  for (;;) {
    bcr_level L;
    L.m = m;
    NK_TRY(nk_dev_alloc(&L.A, (size_t)(m * bb)));
    NK_TRY(nk_dev_alloc(&L.D, (size_t)(m * bb)));
    NK_TRY(nk_dev_alloc(&L.C, (size_t)(m * bb)));
    NK_TRY(nk_dev_alloc(&L.f, (size_t)m * b));
    NK_TRY(nk_dev_alloc(&L.t, (size_t)m * b));
    if (m > 1) {
      const int m2 = (m + 1) / 2;
      NK_TRY(nk_dev_alloc(&L.P, (size_t)(m2 * bb)));
      NK_TRY(nk_dev_alloc(&L.Q, (size_t)(m2 * bb)));
    }
    S->lv.push_back(L);
    if (m == 1) break;
    m = (m + 1) / 2;
  }
  if (b > 128) {
    S->ws_slot = (int64_t)std::max(1, m0 / 2 + 1) * (bb / 2);  // T or W of one depth: batch × n2 × n1 ≤ batch × b²/4 … b²/2 is safe
    NK_TRY(nk_dev_alloc(&S->ws, (size_t)(4 * S->ws_slot)));
  }
  NK_TRY(nk_dev_alloc(&S->d_fail, (size_t)1));
  {
    const char *pv = getenv("NK_BCR_PIVOT");
    if (pv && !strcmp(pv, "always")) S->pivot = true;
    if (pv && !strcmp(pv, "never")) S->pivot_locked = true;
  }
  *out = guard.release();
  return NK_OK;
}

static int bcr_factor_once(nk_bcr *S, nk_csr *Acsr, int *ok);
int nk_bcr_factor(nk_bcr *S, nk_csr *Acsr, int *ok) {
  NK_TRY(bcr_factor_once(S, Acsr, ok));
  if (!*ok && !S->pivot && !S->pivot_locked) {  // a diagonal pivot broke down: this matrix family needs row pivoting
    S->pivot = true;
    NK_TRY(bcr_factor_once(S, Acsr, ok));
  }
  return NK_OK;
}
// 1: switch to row pivoting inside the blocks (returns through *changed whether that is news); the caller refactorises
int nk_bcr_set_pivoting(nk_bcr *S, int on, int *changed) {
  const bool want = on != 0 && !S->pivot_locked;
  if (changed) *changed = (want && !S->pivot) ? 1 : 0;
  if (want) S->pivot = true;
  return NK_OK;
}
static int bcr_factor_once(nk_bcr *S, nk_csr *Acsr, int *ok) {
  nk_ctx *ctx = S->ctx;
  const int b = S->b;
  const int64_t bb = (int64_t)b * b;
  bcr_level &L0 = S->lv[0];
  NK_HIP(hipMemsetAsync(S->d_fail, 0, sizeof(int), ctx->stream));
  NK_HIP(hipMemsetAsync(L0.A, 0, (size_t)(L0.m * bb) * sizeof(double), ctx->stream));
  NK_HIP(hipMemsetAsync(L0.D, 0, (size_t)(L0.m * bb) * sizeof(double), ctx->stream));
  NK_HIP(hipMemsetAsync(L0.C, 0, (size_t)(L0.m * bb) * sizeof(double), ctx->stream));
  {
    const int64_t rows = (int64_t)L0.m * b;
    NK_LAUNCH(ctx, k_bcr_fill, dim3((unsigned)((rows + NK_BLOCK - 1) / NK_BLOCK)), dim3(NK_BLOCK), S->n, b, L0.m,
              (const int32_t *)Acsr->d_rowptr, (const int32_t *)Acsr->d_col, (const double *)Acsr->d_val, L0.A, L0.D, L0.C,
              S->d_fail);
    NK_HIP(hipGetLastError());
  }
  for (size_t l = 0; l < S->lv.size(); ++l) {
    bcr_level &L = S->lv[l];
    const int m = L.m;
    if (m == 1) {
      NK_TRY(bcr_invert(S, L.D, bb, b, b, 1, 0));
      break;
    }
    bcr_level &N = S->lv[l + 1];
    const int m2 = N.m, cP = (m - 1) / 2, cQ = m / 2;
    // odd rows: D_k ← D_k⁻¹
    NK_TRY(bcr_invert(S, L.D + bb, 2 * bb, b, b, m / 2, 0));
    // next level starts from the even rows' diagonal blocks and zero couplings
    NK_LAUNCH(ctx, k_bcr_copy_even, dim3((unsigned)((bb * m2 + NK_BLOCK - 1) / NK_BLOCK)), dim3(NK_BLOCK), bb, (int64_t)m2,
              (const double *)L.D, N.D);
    NK_HIP(hipMemsetAsync(N.A, 0, (size_t)(m2 * bb) * sizeof(double), ctx->stream));
    NK_HIP(hipMemsetAsync(N.C, 0, (size_t)(m2 * bb) * sizeof(double), ctx->stream));
    // Level 0 of a banded matrix: A_i is upper and C_i lower triangular — the products skip the K slabs that are all zero.
    const bool l0 = (l == 0);
    // P_i = A_{2i} D⁻¹_{2i−1} (i = 1 … cP);  D'_i −= P_i C_{2i−1};  A'_i = −P_i A_{2i−1}
    NK_TRY(bcr_gemm(S, cP, b, b, b, 1.0, L.A + 2 * bb, 2 * bb, b, L.D + bb, 2 * bb, b, 0.0, nullptr, 0, 0, L.P + bb, bb, b, l0 ? 1 : 0));
    NK_TRY(bcr_gemm(S, cP, b, b, b, -1.0, L.P + bb, bb, b, L.C + bb, 2 * bb, b, 1.0, N.D + bb, bb, b, N.D + bb, bb, b, l0 ? 2 : 0));
    NK_TRY(bcr_gemm(S, cP, b, b, b, -1.0, L.P + bb, bb, b, L.A + bb, 2 * bb, b, 0.0, nullptr, 0, 0, N.A + bb, bb, b, l0 ? 3 : 0));
    // Q_i = C_{2i} D⁻¹_{2i+1} (i = 0 … cQ − 1);  D'_i −= Q_i A_{2i+1};  C'_i = −Q_i C_{2i+1}
    NK_TRY(bcr_gemm(S, cQ, b, b, b, 1.0, L.C, 2 * bb, b, L.D + bb, 2 * bb, b, 0.0, nullptr, 0, 0, L.Q, bb, b, l0 ? 4 : 0));
    NK_TRY(bcr_gemm(S, cQ, b, b, b, -1.0, L.Q, bb, b, L.A + bb, 2 * bb, b, 1.0, N.D, bb, b, N.D, bb, b, l0 ? 3 : 0));
    NK_TRY(bcr_gemm(S, cQ, b, b, b, -1.0, L.Q, bb, b, L.C + bb, 2 * bb, b, 0.0, nullptr, 0, 0, N.C, bb, b, l0 ? 2 : 0));
  }
  NK_HIP(hipGetLastError());
  int h = 0;
  NK_HIP(hipMemcpyAsync(&h, S->d_fail, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  NK_HIP(hipStreamSynchronize(ctx->stream));
  *ok = (h == 0);
  return NK_OK;
}

// x = A⁻¹ b (device vectors of length n; may alias)
int nk_bcr_solve(nk_bcr *S, const double *d_b, double *d_x) {
  nk_ctx *ctx = S->ctx;
  const int b = S->b;
  bcr_level &L0 = S->lv[0];
  const int64_t padded = (int64_t)L0.m * b;
  if (padded > S->n) NK_HIP(hipMemsetAsync(L0.f + S->n, 0, (size_t)(padded - S->n) * sizeof(double), ctx->stream));
  NK_TRY(nk_blas_copy(ctx, S->n, d_b, L0.f));
  const size_t nl = S->lv.size();
  auto gemv = [&](int mode, int m, int rows, const double *M1, const double *M2, const double *vin, double *out) {
    bcr_gemv_args g;
    g.b = b; g.m = m; g.mode = mode; g.M1 = M1; g.M2 = M2; g.vin = vin; g.out = out;
    if (rows > 0) NK_LAUNCH(ctx, k_bcr_gemv, dim3(rows, b / 32), dim3(256), g);
  };
  for (size_t l = 0; l + 1 < nl; ++l) {
    bcr_level &L = S->lv[l], &N = S->lv[l + 1];
    gemv(0, L.m, N.m, L.P, L.Q, L.f, N.f);
  }
  {
    bcr_level &B = S->lv[nl - 1];
    gemv(3, 1, 1, B.D, nullptr, B.f, B.t);
    NK_TRY(nk_blas_copy(ctx, b, B.t, B.f));
  }
  for (size_t l = nl - 1; l-- > 0;) {
    bcr_level &L = S->lv[l], &N = S->lv[l + 1];
    NK_LAUNCH(ctx, k_bcr_scatter, dim3((unsigned)(((int64_t)N.m * b + NK_BLOCK - 1) / NK_BLOCK)), dim3(NK_BLOCK), b, (int64_t)N.m,
              (const double *)N.f, L.f);
    gemv(1, L.m, L.m / 2, L.A, L.C, L.f, L.t);
    gemv(2, L.m, L.m / 2, L.D, nullptr, L.t, L.f);
  }
  NK_HIP(hipGetLastError());
  return nk_blas_copy(ctx, S->n, L0.f, d_x);
}

int nk_bcr_shape(const nk_bcr *S, int *block, int *levels) {
  if (block) *block = S->b;
  if (levels) *levels = (int)S->lv.size();
  return NK_OK;
}
