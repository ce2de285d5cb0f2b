// Banded LU on the device — the direct `linsolve` of config C2 (`linsolve = nothing` on a concrete sparse J:
// LinearSolve's default sparse factorisation, reused across steps when `reuse_A_if_factorization`
// — lib/NonlinearSolveBase/ext/NonlinearSolveBaseLinearSolveExt.jl:81-86, descent/newton.jl:121-127).
//
// Blocked right-looking LU without pivoting on LAPACK-style band storage AB[(ku + i − j) + j·ldab]
// (ldab = kl + ku + 1, fill-in stays inside the band). Block size NB = 32:
//   k_band_panel   one workgroup: the (NB + kl) × NB panel is factored in LDS, and the two NB×NB triangular
//                  inverses (L11⁻¹, U11⁻¹) are formed so that the later sweeps are matrix–vector products
//   k_band_update  one workgroup per 8 columns: U12 = L11⁻¹ A12, A22 −= L21 U12 (columns are independent)
//   k_band_solve   one persistent workgroup: forward and backward block sweeps
// No pivoting: valid for the diagonally dominant / SPD-like Jacobians of the grid problems; the driver
// verifies ‖J x − b‖ after the solve and reports the linear solve as failed otherwise (then the nonlinear
// driver follows the reference's failure path, lib/NonlinearSolveFirstOrder/src/solve.jl:367-382).
// FP64 work: 2 n kl ku flops (8.6 GFLOP at n = 65 536, kl = ku = 256) — the run time is launch-latency bound
// (2 launches per block column), not MFMA bound.
#include <math.h>

#include <algorithm>

#include "nk_internal.h"

__host__ __device__ static inline int64_t imin64(int64_t a, int64_t b) { return a < b ? a : b; }
constexpr int NB = 32;
constexpr int UPD_COLS = 8;

__device__ __forceinline__ bool in_band(int64_t i, int64_t j, int kl, int ku) { return (j - i) <= ku && (i - j) <= kl; }
__device__ __forceinline__ size_t bidx(int64_t i, int64_t j, int ku, int ldab) { return (size_t)(ku + i - j) + (size_t)j * ldab; }

__global__ __launch_bounds__(NK_BLOCK) void k_band_fill(int64_t nrows, const int32_t *__restrict__ rowptr,
                                                        const int32_t *__restrict__ col, const double *__restrict__ val,
                                                        double *__restrict__ AB, int ku, int ldab) {
  const int64_t r = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x;
  if (r >= nrows) return;
  for (int32_t p = rowptr[r]; p < rowptr[r + 1]; ++p) AB[bidx(r, col[p], ku, ldab)] = val[p];
}

// panel of block column J: rows [j0, j0+NB+kl) × cols [j0, j0+NB), factored in LDS
__global__ __launch_bounds__(1024) void k_band_panel(int64_t n, int kl, int ku, int ldab, double *__restrict__ AB,
                                                     int J, double *__restrict__ invL, double *__restrict__ invU,
                                                     int *__restrict__ fail) {
  extern __shared__ __attribute__((aligned(16))) double sp[];  // (NB+kl) × NB column-major, ld = NB+kl
  const int64_t j0 = (int64_t)J * NB;
  const int nc = (int)imin64(NB, n - j0);
  const int nr = (int)imin64(NB + kl, n - j0);
  const int ld = NB + kl;
  const int t = threadIdx.x, T = blockDim.x;
  for (int e = t; e < nr * nc; e += T) {
    const int c = e / nr, r = e - c * nr;
    const int64_t i = j0 + r, j = j0 + c;
    sp[c * ld + r] = in_band(i, j, kl, ku) ? AB[bidx(i, j, ku, ldab)] : 0.0;
  }
  __syncthreads();
  for (int c = 0; c < nc; ++c) {
    const double piv = sp[c * ld + c];
    if (t == 0 && (piv == 0.0 || !(piv == piv) || isinf(piv))) *fail = 1;
    const double ip = 1.0 / piv;
    for (int r = c + 1 + t; r < nr; r += T) sp[c * ld + r] *= ip;
    __syncthreads();
    const int rem = nc - c - 1, rows = nr - c - 1;
    for (int e = t; e < rem * rows; e += T) {
      const int cc = c + 1 + e / rows, r = c + 1 + (e - (e / rows) * rows);
      sp[cc * ld + r] -= sp[c * ld + r] * sp[cc * ld + c];
    }
    __syncthreads();
  }
  // write the factored panel back
  for (int e = t; e < nr * nc; e += T) {
    const int c = e / nr, r = e - c * nr;
    const int64_t i = j0 + r, j = j0 + c;
    if (in_band(i, j, kl, ku)) AB[bidx(i, j, ku, ldab)] = sp[c * ld + r];
  }
  // Triangular inverses of the diagonal block, formed cooperatively in LDS (row-major NB×NB, zero padded):
  //   XL = L11⁻¹ by forward elimination  (for c: rows r>c: XL[r,:] −= L[r,c]·XL[c,:])
  //   XU = U11⁻¹ by backward elimination (for c = nc−1…0: XU[c,:] /= U[c,c]; rows r<c: XU[r,:] −= U[r,c]·XU[c,:])
  // threads 0..1023: one (row, col) element of each matrix per thread; NB sequential steps, one barrier each.
  __shared__ double XL[NB * NB], XU[NB * NB];
  const int er = t / NB, ec = t % NB;  // this thread's element (T == NB*NB)
  XL[t] = (er == ec) ? 1.0 : 0.0;
  XU[t] = (er == ec) ? 1.0 : 0.0;
  __syncthreads();
  for (int c = 0; c < nc; ++c) {
    const int cu = nc - 1 - c;
    // L: eliminate column c below the diagonal
    const double lrc = (er > c && er < nc) ? sp[c * ld + er] : 0.0;
    const double xlc = XL[c * NB + ec];
    // U: scale row cu, eliminate above
    const double ucc = sp[cu * ld + cu];
    const double xuc = XU[cu * NB + ec] / ucc;
    const double urc = (er < cu) ? sp[cu * ld + er] : 0.0;
    __syncthreads();
    if (er > c && er < nc) XL[t] -= lrc * xlc;
    if (er == cu) XU[t] = xuc;
    else if (er < cu) XU[t] -= urc * xuc;
    __syncthreads();
  }
  // rows/cols ≥ nc of the last (partial) block: identity rows were never touched; zero them so that padded
  // entries cannot leak into the sweeps
  double vl = XL[t], vu = XU[t];
  if (er >= nc || ec >= nc) { vl = 0.0; vu = 0.0; }
  double *iL = invL + (size_t)J * NB * NB, *iU = invU + (size_t)J * NB * NB;
  iL[ec * NB + er] = vl;  // stored column-major: element (row er, col ec)
  iU[ec * NB + er] = vu;
}

// trailing update for block column J: this workgroup owns UPD_COLS columns right of the panel.
// L21 (kl × NB) is staged once in dynamic LDS (zero outside the band) so the rank-NB update reads no global
// memory in its inner loop.
__global__ __launch_bounds__(NK_BLOCK) void k_band_update(int64_t n, int kl, int ku, int ldab, double *__restrict__ AB,
                                                          int J, const double *__restrict__ invL) {
  extern __shared__ __attribute__((aligned(16))) double sL21[];  // kl × NB, column-major (ld = kl)
  __shared__ double sU[NB * UPD_COLS];   // U12 chunk (NB × UPD_COLS)
  __shared__ double sA[NB * UPD_COLS];   // A12 chunk
  __shared__ double sL[NB * NB];
  const int64_t j0 = (int64_t)J * NB;
  const int64_t c0 = j0 + NB + (int64_t)blockIdx.x * UPD_COLS;  // first column of this chunk
  if (c0 >= n) return;
  const int t = threadIdx.x;
  const double *iL = invL + (size_t)J * NB * NB;
  for (int e = t; e < NB * NB; e += NK_BLOCK) sL[e] = iL[e];
  for (int e = t; e < NB * UPD_COLS; e += NK_BLOCK) {
    const int c = e / NB, r = e - c * NB;
    const int64_t i = j0 + r, j = c0 + c;
    sA[e] = (j < n && i < n && in_band(i, j, kl, ku)) ? AB[bidx(i, j, ku, ldab)] : 0.0;
  }
  for (int e = t; e < kl * NB; e += NK_BLOCK) {
    const int q = e / kl, r = e - q * kl;
    const int64_t i = j0 + NB + r, jq = j0 + q;
    sL21[e] = (i < n && jq < n && in_band(i, jq, kl, ku)) ? AB[bidx(i, jq, ku, ldab)] : 0.0;
  }
  __syncthreads();
  // U12 = L11⁻¹ A12
  for (int e = t; e < NB * UPD_COLS; e += NK_BLOCK) {
    const int c = e / NB, r = e - c * NB;
    double s = 0.0;
    for (int q = 0; q <= r; ++q) s += sL[q * NB + r] * sA[c * NB + q];
    sU[e] = s;
  }
  __syncthreads();
  for (int e = t; e < NB * UPD_COLS; e += NK_BLOCK) {
    const int c = e / NB, r = e - c * NB;
    const int64_t i = j0 + r, j = c0 + c;
    if (j < n && i < n && in_band(i, j, kl, ku)) AB[bidx(i, j, ku, ldab)] = sU[e];
  }
  // A22[:, chunk] −= L21 U12, rows j0+NB … j0+NB+kl−1 (consecutive lanes walk down a column: coalesced)
  for (int e = t; e < kl * UPD_COLS; e += NK_BLOCK) {
    const int c = e / kl, r = e - c * kl;
    const int64_t i = j0 + NB + r, j = c0 + c;
    if (i >= n || j >= n || !in_band(i, j, kl, ku)) continue;
    double s = 0.0;
#pragma unroll
    for (int q = 0; q < NB; ++q) s += sL21[q * kl + r] * sU[c * NB + q];
    AB[bidx(i, j, ku, ldab)] -= s;
  }
}

// x ← U⁻¹ L⁻¹ x, one persistent workgroup (block sweeps are inherently sequential)
__global__ __launch_bounds__(1024) void k_band_solve(int64_t n, int kl, int ku, int ldab, const double *__restrict__ AB,
                                                     int nblk, const double *__restrict__ invL,
                                                     const double *__restrict__ invU, double *__restrict__ x) {
  __shared__ double sy[NB], sb[NB];
  const int t = threadIdx.x, T = blockDim.x;
  // forward: y_J = L11⁻¹ b_J ; b[below] −= L21 y_J
  for (int J = 0; J < nblk; ++J) {
    const int64_t j0 = (int64_t)J * NB;
    const int nc = (int)imin64(NB, n - j0);
    if (t < NB) sb[t] = (t < nc) ? x[j0 + t] : 0.0;
    __syncthreads();
    if (t < NB) {
      const double *iL = invL + (size_t)J * NB * NB;
      double s = 0.0;
      for (int q = 0; q <= t; ++q) s += iL[q * NB + t] * sb[q];
      sy[t] = s;
      if (t < nc) x[j0 + t] = s;
    }
    __syncthreads();
    {  // b[below] −= L21 y_J : 4 lanes per row (kl ≤ 256 rows per pass), 8 columns each, shuffle-reduced
      for (int rb = 0; rb < kl; rb += T / 4) {
        const int r = rb + (t >> 2), sub = t & 3;
        const int64_t i = j0 + NB + r;
        double s = 0.0;
        if (r < kl && i < n) {
#pragma unroll
          for (int qq = 0; qq < NB / 4; ++qq) {
            const int q = sub * (NB / 4) + qq;
            const bool ok = (q < nc) && in_band(i, j0 + q, kl, ku);
            const double a = AB[ok ? bidx(i, j0 + q, ku, ldab) : 0];
            s += ok ? a * sy[q] : 0.0;
          }
        }
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        if (sub == 0 && r < kl && i < n) x[i] -= s;
      }
    }
    __syncthreads();
  }
  // backward: x_J = U11⁻¹ (y_J − U12 x_after)
  for (int J = nblk - 1; J >= 0; --J) {
    const int64_t j0 = (int64_t)J * NB;
    const int nc = (int)imin64(NB, n - j0);
    // partial sums of U12 x_after: NB rows × (ku) columns, spread over the workgroup then reduced per row
    __shared__ double part[NB][33];
    const int row = t & (NB - 1), lane = t / NB;  // T/NB = 32 lanes per row
    double s = 0.0;
    if (row < nc) {
      const int64_t i = j0 + row;
      for (int64_t j = j0 + NB + lane; j <= i + ku && j < n; j += T / NB) s += AB[bidx(i, j, ku, ldab)] * x[j];
    }
    part[row][lane] = s;
    __syncthreads();
    if (t < NB) {
      double acc = 0.0;
      for (int l = 0; l < T / NB; ++l) acc += part[t][l];
      sb[t] = (t < nc) ? x[j0 + t] - acc : 0.0;
    }
    __syncthreads();
    if (t < NB) {
      const double *iU = invU + (size_t)J * NB * NB;
      double v = 0.0;
      for (int q = t; q < NB; ++q) v += iU[q * NB + t] * sb[q];
      if (t < nc) x[j0 + t] = v;
    }
    __syncthreads();
  }
}

// ----------------------------------------------------------------------------- host side
int nk_bandlu_create(nk_csr *A, nk_bandlu **out) {
  nk_ctx *ctx = A->ctx;
  NK_REQUIRE(ctx->nranks == 1, "the banded direct solver is single-rank");
  int kl = 0, ku = 0;
  for (int64_t r = 0; r < A->nrows; ++r)
    for (int32_t p = A->h_rowptr[r]; p < A->h_rowptr[r + 1]; ++p) {
      const int64_t d = (int64_t)A->h_col[p] - r;
      if (d > ku) ku = (int)d;
      if (-d > kl) kl = (int)(-d);
    }
  const int64_t n = A->nrows;
  const size_t band_bytes = (size_t)(kl + ku + 1) * n * sizeof(double);
  NK_REQUIRE(band_bytes < ((size_t)64 << 30), "band storage of %zu bytes is too large (bandwidth %d+%d)", band_bytes, kl, ku);
  NK_REQUIRE((size_t)(NB + kl) * NB * sizeof(double) <= 136 * 1024,
             "lower bandwidth %d too large for the LDS panel", kl);
  nk_bandlu *B = new nk_bandlu();
  B->ctx = ctx;
  B->n = n;
  B->kl = kl;
  B->ku = ku;
  B->ldab = kl + ku + 1;
  B->nblk = (int)((n + NB - 1) / NB);
  NK_TRY(nk_dev_alloc(&B->AB, (size_t)B->ldab * n));
  NK_TRY(nk_dev_alloc(&B->invL, (size_t)B->nblk * NB * NB));
  NK_TRY(nk_dev_alloc(&B->invU, (size_t)B->nblk * NB * NB));
  NK_TRY(nk_dev_alloc(&B->tmp, (size_t)n + 1));
  NK_TRY(nk_dev_alloc(&B->d_fail, (size_t)1));
  static bool attr_set = false;
  if (!attr_set) {
    NK_HIP(hipFuncSetAttribute((const void *)k_band_panel, hipFuncAttributeMaxDynamicSharedMemorySize, 136 * 1024));
    NK_HIP(hipFuncSetAttribute((const void *)k_band_update, hipFuncAttributeMaxDynamicSharedMemorySize, 136 * 1024));
    attr_set = true;
  }
  *out = B;
  return NK_OK;
}

void nk_bandlu_destroy(nk_bandlu *B) {
  if (!B) return;
  hipFree(B->AB); hipFree(B->invL); hipFree(B->invU); hipFree(B->tmp); hipFree(B->d_fail);
  delete B;
}

// copy the CSR values into the band and factor; *ok = 0 when a pivot broke down
int nk_bandlu_factor(nk_bandlu *B, nk_csr *A, int *ok) {
  nk_ctx *ctx = B->ctx;
  const int64_t n = B->n;
  NK_HIP(hipMemsetAsync(B->AB, 0, (size_t)B->ldab * n * sizeof(double), ctx->stream));
  NK_HIP(hipMemsetAsync(B->d_fail, 0, sizeof(int), ctx->stream));
  NK_LAUNCH(ctx, k_band_fill, dim3((unsigned)((n + NK_BLOCK - 1) / NK_BLOCK)), dim3(NK_BLOCK), n, A->d_rowptr, A->d_col,
            A->d_val, B->AB, B->ku, B->ldab);
  const size_t lds = (size_t)(NB + B->kl) * NB * sizeof(double);
  for (int J = 0; J < B->nblk; ++J) {
    hipLaunchKernelGGL(k_band_panel, dim3(1), dim3(1024), lds, ctx->stream, n, B->kl, B->ku, B->ldab, B->AB, J, B->invL,
                       B->invU, B->d_fail);
    const int64_t right = imin64(B->ku, n - ((int64_t)J * NB + NB));
    if (right > 0) {
      const int grid = (int)((right + UPD_COLS - 1) / UPD_COLS);
      hipLaunchKernelGGL(k_band_update, dim3(grid), dim3(NK_BLOCK), (size_t)B->kl * NB * sizeof(double), ctx->stream, n,
                         B->kl, B->ku, B->ldab, B->AB, J, (const double *)B->invL);
    }
  }
  NK_HIP(hipGetLastError());
  int h = 0;
  NK_HIP(hipMemcpyAsync(&h, B->d_fail, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  NK_HIP(hipStreamSynchronize(ctx->stream));
  *ok = (h == 0);
  return NK_OK;
}

// x = A⁻¹ b (device vectors; b and x may alias)
int nk_bandlu_solve(nk_bandlu *B, const double *d_b, double *d_x) {
  nk_ctx *ctx = B->ctx;
  NK_TRY(nk_blas_copy(ctx, B->n, d_b, d_x));
  hipLaunchKernelGGL(k_band_solve, dim3(1), dim3(1024), 0, ctx->stream, B->n, B->kl, B->ku, B->ldab,
                     (const double *)B->AB, B->nblk, (const double *)B->invL, (const double *)B->invU, d_x);
  NK_HIP(hipGetLastError());
  return NK_OK;
}

// ----------------------------------------------------------------------------- exported factorisation seam
// (what LinearSolve's `LUFactorization`/`KLUFactorization` cache stands for: factor once, solve many)
extern "C" int nk_lu_create(nk_csr *A, nk_bandlu **out) {
  NK_REQUIRE(A && out, "NULL argument");
  NK_HIP(hipSetDevice(A->ctx->device));
  return nk_bandlu_create(A, out);
}
extern "C" int nk_lu_destroy(nk_bandlu *B) {
  nk_bandlu_destroy(B);
  return NK_OK;
}
extern "C" int nk_lu_factor(nk_bandlu *B, nk_csr *A, int *ok) {
  NK_REQUIRE(B && A && ok, "NULL argument");
  NK_REQUIRE(A->nrows == B->n, "matrix size changed");
  NK_HIP(hipSetDevice(B->ctx->device));
  return nk_bandlu_factor(B, A, ok);
}
extern "C" int nk_lu_solve(nk_bandlu *B, const double *b, double *x, int memspace) {
  NK_REQUIRE(B && b && x, "NULL argument");
  NK_HIP(hipSetDevice(B->ctx->device));
  nk_ctx *ctx = B->ctx;
  if (memspace == NK_DEVICE) return nk_bandlu_solve(B, b, x);
  NK_HIP(hipMemcpyAsync(B->tmp, b, B->n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  NK_TRY(nk_bandlu_solve(B, B->tmp, B->tmp));
  NK_HIP(hipMemcpyAsync(x, B->tmp, B->n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  NK_HIP(hipStreamSynchronize(ctx->stream));
  return NK_OK;
}
extern "C" int nk_lu_info(nk_bandlu *B, int *kl, int *ku, int64_t *band_bytes) {
  NK_REQUIRE(B, "NULL argument");
  if (kl) *kl = B->kl;
  if (ku) *ku = B->ku;
  if (band_bytes) *band_bytes = (int64_t)B->ldab * B->n * 8;
  return NK_OK;
}
