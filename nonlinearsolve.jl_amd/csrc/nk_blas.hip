// Krylov BLAS-1 kernels for gfx950: wavefront-shuffle (64-lane) reductions, 16-byte vector loads,
// two-stage fixed-order (bitwise reproducible) block→grid reductions, fused Gram–Schmidt passes.
//
// Every kernel is HBM-bound streaming work; algorithmic bytes per launch:
//   multidot   : 8 n (nv + 1)            multiaxpy : 8 n (nv + 2)      scale_to : 16 n
//   dot        : 16 n    sumsq/norm_inf : 8 n    axpby : 24 n    copy : 16 n
#include "nk_internal.h"

#define SKIP_GUARD(d_skip) \
  if ((d_skip) != nullptr && *(d_skip) != 0) return;

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double nanmax(double a, double b) {
  return (a != a || b != b) ? __builtin_nan("") : (a > b ? a : b);
}
__device__ __forceinline__ double wave_nanmax(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = nanmax(v, __shfl_xor(v, o, 64));
  return v;
}

// block-wide sum of one value per thread; result valid in thread 0. 256 threads = 4 waves.
__device__ __forceinline__ double block_sum(double v, double *sm /*[4]*/) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sm[wid] = v;
  __syncthreads();
  return sm[0] + sm[1] + sm[2] + sm[3];
}

// ----------------------------------------------------------------------------- stage-2 reducers
// block s reduces partials[s*nblk .. s*nblk+nblk) in a fixed order
__global__ __launch_bounds__(NK_BLOCK) void k_reduce_sum(const double *__restrict__ partials, int nblk,
                                                         double *__restrict__ out, const int *d_skip) {
  SKIP_GUARD(d_skip);
  __shared__ double sm[4];
  const double *p = partials + (size_t)blockIdx.x * nblk;
  double v = 0.0;
  for (int i = threadIdx.x; i < nblk; i += NK_BLOCK) v += p[i];
  v = block_sum(v, sm);
  if (threadIdx.x == 0) out[blockIdx.x] = v;
}
__global__ __launch_bounds__(NK_BLOCK) void k_reduce_nanmax(const double *__restrict__ partials, int nblk,
                                                            double *__restrict__ out, double sign) {
  __shared__ double sm[4];
  const double *p = partials + (size_t)blockIdx.x * nblk;
  double v = -__builtin_inf();
  for (int i = threadIdx.x; i < nblk; i += NK_BLOCK) v = nanmax(v, p[i]);
  v = wave_nanmax(v);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = sign * nanmax(nanmax(sm[0], sm[1]), nanmax(sm[2], sm[3]));
}

// ----------------------------------------------------------------------------- multidot
// partial[(slot)*gridDim.x + blk] = Σ_{i in blk's stripes} V[:,jbase+slot][i] * w[i], slot < nvc
// optional self slot (w·w) written to slot index `self_slot`.
template <int NV>
__global__ __launch_bounds__(NK_BLOCK) void k_multidot(int64_t n, const double *__restrict__ V, int64_t ldv,
                                                       int jbase, int nvc, const double *__restrict__ w,
                                                       double *__restrict__ partials, int self_slot,
                                                       const int *d_skip) {
  SKIP_GUARD(d_skip);
  __shared__ double sm[4];
  double acc[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) acc[j] = 0.0;
  double self = 0.0;
  const int64_t npair = n >> 1;
  const int64_t stride = (int64_t)gridDim.x * NK_BLOCK;
  const double2 *w2 = reinterpret_cast<const double2 *>(w);
  for (int64_t i = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x; i < npair; i += stride) {
    const double2 wv = w2[i];
    if (self_slot >= 0) self += wv.x * wv.x + wv.y * wv.y;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      if (j < nvc) {
        const double2 vv = reinterpret_cast<const double2 *>(V + (size_t)(jbase + j) * ldv)[i];
        acc[j] += wv.x * vv.x + wv.y * vv.y;
      }
    }
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {  // odd tail
    const double wv = w[n - 1];
    if (self_slot >= 0) self += wv * wv;
#pragma unroll
    for (int j = 0; j < NV; ++j)
      if (j < nvc) acc[j] += wv * V[(size_t)(jbase + j) * ldv + n - 1];
  }
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    if (j < nvc) {
      const double s = block_sum(acc[j], sm);
      if (threadIdx.x == 0) partials[(size_t)(jbase + j) * gridDim.x + blockIdx.x] = s;
    }
  }
  if (self_slot >= 0) {
    const double s = block_sum(self, sm);
    if (threadIdx.x == 0) partials[(size_t)self_slot * gridDim.x + blockIdx.x] = s;
  }
}

int nk_blas_multidot(nk_ctx *ctx, int64_t n, int nv, const double *V, int64_t ldv, const double *w,
                     double *d_h, bool with_self, const int *d_skip) {
  NK_REQUIRE(nv >= 0 && nv <= NK_MAX_NV, "multidot: nv=%d out of range", nv);
  const int grid = nk_grid_for(n >> 1, NK_BLOCK * 2, NK_MAX_RED_BLOCKS);
  const int self_slot = with_self ? nv : -1;
  constexpr int CH = 16;
  int j = 0;
  bool self_done = !with_self;
  {
  nk_prof_scope prof_(ctx, NK_K_MULTIDOT, 8.0 * (double)n * (nv + 1));
  while (j < nv || !self_done) {
    const int nvc = (nv - j) < CH ? (nv - j) : CH;
    const int ss = self_done ? -1 : self_slot;
    if (nvc > 8)
      hipLaunchKernelGGL(k_multidot<16>, dim3(grid), dim3(NK_BLOCK), 0, ctx->stream, n, V, ldv, j, nvc, w,
                         ctx->d_partials, ss, d_skip);
    else if (nvc > 4)
      hipLaunchKernelGGL(k_multidot<8>, dim3(grid), dim3(NK_BLOCK), 0, ctx->stream, n, V, ldv, j, nvc, w,
                         ctx->d_partials, ss, d_skip);
    else if (nvc > 1)
      hipLaunchKernelGGL(k_multidot<4>, dim3(grid), dim3(NK_BLOCK), 0, ctx->stream, n, V, ldv, j, nvc, w,
                         ctx->d_partials, ss, d_skip);
    else
      hipLaunchKernelGGL(k_multidot<1>, dim3(grid), dim3(NK_BLOCK), 0, ctx->stream, n, V, ldv, j, nvc, w,
                         ctx->d_partials, ss, d_skip);
    self_done = true;
    j += nvc;
    if (nvc == 0) break;
  }
  }
  const int nslots = nv + (with_self ? 1 : 0);
  if (nslots == 0) return NK_OK;
  {
    nk_prof_scope prof_(ctx, NK_K_REDUCE_SMALL, 8.0 * nslots * grid);
    hipLaunchKernelGGL(k_reduce_sum, dim3(nslots), dim3(NK_BLOCK), 0, ctx->stream, ctx->d_partials, grid, d_h,
                       d_skip);
  }
  NK_HIP(hipGetLastError());
  return nk_comm_allreduce(ctx, d_h, nslots, 0);
}

// ----------------------------------------------------------------------------- multiaxpy
// w += sign * Σ_j h[j] V[:,j]; optionally partial Σ w_new² per block (slot 0 of partials)
__global__ __launch_bounds__(NK_BLOCK) void k_multiaxpy(int64_t n, const double *__restrict__ V, int64_t ldv,
                                                        int nv, const int *__restrict__ d_nv,
                                                        const double *__restrict__ h, double sign,
                                                        double *__restrict__ w, double *__restrict__ partials,
                                                        const int *d_skip) {
  SKIP_GUARD(d_skip);
  __shared__ double sm[4];
  if (d_nv) nv = *d_nv;
  const int64_t npair = n >> 1;
  const int64_t stride = (int64_t)gridDim.x * NK_BLOCK;
  double2 *w2 = reinterpret_cast<double2 *>(w);
  double ss = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x; i < npair; i += stride) {
    double2 a = w2[i];
    int j = 0;
    for (; j + 4 <= nv; j += 4) {
      const double2 v0 = reinterpret_cast<const double2 *>(V + (size_t)(j + 0) * ldv)[i];
      const double2 v1 = reinterpret_cast<const double2 *>(V + (size_t)(j + 1) * ldv)[i];
      const double2 v2 = reinterpret_cast<const double2 *>(V + (size_t)(j + 2) * ldv)[i];
      const double2 v3 = reinterpret_cast<const double2 *>(V + (size_t)(j + 3) * ldv)[i];
      const double c0 = sign * h[j], c1 = sign * h[j + 1], c2 = sign * h[j + 2], c3 = sign * h[j + 3];
      a.x += c0 * v0.x; a.y += c0 * v0.y;
      a.x += c1 * v1.x; a.y += c1 * v1.y;
      a.x += c2 * v2.x; a.y += c2 * v2.y;
      a.x += c3 * v3.x; a.y += c3 * v3.y;
    }
    for (; j < nv; ++j) {
      const double2 v0 = reinterpret_cast<const double2 *>(V + (size_t)j * ldv)[i];
      const double c0 = sign * h[j];
      a.x += c0 * v0.x; a.y += c0 * v0.y;
    }
    w2[i] = a;
    ss += a.x * a.x + a.y * a.y;
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    double a = w[n - 1];
    for (int j = 0; j < nv; ++j) a += sign * h[j] * V[(size_t)j * ldv + n - 1];
    w[n - 1] = a;
    ss += a * a;
  }
  if (partials) {
    const double s = block_sum(ss, sm);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
  }
}

int nk_blas_multiaxpy(nk_ctx *ctx, int64_t n, int nv, const double *V, int64_t ldv, const double *d_h,
                      double sign, double *w, double *d_sumsq, const int *d_skip, const int *d_nv) {
  const int grid = nk_grid_for(n >> 1, NK_BLOCK * 2, NK_MAX_RED_BLOCKS);
  {
    nk_prof_scope prof_(ctx, NK_K_MULTIAXPY, 8.0 * (double)n * (nv + 2));
    hipLaunchKernelGGL(k_multiaxpy, dim3(grid), dim3(NK_BLOCK), 0, ctx->stream, n, V, ldv, nv, d_nv, d_h, sign,
                       w, d_sumsq ? ctx->d_partials : nullptr, d_skip);
  }
  if (d_sumsq) {
    nk_prof_scope prof_(ctx, NK_K_REDUCE_SMALL, 8.0 * grid);
    hipLaunchKernelGGL(k_reduce_sum, dim3(1), dim3(NK_BLOCK), 0, ctx->stream, ctx->d_partials, grid, d_sumsq,
                       d_skip);
    NK_HIP(hipGetLastError());
    return nk_comm_allreduce(ctx, d_sumsq, 1, 0);
  }
  NK_HIP(hipGetLastError());
  return NK_OK;
}

// ----------------------------------------------------------------------------- dot / sumsq / norm_inf / minmax
__global__ __launch_bounds__(NK_BLOCK) void k_dot(int64_t n, const double *__restrict__ x,
                                                  const double *__restrict__ y, double *__restrict__ partials) {
  __shared__ double sm[4];
  double s = 0.0;
  const int64_t npair = n >> 1, stride = (int64_t)gridDim.x * NK_BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x; i < npair; i += stride) {
    const double2 a = reinterpret_cast<const double2 *>(x)[i], b = reinterpret_cast<const double2 *>(y)[i];
    s += a.x * b.x + a.y * b.y;
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) s += x[n - 1] * y[n - 1];
  s = block_sum(s, sm);
  if (threadIdx.x == 0) partials[blockIdx.x] = s;
}
__global__ __launch_bounds__(NK_BLOCK) void k_absmax(int64_t n, const double *__restrict__ x,
                                                     double *__restrict__ partials) {
  __shared__ double sm[4];
  double m = 0.0;
  const int64_t stride = (int64_t)gridDim.x * NK_BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x; i < n; i += stride) m = nanmax(m, fabs(x[i]));
  m = wave_nanmax(m);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = nanmax(nanmax(sm[0], sm[1]), nanmax(sm[2], sm[3]));
}
// slot 0: max(x), slot 1: max(-x)
__global__ __launch_bounds__(NK_BLOCK) void k_minmax(int64_t n, const double *__restrict__ x,
                                                     double *__restrict__ partials) {
  __shared__ double sm[8];
  double mx = -__builtin_inf(), mn = -__builtin_inf();
  const int64_t stride = (int64_t)gridDim.x * NK_BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x; i < n; i += stride) {
    mx = nanmax(mx, x[i]);
    mn = nanmax(mn, -x[i]);
  }
  mx = wave_nanmax(mx);
  mn = wave_nanmax(mn);
  if ((threadIdx.x & 63) == 0) { sm[threadIdx.x >> 6] = mx; sm[4 + (threadIdx.x >> 6)] = mn; }
  __syncthreads();
  if (threadIdx.x == 0) {
    partials[blockIdx.x] = nanmax(nanmax(sm[0], sm[1]), nanmax(sm[2], sm[3]));
    partials[gridDim.x + blockIdx.x] = nanmax(nanmax(sm[4], sm[5]), nanmax(sm[6], sm[7]));
  }
}

int nk_blas_dot(nk_ctx *ctx, int64_t n, const double *x, const double *y, double *d_out) {
  const int grid = nk_grid_for(n >> 1, NK_BLOCK * 2, NK_MAX_RED_BLOCKS);
  hipLaunchKernelGGL(k_dot, dim3(grid), dim3(NK_BLOCK), 0, ctx->stream, n, x, y, ctx->d_partials);
  hipLaunchKernelGGL(k_reduce_sum, dim3(1), dim3(NK_BLOCK), 0, ctx->stream, ctx->d_partials, grid, d_out,
                     (const int *)nullptr);
  NK_HIP(hipGetLastError());
  return nk_comm_allreduce(ctx, d_out, 1, 0);
}
int nk_blas_sumsq(nk_ctx *ctx, int64_t n, const double *x, double *d_out) { return nk_blas_dot(ctx, n, x, x, d_out); }
int nk_blas_norm_inf(nk_ctx *ctx, int64_t n, const double *x, double *d_out) {
  const int grid = nk_grid_for(n, NK_BLOCK * 4, NK_MAX_RED_BLOCKS);
  hipLaunchKernelGGL(k_absmax, dim3(grid), dim3(NK_BLOCK), 0, ctx->stream, n, x, ctx->d_partials);
  hipLaunchKernelGGL(k_reduce_nanmax, dim3(1), dim3(NK_BLOCK), 0, ctx->stream, ctx->d_partials, grid, d_out, 1.0);
  NK_HIP(hipGetLastError());
  return nk_comm_allreduce(ctx, d_out, 1, 1);
}
int nk_blas_minmax(nk_ctx *ctx, int64_t n, const double *x, double *d_out2) {
  const int grid = nk_grid_for(n, NK_BLOCK * 4, NK_MAX_RED_BLOCKS);
  hipLaunchKernelGGL(k_minmax, dim3(grid), dim3(NK_BLOCK), 0, ctx->stream, n, x, ctx->d_partials);
  // out[0] = max(x), out[1] = max(-x); both all-reduced with max, caller negates out[1] → min
  hipLaunchKernelGGL(k_reduce_nanmax, dim3(2), dim3(NK_BLOCK), 0, ctx->stream, ctx->d_partials, grid, d_out2, 1.0);
  NK_HIP(hipGetLastError());
  return nk_comm_allreduce(ctx, d_out2, 2, 1);
}

// ----------------------------------------------------------------------------- elementwise
__global__ __launch_bounds__(NK_BLOCK) void k_axpby(int64_t n, double a, const double *__restrict__ x, double b,
                                                    double *__restrict__ y) {
  const int64_t npair = n >> 1, stride = (int64_t)gridDim.x * NK_BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x; i < npair; i += stride) {
    const double2 xv = reinterpret_cast<const double2 *>(x)[i];
    double2 yv = reinterpret_cast<double2 *>(y)[i];
    yv.x = a * xv.x + b * yv.x;
    yv.y = a * xv.y + b * yv.y;
    reinterpret_cast<double2 *>(y)[i] = yv;
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) y[n - 1] = a * x[n - 1] + b * y[n - 1];
}
__global__ __launch_bounds__(NK_BLOCK) void k_lincomb(int64_t n, double a, const double *__restrict__ x, double b,
                                                      const double *__restrict__ y, double *__restrict__ z) {
  const int64_t stride = (int64_t)gridDim.x * NK_BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x; i < n; i += stride) z[i] = a * x[i] + b * y[i];
}
__global__ __launch_bounds__(NK_BLOCK) void k_scale_to(int64_t n, const double *__restrict__ d_scale,
                                                       const double *__restrict__ x, double *__restrict__ y,
                                                       const int *d_skip) {
  SKIP_GUARD(d_skip);
  const double s = *d_scale;
  const int64_t npair = n >> 1, stride = (int64_t)gridDim.x * NK_BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x; i < npair; i += stride) {
    double2 v = reinterpret_cast<const double2 *>(x)[i];
    v.x *= s;
    v.y *= s;
    reinterpret_cast<double2 *>(y)[i] = v;
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) y[n - 1] = s * x[n - 1];
}
__global__ __launch_bounds__(NK_BLOCK) void k_fill(int64_t n, double a, double *__restrict__ y) {
  const int64_t stride = (int64_t)gridDim.x * NK_BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x; i < n; i += stride) y[i] = a;
}

static inline int ew_grid(int64_t n) { return nk_grid_for(n >> 1, NK_BLOCK * 2, 4096); }

int nk_blas_axpby(nk_ctx *ctx, int64_t n, double a, const double *x, double b, double *y) {
  hipLaunchKernelGGL(k_axpby, dim3(ew_grid(n)), dim3(NK_BLOCK), 0, ctx->stream, n, a, x, b, y);
  NK_HIP(hipGetLastError());
  return NK_OK;
}
int nk_blas_lincomb(nk_ctx *ctx, int64_t n, double a, const double *x, double b, const double *y, double *z) {
  hipLaunchKernelGGL(k_lincomb, dim3(nk_grid_for(n, NK_BLOCK * 2, 4096)), dim3(NK_BLOCK), 0, ctx->stream, n, a, x, b, y, z);
  NK_HIP(hipGetLastError());
  return NK_OK;
}
int nk_blas_scale_to(nk_ctx *ctx, int64_t n, const double *d_scale, const double *x, double *y, const int *d_skip) {
  nk_prof_scope prof_(ctx, NK_K_SCALE, 16.0 * (double)n);
  hipLaunchKernelGGL(k_scale_to, dim3(ew_grid(n)), dim3(NK_BLOCK), 0, ctx->stream, n, d_scale, x, y, d_skip);
  NK_HIP(hipGetLastError());
  return NK_OK;
}
int nk_blas_copy(nk_ctx *ctx, int64_t n, const double *x, double *y) {
  if (n > 0 && x != y) NK_HIP(hipMemcpyAsync(y, x, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
  return NK_OK;
}
int nk_blas_fill(nk_ctx *ctx, int64_t n, double a, double *y) {
  if (n <= 0) return NK_OK;
  if (a == 0.0) {
    NK_HIP(hipMemsetAsync(y, 0, (size_t)n * sizeof(double), ctx->stream));
    return NK_OK;
  }
  hipLaunchKernelGGL(k_fill, dim3(nk_grid_for(n, NK_BLOCK * 4, 4096)), dim3(NK_BLOCK), 0, ctx->stream, n, a, y);
  NK_HIP(hipGetLastError());
  return NK_OK;
}
int nk_scalars_to_host(nk_ctx *ctx, const double *d_src, int count, double *h_dst) {
  NK_REQUIRE(count <= 4 * NK_MAX_NV, "too many scalars");
  NK_HIP(hipMemcpyAsync(ctx->h_pinned, d_src, count * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  NK_HIP(hipStreamSynchronize(ctx->stream));
  for (int i = 0; i < count; ++i) h_dst[i] = ctx->h_pinned[i];
  return NK_OK;
}

// ----------------------------------------------------------------------------- exported BLAS-1 (device pointers)
extern "C" int nk_dot(nk_ctx *ctx, int64_t n, const double *x, const double *y, double *result) {
  NK_REQUIRE(ctx && x && y && result, "NULL argument");
  NK_TRY(nk_blas_dot(ctx, n, x, y, ctx->d_scal));
  return nk_scalars_to_host(ctx, ctx->d_scal, 1, result);
}
extern "C" int nk_nrm2(nk_ctx *ctx, int64_t n, const double *x, double *result) {
  NK_REQUIRE(ctx && x && result, "NULL argument");
  NK_TRY(nk_blas_dot(ctx, n, x, x, ctx->d_scal));
  NK_TRY(nk_scalars_to_host(ctx, ctx->d_scal, 1, result));
  *result = sqrt(*result);
  return NK_OK;
}
extern "C" int nk_norm_inf(nk_ctx *ctx, int64_t n, const double *x, double *result) {
  NK_REQUIRE(ctx && x && result, "NULL argument");
  NK_TRY(nk_blas_norm_inf(ctx, n, x, ctx->d_scal));
  return nk_scalars_to_host(ctx, ctx->d_scal, 1, result);
}
extern "C" int nk_axpy(nk_ctx *ctx, int64_t n, double a, const double *x, double *y) {
  NK_REQUIRE(ctx && x && y, "NULL argument");
  return nk_blas_axpby(ctx, n, a, x, 1.0, y);
}
extern "C" int nk_multidot(nk_ctx *ctx, int64_t n, int nv, const double *V, int64_t ldv, const double *w,
                           double *h_host) {
  NK_REQUIRE(ctx && V && w && h_host, "NULL argument");
  NK_REQUIRE((ldv & 1) == 0, "ldv must be even (16-byte aligned columns)");
  NK_TRY(nk_blas_multidot(ctx, n, nv, V, ldv, w, ctx->d_scal, false, nullptr));
  return nk_scalars_to_host(ctx, ctx->d_scal, nv, h_host);
}
extern "C" int nk_multiaxpy(nk_ctx *ctx, int64_t n, int nv, const double *V, int64_t ldv, const double *h_host,
                            double *w, double *wnorm2) {
  NK_REQUIRE(ctx && V && w && h_host, "NULL argument");
  NK_REQUIRE(nv <= NK_MAX_NV && (ldv & 1) == 0, "bad nv/ldv");
  for (int j = 0; j < nv; ++j) ctx->h_pinned[j] = h_host[j];
  NK_HIP(hipMemcpyAsync(ctx->d_scal, ctx->h_pinned, nv * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  NK_TRY(nk_blas_multiaxpy(ctx, n, nv, V, ldv, ctx->d_scal, -1.0, w, wnorm2 ? ctx->d_scal + NK_MAX_NV : nullptr,
                           nullptr, nullptr));
  if (wnorm2) return nk_scalars_to_host(ctx, ctx->d_scal + NK_MAX_NV, 1, wnorm2);
  NK_HIP(hipStreamSynchronize(ctx->stream));
  return NK_OK;
}
