// Krylov BLAS-1 kernels for gfx950: wavefront-shuffle (64-lane) reductions, 16-byte vector loads,
// two-stage fixed-order (bitwise reproducible) block→grid reductions, fused Gram–Schmidt passes.
//
// Every kernel is HBM-bound streaming work; algorithmic bytes per launch:
//   multidot   : 8 n (nv + 1)            multiaxpy : 8 n (nv + 2)      scale_to : 16 n
//   dot        : 16 n    sumsq/norm_inf : 8 n    axpby : 24 n    copy : 16 n
#include <algorithm>

#include "nk_internal.h"

// Krylov-basis loads. The basis (m+1 columns × 8N bytes ≈ 260 MB at N = 2²⁰) is streamed and never re-used before it
// has left every cache; NK_NT_BASIS=1 marks those loads non-temporal so that they do not evict the Jacobian
// (60 MB, re-read by every SpMV) from L2 / Infinity Cache. Measured on MI355X (Bratu 1024², fixed-work step): the SpMV
// gains 2 % (16.46 → 16.17 µs) but the basis kernels lose 12–14 % (multidot 26.9 → 30.7 µs), 303 → 274 steps/s — off.
#ifndef NK_NT_BASIS
#define NK_NT_BASIS 0
#endif
typedef double nk_d2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 ldv2(const double *col, int64_t i) {
#if NK_NT_BASIS
  const nk_d2v v = __builtin_nontemporal_load(reinterpret_cast<const nk_d2v *>(col) + i);
  return make_double2(v.x, v.y);
#else
  return reinterpret_cast<const double2 *>(col)[i];
#endif
}

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

// NV per-thread accumulators → NV block sums with ONE barrier: butterfly inside each wavefront, lane 0 parks the
// wave's sums in LDS, then thread j adds the four wave sums (fixed order) and stores slot j's partial.
template <int NV>
__device__ __forceinline__ void block_sum_array_store(const double (&acc)[NV], int nvc, double *sm /*[4*NV]*/,
                                                      double *__restrict__ partials, int slot0) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    if (j < nvc) {
      const double v = wave_sum(acc[j]);
      if (lane == 0) sm[wid * NV + j] = v;
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < nvc) {
    const int t = threadIdx.x;
    partials[(size_t)(slot0 + t) * gridDim.x + blockIdx.x] = (sm[t] + sm[NV + t]) + (sm[2 * NV + t] + sm[3 * NV + t]);
  }
}

// ----------------------------------------------------------------------------- stage-2 reducers
// block s reduces partials[s*nblk .. s*nblk+nblk) in a fixed order
// optional per-slot scale (lagged normalisation of the Krylov basis: h_j = s_j · (ṽ_j·w)) for slots < nscaled
__global__ __launch_bounds__(NK_BLOCK) void k_reduce_sum(const double *__restrict__ partials, int nblk,
                                                         double *__restrict__ out, const int *d_skip,
                                                         const double *__restrict__ scales, int nscaled) {
  const int skip = (d_skip != nullptr) ? *d_skip : 0;  // requested together with the partials (one round trip)
  __shared__ double sm[4];
  const double *p = partials + (size_t)blockIdx.x * nblk;
  double v = 0.0;
  for (int i = threadIdx.x; i < nblk; i += NK_BLOCK) v += p[i];
  if (skip) return;
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    v = (sm[0] + sm[1]) + (sm[2] + sm[3]);
    if (scales != nullptr && (int)blockIdx.x < nscaled) v *= scales[blockIdx.x];
    out[blockIdx.x] = v;
  }
}
// Stage-2 reduction AND all-reduce in one launch (several ranks, peer-mapped arenas — nk_ctx.hip): workgroup s sums slot s
// in the fixed order and stores the sum straight into slot s of EVERY rank's arena; the workgroup that takes the last ticket
// releases this rank's flag everywhere, waits for every rank's flag here and combines the nslots values in rank order into
// `out`. The collective part runs even when the cycle is done (d_skip): every rank must execute every collective.
__global__ __launch_bounds__(NK_BLOCK) void k_reduce_sum_allreduce(const double *__restrict__ partials, int nblk,
                                                                   double *__restrict__ out, nk_peer_ar_view pv) {
  __shared__ double sm[4];
  __shared__ unsigned int s_last;
  const int par = (int)(pv.seq & 1), slot = blockIdx.x, nslots = gridDim.x;
  const double *p = partials + (size_t)slot * nblk;
  double v = 0.0;
  for (int i = threadIdx.x; i < nblk; i += NK_BLOCK) v += p[i];
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    v = (sm[0] + sm[1]) + (sm[2] + sm[3]);
    for (int q = 0; q < pv.P; ++q) reinterpret_cast<nk_peer_hdr *>(pv.map[q])->ar_data[par][pv.me][slot] = v;
    __threadfence_system();
    s_last = (atomicAdd(pv.ticket, 1u) == (unsigned)nslots - 1u) ? 1u : 0u;
  }
  __syncthreads();
  if (!s_last) return;
  // ---- the last workgroup of this rank: every slot of mine has landed everywhere
  const int t = threadIdx.x;
  nk_peer_hdr *mine = reinterpret_cast<nk_peer_hdr *>(pv.map[pv.me]);
  if (t == 0) *pv.ticket = 0u;
  __threadfence_system();
  if (t < pv.P) {
    __hip_atomic_store(&reinterpret_cast<nk_peer_hdr *>(pv.map[t])->ar_flag[par][pv.me], pv.seq, __ATOMIC_RELEASE,
                       __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_load(&mine->ar_flag[par][t], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < pv.seq) {
      if (__hip_atomic_load(&mine->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 4) break;
      if (wall_clock64() - t0 > nk_peer_timeout(&mine->err)) { atomicAdd((unsigned long long *)&mine->err, 1ull); break; }
      __builtin_amdgcn_s_sleep(2);
    }
  }
  __syncthreads();
  for (int e = t; e < nslots; e += NK_BLOCK) {
    double acc = mine->ar_data[par][0][e];
    for (int q = 1; q < pv.P; ++q) acc += mine->ar_data[par][q][e];
    out[e] = acc;
  }
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
// NV is the EXACT number of columns (no per-column predicate: hipcc turns a dynamic `if (j < nv)` around each
// load into load→s_waitcnt→use chains with a single load in flight per wave). NV = 0: only the self product.
template <int NV>
__global__ __launch_bounds__(NK_BLOCK) void k_multidot(int64_t n, const double *__restrict__ V, int64_t ldv,
                                                       int jbase, const double *__restrict__ w,
                                                       double *__restrict__ partials, int self_slot,
                                                       const int *d_skip) {
  SKIP_GUARD(d_skip);
  constexpr int NA = NV > 0 ? NV : 1;
  __shared__ double sm[4 * NA + 4];
  double acc[NA];
#pragma unroll
  for (int j = 0; j < NA; ++j) acc[j] = 0.0;
  double self = 0.0;
  const int64_t npair = n >> 1;
  const int64_t stride = (int64_t)gridDim.x * NK_BLOCK;
  const double2 *w2 = reinterpret_cast<const double2 *>(w);
  for (int64_t i = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x; i < npair; i += stride) {
    const double2 wv = w2[i];
    double2 vv[NA];
#pragma unroll
    for (int j = 0; j < NV; ++j) vv[j] = ldv2(V + (size_t)(jbase + j) * ldv, i);
    self += wv.x * wv.x + wv.y * wv.y;
#pragma unroll
    for (int j = 0; j < NV; ++j) acc[j] += wv.x * vv[j].x + wv.y * vv[j].y;
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {  // odd tail
    const double wv = w[n - 1];
    self += wv * wv;
#pragma unroll
    for (int j = 0; j < NV; ++j) acc[j] += wv * V[(size_t)(jbase + j) * ldv + n - 1];
  }
  if (NV > 0) block_sum_array_store<NA>(acc, NV, sm, partials, jbase);
  if (self_slot >= 0) {
    const double s = block_sum(self, sm + 4 * NA);
    if (threadIdx.x == 0) partials[(size_t)self_slot * gridDim.x + blockIdx.x] = s;
  }
}

#define NK_SWITCH_1_16(nvc, F)                                                                               \
  switch (nvc) {                                                                                             \
    case 1: F(1); break;   case 2: F(2); break;   case 3: F(3); break;   case 4: F(4); break;                \
    case 5: F(5); break;   case 6: F(6); break;   case 7: F(7); break;   case 8: F(8); break;                \
    case 9: F(9); break;   case 10: F(10); break; case 11: F(11); break; case 12: F(12); break;              \
    case 13: F(13); break; case 14: F(14); break; case 15: F(15); break; case 16: F(16); break;              \
    default: break;                                                                                          \
  }

int nk_blas_multidot(nk_ctx *ctx, int64_t n, int nv, const double *V, int64_t ldv, const double *w,
                     double *d_h, bool with_self, const int *d_skip, const double *d_scales) {
  NK_REQUIRE(nv >= 0 && nv <= NK_MAX_NV, "multidot: nv=%d out of range", nv);
  const int grid = nk_grid_for(n >> 1, NK_BLOCK * 4, NK_DOT_BLOCKS);
  const int self_slot = with_self ? nv : -1;
  {
    nk_prof_scope prof_(ctx, NK_K_MULTIDOT, 8.0 * (double)n * (nv + 1));
    // balanced chunks of ≤ 16 columns (e.g. 31 → 16 + 15); the self product rides on the first launch
    const int nchunks = nv > 0 ? (nv + 15) / 16 : 0;
    int j = 0;
    for (int c = 0; c < nchunks; ++c) {
      const int nvc = (nv - j + (nchunks - c) - 1) / (nchunks - c);
      const int ss = (c == 0) ? self_slot : -1;
#define MD_LAUNCH(N)                                                                                         \
  NK_LAUNCH(ctx, k_multidot<N>, dim3(grid), dim3(NK_BLOCK), n, V, ldv, j, w, ctx->d_partials, \
                     ss, d_skip)
      NK_SWITCH_1_16(nvc, MD_LAUNCH)
      j += nvc;
    }
    if (nchunks == 0 && with_self) {
      const int j0 = 0;
      const int ss = self_slot;
      (void)j0;
      NK_LAUNCH(ctx, k_multidot<0>, dim3(grid), dim3(NK_BLOCK), n, V, ldv, 0, w, ctx->d_partials,
                         ss, d_skip);
    }
#undef MD_LAUNCH
  }
  const int nslots = nv + (with_self ? 1 : 0);
  ctx->last_red_grid = grid;
  if (nslots == 0 || d_h == nullptr) {
    NK_HIP(hipGetLastError());
    return NK_OK;
  }
  {
    nk_prof_scope prof_(ctx, NK_K_REDUCE_SMALL, 8.0 * nslots * grid);
    NK_LAUNCH(ctx, k_reduce_sum, dim3(nslots), dim3(NK_BLOCK), ctx->d_partials, grid, d_h,
                       d_skip, d_scales, nv);
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
                                                        const int *d_skip, const double *__restrict__ sc, int overwrite,
                                                        const double *__restrict__ uo, double *__restrict__ un, double usign,
                                                        double *__restrict__ upartials) {
  // (un != nullptr — the fused Newton update — is only launched without a skip flag: the update must happen)
  SKIP_GUARD(d_skip);
  __shared__ double sm[4];
  if (d_nv) nv = *d_nv;
  const int64_t npair = n >> 1;
  const int64_t stride = (int64_t)gridDim.x * NK_BLOCK;
  double2 *w2 = reinterpret_cast<double2 *>(w);
  const double2 *uo2 = reinterpret_cast<const double2 *>(uo);
  double2 *un2 = reinterpret_cast<double2 *>(un);
  double ss = 0.0, us = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x; i < npair; i += stride) {
    double2 a = overwrite ? make_double2(0.0, 0.0) : w2[i];
    int j = 0;
    for (; j + 4 <= nv; j += 4) {
      const double2 v0 = ldv2(V + (size_t)(j + 0) * ldv, i);
      const double2 v1 = ldv2(V + (size_t)(j + 1) * ldv, i);
      const double2 v2 = ldv2(V + (size_t)(j + 2) * ldv, i);
      const double2 v3 = ldv2(V + (size_t)(j + 3) * ldv, i);
      double c0 = sign * h[j], c1 = sign * h[j + 1], c2 = sign * h[j + 2], c3 = sign * h[j + 3];
      if (sc) { c0 *= sc[j]; c1 *= sc[j + 1]; c2 *= sc[j + 2]; c3 *= sc[j + 3]; }
      a.x += c0 * v0.x; a.y += c0 * v0.y;
      a.x += c1 * v1.x; a.y += c1 * v1.y;
      a.x += c2 * v2.x; a.y += c2 * v2.y;
      a.x += c3 * v3.x; a.y += c3 * v3.y;
    }
    for (; j < nv; ++j) {
      const double2 v0 = ldv2(V + (size_t)j * ldv, i);
      const double c0 = sign * h[j] * (sc ? sc[j] : 1.0);
      a.x += c0 * v0.x; a.y += c0 * v0.y;
    }
    w2[i] = a;
    ss += a.x * a.x + a.y * a.y;
    if (un) {   // u_new = u_old + usign·x, ‖u_new − u_old‖² (k_newton_update's arithmetic)
      const double2 u = uo2[i];
      double2 r;
      r.x = u.x + usign * a.x; r.y = u.y + usign * a.y;
      un2[i] = r;
      const double dx = r.x - u.x, dy = r.y - u.y;
      us += dx * dx;
      us += dy * dy;
    }
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    double a = overwrite ? 0.0 : w[n - 1];
    for (int j = 0; j < nv; ++j) a += sign * h[j] * (sc ? sc[j] : 1.0) * V[(size_t)j * ldv + n - 1];
    w[n - 1] = a;
    ss += a * a;
    if (un) {
      const double u = uo[n - 1], r = u + usign * a, d = r - u;
      un[n - 1] = r;
      us += d * d;
    }
  }
  if (partials) {
    const double s = block_sum(ss, sm);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
  }
  if (upartials) {
    if (partials) __syncthreads();
    const double s = block_sum(us, sm);
    if (threadIdx.x == 0) upartials[blockIdx.x] = s;
  }
}

int nk_blas_multiaxpy(nk_ctx *ctx, int64_t n, int nv, const double *V, int64_t ldv, const double *d_h,
                      double sign, double *w, double *d_sumsq, const int *d_skip, const int *d_nv,
                      const double *d_scales, bool overwrite, nk_fused_update *fu) {
  const int grid = nk_grid_for(n >> 1, NK_BLOCK * 2, NK_MAX_RED_BLOCKS);
  NK_REQUIRE(!fu || (d_skip == nullptr && d_sumsq == nullptr), "internal: a fused update rides in an unconditional, plain multiaxpy");
  {
    nk_prof_scope prof_(ctx, NK_K_MULTIAXPY, 8.0 * (double)n * (nv + 2 + (fu ? 2 : 0)));
    NK_LAUNCH(ctx, k_multiaxpy, dim3(grid), dim3(NK_BLOCK), n, V, ldv, nv, d_nv, d_h, sign,
                       w, d_sumsq ? ctx->d_partials_ss : nullptr, d_skip, d_scales, overwrite ? 1 : 0,
                       fu ? fu->u_old : (const double *)nullptr, fu ? fu->u_new : (double *)nullptr, fu ? fu->usign : 0.0,
                       fu ? fu->partials : (double *)nullptr);
  }
  if (fu) { fu->grid = grid; fu->done = true; }
  if (d_sumsq == NK_SUMSQ_PARTIALS_ONLY) {  // consumer (k_givens) reduces ctx->d_partials[0..grid) itself
    ctx->last_red_grid = grid;
    NK_HIP(hipGetLastError());
    return NK_OK;
  }
  if (d_sumsq) {
    nk_prof_scope prof_(ctx, NK_K_REDUCE_SMALL, 8.0 * grid);
    NK_LAUNCH(ctx, k_reduce_sum, dim3(1), dim3(NK_BLOCK), ctx->d_partials_ss, grid, d_sumsq,
                       d_skip, (const double *)nullptr, 0);
    NK_HIP(hipGetLastError());
    return nk_comm_allreduce(ctx, d_sumsq, 1, 0);
  }
  NK_HIP(hipGetLastError());
  return NK_OK;
}

// ----------------------------------------------------------------------------- fused Gram–Schmidt pass
// One pass over the basis does BOTH halves of a CGS2 step that touch V:
//     w ← w − Σ_j (h_j s_j) ṽ_j           (first projection, coefficients from the previous multidot)
//     raw2_j = ṽ_j · w_new  (all j)        (inner products of the re-orthogonalisation)
//     ss = ‖w_new‖²
// The ṽ_j values stay in registers between the two uses, so V is read once instead of twice
// (algorithmic bytes 8 n (nv + 2), the same as the plain multiaxpy). NV ≤ 32 columns per launch.
template <int NV>  // EXACT column count (see k_multidot for why)
__global__ __launch_bounds__(NK_BLOCK) void k_fused_axpy_dot(int64_t n, const double *__restrict__ V, int64_t ldv,
                                                             const double *__restrict__ h,
                                                             const double *__restrict__ sc, double *__restrict__ w,
                                                             double *__restrict__ partials, const int *d_skip) {
  SKIP_GUARD(d_skip);
  __shared__ double sm[4 * NV + 4];
  __shared__ double coef[NV];  // h_j s_j, broadcast-read from LDS so they do not occupy 2·NV VGPRs
  if (threadIdx.x < NV) coef[threadIdx.x] = h[threadIdx.x] * sc[threadIdx.x];
  __syncthreads();
  double acc[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) acc[j] = 0.0;
  double ss = 0.0;
  // one element per lane per sweep (8-byte loads): NV values of ṽ_j stay live between the two uses.
  // 32-bit lane offsets on uniform column bases → scalar-base + vector-offset loads.
  const unsigned stride = gridDim.x * NK_BLOCK, nn = (unsigned)n;
  for (unsigned i = blockIdx.x * NK_BLOCK + threadIdx.x; i < nn; i += stride) {
    double vv[NV];
    double a = w[i];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const double *__restrict__ col = V + (size_t)j * ldv;
      vv[j] = col[i];
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) a -= coef[j] * vv[j];
    w[i] = a;
    ss += a * a;
#pragma unroll
    for (int j = 0; j < NV; ++j) acc[j] += vv[j] * a;
  }
  block_sum_array_store<NV>(acc, NV, sm, partials, 0);
  const double s = block_sum(ss, sm + 4 * NV);
  if (threadIdx.x == 0) partials[(size_t)NV * gridDim.x + blockIdx.x] = s;
}

// DCGS2 pass A (see nk_gmres.hip): NV final columns ṽ_0..ṽ_{NV−1}, the pending column p = V[:,NV] (first projection
// of the previous step, not yet re-orthogonalised) and z = V[:,NV+1] = s·A p. Per element, in one sweep over the basis:
//   p ← p − Σ a_j ṽ_j                (the delayed second Gram–Schmidt correction of the previous step)
//   z ← z − Σ b_j ṽ_j − b_NV p       (= A v_NV by linearity and the Arnoldi relation A V = V H̄)
//   g_j = ṽ_j·z (j < NV), g_NV = p·z  (first projection of the new vector, over the corrected basis)
template <int NV>
__global__ __launch_bounds__(NK_BLOCK) void k_dcgs2_pass_a(int64_t n, double *__restrict__ V, int64_t ldv,
                                                           const double *__restrict__ ca_g, const double *__restrict__ cb_g,
                                                           double *__restrict__ partials, const int *d_skip) {
  SKIP_GUARD(d_skip);
  __shared__ double sm[4 * (NV + 1) + 4];
  __shared__ double ca[NV + 1], cb[NV + 1];
  if (threadIdx.x < NV) ca[threadIdx.x] = ca_g[threadIdx.x];
  if (threadIdx.x <= NV) cb[threadIdx.x] = cb_g[threadIdx.x];
  __syncthreads();
  double acc[NV + 1];
#pragma unroll
  for (int j = 0; j <= NV; ++j) acc[j] = 0.0;
  double *__restrict__ pk = V + (size_t)NV * ldv;
  double *__restrict__ zk = V + (size_t)(NV + 1) * ldv;
  const unsigned stride = gridDim.x * NK_BLOCK, nn = (unsigned)n;
  for (unsigned i = blockIdx.x * NK_BLOCK + threadIdx.x; i < nn; i += stride) {
    double vv[NV];
    double pv = pk[i], zv = zk[i];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const double *__restrict__ col = V + (size_t)j * ldv;
      vv[j] = col[i];
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      pv -= ca[j] * vv[j];
      zv -= cb[j] * vv[j];
    }
    zv -= cb[NV] * pv;
    pk[i] = pv;
    zk[i] = zv;
#pragma unroll
    for (int j = 0; j < NV; ++j) acc[j] += vv[j] * zv;
    acc[NV] += pv * zv;
  }
  block_sum_array_store<NV + 1>(acc, NV + 1, sm, partials, 0);
}

int nk_blas_dcgs2_pass_a(nk_ctx *ctx, int64_t n, int k, double *V, int64_t ldv, const double *d_a, const double *d_b,
                         const double *d_scales, double *d_h, const int *d_skip) {
  NK_REQUIRE(k >= 1 && k <= 31, "DCGS2 pass A handles 1..31 final columns (got %d)", k);
  NK_REQUIRE(n < (1ll << 31), "fused pass: local vector too long for 32-bit offsets");
  const int grid = nk_grid_for(n, NK_BLOCK * 4, NK_DOT_BLOCKS);
  {
    nk_prof_scope prof_(ctx, NK_K_MULTIDOT, 8.0 * (double)n * (k + 4));
#define PA_LAUNCH(N) NK_LAUNCH(ctx, k_dcgs2_pass_a<N>, dim3(grid), dim3(NK_BLOCK), n, V, ldv, d_a, d_b, ctx->d_partials, d_skip)
    if (k <= 16) {
      NK_SWITCH_1_16(k, PA_LAUNCH)
    } else {
      switch (k) {
        case 17: PA_LAUNCH(17); break; case 18: PA_LAUNCH(18); break; case 19: PA_LAUNCH(19); break;
        case 20: PA_LAUNCH(20); break; case 21: PA_LAUNCH(21); break; case 22: PA_LAUNCH(22); break;
        case 23: PA_LAUNCH(23); break; case 24: PA_LAUNCH(24); break; case 25: PA_LAUNCH(25); break;
        case 26: PA_LAUNCH(26); break; case 27: PA_LAUNCH(27); break; case 28: PA_LAUNCH(28); break;
        case 29: PA_LAUNCH(29); break; case 30: PA_LAUNCH(30); break; default: PA_LAUNCH(31); break;
      }
    }
#undef PA_LAUNCH
  }
  {
    nk_prof_scope prof_(ctx, NK_K_REDUCE_SMALL, 8.0 * (k + 1) * grid);
    NK_LAUNCH(ctx, k_reduce_sum, dim3(k + 1), dim3(NK_BLOCK), ctx->d_partials, grid, d_h, d_skip, d_scales, k + 1);
  }
  NK_HIP(hipGetLastError());
  return nk_comm_allreduce(ctx, d_h, k + 1, 0);
}

// ----------------------------------------------------------------------------- DCGS2 with one reduction per step
// dot sweep: NV final columns, pending p = V[:,NV], z = V[:,NV+1] (absent when !WITHZ: the flush of a cycle).
// slots: [0,NV) ṽ_j·p | NV: p·p | [NV+1, 2NV+1) ṽ_j·z | 2NV+1: p·z
// Every workgroup owns a contiguous tile of DR·256 rows: its slice of p (and z) stays in registers while the final
// columns stream past eight at a time — 16 accumulators instead of 2·NV+2, so the kernel keeps its occupancy at any NV
// (a first version with all accumulators live ran at 3.0 TB/s). nv is a run-time argument.
constexpr int DR = 8;  // rows per thread
template <bool WITHZ>
__global__ __launch_bounds__(NK_BLOCK) void k_dcgs2r_dots(int64_t n, int nv, const double *__restrict__ V, int64_t ldv,
                                                          double *__restrict__ partials, const int *d_skip) {
  SKIP_GUARD(d_skip);
  constexpr int CH = 2;  // columns per chunk (measured on MI355X: 16 → 3.0, 8 → 4.8, 4 → 5.5, 2 → 6.25, 1 → 5.8 TB/s)
  __shared__ double sm[4 * 2 * CH + 8];
  const double *__restrict__ pk = V + (size_t)nv * ldv;
  const double *__restrict__ zk = V + (size_t)(nv + 1) * ldv;
  const unsigned nn = (unsigned)n, base = blockIdx.x * (NK_BLOCK * DR) + threadIdx.x;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  unsigned idx[DR];
  double pv[DR], zv[DR], msk[DR];
  double aa = 0.0, dd = 0.0;
#pragma unroll
  for (int i = 0; i < DR; ++i) {
    const unsigned r = base + NK_BLOCK * i;
    idx[i] = r < nn ? r : 0;          // clamped: loads stay unconditional
    msk[i] = r < nn ? 1.0 : 0.0;
  }
#pragma unroll
  for (int i = 0; i < DR; ++i) {
    pv[i] = pk[idx[i]] * msk[i];
    zv[i] = WITHZ ? zk[idx[i]] * msk[i] : 0.0;
    aa += pv[i] * pv[i];
    dd += pv[i] * zv[i];
  }
  const int zoff = nv + 1;  // slot of ṽ_0·z
  for (int c0 = 0; c0 < nv; c0 += CH) {
    double acc[2 * CH];
#pragma unroll
    for (int q = 0; q < 2 * CH; ++q) acc[q] = 0.0;
#pragma unroll
    for (int jj = 0; jj < CH; ++jj) {
      const int j = min(c0 + jj, nv - 1);  // columns past nv repeat the last one; their sums are not stored
      const double *__restrict__ col = V + (size_t)j * ldv;
      double vj[DR];
#pragma unroll
      for (int i = 0; i < DR; ++i) vj[i] = col[idx[i]];
#pragma unroll
      for (int i = 0; i < DR; ++i) {
        acc[2 * jj] += vj[i] * pv[i];
        if (WITHZ) acc[2 * jj + 1] += vj[i] * zv[i];
      }
    }
#pragma unroll
    for (int q = 0; q < 2 * CH; ++q) {
      if (WITHZ || (q & 1) == 0) {
        const double v = wave_sum(acc[q]);
        if (lane == 0) sm[wid * 2 * CH + q] = v;
      }
    }
    __syncthreads();
    if (threadIdx.x < 2 * CH) {
      const int q = threadIdx.x, j = c0 + (q >> 1);
      if (j < nv && (WITHZ || (q & 1) == 0)) {
        const double v = (sm[q] + sm[2 * CH + q]) + (sm[4 * CH + q] + sm[6 * CH + q]);
        const int slot = (q & 1) ? zoff + j : j;
        partials[(size_t)slot * gridDim.x + blockIdx.x] = v;
      }
    }
    __syncthreads();
  }
  {
    const double va = wave_sum(aa), vd = wave_sum(dd);
    if (lane == 0) { sm[wid] = va; sm[4 + wid] = vd; }
    __syncthreads();
    if (threadIdx.x == 0) {
      partials[(size_t)nv * gridDim.x + blockIdx.x] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
      if (WITHZ) partials[(size_t)(2 * nv + 1) * gridDim.x + blockIdx.x] = (sm[4] + sm[5]) + (sm[6] + sm[7]);
    }
  }
}

// axpy sweep: p ← p − Σ a_j ṽ_j ; z ← b_{nv+1}·z − Σ b_j ṽ_j − b_nv·p (no reduction). Same shape as the dot sweep: the
// workgroup's row tile of p and z lives in registers, the final columns stream past two at a time.
// ss_partials (last step of a cycle only): per-tile ‖z_new‖² — the norm that closes the cycle's last Hessenberg column.
__global__ __launch_bounds__(NK_BLOCK) void k_dcgs2r_axpy(int64_t n, int nv, double *__restrict__ V, int64_t ldv,
                                                          const double *__restrict__ ca_g, const double *__restrict__ cb_g,
                                                          const int *d_skip, double *__restrict__ ss_partials) {
  SKIP_GUARD(d_skip);
  __shared__ double ca[NK_MAX_NV + 2], cb[NK_MAX_NV + 2];
  __shared__ double s_bk, s_sz, s_w[4];
  if ((int)threadIdx.x <= nv) {  // one zero entry past the end pads an odd column count
    ca[threadIdx.x] = ((int)threadIdx.x < nv) ? ca_g[threadIdx.x] : 0.0;
    cb[threadIdx.x] = ((int)threadIdx.x < nv) ? cb_g[threadIdx.x] : 0.0;
  }
  if (threadIdx.x == 0) { s_bk = cb_g[nv]; s_sz = cb_g[nv + 1]; }
  __syncthreads();
  double *__restrict__ pk = V + (size_t)nv * ldv;
  double *__restrict__ zk = V + (size_t)(nv + 1) * ldv;
  const unsigned nn = (unsigned)n, base = blockIdx.x * (NK_BLOCK * DR) + threadIdx.x;
  unsigned idx[DR];
  double pv[DR], zv[DR];
  const double sz = s_sz, bk = s_bk;
#pragma unroll
  for (int i = 0; i < DR; ++i) {
    const unsigned r = base + NK_BLOCK * i;
    idx[i] = r < nn ? r : 0;
  }
#pragma unroll
  for (int i = 0; i < DR; ++i) {
    pv[i] = pk[idx[i]];
    zv[i] = sz * zk[idx[i]];
  }
  for (int j = 0; j < nv; j += 2) {
    const int j1 = min(j + 1, nv - 1);
    const double *__restrict__ c0 = V + (size_t)j * ldv;
    const double *__restrict__ c1 = V + (size_t)j1 * ldv;
    const double a0 = ca[j], a1 = ca[j + 1], b0 = cb[j], b1 = cb[j + 1];
    double v0[DR], v1[DR];
#pragma unroll
    for (int i = 0; i < DR; ++i) { v0[i] = c0[idx[i]]; v1[i] = c1[idx[i]]; }
#pragma unroll
    for (int i = 0; i < DR; ++i) {
      pv[i] -= a0 * v0[i];
      pv[i] -= a1 * v1[i];
      zv[i] -= b0 * v0[i];
      zv[i] -= b1 * v1[i];
    }
  }
  double ss = 0.0;
#pragma unroll
  for (int i = 0; i < DR; ++i) {
    const unsigned r = base + NK_BLOCK * i;
    if (r < nn) {
      if (nv > 0) pk[r] = pv[i];
      const double zn = zv[i] - bk * pv[i];
      zk[r] = zn;
      ss += zn * zn;
    }
  }
  if (ss_partials != nullptr) {  // uniform
    ss = wave_sum(ss);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = ss;
    __syncthreads();
    if (threadIdx.x == 0) ss_partials[blockIdx.x] = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
  }
}

#define NK_SWITCH_0_31(nvc, F)                                                                                 \
  switch (nvc) {                                                                                               \
    case 0: F(0); break;   case 1: F(1); break;   case 2: F(2); break;   case 3: F(3); break;                  \
    case 4: F(4); break;   case 5: F(5); break;   case 6: F(6); break;   case 7: F(7); break;                  \
    case 8: F(8); break;   case 9: F(9); break;   case 10: F(10); break; case 11: F(11); break;                \
    case 12: F(12); break; case 13: F(13); break; case 14: F(14); break; case 15: F(15); break;                \
    case 16: F(16); break; case 17: F(17); break; case 18: F(18); break; case 19: F(19); break;                \
    case 20: F(20); break; case 21: F(21); break; case 22: F(22); break; case 23: F(23); break;                \
    case 24: F(24); break; case 25: F(25); break; case 26: F(26); break; case 27: F(27); break;                \
    case 28: F(28); break; case 29: F(29); break; case 30: F(30); break; default: F(31); break;                \
  }

int nk_blas_dcgs2r_dots(nk_ctx *ctx, int64_t n, int k, bool flush, const double *V, int64_t ldv, double *d_red,
                        const int *d_skip) {
  NK_REQUIRE(k >= 0 && k <= NK_MAX_NV - 2, "DCGS2-1R handles 0..%d final columns (got %d)", NK_MAX_NV - 2, k);
  NK_REQUIRE(n < (1ll << 31), "fused pass: local vector too long for 32-bit offsets");
  const int64_t tile = (int64_t)NK_BLOCK * DR;
  const int64_t g64 = (n + tile - 1) / tile;
  NK_REQUIRE(g64 <= NK_MAX_ROW_TILES, "DCGS2-1R: %lld local rows exceed %d row tiles of %lld (use NK_ORTHO_DCGS2)",
             (long long)n, NK_MAX_ROW_TILES, (long long)tile);
  const int grid = g64 > 0 ? (int)g64 : 1;
  const int nslots = flush ? k + 1 : 2 * k + 2;
  {
    nk_prof_scope prof_(ctx, NK_K_MULTIDOT, 8.0 * (double)n * (k + (flush ? 1 : 2)));
    if (flush) NK_LAUNCH(ctx, k_dcgs2r_dots<false>, dim3(grid), dim3(NK_BLOCK), n, k, V, ldv, ctx->d_partials, d_skip);
    else NK_LAUNCH(ctx, k_dcgs2r_dots<true>, dim3(grid), dim3(NK_BLOCK), n, k, V, ldv, ctx->d_partials, d_skip);
  }
  const nk_peer_ar_view pv = nk_peer_ar_next(ctx, nslots);
  {
    nk_prof_scope prof_(ctx, NK_K_REDUCE_SMALL, 8.0 * nslots * grid);
    if (pv.seq) NK_LAUNCH(ctx, k_reduce_sum_allreduce, dim3(nslots), dim3(NK_BLOCK), (const double *)ctx->d_partials, grid, d_red, pv);
    else NK_LAUNCH(ctx, k_reduce_sum, dim3(nslots), dim3(NK_BLOCK), ctx->d_partials, grid, d_red, d_skip,
                   (const double *)nullptr, 0);
  }
  NK_HIP(hipGetLastError());
  return pv.seq ? NK_OK : nk_comm_allreduce(ctx, d_red, nslots, 0);
}

// d_ss_out != nullptr: also ‖z_new‖² (all-reduced) into *d_ss_out — the cycle's last step
int nk_blas_dcgs2r_axpy(nk_ctx *ctx, int64_t n, int k, double *V, int64_t ldv, const double *d_a, const double *d_b,
                        const int *d_skip, double *d_ss_out) {
  NK_REQUIRE(k >= 0 && k <= NK_MAX_NV - 2, "DCGS2-1R handles 0..%d final columns (got %d)", NK_MAX_NV - 2, k);
  const int64_t tile = (int64_t)NK_BLOCK * DR;
  const int grid = (int)std::max<int64_t>(1, (n + tile - 1) / tile);
  nk_prof_scope prof_(ctx, NK_K_MULTIAXPY, 8.0 * (double)n * (k + 4));
  NK_LAUNCH(ctx, k_dcgs2r_axpy, dim3(grid), dim3(NK_BLOCK), n, k, V, ldv, d_a, d_b, d_skip,
            d_ss_out ? ctx->d_partials : (double *)nullptr);
  NK_HIP(hipGetLastError());
  if (d_ss_out) {
    NK_LAUNCH(ctx, k_reduce_sum, dim3(1), dim3(NK_BLOCK), ctx->d_partials, grid, d_ss_out, d_skip,
              (const double *)nullptr, 0);
    NK_HIP(hipGetLastError());
    return nk_comm_allreduce(ctx, d_ss_out, 1, 0);
  }
  return NK_OK;
}

// *d_out = Σ partials[0..nblk) in the fixed stage-2 order, all-reduced
// nslots sums of nblk partials each (slot-major), one workgroup per slot
int nk_blas_reduce_slots(nk_ctx *ctx, const double *partials, int nblk, int nslots, double *d_out, const int *d_skip) {
  NK_LAUNCH(ctx, k_reduce_sum, dim3(nslots), dim3(NK_BLOCK), partials, nblk, d_out, d_skip, (const double *)nullptr, 0);
  NK_HIP(hipGetLastError());
  return NK_OK;
}
// the same, all-reduced over the ranks: ONE launch on the peer path (the stage-2 reduction stores its sums straight into
// every rank's arena and the last workgroup combines them), reduction + the transport's all-reduce otherwise
int nk_blas_reduce_slots_allreduce(nk_ctx *ctx, const double *partials, int nblk, int nslots, double *d_out, const int *d_skip) {
  if (nk_ctx_is_single(ctx)) return nk_blas_reduce_slots(ctx, partials, nblk, nslots, d_out, d_skip);
  const nk_peer_ar_view pv = nk_peer_ar_next(ctx, nslots);
  if (pv.seq) {
    NK_LAUNCH(ctx, k_reduce_sum_allreduce, dim3(nslots), dim3(NK_BLOCK), partials, nblk, d_out, pv);
    NK_HIP(hipGetLastError());
    return NK_OK;
  }
  NK_TRY(nk_blas_reduce_slots(ctx, partials, nblk, nslots, d_out, d_skip));
  return nk_comm_allreduce(ctx, d_out, nslots, 0);
}
int nk_blas_reduce_one(nk_ctx *ctx, const double *partials, int nblk, double *d_out, const int *d_skip) {
  NK_LAUNCH(ctx, k_reduce_sum, dim3(1), dim3(NK_BLOCK), partials, nblk, d_out, d_skip, (const double *)nullptr, 0);
  NK_HIP(hipGetLastError());
  return nk_comm_allreduce(ctx, d_out, 1, 0);
}

// d_h2[0..nv) = s_j·(ṽ_j·w_new) and d_h2[nv] = ‖w_new‖² (both all-reduced). Requires nv ≤ 32.
int nk_blas_fused_axpy_dot(nk_ctx *ctx, int64_t n, int nv, const double *V, int64_t ldv, const double *d_h,
                           const double *d_scales, double *w, double *d_h2, const int *d_skip) {
  NK_REQUIRE(nv >= 1 && nv <= 32, "fused pass handles 1..32 columns (got %d)", nv);
  NK_REQUIRE(n < (1ll << 31), "fused pass: local vector too long for 32-bit offsets");
  const int grid = nk_grid_for(n, NK_BLOCK * 4, NK_DOT_BLOCKS);
  {
    nk_prof_scope prof_(ctx, NK_K_MULTIAXPY, 8.0 * (double)n * (nv + 2));
#define FU_LAUNCH(N)                                                                                            \
  NK_LAUNCH(ctx, k_fused_axpy_dot<N>, dim3(grid), dim3(NK_BLOCK), n, V, ldv, d_h, d_scales, w, \
                     ctx->d_partials, d_skip)
    if (nv <= 16) {
      NK_SWITCH_1_16(nv, FU_LAUNCH)
    } else {
      switch (nv) {
        case 17: FU_LAUNCH(17); break; case 18: FU_LAUNCH(18); break; case 19: FU_LAUNCH(19); break;
        case 20: FU_LAUNCH(20); break; case 21: FU_LAUNCH(21); break; case 22: FU_LAUNCH(22); break;
        case 23: FU_LAUNCH(23); break; case 24: FU_LAUNCH(24); break; case 25: FU_LAUNCH(25); break;
        case 26: FU_LAUNCH(26); break; case 27: FU_LAUNCH(27); break; case 28: FU_LAUNCH(28); break;
        case 29: FU_LAUNCH(29); break; case 30: FU_LAUNCH(30); break; case 31: FU_LAUNCH(31); break;
        default: FU_LAUNCH(32); break;
      }
    }
#undef FU_LAUNCH
  }
  ctx->last_red_grid = grid;
  if (d_h2 == NK_SUMSQ_PARTIALS_ONLY) {  // single rank: the consumer (k_givens_dcgs2) reduces ctx->d_partials itself
    NK_HIP(hipGetLastError());
    return NK_OK;
  }
  {
    nk_prof_scope prof_(ctx, NK_K_REDUCE_SMALL, 8.0 * (nv + 1) * grid);
    NK_LAUNCH(ctx, k_reduce_sum, dim3(nv + 1), dim3(NK_BLOCK), ctx->d_partials, grid, d_h2, d_skip,
                       d_scales, nv);
  }
  NK_HIP(hipGetLastError());
  return nk_comm_allreduce(ctx, d_h2, nv + 1, 0);
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
  NK_LAUNCH(ctx, k_dot, dim3(grid), dim3(NK_BLOCK), n, x, y, ctx->d_partials);
  NK_LAUNCH(ctx, k_reduce_sum, dim3(1), dim3(NK_BLOCK), ctx->d_partials, grid, d_out,
                     (const int *)nullptr, (const double *)nullptr, 0);
  NK_HIP(hipGetLastError());
  return nk_comm_allreduce(ctx, d_out, 1, 0);
}
int nk_blas_sumsq(nk_ctx *ctx, int64_t n, const double *x, double *d_out) { return nk_blas_dot(ctx, n, x, x, d_out); }
int nk_blas_norm_inf(nk_ctx *ctx, int64_t n, const double *x, double *d_out) {
  const int grid = nk_grid_for(n, NK_BLOCK * 4, NK_MAX_RED_BLOCKS);
  NK_LAUNCH(ctx, k_absmax, dim3(grid), dim3(NK_BLOCK), n, x, ctx->d_partials);
  NK_LAUNCH(ctx, k_reduce_nanmax, dim3(1), dim3(NK_BLOCK), ctx->d_partials, grid, d_out, 1.0);
  NK_HIP(hipGetLastError());
  return nk_comm_allreduce(ctx, d_out, 1, 1);
}
int nk_blas_minmax(nk_ctx *ctx, int64_t n, const double *x, double *d_out2) {
  const int grid = nk_grid_for(n, NK_BLOCK * 4, NK_MAX_RED_BLOCKS);
  NK_LAUNCH(ctx, k_minmax, dim3(grid), dim3(NK_BLOCK), n, x, ctx->d_partials);
  // out[0] = max(x), out[1] = max(-x); both all-reduced with max, caller negates out[1] → min
  NK_LAUNCH(ctx, k_reduce_nanmax, dim3(2), dim3(NK_BLOCK), ctx->d_partials, grid, d_out2, 1.0);
  NK_HIP(hipGetLastError());
  return nk_comm_allreduce(ctx, d_out2, 2, 1);
}

// y = x with the per-block Σ x² on the way (the start of a GMRES cycle: b → column 0 and ‖b‖² in one pass)
__global__ __launch_bounds__(NK_BLOCK) void k_copy_sumsq(int64_t n, const double *__restrict__ x, double *__restrict__ y,
                                                         double *__restrict__ partials) {
  __shared__ double sm[4];
  double s = 0.0;
  const int64_t npair = n >> 1, stride = (int64_t)gridDim.x * NK_BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x; i < npair; i += stride) {
    const double2 a = reinterpret_cast<const double2 *>(x)[i];
    reinterpret_cast<double2 *>(y)[i] = a;
    s += a.x * a.x + a.y * a.y;
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) { y[n - 1] = x[n - 1]; s += x[n - 1] * x[n - 1]; }
  s = block_sum(s, sm);
  if (threadIdx.x == 0) partials[blockIdx.x] = s;
}
// stage 1 alone: y = x and the per-workgroup partial sums of Σ x² in ctx->d_partials[0..grid) — for a consumer kernel that
// reduces them itself, in k_reduce_sum's order (one rank)
int nk_blas_copy_sumsq_stage1(nk_ctx *ctx, int64_t n, const double *x, double *y, int *grid_out) {
  const int grid = nk_grid_for(n >> 1, NK_BLOCK * 2, NK_MAX_RED_BLOCKS);
  NK_LAUNCH(ctx, k_copy_sumsq, dim3(grid), dim3(NK_BLOCK), n, x, y, ctx->d_partials);
  NK_HIP(hipGetLastError());
  *grid_out = grid;
  return NK_OK;
}
int nk_blas_copy_sumsq(nk_ctx *ctx, int64_t n, const double *x, double *y, double *d_out) {
  const int grid = nk_grid_for(n >> 1, NK_BLOCK * 2, NK_MAX_RED_BLOCKS);
  NK_LAUNCH(ctx, k_copy_sumsq, dim3(grid), dim3(NK_BLOCK), n, x, y, ctx->d_partials);
  NK_LAUNCH(ctx, k_reduce_sum, dim3(1), dim3(NK_BLOCK), ctx->d_partials, grid, d_out,
            (const int *)nullptr, (const double *)nullptr, 0);
  NK_HIP(hipGetLastError());
  return nk_comm_allreduce(ctx, d_out, 1, 0);
}

// max|x| (NaN-propagating) and Σ x² in one pass; stage 2 also folds a third set of partial sums left by another kernel
__global__ __launch_bounds__(NK_BLOCK) void k_absmax_sumsq(int64_t n, const double *__restrict__ x,
                                                           double *__restrict__ partials) {
  __shared__ double sm[8];
  double m = 0.0, s = 0.0;
  const int64_t stride = (int64_t)gridDim.x * NK_BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x; i < n; i += stride) {
    const double v = x[i];
    m = nanmax(m, fabs(v));
    s += v * v;
  }
  m = wave_nanmax(m);
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) { sm[threadIdx.x >> 6] = m; sm[4 + (threadIdx.x >> 6)] = s; }
  __syncthreads();
  if (threadIdx.x == 0) {
    partials[blockIdx.x] = nanmax(nanmax(sm[0], sm[1]), nanmax(sm[2], sm[3]));
    partials[gridDim.x + blockIdx.x] = (sm[4] + sm[5]) + (sm[6] + sm[7]);
  }
}
// h_dst != nullptr: the results also go to coherent pinned host memory and the sequence word is released (what k_publish does
// as a launch of its own — the once-per-step norms of the Newton driver save that launch on one rank)
__global__ __launch_bounds__(NK_BLOCK) void k_reduce_inf2(const double *__restrict__ partials, int nblk,
                                                          const double *__restrict__ extra, int extra_n,
                                                          double *__restrict__ out, double *h_dst, uint64_t *h_seq, uint64_t seq) {
  __shared__ double sm[12];
  nk_fold_norms f;
  f.partials = partials; f.nblk = nblk; f.extra = extra; f.extra_n = extra_n; f.out = out; f.h_dst = h_dst; f.h_seq = h_seq; f.seq = seq;
  nk_reduce_inf2_body(f, sm);
}
int nk_blas_norms_inf2(nk_ctx *ctx, int64_t n, const double *x, double *d_out, const double *extra_partials, int extra_n,
                       int have_partials) {
  const int grid = have_partials > 0 ? have_partials : nk_grid_for(n, NK_BLOCK * 4, NK_MAX_RED_BLOCKS);
  if (have_partials <= 0) NK_LAUNCH(ctx, k_absmax_sumsq, dim3(grid), dim3(NK_BLOCK), n, x, ctx->d_partials);
  NK_LAUNCH(ctx, k_reduce_inf2, dim3(1), dim3(NK_BLOCK), (const double *)ctx->d_partials, grid, extra_partials, extra_n, d_out,
            (double *)nullptr, (uint64_t *)nullptr, (uint64_t)0);
  NK_HIP(hipGetLastError());
  return nk_comm_allreduce_mixed(ctx, d_out, extra_partials ? 3 : 2, 0, 1);  // [max, +, +]: one message on the peer path
}
__global__ __launch_bounds__(NK_BLOCK) void k_publish(const double *__restrict__ src, int count, double *h_dst, uint64_t *h_seq,
                                                      uint64_t seq);
// the same, with the 2–3 results delivered to the host (`h_out`): on one rank the stage-2 launch publishes them itself
int nk_blas_norms_inf2_to_host(nk_ctx *ctx, int64_t n, const double *x, double *d_out, const double *extra_partials, int extra_n,
                               double *h_out, const std::function<int()> &before_wait, int have_partials,
                               const std::function<int(const nk_fold_norms &, bool *)> &fold_into) {
  const int count = extra_partials ? 3 : 2;
  static const bool legacy = getenv("NK_FETCH_MEMCPY") != nullptr || getenv("NK_NORMS_SEPARATE_PUBLISH") != nullptr;
  if (!nk_ctx_is_single(ctx) || legacy) {
    NK_TRY(nk_blas_norms_inf2(ctx, n, x, d_out, extra_partials, extra_n, have_partials));
    if (legacy) return nk_scalars_to_host(ctx, d_out, count, h_out);
    // several ranks: the publish is a launch of its own; work the caller wants in the queue behind it goes in before the wait
    const uint64_t seq = ++ctx->seq;
    NK_LAUNCH(ctx, k_publish, dim3(1), dim3(NK_BLOCK), (const double *)d_out, count, ctx->h_pinned_dev, ctx->h_seq_dev, seq);
    NK_HIP(hipGetLastError());
    if (before_wait) NK_TRY(before_wait());
    volatile uint64_t *hs = ctx->h_seq;
    NK_TRY(nk_spin_wait(ctx, [&] { return __atomic_load_n(hs, __ATOMIC_ACQUIRE) == seq; }, "published norms"));
    for (int i = 0; i < count; ++i) h_out[i] = ctx->h_pinned[i];
    return NK_OK;
  }
  const int grid = have_partials > 0 ? have_partials : nk_grid_for(n, NK_BLOCK * 4, NK_MAX_RED_BLOCKS);
  const uint64_t seq = ++ctx->seq;
  if (have_partials <= 0) NK_LAUNCH(ctx, k_absmax_sumsq, dim3(grid), dim3(NK_BLOCK), n, x, ctx->d_partials);
  bool folded = false;
  static const bool fold_off = getenv("NK_FOLD_NORMS") && atoi(getenv("NK_FOLD_NORMS")) == 0;   // A/B switch
  if (fold_into && have_partials > 0 && !fold_off) {
    nk_fold_norms f;
    f.partials = ctx->d_partials; f.nblk = grid; f.extra = extra_partials; f.extra_n = extra_n; f.out = d_out;
    f.h_dst = ctx->h_pinned_dev; f.h_seq = ctx->h_seq_dev; f.seq = seq;
    NK_TRY(fold_into(f, &folded));
  }
  if (!folded)
    NK_LAUNCH(ctx, k_reduce_inf2, dim3(1), dim3(NK_BLOCK), (const double *)ctx->d_partials, grid, extra_partials, extra_n, d_out,
              ctx->h_pinned_dev, ctx->h_seq_dev, seq);
  NK_HIP(hipGetLastError());
  if (before_wait) NK_TRY(before_wait());   // (the host's round trip for these scalars then overlaps with that work)
  volatile uint64_t *hs = ctx->h_seq;
  NK_TRY(nk_spin_wait(ctx, [&] { return __atomic_load_n(hs, __ATOMIC_ACQUIRE) == seq; }, "published norms"));
  for (int i = 0; i < count; ++i) h_out[i] = ctx->h_pinned[i];
  return NK_OK;
}

// ----------------------------------------------------------------------------- several reductions in one pass
// Up to NK_MR_MAX inner products Σ x_q·y_q and one NaN-propagating max|a| over the same row range in ONE sweep, one
// stage-2 launch (which also folds `extra_slots` sets of per-block partial sums another kernel left behind) — the
// trust-region step's ρ quantities arrive with one fetch instead of six reductions of two launches each.
struct nk_mr_spec {
  const double *x[NK_MR_MAX], *y[NK_MR_MAX];
  const double *amax;
  int ndots;
};
__global__ __launch_bounds__(NK_BLOCK) void k_multi_reduce(int64_t n, nk_mr_spec sp, double *__restrict__ partials) {
  __shared__ double sm[4 * (NK_MR_MAX + 1)];
  double acc[NK_MR_MAX], mx = 0.0;
#pragma unroll
  for (int q = 0; q < NK_MR_MAX; ++q) acc[q] = 0.0;
  const int64_t stride = (int64_t)gridDim.x * NK_BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x; i < n; i += stride) {
#pragma unroll
    for (int q = 0; q < NK_MR_MAX; ++q)
      if (q < sp.ndots) acc[q] += sp.x[q][i] * sp.y[q][i];
    if (sp.amax != nullptr) mx = nanmax(mx, fabs(sp.amax[i]));
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < NK_MR_MAX; ++q) {
    if (q < sp.ndots) {
      const double v = wave_sum(acc[q]);
      if (lane == 0) sm[wid * (NK_MR_MAX + 1) + q] = v;
    }
  }
  mx = wave_nanmax(mx);
  if (lane == 0) sm[wid * (NK_MR_MAX + 1) + NK_MR_MAX] = mx;
  __syncthreads();
  const int t = threadIdx.x;
  constexpr int L = NK_MR_MAX + 1;
  if (t < sp.ndots) partials[(size_t)t * gridDim.x + blockIdx.x] = (sm[t] + sm[L + t]) + (sm[2 * L + t] + sm[3 * L + t]);
  if (t == NK_MR_MAX && sp.amax != nullptr)
    partials[(size_t)sp.ndots * gridDim.x + blockIdx.x] =
        nanmax(nanmax(sm[NK_MR_MAX], sm[L + NK_MR_MAX]), nanmax(sm[2 * L + NK_MR_MAX], sm[3 * L + NK_MR_MAX]));
}
// block s < ndots: Σ partials[s][:] ; block ndots (if has_max): nanmax ; blocks after that: Σ extra[e][:]
__global__ __launch_bounds__(NK_BLOCK) void k_multi_reduce2(const double *__restrict__ partials, int nblk, int ndots, int has_max,
                                                            const double *__restrict__ extra, int extra_n,
                                                            double *__restrict__ out) {
  __shared__ double sm[4];
  const int b = blockIdx.x;
  const int nmain = ndots + (has_max ? 1 : 0);
  if (has_max && b == ndots) {
    const double *p = partials + (size_t)b * nblk;
    double v = -__builtin_inf();
    for (int i = threadIdx.x; i < nblk; i += NK_BLOCK) v = nanmax(v, p[i]);
    v = wave_nanmax(v);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) out[b] = nanmax(nanmax(sm[0], sm[1]), nanmax(sm[2], sm[3]));
    return;
  }
  const double *p = (b < nmain) ? partials + (size_t)b * nblk : extra + (size_t)(b - nmain) * extra_n;
  const int cnt = (b < nmain) ? nblk : extra_n;
  double v = 0.0;
  for (int i = threadIdx.x; i < cnt; i += NK_BLOCK) v += p[i];
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) out[b] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
}
// d_out = [dot_0 … dot_{ndots−1}, max|amax| (if amax), extra sums …]; dots and extras all-reduced with +, the max with max
int nk_blas_multi_reduce(nk_ctx *ctx, int64_t n, int ndots, const double *const *xs, const double *const *ys,
                         const double *amax, const double *extra_partials, int extra_slots, int extra_n, double *d_out) {
  NK_REQUIRE(ndots >= 0 && ndots <= NK_MR_MAX, "multi-reduce: at most %d inner products", NK_MR_MAX);
  nk_mr_spec sp;
  for (int q = 0; q < NK_MR_MAX; ++q) { sp.x[q] = q < ndots ? xs[q] : nullptr; sp.y[q] = q < ndots ? ys[q] : nullptr; }
  sp.amax = amax;
  sp.ndots = ndots;
  const int grid = nk_grid_for(n, NK_BLOCK * 4, NK_MAX_RED_BLOCKS);
  const int has_max = amax ? 1 : 0;
  if (ndots || has_max) NK_LAUNCH(ctx, k_multi_reduce, dim3(grid), dim3(NK_BLOCK), n, sp, ctx->d_partials);
  NK_LAUNCH(ctx, k_multi_reduce2, dim3(ndots + has_max + extra_slots), dim3(NK_BLOCK), (const double *)ctx->d_partials, grid,
            ndots, has_max, extra_partials, extra_n, d_out);
  NK_HIP(hipGetLastError());
  return nk_comm_allreduce_mixed(ctx, d_out, ndots + has_max + extra_slots, ndots, ndots + has_max);
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
  NK_LAUNCH(ctx, k_axpby, dim3(ew_grid(n)), dim3(NK_BLOCK), n, a, x, b, y);
  NK_HIP(hipGetLastError());
  return NK_OK;
}
int nk_blas_lincomb(nk_ctx *ctx, int64_t n, double a, const double *x, double b, const double *y, double *z) {
  NK_LAUNCH(ctx, k_lincomb, dim3(nk_grid_for(n, NK_BLOCK * 2, 4096)), dim3(NK_BLOCK), n, a, x, b, y, z);
  NK_HIP(hipGetLastError());
  return NK_OK;
}
int nk_blas_scale_to(nk_ctx *ctx, int64_t n, const double *d_scale, const double *x, double *y, const int *d_skip) {
  nk_prof_scope prof_(ctx, NK_K_SCALE, 16.0 * (double)n);
  NK_LAUNCH(ctx, k_scale_to, dim3(ew_grid(n)), dim3(NK_BLOCK), n, d_scale, x, y, d_skip);
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
  NK_LAUNCH(ctx, k_fill, dim3(nk_grid_for(n, NK_BLOCK * 4, 4096)), dim3(NK_BLOCK), n, a, y);
  NK_HIP(hipGetLastError());
  return NK_OK;
}
// Device scalars → host without a copy engine and without a stream synchronisation: a one-workgroup kernel stores them
// into coherent pinned memory, then releases a sequence word the host polls (a D2H hipMemcpyAsync + hipStreamSynchronize
// cost ≈ 14 µs of blit kernel plus the wake-up latency per call — 2–8 calls per nonlinear step).
__global__ __launch_bounds__(NK_BLOCK) void k_publish(const double *__restrict__ src, int count, double *h_dst,
                                                      uint64_t *h_seq, uint64_t seq) {
  if ((int)threadIdx.x < count) h_dst[threadIdx.x] = src[threadIdx.x];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(h_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
int nk_scalars_to_host(nk_ctx *ctx, const double *d_src, int count, double *h_dst, const std::function<int()> &before_wait) {
  NK_REQUIRE(count <= NK_BLOCK && count <= 4 * NK_MAX_NV, "too many scalars");
  static const bool legacy = getenv("NK_FETCH_MEMCPY") != nullptr;  // A/B switch: copy + synchronise
  if (legacy) {
    NK_HIP(hipMemcpyAsync(ctx->h_pinned, d_src, count * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    NK_HIP(hipStreamSynchronize(ctx->stream));
  } else {
    const uint64_t seq = ++ctx->seq;
    NK_LAUNCH(ctx, k_publish, dim3(1), dim3(NK_BLOCK), d_src, count, ctx->h_pinned_dev, ctx->h_seq_dev, seq);
    NK_HIP(hipGetLastError());
    if (before_wait) NK_TRY(before_wait());   // (work the caller wants in the queue while the host waits)
    volatile uint64_t *hs = ctx->h_seq;
    NK_TRY(nk_spin_wait(ctx, [&] { return __atomic_load_n(hs, __ATOMIC_ACQUIRE) == seq; }, "published scalars"));
  }
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
  NK_TRY(nk_blas_multidot(ctx, n, nv, V, ldv, w, ctx->d_scal, false, nullptr, nullptr));
  return nk_scalars_to_host(ctx, ctx->d_scal, nv, h_host);
}
// bench/test export of the fused CGS2 pass: w ← w − V(h∘s); h2[j] = s_j ṽ_j·w_new (j<nv), h2[nv] = ‖w_new‖²
extern "C" int nk_fused_axpy_dot(nk_ctx *ctx, int64_t n, int nv, const double *V, int64_t ldv, const double *h_host,
                                 const double *s_host, double *w, double *h2_host) {
  NK_REQUIRE(ctx && V && w && h_host && s_host && h2_host, "NULL argument");
  NK_REQUIRE(nv >= 1 && nv <= 32, "nv out of range");
  for (int j = 0; j < nv; ++j) { ctx->h_pinned[j] = h_host[j]; ctx->h_pinned[NK_MAX_NV + j] = s_host[j]; }
  NK_HIP(hipMemcpyAsync(ctx->d_scal, ctx->h_pinned, 2 * NK_MAX_NV * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  NK_TRY(nk_blas_fused_axpy_dot(ctx, n, nv, V, ldv, ctx->d_scal, ctx->d_scal + NK_MAX_NV, w, ctx->d_scal + 2 * NK_MAX_NV,
                                nullptr));
  return nk_scalars_to_host(ctx, ctx->d_scal + 2 * NK_MAX_NV, nv + 1, h2_host);
}
extern "C" int nk_multiaxpy(nk_ctx *ctx, int64_t n, int nv, const double *V, int64_t ldv, const double *h_host,
                            double *w, double *wnorm2) {
  NK_REQUIRE(ctx && V && w && h_host, "NULL argument");
  NK_REQUIRE(nv <= NK_MAX_NV && (ldv & 1) == 0, "bad nv/ldv");
  for (int j = 0; j < nv; ++j) ctx->h_pinned[j] = h_host[j];
  NK_HIP(hipMemcpyAsync(ctx->d_scal, ctx->h_pinned, nv * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  NK_TRY(nk_blas_multiaxpy(ctx, n, nv, V, ldv, ctx->d_scal, -1.0, w, wnorm2 ? ctx->d_scal + NK_MAX_NV : nullptr,
                           nullptr, nullptr, nullptr));
  if (wnorm2) return nk_scalars_to_host(ctx, ctx->d_scal + NK_MAX_NV, 1, wnorm2);
  NK_HIP(hipStreamSynchronize(ctx->stream));
  return NK_OK;
}
