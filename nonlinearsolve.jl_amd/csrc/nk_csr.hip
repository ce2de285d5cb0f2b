// Row-partitioned CSR (f64 values, i32 local column ids) and the SpMV kernels.
//
// SpMV design ("CSR-stream", LDS-staged partial products): a 256-thread workgroup owns a contiguous range
// of rows whose non-zeros fit one LDS tile. Phase 1 streams vals/colind of the whole tile with fully
// coalesced lane-contiguous loads (no per-row divergence), gathers x, and parks val*x in LDS. Phase 2 gives
// each lane one row and sums that row's products from LDS in CSR order — so the result is bit-identical
// to a sequential CPU row sum, with no atomics. Row-block boundaries are computed once on the host.
// The blockIdx→row-block map is XCD-aware: consecutive row blocks go to the same XCD (block b runs on
// XCD b%8) so that the x gathers of neighbouring grid lines hit that XCD's private 4 MiB L2.
//
// Algorithmic bytes per launch: 12 nnz + 4 (nrows+1) + 8 nrows (y) + 8 nrows (x)  ≈ 80 N for 5-pt Bratu.
#include <algorithm>
#include <stdlib.h>
#include <string.h>

#include "nk_internal.h"

constexpr int SPMV_TILE_MAX = 4096;  // largest LDS tile a variant may use (32 KiB)
constexpr int NXCD = 8;

__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  // bijective for any nblk: XCD x gets the contiguous chunk [x*q + min(x,r), ...) of row blocks
  const int q = nblk / NXCD, r = nblk % NXCD;
  const int x = bid % NXCD, k = bid / NXCD;
  return x * q + (x < r ? x : r) + k;
}

// TILE: non-zeros per workgroup tile (LDS = 8·TILE bytes). HALO: columns ≥ nlocal live in the halo buffer.
// The tile phase is written branch-free on purpose: hipcc turns `if (k < nnzb)` around a load into an exec-mask
// branch plus a full `s_waitcnt vmcnt(0)` per element (seen in the ISA: one load in flight per wave). Instead
// every lane issues ALL its col/val loads unconditionally (the arrays are padded by one tile), then all x
// gathers (out-of-tile lanes gather x[0]), then the LDS writes — TILE/256 independent loads in flight per lane.
// Row epilogue. mode 0: y[row] = scale·s. mode 1 (fused Chebyshev step, x = d_old): r −= s; d_new = c1·d_old + c2·r;
// yacc += d_new — saves the separate 56 n-byte vector update and a kernel boundary per polynomial degree (epi.dinv: the step
// runs on D⁻¹A — the algebraic multigrid's smoother, nk_amg.hip). mode 2: y = b − A x. mode 3: shifted. mode 4: r −= A x.
__device__ __forceinline__ void spmv_store_row(int row, double s, double *__restrict__ y, const double *out_scale,
                                               double os, const double *__restrict__ x, const nk_spmv_epi &epi) {
  if (epi.mode == 0) {
    y[row] = out_scale ? os * s : s;
  } else if (epi.mode == 3) {  // Newton-basis step of the s-step Arnoldi process: y = scale·(A x − θ x)
    const double v = s - (*epi.theta) * x[row];
    y[row] = out_scale ? os * v : v;
  } else if (epi.mode == 2) {  // fused residual: y = b − A x (b = epi.r), as the stencil kernels' mode 2
    y[row] = epi.r[row] - s;
  } else if (epi.mode == 4) {  // residual update only: r −= A x
    epi.r[row] -= s;
  } else {                     // Chebyshev step (x = d_old); with epi.dinv on D⁻¹A (the multigrid smoothers)
    const double rr = epi.r[row] - s;
    epi.r[row] = rr;
    const double dn = epi.c1 * x[row] + epi.c2 * (epi.dinv ? epi.dinv[row] * rr : rr);
    epi.dnew[row] = dn;
    epi.yacc[row] += dn;
  }
}

// The halo exchange inside the SpMV launch (peer-mapped arenas, several ranks): the first `nsegs` workgroups to be
// dispatched first store this rank's entries for neighbour s into that neighbour's receive area and release the flag over
// there; a workgroup whose row block reads halo columns (descriptor index ≥ wait_from — they are ordered last) waits for
// every neighbour's flag before it gathers. Pushes never wait for anything, so no cycle can form; one launch less per
// operator application than the separate exchange kernel.
struct nk_spmv_xchg {
  const nk_peer_seg *segs = nullptr;
  const int32_t *send_idx = nullptr;
  uint64_t *err = nullptr;
  uint64_t seq = 0;
  int nsegs = 0, wait_from = 0;
};
template <int TILE, bool HALO, bool REMAP>
__global__ __launch_bounds__(NK_BLOCK) void k_spmv_stream(
    int nblk, const int4 *__restrict__ rowblocks, const int32_t *__restrict__ rowptr,
    const int32_t *__restrict__ col, const double *__restrict__ val, const double *__restrict__ x,
    const double *__restrict__ xhalo, int32_t nlocal, double *__restrict__ y, const int *d_skip,
    const double *__restrict__ out_scale, const nk_spmv_epi epi, const int16_t *__restrict__ col16, int n16,
    const nk_spmv_xchg xc) {
  const int b = REMAP ? xcd_remap(blockIdx.x, nblk) : (int)blockIdx.x;
  if (HALO && xc.segs != nullptr) {  // (uniform per workgroup; runs even when the cycle is done: flags stay in step)
    if ((int)blockIdx.x < xc.nsegs) {
      // (fields read through the pointer: a by-value copy of the segment indexed by the sequence's parity lands in private memory
      //  — 56 B of scratch per lane in every HALO instance — and turns `dst` into a generic pointer, its stores into FLAT ones)
      const nk_peer_seg *__restrict__ sg = xc.segs + blockIdx.x;
      double *__restrict__ dst = (xc.seq & 1) ? sg->dst[1] : sg->dst[0];
      const int64_t send_cnt = sg->send_cnt;
      const int32_t *idx = xc.send_idx + sg->send_off;
      for (int64_t i = threadIdx.x; i < send_cnt; i += NK_BLOCK) dst[i] = x[idx[i]];
      __threadfence_system();
      __syncthreads();
      if (threadIdx.x == 0) __hip_atomic_store(sg->flag_remote, xc.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (b >= xc.wait_from) {
      if ((int)threadIdx.x < xc.nsegs) {
        const uint64_t *fl = xc.segs[threadIdx.x].flag_local;
        const unsigned long long t0 = wall_clock64();
        while (__hip_atomic_load(fl, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < xc.seq) {
          if (__hip_atomic_load(xc.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 4) break;
          if (wall_clock64() - t0 > nk_peer_timeout(xc.err)) { atomicAdd((unsigned long long *)xc.err, 1ull); break; }
          __builtin_amdgcn_s_sleep(2);
        }
      }
      __syncthreads();
    }
  }
  if (d_skip != nullptr && *d_skip != 0) return;
  const double os = out_scale ? *out_scale : 1.0;
  __shared__ double prod[TILE];
  __shared__ double red[4];
  const int4 desc = rowblocks[b];  // {first row, end row, first nnz, end nnz}: one 16-byte scalar load per block
  const int r0 = desc.x, r1 = desc.y, p0 = desc.z, p1 = desc.w;
  const int nnzb = p1 - p0;
  if (nnzb <= TILE) {
    // row bounds of this lane's first two rows, requested before the tile streams in (they are only needed
    // after the barrier; issuing them here takes an L2 round trip off the block's critical path)
    const int rA = r0 + threadIdx.x, rB = rA + NK_BLOCK;
    const int rAc = rA < r1 ? rA : r0, rBc = rB < r1 ? rB : r0;  // clamped: loads stay unconditional
    const int aA = rowptr[rAc], eA = rowptr[rAc + 1], aB = rowptr[rBc], eB = rowptr[rBc + 1];
    constexpr int PER = TILE / NK_BLOCK;
    int c[PER];
    double v[PER], xv[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) v[i] = val[p0 + threadIdx.x + NK_BLOCK * i];  // in bounds: padded by SPMV_TILE_MAX
    if (b < n16) {  // uniform: this block's columns are stored as 16-bit offsets from its first row
#pragma unroll
      for (int i = 0; i < PER; ++i) c[i] = r0 + (int)col16[p0 + threadIdx.x + NK_BLOCK * i];
    } else {
#pragma unroll
      for (int i = 0; i < PER; ++i) c[i] = col[p0 + threadIdx.x + NK_BLOCK * i];
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int k = threadIdx.x + NK_BLOCK * i;
      const int cc = (k < nnzb) ? c[i] : 0;
      if (HALO) {
        const double *__restrict__ base = (cc < nlocal) ? x : (xhalo - nlocal);
        xv[i] = base[cc];
      } else {
        xv[i] = x[cc];
      }
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) prod[threadIdx.x + NK_BLOCK * i] = v[i] * xv[i];
    __syncthreads();
    if (rA < r1) {
      double s = 0.0;
      for (int k = aA - p0; k < eA - p0; ++k) s += prod[k];
      spmv_store_row(rA, s, y, out_scale, os, x, epi);
    }
    if (rB < r1) {
      double s = 0.0;
      for (int k = aB - p0; k < eB - p0; ++k) s += prod[k];
      spmv_store_row(rB, s, y, out_scale, os, x, epi);
    }
    for (int r = rB + NK_BLOCK; r < r1; r += NK_BLOCK) {  // blocks of (almost) empty rows
      const int a = rowptr[r] - p0, e = rowptr[r + 1] - p0;
      double s = 0.0;
      for (int k = a; k < e; ++k) s += prod[k];
      spmv_store_row(r, s, y, out_scale, os, x, epi);
    }
  } else {
    // a single long row: the whole workgroup reduces it (fixed order → deterministic)
    double s = 0.0;
    for (int k = threadIdx.x; k < nnzb; k += NK_BLOCK) {
      const int cc = col[p0 + k];
      const double xx = (cc < nlocal) ? x[cc] : xhalo[cc - nlocal];
      s += val[p0 + k] * xx;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) spmv_store_row(r0, red[0] + red[1] + red[2] + red[3], y, out_scale, os, x, epi);
  }
}

static int spmv_tile_from_env() {
  const char *e = getenv("NK_SPMV_TILE");
  int t = e ? atoi(e) : 1024;  // measured best on MI355X (tools/microbench.py)
  if (t != 512 && t != 1024 && t != 2048 && t != 4096) t = 1024;
  return t;
}
static int spmv_variant_from_env() {  // 0 rolled, 1 batched loads, 2 rolled/no XCD remap
  const char *e = getenv("NK_SPMV_VARIANT");
  return e ? atoi(e) : 0;
}

static void build_rowblocks(const std::vector<int32_t> &rowptr, std::vector<int32_t> &rb, int SPMV_TILE) {
  const int64_t nrows = (int64_t)rowptr.size() - 1;
  rb.clear();
  rb.push_back(0);
  int64_t r = 0;
  while (r < nrows) {
    int64_t e = r;
    const int64_t base = rowptr[r];
    while (e < nrows && (rowptr[e + 1] - base) <= SPMV_TILE && (e - r) < 4 * NK_BLOCK) ++e;
    if (e == r) e = r + 1;  // one row longer than a tile → long-row path
    rb.push_back((int32_t)e);
    r = e;
  }
}

int nk_csr_create_local(nk_ctx *ctx, int64_t nrows, int64_t n_global, int64_t row_begin,
                        const std::vector<int32_t> &rowptr, const std::vector<int64_t> &gcol,
                        const double *vals_host, nk_csr **out, bool local_only) {
  const int64_t nnz = (int64_t)gcol.size();
  NK_REQUIRE(nnz < (1ll << 31) && nrows < (1ll << 31), "local CSR too large for int32 indices");
  nk_csr *A = new nk_csr();
  auto guard = nk_make_guard(A, [](nk_csr *a) { nk_csr_destroy(a); });
  A->ctx = ctx;
  A->nrows = nrows;
  A->n_global = n_global;
  A->row_begin = row_begin;
  A->nnz = nnz;
  A->h_rowptr = rowptr;
  A->h_col.resize(nnz);
  // map global columns → local index space [owned | halo]
  const int64_t lo = row_begin, hi = row_begin + nrows;
  std::vector<int64_t> halo;
  for (int64_t k = 0; k < nnz; ++k) {
    const int64_t g = gcol[k];
    if (g < 0 || g >= n_global) {
      NK_FAIL(NK_E_INVALID, "column index %lld out of range [0,%lld)", (long long)g, (long long)n_global);
    }
    if (g < lo || g >= hi) halo.push_back(g);
  }
  std::sort(halo.begin(), halo.end());
  halo.erase(std::unique(halo.begin(), halo.end()), halo.end());
  if ((ctx->nranks == 1 || local_only) && !halo.empty()) {
    NK_FAIL(NK_E_INVALID, "single-rank CSR must be square: column outside the local row range");
  }
  A->halo_gcols = halo;
  for (int64_t k = 0; k < nnz; ++k) {
    const int64_t g = gcol[k];
    if (g >= lo && g < hi) A->h_col[k] = (int32_t)(g - lo);
    else A->h_col[k] = (int32_t)(nrows + (std::lower_bound(halo.begin(), halo.end(), g) - halo.begin()));
  }
  std::vector<int32_t> rb;
  A->tile = spmv_tile_from_env();
  A->variant = spmv_variant_from_env();
  build_rowblocks(rowptr, rb, A->tile);
  A->nblocks = (int)rb.size() - 1;
  NK_TRY(nk_dev_alloc(&A->d_rowptr, (size_t)nrows + 1));
  NK_TRY(nk_dev_alloc(&A->d_col, (size_t)nnz + SPMV_TILE_MAX));  // padded: the tile phase reads whole tiles
  NK_TRY(nk_dev_alloc(&A->d_val, (size_t)nnz + SPMV_TILE_MAX));
  NK_HIP(nk_memset(ctx, A->d_col + nnz, 0, SPMV_TILE_MAX * sizeof(int32_t)));
  NK_HIP(nk_memset(ctx, A->d_val + nnz, 0, SPMV_TILE_MAX * sizeof(double)));
  // block descriptors, those that read no halo column first: with halo overlap the SpMV launches the two groups
  // separately (interior while the exchange is in flight, boundary after it); on one rank every block is interior
  std::vector<int32_t> desc(4 * (size_t)A->nblocks);
  {
    std::vector<int> order;
    order.reserve(A->nblocks);
    std::vector<char> touches(A->nblocks, 0);
    for (int b = 0; b < A->nblocks; ++b)
      for (int32_t k = rowptr[rb[b]]; k < rowptr[rb[b + 1]]; ++k)
        if (A->h_col[k] >= nrows) { touches[b] = 1; break; }
    for (int b = 0; b < A->nblocks; ++b) if (!touches[b]) order.push_back(b);
    A->nblocks_interior = (int)order.size();
    for (int b = 0; b < A->nblocks; ++b) if (touches[b]) order.push_back(b);
    for (int i = 0; i < A->nblocks; ++i) {
      const int b = order[i];
      desc[4 * i + 0] = rb[b];
      desc[4 * i + 1] = rb[b + 1];
      desc[4 * i + 2] = rowptr[rb[b]];
      desc[4 * i + 3] = rowptr[rb[b + 1]];
    }
  }
  // 16-bit block-relative columns for the interior blocks (all or none; NK_SPMV_COL32=1 keeps 32-bit ids everywhere)
  if (nnz && A->nblocks_interior > 0 && getenv("NK_SPMV_COL32") == nullptr) {
    std::vector<int16_t> c16((size_t)nnz, 0);
    bool fits = true;
    for (int i = 0; i < A->nblocks_interior && fits; ++i) {
      const int32_t r0 = desc[4 * i], p0 = desc[4 * i + 2], p1 = desc[4 * i + 3];
      for (int32_t k = p0; k < p1; ++k) {
        const int32_t dlt = A->h_col[k] - r0;
        if (dlt < -32768 || dlt > 32767) { fits = false; break; }
        c16[k] = (int16_t)dlt;
      }
    }
    if (fits) {
      NK_HIP(hipMalloc((void **)&A->d_col16, ((size_t)nnz + SPMV_TILE_MAX) * sizeof(int16_t)));
      NK_HIP(nk_memset(ctx, A->d_col16, 0, ((size_t)nnz + SPMV_TILE_MAX) * sizeof(int16_t)));
      NK_HIP(nk_memcpy(ctx, A->d_col16, c16.data(), (size_t)nnz * sizeof(int16_t), hipMemcpyHostToDevice));
      A->n16 = A->nblocks_interior;
    }
  }
  NK_TRY(nk_dev_alloc(&A->d_rowblocks, desc.size() + 4));
  NK_HIP(nk_memcpy(ctx, A->d_rowptr, rowptr.data(), (nrows + 1) * sizeof(int32_t), hipMemcpyHostToDevice));
  if (nnz) NK_HIP(nk_memcpy(ctx, A->d_col, A->h_col.data(), nnz * sizeof(int32_t), hipMemcpyHostToDevice));
  if (!desc.empty()) NK_HIP(nk_memcpy(ctx, A->d_rowblocks, desc.data(), desc.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  if (vals_host && nnz) NK_HIP(nk_memcpy(ctx, A->d_val, vals_host, nnz * sizeof(double), hipMemcpyHostToDevice));
  else if (nnz) NK_HIP(nk_memset(ctx, A->d_val, 0, nnz * sizeof(double)));

  // ---- halo plan (collective): tell every owner which of its entries we need
  A->local_only = local_only;
  if (ctx->nranks > 1 && !local_only) {
    const int P = ctx->nranks;
    // 1. everyone learns all row ranges: all-reduce a zero vector with our begin in slot `rank`
    std::vector<double> hb(P + 1, 0.0);
    hb[ctx->rank] = (double)row_begin;
    if (ctx->rank == P - 1) hb[P] = (double)(row_begin + nrows);
    double *d_tmp = nullptr;
    NK_TRY(nk_dev_alloc(&d_tmp, (size_t)P * P + P + 1));
    NK_HIP(nk_memcpy(ctx, d_tmp, hb.data(), (P + 1) * sizeof(double), hipMemcpyHostToDevice));
    NK_TRY(nk_comm_allreduce(ctx, d_tmp, P + 1, 0));
    NK_HIP(hipStreamSynchronize(ctx->stream));
    NK_HIP(nk_memcpy(ctx, hb.data(), d_tmp, (P + 1) * sizeof(double), hipMemcpyDeviceToHost));
    std::vector<int64_t> begin(P + 1);
    for (int p = 0; p <= P; ++p) begin[p] = (int64_t)hb[p];
    // 2. needs per owner
    std::vector<std::vector<int64_t>> need(P);
    for (int64_t g : halo) {
      int p = (int)(std::upper_bound(begin.begin(), begin.end(), g) - begin.begin()) - 1;
      if (p < 0 || p >= P || p == ctx->rank) {
        hipFree(d_tmp);
        NK_FAIL(NK_E_INVALID, "halo column %lld has no owner", (long long)g);
      }
      need[p].push_back(g);
    }
    // 3. exchange counts: P×P matrix, row = requester, col = owner
    std::vector<double> cnt((size_t)P * P, 0.0);
    for (int p = 0; p < P; ++p) cnt[(size_t)ctx->rank * P + p] = (double)need[p].size();
    NK_HIP(nk_memcpy(ctx, d_tmp, cnt.data(), (size_t)P * P * sizeof(double), hipMemcpyHostToDevice));
    NK_TRY(nk_comm_allreduce(ctx, d_tmp, P * P, 0));
    NK_HIP(hipStreamSynchronize(ctx->stream));
    NK_HIP(nk_memcpy(ctx, cnt.data(), d_tmp, (size_t)P * P * sizeof(double), hipMemcpyDeviceToHost));
    hipFree(d_tmp);
    // 4. exchange index lists (int64 global ids) — what I need from p ↔ what p needs from me
    std::vector<int64_t> soff(P, 0), sbytes(P, 0), roff(P, 0), rbytes(P, 0);
    std::vector<int64_t> sendflat;
    int64_t rtotal = 0;
    for (int p = 0; p < P; ++p) {
      soff[p] = (int64_t)sendflat.size() * 8;
      sbytes[p] = (int64_t)need[p].size() * 8;
      sendflat.insert(sendflat.end(), need[p].begin(), need[p].end());
      roff[p] = rtotal * 8;
      const int64_t c = (int64_t)cnt[(size_t)p * P + ctx->rank];
      rbytes[p] = c * 8;
      rtotal += c;
    }
    int64_t *d_s = nullptr, *d_r = nullptr;
    NK_TRY(nk_dev_alloc(&d_s, sendflat.size() + 1));
    NK_TRY(nk_dev_alloc(&d_r, (size_t)rtotal + 1));
    if (!sendflat.empty())
      NK_HIP(nk_memcpy(ctx, d_s, sendflat.data(), sendflat.size() * 8, hipMemcpyHostToDevice));
    NK_TRY(nk_comm_alltoallv(ctx, d_s, soff.data(), sbytes.data(), d_r, roff.data(), rbytes.data()));
    NK_HIP(hipStreamSynchronize(ctx->stream));
    std::vector<int64_t> wanted((size_t)rtotal);
    if (rtotal) NK_HIP(nk_memcpy(ctx, wanted.data(), d_r, (size_t)rtotal * 8, hipMemcpyDeviceToHost));
    hipFree(d_s);
    hipFree(d_r);
    std::vector<std::vector<int32_t>> send_idx(P);
    std::vector<int64_t> recv_cnt(P, 0);
    for (int p = 0; p < P; ++p) {
      const int64_t c = rbytes[p] / 8, o = roff[p] / 8;
      for (int64_t t = 0; t < c; ++t) {
        const int64_t g = wanted[o + t];
        if (g < lo || g >= hi) NK_FAIL(NK_E_INVALID, "peer %d asked for a row this rank does not own", p);
        send_idx[p].push_back((int32_t)(g - lo));
      }
      recv_cnt[p] = (int64_t)need[p].size();
    }
    // halo slots are sorted by global id and owners are ordered by rank → recv layout == halo order
    NK_TRY(nk_halo_setup(ctx, &A->halo, send_idx, recv_cnt));
  }
  *out = guard.release();
  return NK_OK;
}

template <typename I>
static void widen_indices(const void *p, int64_t count, int base, std::vector<int64_t> &out) {
  const I *q = (const I *)p;
  out.resize(count);
  for (int64_t i = 0; i < count; ++i) out[i] = (int64_t)q[i] - base;
}

extern "C" int nk_csr_create(nk_ctx *ctx, int64_t nrows_local, int64_t n_global, int64_t row_begin, int64_t nnz,
                             int index_bits, int index_base, const void *rowptr, const void *colind,
                             const double *vals, int memspace, nk_csr **out) {
  NK_REQUIRE(ctx && rowptr && (colind || nnz == 0) && out, "NULL argument");
  NK_REQUIRE(index_bits == 32 || index_bits == 64, "index_bits must be 32 or 64");
  NK_REQUIRE(index_base == 0 || index_base == 1, "index_base must be 0 or 1");
  NK_REQUIRE(nrows_local >= 0 && nnz >= 0 && row_begin >= 0 && row_begin + nrows_local <= n_global, "bad sizes");
  NK_HIP(hipSetDevice(ctx->device));
  const size_t isz = index_bits / 8;
  std::vector<char> hrp((nrows_local + 1) * isz), hci(nnz * isz);
  std::vector<double> hv;
  const void *rp = rowptr, *ci = colind;
  const double *vv = vals;
  if (memspace == NK_DEVICE) {
    NK_HIP(nk_memcpy(ctx, hrp.data(), rowptr, hrp.size(), hipMemcpyDeviceToHost));
    if (nnz) NK_HIP(nk_memcpy(ctx, hci.data(), colind, hci.size(), hipMemcpyDeviceToHost));
    rp = hrp.data();
    ci = hci.data();
    if (vals) {
      hv.resize(nnz);
      if (nnz) NK_HIP(nk_memcpy(ctx, hv.data(), vals, nnz * sizeof(double), hipMemcpyDeviceToHost));
      vv = hv.data();
    }
  }
  std::vector<int64_t> rp64, gc;
  if (index_bits == 32) {
    widen_indices<int32_t>(rp, nrows_local + 1, index_base, rp64);
    widen_indices<int32_t>(ci, nnz, index_base, gc);
  } else {
    widen_indices<int64_t>(rp, nrows_local + 1, index_base, rp64);
    widen_indices<int64_t>(ci, nnz, index_base, gc);
  }
  NK_REQUIRE(rp64[0] == 0 && rp64[nrows_local] == nnz, "rowptr does not span [0, nnz]");
  std::vector<int32_t> rp32(nrows_local + 1);
  for (int64_t i = 0; i <= nrows_local; ++i) {
    if (i > 0) NK_REQUIRE(rp64[i] >= rp64[i - 1], "rowptr not monotone at row %lld", (long long)i);
    rp32[i] = (int32_t)rp64[i];
  }
  return nk_csr_create_local(ctx, nrows_local, n_global, row_begin, rp32, gc, vv, out);
}

// Julia's SparseMatrixCSC fields as they are (1-based Int64 colptr / rowval / nzval of the WHOLE n × n matrix, present on
// every rank): this rank keeps the rows [row_begin, row_begin + nrows_local) as its CSR slice; columns stay global and the
// halo plan is built collectively, exactly as for nk_csr_create.
extern "C" int nk_csr_create_from_csc_rows(nk_ctx *ctx, int64_t n, int64_t nnz, int index_bits, int index_base,
                                           const void *colptr, const void *rowval, const double *nzval,
                                           int64_t row_begin, int64_t nrows_local, nk_csr **out) {
  NK_REQUIRE(ctx && colptr && rowval && out, "NULL argument");
  NK_REQUIRE(index_bits == 32 || index_bits == 64, "index_bits must be 32 or 64");
  NK_REQUIRE(row_begin >= 0 && nrows_local >= 0 && row_begin + nrows_local <= n, "bad row range");
  NK_REQUIRE(ctx->nranks > 1 || (row_begin == 0 && nrows_local == n), "a single rank owns every row");
  NK_HIP(hipSetDevice(ctx->device));
  std::vector<int64_t> cp, rv;
  if (index_bits == 32) {
    widen_indices<int32_t>(colptr, n + 1, index_base, cp);
    widen_indices<int32_t>(rowval, nnz, index_base, rv);
  } else {
    widen_indices<int64_t>(colptr, n + 1, index_base, cp);
    widen_indices<int64_t>(rowval, nnz, index_base, rv);
  }
  NK_REQUIRE(cp[0] == 0 && cp[n] == nnz, "colptr does not span [0, nnz]");
  const int64_t lo = row_begin, hi = row_begin + nrows_local;
  std::vector<int32_t> rp(nrows_local + 1, 0);
  int64_t nloc = 0;
  for (int64_t k = 0; k < nnz; ++k) {
    NK_REQUIRE(rv[k] >= 0 && rv[k] < n, "row index out of range");
    if (rv[k] >= lo && rv[k] < hi) { rp[rv[k] - lo + 1]++; ++nloc; }
  }
  for (int64_t i = 0; i < nrows_local; ++i) rp[i + 1] += rp[i];
  std::vector<int64_t> gc(nloc);
  std::vector<double> vals(nloc, 0.0);
  std::vector<int32_t> fill(rp.begin(), rp.end() - 1);
  for (int64_t c = 0; c < n; ++c)
    for (int64_t k = cp[c]; k < cp[c + 1]; ++k) {
      if (rv[k] < lo || rv[k] >= hi) continue;
      const int32_t pos = fill[rv[k] - lo]++;
      gc[pos] = c;  // columns ascend within each row because c ascends
      if (nzval) vals[pos] = nzval[k];
    }
  NK_TRY(nk_csr_create_local(ctx, nrows_local, n, row_begin, rp, gc, nzval ? vals.data() : nullptr, out));
  // remember where every local entry sits in the CSC value array: nk_csr_set_values_csc then refreshes the values of a new
  // Jacobian with one gather instead of a new conversion (entries keep their order inside a row through nk_csr_create_local)
  {
    nk_csr *A = *out;
    std::vector<int32_t> src(nloc > 0 ? nloc : 1, 0);
    std::vector<int32_t> fill2(rp.begin(), rp.end() - 1);
    for (int64_t c = 0; c < n; ++c)
      for (int64_t k = cp[c]; k < cp[c + 1]; ++k)
        if (rv[k] >= lo && rv[k] < hi) src[fill2[rv[k] - lo]++] = (int32_t)k;
    A->csc_nnz = nnz;
    if (nnz < (1ll << 31) && nk_dev_alloc(&A->d_csc_src, (size_t)nloc + 1) == NK_OK && nloc > 0)
      NK_HIP(nk_memcpy(ctx, A->d_csc_src, src.data(), nloc * sizeof(int32_t), hipMemcpyHostToDevice));
  }
  return NK_OK;
}
__global__ __launch_bounds__(NK_BLOCK) void k_gather_values(int64_t nnz, const int32_t *__restrict__ src,
                                                            const double *__restrict__ in, double *__restrict__ val) {
  const int64_t stride = (int64_t)gridDim.x * NK_BLOCK;
  for (int64_t e = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x; e < nnz; e += stride) val[e] = in[src[e]];
}
// New values for a matrix that came in through nk_csr_create_from_csc(_rows), in the CSC order of that call (Julia:
// `nonzeros(A)` of a SparseMatrixCSC with the same pattern — what `f.jac(J, u, p)` refreshes every Newton step).
extern "C" int nk_csr_set_values_csc(nk_csr *A, const double *nzval, int64_t nnz_csc, int memspace) {
  NK_REQUIRE(A && nzval, "NULL argument");
  NK_REQUIRE(A->d_csc_src != nullptr, "the matrix was not created from CSC arrays");
  NK_REQUIRE(nnz_csc == A->csc_nnz, "nzval has %lld entries, the CSC pattern had %lld", (long long)nnz_csc, (long long)A->csc_nnz);
  nk_ctx *ctx = A->ctx;
  NK_HIP(hipSetDevice(ctx->device));
  const double *d_in = nzval;
  if (memspace != NK_DEVICE) {
    if (!A->d_csc_stage) NK_TRY(nk_dev_alloc(&A->d_csc_stage, (size_t)A->csc_nnz + 1));
    NK_HIP(hipMemcpyAsync(A->d_csc_stage, nzval, A->csc_nnz * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    d_in = A->d_csc_stage;
  }
  if (A->nnz > 0) {
    NK_LAUNCH(ctx, k_gather_values, dim3(nk_grid_for(A->nnz, NK_BLOCK * 4, 4096)), dim3(NK_BLOCK), A->nnz,
              (const int32_t *)A->d_csc_src, d_in, A->d_val);
    NK_HIP(hipGetLastError());
  }
  if (memspace != NK_DEVICE) NK_HIP(hipStreamSynchronize(ctx->stream));   // the caller's host array may change after the call
  A->t_values_stale = true; A->bounds_valid = false; A->bounds_pending = false;
  return NK_OK;
}
// the same with the library's default partition (contiguous row ranges, nk_partition_range with granule 1)
extern "C" int nk_csr_create_from_csc(nk_ctx *ctx, int64_t n, int64_t nnz, int index_bits, int index_base,
                                      const void *colptr, const void *rowval, const double *nzval, nk_csr **out) {
  NK_REQUIRE(ctx, "NULL argument");
  int64_t b = 0, e = n;
  NK_TRY(nk_partition_range(n, 1, ctx->nranks, ctx->rank, &b, &e));
  return nk_csr_create_from_csc_rows(ctx, n, nnz, index_bits, index_base, colptr, rowval, nzval, b, e - b, out);
}

extern "C" int nk_csr_destroy(nk_csr *A) {
  if (!A) return NK_OK;
  hipFree(A->d_rowptr);
  hipFree(A->d_col);
  hipFree(A->d_col16);
  hipFree(A->d_val);
  hipFree(A->d_rowblocks);
  hipFree(A->d_tperm);
  hipFree(A->d_ones);
  hipFree(A->d_diagpos);
  hipFree(A->d_gersh);
  hipFree(A->d_bounds);
  hipFree(A->d_csc_src);
  hipFree(A->d_csc_stage);
  hipFree(A->d_tz);
  hipFree(A->d_trecv);
  hipFree(A->d_role);
  hipFree(A->d_node);
  hipFree(A->d_xtmp);
  hipFree(A->d_ytmp);
  hipFree(A->d_color);
  hipFree(A->d_nnzcolor);
  hipFree(A->d_seed);
  hipFree(A->d_B);
  nk_powers_plan_destroy(A->pw);
  nk_halo_free(&A->halo);
  if (A->T) nk_csr_destroy(A->T);
  delete A;
  return NK_OK;
}
extern "C" int nk_csr_set_values(nk_csr *A, const double *vals, int memspace) {
  NK_REQUIRE(A && vals, "NULL argument");
  NK_HIP(hipMemcpyAsync(A->d_val, vals, A->nnz * sizeof(double),
                        memspace == NK_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, A->ctx->stream));
  if (memspace != NK_DEVICE) NK_HIP(hipStreamSynchronize(A->ctx->stream));
  A->t_values_stale = true; A->bounds_valid = false; A->bounds_pending = false;
  return NK_OK;
}
extern "C" int nk_csr_get_values(nk_csr *A, double *vals, int memspace) {
  NK_REQUIRE(A && vals, "NULL argument");
  NK_HIP(hipMemcpyAsync(vals, A->d_val, A->nnz * sizeof(double),
                        memspace == NK_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, A->ctx->stream));
  NK_HIP(hipStreamSynchronize(A->ctx->stream));
  return NK_OK;
}
extern "C" int nk_csr_info(nk_csr *A, int64_t *nrows_local, int64_t *n_global, int64_t *nnz, int64_t *n_halo) {
  NK_REQUIRE(A, "NULL argument");
  if (nrows_local) *nrows_local = A->nrows;
  if (n_global) *n_global = A->n_global;
  if (nnz) *nnz = A->nnz;
  if (n_halo) *n_halo = (int64_t)A->halo_gcols.size();
  return NK_OK;
}
extern "C" double *nk_csr_values_device(nk_csr *A) {
  if (A) A->raw_exposed = true;   // the caller may write the values behind the library's back: cached spectrum bounds are off
  return A ? A->d_val : nullptr;
}

// ----------------------------------------------------------------------------- Gershgorin bounds of the spectrum's real part
// Every eigenvalue lies in a disc |λ − a_ii| ≤ r_i = Σ_{j≠i} |a_ij|, so Re λ ∈ [min_i (a_ii − r_i), max_i (a_ii + r_i)] — the
// interval the s-step Arnoldi process places the shifts of its Newton basis on (nk_sstep.hip). Same traversal as the SpMV: a
// workgroup streams its row block's values and columns with lane-contiguous loads into LDS, then every lane walks one row
// in CSR order (so the row sums equal a sequential CPU sum bit for bit); max is exact in any order. Halo columns are never
// the diagonal. Output per block: {max_i −(a_ii − r_i), max_i (a_ii + r_i)}.
__global__ __launch_bounds__(NK_BLOCK) void k_csr_gershgorin(int nblk, int tile, const int4 *__restrict__ rowblocks,
                                                             const int32_t *__restrict__ rowptr,
                                                             const int32_t *__restrict__ col,
                                                             const double *__restrict__ val, double *__restrict__ part) {
  extern __shared__ double g_sv[];
  int32_t *g_sc = reinterpret_cast<int32_t *>(g_sv + tile);
  __shared__ double red[8];
  const int b = blockIdx.x;
  const int4 desc = rowblocks[b];
  const int r0 = desc.x, r1 = desc.y, p0 = desc.z, p1 = desc.w;
  const int nnzb = p1 - p0;
  double mlo = -INFINITY, mhi = -INFINITY;
  if (nnzb <= tile) {
    for (int k = threadIdx.x; k < nnzb; k += NK_BLOCK) {
      g_sv[k] = val[p0 + k];
      g_sc[k] = col[p0 + k];
    }
    __syncthreads();
    for (int r = r0 + threadIdx.x; r < r1; r += NK_BLOCK) {
      const int a = rowptr[r] - p0, e = rowptr[r + 1] - p0;
      double rad = 0.0, d = 0.0;
      for (int k = a; k < e; ++k) {
        const double v = g_sv[k];
        if (g_sc[k] == r) d += v;
        else rad += fabs(v);
      }
      mlo = fmax(mlo, -(d - rad));
      mhi = fmax(mhi, d + rad);
    }
  } else {  // a single long row
    double rad = 0.0, d = 0.0;
    for (int k = threadIdx.x; k < nnzb; k += NK_BLOCK) {
      const double v = val[p0 + k];
      if (col[p0 + k] == r0) d += v;
      else rad += fabs(v);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { rad += __shfl_xor(rad, o, 64); d += __shfl_xor(d, o, 64); }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = rad; red[4 + (threadIdx.x >> 6)] = d; }
    __syncthreads();
    rad = (red[0] + red[1]) + (red[2] + red[3]);
    d = (red[4] + red[5]) + (red[6] + red[7]);
    mlo = -(d - rad);
    mhi = d + rad;
    __syncthreads();
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { mlo = fmax(mlo, __shfl_xor(mlo, o, 64)); mhi = fmax(mhi, __shfl_xor(mhi, o, 64)); }
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = mlo; red[4 + (threadIdx.x >> 6)] = mhi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[b] = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    part[nblk + b] = fmax(fmax(red[4], red[5]), fmax(red[6], red[7]));
  }
}
__global__ __launch_bounds__(1024) void k_max2_final(int nblk, const double *__restrict__ part, double *__restrict__ out2) {
  __shared__ double red[32];
  double a = -INFINITY, c = -INFINITY;
  for (int i = threadIdx.x; i < nblk; i += 1024) { a = fmax(a, part[i]); c = fmax(c, part[nblk + i]); }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { a = fmax(a, __shfl_xor(a, o, 64)); c = fmax(c, __shfl_xor(c, o, 64)); }
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = a; red[16 + (threadIdx.x >> 6)] = c; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 16; ++w) { a = fmax(a, red[w]); c = fmax(c, red[16 + w]); }
    out2[0] = a;
    out2[1] = c;
  }
}
// a fill kernel left {max −lo, max hi} per block: the reduction into d_bounds waits for the first reader — the s-step form's
// begin kernel folds it in (nk_csr_take_pending_bounds), anyone else gets a k_max2_final launch
int nk_csr_bounds_from_partials(nk_csr *A, const double *d_part, int nblk) {
  if (!A->d_bounds) NK_TRY(nk_dev_alloc(&A->d_bounds, (size_t)2));
  A->bounds_part = d_part;
  A->bounds_nblk = nblk;
  A->bounds_pending = true;
  A->bounds_valid = true;
  return NK_OK;
}
static int csr_settle_bounds(nk_csr *A) {
  if (!A->bounds_pending) return NK_OK;
  NK_LAUNCH(A->ctx, k_max2_final, dim3(1), dim3(1024), A->bounds_nblk, A->bounds_part, A->d_bounds);
  NK_HIP(hipGetLastError());
  A->bounds_pending = false;
  return NK_OK;
}
bool nk_csr_take_pending_bounds(nk_csr *A, const double **part, int *nblk, double **dst) {
  if (!(A->bounds_valid && A->bounds_pending && !A->raw_exposed && A->d_bounds && A->ctx->nranks == 1)) return false;
  *part = A->bounds_part;
  *nblk = A->bounds_nblk;
  *dst = A->d_bounds;
  // (bounds_pending stays set until the caller has ENQUEUED the kernel that performs the reduction —
  //  nk_csr_commit_pending_bounds: a solve that fails before that must not leave d_bounds marked as reduced)
  return true;
}
void nk_csr_commit_pending_bounds(nk_csr *A) { A->bounds_pending = false; }
void nk_csr_invalidate_bounds(nk_csr *A) { A->bounds_valid = false; A->bounds_pending = false; }
double *nk_csr_bounds_word(nk_csr *A) {
  if (!A->d_bounds && nk_dev_alloc(&A->d_bounds, (size_t)2) != NK_OK) return nullptr;
  return A->d_bounds;
}
int nk_csr_gershgorin_dev(nk_csr *A, double *d_out2, const double **where) {
  nk_ctx *ctx = A->ctx;
  NK_REQUIRE(A->nblocks > 0, "Gershgorin bounds of an empty matrix");
  if (A->bounds_valid && !A->raw_exposed && A->d_bounds && ctx->nranks == 1) {   // the fill kernel computed the discs on the fly
    NK_TRY(csr_settle_bounds(A));
    *where = A->d_bounds;
    return NK_OK;
  }
  *where = d_out2;
  if (!A->d_gersh || A->gersh_cap < 2 * A->nblocks) {
    hipFree(A->d_gersh);
    A->d_gersh = nullptr;
    NK_TRY(nk_dev_alloc(&A->d_gersh, (size_t)2 * A->nblocks + 2));
    A->gersh_cap = 2 * A->nblocks + 2;
  }
  nk_prof_scope prof_(ctx, NK_K_OTHER, 12.0 * (double)A->nnz + 4.0 * (double)(A->nrows + 1));
  hipLaunchKernelGGL(k_csr_gershgorin, dim3(A->nblocks), dim3(NK_BLOCK), (size_t)A->tile * 12, ctx->stream, A->nblocks,
                     A->tile, (const int4 *)A->d_rowblocks, A->d_rowptr, A->d_col, A->d_val, A->d_gersh);
  NK_LAUNCH(ctx, k_max2_final, dim3(1), dim3(1024), A->nblocks, (const double *)A->d_gersh, d_out2);
  NK_HIP(hipGetLastError());
  return NK_OK;
}

int nk_csr_spmv_dev(nk_csr *A, const double *d_x, double *d_y, const int *d_skip, const double *d_out_scale,
                    const nk_spmv_epi *epi) {
  nk_ctx *ctx = A->ctx;
  nk_spmv_epi ep{};
  if (epi) ep = *epi;
  // halo overlap: interior row blocks run while the exchange is in flight on the communication stream
  const bool overlap = A->halo.active() && ctx->halo_overlap && ctx->nranks > 1 &&
                       A->nblocks_interior > 0 && A->nblocks_interior < A->nblocks;
  // peer-mapped halo plan: the exchange rides inside the SpMV launch (NK_PEER_UNFUSED=1 keeps the separate kernel)
  static const bool unfused = getenv("NK_PEER_UNFUSED") != nullptr;
  nk_spmv_xchg xc;
  const bool inkernel = A->halo.peer && !overlap && !unfused && A->halo.n_recv > 0 && A->halo.nsegs > 0 &&
                        A->halo.nsegs <= A->nblocks && A->halo.nsegs <= NK_BLOCK;
  if (inkernel) {
    nk_halo &H = A->halo;
    xc.segs = H.d_segs;
    xc.send_idx = H.d_send_idx;
    xc.err = nk_peer_err_ptr(ctx);
    xc.seq = ++H.seq;
    xc.nsegs = H.nsegs;
    xc.wait_from = A->nblocks_interior;
    H.d_recv = H.recv_buf[xc.seq & 1];
    ctx->stats.halo_exchanges++;
  } else if (A->halo.active()) {
    NK_TRY(overlap ? nk_halo_exchange_begin(ctx, &A->halo, d_x) : nk_halo_exchange(ctx, &A->halo, d_x));
  }
  ctx->stats.op_applies++;
  nk_prof_scope prof_(ctx, NK_K_SPMV, 12.0 * (double)A->nnz + 4.0 * (double)(A->nrows + 1) + 16.0 * (double)A->nrows);
  if (A->nblocks > 0) {
#define SPMV_LAUNCH(T, H, R)                                                                                      \
  NK_LAUNCH(ctx, (k_spmv_stream<T, H, R>), dim3(nb_), dim3(NK_BLOCK), nb_,                                         \
            (const int4 *)A->d_rowblocks + b0_, A->d_rowptr, A->d_col, A->d_val, d_x, A->halo.d_recv, (int32_t)A->nrows, \
            d_y, d_skip, d_out_scale, ep, (const int16_t *)A->d_col16, A->n16 - b0_, xc)
#define SPMV_TILES(H, R)                                  \
  if (A->tile == 512) SPMV_LAUNCH(512, H, R);             \
  else if (A->tile == 2048) SPMV_LAUNCH(2048, H, R);      \
  else if (A->tile == 4096) SPMV_LAUNCH(4096, H, R);      \
  else SPMV_LAUNCH(1024, H, R)
    const bool halo = A->halo.n_recv > 0;
    {
      const int nparts = overlap ? 2 : 1;
      for (int part = 0; part < nparts; ++part) {
        const int b0_ = (overlap && part == 1) ? A->nblocks_interior : 0;
        const int nb_ = overlap ? (part == 0 ? A->nblocks_interior : A->nblocks - A->nblocks_interior) : A->nblocks;
        if (overlap && part == 1) NK_TRY(nk_halo_exchange_end(ctx, &A->halo));
        if (A->variant == 2) {
          if (halo) { SPMV_TILES(true, false); } else { SPMV_TILES(false, false); }
        } else {
          if (halo) { SPMV_TILES(true, true); } else { SPMV_TILES(false, true); }
        }
      }
    }
#undef SPMV_TILES
#undef SPMV_LAUNCH
  }
  NK_HIP(hipGetLastError());
  return NK_OK;
}

// same pattern, partition and halo plan, own (zeroed) values
nk_csr_valstate nk_csr_get_valstate(const nk_csr *A) {
  nk_csr_valstate v;
  v.d_val = A->d_val; v.d_gersh = A->d_gersh; v.gersh_cap = A->gersh_cap;
  v.t_values_stale = A->t_values_stale; v.bounds_valid = A->bounds_valid; v.bounds_pending = A->bounds_pending;
  v.bounds_part = A->bounds_part; v.bounds_nblk = A->bounds_nblk;
  return v;
}
void nk_csr_set_valstate(nk_csr *A, const nk_csr_valstate &v) {
  A->d_val = v.d_val; A->d_gersh = v.d_gersh; A->gersh_cap = v.gersh_cap;
  A->t_values_stale = v.t_values_stale; A->bounds_valid = v.bounds_valid; A->bounds_pending = v.bounds_pending;
  A->bounds_part = v.bounds_part; A->bounds_nblk = v.bounds_nblk;
}
int nk_csr_alloc_values(nk_csr *A, double **out) {
  NK_TRY(nk_dev_alloc(out, (size_t)A->nnz + SPMV_TILE_MAX));
  NK_HIP(nk_memset(A->ctx, *out, 0, ((size_t)A->nnz + SPMV_TILE_MAX) * sizeof(double)));
  return NK_OK;
}
int nk_csr_clone_pattern(nk_csr *A, nk_csr **out) {
  NK_REQUIRE(A && out, "NULL argument");
  std::vector<int64_t> gc((size_t)A->nnz);
  for (int64_t k = 0; k < A->nnz; ++k) {
    const int32_t c = A->h_col[k];
    gc[k] = c < A->nrows ? A->row_begin + c : A->halo_gcols[c - A->nrows];
  }
  return nk_csr_create_local(A->ctx, A->nrows, A->n_global, A->row_begin, A->h_rowptr, gc, nullptr, out);
}

// ----------------------------------------------------------------------------- transpose
// y = Aᵀ x for a row-partitioned A (steepest.jl:75-77, trust_region.jl:410 need Jᵀ fu with a concrete J). The local block
// [owned | halo] is transposed as a whole: T = blockᵀ has nrows + n_halo rows. Its first nrows outputs are this rank's
// own contributions to y; the others belong to entries the neighbours own and travel back through the halo plan in
// reverse (what a rank receives in the forward exchange it now sends, and vice versa). The owners add them to y peer by
// peer in rank order — a fixed order, so the result is bitwise reproducible.
__global__ __launch_bounds__(NK_BLOCK) void k_permute_vals(int64_t nnz, const int32_t *__restrict__ perm,
                                                           const double *__restrict__ src, double *__restrict__ dst) {
  const int64_t i = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x;
  if (i < nnz) dst[i] = src[perm[i]];
}
__global__ __launch_bounds__(NK_BLOCK) void k_scatter_add(int64_t n, const int32_t *__restrict__ idx,
                                                          const double *__restrict__ v, double *__restrict__ y) {
  const int64_t i = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x;
  if (i < n) y[idx[i]] += v[i];  // a peer asks for each entry at most once: no two lanes share a target
}

static int build_transpose(nk_csr *A) {
  const int64_t n = A->nrows, nnz = A->nnz, nh = (int64_t)A->halo_gcols.size(), nt = n + nh;
  std::vector<int32_t> rp(nt + 1, 0), perm(nnz);
  std::vector<int64_t> gc(nnz);
  for (int64_t k = 0; k < nnz; ++k) rp[A->h_col[k] + 1]++;
  for (int64_t i = 0; i < nt; ++i) rp[i + 1] += rp[i];
  std::vector<int32_t> fill(rp.begin(), rp.end() - 1);
  for (int64_t r = 0; r < n; ++r)
    for (int32_t k = A->h_rowptr[r]; k < A->h_rowptr[r + 1]; ++k) {
      const int32_t pos = fill[A->h_col[k]]++;
      gc[pos] = r;
      perm[pos] = k;
    }
  NK_TRY(nk_csr_create_local(A->ctx, nt, nt, 0, rp, gc, nullptr, &A->T, true));
  NK_TRY(nk_dev_alloc(&A->d_tperm, (size_t)nnz));
  if (nnz) NK_HIP(nk_memcpy(A->ctx, A->d_tperm, perm.data(), nnz * sizeof(int32_t), hipMemcpyHostToDevice));
  if (nh) {
    NK_TRY(nk_dev_alloc(&A->d_tz, (size_t)nt + 1));
    NK_TRY(nk_dev_alloc(&A->d_trecv, (size_t)A->halo.n_send + 1));
  }
  A->t_values_stale = true; A->bounds_valid = false; A->bounds_pending = false;
  return NK_OK;
}

int nk_csr_spmv_t_dev(nk_csr *A, const double *d_x, double *d_y) {
  nk_ctx *ctx = A->ctx;
  if (!A->T) NK_TRY(build_transpose(A));
  if (A->t_values_stale && A->nnz) {
    const int grid = (int)((A->nnz + NK_BLOCK - 1) / NK_BLOCK);
    NK_LAUNCH(ctx, k_permute_vals, dim3(grid), dim3(NK_BLOCK), A->nnz, A->d_tperm, A->d_val, A->T->d_val);
    A->t_values_stale = false;
  }
  const int64_t n = A->nrows, nh = (int64_t)A->halo_gcols.size();
  if (nh == 0 && A->halo.n_send == 0) return nk_csr_spmv_dev(A->T, d_x, d_y, nullptr);
  // T x: (n + nh) outputs. (T's SpMV reads x as a vector of n + nh entries only through its column ids, all < n.)
  if (nh) {
    NK_TRY(nk_csr_spmv_dev(A->T, d_x, A->d_tz, nullptr));
    NK_TRY(nk_blas_copy(ctx, n, A->d_tz, d_y));
  } else {
    NK_TRY(nk_csr_spmv_dev(A->T, d_x, d_y, nullptr));
  }
  // reverse exchange: my halo slots (grouped by owner, rank order) go to their owners; theirs arrive in my send layout
  const int P = ctx->nranks;
  nk_halo &H = A->halo;
  std::vector<int64_t> so(P), sb(P), ro(P), rb(P);
  for (int p = 0; p < P; ++p) {
    so[p] = H.recv_off[p] * 8;
    sb[p] = (p == ctx->rank) ? 0 : H.recv_cnt[p] * 8;
    ro[p] = H.send_off[p] * 8;
    rb[p] = (p == ctx->rank) ? 0 : H.send_cnt[p] * 8;
  }
  if (!A->d_trecv && H.n_send) NK_TRY(nk_dev_alloc(&A->d_trecv, (size_t)H.n_send + 1));
  ctx->stats.halo_exchanges++;
  NK_TRY(nk_comm_alltoallv(ctx, nh ? (const void *)(A->d_tz + n) : (const void *)d_y, so.data(), sb.data(), A->d_trecv,
                           ro.data(), rb.data()));
  for (int p = 0; p < P; ++p) {  // owners accumulate peer by peer, in rank order
    const int64_t c = (p == ctx->rank) ? 0 : H.send_cnt[p];
    if (c == 0) continue;
    NK_LAUNCH(ctx, k_scatter_add, dim3((unsigned)((c + NK_BLOCK - 1) / NK_BLOCK)), dim3(NK_BLOCK), c,
              (const int32_t *)H.d_send_idx + H.send_off[p], (const double *)A->d_trecv + H.send_off[p], d_y);
  }
  NK_HIP(hipGetLastError());
  return NK_OK;
}

// out_j = Σ_i A_ij² = ((A∘A)ᵀ·1)_j — the transposed product with squared values and a vector of ones, so it inherits the
// fixed summation order and the rank-ordered reverse exchange of nk_csr_spmv_t_dev (bitwise reproducible, any partition)
__global__ __launch_bounds__(NK_BLOCK) void k_permute_vals_sq(int64_t nnz, const int32_t *__restrict__ perm,
                                                              const double *__restrict__ src, double *__restrict__ dst) {
  const int64_t i = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x;
  if (i < nnz) { const double v = src[perm[i]]; dst[i] = v * v; }
}
int nk_csr_colsumsq_dev(nk_csr *A, double *d_out) {
  nk_ctx *ctx = A->ctx;
  if (!A->T) NK_TRY(build_transpose(A));
  if (!A->d_ones) {
    NK_TRY(nk_dev_alloc(&A->d_ones, (size_t)A->nrows + 1));
    NK_TRY(nk_blas_fill(ctx, A->nrows + 1, 1.0, A->d_ones));
  }
  if (A->nnz) {
    const int grid = (int)((A->nnz + NK_BLOCK - 1) / NK_BLOCK);
    NK_LAUNCH(ctx, k_permute_vals_sq, dim3(grid), dim3(NK_BLOCK), A->nnz, A->d_tperm, A->d_val, A->T->d_val);
  }
  A->t_values_stale = false;                    // T holds the squares for this one product …
  const int st = nk_csr_spmv_t_dev(A, A->d_ones, d_out);
  A->t_values_stale = true; A->bounds_valid = false; A->bounds_pending = false;                     // … and must be refreshed before the next Aᵀ x
  return st;
}
extern "C" int nk_csr_colsumsq(nk_csr *A, double *out, int memspace) {
  NK_REQUIRE(A && out, "NULL argument");
  NK_HIP(hipSetDevice(A->ctx->device));
  if (memspace == NK_DEVICE) return nk_csr_colsumsq_dev(A, out);
  if (!A->d_ytmp) NK_TRY(nk_dev_alloc(&A->d_ytmp, (size_t)A->nrows + 1));
  NK_TRY(nk_csr_colsumsq_dev(A, A->d_ytmp));
  NK_HIP(hipMemcpyAsync(out, A->d_ytmp, A->nrows * sizeof(double), hipMemcpyDeviceToHost, A->ctx->stream));
  NK_HIP(hipStreamSynchronize(A->ctx->stream));
  return NK_OK;
}

// A ← A + σ I: `dampen_jacobian!!(J_cache, J, D::Number)` (descent/damped_newton.jl:352-366) on the device CSR
__global__ __launch_bounds__(NK_BLOCK) void k_add_diag(int64_t nrows, const int32_t *__restrict__ pos, double sigma,
                                                       const double *__restrict__ m, double *__restrict__ val) {
  const int64_t r = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x;
  if (r < nrows) val[pos[r]] += m ? sigma * m[r] : sigma;
}
int nk_csr_add_to_diagonal_dev(nk_csr *A, double sigma, const double *d_m) {
  if (!A->d_diagpos) {
    std::vector<int32_t> pos((size_t)A->nrows);
    for (int64_t r = 0; r < A->nrows; ++r) {
      int32_t found = -1;
      for (int32_t k = A->h_rowptr[r]; k < A->h_rowptr[r + 1]; ++k)
        if (A->h_col[k] == r) { found = k; break; }
      NK_REQUIRE(found >= 0, "row %lld stores no diagonal entry: the damping J + sigma I needs one in the pattern", (long long)r);
      pos[r] = found;
    }
    NK_TRY(nk_dev_alloc(&A->d_diagpos, (size_t)A->nrows + 1));
    if (A->nrows) NK_HIP(nk_memcpy(A->ctx, A->d_diagpos, pos.data(), A->nrows * sizeof(int32_t), hipMemcpyHostToDevice));
  }
  if (A->nrows)
    NK_LAUNCH(A->ctx, k_add_diag, dim3((unsigned)((A->nrows + NK_BLOCK - 1) / NK_BLOCK)), dim3(NK_BLOCK), A->nrows,
              (const int32_t *)A->d_diagpos, sigma, d_m, A->d_val);
  NK_HIP(hipGetLastError());
  A->t_values_stale = true; A->bounds_valid = false; A->bounds_pending = false;
  return NK_OK;
}

// ----------------------------------------------------------------------------- assembled normal matrix JᵀJ + λ·diag(d)
// (DampedNewtonDescent with a FACTORISING linear solver: the reference solves min ‖[J; √(λDᵀD)] x − [f; 0]‖ by QR,
// descent/damped_newton.jl:258-296; the device factorises the normal equations of the same problem, whose matrix it has to
// assemble.) Symbolic product once per pattern on the host — for every non-zero (i, j) of JᵀJ the list of pairs (p, q) of J's
// non-zeros with row(p) = row(q), col(p) = i, col(q) = j, in row order — then one thread per output non-zero sums its
// products in that fixed order. Single rank.
struct nk_normal_plan {
  nk_csr *N = nullptr;          // the pattern of JᵀJ (owned)
  int32_t *d_ptr = nullptr;     // nnz(N) + 1 offsets into the pair lists
  int32_t *d_pa = nullptr, *d_pb = nullptr;
  int32_t *d_diagrow = nullptr; // row of a diagonal non-zero, −1 elsewhere
};
__global__ __launch_bounds__(NK_BLOCK) void k_normal_values(int64_t nnzN, const int32_t *__restrict__ ptr,
                                                            const int32_t *__restrict__ pa, const int32_t *__restrict__ pb,
                                                            const int32_t *__restrict__ diagrow, const double *__restrict__ val,
                                                            double lambda, const double *__restrict__ d, double *__restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x;
  if (e >= nnzN) return;
  double s = 0.0;
  for (int32_t t = ptr[e]; t < ptr[e + 1]; ++t) s += val[pa[t]] * val[pb[t]];
  const int32_t dr = diagrow[e];
  if (dr >= 0 && d != nullptr) s += lambda * d[dr];
  out[e] = s;
}
void nk_normal_plan_destroy(nk_normal_plan *Pn) {
  if (!Pn) return;
  if (Pn->N) nk_csr_destroy(Pn->N);
  hipFree(Pn->d_ptr); hipFree(Pn->d_pa); hipFree(Pn->d_pb); hipFree(Pn->d_diagrow);
  delete Pn;
}
int nk_normal_plan_create(nk_csr *J, nk_normal_plan **out) {
  nk_ctx *ctx = J->ctx;
  NK_REQUIRE(ctx->nranks == 1 && J->halo_gcols.empty(), "the assembled normal matrix is built on one rank");
  const int64_t n = J->nrows;
  // CSC view of J: for every column the (row, position) pairs in row order
  std::vector<int32_t> cptr(n + 1, 0), crow(J->nnz), cpos(J->nnz);
  for (int64_t k = 0; k < J->nnz; ++k) cptr[J->h_col[k] + 1]++;
  for (int64_t i = 0; i < n; ++i) cptr[i + 1] += cptr[i];
  {
    std::vector<int32_t> fill(cptr.begin(), cptr.end() - 1);
    for (int64_t r = 0; r < n; ++r)
      for (int32_t k = J->h_rowptr[r]; k < J->h_rowptr[r + 1]; ++k) {
        const int32_t pos = fill[J->h_col[k]]++;
        crow[pos] = (int32_t)r;
        cpos[pos] = k;
      }
  }
  std::vector<int32_t> rp(n + 1, 0), ptr, pa, pb, diagrow;
  std::vector<int64_t> gc;
  std::vector<std::pair<int32_t, std::pair<int32_t, int32_t>>> tmp;  // (j, (p, q)) of one output row
  ptr.push_back(0);
  for (int64_t i = 0; i < n; ++i) {
    tmp.clear();
    for (int32_t c = cptr[i]; c < cptr[i + 1]; ++c) {
      const int32_t r = crow[c], p = cpos[c];
      for (int32_t q = J->h_rowptr[r]; q < J->h_rowptr[r + 1]; ++q) tmp.push_back({J->h_col[q], {p, q}});
    }
    std::stable_sort(tmp.begin(), tmp.end(), [](const auto &a, const auto &b) { return a.first < b.first; });
    bool has_diag = false;
    for (size_t t = 0; t < tmp.size();) {
      const int32_t j = tmp[t].first;
      if (j > i && !has_diag) {  // a structurally missing diagonal still receives the damping
        gc.push_back(i); diagrow.push_back((int32_t)i); ptr.push_back((int32_t)pa.size());
        has_diag = true;
      }
      for (; t < tmp.size() && tmp[t].first == j; ++t) { pa.push_back(tmp[t].second.first); pb.push_back(tmp[t].second.second); }
      gc.push_back(j);
      diagrow.push_back(j == i ? (int32_t)i : -1);
      if (j == i) has_diag = true;
      ptr.push_back((int32_t)pa.size());
    }
    if (!has_diag) { gc.push_back(i); diagrow.push_back((int32_t)i); ptr.push_back((int32_t)pa.size()); }
    rp[i + 1] = (int32_t)gc.size();
  }
  nk_normal_plan *Pn = new nk_normal_plan();
  auto guard = nk_make_guard(Pn, [](nk_normal_plan *q) { nk_normal_plan_destroy(q); });
  NK_TRY(nk_csr_create_local(ctx, n, n, 0, rp, gc, nullptr, &Pn->N, true));
  const size_t nnzN = gc.size(), np = pa.size();
  NK_TRY(nk_dev_alloc(&Pn->d_ptr, nnzN + 1));
  NK_TRY(nk_dev_alloc(&Pn->d_pa, np + 1));
  NK_TRY(nk_dev_alloc(&Pn->d_pb, np + 1));
  NK_TRY(nk_dev_alloc(&Pn->d_diagrow, nnzN + 1));
  NK_HIP(nk_memcpy(ctx, Pn->d_ptr, ptr.data(), (nnzN + 1) * sizeof(int32_t), hipMemcpyHostToDevice));
  if (np) NK_HIP(nk_memcpy(ctx, Pn->d_pa, pa.data(), np * sizeof(int32_t), hipMemcpyHostToDevice));
  if (np) NK_HIP(nk_memcpy(ctx, Pn->d_pb, pb.data(), np * sizeof(int32_t), hipMemcpyHostToDevice));
  NK_HIP(nk_memcpy(ctx, Pn->d_diagrow, diagrow.data(), nnzN * sizeof(int32_t), hipMemcpyHostToDevice));
  *out = guard.release();
  return NK_OK;
}
nk_csr *nk_normal_plan_matrix(nk_normal_plan *Pn) { return Pn->N; }
// N ← JᵀJ + λ·diag(d) (d may be nullptr)
int nk_normal_plan_values(nk_normal_plan *Pn, nk_csr *J, double lambda, const double *d_diag) {
  nk_csr *N = Pn->N;
  if (N->nnz) {
    NK_LAUNCH(J->ctx, k_normal_values, dim3((unsigned)((N->nnz + NK_BLOCK - 1) / NK_BLOCK)), dim3(NK_BLOCK), N->nnz,
              (const int32_t *)Pn->d_ptr, (const int32_t *)Pn->d_pa, (const int32_t *)Pn->d_pb, (const int32_t *)Pn->d_diagrow,
              (const double *)J->d_val, lambda, d_diag, N->d_val);
    NK_HIP(hipGetLastError());
  }
  N->t_values_stale = true; N->bounds_valid = false; N->bounds_pending = false;
  return NK_OK;
}

static int stage_in(nk_csr *A, const double *x, int memspace, const double **dx) {
  if (memspace == NK_DEVICE) {
    *dx = x;
    return NK_OK;
  }
  if (!A->d_xtmp) NK_TRY(nk_dev_alloc(&A->d_xtmp, (size_t)A->nrows + 1));
  NK_HIP(hipMemcpyAsync(A->d_xtmp, x, A->nrows * sizeof(double), hipMemcpyHostToDevice, A->ctx->stream));
  *dx = A->d_xtmp;
  return NK_OK;
}

static int spmv_any(nk_csr *A, const double *x, double *y, int memspace, bool transpose) {
  NK_REQUIRE(A && x && y, "NULL argument");
  NK_HIP(hipSetDevice(A->ctx->device));
  const double *dx;
  NK_TRY(stage_in(A, x, memspace, &dx));
  double *dy = y;
  if (memspace != NK_DEVICE) {
    if (!A->d_ytmp) NK_TRY(nk_dev_alloc(&A->d_ytmp, (size_t)A->nrows + 1));
    dy = A->d_ytmp;
  }
  NK_TRY(transpose ? nk_csr_spmv_t_dev(A, dx, dy) : nk_csr_spmv_dev(A, dx, dy, nullptr));
  if (memspace != NK_DEVICE) {
    NK_HIP(hipMemcpyAsync(y, dy, A->nrows * sizeof(double), hipMemcpyDeviceToHost, A->ctx->stream));
    NK_HIP(hipStreamSynchronize(A->ctx->stream));
  }
  return NK_OK;
}
extern "C" int nk_spmv(nk_csr *A, const double *x, double *y, int memspace) { return spmv_any(A, x, y, memspace, false); }
extern "C" int nk_spmv_t(nk_csr *A, const double *x, double *y, int memspace) { return spmv_any(A, x, y, memspace, true); }
