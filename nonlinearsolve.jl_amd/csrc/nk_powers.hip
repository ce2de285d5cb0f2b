// Matrix powers with the matrix RESIDENT ON THE CHIP: Y_p = os_p·(A Y_{p−1} − θ_p Y_{p−1}), p = 0 … s−1, Y_{−1} = x, in ONE
// launch — the s operator applications an s-step Arnoldi block makes back to back (nk_sstep.hip) with nothing in between.
//
// Why: the streaming SpMV (nk_csr.hip) reads the whole matrix for every application — 15 × 74 MB per block at 1024², 2.2 GB of
// the 3.9 GB a fixed-work Newton step moves. An MI355X has 256 CUs × 512 KB of vector registers = 128 MB; a matrix of up to
// ≈ 8 M non-zeros fits there. So: one 1024-thread workgroup per CU owns a BAND of 1024·RPT consecutive rows and loads its
// slice of val / col into registers ONCE per launch (thread t holds rows t, t + 1024, … of its band, ≤ W entries each, in CSR
// order). The band's slice of the current vector lives in LDS ([halo above | own rows | halo below], two buffers); a power is
// RPT·W LDS gathers and multiply-adds per thread, one 8-byte store per row into the basis column — and the only data that
// crosses workgroups are the ≤ 1024 rows either neighbour band reads (a grid line of the 5-point stencil): the boundary rows
// are stored write-through (`sc1`), a per-band flag word is released behind them, the neighbour polls it and reads the rows
// with `sc1` loads (MI355X_MICROARCH.md, hand-off forms). No grid-wide synchronisation, no second read of the matrix.
// HBM traffic per power: the 8 n bytes of the new column. Row sums are formed exactly as the streaming kernel forms them
// (products rounded, added in CSR order from 0.0, same epilogue), so the columns are BIT-IDENTICAL to s streaming launches.
//
// The hand-off is written for gfx950 and says so: boundary rows leave as `sc1` stores (RELAXED agent-scope atomics → write-through
// past the XCD's L2), the wavefront drains them with `s_waitcnt vmcnt(0)` (on gfx9 vmcnt counts stores as well as loads), a barrier,
// then the RELAXED agent-scope flag store; the reader polls with `sc1` loads and reads the rows with `sc1` loads. That is a
// release / acquire pair spelled out instruction by instruction: a compiler-emitted agent-scope release would write back the whole
// L2 (`buffer_wbl2`) once per power and band — measured in rounds 3–5 as the slow form (MI355X_MICROARCH.md, hand-off forms) —, so
// the contract is tied to this target, checked by tests that compare every word under uneven load (tests/test_gpu_powers.py).
// Eligibility (host, once per pattern — nk_csr_powers_plan): one rank, no halo; rows ≤ #CUs × 1024 × RPT_max; every row ≤ W
// entries; every column of band b inside [first row of b − 1024, last row of b + 1024]. Everything else keeps the streaming
// kernel. All workgroups must be resident at once (grid ≤ #CUs, one per CU by its LDS footprint); every wait is bounded by a
// wall-clock time-out that raises the plan's error word (host-visible), after which the object falls back for good.
#include <algorithm>
#include <stdlib.h>
#include <vector>

#include "nk_internal.h"

constexpr int PW_T = 1024;          // threads per workgroup = rows per slice
constexpr int PW_HALO = 1024;       // rows a band may read from either neighbour
constexpr int PW_FLAG_STRIDE = 16;  // uint64 words between two bands' flags (128 B)
constexpr int PW_NXCD = 8;

struct pw_args {
  int nrows, nb, s, variant;   // variant bit 8 (development hook): band 0 never publishes — its neighbour times out
  const int32_t *rowptr, *col;
  const double *val;
  const double *x0;
  double *Y;
  int64_t ldy;
  const double *scal_first, *scal_rest, *theta;
  const int *d_skip;
  uint64_t *flags;
  uint64_t base;   // band b has published power p ⇔ flags[b] ≥ base + p + 1 (base grows with every launch: no reset)
  uint64_t *err;   // [0] time-outs (sticky), [1] bound of a wait in ticks of the 100 MHz wall clock
  // GEN = 1: the matrix is not stored — the 5-point Bratu Jacobian c_lap·Δ_h − diag(d) on an ns × ns grid, rows lexicographic
  int ns;
  double c_lap;
  const double *diag;
  // k_spmv_powers_seg: SEG segments of seg_len rows each (nb bands per segment), bands of a segment on a ring or not
  int seg_len, ring;
  // several ranks (k_spmv_powers<…, PEER = true>): the matrix is row-partitioned, rank r's first band and rank r − 1's last band
  // are neighbours like any two bands — their boundary slices change hands through the peer-mapped arenas (system-scope stores
  // into the neighbour's receive area, a flag released behind them; what nk_csr.hip's in-launch halo exchange does per
  // operator application, here once per power inside ONE launch)
  struct {
    int up, dn;                          // there is a rank above / below
    const int32_t *halo_vl;              // halo slot → row relative to this rank's first row (negative: above)
    double *push_up, *push_dn;           // the neighbours' receive areas for my slices (4 buffers of PW_T), in MY address space
    uint64_t *flag_up, *flag_dn;         // the neighbours' flags for me
    const double *recv_up, *recv_dn;     // my receive areas (4 buffers of PW_T each)
    const uint64_t *myflag_up, *myflag_dn;
    uint64_t *err;                       // the arena's time-out counter (nk_peer_timeout)
    int ebuf;                            // buffer pair of this launch: (epoch & 1)·2 — a neighbour is never two launches ahead
  } pr;
};

// development: phase time stamps of every band's wavefront 0 (100 MHz wall clock), builds with -DNK_PW_STAMPS only (make stamps;
// tools/pw_stamps.py): [band][power][PW_NST]
constexpr int PW_NST = 12, PW_STP = 16;
__device__ unsigned long long *g_pw_stamp = nullptr;
#ifdef NK_PW_STAMPS
#define PW_STAMP(p, i) do { if (t == 0 && (p) < PW_STP) s_pw_st[(p) * PW_NST + (i)] = wall_clock64(); } while (0)
#else
#define PW_STAMP(p, i) do { } while (0)
#endif

__device__ __forceinline__ int pw_band_of(int bid, int nb) {   // block b runs on XCD b % 8: neighbouring bands share an L2
  const int q = nb / PW_NXCD, r = nb % PW_NXCD;
  const int x = bid % PW_NXCD, k = bid / PW_NXCD;
  return x * q + (x < r ? x : r) + k;
}
__device__ __forceinline__ void pw_store_sc1(double *p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double pw_load_sc1(const double *p) {
  return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED,
                                                           __HIP_MEMORY_SCOPE_AGENT));
}
// one lane: until *f ≥ target; false on a time-out (ours or anybody's)
__device__ __forceinline__ bool pw_wait(const uint64_t *f, uint64_t target, uint64_t *err) {
  if (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) return true;
  const unsigned long long t0 = wall_clock64(), lim = err[1];
  for (unsigned it = 1;; ++it) {
    __builtin_amdgcn_s_sleep(1);
    if (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) return true;
    if ((it & 63u) == 0) {
      if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) return false;
      if (wall_clock64() - t0 > lim) {   // (a plain store: the word lives in pinned host memory, sticky, any non-zero value will do)
        __hip_atomic_store(err, (uint64_t)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return false;
      }
    }
  }
}
// a neighbour RANK's flag (system scope); same contract as pw_wait, the bound is the arena's
__device__ __forceinline__ bool pw_wait_peer(const uint64_t *f, uint64_t target, uint64_t *perr, uint64_t *err) {
  if (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) >= target) return true;
  const unsigned long long t0 = wall_clock64(), lim = nk_peer_timeout(perr);
  for (unsigned it = 1;; ++it) {
    __builtin_amdgcn_s_sleep(1);
    if (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) >= target) return true;
    if ((it & 63u) == 0) {
      if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) return false;
      if (__hip_atomic_load(perr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 4 || wall_clock64() - t0 > lim) {
        atomicAdd((unsigned long long *)perr, 1ull);
        __hip_atomic_store(err, (uint64_t)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return false;
      }
    }
  }
}
__device__ __forceinline__ void pw_store_sys(double *p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ double pw_load_sys(const double *p) {
  return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED,
                                                           __HIP_MEMORY_SCOPE_SYSTEM));
}
// order in which a power visits the band's slices: the two the neighbours read first, the interior ones behind them
template <int RPT>
__device__ __forceinline__ constexpr int pw_slice(int q) {
  return q == 0 ? 0 : (q == 1 ? RPT - 1 : q - 1);
}

template <int RPT, int W, int GEN, bool PEER = false>
__global__ __launch_bounds__(PW_T) void k_spmv_powers(const pw_args a) {
  if (a.d_skip != nullptr && *a.d_skip != 0) return;   // (the flag is replicated: every rank takes the same branch)
  extern __shared__ double pw_x[];
  __shared__ int s_abort;
#ifdef NK_PW_STAMPS
  __shared__ unsigned long long s_pw_st[PW_STP * PW_NST];
  for (int i = threadIdx.x; i < PW_STP * PW_NST; i += PW_T) s_pw_st[i] = 0;
#endif
  constexpr int RB = PW_T * RPT, XN = RB + 2 * PW_HALO;
  constexpr int NBND = RPT > 1 ? 2 : 1;   // slices a neighbour reads
  const int t = threadIdx.x;
  const int b = pw_band_of(blockIdx.x, a.nb);
  const int r0 = b * RB;
  double *xa = pw_x, *xb = pw_x + XN;
  // PEER: the first band's upper and the last band's lower neighbour are bands of other ranks
  const bool pup = PEER && b == 0 && a.pr.up != 0, pdn = PEER && b == a.nb - 1 && a.pr.dn != 0;

  // ---- the band's matrix slice → registers (once per launch). A slice's entries are contiguous in val / col: the workgroup
  // streams them with lane-contiguous loads into LDS (the vector buffers are idle until the first power) and every thread then
  // picks its row's ≤ W entries from there — a thread reading its own row straight from memory touches every 128-byte line of
  // the slice W times over (measured: 17.7 → 12 µs for this phase at 1024²). val / col are padded by a tile: k + j stays in bounds.
  double v[RPT][W];
  int ci[RPT][W];
  int len[RPT];
  if constexpr (GEN == 1) {
    // matrix-free: the band's rows of the stencil Jacobian are GENERATED into the same register layout (entries in CSR order:
    // south, west, centre, east, north where they exist) — the matrix-free operator's s applications in one launch as well
    static_assert(W >= 5, "the 5-point stencil needs five slots per row");
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      const int r = r0 + t + PW_T * i;
      const int rc = r < a.nrows ? r : a.nrows - 1;
      const int gj = rc / a.ns, gi = rc - gj * a.ns;
      const double dd = a.diag[rc];
      const int base = rc - (r0 - PW_HALO), own = PW_HALO + t + PW_T * i;
      const bool hs = gj > 0, hw = gi > 0, he = gi + 1 < a.ns, hn = gj + 1 < a.ns;
      const int pW = hs ? 1 : 0, pC = pW + (hw ? 1 : 0), pE = pC + 1, pN = pE + (he ? 1 : 0), n = pN + (hn ? 1 : 0);
      const double off = -a.c_lap, ctr = 4.0 * a.c_lap - dd;
#pragma unroll
      for (int j = 0; j < W; ++j) {   // (selects only: the slot index of every stencil point is a prefix count)
        double vv = 0.0;
        int cc = own;
        if (hs && j == 0) { vv = off; cc = base - a.ns; }
        if (hw && j == pW) { vv = off; cc = base - 1; }
        if (j == pC) { vv = ctr; cc = base; }
        if (he && j == pE) { vv = off; cc = base + 1; }
        if (hn && j == pN) { vv = off; cc = base + a.ns; }
        v[i][j] = vv;
        ci[i][j] = cc;
      }
      len[i] = r < a.nrows ? n : 0;
    }
  } else if constexpr (W > 8) {
    // wide rows (W = 16): the slice does not fit the LDS staging area — every thread reads its own row
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      const int r = r0 + t + PW_T * i;
      const int rc = r < a.nrows ? r : a.nrows - 1;
      const int k0 = a.rowptr[rc], k1 = a.rowptr[rc + 1];
      len[i] = r < a.nrows ? k1 - k0 : 0;
#pragma unroll
      for (int j = 0; j < W; ++j) {
        v[i][j] = a.val[k0 + j];
        int c = a.col[k0 + j];
        if (PEER && (pup || pdn) && j < len[i] && c >= a.nrows) c = a.pr.halo_vl[c - a.nrows];
        ci[i][j] = (j < len[i]) ? c - (r0 - PW_HALO) : PW_HALO + t + PW_T * i;
      }
    }
  } else {
    double *sv = pw_x;                                             // PW_T·W values …
    int *sc = reinterpret_cast<int *>(pw_x + (size_t)PW_T * W);    // … and as many column ids
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      const int rfirst = r0 + PW_T * i;
      const int r = rfirst + t;
      const int rc = r < a.nrows ? r : a.nrows - 1;
      const int k0 = a.rowptr[rc], k1 = a.rowptr[rc + 1];
      const int rl = rfirst < a.nrows ? (rfirst + PW_T <= a.nrows ? rfirst + PW_T : a.nrows) : rfirst;   // end row of the slice
      const int kb = rfirst < a.nrows ? a.rowptr[rfirst] : 0, ke = rfirst < a.nrows ? a.rowptr[rl] : 0;
      len[i] = r < a.nrows ? k1 - k0 : 0;
#pragma unroll
      for (int j = 0; j < W; ++j) {
        const int e = t + PW_T * j;
        if (kb + e < ke) { sv[e] = a.val[kb + e]; sc[e] = a.col[kb + e]; }
      }
      __syncthreads();
#pragma unroll
      for (int j = 0; j < W; ++j) {
        const int e = (j < len[i]) ? k0 - kb + j : 0;
        const double vv = sv[e];
        int c = sc[e];
        if (PEER && (pup || pdn) && j < len[i] && c >= a.nrows) c = a.pr.halo_vl[c - a.nrows];   // a halo slot → its row next door
        v[i][j] = (j < len[i]) ? vv : 0.0;
        ci[i][j] = (j < len[i]) ? c - (r0 - PW_HALO) : PW_HALO + t + PW_T * i;   // (slots past the row's end: never summed)
      }
      __syncthreads();
    }
  }
  for (int idx = t; idx < XN; idx += PW_T) {
    const int g = r0 - PW_HALO + idx;
    xa[idx] = (g >= 0 && g < a.nrows) ? a.x0[g] : 0.0;
    xb[idx] = 0.0;
  }
  if (t == 0) s_abort = 0;
  __syncthreads();
  if (PEER && (pup || pdn)) {
    // the start vector's rows next door: my first / last slice goes to the neighbour rank, its last / first comes back
    // (buffer ebuf + 0; "power −1" is published as the launch's base value itself)
    if (pup) pw_store_sys(a.pr.push_up + (size_t)a.pr.ebuf * PW_T + t, a.x0[t]);
    if (pdn) pw_store_sys(a.pr.push_dn + (size_t)a.pr.ebuf * PW_T + t, a.x0[a.nrows - PW_T + t]);
    __threadfence_system();
    __syncthreads();
    if (t == 0 && pup) __hip_atomic_store(a.pr.flag_up, a.base, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (t == 64 && pdn) __hip_atomic_store(a.pr.flag_dn, a.base, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (t == 0 && pup && !pw_wait_peer(a.pr.myflag_up, a.base, a.pr.err, a.err)) s_abort = 1;
    if (t == 64 && pdn && !pw_wait_peer(a.pr.myflag_dn, a.base, a.pr.err, a.err)) s_abort = 1;
    __syncthreads();
    if (s_abort) return;
    if (pup) xa[t] = pw_load_sys(a.pr.recv_up + (size_t)a.pr.ebuf * PW_T + t);
    if (pdn) xa[PW_HALO + RB + t] = pw_load_sys(a.pr.recv_dn + (size_t)a.pr.ebuf * PW_T + t);
    __syncthreads();
  }

  const bool shifted = a.theta != nullptr;
  for (int p = 0; p < a.s; ++p) {
    const double *xin = (p & 1) ? xb : xa;
    double *xout = (p & 1) ? xa : xb;
    const double *osp = (p == 0) ? a.scal_first : a.scal_rest;
    const double os = osp ? *osp : 1.0;
    const double th = shifted ? a.theta[p] : 0.0;
    double *ycol = a.Y + (int64_t)p * a.ldy;
    const bool pub = p + 1 < a.s;
    PW_STAMP(p, 0);
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
      const int i = pw_slice<RPT>(q);
      const int lr = t + PW_T * i, r = r0 + lr;
      double sum = 0.0;
#pragma unroll
      for (int j = 0; j < W; ++j) {
        const double pr = v[i][j] * xin[ci[i][j]];
        sum = (j < len[i]) ? sum + pr : sum;
      }
      double out = shifted ? sum - th * xin[PW_HALO + lr] : sum;
      out = osp ? os * out : out;
      if (r < a.nrows) {
        xout[PW_HALO + lr] = out;
        if (q < NBND) pw_store_sc1(ycol + r, out);   // rows a neighbour band reads: write-through
        else ycol[r] = out;
      }
      if (PEER && pub) {   // … and the slice a neighbour RANK reads: into its receive area (buffer ebuf + (p + 1) mod 2)
        const size_t boff = (size_t)(a.pr.ebuf + ((p + 1) & 1)) * PW_T + t;
        if (pup && i == 0) pw_store_sys(a.pr.push_up + boff, out);
        if (pdn && i == RPT - 1) pw_store_sys(a.pr.push_dn + boff, out);
      }
      if (q == NBND - 1) PW_STAMP(p, 1);   // boundary slices computed, their stores issued
      if (pub && (a.variant & 255) == 0 && q == NBND - 1) {   // publish as early as possible: behind the boundary slices
        if (PEER && (pup || pdn)) __threadfence_system();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PW_STAMP(p, 2);                      // … drained
        __syncthreads();
        PW_STAMP(p, 3);                      // … by every wavefront
        if (t == 0 && !((a.variant & 256) && b == 0))
          __hip_atomic_store(a.flags + (size_t)b * PW_FLAG_STRIDE, a.base + (uint64_t)p + 1, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
        if (PEER) {
          if (t == 64 && pup) __hip_atomic_store(a.pr.flag_up, a.base + (uint64_t)p + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
          if (t == 128 && pdn) __hip_atomic_store(a.pr.flag_dn, a.base + (uint64_t)p + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
    }
    PW_STAMP(p, 4);                          // interior slices computed, their stores issued
    if (!pub) break;
    if ((a.variant & 255) != 0) {   // publish behind the whole band: the interior rows overlap the boundary rows' write-through
      if (PEER && (pup || pdn)) __threadfence_system();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (t == 0)
        __hip_atomic_store(a.flags + (size_t)b * PW_FLAG_STRIDE, a.base + (uint64_t)p + 1, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      if (PEER) {
        if (t == 64 && pup) __hip_atomic_store(a.pr.flag_up, a.base + (uint64_t)p + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        if (t == 128 && pdn) __hip_atomic_store(a.pr.flag_dn, a.base + (uint64_t)p + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    // ---- the neighbours' boundary rows of power p → the halo parts of the next input buffer
    if (t == 0) {
      if (b > 0) {
        if (!pw_wait(a.flags + (size_t)(b - 1) * PW_FLAG_STRIDE, a.base + (uint64_t)p + 1, a.err)) s_abort = 1;
      } else if (PEER && pup) {
        if (!pw_wait_peer(a.pr.myflag_up, a.base + (uint64_t)p + 1, a.pr.err, a.err)) s_abort = 1;
      }
    }
    PW_STAMP(p, 5);                          // the upper neighbour's flag seen (wavefront 0 polls it)
    if (t == 64) {
      if (b + 1 < a.nb) {
        if (!pw_wait(a.flags + (size_t)(b + 1) * PW_FLAG_STRIDE, a.base + (uint64_t)p + 1, a.err)) s_abort = 1;
      } else if (PEER && pdn) {
        if (!pw_wait_peer(a.pr.myflag_dn, a.base + (uint64_t)p + 1, a.pr.err, a.err)) s_abort = 1;
      }
    }
    __syncthreads();
    PW_STAMP(p, 6);                          // both flags seen
    if (s_abort) return;
    {
      const int gu = r0 - PW_HALO + t, gd = r0 + RB + t;
      double hu = 0.0, hd = 0.0;
      if (b > 0) hu = pw_load_sc1(ycol + gu);
      else if (PEER && pup) hu = pw_load_sys(a.pr.recv_up + (size_t)(a.pr.ebuf + ((p + 1) & 1)) * PW_T + t);
      if (gd < a.nrows) hd = pw_load_sc1(ycol + gd);
      else if (PEER && pdn) hd = pw_load_sys(a.pr.recv_dn + (size_t)(a.pr.ebuf + ((p + 1) & 1)) * PW_T + t);
#ifdef NK_PW_STAMPS
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      PW_STAMP(p, 7);                        // this wavefront's halo rows are in
#endif
      xout[t] = hu;
      xout[PW_HALO + RB + t] = hd;
    }
    __syncthreads();
    PW_STAMP(p, 9);
  }
#ifdef NK_PW_STAMPS
  __syncthreads();
  if (g_pw_stamp != nullptr)
    for (int i = t; i < PW_STP * PW_NST; i += PW_T) g_pw_stamp[(size_t)b * PW_STP * PW_NST + i] = s_pw_st[i];
#endif
}

// The same kernel for matrices that are banded only SEGMENT BY SEGMENT: n = SEG·M rows, segment σ = rows [σM, (σ+1)M) — the
// species of a reaction–diffusion system in the reference's (i, j, species) ordering (docs/src/tutorials/large_systems.md: the
// Brusselator couples u(i, j) with v(i, j), M rows apart) — and / or whose bands close to a RING (periodic boundaries: the first
// grid line couples with the last). A workgroup owns band b OF EVERY SEGMENT (rows σM + b·RB … of each), so the coupling
// between segments never leaves the workgroup; its vector buffer is SEG × [halo above | RB own rows | halo below]; the hand-off
// is the one of k_spmv_powers per segment (one flag per band behind all of them), with band 0's upper neighbour being band
// nb − 1 on a ring. A column c of a row of band b maps to segment c / M, offset (c mod M) − b·RB, shifted by ±M on a ring when
// that brings it into [−1024, RB + 1024) — fixed at load time, the power loop is the plain one. Bit-identical row sums as before.
template <int RPS, int W, int SEG>
__global__ __launch_bounds__(PW_T) void k_spmv_powers_seg(const pw_args a) {
  if (a.d_skip != nullptr && *a.d_skip != 0) return;
  extern __shared__ double pw_x[];
  __shared__ int s_abort;
  constexpr int RB = PW_T * RPS, XS = RB + 2 * PW_HALO, XN = SEG * XS, NSL = SEG * RPS;
  static_assert(W <= 8, "the slices are staged through LDS");
  const int t = threadIdx.x;
  const int b = pw_band_of(blockIdx.x, a.nb);
  const int M = a.seg_len, b0 = b * RB;
  const bool ring = a.ring != 0;
  double *xa = pw_x, *xb = pw_x + XN;

  double v[NSL][W];
  int ci[NSL][W];
  int len[NSL];
  {
    double *sv = pw_x;
    int *sc = reinterpret_cast<int *>(pw_x + (size_t)PW_T * W);
#pragma unroll
    for (int i = 0; i < NSL; ++i) {
      const int sg = i / RPS, j = i % RPS;
      const int lfirst = b0 + PW_T * j;                 // the slice's first row within its segment
      const bool any = lfirst < M;                       // (uniform)
      const bool valid = lfirst + t < M;
      const int rfirst = sg * M + lfirst, r = rfirst + t;
      const int rc = valid ? r : (any ? sg * M + M - 1 : 0);
      const int k0 = a.rowptr[rc], k1 = a.rowptr[rc + 1];
      const int rl = any ? sg * M + (lfirst + PW_T <= M ? lfirst + PW_T : M) : 0;
      const int kb = any ? a.rowptr[rfirst] : 0, ke = any ? a.rowptr[rl] : 0;
      len[i] = valid ? k1 - k0 : 0;
#pragma unroll
      for (int jj = 0; jj < W; ++jj) {
        const int e = t + PW_T * jj;
        if (kb + e < ke) { sv[e] = a.val[kb + e]; sc[e] = a.col[kb + e]; }
      }
      __syncthreads();
#pragma unroll
      for (int jj = 0; jj < W; ++jj) {
        const int e = (jj < len[i]) ? k0 - kb + jj : 0;
        const double vv = sv[e];
        const int c = sc[e];
        const int sgc = SEG > 1 ? c / M : 0;
        int d = c - sgc * M - b0;
        if (ring) d = d < -PW_HALO ? d + M : (d >= RB + PW_HALO ? d - M : d);
        v[i][jj] = (jj < len[i]) ? vv : 0.0;
        ci[i][jj] = (jj < len[i]) ? sgc * XS + PW_HALO + d : sg * XS + PW_HALO + PW_T * j + t;
      }
      __syncthreads();
    }
  }
  for (int idx = t; idx < XN; idx += PW_T) {
    const int sg = idx / XS;
    int l = b0 - PW_HALO + (idx - sg * XS);
    if (ring) l = l < 0 ? l + M : (l >= M ? l - M : l);
    xa[idx] = (l >= 0 && l < M) ? a.x0[sg * M + l] : 0.0;
    xb[idx] = 0.0;
  }
  if (t == 0) s_abort = 0;
  __syncthreads();

  const bool shifted = a.theta != nullptr;
  const int nup = b > 0 ? b - 1 : (ring ? a.nb - 1 : -1), ndn = b + 1 < a.nb ? b + 1 : (ring ? 0 : -1);
  for (int p = 0; p < a.s; ++p) {
    const double *xin = (p & 1) ? xb : xa;
    double *xout = (p & 1) ? xa : xb;
    const double *osp = (p == 0) ? a.scal_first : a.scal_rest;
    const double os = osp ? *osp : 1.0;
    const double th = shifted ? a.theta[p] : 0.0;
    double *ycol = a.Y + (int64_t)p * a.ldy;
    const bool pub = p + 1 < a.s;
#pragma unroll
    for (int i = 0; i < NSL; ++i) {
      const int sg = i / RPS, j = i % RPS;
      const int lrow = b0 + PW_T * j + t, own = sg * XS + PW_HALO + PW_T * j + t;
      double sum = 0.0;
#pragma unroll
      for (int jj = 0; jj < W; ++jj) {
        const double pr = v[i][jj] * xin[ci[i][jj]];
        sum = (jj < len[i]) ? sum + pr : sum;
      }
      double out = shifted ? sum - th * xin[own] : sum;
      out = osp ? os * out : out;
      if (lrow < M) {
        xout[own] = out;
        if (pub && (j == 0 || j == RPS - 1)) pw_store_sc1(ycol + sg * M + lrow, out);   // rows a neighbour band reads
        else ycol[sg * M + lrow] = out;
      }
    }
    if (!pub) break;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0 && !((a.variant & 256) && b == 0))
      __hip_atomic_store(a.flags + (size_t)b * PW_FLAG_STRIDE, a.base + (uint64_t)p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t == 0 && nup >= 0) {
      if (!pw_wait(a.flags + (size_t)nup * PW_FLAG_STRIDE, a.base + (uint64_t)p + 1, a.err)) s_abort = 1;
    }
    if (t == 64 && ndn >= 0) {
      if (!pw_wait(a.flags + (size_t)ndn * PW_FLAG_STRIDE, a.base + (uint64_t)p + 1, a.err)) s_abort = 1;
    }
    __syncthreads();
    if (s_abort) return;
    {
      double hu[SEG], hd[SEG];
      int lu = b0 - PW_HALO + t, ld = b0 + RB + t;
      if (ring) { lu = lu < 0 ? lu + M : lu; ld = ld >= M ? ld - M : ld; }
#pragma unroll
      for (int sg = 0; sg < SEG; ++sg) {
        hu[sg] = (nup >= 0 && lu >= 0 && lu < M) ? pw_load_sc1(ycol + sg * M + lu) : 0.0;
        hd[sg] = (ndn >= 0 && ld < M) ? pw_load_sc1(ycol + sg * M + ld) : 0.0;
      }
#pragma unroll
      for (int sg = 0; sg < SEG; ++sg) {
        xout[sg * XS + t] = hu[sg];
        xout[sg * XS + PW_HALO + RB + t] = hd[sg];
      }
    }
    __syncthreads();
  }
}

// ----------------------------------------------------------------------------- host side
struct nk_powers_plan {
  int rpt = 0, w = 0, nb = 0;
  int seg = 0, ring = 0, seg_len = 0;   // seg > 0: k_spmv_powers_seg<rpt, w, seg> (rpt slices per segment and band)
  uint64_t *d_flags = nullptr;
  uint64_t *h_err = nullptr, *h_err_dev = nullptr;   // pinned, coherent: {time-outs, bound in ticks}
  uint64_t epoch = 0;
  bool broken = false;
  int strikes = 0;   // launches that timed out so far: the plan comes back with the next linear solve, three strikes switch it off for good
  // several ranks: the neighbours' and my receive areas in the peer-mapped arenas, the halo slots' rows next door
  bool peer = false;
  nk_peer_powers pp;
  int32_t *d_halo_vl = nullptr;
};
void nk_powers_plan_destroy(nk_powers_plan *P) {
  if (!P) return;
  hipFree(P->d_flags);
  hipFree(P->d_halo_vl);
  if (P->h_err) hipHostFree(P->h_err);
  delete P;
}
static int pw_variant() {
  static const int v = getenv("NK_PW_VARIANT") ? atoi(getenv("NK_PW_VARIANT")) : 0;
  // development hook: NK_PW_DEBUG_STALL_LAUNCH = n makes band 0 of the process's n-th launch withhold its flag (the time-out path)
  // (a comma-separated list: several torn launches)
  static const std::vector<long> stalls = [] {
    std::vector<long> v;
    const char *e = getenv("NK_PW_DEBUG_STALL_LAUNCH");
    while (e && *e) {
      char *end = nullptr;
      const long x = strtol(e, &end, 10);
      if (end == e) break;
      v.push_back(x);
      e = (*end == ',') ? end + 1 : end;
    }
    return v;
  }();
  static long launches = 0;
  ++launches;
  bool stall = false;
  for (long x : stalls) stall = stall || (x > 0 && launches == x);
  return (v & 255) | (stall ? 256 : 0);
}
static bool pw_enabled() {
  static const bool on = !(getenv("NK_SPMV_POWERS") && atoi(getenv("NK_SPMV_POWERS")) == 0);
  return on;
}

// (the LDS limit is an attribute of the function ON A DEVICE: set on every call — a process-wide "done" flag left a second
//  context on another GPU at the default 64 KB)
template <int RPT, int W, int GEN, bool PEER = false>
static int pw_launch(nk_ctx *ctx, const pw_args &a, bool query, int *occ) {
  constexpr size_t lds_x = (size_t)2 * (PW_T * RPT + 2 * PW_HALO) * sizeof(double), lds_m = (W > 8 || GEN == 1) ? 0 : (size_t)PW_T * W * 12;
  constexpr size_t lds = lds_x > lds_m ? lds_x : lds_m;
  if (lds > 64 * 1024)
    NK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_spmv_powers<RPT, W, GEN, PEER>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  if (query) {
    NK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(occ, k_spmv_powers<RPT, W, GEN, PEER>, PW_T, lds));
    return NK_OK;
  }
  hipEvent_t e0 = nullptr, e1 = nullptr;
  // NK_PW_COOP=1: a cooperative launch — the runtime refuses a grid that cannot be resident on an EMPTY device and keeps cooperative
  // launches of different streams apart; it does not keep ordinary kernels of other streams off the compute units (the time-out
  // stays the safety net). Measured: profiles/r05_*_powers_coop_ab.txt.
  static const bool coop = getenv("NK_PW_COOP") && atoi(getenv("NK_PW_COOP")) != 0;
  if (ctx->prof.on && nk_prof_next(ctx, &e0, &e1))
    hipExtLaunchKernelGGL((k_spmv_powers<RPT, W, GEN, PEER>), dim3(a.nb), dim3(PW_T), lds, ctx->stream, e0, e1, 0, a);
  else if (coop) {
    pw_args ac = a;
    void *args[] = {(void *)&ac};
    NK_HIP(hipLaunchCooperativeKernel(reinterpret_cast<const void *>(&k_spmv_powers<RPT, W, GEN, PEER>), dim3(a.nb), dim3(PW_T), args,
                                      (unsigned int)lds, ctx->stream));
  } else
    hipLaunchKernelGGL((k_spmv_powers<RPT, W, GEN, PEER>), dim3(a.nb), dim3(PW_T), lds, ctx->stream, a);
  NK_HIP(hipGetLastError());
  return NK_OK;
}
static int pw_dispatch(nk_ctx *ctx, int rpt, int w, const pw_args &a, bool query, int *occ, int gen = 0, bool peer = false) {
  if (peer) {   // several ranks (stored matrices)
#define PW_PEER(R, WW) if (rpt == R && w == WW) return pw_launch<R, WW, 0, true>(ctx, a, query, occ)
    PW_PEER(1, 5); PW_PEER(2, 5); PW_PEER(4, 5); PW_PEER(6, 5);
    PW_PEER(1, 8); PW_PEER(2, 8);
    PW_PEER(1, 16);
#undef PW_PEER
    NK_FAIL(NK_E_INVALID, "internal: no matrix-powers kernel for %d rows per thread × %d entries per row", rpt, w);
  }
#define PW_CASE(R, WW, G) if (rpt == R && w == WW && gen == G) return pw_launch<R, WW, G>(ctx, a, query, occ)
  PW_CASE(1, 5, 1); PW_CASE(2, 5, 1); PW_CASE(4, 5, 1); PW_CASE(6, 5, 1);
  PW_CASE(1, 5, 0); PW_CASE(2, 5, 0); PW_CASE(4, 5, 0); PW_CASE(6, 5, 0);
  PW_CASE(1, 8, 0); PW_CASE(2, 8, 0);
  PW_CASE(1, 16, 0);
#undef PW_CASE
  NK_FAIL(NK_E_INVALID, "internal: no matrix-powers kernel for %d rows per thread × %d entries per row", rpt, w);
}

template <int RPS, int W, int SEG>
static int pw_launch_seg(nk_ctx *ctx, const pw_args &a, bool query, int *occ) {
  constexpr size_t lds_x = (size_t)2 * SEG * (PW_T * RPS + 2 * PW_HALO) * sizeof(double), lds_m = (size_t)PW_T * W * 12;
  constexpr size_t lds = lds_x > lds_m ? lds_x : lds_m;
  if (lds > 64 * 1024)
    NK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_spmv_powers_seg<RPS, W, SEG>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)lds));
  if (query) {
    NK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(occ, k_spmv_powers_seg<RPS, W, SEG>, PW_T, lds));
    return NK_OK;
  }
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (ctx->prof.on && nk_prof_next(ctx, &e0, &e1))
    hipExtLaunchKernelGGL((k_spmv_powers_seg<RPS, W, SEG>), dim3(a.nb), dim3(PW_T), lds, ctx->stream, e0, e1, 0, a);
  else
    hipLaunchKernelGGL((k_spmv_powers_seg<RPS, W, SEG>), dim3(a.nb), dim3(PW_T), lds, ctx->stream, a);
  NK_HIP(hipGetLastError());
  return NK_OK;
}
// (slices per segment, slots per row, segments) the segmented kernel is compiled for: SEG·RPS·W ≤ 20 register-resident slots
static bool pw_seg_shape(int rps, int w, int seg) {
  if (seg == 1) return (w == 5 && (rps == 1 || rps == 2 || rps == 4)) || (w == 8 && (rps == 1 || rps == 2));
  if (seg == 2) return (w == 5 && (rps == 1 || rps == 2)) || (w == 8 && rps == 1);
  return false;
}
static int pw_dispatch_seg(nk_ctx *ctx, int rps, int w, int seg, const pw_args &a, bool query, int *occ) {
#define PW_SEG(R, WW, SG) if (rps == R && w == WW && seg == SG) return pw_launch_seg<R, WW, SG>(ctx, a, query, occ)
  PW_SEG(1, 5, 1); PW_SEG(2, 5, 1); PW_SEG(4, 5, 1); PW_SEG(1, 8, 1); PW_SEG(2, 8, 1);
  PW_SEG(1, 5, 2); PW_SEG(2, 5, 2); PW_SEG(1, 8, 2);
#undef PW_SEG
  NK_FAIL(NK_E_INVALID, "internal: no segmented matrix-powers kernel for %d slices × %d slots × %d segments", rps, w, seg);
}

static int pw_plan_new(nk_ctx *ctx, int rpt, int w, int nb, nk_powers_plan **out) {
  nk_powers_plan *P = new nk_powers_plan();
  auto guard = nk_make_guard(P, [](nk_powers_plan *p) { nk_powers_plan_destroy(p); });
  P->rpt = rpt; P->w = w; P->nb = nb;
  NK_TRY(nk_dev_alloc(&P->d_flags, (size_t)nb * PW_FLAG_STRIDE));
  NK_HIP(nk_memset(ctx, P->d_flags, 0, (size_t)nb * PW_FLAG_STRIDE * sizeof(uint64_t)));
  NK_HIP(hipHostMalloc((void **)&P->h_err, 2 * sizeof(uint64_t), hipHostMallocMapped | hipHostMallocCoherent));
  NK_HIP(hipHostGetDevicePointer((void **)&P->h_err_dev, P->h_err, 0));
  P->h_err[0] = 0;
  {
    const char *e = getenv("NK_PW_TIMEOUT_MS");
    const double ms = e ? atof(e) : 250.0;
    P->h_err[1] = (uint64_t)((ms > 1.0 ? ms : 1.0) * 1.0e5);   // 100 MHz wall clock
  }
  *out = guard.release();
  return NK_OK;
}

// Which layout of the resident kernel a pattern fits — host arithmetic only (no device): candidates in the order plain bands, one
// segment on a ring, two segments, two segments on rings. `skip` candidates have been rejected by the caller (occupancy).
struct pw_layout {
  int kind = 0;   // 0 none, 1 plain bands (k_spmv_powers), 2 segments / ring (k_spmv_powers_seg)
  int rpt = 0, w = 0, nb = 0, seg = 0, ring = 0;
  int64_t M = 0;
};
static pw_layout pw_find_layout(int64_t n, const int32_t *rowptr, const int32_t *col, int num_cus, int skip) {
  pw_layout L;
  if (n < 1 || num_cus < 1) return L;
  int maxlen = 0;
  for (int64_t r = 0; r < n; ++r) maxlen = std::max(maxlen, (int)(rowptr[r + 1] - rowptr[r]));
  const int w = maxlen <= 5 ? 5 : (maxlen <= 8 ? 8 : (maxlen <= 16 ? 16 : 0));
  if (!w) return L;
  for (int lay = skip; lay < 4; ++lay) {
    if (lay == 0) {   // plain bands of 1024·rpt rows, every column of a band within one slice of it
      const int64_t per_cu = (n + num_cus - 1) / num_cus;
      const int need = (int)((per_cu + PW_T - 1) / PW_T);
      const int rpt = need <= 1 ? 1 : (need <= 2 ? 2 : (need <= 4 ? 4 : (need <= 6 ? 6 : 0)));
      if (!rpt) continue;
      if ((w == 8 && rpt > 2) || (w == 16 && rpt > 1)) continue;   // register budget (128 VGPRs at 1024 threads): RPT·W ≤ 30 slots
      const int64_t rb = (int64_t)PW_T * rpt;
      bool ok = true;
      for (int64_t r = 0; r < n && ok; ++r) {
        const int64_t b0 = (r / rb) * rb, lo = b0 - PW_HALO, hi = b0 + rb + PW_HALO;
        for (int32_t k = rowptr[r]; k < rowptr[r + 1]; ++k)
          if (col[k] < lo || col[k] >= hi) { ok = false; break; }
      }
      if (!ok) continue;
      L.kind = 1; L.rpt = rpt; L.w = w; L.nb = (int)((n + rb - 1) / rb); L.seg = 1; L.ring = 0; L.M = n;
      return L;
    }
    // banded segment by segment and / or on a ring (k_spmv_powers_seg)
    const int seg = lay == 1 ? 1 : 2, ring = lay == 2 ? 0 : 1;
    if (n % seg) continue;
    const int64_t M = n / seg, pc = (M + num_cus - 1) / num_cus;
    const int nd = (int)((pc + PW_T - 1) / PW_T);
    const int rps = nd <= 1 ? 1 : (nd <= 2 ? 2 : (nd <= 4 ? 4 : 0));
    if (!rps || !pw_seg_shape(rps, w, seg)) continue;
    const int64_t RB = (int64_t)PW_T * rps;
    if (ring && (M % RB)) continue;                 // (a ragged last band has no well-defined ring neighbour rows)
    bool ok = true;
    for (int64_t r = 0; r < n && ok; ++r) {
      const int64_t b0 = ((r % M) / RB) * RB;
      for (int32_t k = rowptr[r]; k < rowptr[r + 1]; ++k) {
        if (col[k] < 0 || col[k] >= n) { ok = false; break; }
        int64_t d = (int64_t)col[k] % M - b0;
        if (ring) d = d < -PW_HALO ? d + M : (d >= RB + PW_HALO ? d - M : d);
        if (d < -PW_HALO || d >= RB + PW_HALO) { ok = false; break; }
      }
    }
    if (!ok) continue;
    L.kind = 2; L.rpt = rps; L.w = w; L.nb = (int)((M + RB - 1) / RB); L.seg = seg; L.ring = ring; L.M = M;
    return L;
  }
  return L;
}
static int pw_layout_index(const pw_layout &L) { return L.kind == 1 ? 0 : (L.seg == 1 ? 1 : (L.ring ? 3 : 2)); }
extern "C" int nk_csr_powers_layout(int64_t nrows, const int32_t *rowptr, const int32_t *col, int num_cus, int layout[6]) {
  NK_REQUIRE(rowptr && col && layout && nrows >= 0 && num_cus >= 1, "bad argument");
  const pw_layout L = pw_find_layout(nrows, rowptr, col, num_cus, 0);
  layout[0] = L.kind; layout[1] = L.rpt; layout[2] = L.w; layout[3] = L.nb; layout[4] = L.seg; layout[5] = L.ring;
  return NK_OK;
}

// Builds (once per pattern) the plan of the resident matrix-powers kernel; A->pw stays NULL when the matrix is not eligible.
// Several ranks (COLLECTIVE — every rank reaches this with its share of the same matrix): the plain band layout on every rank,
// whole bands, and every halo column within one slice of the rank's first / last row — i.e. owned by the rank next door, among
// its last / first 1024 rows. The ranks agree (nk_peer_powers_setup); a single "no" keeps the streaming kernel everywhere.
static int pw_plan_ranks(nk_csr *A) {
  nk_ctx *ctx = A->ctx;
  const int64_t n = A->nrows;
  const int nh = (int)A->halo_gcols.size();
  std::vector<int32_t> hvl(nh > 0 ? nh : 1, 0), vcol;
  // (ranks sharing a device: each rank's kernel wants every CU while it waits for the other's — not there; nk_ctx.hip)
  bool ok = pw_enabled() && ctx->peer.on && !ctx->device_shared && n >= PW_T && A->nnz >= 1;
  static const bool ranks_off = getenv("NK_PW_RANKS") && atoi(getenv("NK_PW_RANKS")) == 0;   // A/B switch
  ok = ok && !ranks_off;
  for (int h = 0; h < nh && ok; ++h) {
    const int64_t vl = A->halo_gcols[h] - A->row_begin;
    ok = (vl < 0 && vl >= -PW_HALO && ctx->rank > 0) || (vl >= n && vl < n + PW_HALO && ctx->rank + 1 < ctx->nranks);
    hvl[h] = (int32_t)vl;
  }
  pw_layout L;
  if (ok) {
    vcol.resize(A->h_col.size());
    for (size_t k = 0; k < vcol.size(); ++k) vcol[k] = A->h_col[k] < n ? A->h_col[k] : hvl[A->h_col[k] - n];
    L = pw_find_layout(n, A->h_rowptr.data(), vcol.data(), ctx->num_cus, 0);
    ok = L.kind == 1 && n % ((int64_t)PW_T * L.rpt) == 0;
    if (ok) {
      pw_args probe{};
      int occ = 0;
      NK_TRY(pw_dispatch(ctx, L.rpt, L.w, probe, true, &occ, 0, true));
      ok = occ >= 1 && L.nb <= ctx->num_cus * occ;
    }
  }
  nk_peer_powers pp;
  bool all = false;
  NK_TRY(nk_peer_powers_setup(ctx, ok, &pp, &all));
  if (!all) return NK_OK;
  NK_TRY(pw_plan_new(A->ctx, L.rpt, L.w, L.nb, &A->pw));
  A->pw->peer = true;
  A->pw->pp = pp;
  NK_HIP(hipMalloc((void **)&A->pw->d_halo_vl, hvl.size() * sizeof(int32_t)));
  NK_HIP(nk_memcpy(ctx, A->pw->d_halo_vl, hvl.data(), hvl.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  return NK_OK;
}
static int pw_plan(nk_csr *A) {
  if (A->pw_tried) return NK_OK;
  A->pw_tried = true;
  nk_ctx *ctx = A->ctx;
  if (ctx->nranks > 1 && A->local_only) return NK_OK;   // (a rank-local helper matrix: no collective decision to take part in)
  if (ctx->nranks > 1) return pw_plan_ranks(A);
  if (!pw_enabled() || !A->halo_gcols.empty() || A->nrows < 1 || A->nnz < 1) return NK_OK;
  pw_args probe{};
  for (int skip = 0; skip < 4;) {
    const pw_layout L = pw_find_layout(A->nrows, A->h_rowptr.data(), A->h_col.data(), ctx->num_cus, skip);
    if (!L.kind) return NK_OK;
    int occ = 0;   // all workgroups must be resident at once
    if (L.kind == 1) NK_TRY(pw_dispatch(ctx, L.rpt, L.w, probe, true, &occ));
    else NK_TRY(pw_dispatch_seg(ctx, L.rpt, L.w, L.seg, probe, true, &occ));
    if (occ >= 1 && L.nb <= ctx->num_cus * occ) {
      NK_TRY(pw_plan_new(A->ctx, L.rpt, L.w, L.nb, &A->pw));
      if (L.kind == 2) { A->pw->seg = L.seg; A->pw->ring = L.ring; A->pw->seg_len = (int)L.M; }
      return NK_OK;
    }
    skip = pw_layout_index(L) + 1;
  }
  return NK_OK;
}
bool nk_csr_powers_ready(nk_csr *A) {
  if (!A->pw_tried && pw_plan(A) != NK_OK) return false;
  return A->pw != nullptr && !A->pw->broken && A->pw->h_err[0] == 0;
}
// a time-out of an earlier launch (workgroups not resident together: another stream or process held compute units): the plan is
// parked — NK_E_HIP once, the caller reruns what it can on the streaming kernel
int nk_csr_powers_check(nk_csr *A) {
  if (!A->pw || A->pw->broken || A->pw->h_err[0] == 0) return NK_OK;
  A->pw->broken = true;
  A->pw->strikes++;
  NK_FAIL(NK_E_HIP, "resident matrix-powers kernel: a workgroup waited longer than NK_PW_TIMEOUT_MS for its neighbour band "
                    "(NK_SPMV_POWERS=0 keeps the streaming SpMV)");
}
// … and comes back with the next linear solve (what held the compute units is usually gone by then): the error word is cleared,
// the flags need nothing (they are monotone: the next launch's base lies above everything a torn launch left behind). After three
// time-outs the matrix keeps the streaming kernel. One rank only: on several ranks every rank would have to take the same decision.
static void pw_rearm(nk_powers_plan *P) {
  if (!P || !P->broken || P->peer || P->strikes >= 3) return;
  P->h_err[0] = 0;
  P->broken = false;
}
void nk_csr_powers_rearm(nk_csr *A) { if (A) pw_rearm(A->pw); }
void nk_problem_powers_rearm(nk_problem *P) { if (P) pw_rearm(P->pw); }
int nk_csr_powers_dev(nk_csr *A, const double *d_x0, double *d_Y, int64_t ldy, int s, const double *d_scal_first,
                      const double *d_scal_rest, const double *d_theta, const int *d_skip) {
  NK_REQUIRE(nk_csr_powers_ready(A), "internal: matrix powers on a matrix without a plan");
  NK_REQUIRE(s >= 1 && s <= 200, "matrix powers: 1 ≤ s ≤ 200");
  nk_ctx *ctx = A->ctx;
  nk_powers_plan *P = A->pw;
  pw_args a{};
  a.nrows = (int)A->nrows; a.nb = P->nb; a.s = s; a.variant = pw_variant();
  a.rowptr = A->d_rowptr; a.col = A->d_col; a.val = A->d_val;
  a.x0 = d_x0; a.Y = d_Y; a.ldy = ldy;
  a.scal_first = d_scal_first; a.scal_rest = d_scal_rest; a.theta = d_theta; a.d_skip = d_skip;
  a.flags = P->d_flags; a.base = (++P->epoch) << 8; a.err = P->h_err_dev;
  ctx->stats.op_applies += s;
  nk_prof_scope prof_(ctx, NK_K_POWERS,
                      (double)s * (12.0 * (double)A->nnz + 4.0 * (double)(A->nrows + 1) + 16.0 * (double)A->nrows));
  if (P->seg > 0) {
    a.seg_len = P->seg_len; a.ring = P->ring;
    return pw_dispatch_seg(ctx, P->rpt, P->w, P->seg, a, false, nullptr);
  }
  if (P->peer) {
    a.pr.up = ctx->rank > 0; a.pr.dn = ctx->rank + 1 < ctx->nranks;
    a.pr.halo_vl = P->d_halo_vl;
    a.pr.push_up = P->pp.push_up; a.pr.push_dn = P->pp.push_dn; a.pr.flag_up = P->pp.flag_up; a.pr.flag_dn = P->pp.flag_dn;
    a.pr.recv_up = P->pp.recv_up; a.pr.recv_dn = P->pp.recv_dn; a.pr.myflag_up = P->pp.myflag_up; a.pr.myflag_dn = P->pp.myflag_dn;
    a.pr.err = nk_peer_err_ptr(ctx);
    a.pr.ebuf = (int)(P->epoch & 1) * 2;
    ctx->stats.halo_exchanges++;   // (ONE exchange protocol per launch: s − 1 hand-offs inside it)
    return pw_dispatch(ctx, P->rpt, P->w, a, false, nullptr, 0, true);
  }
  return pw_dispatch(ctx, P->rpt, P->w, a, false, nullptr);
}

// ---- the matrix-free Bratu operator (J = c_lap·Δ_h − diag(c_exp·exp u), nk_problems.hip): the same kernel with the band's rows
// generated instead of loaded. One rank, the whole grid on it, grid side ≤ 1024 (the halo is one grid line).
static int pw_problem_plan(nk_problem *P) {
  if (P->pw_tried) return NK_OK;
  P->pw_tried = true;
  nk_ctx *ctx = P->ctx;
  if (!pw_enabled() || P->kind != NK_PROBLEM_BRATU2D || ctx->nranks != 1 || P->replicated || P->j0 != 0 || P->j1 != P->ns ||
      P->ns < 2 || P->ns > PW_HALO || P->n_local != P->ns * P->ns)
    return NK_OK;
  const int64_t n = P->n_local, per_cu = (n + ctx->num_cus - 1) / ctx->num_cus;
  const int need = (int)((per_cu + PW_T - 1) / PW_T);
  const int rpt = need <= 1 ? 1 : (need <= 2 ? 2 : (need <= 4 ? 4 : (need <= 6 ? 6 : 0)));
  if (!rpt) return NK_OK;
  const int nb = (int)((n + (int64_t)PW_T * rpt - 1) / ((int64_t)PW_T * rpt));
  pw_args probe{};
  int occ = 0;
  NK_TRY(pw_dispatch(ctx, rpt, 5, probe, true, &occ, 1));
  if (occ < 1 || nb > ctx->num_cus * occ) return NK_OK;
  return pw_plan_new(P->ctx, rpt, 5, nb, &P->pw);
}
bool nk_problem_powers_ready(nk_problem *P) {
  if (!P->pw_tried && pw_problem_plan(P) != NK_OK) return false;
  return P->pw != nullptr && !P->pw->broken && P->pw->h_err[0] == 0;
}
int nk_problem_powers_check(nk_problem *P) {
  if (!P->pw || P->pw->broken || P->pw->h_err[0] == 0) return NK_OK;
  P->pw->broken = true;
  P->pw->strikes++;
  NK_FAIL(NK_E_HIP, "resident matrix-powers kernel (matrix-free): a workgroup waited longer than NK_PW_TIMEOUT_MS for its "
                    "neighbour band (NK_SPMV_POWERS=0 keeps the per-column JVP)");
}
int nk_problem_powers_dev(nk_problem *P, const double *d_u, const double *d_x0, double *d_Y, int64_t ldy, int s,
                          const double *d_scal_first, const double *d_scal_rest, const double *d_theta, const int *d_skip) {
  NK_REQUIRE(nk_problem_powers_ready(P), "internal: matrix powers on a problem without a plan");
  nk_ctx *ctx = P->ctx;
  if (P->d_u_lin != d_u || !P->d_diag) NK_TRY(nk_problem_jvp_prepare(P, d_u));   // d = c_exp·exp(u) at the linearisation point
  nk_powers_plan *Q = P->pw;
  pw_args a{};
  a.nrows = (int)P->n_local; a.nb = Q->nb; a.s = s; a.variant = pw_variant();
  a.x0 = d_x0; a.Y = d_Y; a.ldy = ldy;
  a.scal_first = d_scal_first; a.scal_rest = d_scal_rest; a.theta = d_theta; a.d_skip = d_skip;
  a.flags = Q->d_flags; a.base = (++Q->epoch) << 8; a.err = Q->h_err_dev;
  a.ns = (int)P->ns; a.c_lap = P->c_lap; a.diag = P->d_diag;
  ctx->stats.op_applies += s;
  nk_prof_scope prof_(ctx, NK_K_POWERS, (double)s * 24.0 * (double)P->n_local);
  return pw_dispatch(ctx, Q->rpt, 5, a, false, nullptr, 1);
}

// Y[:, p] = scale·(A − θ_p I) Y[:, p−1], Y[:, −1] = x (θ = NULL: plain powers). The resident kernel where the matrix is
// eligible (*resident = 1), s streaming launches otherwise — bit-identical either way.
extern "C" int nk_csr_powers(nk_csr *A, const double *x, double *Y, int64_t ldy, int s, const double *theta, double scale,
                             int memspace, int *resident) {
  NK_REQUIRE(A && x && Y && s >= 1 && s <= 64 && ldy >= A->nrows, "bad argument");
  nk_ctx *ctx = A->ctx;
  NK_HIP(hipSetDevice(ctx->device));
  const int64_t n = A->nrows;
  double *dx = nullptr, *dY = nullptr, *dsc = nullptr;
  auto cleanup = [&]() { if (memspace != NK_DEVICE) { hipFree(dx); hipFree(dY); } hipFree(dsc); };
  if (memspace == NK_DEVICE) { dx = const_cast<double *>(x); dY = Y; }
  else {
    NK_TRY(nk_dev_alloc(&dx, (size_t)n));
    if (nk_dev_alloc(&dY, (size_t)ldy * s) != NK_OK) { cleanup(); return NK_E_NOMEM; }
    NK_HIP(nk_memcpy(ctx, dx, x, n * sizeof(double), hipMemcpyHostToDevice));
  }
  if (nk_dev_alloc(&dsc, (size_t)s + 1) != NK_OK) { cleanup(); return NK_E_NOMEM; }
  std::vector<double> hs((size_t)s + 1, 0.0);
  hs[0] = scale;
  if (theta) for (int p = 0; p < s; ++p) hs[1 + p] = theta[p];
  nk_memcpy(ctx, dsc, hs.data(), hs.size() * sizeof(double), hipMemcpyHostToDevice);
  nk_csr_powers_rearm(A);
  const bool res = nk_csr_powers_ready(A);
  bool res_used = res;
  int rc = NK_OK;
  auto streaming = [&]() -> int {
    int r = NK_OK;
    for (int p = 0; p < s && r == NK_OK; ++p) {
      nk_spmv_epi ep;
      if (theta) { ep.mode = 3; ep.theta = dsc + 1 + p; }
      r = nk_csr_spmv_dev(A, p == 0 ? dx : dY + (size_t)(p - 1) * ldy, dY + (size_t)p * ldy, nullptr, dsc, &ep);
    }
    return r;
  };
  rc = res ? nk_csr_powers_dev(A, dx, dY, ldy, s, dsc, dsc, theta ? dsc + 1 : nullptr, nullptr) : streaming();
  if (rc == NK_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) { nk_set_error("stream error in nk_csr_powers"); rc = NK_E_HIP; }
  if (rc == NK_OK && res && nk_csr_powers_check(A) != NK_OK) {
    // the resident launch timed out (its workgroups were not on the chip together): the columns are garbage — the s streaming
    // launches produce the same bits (one rank; on several ranks the others are in the same launch and time out as well)
    res_used = false;
    rc = streaming();
    if (rc == NK_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) { nk_set_error("stream error in nk_csr_powers"); rc = NK_E_HIP; }
  }
  if (rc == NK_OK && memspace != NK_DEVICE)
    if (nk_memcpy(ctx, Y, dY, (size_t)ldy * s * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) rc = NK_E_HIP;
  cleanup();
  if (resident) *resident = res_used ? 1 : 0;
  return rc;
}

// development (builds with -DNK_PW_STAMPS): the bands' phase stamps of the LAST launch → out[nb][PW_STP][PW_NST]
extern "C" int nk_pw_debug_stamps(int enable, unsigned long long *out, int nbands) {
  static unsigned long long *d_st = nullptr;
  const size_t words = (size_t)1024 * PW_STP * PW_NST;
  if (enable && !d_st) {
    if (hipMalloc(&d_st, words * sizeof(unsigned long long)) != hipSuccess) return NK_E_NOMEM;
    hipMemset(d_st, 0, words * sizeof(unsigned long long));   // (development hook: the caller synchronises the device)
    hipDeviceSynchronize();
    hipMemcpyToSymbol(HIP_SYMBOL(g_pw_stamp), &d_st, sizeof(d_st));
  }
  if (!enable && d_st) {
    unsigned long long *z = nullptr;
    hipMemcpyToSymbol(HIP_SYMBOL(g_pw_stamp), &z, sizeof(z));
  }
  if (out && d_st && nbands > 0 && nbands <= 1024) {
    hipDeviceSynchronize();
    hipMemcpy(out, d_st, (size_t)nbands * PW_STP * PW_NST * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  }
  return NK_OK;
}
